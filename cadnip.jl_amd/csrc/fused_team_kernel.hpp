// fused_team_kernel.hpp -- the fused Newton kernel for FEW instances: a TEAM of NW waves per sweep instance (k_fteam<NW>).
//
// k_fused2 (fused2_kernel.hpp) gives every instance one wave and gets its throughput from thousands of instances in flight; a single
// transient (BASELINE.json configs 2 and 3) or a small sweep (config 4: 128 points per GPU) leaves the chip empty and pays the
// latency of one wave's serial instruction stream: 14-15 us per Newton iteration on the DFF, twice what one CPU core needs.  Here
// one workgroup of NW waves (one per SIMD of a CU) owns one instance, and the independent parts of a Newton round run side by side:
//   * stamping: the three independent parts of the sp_mos1 load section (junction currents | depletion charges | channel current,
//     devices.hpp: stamp_mos1_team) on three waves, all other device blocks (pinned R / C block, sources, ...) on the fourth;
//     every contribution is an LDS atomic into the shared work array W, exactly as in k_fused2;
//   * linear solve: the entry program's passes of one dependency level are dealt to the waves, a workgroup barrier closes the level
//     (as k_lu_f2_mw, lu_f2.hip, but with k_fused2's software pipeline on every wave's own step list); the dense core is wave 0's;
//   * update: unknown i belongs to thread i (history / predictor / error weights in registers, one element per thread for n <= 256),
//     the norms are summed per wave with DPP and across waves through LDS; every wave takes the controller's decisions itself from
//     the same sums (tran_ctrl.hpp: the StepState is wave-uniform and identical in all waves of the team) -- nothing is broadcast;
//   * the sp_mos1 parameter rows of the resident instance are staged in LDS once per residence (k_fused2 reads them from L2 in
//     every round -- two of them, the junction's critical voltages, inside the limiting code's branches).
// Same tables, same argument block, same program, same Newton modes as k_fused2 (transient, direct residuals, lean device set); the
// host picks the kernel by batch size (fused2.hip: team_waves).  Summation order inside a matrix word differs from k_fused2 (three
// waves add into it), so the two agree to rounding.
#pragma once
#include "fused2_kernel.hpp"

namespace cadnip {

// controller vector policy (tran_ctrl.hpp) of a team: thread t owns the unknowns t, t + NT, ...; the first KPF of them live in registers
template <int NT_, int KPF_>
struct TeamVecs {
  static constexpr int KPF = KPF_;
  static constexpr int NT = NT_;
  static constexpr bool OWN_REDUCE = true;
  double *us, *betas; const double* W; const u16* qinv;
  double* red;                 // [NT / 64][4] exchange words of the update's reductions
  int qi[KPF];                 // W word of this thread's unknowns' Newton step (a property of the circuit)
  size_t vo;
  const double* lw;
  double r_u0[KPF], r_u1[KPF], r_u2[KPF], r_up[KPF], pf_at[KPF], pf_em[KPF];
  __device__ __forceinline__ double* p_u0() const { return kargs()->t.u0 + vo; }
  __device__ __forceinline__ double* p_u1() const { return kargs()->t.u1 + vo; }
  __device__ __forceinline__ double* p_u2() const { return kargs()->t.u2 + vo; }
  __device__ __forceinline__ double* p_up() const { return kargs()->t.up + vo; }
  __device__ __forceinline__ double mem_u0(int i) const { return p_u0()[i]; }
  __device__ __forceinline__ double mem_u1(int i) const { return p_u1()[i]; }
  __device__ __forceinline__ void load_history(int n, int tid) {
    const double *u0 = p_u0(), *u1 = p_u1(), *u2 = p_u2(), *up = p_up();
#pragma unroll
    for (int k = 0; k < KPF; ++k) {
      const int i = tid + NT * k < n ? tid + NT * k : 0;
      r_u0[k] = u0[i]; r_u1[k] = u1[i]; r_u2[k] = u2[i]; r_up[k] = up[i];
    }
  }
  __device__ __forceinline__ void store_history(int n, int tid) const {
    double *u0 = p_u0(), *u1 = p_u1(), *u2 = p_u2(), *up = p_up();
#pragma unroll
    for (int k = 0; k < KPF; ++k) {
      const int i = tid + NT * k;
      if (i < n) { u0[i] = r_u0[k]; u1[i] = r_u1[k]; u2[i] = r_u2[k]; up[i] = r_up[k]; }
    }
  }
  __device__ __forceinline__ void history_to_memory(int n, int tid) const {
    double *u0 = p_u0(), *u1 = p_u1();
#pragma unroll
    for (int k = 0; k < KPF; ++k) { const int i = tid + NT * k; if (i < n) { u0[i] = r_u0[k]; u1[i] = r_u1[k]; } }
    __syncthreads();     // read back by threads of other waves (save_outputs); release / acquire at workgroup scope
  }
  // error weights: per unknown, the same for every instance -- read once per launch
  __device__ __forceinline__ void load_weights(const TranArgs& a, int tid) {
#pragma unroll
    for (int k = 0; k < KPF; ++k) {
      const int i = tid + NT * k < a.n ? tid + NT * k : 0;
      pf_at[k] = a.atol[i]; pf_em[k] = a.emask[i]; qi[k] = qinv[i];
    }
  }
  __device__ __forceinline__ double get_delta(int i, int k) const { return W[k >= 0 ? qi[k] : (int)qinv[i]]; }
  // Behind the barrier of the update's reductions the right-hand side and the trash words are dead: clear them here for the next round
  // (both kinds of round need that; a refactoring round clears the matrix words on top, k_fteam: begin_round), so that the end of a round
  // on kept factors is ONE barrier.
  int z0, z1;                  // the words [z0, z1) = rhs | trash of the work array
  __device__ __forceinline__ void step_consumed(int tid) const {
    double* Wm = const_cast<double*>(W);
    for (int i = z0 + tid; i < z1; i += NT) Wm[i] = 0.0;
  }
  // the update's two norms and its failure flag over the team: DPP sums per wave, one LDS exchange, ONE barrier; every thread adds the
  // waves' words in the same order.  (Letting every lane add into its wave's word with an LDS atomic looked cheaper -- 20 instructions
  // instead of 110 -- and was slower: 64 same-address fp64 atomics serialise, +1.1 k cycles per round.)
  __device__ __forceinline__ void reduce3(double& s1, double& s2, int& bad) const {
    const double p1 = wave_sum(s1), p2 = wave_sum(s2);
    const int pb = wave_any(bad);
    double* mine = red + 4 * (threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0) { mine[0] = p1; mine[1] = p2; mine[2] = pb ? 1.0 : 0.0; }
    __syncthreads();
    double t1 = 0.0, t2 = 0.0, tb = 0.0;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) { t1 += red[4 * k]; t2 += red[4 * k + 1]; tb += red[4 * k + 2]; }
    s1 = t1; s2 = t2; bad = tb != 0.0;
  }
  __device__ __forceinline__ double get_u(int i) const { return us[i]; }
  __device__ __forceinline__ void set_u(int i, double v) const { us[i] = v; }
  __device__ __forceinline__ double get_beta(int i) const { return betas[i]; }
  __device__ __forceinline__ void set_beta(int i, double v) const { betas[i] = v; }
  __device__ __forceinline__ void set_du(int, double) const {}
  __device__ __forceinline__ double get_lw(int i) const { return lw[i]; }
  __device__ __forceinline__ double h0(int i, int k) const { return k >= 0 ? r_u0[k] : p_u0()[i]; }
  __device__ __forceinline__ double h1(int i, int k) const { return k >= 0 ? r_u1[k] : p_u1()[i]; }
  __device__ __forceinline__ double h2(int i, int k) const { return k >= 0 ? r_u2[k] : p_u2()[i]; }
  __device__ __forceinline__ double hp(int i, int k) const { return k >= 0 ? r_up[k] : p_up()[i]; }
  __device__ __forceinline__ double h3(int i, int) const { return (kargs()->t.u3 + vo)[i]; }          // (max_order = 3: the fourth history point stays in memory)
  __device__ __forceinline__ void set_h3(int i, int, double v) { (kargs()->t.u3 + vo)[i] = v; }
  __device__ __forceinline__ void set_h0(int i, int k, double v) { if (k >= 0) r_u0[k] = v; else p_u0()[i] = v; }
  __device__ __forceinline__ void set_h1(int i, int k, double v) { if (k >= 0) r_u1[k] = v; else p_u1()[i] = v; }
  __device__ __forceinline__ void set_h2(int i, int k, double v) { if (k >= 0) r_u2[k] = v; else p_u2()[i] = v; }
  __device__ __forceinline__ void set_hp(int i, int k, double v) { if (k >= 0) r_up[k] = v; else p_up()[i] = v; }
  __device__ __forceinline__ double atol_of(const TranArgs& a, int i, int k) const { return k >= 0 ? pf_at[k] : a.atol[i]; }
  __device__ __forceinline__ double emask_of(const TranArgs& a, int i, int k) const { return k >= 0 ? pf_em[k] : a.emask[i]; }
};

// STEP: ONE Newton iteration of the DAE form for cadnip_newton_step_fused (api.hip) instead of a transient: u, du, gamma, t come from the
// caller (TranArgs: u, du, gamma, tcur), the residual C du + G u - b and its norm go to f.step_resid / f.step_norm, the Newton step J^-1 resid
// to TranArgs::delta; f.step_refresh says whether J = G + gamma C is refactored (and kept in f.lufac) or the kept factors solve.
template <int NW, bool STEP>
__global__ void __launch_bounds__(64 * NW) k_fteam(F2Args f) {
  static_assert(NW == 2 || NW == 4, "teams of two or four waves");
  constexpr int NT = 64 * NW;
  constexpr int KPF = 2;                                   // unknowns per thread held in registers (n <= 2 NT; beyond: HBM)
  extern __shared__ double sm[];
  __shared__ int s_next;
  const int tid = threadIdx.x, lane0 = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), n = f.n;
  {
    const uint2* src = (const uint2*)(f.tab + f.tab_lo);      // the lean range of the tables: permutations, stamp tables, node tables
    uint2* dst = (uint2*)sm;
    for (int i = tid; i < f.tab_len / 2; i += NT) dst[i] = src[i];
  }
  __syncthreads();
  const TranArgs& a = f.t;
  const unsigned* tab = (const unsigned*)sm;
  const int tab_dbl = f.tab_len / 2;
  const int nW = f.nnz_lu + n + F2_TRASH;                 // LU | rhs | trash
  double* W = sm + tab_dbl;
  if (tid == 0) { W[nW] = 0.0; W[nW + 1] = 1.0; }         // the linear solve's constant words, directly behind the trash words (f2_build_team); never cleared
  double* red = W + nW + 2;                               // [NW][4] exchange words of the update's reductions
  double* us = red + 4 * NW;
  double* betas = us + n;
  double* parc = betas + n;                               // sp_mos1 parameter rows of the resident instance (F2Block::lds_par)
  const int par_words = f.par_words;
  uint4* tdesc = (uint4*)(parc + ((par_words + 1) & ~1));  // step descriptors of the linear solve (f2_program.cpp: f2_build_steps, 16 bytes per lane and step)
  for (int i = tid; i < f.team_desc_len; i += NT) ((u64*)tdesc)[i] = f.team_desc[i];
  // Reproducible sums.  Four waves adding into one word with LDS atomics would do so in the order in which they happen to arrive, and a
  // floating-point sum depends on that order: the last bits of a transient would change from run to run.  So only wave 0 accumulates into
  // W itself; every other wave has a private copy of the work array (matrix words and right-hand side), and after the stamping barrier the
  // copies are added to W in wave order (and cleared for the next round by the thread that reads them).  Within a wave the atomics of one
  // instruction are applied in lane order and the instructions in program order (as in k_fused2), so every word's sum is a fixed sequence.
  double* const WP = (double*)((u64*)tdesc + f.team_desc_len);  // [NW - 1][nW]
  for (int i = tid; i < (NW - 1) * nW; i += NT) WP[i] = 0.0;
  double* const Wacc = w == 0 ? W : WP + (size_t)(w - 1) * nW;

  const int tlo = f.tab_lo;
  const u16* gpos = (const u16*)(tab + (f.off[S_GPOS] - tlo));
  const u64* cdesc = (const u64*)(tab + (f.off[S_CDESC] - tlo));
  const u16* brow = (const u16*)(tab + (f.off[S_BROW] - tlo));
  const u16* qinv = (const u16*)(tab + (f.off[S_QINV] - tlo));
  const short* nodes = (const short*)(tab + (f.off[S_NODES] - tlo));
  const u16* rowof = (const u16*)(tab + (f.off[S_ROWOF] - tlo));
  // what this wave stamps: a part of every sp_mos1 device, or all the other device blocks
  const int roles = NW == 4 ? (w == 0 ? M1_ROLE_J : w == 1 ? M1_ROLE_Q : w == 2 ? M1_ROLE_CH : 0) : (w == 0 ? (M1_ROLE_J | M1_ROLE_CH) : M1_ROLE_Q);
  const bool others = w == NW - 1;
  // pinned stamp targets of the first capacitor / resistor block and the first independent-source block (as k_fused2)
  unsigned rc_gp[2][2] = {{0, 0}, {0, 0}}, rc_row[2] = {0, 0}, rc_nd[2] = {0, 0}, src_gp[2] = {0, 0}, src_row[2] = {0, 0}, src_nd[2] = {0, 0};
  int rc_count = 0, rc_type = 0, src_count = 0, src_type = 0;
  if (others) {
    const unsigned tr = (unsigned)(f.nnz_lu + n + lane0);
    auto row_of = [&](int node) -> unsigned { return node < 0 ? tr : (unsigned)rowof[node]; };
    if (f.rc_blk >= 0) {
      const F2Block B = load_block(f.blk, f.rc_blk);
      rc_count = B.count; rc_type = B.type;
      const short* nd = nodes + B.nodes_off;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int dev = lane0 + 64 * q;
        unsigned p[4] = {tr, tr, tr, tr};
        int np = -1, nn = -1;
        if (dev < B.count) {
          np = nd[dev]; nn = nd[B.count + dev];
#pragma unroll
          for (int k = 0; k < 4; ++k)
            p[k] = B.type == CADNIP_DEV_CAPACITOR ? (unsigned)cdesc[B.c_base + k * B.count + dev] & 0xFFFFu : (unsigned)gpos[B.g_base + k * B.count + dev];
        }
        rc_gp[q][0] = p[0] | p[1] << 16; rc_gp[q][1] = p[2] | p[3] << 16;
        rc_row[q] = row_of(np) | row_of(nn) << 16;
        rc_nd[q] = ((unsigned)np & 0xFFFFu) | ((unsigned)nn & 0xFFFFu) << 16;
      }
    }
    if (f.src_blk >= 0) {
      const F2Block B = load_block(f.blk, f.src_blk);
      src_count = B.count; src_type = B.type;
      const short* nd = nodes + B.nodes_off;
      unsigned p[4] = {tr, tr, tr, tr};
      int np = -1, nn = -1, ni = -1;
      if (lane0 < B.count) {
        np = nd[lane0]; nn = nd[B.count + lane0];
        if (B.type == CADNIP_DEV_VSOURCE) {
          ni = nd[2 * B.count + lane0];
#pragma unroll
          for (int k = 0; k < 4; ++k) p[k] = (unsigned)gpos[B.g_base + k * B.count + lane0];
        }
      }
      src_gp[0] = p[0] | p[1] << 16; src_gp[1] = p[2] | p[3] << 16;
      src_row[0] = row_of(np) | row_of(nn) << 16; src_row[1] = row_of(ni);
      src_nd[0] = ((unsigned)np & 0xFFFFu) | ((unsigned)nn & 0xFFFFu) << 16; src_nd[1] = (unsigned)ni & 0xFFFFu;
    }
  }
#ifdef CADNIP_TRACE
  if (blockIdx.x == 0 && tid == 0) g_trace_last = clock64();
#endif
  // The first pass over the first sp_mos1 block (all of them when the circuit has at most 32 MOSFETs) runs on a register-resident view of
  // this lane's device (devices.hpp: M1RegView): node indices and charge-form flags are properties of the circuit and read here, once per
  // launch; its stamp tables' base pointers likewise.  What is left for the per-round block loop -- further passes, further blocks, blocks
  // beyond the pinned ones on the last wave -- is usually nothing: `more` says so and the loop is skipped.
  M1RegView rv;
  int vdep0 = 0, m1_blk0 = -1, m1_count0 = 0, m1_npar0 = 0;
  bool m1_valid0 = false, more = false;
  const u16* m1_gpos = gpos; const u64* m1_cdesc = cdesc; const u16* m1_brow = brow;
  const double* m1_par0 = nullptr;
  for (int bi = 0; bi < f.n_blk; ++bi) {
    const F2Block B = load_block(f.blk, bi);
    if (B.type == CADNIP_DEV_MOS1 && B.mos1_plain) {
      if (m1_blk0 < 0) {
        m1_blk0 = bi; m1_count0 = B.count; m1_npar0 = B.n_par;
        const int dv = lane0 >> 1;
        m1_valid0 = dv < B.count;
        const int dev = m1_valid0 ? dv : B.count - 1;
        vdep0 = B.ipar[dev];
        const short* nd = nodes + B.nodes_off;
#pragma unroll
        for (int k = 0; k < 14; ++k) rv.nd[k] = nd[k * B.count + dev];
        rv.initjct = 0;
        m1_gpos = gpos + B.g_base; m1_cdesc = cdesc + B.c_base; m1_brow = brow + B.b_base;
        m1_par0 = B.par;
        if (roles && B.count > 32) more = true;
      } else if (roles) more = true;
    } else if (others && !((bi == f.rc_blk && B.count <= 128) || (bi == f.src_blk && B.count <= 64))) more = true;
  }
#pragma unroll
  for (int k = 0; k < M1_NPAR; ++k) rv.p[k] = 0.0;
  TeamVecs<NT, KPF> vec;
  vec.us = us; vec.betas = betas; vec.W = W; vec.qinv = qinv; vec.red = red; vec.z0 = f.nnz_lu; vec.z1 = nW;
  if constexpr (!STEP) vec.load_weights(a, tid);
  // The team works through instances one after the other: its first one by position, further ones from the grid's queue.
  int inst = blockIdx.x;
  int budget = f.rounds;
  for (;;) {
    StepState st;
    bool have = false;
    while (inst < f.B) {
      if constexpr (STEP) {
        // the caller's point and leading coefficient; the refactor decision of begin_round follows f.step_refresh
        st.t = st.h = st.hprev = st.hpp = 0.0; st.tn = kargs()->t.tcur[inst]; st.a0 = kargs()->t.gamma[inst];
        if (tid == 0) { if (kargs()->t.t) kargs()->t.t[inst] = st.tn; if (kargs()->t.h) kargs()->t.h[inst] = st.a0; }    // the handle's copies (api.hip)
        st.nhist = 1; st.ord = 1; st.k = 1; st.status = 0; st.bp = st.si = 0; st.c_newton = st.c_accept = st.c_reject = st.c_fail = 0;
        st.t_break = st.t_save = 0.0; st.a0f = st.a0; st.ss = 20.0; st.dnp = 0.0; st.dsc = 1.0; st.mflags = f.step_refresh ? MN_NEED : MN_VALID; st.hp3 = 0.0;
        make_uniform(st);
        have = true; break;
      }
      const TranStateView sv = state_view();
      st = load_state(sv, inst); make_uniform(st);
      if (st.status == 0) { have = true; break; }
      __syncthreads();                                     // the previous hand-out has been read by every wave
      if (tid == 0) s_next = atomicAdd(kargs()->queue, 1);
      __syncthreads();
      inst = (int)gridDim.x + __builtin_amdgcn_readfirstlane(s_next);
    }
    if (!have || budget <= 0) break;
    const size_t vo = (size_t)inst * n;
    double* lw = a.use_pcnr ? kargs()->t.limit_w + vo : nullptr;
    vec.vo = vo; vec.lw = lw;
    if constexpr (STEP) {
      // beta = du - a0 u, so that the devices' a0 u + beta is the caller's du
      const F2ArgsK ka = kargs();
      const double *ug = ka->t.u + vo, *dug = ka->t.du + vo;
      const double a0 = st.a0;
      for (int i = tid; i < n; i += NT) { const double x = ug[i]; us[i] = x; betas[i] = fma(-a0, x, dug[i]); }
    } else {
      vec.load_history(n, tid);
      const F2ArgsK ka = kargs();
      const double *ug = ka->t.u + vo, *betag = ka->t.beta + vo;
      for (int i = tid; i < n; i += NT) { us[i] = ug[i]; betas[i] = betag[i]; }
    }
    if (par_words > 0)
    for (int bi = 0; bi < f.n_blk; ++bi) {                  // parameter rows of the sp_mos1 blocks beyond the register-resident pass: one coalesced copy per residence
      const F2Block B = load_block(f.blk, bi);
      if (B.lds_par < 0) continue;
      const int words = B.n_par * B.count;
      const double* src = B.par + (size_t)inst * words;
      for (int i = tid; i < words; i += NT) parc[B.lds_par + i] = src[i];
    }
    const bool mn = STEP || a.newton_mode != 0;
    if (mn && (st.mflags & MN_VALID)) {
      const double* src = kargs()->lufac + (size_t)inst * f.nnz_lu;
      for (int i = tid; i < f.nnz_lu; i += NT) W[i] = src[i];
    }
    if (roles && m1_blk0 >= 0)                               // this lane's device, this lane's junction side: into registers for the residence
      m1_load_params(rv, m1_par0 + (size_t)inst * m1_npar0 * m1_count0, m1_count0, m1_valid0 ? (lane0 >> 1) : m1_count0 - 1, (lane0 & 1) != 0);
    double rc_val[2] = {0.0, 0.0};
    if (others && f.rc_blk >= 0) {
      const F2Block B = load_block(f.blk, f.rc_blk);
      const double* par = B.par + (size_t)inst * B.n_par * B.count;
#pragma unroll
      for (int q = 0; q < 2; ++q) { const int dev = lane0 + 64 * q; rc_val[q] = par[dev < B.count ? dev : 0]; }
    }
    double src_val = 0.0, src_t = 0.0;
    bool src_have = false;
    int src_seg = 0;
    // Does the next round refactor (Newton mode 1: IDA's lsetup conditions, as k_fused2)?  Then the whole work array is cleared, else only
    // the right-hand side and the trash words.  Runs at the END of a round (and once at pick-up), so that one barrier closes the
    // update and the clearing together.
    bool refresh = true;
    int step_rep = 0;
    auto begin_round = [&]() {
      refresh = true;
      if (mn) {
        refresh = (st.mflags & MN_NEED) || !(st.mflags & MN_VALID) ||
                  (st.k == 0 && (st.a0 < 0.6 * st.a0f || st.a0 * 0.6 > st.a0f || (st.mflags >> MN_SINCE_SHIFT) >= 20));
        if (refresh) { st.a0f = st.a0; st.ss = 20.0; st.mflags = MN_VALID | MN_JCUR; st.dsc = 1.0; }
        else st.dsc = st.a0 == st.a0f ? 1.0 : fast_div(2.0, 1.0 + fast_div(st.a0, st.a0f));
      }
      if (refresh) { for (int i = tid; i < (f.nnz_lu >> 1); i += NT) ((double2*)W)[i] = make_double2(0.0, 0.0); if (tid == 0 && (f.nnz_lu & 1)) W[f.nnz_lu - 1] = 0.0; }
    };
    __syncthreads();                                        // the kept factors are in place before they are cleared (a refactoring round)
    vec.step_consumed(tid);                                 // rhs | trash: cleared by the update from now on
    begin_round();
    __syncthreads();
    for (;;) {
      int lane = lane0;
      asm volatile("" : "+v"(lane));
      CADNIP_TRACE_POINT(17);
      const double tcur = st.tn, a0 = st.a0;
      const unsigned trash_w = (unsigned)(f.nnz_lu + n + lane);
      // ---- stamp: every wave its share
      const int skip = STEP ? f.step_skip : 0;                // (measurement: cadnip_debug_step_time)
      if (others && !(skip & 1)) {
        auto at = [&](unsigned nd16) -> double { const double x = us[nd16 == 0xFFFFu ? 0u : nd16]; return nd16 == 0xFFFFu ? 0.0 : x; };
        auto dat = [&](unsigned nd16) -> double {
          const unsigned i = nd16 == 0xFFFFu ? 0u : nd16;
          const double x = a0 * us[i] + betas[i];
          return nd16 == 0xFFFFu ? 0.0 : x;
        };
        if (f.rc_blk >= 0) {
          const bool cap = rc_type == CADNIP_DEV_CAPACITOR;
          double jv[2], cur[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const unsigned np = rc_nd[q] & 0xFFFFu, nn = rc_nd[q] >> 16;
            const double xp = cap ? dat(np) : at(np), xn = cap ? dat(nn) : at(nn);
            jv[q] = cap ? a0 * rc_val[q] : rc_val[q];
            cur[q] = rc_val[q] * (xp - xn);
          }
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            if (q == 1 && rc_count <= 64) break;
            if (refresh) {
              atomicAdd(&Wacc[rc_gp[q][0] & 0xFFFFu], jv[q]); atomicAdd(&Wacc[rc_gp[q][0] >> 16], -jv[q]);
              atomicAdd(&Wacc[rc_gp[q][1] & 0xFFFFu], -jv[q]); atomicAdd(&Wacc[rc_gp[q][1] >> 16], jv[q]);
            }
            atomicAdd(&Wacc[rc_row[q] & 0xFFFFu], cur[q]); atomicAdd(&Wacc[rc_row[q] >> 16], -cur[q]);
          }
        }
        if (f.src_blk >= 0) {
          if (!src_have || tcur != src_t) {
            const F2Block B = load_block(f.blk, f.src_blk);
            const double* par = B.par + (size_t)inst * B.n_par * B.count;
            LdsCtx d{nodes + B.nodes_off, B.ipar, par, f.wave, B.count, lane < B.count ? lane : 0, tcur, 1, 0};
            src_val = source_value(d, par_of(d, 0), par_of(d, 1), &src_seg);
            src_t = tcur; src_have = true;
          }
          if (src_type == CADNIP_DEV_VSOURCE) {
            const double ui = at(src_nd[1]), vd = at(src_nd[0] & 0xFFFFu) - at(src_nd[0] >> 16) - src_val;
            if (refresh) {
              atomicAdd(&Wacc[src_gp[0] & 0xFFFFu], 1.0); atomicAdd(&Wacc[src_gp[0] >> 16], -1.0);
              atomicAdd(&Wacc[src_gp[1] & 0xFFFFu], 1.0); atomicAdd(&Wacc[src_gp[1] >> 16], -1.0);
            }
            atomicAdd(&Wacc[src_row[0] & 0xFFFFu], ui); atomicAdd(&Wacc[src_row[0] >> 16], -ui);
            atomicAdd(&Wacc[src_row[1]], vd);
          } else {
            atomicAdd(&Wacc[src_row[0] & 0xFFFFu], -src_val); atomicAdd(&Wacc[src_row[0] >> 16], src_val);
          }
        }
        CADNIP_TRACE_POINT(13);
      }
      if (roles && m1_blk0 >= 0 && !(skip & 1)) {
        const int side = lane & 1;
        rv.initjct = 0;
        if (!refresh) {       // round on kept factors: residuals only
          AccumOutT<true, true, false> s{Wacc, betas, a0, m1_gpos, m1_cdesc, m1_brow, m1_count0, m1_valid0 ? (lane >> 1) : m1_count0 - 1, m1_valid0 ? 0u : trash_w, us, rowof, trash_w};
          stamp_mos1_team(rv, us, s, lw, side, m1_valid0, roles, vdep0);
        } else {
          AccumOutT<true, true> s{Wacc, betas, a0, m1_gpos, m1_cdesc, m1_brow, m1_count0, m1_valid0 ? (lane >> 1) : m1_count0 - 1, m1_valid0 ? 0u : trash_w, us, rowof, trash_w};
          stamp_mos1_team(rv, us, s, lw, side, m1_valid0, roles, vdep0);
        }
      }
      if (more && !(skip & 1))
      for (int bi = 0; bi < f.n_blk; ++bi) {
        const F2Block B = load_block(f.blk, bi);
        if (B.type == CADNIP_DEV_MOS1 && B.mos1_plain) {
          if (roles == 0) continue;
          const double* par = parc + B.lds_par;
          const int side = lane & 1;
          for (int d0 = bi == m1_blk0 ? 32 : 0; d0 < B.count; d0 += 32) {
            const int dv = d0 + (lane >> 1);
            const bool valid = dv < B.count;
            LdsCtx d{nodes + B.nodes_off, B.ipar, par, f.wave, B.count, valid ? dv : B.count - 1, tcur, 1, 0};
            const int vdep = B.ipar[d.dev];
            if (!refresh) {       // round on kept factors: residuals only
              AccumOutT<true, true, false> s{Wacc, betas, a0, gpos + B.g_base, cdesc + B.c_base, brow + B.b_base, B.count, d.dev, valid ? 0u : trash_w, us, rowof, trash_w};
              stamp_mos1_team(d, us, s, lw, side, valid, roles, vdep);
            } else {
              AccumOutT<true, true> s{Wacc, betas, a0, gpos + B.g_base, cdesc + B.c_base, brow + B.b_base, B.count, d.dev, valid ? 0u : trash_w, us, rowof, trash_w};
              stamp_mos1_team(d, us, s, lw, side, valid, roles, vdep);
            }
          }
          continue;
        }
        if (!others) continue;
        if ((bi == f.rc_blk && rc_count <= 128) || (bi == f.src_blk && src_count <= 64)) continue;   // all of it was pinned
        const double* par = B.par + (size_t)inst * B.n_par * B.count;
        int dev0 = lane;
        if (bi == f.rc_blk) dev0 = lane + 128;
        if (bi == f.src_blk) dev0 = lane + 64;
        for (int dev = dev0; dev < B.count; dev += 64) {
          LdsCtx d{nodes + B.nodes_off, B.ipar, par, f.wave, B.count, dev, tcur, 1, 0};
          if (!refresh) {
            AccumOutT<false, true, false> s{Wacc, betas, a0, gpos + B.g_base, cdesc + B.c_base, brow + B.b_base, B.count, dev, 0u, us, rowof, trash_w};
            dispatch_stamp2<true>(B.type, d, us, s, lw);
          } else {
            AccumOutT<false, true> s{Wacc, betas, a0, gpos + B.g_base, cdesc + B.c_base, brow + B.b_base, B.count, dev, 0u, us, rowof, trash_w};
            dispatch_stamp2<true>(B.type, d, us, s, lw);
          }
        }
      }
      CADNIP_TRACE_POINT(8);
      __syncthreads();
      // the other waves' contributions, in wave order (a round on kept factors has stamped the right-hand side only)
      if (!(skip & 2))
      for (int i = (refresh ? 0 : f.nnz_lu) + tid; i < f.nnz_lu + n; i += NT) {
        double acc = W[i];
#pragma unroll
        for (int k = 0; k < NW - 1; ++k) { double* q = WP + (size_t)k * nW + i; acc += *q; *q = 0.0; }
        W[i] = acc;
      }
      __syncthreads();
      if constexpr (STEP) {
        // the residual as the caller sees it (unknown order) and its 2-norm, before the linear solve overwrites the right-hand side
        double* ro = kargs()->step_resid ? kargs()->step_resid + vo : nullptr;
        double s2 = 0.0;
        for (int i = tid; i < n; i += NT) { const double r = W[rowof[i]]; s2 += r * r; if (ro) ro[i] = r; }
        double dummy = 0.0; int nob = 0;
        vec.reduce3(s2, dummy, nob);
        if (tid == 0 && kargs()->step_norm) kargs()->step_norm[inst] = sqrt(s2);
      }
      CADNIP_TRACE_POINT(1);
      // ---- refactor + forward + backward substitution: straight-line steps (f2_program.cpp: f2_build_steps, list-scheduled).  A step gives every
      // thread of the team one entry share W[pos] = (W[pos] - sum of up to three W[a] W[b], summed over the entry's lane group) / W[piv]; the
      // next step's descriptor is read (LDS) together with this step's operands.  Lanes without work and entries without a division point at the constant words
      // (0.0, 1.0) behind the work array.  Every step ends with a workgroup barrier.
      int bad = 0;
      // Teams of four run the level-aligned ONE-term layout (f2_build_team, 8-byte descriptors): with 256 lanes a level fits a step anyway and
      // the step is shorter (four operand reads); teams of two the list-scheduled THREE-term layout (f2_build_steps, 16 bytes): 17 instead of
      // 22 steps on the flip-flop.  (Measured: three terms on four waves 14.3 -> 15.7 ms per transient, on two waves 17.2 -> 16.6 ms.)
      auto run_steps = [&](const int s_first, const int s_count) {
        if (s_count <= 0) return;
        if constexpr (NW == 4) {
          const u64* dp = (const u64*)tdesc + (size_t)s_first * NT + tid;
          u64 D = dp[0];
          for (int si = 0; si < s_count; ++si) {
            const unsigned lo = (unsigned)D, hi = (unsigned)(D >> 32);
            double* const pp = W + (lo & 0x7FFFu);
            const double piv = W[(lo >> 16) & 0x7FFFu], av = W[hi & 0x7FFFu], bv = W[(hi >> 16) & 0x7FFFu];
            const double acc0 = *pp;
            const u64 Dn = dp[(size_t)(si + 1) * NT];
            const unsigned lg = (lo >> 31) | ((hi >> 14) & 2u) | ((hi >> 29) & 4u);
            double part = av * bv;
            { const double o = dpp_f64<0xB1>(part); part += lg >= 1 ? o : 0.0; }
            { const double o = dpp_f64<0x4E>(part); part += lg >= 2 ? o : 0.0; }
            { const double o = dpp_f64<0x141>(part); part += lg >= 3 ? o : 0.0; }
            { const double o = dpp_f64<0x140>(part); part += lg >= 4 ? o : 0.0; }
            if (piv == 0.0 || !isfinite(piv)) bad = 1;
            const double acc = fast_div(acc0 - part, piv);
            // every lane divides and stores -- the lanes that are not their group's leader into their trash word: no branch, and the entry's
            // old value is read with the other operands instead of behind the group sum
            *((lo & 0x8000u) ? pp : W + trash_w) = acc;
            __syncthreads();
            D = Dn;
          }
        } else {
          const uint4* dp = tdesc + (size_t)s_first * NT + tid;
          uint4 D = dp[0];
          for (int si = 0; si < s_count; ++si) {
            double* const pp = W + (D.x & 0x7FFFu);
            const double piv = W[(D.x >> 16) & 0x7FFFu];
            const double a0v = W[D.y & 0x7FFFu], b0v = W[(D.y >> 16) & 0x7FFFu];
            const double a1v = W[D.z & 0x7FFFu], b1v = W[(D.z >> 16) & 0x7FFFu];
            const double a2v = W[D.w & 0x7FFFu], b2v = W[(D.w >> 16) & 0x7FFFu];
            const double acc0 = *pp;
            const uint4 Dn = dp[(size_t)(si + 1) * NT];
            const unsigned lg = (D.x >> 31) | ((D.y >> 14) & 2u) | ((D.y >> 29) & 4u);
            double part = fma(a2v, b2v, fma(a1v, b1v, a0v * b0v));
            { const double o = dpp_f64<0xB1>(part); part += lg >= 1 ? o : 0.0; }
            { const double o = dpp_f64<0x4E>(part); part += lg >= 2 ? o : 0.0; }
            { const double o = dpp_f64<0x141>(part); part += lg >= 3 ? o : 0.0; }
            { const double o = dpp_f64<0x140>(part); part += lg >= 4 ? o : 0.0; }
            if (piv == 0.0 || !isfinite(piv)) bad = 1;
            const double acc = fast_div(acc0 - part, piv);
            *((D.x & 0x8000u) ? pp : W + trash_w) = acc;
            __syncthreads();
            D = Dn;
          }
        }
      };
      if (skip & 4) {}
      else if (refresh) run_steps(0, f.ts_pre);
      else run_steps(f.ts_pre + f.ts_post, f.ts_fwd);
      CADNIP_TRACE_POINT(4);
      if (f.nc > 0 && !(skip & 8)) {
        if (w == 0) {
          const int yc0 = f.nnz_lu + n - f.nc;
          if (refresh) {
            if (f.nc == 8) dense_core_solve<8, 0>(W, f.dn0, yc0, lane, bad, mn);
            else if (f.nc == 12) dense_core_solve<12, 0>(W, f.dn0, yc0, lane, bad, mn);
            else dense_core_solve<F2_NCMAX, 0>(W, f.dn0, yc0, lane, bad, mn);
          } else {
            if (f.nc == 8) dense_core_solve<8, 1>(W, f.dn0, yc0, lane, bad, false);
            else if (f.nc == 12) dense_core_solve<12, 1>(W, f.dn0, yc0, lane, bad, false);
            else dense_core_solve<F2_NCMAX, 1>(W, f.dn0, yc0, lane, bad, false);
          }
        }
        __syncthreads();
      }
      CADNIP_TRACE_POINT(5);
      if (!(skip & 4)) run_steps(f.ts_pre, f.ts_post);
      CADNIP_TRACE_POINT(3);
      if constexpr (STEP) {
        double* dout = kargs()->t.delta + vo;                 // (possibly mapped host memory: plain stores, a system-scope fence behind them)
        for (int i = tid; i < n; i += NT) { const double dd = W[qinv[i]]; if (!isfinite(dd)) bad = 1; dout[i] = dd; }
        if (__syncthreads_or(bad) && tid == 0) kargs()->t.flags[inst] = 1;
        __threadfence_system();
        if (++step_rep < f.step_reps) {                       // measurement: the same iteration again
          __syncthreads();
          vec.step_consumed(tid);
          st.mflags = f.step_refresh ? MN_NEED : MN_VALID; st.k = 1;
          begin_round();
          __syncthreads();
          continue;
        }
        st.status = 1;                                        // one iteration: done
        --budget;
        break;
      }
      // ---- Newton update + step controller: thread t owns unknown t; every wave decides for itself from the same sums
      tran_update_body(a, vec, st, inst, tid, bad);
      make_uniform(st);
      CADNIP_TRACE_POINT(16);
      if (st.status != 0) { --budget; break; }
      if (--budget <= 0) break;
      begin_round();                                        // (a refactoring round clears the matrix words: every thread is past the linear solve -- the update's barrier)
      __syncthreads();
    }
    __syncthreads();
    {
      const F2ArgsK ka = kargs();
      if constexpr (STEP) {
        if (f.step_refresh) {                                 // the factors stay for the caller's next iterations on this Jacobian
          double* dst = ka->lufac + (size_t)inst * f.nnz_lu;
          for (int i = tid; i < f.nnz_lu; i += NT) dst[i] = W[i];
        }
      } else {
      double *ug = ka->t.u + vo, *betag = ka->t.beta + vo, *dug = ka->t.du + vo;
      const double a0 = st.a0;
      for (int i = tid; i < n; i += NT) { double x = us[i], b = betas[i]; ug[i] = x; betag[i] = b; dug[i] = a0 * x + b; }
      if (mn && (st.mflags & MN_VALID) && st.status == 0) {
        double* dst = ka->lufac + (size_t)inst * f.nnz_lu;
        for (int i = tid; i < f.nnz_lu; i += NT) dst[i] = W[i];
      }
      vec.store_history(n, tid);
      store_state(state_view(), inst, tid, st);
      }
    }
    __syncthreads();
    if (st.status == 0) break;                // out of budget in the middle of this instance: the next launch resumes it
    if (tid == 0) s_next = atomicAdd(kargs()->queue, 1);
    __syncthreads();
    inst = (int)gridDim.x + __builtin_amdgcn_readfirstlane(s_next);
  }
}

int fteam_launch(int nw, int grid, size_t shmem, hipStream_t stream, const F2Args& f);   // fused_team.hip
int fteam_launch_step(int grid, size_t shmem, hipStream_t stream, const F2Args& f);      // ... k_fteam<4, true>

}  // namespace cadnip
