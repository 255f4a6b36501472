// fused2_kernel.hpp -- the fused Newton kernel itself (template k_fused2<WPB, DC, VAR>) and its argument block.  Included by
// fused2.hip (host side: tables, launch geometry) and by the three translation units that instantiate the kernel, one per
// variant (fused2_v0/1/2.hip), so that the variants compile in parallel.  See fused2.hip for the design notes.
#pragma once
#include <hip/hip_runtime.h>
#include "devices.hpp"
#include "internal.hpp"
#include "tran_ctrl.hpp"

namespace cadnip {

typedef unsigned short u16;
typedef unsigned long long u64;
#define NOPOS 0xFFFFu
#define F2_NCMAX 16   // largest core the in-register dense solve is unrolled for
#define F2_JU 18      // J*u entries per lane and chunk
#define F2_MAX_BLOCKS 32   // device blocks of one circuit (one per built-in type, one per Verilog-A module); more: per-op path
#define F2_TRASH 64   // per-instance trash words (one per lane) that absorb stamps into ground rows / columns

// table sections (offsets in 32-bit words, every section 8-byte aligned)
enum { S_GPOS = 0, S_CDESC, S_BROW, S_NZ, S_ENT, S_TERM, S_LEV, S_QINV, S_NODES, S_ROWOF, S_LOADPOS, S_NSEC };   // S_LOADPOS: csr entry -> W word (lu_f2.hip)

struct F2Block {
  const int* ipar; const double* par;
  int type, count, n_par, g_base, c_base, b_base, nodes_off, mos1_plain;
  int lds_par;               // team kernel (fused_team_kernel.hpp): word offset of the block's parameter rows in the LDS parameter area, -1 = not staged
};

// The device blocks are read through a pointer: an array inside the by-value argument struct, indexed by the block
// loop's counter, would be copied to scratch memory and re-read from there (an HBM-latency load) at every use.
typedef const __attribute__((address_space(4))) F2Block* F2BlockPtr;   // constant address space: uniform reads become scalar loads

__device__ __forceinline__ F2Block load_block(const F2Block* blk, int i) {
  F2BlockPtr q = (F2BlockPtr)blk + i;
  F2Block b;
  b.ipar = q->ipar; b.par = q->par; b.type = q->type; b.count = q->count; b.n_par = q->n_par; b.g_base = q->g_base; b.c_base = q->c_base;
  b.b_base = q->b_base; b.nodes_off = q->nodes_off; b.mos1_plain = q->mos1_plain; b.lds_par = q->lds_par;
  return b;
}

struct F2Args {
  const F2Block* blk;        // [n_blk] in device memory
  int n_blk, rc_blk;         // rc_blk: index of the first capacitor / resistor block (-1 = none)
  int src_blk;               // index of the first independent-source block (-1 = none)
  const double* wave;
  const unsigned* tab;       // packed tables in global memory
  int off[S_NSEC];
  int tab_len;               // 32-bit words the kernel stages in LDS (a multiple of 4) ...
  int tab_lo;                // ... starting at this word of `tab` (a multiple of 4): the lean kernels leave the pass program and the J*u list out
  int n, nnz, nnz_lu, rounds, B;   // nnz_lu: words of W before the rhs (sparse L\U entries + dense core block)
  int n_pre, n_post, nc, dn0;       // passes before / after the dense core solve, core size, first word of the core block
  int n_fwd;                        // Newton mode 1: passes of the forward substitution alone (kept factors), behind the pre / post passes
  double* lufac;                    // [B][nnz_lu] kept factors of the instances that are not resident (between launches / while queued)
  const unsigned long long* team_desc; int team_desc_len;   // team kernel (fused_team_kernel.hpp): step descriptors of the linear solve (f2_program.cpp: f2_build_team)
  int ts_pre, ts_post, ts_fwd;      // ... steps of the pre-core, post-core and forward-only lists
  int par_words;                    // team kernel: doubles of the LDS-staged sp_mos1 parameter rows (F2Block::lds_par)
  int step_refresh; double *step_resid, *step_norm;   // team kernel, STEP mode (cadnip_newton_step_fused): refactor or use f.lufac; optional outputs
  int step_reps, step_skip;         // STEP mode, measurement only (cadnip_debug_step_time): repeat the same iteration, leave phases out (1 stamping, 2 combine, 4 linear-solve steps, 8 dense core)
  // DC mode (k_fused2<WPB, true>): PCNR / plain Newton on G u = b (driver.hip: k_dc_check, k_dc_update)
  double dc_abstol; int dc_maxiters, dc_pcnr, dc_mode, dc_initjct; int* dcstate;
  const int* cold;           // [B] DC mode: 1 = the instance starts cold (initjct applies to it), driver.hip: k_dc_init
  int* queue;                // next not-yet-resident instance (relative to gridDim.x * WPB); zeroed before every launch
  TranArgs t;
};

// Scalar registers are the scarce resource next to vector registers: the argument block alone is > 120 dwords, and a
// kernel argument stays live (i.e. spilled to vector lanes, reloaded with v_readlane) from the entry block to its last
// use.  Pointers that are needed only when an instance is picked up, handed back or writes outputs are therefore not
// taken from `f` but fetched from the kernarg segment where they are used (scalar loads, constant cache); the opaque
// copy of the segment pointer keeps those loads from being hoisted back to the top.
typedef const __attribute__((address_space(4))) F2Args* F2ArgsK;
__device__ __forceinline__ F2ArgsK kargs() {
  F2ArgsK p = (F2ArgsK)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}
// per-instance controller state arrays (member names as in TranArgs: load_state / store_state take either)
struct TranStateView {
  double *t, *h, *hprev, *hpp, *tcur, *gamma;
  int *nhist, *order, *k, *status, *bp_idx, *save_idx, *active;
  long long* cnt;
  const double *breaks, *save_t;
  int n_break, n_save;
  int newton_mode; double *mn_a0f, *mn_ss, *mn_dnp; int* mn_flags;
  int max_order; double* hp3;
};
__device__ __forceinline__ TranStateView state_view() {
  const F2ArgsK p = kargs();
  TranStateView v;
  v.t = p->t.t; v.h = p->t.h; v.hprev = p->t.hprev; v.hpp = p->t.hpp; v.tcur = p->t.tcur; v.gamma = p->t.gamma;
  v.nhist = p->t.nhist; v.order = p->t.order; v.k = p->t.k; v.status = p->t.status; v.bp_idx = p->t.bp_idx; v.save_idx = p->t.save_idx;
  v.active = p->t.active; v.cnt = p->t.cnt; v.breaks = p->t.breaks; v.save_t = p->t.save_t; v.n_break = p->t.n_break; v.n_save = p->t.n_save;
  v.newton_mode = p->t.newton_mode; v.mn_a0f = p->t.mn_a0f; v.mn_ss = p->t.mn_ss; v.mn_dnp = p->t.mn_dnp; v.mn_flags = p->t.mn_flags;
  v.max_order = p->t.max_order; v.hp3 = p->t.hp3;
  return v;
}

// stamp writer: accumulates into J (LU positions) and the residual; all targets are offsets into W.
// GUARD: a lane whose `sink` is non-zero (no device behind it) sends every stamp to that trash word instead.
// DIRECT: the residual comes from the devices (Rn, devices.hpp) -- b stamps and the C*beta terms are dropped here and
// the kernel skips its J*u product.
// JAC = false: a round on kept factors (Newton mode 1) -- the matrix stamps G / C are no-ops, so everything the devices compute only
// for them (the partial derivatives) is dead code to the compiler; the residual stamps remain.
template <bool GUARD, bool DIRECT_, bool JAC = true>
struct AccumOutT {
  static constexpr bool DIRECT = DIRECT_;
  static_assert(JAC || DIRECT_, "a residual-only pass needs the direct residuals");
  double* W; const double* betas; double a0;
  const u16* gpos; const u64* cdesc; const u16* brow;   // already offset to this device block
  int count, dev;
  unsigned sink;
  const double* us; const u16* rowof; unsigned trash;    // DIRECT: u, unknown index -> rhs word, this lane's trash word
  __device__ __forceinline__ unsigned tg(unsigned p) const { return GUARD && sink ? sink : p; }
  __device__ __forceinline__ double du(int node) const { return node < 0 ? 0.0 : a0 * us[node] + betas[node]; }
  __device__ __forceinline__ void Rn(int node, double v) const {
    if (!DIRECT) return;
    atomicAdd(&W[tg(node < 0 ? trash : (unsigned)rowof[node])], v);
  }
  __device__ __forceinline__ void G(int k, double v) const {
    if (!JAC) return;
    if (__builtin_constant_p(v) && v == 0.0) return;     // structurally zero stamps cost nothing
    atomicAdd(&W[tg(gpos[k * count + dev])], v);
  }
  __device__ __forceinline__ void C(int k, double v) const {
    if (!JAC) return;
    if (__builtin_constant_p(v) && v == 0.0) return;
    const u64 d = cdesc[k * count + dev];
    atomicAdd(&W[tg((unsigned)d & 0xFFFFu)], a0 * v);
    if (!DIRECT) atomicAdd(&W[tg((unsigned)(d >> 16) & 0xFFFFu)], v * betas[(unsigned)(d >> 32) & 0xFFFFu]);
  }
  __device__ __forceinline__ void B(int k, double v) const {
    if (DIRECT || (__builtin_constant_p(v) && v == 0.0)) return;
    atomicAdd(&W[tg(brow[k * count + dev])], -v);
  }
  // batch forms: all table reads of the batch are issued before its first atomic, so a batch costs one LDS
  // round trip plus the atomics' issue slots instead of one dependent read -> atomic chain per stamp
  template <int N> __device__ __forceinline__ void Gv(int k0, const double (&v)[N]) const {
    if (!JAC) return;
    unsigned p[N];
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = gpos[(k0 + i) * count + dev];
#pragma unroll
    for (int i = 0; i < N; ++i) if (!(__builtin_constant_p(v[i]) && v[i] == 0.0)) atomicAdd(&W[tg(p[i])], v[i]);
  }
  template <int N> __device__ __forceinline__ void Gk(const int (&k)[N], const double (&v)[N]) const {
    if (!JAC) return;
    unsigned p[N];
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = gpos[k[i] * count + dev];
#pragma unroll
    for (int i = 0; i < N; ++i) atomicAdd(&W[tg(p[i])], v[i]);
  }
  template <int N> __device__ __forceinline__ void Cv(int k0, const double (&v)[N]) const {
    if (!JAC) return;
    u64 d[N];
    double bt[N];
#pragma unroll
    for (int i = 0; i < N; ++i) d[i] = cdesc[(k0 + i) * count + dev];
    if (!DIRECT) {
#pragma unroll
      for (int i = 0; i < N; ++i) bt[i] = betas[(unsigned)(d[i] >> 32) & 0xFFFFu];
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (!(__builtin_constant_p(v[i]) && v[i] == 0.0)) {
        atomicAdd(&W[tg((unsigned)d[i] & 0xFFFFu)], a0 * v[i]);
        if (!DIRECT) atomicAdd(&W[tg((unsigned)(d[i] >> 16) & 0xFFFFu)], v[i] * bt[i]);
      }
  }
  template <int N> __device__ __forceinline__ void Bv(int k0, const double (&v)[N]) const {
    if (DIRECT) return;
    unsigned p[N];
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = brow[(k0 + i) * count + dev];
#pragma unroll
    for (int i = 0; i < N; ++i) if (!(__builtin_constant_p(v[i]) && v[i] == 0.0)) atomicAdd(&W[tg(p[i])], -v[i]);
  }
};

typedef DevCtxT<short> LdsCtx;

// LEAN: the circuit holds linear elements, independent sources and lane-paired sp_mos1 only (those are stamped by
// stamp_mos1_pair in the kernel body).  The models left out are the register-hungry ones: with them compiled in, the
// kernel no longer fits 256 VGPRs without spilling (33 spilled, -10 % on the DFF sweep).
template <bool LEAN, class Out>
__device__ __forceinline__ void dispatch_stamp2(int type, const LdsCtx& d, const double* u, const Out& s, double* lw) {
  if constexpr (LEAN) {
    switch (type) {
      case CADNIP_DEV_RESISTOR: stamp_resistor(d, u, s, lw); break;
      case CADNIP_DEV_CAPACITOR: stamp_capacitor(d, u, s, lw); break;
      case CADNIP_DEV_INDUCTOR: stamp_inductor(d, u, s, lw); break;
      case CADNIP_DEV_VSOURCE: stamp_vsource(d, u, s, lw); break;
      case CADNIP_DEV_ISOURCE: stamp_isource(d, u, s, lw); break;
      case CADNIP_DEV_VCVS: stamp_vcvs(d, u, s, lw); break;
      case CADNIP_DEV_VCCS: stamp_vccs(d, u, s, lw); break;
      case CADNIP_DEV_CCVS: stamp_ccvs(d, u, s, lw); break;
      case CADNIP_DEV_CCCS: stamp_cccs(d, u, s, lw); break;
    }
    return;
  }
  switch (type) {
    case CADNIP_DEV_RESISTOR: stamp_resistor(d, u, s, lw); break;
    case CADNIP_DEV_CAPACITOR: stamp_capacitor(d, u, s, lw); break;
    case CADNIP_DEV_INDUCTOR: stamp_inductor(d, u, s, lw); break;
    case CADNIP_DEV_VSOURCE: stamp_vsource(d, u, s, lw); break;
    case CADNIP_DEV_ISOURCE: stamp_isource(d, u, s, lw); break;
    case CADNIP_DEV_VCVS: stamp_vcvs(d, u, s, lw); break;
    case CADNIP_DEV_VCCS: stamp_vccs(d, u, s, lw); break;
    case CADNIP_DEV_CCVS: stamp_ccvs(d, u, s, lw); break;
    case CADNIP_DEV_CCCS: stamp_cccs(d, u, s, lw); break;
    case CADNIP_DEV_DIODE: stamp_diode(d, u, s, lw); break;
    case CADNIP_DEV_DIODECAP: stamp_diodecap(d, u, s, lw); break;
    case CADNIP_DEV_SIMPLEMOS: stamp_simplemos(d, u, s, lw); break;
    case CADNIP_DEV_MOS1: stamp_mos1(d, u, s, lw); break;
    case CADNIP_DEV_BVSOURCE: stamp_bvsource(d, u, s, lw); break;
    case CADNIP_DEV_BISOURCE: stamp_bisource(d, u, s, lw); break;
    case CADNIP_DEV_VA: stamp_va(d, u, s, lw); break;
  }
}

// controller vector policy of the fused kernel (see tran_ctrl.hpp): u, beta and the Newton step in LDS
struct FusedVecs {
  // per-lane elements kept in registers (covers n <= 256): the history u0 / u1 / u2 and the predictor for the whole
  // residence of the instance, the error weights from `prefetch` to the update.  Elements beyond stay in HBM.
  static constexpr int KPF = 4;
  static constexpr bool OWN_REDUCE = false;
  static constexpr int NT = 64;          // one wave per instance (tran_ctrl.hpp: grp_sum / grp_sync)
  double *us, *betas; const double* W; const u16* qinv;
  size_t vo;                   // this instance's offset into the per-unknown vectors; their bases come from the kernarg segment
  const double* lw;
  double r_u0[KPF], r_u1[KPF], r_u2[KPF], r_up[KPF], pf_at[KPF], pf_em[KPF];
  __device__ __forceinline__ double* p_u0() const { return kargs()->t.u0 + vo; }
  __device__ __forceinline__ double* p_u1() const { return kargs()->t.u1 + vo; }
  __device__ __forceinline__ double* p_u2() const { return kargs()->t.u2 + vo; }
  __device__ __forceinline__ double* p_up() const { return kargs()->t.up + vo; }
  __device__ __forceinline__ double mem_u0(int i) const { return p_u0()[i]; }
  __device__ __forceinline__ double mem_u1(int i) const { return p_u1()[i]; }
  __device__ __forceinline__ void load_history(int n, int lane) {
    const double *u0 = p_u0(), *u1 = p_u1(), *u2 = p_u2(), *up = p_up();
#pragma unroll
    for (int k = 0; k < KPF; ++k) {
      const int i = lane + 64 * k < n ? lane + 64 * k : 0;
      r_u0[k] = u0[i]; r_u1[k] = u1[i]; r_u2[k] = u2[i]; r_up[k] = up[i];
    }
  }
  __device__ __forceinline__ void store_history(int n, int lane) const {
    double *u0 = p_u0(), *u1 = p_u1(), *u2 = p_u2(), *up = p_up();
#pragma unroll
    for (int k = 0; k < KPF; ++k) {
      const int i = lane + 64 * k;
      if (i < n) { u0[i] = r_u0[k]; u1[i] = r_u1[k]; u2[i] = r_u2[k]; up[i] = r_up[k]; }
    }
  }
  __device__ __forceinline__ void history_to_memory(int n, int lane) const {
    double *u0 = p_u0(), *u1 = p_u1();
#pragma unroll
    for (int k = 0; k < KPF; ++k) { const int i = lane + 64 * k; if (i < n) { u0[i] = r_u0[k]; u1[i] = r_u1[k]; } }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // read back by other lanes of this wave (save_outputs)
    __builtin_amdgcn_wave_barrier();
  }
  __device__ __forceinline__ void prefetch(const TranArgs& a, int lane) {
#pragma unroll
    for (int k = 0; k < KPF; ++k) {
      const int i = lane + 64 * k < a.n ? lane + 64 * k : 0;   // clamped, no select on the loaded value: nothing waits here
      pf_at[k] = a.atol[i]; pf_em[k] = a.emask[i];
    }
  }
  __device__ __forceinline__ double get_delta(int i, int) const { return W[qinv[i]]; }
  __device__ __forceinline__ void step_consumed(int) const {}
  __device__ __forceinline__ double get_u(int i) const { return us[i]; }
  __device__ __forceinline__ void set_u(int i, double v) const { us[i] = v; }
  __device__ __forceinline__ double get_beta(int i) const { return betas[i]; }
  __device__ __forceinline__ void set_beta(int i, double v) const { betas[i] = v; }
  __device__ __forceinline__ void set_du(int, double) const {}   // du is rebuilt from u and beta when the kernel exits
  __device__ __forceinline__ double get_lw(int i) const { return lw[i]; }
  // k: compile-time ordinal of a register-held element, or -1 (tran_ctrl.hpp: each_elem)
  __device__ __forceinline__ double h0(int i, int k) const { return k >= 0 ? r_u0[k] : p_u0()[i]; }
  __device__ __forceinline__ double h1(int i, int k) const { return k >= 0 ? r_u1[k] : p_u1()[i]; }
  __device__ __forceinline__ double h2(int i, int k) const { return k >= 0 ? r_u2[k] : p_u2()[i]; }
  __device__ __forceinline__ double hp(int i, int k) const { return k >= 0 ? r_up[k] : p_up()[i]; }
  // (max_order = 3: the fourth history point stays in memory -- the kernel has no registers to spare, and the default order never reads it)
  __device__ __forceinline__ double h3(int i, int) const { return (kargs()->t.u3 + vo)[i]; }
  __device__ __forceinline__ void set_h3(int i, int, double v) { (kargs()->t.u3 + vo)[i] = v; }
  __device__ __forceinline__ void set_h0(int i, int k, double v) { if (k >= 0) r_u0[k] = v; else p_u0()[i] = v; }
  __device__ __forceinline__ void set_h1(int i, int k, double v) { if (k >= 0) r_u1[k] = v; else p_u1()[i] = v; }
  __device__ __forceinline__ void set_h2(int i, int k, double v) { if (k >= 0) r_u2[k] = v; else p_u2()[i] = v; }
  __device__ __forceinline__ void set_hp(int i, int k, double v) { if (k >= 0) r_up[k] = v; else p_up()[i] = v; }
  __device__ __forceinline__ double atol_of(const TranArgs& a, int i, int k) const { return k >= 0 ? pf_at[k] : a.atol[i]; }
  __device__ __forceinline__ double emask_of(const TranArgs& a, int i, int k) const { return k >= 0 ? pf_em[k] : a.emask[i]; }
};

// The core of the linear system: the Schur complement of the last NC pivots (accumulated in W by the entry program), one
// row per lane, eliminated and solved in registers.  Pivot rows are broadcast with v_readlane; no LDS traffic and no fences
// inside.  Static pivot order like the rest of the factorisation; a zero / non-finite pivot raises `bad`.
// MODE 0: eliminate and solve; `keep` writes the factors back (multipliers below the diagonal, U above it, the reciprocal pivots on it) for
// later MODE 1 calls, which only carry the right-hand side through them (Newton mode 1: a round on kept factors).
template <int NC, int MODE>
__device__ __forceinline__ void dense_core_solve(double* W, int dn0, int yc0, int lane, int& bad, bool keep) {
  const int row = lane < NC ? lane : 0;
  double* S = W + dn0 + row * NC;
  double A[NC], rp[NC], bc = W[yc0 + row];
#pragma unroll
  for (int j = 0; j < NC; ++j) A[j] = S[j];
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const double pkk = readlane_f64(A[k], k);
    if (MODE == 0 && (pkk == 0.0 || !isfinite(pkk))) bad = 1;
    rp[k] = MODE == 0 ? fast_div(1.0, pkk) : pkk;          // kept factors carry 1 / pivot on the diagonal
    const double m = lane > k ? (MODE == 0 ? A[k] * rp[k] : A[k]) : 0.0;       // rows up to k keep their (final) U rows
    if (MODE == 0) {
#pragma unroll
      for (int j = k + 1; j < NC; ++j) A[j] = fma(-m, readlane_f64(A[j], k), A[j]);
      if (lane > k) A[k] = m;
    }
    bc = fma(-m, readlane_f64(bc, k), bc);
  }
#pragma unroll
  for (int k = NC - 1; k >= 0; --k) {
    const double xk = readlane_f64(bc * rp[k], k);         // lane k holds the reduced rhs of row k
    bc = lane == k ? xk : (lane < k ? fma(-A[k], xk, bc) : bc);
  }
  if (lane < NC) {
    W[yc0 + lane] = bc;
    if (MODE == 0 && keep) {
#pragma unroll
      for (int j = 0; j < NC; ++j) S[j] = j == lane ? rp[j] : A[j];
    }
  }
}

// VAR: 0 = direct residuals, lean device set; 1 = direct residuals, every device type; 2 = assembled residual
// r = J u + C beta - b (diagnostic, CADNIP_F2_NODIRECT=1), every device type
template <int WPB, bool DC, int VAR>
__global__ void __launch_bounds__(64 * WPB) k_fused2(F2Args f) {
  constexpr bool DIRECT = VAR != 2, LEAN = VAR == 0;
  extern __shared__ double sm[];
  // w is the same for all lanes of a wave: say so (readfirstlane), or every address derived from it lives in VGPRs
  const int tid = threadIdx.x, lane0 = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), n = f.n;
  // ---- shared tables: one cooperative copy per launch
  {
    const uint2* src = (const uint2*)(f.tab + f.tab_lo);
    uint2* dst = (uint2*)sm;
    for (int i = tid; i < f.tab_len / 2; i += 64 * WPB) dst[i] = src[i];
  }
  const TranArgs& a = f.t;
  const unsigned* tab = (const unsigned*)sm;
  const int tab_dbl = f.tab_len / 2;
  // lean variant: the linear solve as straight-line steps (f2_program.cpp: f2_build_steps), their 16-byte lane descriptors staged behind the tables
  const uint4* tdesc = (const uint4*)(sm + tab_dbl);
  const int desc_dbl = LEAN ? f.team_desc_len : 0;
  if constexpr (LEAN) {
    const uint2* src = (const uint2*)f.team_desc;
    uint2* dst = (uint2*)(sm + tab_dbl);
    for (int i = tid; i < f.team_desc_len; i += 64 * WPB) dst[i] = src[i];
  }
  __syncthreads();
  const int nW = f.nnz_lu + n + F2_TRASH;                 // LU | rhs | trash : zeroed every round
  const int per = nW + 2 + 2 * n;                         // ... | the steps' constant words 0.0, 1.0 | u | beta
  double* W = sm + tab_dbl + desc_dbl + (size_t)w * per;
  double* us = W + nW + 2;
  double* betas = us + n;
  if (lane0 == 0) { W[nW] = 0.0; W[nW + 1] = 1.0; }
  const int tlo = f.tab_lo;                               // (a section's offset minus tab_lo first: every pointer formed here lies inside the staged range)
  const u16* gpos = (const u16*)(tab + (f.off[S_GPOS] - tlo));
  const u64* cdesc = (const u64*)(tab + (f.off[S_CDESC] - tlo));
  const u16* brow = (const u16*)(tab + (f.off[S_BROW] - tlo));
  const u64* nzd = (const u64*)(tab + (LEAN ? 0 : f.off[S_NZ]));
  const u64* laned = (const u64*)(tab + (LEAN ? 0 : f.off[S_ENT]));
  const unsigned* term = tab + (LEAN ? 0 : f.off[S_TERM]);
  // pass descriptors are wave-uniform: read them with scalar loads from the table's copy in global memory (constant
  // cache) instead of an LDS read plus two v_readfirstlane per pass
  typedef const __attribute__((address_space(4))) u64* PassPtr;
  const PassPtr passd = (PassPtr)(const u64*)(f.tab + f.off[S_LEV]);
  const u16* qinv = (const u16*)(tab + (f.off[S_QINV] - tlo));
  const short* nodes = (const short*)(tab + (f.off[S_NODES] - tlo));
  const u16* rowof = (const u16*)(tab + (f.off[S_ROWOF] - tlo));
  // Pinned stamp targets (DIRECT variants): the first capacitor / resistor block (two devices per lane) and the first
  // independent-source block (one per lane) are stamped from registers -- four matrix words, the residual words of
  // their rows and their unknowns, as 16-bit offsets.  They depend on the circuit only, so they are read once per
  // launch; a lane without a device points everything at its trash word / ground.
  unsigned rc_gp[2][2] = {{0, 0}, {0, 0}}, rc_row[2] = {0, 0}, rc_nd[2] = {0, 0}, src_gp[2] = {0, 0}, src_row[2] = {0, 0}, src_nd[2] = {0, 0};
  int rc_count = 0, rc_type = 0, src_count = 0, src_type = 0;
  if constexpr (DIRECT) {
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
  // A wave works through instances one after the other: its first one by position, further ones from a queue shared
  // by the grid (instances beyond the resident waves, in index order).  `budget` caps the wave's rounds per launch.
  int inst = blockIdx.x * WPB + w;
  int budget = f.rounds;
  for (;;) {
  StepState st;
  bool have = false;
  while (inst < f.B) {
    const TranStateView sv = state_view();
    if (DC) st.status = __builtin_amdgcn_readfirstlane(sv.status[inst]);
    else { st = load_state(sv, inst); make_uniform(st); }
    if (st.status == 0) { have = true; break; }
    int nx = 0;
    if (lane0 == 0) nx = atomicAdd(kargs()->queue, 1);
    inst = (int)gridDim.x * WPB + __builtin_amdgcn_readfirstlane(nx);
  }
  if (!have || budget <= 0) break;
  const size_t vo = (size_t)inst * n;
  // limit_w is read only by the PCNR corrector of the update: without it the stamps need not write it (each write is an
  // HBM store that later vector-memory waits would queue behind)
  double* lw = (DC ? f.dc_pcnr : a.use_pcnr) ? kargs()->t.limit_w + vo : nullptr;
  FusedVecs vec{us, betas, W, qinv, vo, lw};
  if (!DC) vec.load_history(n, lane0);
  {
    const F2ArgsK ka = kargs();
    const double *ug = ka->t.u + vo, *betag = ka->t.beta + vo;
    for (int i = lane0; i < n; i += 64) { us[i] = ug[i]; betas[i] = DC ? 0.0 : betag[i]; }
  }
  // Newton mode 1: the factors this instance kept when it last left a wave
  const bool mn = !DC && LEAN && a.newton_mode != 0;     // (lean variant only: the host refuses the mode otherwise, fused2.hip)
  if (mn && (st.mflags & MN_VALID)) {
    const double* src = kargs()->lufac + (size_t)inst * f.nnz_lu;
    for (int i = lane0; i < f.nnz_lu; i += 64) W[i] = src[i];
  }
  // DC state of this instance: settle flag of the PCNR loop (solve.jl:640-657), Newton solves done in this launch
  int dc_state = 0, dc_iters = 0, dc_first = 0;
  if (DC) { dc_state = __builtin_amdgcn_readfirstlane(kargs()->dcstate[inst]); dc_first = f.dc_initjct && __builtin_amdgcn_readfirstlane(kargs()->cold[inst]); }
  double rc_val[2] = {0.0, 0.0};   // values of the first capacitor / resistor block: constant for the instance, kept in registers
  if (f.rc_blk >= 0) {
    const F2Block B = load_block(f.blk, f.rc_blk);
    const double* par = B.par + (size_t)inst * B.n_par * B.count;
#pragma unroll
    for (int q = 0; q < 2; ++q) { const int dev = lane0 + 64 * q; rc_val[q] = par[dev < B.count ? dev : 0]; }
  }
  // values of the first independent-source block: functions of time only, kept over the Newton rounds of a time point
  double src_val = 0.0, src_t = 0.0;
  bool src_have = false;
  int src_seg = 0;             // PWL segment of the last evaluation (devices.hpp: pwl_at_time)
  for (; budget > 0; --budget) {
    // Addresses derived from the lane id are loop invariant; hoisted out of the round loop they would have to live in
    // (and spill from) vector registers for the whole instance.  An opaque copy per round keeps them local.
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    CADNIP_TRACE_POINT(17);
    // Newton mode 1: does this round refactor (IDA's lsetup conditions, tran_ctrl.hpp)?  Otherwise the factors in W stay and only
    // the right-hand side (and the trash words) are cleared.
    bool refresh = true;
    if (mn) {
      refresh = (st.mflags & MN_NEED) || !(st.mflags & MN_VALID) ||
                (st.k == 0 && (st.a0 < 0.6 * st.a0f || st.a0 * 0.6 > st.a0f || (st.mflags >> MN_SINCE_SHIFT) >= 20));
      if (refresh) { st.a0f = st.a0; st.ss = 20.0; st.mflags = MN_VALID | MN_JCUR; st.dsc = 1.0; }
      else st.dsc = st.a0 == st.a0f ? 1.0 : fast_div(2.0, 1.0 + fast_div(st.a0, st.a0f));
    }
    if (refresh) {
      for (int i = lane; i < (nW >> 1); i += 64) ((double2*)W)[i] = make_double2(0.0, 0.0);   // nW is even (f2_program.cpp), W 16-byte aligned
    } else {
      for (int i = f.nnz_lu + lane; i < nW; i += 64) W[i] = 0.0;
    }
    CADNIP_WAVE_SYNC();
    CADNIP_TRACE_POINT(0);
    // ---- stamp: accumulate J (at LU positions) and the C*beta - b part of the residual
    const double tcur = DC ? 0.0 : st.tn, a0 = DC ? 0.0 : st.a0;
    const int dmode = DC ? f.dc_mode : 1, dinit = DC ? dc_first : 0;
    const unsigned trash_w = (unsigned)(f.nnz_lu + n + lane);
    dc_first = 0;                                       // initjct is armed for the first stamping only (solve.jl:624,632)
    if constexpr (DIRECT) {
      // ---- pinned blocks: no table reads, no block header -- operand reads, then the atomics
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
          if (q == 1 && rc_count <= 64) break;             // (the second device of a lane exists only beyond 64: its stamps would all go to the trash word)
          if (refresh) {
            atomicAdd(&W[rc_gp[q][0] & 0xFFFFu], jv[q]); atomicAdd(&W[rc_gp[q][0] >> 16], -jv[q]);
            atomicAdd(&W[rc_gp[q][1] & 0xFFFFu], -jv[q]); atomicAdd(&W[rc_gp[q][1] >> 16], jv[q]);
          }
          atomicAdd(&W[rc_row[q] & 0xFFFFu], cur[q]); atomicAdd(&W[rc_row[q] >> 16], -cur[q]);
        }
      }
      if (f.src_blk >= 0) {
        if (DC || !src_have || tcur != src_t) {
          const F2Block B = load_block(f.blk, f.src_blk);
          const double* par = B.par + (size_t)inst * B.n_par * B.count;
          LdsCtx d{nodes + B.nodes_off, B.ipar, par, f.wave, B.count, lane < B.count ? lane : 0, tcur, dmode, dinit};
          src_val = source_value(d, par_of(d, 0), par_of(d, 1), &src_seg);
          src_t = tcur; src_have = true;
        }
        if (src_type == CADNIP_DEV_VSOURCE) {
          // branch rows / columns +-1 (devices.hpp: branch4); KCL rows carry u[I], the branch row V(p) - V(n) - v
          const double ui = at(src_nd[1]), vd = at(src_nd[0] & 0xFFFFu) - at(src_nd[0] >> 16) - src_val;
          if (refresh) {
            atomicAdd(&W[src_gp[0] & 0xFFFFu], 1.0); atomicAdd(&W[src_gp[0] >> 16], -1.0);
            atomicAdd(&W[src_gp[1] & 0xFFFFu], 1.0); atomicAdd(&W[src_gp[1] >> 16], -1.0);
          }
          atomicAdd(&W[src_row[0] & 0xFFFFu], ui); atomicAdd(&W[src_row[0] >> 16], -ui);
          atomicAdd(&W[src_row[1]], vd);
        } else {
          atomicAdd(&W[src_row[0] & 0xFFFFu], -src_val); atomicAdd(&W[src_row[0] >> 16], src_val);
        }
      }
      CADNIP_TRACE_POINT(13);
    }
    for (int bi = 0; bi < f.n_blk; ++bi) {
      if (DIRECT && ((bi == f.rc_blk && rc_count <= 128) || (bi == f.src_blk && src_count <= 64))) continue;   // all of it was pinned
      const F2Block B = load_block(f.blk, bi);
      const double* par = B.par + (size_t)inst * B.n_par * B.count;
      int dev0 = lane;
      if (DIRECT && bi == f.rc_blk) dev0 = lane + 128;
      if (DIRECT && bi == f.src_blk) dev0 = lane + 64;
      if (!DIRECT && bi == f.rc_blk) {
        // first capacitor / resistor block: its (round-invariant) values were fetched once per instance
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int dev = lane + 64 * q;
          if (dev < B.count) {
            AccumOutT<false, DIRECT> s{W, betas, a0, gpos + B.g_base, cdesc + B.c_base, brow + B.b_base, B.count, dev, 0u, us, rowof, trash_w};
            if (B.type == CADNIP_DEV_CAPACITOR) capacitance4(s, 0, rc_val[q]); else conductance4(s, 0, rc_val[q]);
            if (DIRECT) {
              const short* nd = nodes + B.nodes_off;
              const int np = nd[dev], nn = nd[B.count + dev];
              residual2(s, np, nn, B.type == CADNIP_DEV_CAPACITOR ? rc_val[q] * (s.du(np) - s.du(nn)) : rc_val[q] * (volt(us, np) - volt(us, nn)));
            }
          }
        }
        dev0 = lane + 128;
      }
      if (!DIRECT && bi == f.src_blk) {
        const bool on = lane < B.count;
        LdsCtx d{nodes + B.nodes_off, B.ipar, par, f.wave, B.count, on ? lane : 0, tcur, dmode, dinit};
        if (DC || !src_have || tcur != src_t) {
          src_val = source_value(d, par_of(d, 0), par_of(d, 1), &src_seg);
          src_t = tcur; src_have = true;
        }
        if (on) {
          AccumOutT<false, DIRECT> s{W, betas, a0, gpos + B.g_base, cdesc + B.c_base, brow + B.b_base, B.count, lane, 0u, us, rowof, trash_w};
          if (B.type == CADNIP_DEV_VSOURCE) stamp_vsource_value(d, us, s, src_val); else stamp_isource_value(d, s, src_val);
        }
        dev0 = lane + 64;
      }
      if (B.type == CADNIP_DEV_MOS1 && B.mos1_plain) {
        // two lanes per MOSFET (devices.hpp: stamp_mos1_pair), 32 devices per wave pass
        const int side = lane & 1;
        for (int d0 = 0; d0 < B.count; d0 += 32) {
          const int dv = d0 + (lane >> 1);
          const bool valid = dv < B.count;
          LdsCtx d{nodes + B.nodes_off, B.ipar, par, f.wave, B.count, valid ? dv : B.count - 1, tcur, dmode, dinit};
          if (LEAN && !DC && !refresh) {       // round on kept factors: residuals only
            AccumOutT<true, DIRECT, !LEAN || DC> s{W, betas, a0, gpos + B.g_base, cdesc + B.c_base, brow + B.b_base, B.count, d.dev, valid ? 0u : trash_w, us, rowof, trash_w};
            stamp_mos1_pair(d, us, s, lw, side, valid);
          } else {
            AccumOutT<true, DIRECT> s{W, betas, a0, gpos + B.g_base, cdesc + B.c_base, brow + B.b_base, B.count, d.dev, valid ? 0u : trash_w, us, rowof, trash_w};
            stamp_mos1_pair(d, us, s, lw, side, valid);
          }
        }
        dev0 = B.count;
      }
      for (int dev = dev0; dev < B.count; dev += 64) {
        LdsCtx d{nodes + B.nodes_off, B.ipar, par, f.wave, B.count, dev, tcur, dmode, dinit};
        if (LEAN && !DC && !refresh) {
          AccumOutT<false, DIRECT, !LEAN || DC> s{W, betas, a0, gpos + B.g_base, cdesc + B.c_base, brow + B.b_base, B.count, dev, 0u, us, rowof, trash_w};
          dispatch_stamp2<LEAN>(B.type, d, us, s, lw);
        } else {
          AccumOutT<false, DIRECT> s{W, betas, a0, gpos + B.g_base, cdesc + B.c_base, brow + B.b_base, B.count, dev, 0u, us, rowof, trash_w};
          dispatch_stamp2<LEAN>(B.type, d, us, s, lw);
        }
      }
      CADNIP_TRACE_POINT(8 + bi);
    }
    CADNIP_WAVE_SYNC();
    CADNIP_TRACE_POINT(1);
    if (!DC) vec.prefetch(a, lane);   // HBM operands of the update: in flight while the linear solve runs out of LDS
    // ---- r += J*u  (J still unfactored in the LU array).  F2_JU entries per lane in flight: all descriptor reads, then
    // all operand reads, then all atomics -- three LDS round trips per chunk of 64 * F2_JU entries (one chunk on the DFF)
    for (int p0 = 0; !DIRECT && p0 < f.nnz; p0 += 64 * F2_JU) {
      u64 d[F2_JU];
      double v[F2_JU];
#pragma unroll
      for (int q = 0; q < F2_JU; ++q) { const int p = p0 + q * 64 + lane; d[q] = nzd[p < f.nnz ? p : f.nnz - 1]; }
#pragma unroll
      for (int q = 0; q < F2_JU; ++q) v[q] = W[(unsigned)d[q] & 0xFFFFu] * us[(unsigned)(d[q] >> 32) & 0xFFFFu];
#pragma unroll
      for (int q = 0; q < F2_JU; ++q) {
        const bool ok = p0 + q * 64 + lane < f.nnz;
        atomicAdd(&W[ok ? ((unsigned)(d[q] >> 16) & 0xFFFFu) : (unsigned)(f.nnz_lu + n + lane)], ok ? v[q] : 0.0);
      }
    }
    CADNIP_WAVE_SYNC();
    CADNIP_TRACE_POINT(2);
    if (DC) {
      // ---- k_dc_check (driver.hip) on F = G u - b, which is what the rhs words hold now
      double s2 = 0.0;
      int nonfinite = 0;
      for (int i = lane; i < n; i += 64) { const double fv = W[f.nnz_lu + i]; if (!isfinite(fv)) nonfinite = 1; s2 += fv * fv; }
      s2 = wave_sum(s2);
      nonfinite = wave_any(nonfinite);
      const bool pc = f.dc_pcnr && a.n_limits > 0;
      int action = 0;
      const long long done = kargs()->t.cnt[(size_t)inst * 4] + dc_iters;
      if (nonfinite) st.status = -1;
      else if (pc && dc_state == 0 && done >= f.dc_maxiters) st.status = -3;   // PCNR: no residual test after the last solve (driver.hip: k_dc_check)
      else if (sqrt(s2) < f.dc_abstol) {
        if (!pc) st.status = 1;
        else if (dc_state == 0) {                         // settle the limit slots, verify on the next stamping
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
          for (int i = n - a.n_limits + lane; i < n; i += 64) us[i] = lw[i];
          dc_state = 1; action = 1;
        } else st.status = 1;
      } else dc_state = 0;
      if (st.status == 0 && !action && done >= f.dc_maxiters) st.status = -3;
      CADNIP_WAVE_SYNC();
      if (st.status != 0) { --budget; break; }
      if (action) continue;
    }
    // ---- refactor + forward + backward substitution: one entry-wise program, executed pass by pass.  A pass gives
    // every lane one descriptor (entry, its share of the entry's terms, the width 2^lg of the entry's lane group);
    // a dependency level is one or more passes and ends with a fence.  Software pipeline: pass descriptors are
    // fetched two passes ahead, lane descriptor and first term one pass ahead, so the chain inside a pass is
    // operand reads -> fma -> DPP sum -> [divide] -> write.
    int bad = 0;
    auto run_passes = [&](const int p_first, const int p_count) {
      if (p_count <= 0) return;
      u64 pd = passd[p_first], pd1 = passd[p_first + 1];
      u64 D;
      unsigned T0;
      {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pd), hi = __builtin_amdgcn_readfirstlane((unsigned)(pd >> 32));
        const int T = hi & 0x7F;
        D = laned[lo + (lane < T ? lane : T - 1)];
        T0 = term[(unsigned)(D >> 32) & 0xFFFFu];
      }
      for (int pi = p_first; pi < p_first + p_count; ++pi) {
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(pd >> 32));
        const int T = hi & 0x7F, maxlg = (hi >> 8) & 7, hasdiv = (hi >> 11) & 1, fence = (hi >> 12) & 1, multi = (hi >> 13) & 1;
        const bool act = lane < T;
        const unsigned pos = (unsigned)D & 0xFFFFu, dg = (unsigned)(D >> 16) & 0xFFFFu, t0 = (unsigned)(D >> 32) & 0xFFFFu;
        const unsigned dhi = (unsigned)(D >> 48);
        const int nt = act ? (int)(dhi & 0xFFu) : 0, lg = (dhi >> 8) & 7;
        const bool leader = act && ((dhi >> 12) & 1u);
        // (1) operands of this pass
        const double acc0 = W[pos];
        double piv = W[dg == NOPOS ? pos : dg];
        const double av = W[T0 & 0xFFFFu], bv = W[T0 >> 16];
        // (2) prefetch for the next passes
        const unsigned lo1 = __builtin_amdgcn_readfirstlane((unsigned)pd1), hi1 = __builtin_amdgcn_readfirstlane((unsigned)(pd1 >> 32));
        const int T1 = hi1 & 0x7F;
        const u64 Dn = laned[lo1 + (lane < T1 ? lane : (T1 > 0 ? T1 - 1 : 0))];
        const u64 pd2 = passd[pi + 2];
        __builtin_amdgcn_sched_barrier(0);
        // (3) dot product share, group sum, finish
        double part = nt > 0 ? av * bv : 0.0;
        const unsigned T0n = term[(unsigned)(Dn >> 32) & 0xFFFFu];   // next pass's first term: its read overlaps the arithmetic below
        __builtin_amdgcn_sched_barrier(0);
        if (multi)
          for (int t = 1; t < nt; ++t) { const unsigned tm = term[t0 + t]; part = fma(W[tm & 0xFFFFu], W[tm >> 16], part); }
        if (maxlg >= 1) { const double o = dpp_f64<0xB1>(part); part += lg >= 1 ? o : 0.0; }
        if (maxlg >= 2) { const double o = dpp_f64<0x4E>(part); part += lg >= 2 ? o : 0.0; }
        if (maxlg >= 3) { const double o = dpp_f64<0x141>(part); part += lg >= 3 ? o : 0.0; }
        if (maxlg >= 4) { const double o = dpp_f64<0x140>(part); part += lg >= 4 ? o : 0.0; }
        double acc = acc0 - part;
        if (hasdiv) {
          if (dg == NOPOS) piv = 1.0;
          else if (act && (piv == 0.0 || !isfinite(piv))) bad = 1;
          acc = fast_div(acc, piv);
        }
        if (leader) W[pos] = acc;
        if (fence) CADNIP_WAVE_SYNC();
        D = Dn; T0 = T0n; pd = pd1; pd1 = pd2;
      }
    };
    // lean variant: the same program as straight-line steps -- one 16-byte descriptor per lane and step (entry, pivot, three multiply-add
    // terms; f2_program.cpp: f2_build_steps), the next step's descriptor read with this step's operands; no term lists, no inner loops; the
    // step's widest lane group and whether anything divides are the same bits in every lane's descriptor (scalar branches).  A wave's LDS
    // operations execute in order, so steps need no fence between them.
    auto run_steps = [&](const int s_first, const int s_count) {
      if (s_count <= 0) return;
      const uint4* dp = tdesc + (size_t)s_first * 64 + lane;
      uint4 D = dp[0];
      for (int si = 0; si < s_count; ++si) {
        double* const pp = W + (D.x & 0x7FFFu);
        const double piv = W[(D.x >> 16) & 0x7FFFu];
        const double a0v = W[D.y & 0x7FFFu], b0v = W[(D.y >> 16) & 0x7FFFu];
        const double a1v = W[D.z & 0x7FFFu], b1v = W[(D.z >> 16) & 0x7FFFu];
        const double a2v = W[D.w & 0x7FFFu], b2v = W[(D.w >> 16) & 0x7FFFu];
        const double acc0 = *pp;
        const uint4 Dn = dp[(size_t)(si + 1) * 64];
        const unsigned fz = __builtin_amdgcn_readfirstlane(D.z), fw = __builtin_amdgcn_readfirstlane(D.w);
        const unsigned maxlg = ((fz >> 15) & 1u) | ((fz >> 30) & 2u) | ((fw >> 13) & 4u);
        const unsigned lg = (D.x >> 31) | ((D.y >> 14) & 2u) | ((D.y >> 29) & 4u);
        double part = fma(a2v, b2v, fma(a1v, b1v, a0v * b0v));
        if (maxlg >= 1) { const double o = dpp_f64<0xB1>(part); part += lg >= 1 ? o : 0.0; }
        if (maxlg >= 2) { const double o = dpp_f64<0x4E>(part); part += lg >= 2 ? o : 0.0; }
        if (maxlg >= 3) { const double o = dpp_f64<0x141>(part); part += lg >= 3 ? o : 0.0; }
        if (maxlg >= 4) { const double o = dpp_f64<0x140>(part); part += lg >= 4 ? o : 0.0; }
        double acc = acc0 - part;
        if (fw >> 31) {
          if (piv == 0.0 || !isfinite(piv)) bad = 1;        // (lanes and entries without a division read the constant 1.0)
          acc = fast_div(acc, piv);
        }
        // every lane stores -- the lanes that are not their group's leader into their trash word: no branch
        *((D.x & 0x8000u) ? pp : W + trash_w) = acc;
        D = Dn;
      }
      CADNIP_WAVE_SYNC();
    };
    if constexpr (LEAN) {
      if (refresh) run_steps(0, f.ts_pre);
      else run_steps(f.ts_pre + f.ts_post, f.ts_fwd);       // kept factors: the forward substitution alone
    } else {
      if (refresh) run_passes(0, f.n_pre);
      else run_passes(f.n_pre + f.n_post, f.n_fwd);
    }
    CADNIP_TRACE_POINT(4);
    if (f.nc > 0) {
      const int yc0 = f.nnz_lu + n - f.nc;
      if (refresh) {
        if (f.nc == 8) dense_core_solve<8, 0>(W, f.dn0, yc0, lane, bad, mn);
        else if (f.nc == 12) dense_core_solve<12, 0>(W, f.dn0, yc0, lane, bad, mn);
        else dense_core_solve<F2_NCMAX, 0>(W, f.dn0, yc0, lane, bad, mn);
      } else if (LEAN && !DC) {
        if (f.nc == 8) dense_core_solve<8, 1>(W, f.dn0, yc0, lane, bad, false);
        else if (f.nc == 12) dense_core_solve<12, 1>(W, f.dn0, yc0, lane, bad, false);
        else dense_core_solve<F2_NCMAX, 1>(W, f.dn0, yc0, lane, bad, false);
      }
      CADNIP_WAVE_SYNC();
    }
    CADNIP_TRACE_POINT(5);
    if constexpr (LEAN) run_steps(f.ts_pre, f.ts_post); else run_passes(f.n_pre, f.n_post);
    CADNIP_TRACE_POINT(3);
    // ---- Newton update + step controller (registers / LDS; HBM only for history and outputs)
    if (DC) {
      // ---- k_dc_update: u -= delta, PCNR corrector u[lim] = limit_w (solve.jl:667-690)
      for (int i = lane; i < n; i += 64) { const double dd = W[qinv[i]]; if (!isfinite(dd)) bad = 1; us[i] -= dd; }
      bad = wave_any(bad);
      CADNIP_WAVE_SYNC();
      if (f.dc_pcnr && a.n_limits > 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // limit_w was stored to HBM by other lanes of this wave
        for (int i = n - a.n_limits + lane; i < n; i += 64) us[i] = lw[i];
      }
      if (bad) st.status = -2; else dc_iters += 1;        // a failed solve is not an iteration (driver.hip: k_dc_update)
    } else {
      tran_update_body(a, vec, st, inst, lane, bad);
      make_uniform(st);
    }
    CADNIP_WAVE_SYNC();
    CADNIP_TRACE_POINT(16);
    if (st.status != 0) { --budget; break; }
  }
  {
    const F2ArgsK ka = kargs();
    double* ug = ka->t.u + vo;
    if (DC) {
      for (int i = lane0; i < n; i += 64) ug[i] = us[i];
      if (lane0 == 0) {
        ka->t.status[inst] = st.status; ka->dcstate[inst] = dc_state; ka->t.active[inst] = st.status == 0 ? 1 : 0;
        ka->t.cnt[(size_t)inst * 4] += dc_iters;
      }
    } else {
      double *betag = ka->t.beta + vo, *dug = ka->t.du + vo;
      const double a0 = st.a0;
      for (int i = lane0; i < n; i += 64) { double x = us[i], b = betas[i]; ug[i] = x; betag[i] = b; dug[i] = a0 * x + b; }
      if (mn && (st.mflags & MN_VALID) && st.status == 0) {    // the factors travel with the instance
        double* dst = ka->lufac + (size_t)inst * f.nnz_lu;
        for (int i = lane0; i < f.nnz_lu; i += 64) dst[i] = W[i];
      }
      vec.store_history(n, lane0);
      store_state(state_view(), inst, lane0, st);
    }
  }
  CADNIP_WAVE_SYNC();
  if (st.status == 0) break;                // out of budget in the middle of this instance: the next launch resumes it
  int nx = 0;
  if (lane0 == 0) nx = atomicAdd(kargs()->queue, 1);
  inst = (int)gridDim.x * WPB + __builtin_amdgcn_readfirstlane(nx);
  }
}


// launch of one instantiation (defined in fused2_v<VAR>.hip); shmem > 64 KB is enabled there
template <int VAR> int f2_launch_variant(int wpb, bool dc, int grid, size_t shmem, hipStream_t stream, const F2Args& f);

}  // namespace cadnip
