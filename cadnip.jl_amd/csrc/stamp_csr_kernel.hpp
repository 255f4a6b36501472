// stamp_csr_kernel.hpp -- the per-op stamping kernel template (see stamp_csr.hip for the design): shared by stamp_csr.hip (built-in
// device types) and the generated translation units of the external Verilog-A models (va_ext/<module>.hip, one kernel each).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "devices.hpp"
#include "internal.hpp"
#include "tran_ctrl.hpp"   // CADNIP_WAVE_SYNC

namespace cadnip {

#define N_CLS 5   // record classes of the reduction (k_stamp_csr); 5 = padding step
enum { TGT_STORE = 0, TGT_RMW = 1, TGT_ATOMIC = 2, TGT_PARTIAL = 3 };   // bits 30-31 of a target word; bits 28-29: 0 G, 1 C, 2 b; bits 0-27: index
                                                                       // (TGT_PARTIAL: index = LDS scratch word of the tile, relative to the tile)

// LDS staging writer: slot (k, dev) of this lane's instance tile; same interface as SlotOut (devices.hpp)
// REMAP: the block's plan packs its rows (sp_mos1 and the generated models: many slots, many of them without a target); the small device types
// keep one row per slot, addressed arithmetically -- a table look-up per staged value and one more memory round trip at the start of a wave that
// lives for ten microseconds cost their kernels 30 - 60 %
template <bool REMAP>
struct LdsOutT {
  static constexpr bool DIRECT = false;
  __device__ __forceinline__ void Rn(int, double) const {}
  __device__ __forceinline__ double du(int) const { return 0.0; }
  double* tile;                          // this lane's instance tile
  const unsigned short *rg, *rc, *rb;    // LDS: word offset of the row of G / C / b slot k inside a tile (build_stamp_plan: the rows some target reads are
                                         // packed; every other slot -- ground rows / columns, the unused form of a reactive branch -- shares one trash row)
  int ldev;                              // this lane's device within the tile
  bool on;                               // false: no device behind this lane (or an inactive instance)
  int cs, og, oc, ob;                    // !REMAP: devices per row, first row of the G / C / b slots
  __device__ __forceinline__ void G(int k, double v) const { if (on) tile[(REMAP ? (int)rg[k] : (og + k) * cs) + ldev] = v; }
  __device__ __forceinline__ void C(int k, double v) const { if (on) tile[(REMAP ? (int)rc[k] : (oc + k) * cs) + ldev] = v; }
  __device__ __forceinline__ void B(int k, double v) const { if (on) tile[(REMAP ? (int)rb[k] : (ob + k) * cs) + ldev] = v; }
  template <int N> __device__ __forceinline__ void Gv(int k0, const double (&v)[N]) const {
#pragma unroll
    for (int i = 0; i < N; ++i) G(k0 + i, v[i]);
  }
  template <int N> __device__ __forceinline__ void Gk(const int (&k)[N], const double (&v)[N]) const {
#pragma unroll
    for (int i = 0; i < N; ++i) G(k[i], v[i]);
  }
  template <int N> __device__ __forceinline__ void Cv(int k0, const double (&v)[N]) const {
#pragma unroll
    for (int i = 0; i < N; ++i) C(k0 + i, v[i]);
  }
  template <int N> __device__ __forceinline__ void Bv(int k0, const double (&v)[N]) const {
#pragma unroll
    for (int i = 0; i < N; ++i) B(k0 + i, v[i]);
  }
};

struct CsrStampArgs {
  const int* nodes; const int* ipar; const double* par; const double* wave;
  const double* u; const double* t; const int* active; const int* cold;
  double *G, *C, *b, *limit_w; int* nonfinite;
  const unsigned char* diag_flag; const double* gshunt; const double* srcFact;
  const int* step_ptr; const int* step_info; const uint4* tgt_rec;   // [n_chunks + 1] step ranges, per step class | new-level flag << 8, STEP_W records (16 B each) per step (build_stamp_plan)
  int B, count, n, nnz, n_par, n_g, n_c, n_b, cs, n_chunks, ipw, lpd, mode, initjct, zero_first, n_levels, n_scratch;
  int u_lds;   // the unknowns of the tile's instances are staged in LDS (small circuits): node voltages are then LDS reads
  const double* cache; int n_cache;                 // generated external models: setup-pass results [B][n_cache][count] (k_va_setup)
  double* dump; int ns, dump_g, dump_c, dump_b;   // operating-point read-out only (cadnip_get_contributions): the staged per-device
                                                    // contributions written out as [B][ns], slot (k, dev) of array A at A_base + k * count + dev
  const unsigned short* rowoff; int n_rows;        // row offsets of the slots (null: identity, n_rows = all slots), rows of a tile
  int dump_only;                                    // operating-point read-out pass: stage with the identity layout, write `dump`, skip the reduction

};

#ifdef CADNIP_TRACE
// diagnostic build: cycles between the phases of the sp_mos1 kernel, summed over all waves and launches
static __device__ unsigned long long g_sc_sum[8], g_sc_cnt;
#define SC_POINT(id) do { if (TYPE == CADNIP_DEV_MOS1) { unsigned long long _t = clock64(); if (threadIdx.x == 0) atomicAdd(&g_sc_sum[id], _t - sc_last); sc_last = _t; } } while (0)
#else
#define SC_POINT(id) do {} while (0)
#endif

#define STEP_W 128   // records per reduction step (two per lane)
#define PIPE 8       // steps whose records are in flight / in registers

// EXT: void for the built-in device types; for a generated external model (va_ext/<module>.hip) a struct with
//   template <class Ctx, class Out> static __device__ void stamp(const Ctx&, const double* u, const Out&, double* lw, int dir);
// the kernel is then that model's own (its register budget is not the worst model's).
#ifndef CADNIP_STAMP_KERNEL_ATTR
#define CADNIP_STAMP_KERNEL_ATTR          // a generated unit may ask for a register budget of its own (e.g. amdgpu_waves_per_eu)
#endif
template <int TYPE, class EXT = void>
__global__ void __launch_bounds__(64) CADNIP_STAMP_KERNEL_ATTR k_stamp_csr(CsrStampArgs a) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
#ifdef CADNIP_TRACE
  unsigned long long sc_last = clock64();
  if (TYPE == CADNIP_DEV_MOS1 && threadIdx.x == 0) atomicAdd(&g_sc_cnt, 1ull);
#endif
  const int chunk = blockIdx.x % a.n_chunks, grp = blockIdx.x / a.n_chunks;
  // ---- the reduction's records.  The chunk's records form *steps* of STEP_W = 128 (two per lane), every step homogeneous
  // in class and level (build_stamp_plan pads).  The reduce loop is ROLLED -- this kernel runs once per wave, so every
  // instruction it executes is an instruction-cache miss waiting to happen, and a 32-fold unrolled loop ran at ~800 cycles
  // per step for that reason alone -- and keeps a shift register of PIPE steps' records: the first PIPE are requested
  // here, before the stamp phase, a new one enters with every step.  step_lane: lane s holds the descriptor of step s
  // (class | first-step-of-a-level flag << 8), read with v_readlane.
  const int s0 = a.step_ptr[chunk], n_steps = a.dump_only ? 0 : a.step_ptr[chunk + 1] - s0;
  const int step_lane = a.step_info[s0 + (lane < n_steps ? lane : 0)];   // steps beyond 64 read their descriptor from memory
  const uint4* recs = a.tgt_rec + (size_t)s0 * STEP_W + lane;
  uint4 pipe[PIPE][2];                                                    // stage p holds the records of step q + p
#pragma unroll
  for (int p = 0; p < PIPE; ++p) {
    const int st = p < n_steps ? p : 0;
    pipe[p][0] = recs[(size_t)st * STEP_W]; pipe[p][1] = recs[(size_t)st * STEP_W + 64];
  }
  const int nslots = a.n_g + a.n_c + a.n_b, tile_words = a.n_rows * a.cs + a.n_scratch;   // staged rows | scratch of the reduction tree
  // lane -> (instance of the tile, device of the chunk, side of a lane pair)
  const int dl = lane / a.lpd, side = lane - dl * a.lpd;
  const int ii = a.n_chunks == 1 ? dl / a.count : 0;
  const int ldev = a.n_chunks == 1 ? dl - ii * a.count : dl;
  const int dev = chunk * a.cs + ldev;
  const int inst = grp * a.ipw + ii;
  const bool lane_on = ii < a.ipw && inst < a.B && dev < a.count && ldev < a.cs;
  const bool valid = lane_on && a.active[inst] != 0;
  const int inst_c = inst < a.B ? inst : a.B - 1;                 // clamped: every lane runs the device code (lane-pair DPP)
  double* tile = lds + (size_t)(ii < a.ipw ? ii : 0) * tile_words;
  if (a.zero_first) {
    // (tile_words is even for every type that needs this: n_g + n_c + n_b of sp_mos1 = 114; an odd tail is covered below)
    const int words = a.ipw * tile_words;
    for (int i = lane; i < (words >> 1); i += 64) ((double2*)lds)[i] = make_double2(0.0, 0.0);
    if ((words & 1) && lane == 0) lds[words - 1] = 0.0;
    CADNIP_WAVE_SYNC();
  }
  if (lane < a.ipw) lds[(size_t)(lane + 1) * tile_words - 1] = 0.0;   // the tile's zero word: operand slots a record does not use
  // per-instance scalars of the tile's instances, behind the tiles: the reduction reads them per instance
  double* inst_par = lds + (size_t)a.ipw * tile_words;             // [ipw][3]: active, gshunt, srcFact
  if (lane < a.ipw) {
    const int i2 = grp * a.ipw + lane;
    const bool in = i2 < a.B;
    inst_par[3 * lane] = in && a.active[i2] ? 1.0 : 0.0;
    inst_par[3 * lane + 1] = in ? a.gshunt[i2] : 0.0;
    inst_par[3 * lane + 2] = in ? a.srcFact[i2] : 1.0;
  }
  double* u_tile = inst_par + 3 * a.ipw;                           // [ipw][n] when u_lds
  constexpr bool REMAP = TYPE == CADNIP_DEV_MOS1 || TYPE == CADNIP_DEV_VA;
  unsigned short* rowoff = (unsigned short*)(u_tile + (a.u_lds ? (size_t)a.ipw * a.n : 0));   // [nslots]
  if (REMAP) {
    for (int i = lane; i < nslots; i += 64) rowoff[i] = a.rowoff ? a.rowoff[i] : (unsigned short)(i * a.cs);
    if (!a.u_lds) CADNIP_WAVE_SYNC();
  }
  if (a.u_lds) {
    for (int r = 0; r < a.ipw; ++r) {
      const int i2 = grp * a.ipw + r < a.B ? grp * a.ipw + r : a.B - 1;
      const double* ug = a.u + (size_t)i2 * a.n;
      for (int i = lane; i < a.n; i += 64) u_tile[(size_t)r * a.n + i] = ug[i];
    }
    CADNIP_WAVE_SYNC();
  }
  SC_POINT(0);
  {
    DevCtx d{a.nodes, a.ipar, a.par + (size_t)inst_c * a.n_par * a.count, a.wave, a.count, dev < a.count ? dev : a.count - 1, a.t[inst_c], a.mode, (a.initjct && a.cold[inst_c]) ? 1 : 0,
             a.cache ? a.cache + (size_t)inst_c * a.n_cache * a.count : nullptr};
    LdsOutT<REMAP> s{tile, rowoff, rowoff + a.n_g, rowoff + a.n_g + a.n_c, ldev, valid, a.cs, 0, a.n_g, a.n_g + a.n_c};
    const double* u = a.u_lds ? u_tile + (size_t)(ii < a.ipw ? ii : 0) * a.n : a.u + (size_t)inst_c * a.n;
    double* lw = valid ? a.limit_w + (size_t)inst_c * a.n : nullptr;
    if (TYPE == CADNIP_DEV_RESISTOR) stamp_resistor(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_CAPACITOR) stamp_capacitor(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_INDUCTOR) stamp_inductor(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_VSOURCE) stamp_vsource(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_ISOURCE) stamp_isource(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_VCVS) stamp_vcvs(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_VCCS) stamp_vccs(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_CCVS) stamp_ccvs(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_CCCS) stamp_cccs(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_DIODE) stamp_diode(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_DIODECAP) stamp_diodecap(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_SIMPLEMOS) stamp_simplemos(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_MOS1) {
      if (a.lpd == 2) stamp_mos1_pair(d, u, s, lw, side, valid);   // two lanes per MOSFET (devices.hpp)
      else stamp_mos1(d, u, s, lw);
    }
    else if (TYPE == CADNIP_DEV_BVSOURCE) stamp_bvsource(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_BISOURCE) stamp_bisource(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_VA) {
      if constexpr (std::is_void<EXT>::value) stamp_va(d, u, s, lw);
      else EXT::stamp(d, u, s, lw, side);                         // external model: lane `side` of the device's group carries direction `side`
    }
  }
  CADNIP_WAVE_SYNC();
  if (a.dump_only && a.dump && valid && side == 0) {
    double* D = a.dump + (size_t)inst * a.ns;
    // (dump_only pass: identity layout, every slot has its own row)
    for (int k = 0; k < a.n_g; ++k) D[a.dump_g + k * a.count + dev] = tile[k * a.cs + ldev];
    for (int k = 0; k < a.n_c; ++k) D[a.dump_c + k * a.count + dev] = tile[(a.n_g + k) * a.cs + ldev];
    for (int k = 0; k < a.n_b; ++k) D[a.dump_b + k * a.count + dev] = tile[(a.n_g + a.n_c + k) * a.cs + ldev];
  }
  SC_POINT(1);
  // ---- segmented reduction: one target per lane and step, contributions summed in COO order out of LDS.  A target is
  // one 16-byte record (destination, count, up to five staging offsets inline); RED_U records per lane are requested
  // before the first is used, so the batch costs one memory latency, not one per target.
  // Step by step: class 0 / 1 / 2 = sole writer of a word of G / C / b (plain store), 3 = partial sum into LDS scratch,
  // 4 = the rest (read-modify-write behind an earlier kernel, atomics between tiles), 5 = padding.  Class, destination and
  // the instance's scalars are wave-uniform, so a record costs five LDS reads issued together (operand slots it does not
  // use point at the tile's zero word), four adds and one store.
  // everything a record needs about its instance, gathered once (not per record: the compiler would re-read kernel
  // arguments and LDS words in every step)
  struct InstCtx { double *Gb, *Cb, *bb; int* nf; double sf, gsh; int tile_off; };   // (the tile is addressed by offset: an LDS pointer inside a struct decays to a generic one)
  auto inst_ctx = [&](int ri) {
    const int rinst = grp * a.ipw + ri;
    InstCtx c;
    c.tile_off = ri * tile_words;
    c.Gb = a.G + (size_t)rinst * a.nnz; c.Cb = a.C + (size_t)rinst * a.nnz; c.bb = a.b + (size_t)rinst * a.n;
    c.nf = a.nonfinite + rinst;
    c.gsh = uniform_f64(inst_par[3 * ri + 1]); c.sf = uniform_f64(inst_par[3 * ri + 2]);
    return c;
  };
  const unsigned char* const diag_flag = a.diag_flag;
  auto reduce_instance = [&](const InstCtx& c, const uint4& ra, const uint4& rb, int cls) {
    double* src = lds + c.tile_off;
    const double a0 = src[ra.y >> 16], a1 = src[ra.z & 0xFFFFu], a2 = src[ra.z >> 16], a3 = src[ra.w & 0xFFFFu], a4 = src[ra.w >> 16];
    const double b0 = src[rb.y >> 16], b1 = src[rb.z & 0xFFFFu], b2 = src[rb.z >> 16], b3 = src[rb.w & 0xFFFFu], b4 = src[rb.w >> 16];
    double acc[2] = {(((a0 + a1) + a2) + a3) + a4, (((b0 + b1) + b2) + b3) + b4};
    const unsigned e[2] = {ra.x & 0x0FFFFFFFu, rb.x & 0x0FFFFFFFu};
    const bool on[2] = {(ra.y & 0xFFFFu) != 0u, (rb.y & 0xFFFFu) != 0u};   // padding records inside a step have count 0
    const unsigned wx[2] = {ra.x, rb.x};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (!on[k]) continue;
      double v = acc[k];
      switch (cls) {                                 // wave-uniform
        case 0:
          if (c.gsh != 0.0 && diag_flag[e[k]]) v += c.gsh;                                     // precompile.jl:529-534
          if (!isfinite(v)) *c.nf = 1;
          c.Gb[e[k]] = v;
          break;
        case 1:
          if (!isfinite(v)) *c.nf = 1;
          c.Cb[e[k]] = v;
          break;
        case 2:
          if (c.sf < 1.0) v *= c.sf;                                                           // precompile.jl:524-527
          if (!isfinite(v)) *c.nf = 1;
          c.bb[e[k]] = v;
          break;
        case 3: src[e[k]] = v; break;
        default: {
          const unsigned arr = (wx[k] >> 28) & 3u, md = wx[k] >> 30;
          if (arr == 2u && c.sf < 1.0) v *= c.sf;
          if (!isfinite(v)) *c.nf = 1;
          double* dst = (arr == 0u ? c.Gb : arr == 1u ? c.Cb : c.bb) + e[k];
          if (md == TGT_RMW) *dst += v;
          else unsafeAtomicAdd(dst, v);
        }
      }
    }
  };
  const bool single = a.ipw == 1;
  const bool single_on = single && grp < a.B && inst_par[0] != 0.0;                 // (ipw == 1: the tile's instance is grp)
  const InstCtx c0 = inst_ctx(0);
  // The loop body is PIPE steps (stage p of the register window serves steps p, p + PIPE, ...): a stage is refilled right
  // after it has been consumed.  Result stores are acknowledged in order with the record reads (one vmcnt counter), so the
  // window also decides how many steps of stores may be in flight: with 4 the loop ran at the store latency / 4 per step.
  auto step = [&](int q, const uint4& ra, const uint4& rb) {
    const int info = q < 64 ? __builtin_amdgcn_readlane(step_lane, q) : a.step_info[s0 + q];
    const int cls = info & 0xFF;
    if (info & 0x100) CADNIP_WAVE_SYNC();          // first step of a level: the partial sums below it are complete
    if (cls >= 5) return;
    if (single) { if (single_on) reduce_instance(c0, ra, rb, cls); return; }
    for (int ri = 0; ri < a.ipw; ++ri)
      if (grp * a.ipw + ri < a.B && inst_par[3 * ri] != 0.0) reduce_instance(inst_ctx(ri), ra, rb, cls);   // wave-uniform
  };
  for (int q0 = 0; q0 < n_steps; q0 += PIPE) {
#pragma unroll
    for (int p = 0; p < PIPE; ++p) {
      const int q = q0 + p;
      const uint4 ra = pipe[p][0], rb = pipe[p][1];
      const int st = q + PIPE < n_steps ? q + PIPE : 0;                              // past the end: a harmless re-read of step 0
      pipe[p][0] = recs[(size_t)st * STEP_W]; pipe[p][1] = recs[(size_t)st * STEP_W + 64];
      if (q < n_steps) step(q, ra, rb);                                              // wave-uniform
    }
  }
  CADNIP_WAVE_SYNC();
  SC_POINT(7);
}


// launch of one instantiation (dynamic LDS above 64 KB is enabled here)
template <int TYPE, class EXT = void>
static inline int launch_stamp_kernel(const CsrStampArgs& a, unsigned grid, size_t shmem, hipStream_t stream) {
  if (shmem > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_stamp_csr<TYPE, EXT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL((k_stamp_csr<TYPE, EXT>), dim3(grid), dim3(64), shmem, stream, a);
  return CADNIP_OK;
}

// Setup pass of a generated external model (setup_va_<module>): every bias-independent statement of the module, one thread per
// (instance, device), whenever the block's parameters change (cadnip_set_params).  What the per-call stamp function reads of it lies
// in the block's cache [B][n_cache][count].
struct VaSetupArgs { const int* nodes; const int* ipar; const double* par; const double* wave; double* cache; int B, count, n_par, n_cache, mode; };
template <class EXT>
__global__ void __launch_bounds__(64) k_va_setup(VaSetupArgs a) {
  const int idx = blockIdx.x * 64 + threadIdx.x;
  if (idx >= a.B * a.count) return;
  const int inst = idx / a.count, dev = idx - inst * a.count;
  DevCtx d{a.nodes, a.ipar, a.par + (size_t)inst * a.n_par * a.count, a.wave, a.count, dev, 0.0, a.mode, 0, nullptr};
  EXT::setup(d, a.cache + (size_t)inst * a.n_cache * a.count + dev, a.count);
}
template <class EXT>
static inline int launch_va_setup_kernel(const VaSetupArgs& a, hipStream_t stream) {
  const int total = a.B * a.count;
  hipLaunchKernelGGL((k_va_setup<EXT>), dim3((total + 63) / 64), dim3(64), 0, stream, a);
  return CADNIP_OK;
}

}  // namespace cadnip
