// lu_f2.hip -- per-op refactor + solve (launch_factor_solve: the KLU step of a per-op Newton round, sweeps.jl:600 /
// solve.jl:667-670) executed with the fused kernel's entry program: J = G + gamma C scattered into a per-wave work array
// in LDS, the unified factor / forward / backward program of f2_program.cpp run pass by pass (lane-packed entries, DPP
// group sums, the dense core in registers), x gathered back.  One wave per sweep instance, LU_WPB instances per workgroup
// sharing ONE copy of the program tables in LDS -- the first per-op LU (k_lu, kernels.hip) fetched every level's indices
// from global memory behind three dependent loads and took 52 us for the DFF at B = 1024.
// k_lu remains for what this program cannot do: factor-only / solve-only (the callback ABI's cadnip_factor /
// cadnip_solve keep the factors in HBM) and circuits whose program does not fit.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdlib.h>
#include "fused2_kernel.hpp"

namespace cadnip {

struct LuF2Args {
  const unsigned* tab; int off[S_NSEC]; int tab_len;
  int tab_lo;                                   // k_lu_f2s: the staged table range starts at this word (load map, permutations)
  const uint4* steps; int steps_len, ts_pre, ts_post;   // k_lu_f2s: the one-wave step program (f2_build_steps), 16-byte lane descriptors
  const double *G, *C, *gamma, *rhs; double* x;
  const int* active; int* flags;
  int B, n, nnz, lu_words, n_pre, n_post, nc, dn0;
};

template <int WPB>
__global__ void __launch_bounds__(64 * WPB) k_lu_f2(LuF2Args f) {
  extern __shared__ double sm[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), n = f.n;
  {
    const uint2* src = (const uint2*)f.tab;
    uint2* dst = (uint2*)sm;
    for (int i = tid; i < f.tab_len / 2; i += 64 * WPB) dst[i] = src[i];
  }
  __syncthreads();
  const int inst = blockIdx.x * WPB + w;
  if (inst >= f.B || !f.active[inst]) return;
  const unsigned* tab = (const unsigned*)sm;
  const int nW = f.lu_words + n + F2_TRASH;
  double* W = sm + f.tab_len / 2 + (size_t)w * nW;
  const u16* loadpos = (const u16*)(tab + f.off[S_LOADPOS]);
  const u64* laned = (const u64*)(tab + f.off[S_ENT]);
  const unsigned* term = tab + f.off[S_TERM];
  typedef const __attribute__((address_space(4))) u64* PassPtr;
  const PassPtr passd = (PassPtr)(const u64*)(f.tab + f.off[S_LEV]);
  const u16* qinv = (const u16*)(tab + f.off[S_QINV]);
  const u16* rowof = (const u16*)(tab + f.off[S_ROWOF]);
  for (int i = lane; i < (nW >> 1); i += 64) ((double2*)W)[i] = make_double2(0.0, 0.0);   // nW is even (f2_program.cpp)
  CADNIP_WAVE_SYNC();
  {
    // J = G + gamma C at its L\U positions (every pattern entry has its own word: plain stores), rhs in pivot-row order
    const double* G = f.G + (size_t)inst * f.nnz;
    const double* C = f.C + (size_t)inst * f.nnz;
    const double gam = f.gamma[inst];
    for (int e = lane; e < f.nnz; e += 64) W[loadpos[e]] = G[e] + gam * C[e];   // coalesced reads of the CSR arrays
    const double* rhs = f.rhs + (size_t)inst * n;
    for (int i = lane; i < n; i += 64) W[rowof[i]] = rhs[i];
  }
  CADNIP_WAVE_SYNC();
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
      const double acc0 = W[pos];
      double piv = W[dg == NOPOS ? pos : dg];
      const double av = W[T0 & 0xFFFFu], bv = W[T0 >> 16];
      const unsigned lo1 = __builtin_amdgcn_readfirstlane((unsigned)pd1), hi1 = __builtin_amdgcn_readfirstlane((unsigned)(pd1 >> 32));
      const int T1 = hi1 & 0x7F;
      const u64 Dn = laned[lo1 + (lane < T1 ? lane : (T1 > 0 ? T1 - 1 : 0))];
      const u64 pd2 = passd[pi + 2];
      double part = nt > 0 ? av * bv : 0.0;
      const unsigned T0n = term[(unsigned)(Dn >> 32) & 0xFFFFu];
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
  run_passes(0, f.n_pre);
  if (f.nc > 0) {
    const int yc0 = f.lu_words + n - f.nc;
    if (f.nc == 8) dense_core_solve<8, 0>(W, f.dn0, yc0, lane, bad, false);
    else if (f.nc == 12) dense_core_solve<12, 0>(W, f.dn0, yc0, lane, bad, false);
    else dense_core_solve<F2_NCMAX, 0>(W, f.dn0, yc0, lane, bad, false);
    CADNIP_WAVE_SYNC();
  }
  run_passes(f.n_pre, f.n_post);
  double* x = f.x + (size_t)inst * n;
  for (int i = lane; i < n; i += 64) { const double v = W[qinv[i]]; if (!isfinite(v)) bad = 1; x[i] = v; }
  if (wave_any(bad) && lane == 0) atomicOr(&f.flags[inst], 1);
}

// The same kernel on the fused sweep kernel's STEP program (f2_program.cpp: f2_build_steps; fused2_kernel.hpp: run_steps): straight-line
// steps with three terms per lane, list-scheduled -- 7 + 6 steps instead of 9 + 8 passes on the flip-flop, and no term lists.  Staged in LDS:
// the load map and the permutations of the table, the step descriptors, the instances' work arrays.  Taken when the descriptors fit (a long
// dependency chain has one step per link: the pass program is the compact form).
template <int WPB>
__global__ void __launch_bounds__(64 * WPB) k_lu_f2s(LuF2Args f) {
  extern __shared__ double sm[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), n = f.n;
  {
    const uint2* src = (const uint2*)(f.tab + f.tab_lo);
    uint2* dst = (uint2*)sm;
    for (int i = tid; i < f.tab_len / 2; i += 64 * WPB) dst[i] = src[i];
    const uint2* s2 = (const uint2*)f.steps;
    uint2* d2 = (uint2*)(sm + f.tab_len / 2);
    for (int i = tid; i < f.steps_len; i += 64 * WPB) d2[i] = s2[i];
  }
  __syncthreads();
  const int inst = blockIdx.x * WPB + w;
  if (inst >= f.B || !f.active[inst]) return;
  const unsigned* tab = (const unsigned*)sm;
  const uint4* tdesc = (const uint4*)(sm + f.tab_len / 2);
  const int nW = f.lu_words + n + F2_TRASH;
  double* W = sm + f.tab_len / 2 + f.steps_len + (size_t)w * (nW + 2);
  const u16* loadpos = (const u16*)(tab + (f.off[S_LOADPOS] - f.tab_lo));
  const u16* qinv = (const u16*)(tab + (f.off[S_QINV] - f.tab_lo));
  const u16* rowof = (const u16*)(tab + (f.off[S_ROWOF] - f.tab_lo));
  for (int i = lane; i < (nW >> 1); i += 64) ((double2*)W)[i] = make_double2(0.0, 0.0);   // nW is even (f2_program.cpp)
  if (lane == 0) { W[nW] = 0.0; W[nW + 1] = 1.0; }                                        // the steps' constant words
  CADNIP_WAVE_SYNC();
  {
    const double* G = f.G + (size_t)inst * f.nnz;
    const double* C = f.C + (size_t)inst * f.nnz;
    const double gam = f.gamma[inst];
    for (int e = lane; e < f.nnz; e += 64) W[loadpos[e]] = G[e] + gam * C[e];
    const double* rhs = f.rhs + (size_t)inst * n;
    for (int i = lane; i < n; i += 64) W[rowof[i]] = rhs[i];
  }
  CADNIP_WAVE_SYNC();
  int bad = 0;
  const unsigned trash_w = (unsigned)(f.lu_words + n + lane);
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
        if (piv == 0.0 || !isfinite(piv)) bad = 1;
        acc = fast_div(acc, piv);
      }
      *((D.x & 0x8000u) ? pp : W + trash_w) = acc;
      D = Dn;
    }
    CADNIP_WAVE_SYNC();
  };
  run_steps(0, f.ts_pre);
  if (f.nc > 0) {
    const int yc0 = f.lu_words + n - f.nc;
    if (f.nc == 8) dense_core_solve<8, 0>(W, f.dn0, yc0, lane, bad, false);
    else if (f.nc == 12) dense_core_solve<12, 0>(W, f.dn0, yc0, lane, bad, false);
    else dense_core_solve<F2_NCMAX, 0>(W, f.dn0, yc0, lane, bad, false);
    CADNIP_WAVE_SYNC();
  }
  run_steps(f.ts_pre, f.ts_post);
  double* x = f.x + (size_t)inst * n;
  for (int i = lane; i < n; i += 64) { const double v = W[qinv[i]]; if (!isfinite(v)) bad = 1; x[i] = v; }
  if (wave_any(bad) && lane == 0) atomicOr(&f.flags[inst], 1);
}

// Few instances (a single transient, a handful of corners): one wave per instance leaves the chip empty and walks the passes one after
// the other -- the ring oscillator of BASELINE.json's config 5 (n = 371) has 105 of them in 46 dependency levels.  Here WPI waves share
// one instance: the passes of a level are dealt round-robin to the waves (they touch different entries and read only earlier levels), a
// workgroup barrier closes the level, the dense core is wave 0's.  No software pipeline: with so few waves the kernel is latency-bound anyway.
template <int WPI>
__global__ void __launch_bounds__(64 * WPI) k_lu_f2_mw(LuF2Args f) {
  extern __shared__ double sm[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), n = f.n;
  {
    const uint2* src = (const uint2*)f.tab;
    uint2* dst = (uint2*)sm;
    for (int i = tid; i < f.tab_len / 2; i += 64 * WPI) dst[i] = src[i];
  }
  const int inst = blockIdx.x;
  if (inst >= f.B || !f.active[inst]) return;                 // (uniform over the workgroup: nobody is left waiting at a barrier)
  const unsigned* tab = (const unsigned*)sm;
  const int nW = f.lu_words + n + F2_TRASH;
  double* W = sm + f.tab_len / 2;
  const u16* loadpos = (const u16*)(tab + f.off[S_LOADPOS]);
  const u64* laned = (const u64*)(tab + f.off[S_ENT]);
  const unsigned* term = tab + f.off[S_TERM];
  typedef const __attribute__((address_space(4))) u64* PassPtr;
  const PassPtr passd = (PassPtr)(const u64*)(f.tab + f.off[S_LEV]);
  const u16* qinv = (const u16*)(tab + f.off[S_QINV]);
  const u16* rowof = (const u16*)(tab + f.off[S_ROWOF]);
  for (int i = tid; i < (nW >> 1); i += 64 * WPI) ((double2*)W)[i] = make_double2(0.0, 0.0);
  __syncthreads();
  {
    const double* G = f.G + (size_t)inst * f.nnz;
    const double* C = f.C + (size_t)inst * f.nnz;
    const double gam = f.gamma[inst];
    for (int e = tid; e < f.nnz; e += 64 * WPI) W[loadpos[e]] = G[e] + gam * C[e];
    const double* rhs = f.rhs + (size_t)inst * n;
    for (int i = tid; i < n; i += 64 * WPI) W[rowof[i]] = rhs[i];
  }
  __syncthreads();
  int bad = 0;
  auto run_passes = [&](const int p_first, const int p_count) {
    int in_level = 0;
    for (int pi = p_first; pi < p_first + p_count; ++pi) {
      const u64 pd = passd[pi];
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pd), hi = __builtin_amdgcn_readfirstlane((unsigned)(pd >> 32));
      const int T = hi & 0x7F, maxlg = (hi >> 8) & 7, hasdiv = (hi >> 11) & 1, fence = (hi >> 12) & 1;
      if (in_level % WPI == w && T > 0) {
        const bool act = lane < T;
        const u64 D = laned[lo + (act ? lane : T - 1)];
        const unsigned pos = (unsigned)D & 0xFFFFu, dg = (unsigned)(D >> 16) & 0xFFFFu, t0 = (unsigned)(D >> 32) & 0xFFFFu;
        const unsigned dhi = (unsigned)(D >> 48);
        const int nt = act ? (int)(dhi & 0xFFu) : 0, lg = (dhi >> 8) & 7;
        const bool leader = act && ((dhi >> 12) & 1u);
        const double acc0 = W[pos];
        double piv = W[dg == NOPOS ? pos : dg];
        double part = 0.0;
        for (int t = 0; t < nt; ++t) { const unsigned tm = term[t0 + t]; part = fma(W[tm & 0xFFFFu], W[tm >> 16], part); }
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
      }
      ++in_level;
      if (fence) { __syncthreads(); in_level = 0; }
    }
  };
  run_passes(0, f.n_pre);
  if (f.nc > 0) {
    if (w == 0) {
      const int yc0 = f.lu_words + n - f.nc;
      if (f.nc == 8) dense_core_solve<8, 0>(W, f.dn0, yc0, lane, bad, false);
      else if (f.nc == 12) dense_core_solve<12, 0>(W, f.dn0, yc0, lane, bad, false);
      else dense_core_solve<F2_NCMAX, 0>(W, f.dn0, yc0, lane, bad, false);
    }
    __syncthreads();
  }
  run_passes(f.n_pre, f.n_post);
  double* x = f.x + (size_t)inst * n;
  for (int i = tid; i < n; i += 64 * WPI) { const double v = W[qinv[i]]; if (!isfinite(v)) bad = 1; x[i] = v; }
  if (wave_any(bad) && lane == 0) atomicOr(&f.flags[inst], 1);
}

// The same refactor + solve as straight-line STEPS for a team of four waves per instance (f2_program.cpp: f2_build_steps with nw = 4):
// every thread owns one 16-byte descriptor per step -- word offsets of the entry, its pivot and the factors of up to three multiply-add
// terms; steps are list-scheduled (an entry runs in the first step behind its operands' last writers that has a lane group free), a
// barrier closes each.  The descriptors are per thread: they are fetched from global memory a CHUNK of steps ahead into registers (the
// loads of chunk c + 1 are in flight while chunk c runs), nothing of the program lives in LDS -- only the work array does, so a circuit
// whose tables do not fit beside it (the PSP103 ring: 87 steps x 256 threads; 116 with one term per lane and level-aligned steps) runs
// here all the same.  For few instances -- where k_lu_f2's one wave per instance walks 100+ passes alone.
struct LuStepArgs {
  const uint4* desc; int n_pre, n_post;
  const u16 *loadpos, *rowof, *qinv;              // global copies of the table sections (csr entry -> W word, unknown -> rhs word, unknown -> solution word)
  const double *G, *C, *gamma, *rhs; double* x;
  const int* active; int* flags;
  int B, n, nnz, lu_words, nc, dn0;
};
#define LU_CHUNK 8
template <int NW>
__global__ void __launch_bounds__(64 * NW) k_lu_steps(LuStepArgs f) {
  constexpr int NT = 64 * NW;
  extern __shared__ double sm[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), n = f.n;
  const int inst = blockIdx.x;
  if (inst >= f.B || !f.active[inst]) return;                 // (uniform over the workgroup)
  const int nW = f.lu_words + n + F2_TRASH;
  double* W = sm;
  for (int i = tid; i < (nW >> 1); i += NT) ((double2*)W)[i] = make_double2(0.0, 0.0);
  if (tid == 0) { W[nW] = 0.0; W[nW + 1] = 1.0; }            // the steps' constant words (f2_build_steps)
  // this thread's descriptors of the first chunk: requested before the matrix is loaded
  const uint4* dp = f.desc + tid;
  uint4 cur[LU_CHUNK], nxt[LU_CHUNK];
  const int n_steps = f.n_pre + f.n_post;
#pragma unroll
  for (int k = 0; k < LU_CHUNK; ++k) cur[k] = dp[(size_t)(k < n_steps ? k : 0) * NT];
  __syncthreads();
  {
    const double* G = f.G + (size_t)inst * f.nnz;
    const double* C = f.C + (size_t)inst * f.nnz;
    const double gam = f.gamma[inst];
    for (int e = tid; e < f.nnz; e += NT) W[f.loadpos[e]] = G[e] + gam * C[e];
    const double* rhs = f.rhs + (size_t)inst * n;
    for (int i = tid; i < n; i += NT) W[f.rowof[i]] = rhs[i];
  }
  __syncthreads();
  int bad = 0;
  const unsigned trash_w = (unsigned)(f.lu_words + n + lane);
  auto step = [&](const uint4 D) {
    double* const pp = W + (D.x & 0x7FFFu);
    const double piv = W[(D.x >> 16) & 0x7FFFu];
    const double a0v = W[D.y & 0x7FFFu], b0v = W[(D.y >> 16) & 0x7FFFu];
    const double a1v = W[D.z & 0x7FFFu], b1v = W[(D.z >> 16) & 0x7FFFu];
    const double a2v = W[D.w & 0x7FFFu], b2v = W[(D.w >> 16) & 0x7FFFu];
    const double acc0 = *pp;
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
  };
  // chunks of LU_CHUNK steps; the dense core sits between step n_pre - 1 and step n_pre
  for (int c0 = 0; c0 < n_steps; c0 += LU_CHUNK) {
#pragma unroll
    for (int k = 0; k < LU_CHUNK; ++k) { const int sidx = c0 + LU_CHUNK + k; nxt[k] = dp[(size_t)(sidx < n_steps ? sidx : 0) * NT]; }
#pragma unroll
    for (int k = 0; k < LU_CHUNK; ++k) {
      const int sidx = c0 + k;
      if (sidx == f.n_pre && f.nc > 0) {                     // (uniform)
        if (w == 0) {
          const int yc0 = f.lu_words + n - f.nc;
          if (f.nc == 8) dense_core_solve<8, 0>(W, f.dn0, yc0, lane, bad, false);
          else if (f.nc == 12) dense_core_solve<12, 0>(W, f.dn0, yc0, lane, bad, false);
          else dense_core_solve<F2_NCMAX, 0>(W, f.dn0, yc0, lane, bad, false);
        }
        __syncthreads();
      }
      if (sidx < n_steps) step(cur[k]);
    }
#pragma unroll
    for (int k = 0; k < LU_CHUNK; ++k) cur[k] = nxt[k];
  }
  if (f.n_post == 0 && f.nc > 0) {                            // (no step behind the core: it has not run yet)
    if (w == 0) {
      const int yc0 = f.lu_words + n - f.nc;
      if (f.nc == 8) dense_core_solve<8, 0>(W, f.dn0, yc0, lane, bad, false);
      else if (f.nc == 12) dense_core_solve<12, 0>(W, f.dn0, yc0, lane, bad, false);
      else dense_core_solve<F2_NCMAX, 0>(W, f.dn0, yc0, lane, bad, false);
    }
    __syncthreads();
  }
  double* x = f.x + (size_t)inst * n;
  for (int i = tid; i < n; i += NT) { const double v = W[f.qinv[i]]; if (!isfinite(v)) bad = 1; x[i] = v; }
  if (wave_any(bad) && lane == 0) atomicOr(&f.flags[inst], 1);
}

// 0 = done with the program kernel; 1 = not applicable (the caller falls back to k_lu); < 0 never
int launch_factor_solve_f2(CadnipHandle* h, const double* d_rhs, double* d_x) {
  if (!h->analyzed || !fused2_tables_ready(h)) return 1;   // (only the linear-solve prefix of the tables has to fit: checked below)
  ProfScope ps(h, "lu_factor_solve");
  LuF2Args f;
  f.tab = h->d_f2tab; for (int i = 0; i < S_NSEC; ++i) f.off[i] = h->f2off[i]; f.tab_len = h->f2_lu_len;
  f.tab_lo = 0; f.steps = nullptr; f.steps_len = 0; f.ts_pre = f.ts_post = 0;
  f.G = h->d_G; f.C = h->d_C; f.gamma = h->d_gamma; f.rhs = d_rhs; f.x = d_x; f.active = h->d_active; f.flags = h->d_flags;
  f.B = h->B; f.n = h->n; f.nnz = h->nnz; f.lu_words = h->f2_lu_words; f.n_pre = h->f2_n_pre; f.n_post = h->f2_n_post; f.nc = h->f2_nc; f.dn0 = h->f2_dn0;
  const size_t tab_dbl = (size_t)h->f2_lu_len / 2, per = (size_t)h->f2_lu_words + h->n + F2_TRASH;
  // at most two instances per CU: the straight-line steps of a team of four waves per instance (k_lu_steps).  CADNIP_LU_STEPS = 0 | 1 forces the choice
  {
    const char* e = getenv("CADNIP_LU_STEPS");
    const bool steps = e ? atoi(e) != 0 : h->B <= 2 * h->n_cu_hint();
    const size_t shmem_s = (per + 2) * 8;
    if (steps && h->d_steps4 && shmem_s <= 160 * 1024) {
      LuStepArgs g;
      g.desc = (const uint4*)h->d_steps4; g.n_pre = h->steps4[0]; g.n_post = h->steps4[1];
      g.loadpos = (const u16*)(h->d_f2tab + h->f2off[S_LOADPOS]); g.rowof = (const u16*)(h->d_f2tab + h->f2off[S_ROWOF]); g.qinv = (const u16*)(h->d_f2tab + h->f2off[S_QINV]);
      g.G = h->d_G; g.C = h->d_C; g.gamma = h->d_gamma; g.rhs = d_rhs; g.x = d_x; g.active = h->d_active; g.flags = h->d_flags;
      g.B = h->B; g.n = h->n; g.nnz = h->nnz; g.lu_words = h->f2_lu_words; g.nc = h->f2_nc; g.dn0 = h->f2_dn0;
      if (shmem_s > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_lu_steps<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem_s));
      hipLaunchKernelGGL(k_lu_steps<4>, dim3(h->B), dim3(256), shmem_s, h->stream, g);
      HIP_TRY(hipGetLastError());
      return CADNIP_OK;
    }
  }
  // a few instances of a circuit with many passes: several waves per instance (k_lu_f2_mw).  CADNIP_LU_WPI forces 1 / 4 (diagnostic)
  {
    const char* e = getenv("CADNIP_LU_WPI");
    const int wpi = e ? atoi(e) : (h->B * 4 <= h->n_cu_hint() && h->f2_n_pre + h->f2_n_post >= 32 ? 4 : 1);
    const size_t shmem_mw = (tab_dbl + per) * 8;
    if (wpi == 4 && shmem_mw <= 160 * 1024) {
      if (shmem_mw > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_lu_f2_mw<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem_mw));
      hipLaunchKernelGGL(k_lu_f2_mw<4>, dim3(h->B), dim3(256), shmem_mw, h->stream, f);
      HIP_TRY(hipGetLastError());
      return CADNIP_OK;
    }
  }
  // many instances: one wave each.  On the step program when its descriptors fit beside eight work arrays (CADNIP_LU_F2S = 0 keeps the passes)
  if (h->d_steps1 && !(getenv("CADNIP_LU_F2S") && atoi(getenv("CADNIP_LU_F2S")) == 0)) {
    const int lo = h->f2off[S_LOADPOS] & ~3;
    const size_t tabs = (size_t)(h->f2_lu_len - lo) / 2, desc = (size_t)h->steps1_len, pers = per + 2;
    int wpb = 8;
    while (wpb > 1 && h->B < 256 * wpb / 2) wpb >>= 1;
    const size_t shmem = (tabs + desc + wpb * pers) * 8;
    if ((tabs + desc + 8 * pers) * 8 <= 160 * 1024) {
      f.tab_lo = lo; f.tab_len = h->f2_lu_len - lo;
      f.steps = (const uint4*)h->d_steps1; f.steps_len = h->steps1_len; f.ts_pre = h->steps1[0]; f.ts_post = h->steps1[1];
      const int grid = (h->B + wpb - 1) / wpb;
#define LAUNCH(W) do { if (shmem > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_lu_f2s<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
    hipLaunchKernelGGL(k_lu_f2s<W>, dim3(grid), dim3(64 * W), shmem, h->stream, f); } while (0)
      if (wpb == 8) LAUNCH(8); else if (wpb == 4) LAUNCH(4); else if (wpb == 2) LAUNCH(2); else LAUNCH(1);
#undef LAUNCH
      HIP_TRY(hipGetLastError());
      return CADNIP_OK;
    }
  }
  int wpb = 8;
  while (wpb > 1 && ((tab_dbl + wpb * per) * 8 > 160 * 1024 || h->B < 256 * wpb / 2)) wpb >>= 1;
  const size_t shmem = (tab_dbl + wpb * per) * 8;
  if (shmem > 160 * 1024) return 1;
  const int grid = (h->B + wpb - 1) / wpb;
#define LAUNCH(W) do { if (shmem > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_lu_f2<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
    hipLaunchKernelGGL(k_lu_f2<W>, dim3(grid), dim3(64 * W), shmem, h->stream, f); } while (0)
  if (wpb == 8) LAUNCH(8); else if (wpb == 4) LAUNCH(4); else if (wpb == 2) LAUNCH(2); else LAUNCH(1);
#undef LAUNCH
  HIP_TRY(hipGetLastError());
  return CADNIP_OK;
}

}  // namespace cadnip
