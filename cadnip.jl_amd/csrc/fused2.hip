// fused2.hip -- fused Newton kernel, second generation: WPB sweep instances per workgroup, one wave each,
// the circuit *structure* resident in LDS and shared by the workgroup's instances.
//
// Why: for DFF-class circuits one Newton iteration of one instance is a chain of short, dependent phases
// (stamp -> assemble -> refactor over ~30 dependency levels -> two triangular solves -> update).  With one
// wave per instance nothing hides a global-memory index load inside such a chain, and the LDS footprint
// of fused v1 (slot buffer + G + C) admits only ~3 waves per CU.  Here
//   * every index / program array (slot -> LU position, CSR pattern, LU entry program, level schedules)
//     is copied once per launch into LDS as uint16 and read from there by all WPB instances;
//   * there is no slot buffer and no G / C: each stamp value is accumulated straight into the instance's
//     LDS-resident Jacobian  J = G + a0*C  (at its LU position) with ds_add_f64, and into the residual via
//        r = C*du + G*u - b = J*u + C*beta - b      (du = a0*u + beta, BDF),
//     so per instance only  lu[nnz_lu] + u,beta,r,y[n]  live in LDS (16 KB for the DFF);
//   * a wave never waits for another wave: all synchronisation is wave-level (tran_ctrl.hpp).
// Summation order inside an nz differs from the per-op path (slot-major instead of COO order), so results
// agree with it to rounding (1e-13 relative), not bit for bit; the per-op path remains the reference ABI.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "devices.hpp"
#include "internal.hpp"
#include "tran_ctrl.hpp"

namespace cadnip {

typedef unsigned short u16;
#define NOPOS 0xFFFFu

enum { T_GPOS = 0, T_CPOS, T_CROW, T_CCOL, T_BROW, T_NZROW, T_COLIDX, T_LOADDST, T_ENTPOS, T_ENTDIAG, T_ENTPTR, T_TERMA, T_TERMB, T_LEVPTR,
       T_LUROWPTR, T_LUCOL, T_LUDIAG, T_RPERM, T_CPERM, T_FWDROWS, T_FWDLEV, T_BWDROWS, T_BWDLEV, T_NTAB };

struct F2Block {
  const int* nodes; const int* ipar; const double* par;
  int type, count, n_par, g_base, c_base, b_base;
};

struct F2Args {
  F2Block blk[CADNIP_DEV_NTYPES];
  int n_blk;
  const double* wave;
  const u16* tab;            // packed uint16 tables in global memory
  int off[T_NTAB];           // offsets (in u16 units) of each table inside `tab`
  int tab_len;               // total u16 count (even)
  int n, nnz, nnz_lu, n_lev, n_fwd_lev, n_bwd_lev, rounds, B;
  unsigned long long* prof;   // diagnostic: per-phase cycle sums [8] (null in production launches)
  TranArgs t;
};

// stamp writer that accumulates straight into J (LU array) and the residual
struct AccumOut {
  double* lu; double* r; const double* beta; double a0;
  const u16 *gpos, *cpos, *crow, *ccol, *brow;   // already offset to this device block
  int count, dev;
  // exact zeros are skipped: the accumulators start at +0.0, so adding them would change nothing
  __device__ __forceinline__ void G(int k, double v) const {
    if (v == 0.0) return;
    unsigned p = gpos[k * count + dev];
    if (p != NOPOS) atomicAdd(&lu[p], v);
  }
  __device__ __forceinline__ void C(int k, double v) const {
    if (v == 0.0) return;
    int idx = k * count + dev;
    unsigned p = cpos[idx];
    if (p != NOPOS) { atomicAdd(&lu[p], a0 * v); atomicAdd(&r[crow[idx]], v * beta[ccol[idx]]); }
  }
  __device__ __forceinline__ void B(int k, double v) const {
    if (v == 0.0) return;
    unsigned row = brow[k * count + dev];
    if (row != NOPOS) atomicAdd(&r[row], -v);
  }
};

template <class Out>
__device__ __forceinline__ void dispatch_stamp2(int type, const DevCtx& d, const double* u, const Out& s, double* lw) {
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
  }
}

template <int WPB>
__global__ void __launch_bounds__(64 * WPB) k_fused2(F2Args f) {
  extern __shared__ double sm[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = f.n;
  // ---- shared tables: one cooperative copy per launch
  u16* tab = (u16*)sm;
  {
    const unsigned* src = (const unsigned*)f.tab;
    unsigned* dst = (unsigned*)tab;
    for (int i = tid; i < f.tab_len / 2; i += 64 * WPB) dst[i] = src[i];
  }
  __syncthreads();
  const int inst = blockIdx.x * WPB + w;
  if (inst >= f.B) return;
  const int tab_dbl = (f.tab_len * 2 + 7) / 8;            // doubles occupied by the tables
  const int per = f.nnz_lu + 4 * n;                       // doubles per instance
  double* lu = sm + tab_dbl + (size_t)w * per;
  double* us = lu + f.nnz_lu;
  double* betas = us + n;
  double* rs = betas + n;
  double* y = rs + n;
  const u16 *gpos = tab + f.off[T_GPOS], *cpos = tab + f.off[T_CPOS], *crow = tab + f.off[T_CROW], *ccol = tab + f.off[T_CCOL], *brow = tab + f.off[T_BROW];
  const u16 *nzrow = tab + f.off[T_NZROW], *colidx = tab + f.off[T_COLIDX], *load_dst = tab + f.off[T_LOADDST];
  const u16 *ent_pos = tab + f.off[T_ENTPOS], *ent_diag = tab + f.off[T_ENTDIAG], *ent_ptr = tab + f.off[T_ENTPTR];
  const u16 *term_a = tab + f.off[T_TERMA], *term_b = tab + f.off[T_TERMB], *lev_ptr = tab + f.off[T_LEVPTR];
  const u16 *lu_rowptr = tab + f.off[T_LUROWPTR], *lu_col = tab + f.off[T_LUCOL], *lu_diag = tab + f.off[T_LUDIAG];
  const u16 *rperm = tab + f.off[T_RPERM], *cperm = tab + f.off[T_CPERM];
  const u16 *fwd_rows = tab + f.off[T_FWDROWS], *fwd_lev = tab + f.off[T_FWDLEV], *bwd_rows = tab + f.off[T_BWDROWS], *bwd_lev = tab + f.off[T_BWDLEV];
  const TranArgs& a = f.t;
  const double* ug = a.u + (size_t)inst * n;
  const double* betag = a.beta + (size_t)inst * n;
  double* delta = a.delta + (size_t)inst * n;
  double* lw = a.limit_w + (size_t)inst * n;
#define F2_STAMP(k) do { if (f.prof) { unsigned long long _t = clock64(); if (lane == 0) atomicAdd(&f.prof[k], _t - tstamp); tstamp = _t; } } while (0)
  unsigned long long tstamp = f.prof ? clock64() : 0;
  for (int round = 0; round < f.rounds; ++round) {
    if (a.status[inst] != 0) break;
    const double tcur = a.tcur[inst], a0 = a.gamma[inst];
    for (int i = lane; i < n; i += 64) { us[i] = ug[i]; betas[i] = betag[i]; rs[i] = 0.0; }
    for (int p = lane; p < f.nnz_lu; p += 64) lu[p] = 0.0;
    CADNIP_WAVE_SYNC();
    F2_STAMP(0);
    // ---- stamp: accumulate J (at LU positions) and the C*beta - b part of the residual
    for (int bi = 0; bi < f.n_blk; ++bi) {
      const F2Block& B = f.blk[bi];
      for (int dev = lane; dev < B.count; dev += 64) {
        DevCtx d{B.nodes, B.ipar, B.par + (size_t)inst * B.n_par * B.count, f.wave, B.count, dev, tcur, 1, 0};
        AccumOut s{lu, rs, betas, a0, gpos + B.g_base, cpos + B.c_base, crow + B.c_base, ccol + B.c_base, brow + B.b_base, B.count, dev};
        dispatch_stamp2(B.type, d, us, s, lw);
      }
    }
    CADNIP_WAVE_SYNC();
    F2_STAMP(1);
    // ---- r += J*u  (J still unfactored in the LU array)
    for (int p = lane; p < f.nnz; p += 64) atomicAdd(&rs[nzrow[p]], lu[load_dst[p]] * us[colidx[p]]);
    CADNIP_WAVE_SYNC();
    F2_STAMP(2);
    // ---- refactor (entry-wise left-looking, level by level)
    for (int lev = 0; lev < f.n_lev; ++lev) {
      const int e0 = lev_ptr[lev], e1 = lev_ptr[lev + 1], E = e1 - e0;
      int lpe = 1;                                   // lanes per entry: split each dot product over lpe lanes
      while (lpe * 2 * E <= 64) lpe *= 2;
      if (lpe == 1) {
        for (int e = e0 + lane; e < e1; e += 64) {
          const int pos = ent_pos[e];
          double acc = lu[pos];
          const int t1 = ent_ptr[e + 1];
          for (int t = ent_ptr[e]; t < t1; ++t) acc -= lu[term_a[t]] * lu[term_b[t]];
          const unsigned dg = ent_diag[e];
          if (dg != NOPOS) acc /= lu[dg];
          lu[pos] = acc;
        }
      } else {
        const int idx = lane / lpe, sub = lane - idx * lpe, e = e0 + idx;
        double part = 0.0;
        if (idx < E) { const int t1 = ent_ptr[e + 1]; for (int t = ent_ptr[e] + sub; t < t1; t += lpe) part += lu[term_a[t]] * lu[term_b[t]]; }
        for (int o = lpe >> 1; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if (idx < E && sub == 0) {
          const int pos = ent_pos[e];
          double acc = lu[pos] - part;
          const unsigned dg = ent_diag[e];
          if (dg != NOPOS) acc /= lu[dg];
          lu[pos] = acc;
        }
      }
      CADNIP_WAVE_SYNC();
    }
    F2_STAMP(3);
    int bad = 0;
    for (int i = lane; i < n; i += 64) { double dd = lu[lu_diag[i]]; if (dd == 0.0 || !isfinite(dd)) bad = 1; }
    if (bad) atomicOr(&a.flags[inst], 1);
    for (int i = lane; i < n; i += 64) y[i] = rs[rperm[i]];
    CADNIP_WAVE_SYNC();
    for (int lev = 0; lev < f.n_fwd_lev; ++lev) {
      const int r0 = fwd_lev[lev], r1 = fwd_lev[lev + 1], R = r1 - r0;
      int lpe = 1;
      while (lpe * 2 * R <= 64) lpe *= 2;
      if (lpe == 1) {
        for (int r = r0 + lane; r < r1; r += 64) {
          const int i = fwd_rows[r];
          double acc = y[i];
          const int p1 = lu_diag[i];
          for (int p = lu_rowptr[i]; p < p1; ++p) acc -= lu[p] * y[lu_col[p]];
          y[i] = acc;
        }
      } else {
        const int idx = lane / lpe, sub = lane - idx * lpe;
        const int i = idx < R ? fwd_rows[r0 + idx] : 0;
        double part = 0.0;
        if (idx < R) { const int p1 = lu_diag[i]; for (int p = lu_rowptr[i] + sub; p < p1; p += lpe) part += lu[p] * y[lu_col[p]]; }
        for (int o = lpe >> 1; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if (idx < R && sub == 0) y[i] -= part;
      }
      CADNIP_WAVE_SYNC();
    }
    for (int lev = 0; lev < f.n_bwd_lev; ++lev) {
      const int r0 = bwd_lev[lev], r1 = bwd_lev[lev + 1], R = r1 - r0;
      int lpe = 1;
      while (lpe * 2 * R <= 64) lpe *= 2;
      if (lpe == 1) {
        for (int r = r0 + lane; r < r1; r += 64) {
          const int i = bwd_rows[r];
          double acc = y[i];
          const int dp = lu_diag[i], p1 = lu_rowptr[i + 1];
          for (int p = dp + 1; p < p1; ++p) acc -= lu[p] * y[lu_col[p]];
          y[i] = acc / lu[dp];
        }
      } else {
        const int idx = lane / lpe, sub = lane - idx * lpe;
        const int i = idx < R ? bwd_rows[r0 + idx] : 0;
        const int dp = lu_diag[i];
        double part = 0.0;
        if (idx < R) { const int p1 = lu_rowptr[i + 1]; for (int p = dp + 1 + sub; p < p1; p += lpe) part += lu[p] * y[lu_col[p]]; }
        for (int o = lpe >> 1; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if (idx < R && sub == 0) y[i] = (y[i] - part) / lu[dp];
      }
      CADNIP_WAVE_SYNC();
    }
    for (int i = lane; i < n; i += 64) delta[cperm[i]] = y[i];
    CADNIP_WAVE_SYNC();
    F2_STAMP(4);
    tran_update_body(a, inst, lane);
    CADNIP_WAVE_SYNC();
    F2_STAMP(5);
    if (f.prof && lane == 0) atomicAdd(&f.prof[7], 1ull);
  }
}

// ---- host side: build the packed uint16 tables once per (structure, LU program) -----------------------------
struct F2Tables {
  std::vector<u16> data;
  int off[T_NTAB];
  void add(int which, const std::vector<int>& v) {
    off[which] = (int)data.size();
    for (int x : v) data.push_back(x < 0 ? (u16)NOPOS : (u16)x);
    if (data.size() & 1) data.push_back(0);
  }
};

static bool f2_prepare(CadnipHandle* h, F2Tables& T, std::vector<int>& gs, std::vector<int>& cs, std::vector<int>& cr, std::vector<int>& cc, std::vector<int>& br,
                       const std::vector<int>& g_ptr, const std::vector<int>& g_slots, const std::vector<int>& c_ptr, const std::vector<int>& c_slots,
                       const std::vector<int>& b_ptr, const std::vector<int>& b_slots) {
  const LUProgram& P = h->lu;
  if (h->n >= 65535 || P.nnz_lu >= 65535 || h->ns_g >= 65535 || h->ns_c >= 65535 || h->ns_b >= 65535 || (int)P.term_a.size() >= 65535) return false;
  std::vector<int> dst(h->nnz, 0);
  for (size_t k = 0; k < P.load_src.size(); ++k) dst[P.load_src[k]] = P.load_dst[k];
  gs.assign(h->ns_g, -1); cs.assign(h->ns_c, -1); cr.assign(h->ns_c, -1); cc.assign(h->ns_c, -1); br.assign(h->ns_b, -1);
  for (int i = 0; i < h->n; ++i)
    for (int e = h->h_rowptr[i]; e < h->h_rowptr[i + 1]; ++e) {
      for (int p = g_ptr[e]; p < g_ptr[e + 1]; ++p) gs[g_slots[p]] = dst[e];
      for (int p = c_ptr[e]; p < c_ptr[e + 1]; ++p) { cs[c_slots[p]] = dst[e]; cr[c_slots[p]] = i; cc[c_slots[p]] = h->h_colidx[e]; }
    }
  for (int i = 0; i < h->n; ++i) for (int p = b_ptr[i]; p < b_ptr[i + 1]; ++p) br[b_slots[p]] = i;
  T.add(T_GPOS, gs); T.add(T_CPOS, cs); T.add(T_CROW, cr); T.add(T_CCOL, cc); T.add(T_BROW, br);
  std::vector<int> nzrow(h->nnz);
  for (int i = 0; i < h->n; ++i) for (int e = h->h_rowptr[i]; e < h->h_rowptr[i + 1]; ++e) nzrow[e] = i;
  T.add(T_NZROW, nzrow); T.add(T_COLIDX, h->h_colidx); T.add(T_LOADDST, dst);
  T.add(T_ENTPOS, P.ent_pos); T.add(T_ENTDIAG, P.ent_diag); T.add(T_ENTPTR, P.ent_ptr); T.add(T_TERMA, P.term_a); T.add(T_TERMB, P.term_b); T.add(T_LEVPTR, P.lev_ptr);
  T.add(T_LUROWPTR, P.lu_rowptr); T.add(T_LUCOL, P.lu_col); T.add(T_LUDIAG, P.lu_diag); T.add(T_RPERM, P.rperm); T.add(T_CPERM, P.cperm);
  T.add(T_FWDROWS, P.fwd_rows); T.add(T_FWDLEV, P.fwd_lev_ptr); T.add(T_BWDROWS, P.bwd_rows); T.add(T_BWDLEV, P.bwd_lev_ptr);
  return true;
}

int launch_fused2_rounds(CadnipHandle* h, const TranArgs& t, int rounds) {
  if (!h->analyzed) return CADNIP_NOTREADY;
  if (h->spec.gshunt != 0.0 || h->spec.srcFact < 1.0) return CADNIP_BADARG;   // homotopies run on the per-op path
  static_assert(T_NTAB <= 32, "f2off too small");
  if (!h->d_f2tab || h->fused2_dirty) {
    // host copies of the gather lists are needed to invert them: read back once
    std::vector<int> g_ptr(h->nnz + 1), c_ptr(h->nnz + 1), b_ptr(h->n + 1);
    HIP_TRY(hipMemcpy(g_ptr.data(), h->d_g_ptr, g_ptr.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(c_ptr.data(), h->d_c_ptr, c_ptr.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(b_ptr.data(), h->d_b_ptr, b_ptr.size() * 4, hipMemcpyDeviceToHost));
    std::vector<int> g_slots(g_ptr.back()), c_slots(c_ptr.back()), b_slots(b_ptr.back());
    if (!g_slots.empty()) HIP_TRY(hipMemcpy(g_slots.data(), h->d_g_slots, g_slots.size() * 4, hipMemcpyDeviceToHost));
    if (!c_slots.empty()) HIP_TRY(hipMemcpy(c_slots.data(), h->d_c_slots, c_slots.size() * 4, hipMemcpyDeviceToHost));
    if (!b_slots.empty()) HIP_TRY(hipMemcpy(b_slots.data(), h->d_b_slots, b_slots.size() * 4, hipMemcpyDeviceToHost));
    F2Tables T;
    std::vector<int> gs, cs, cr, cc, br;
    if (!f2_prepare(h, T, gs, cs, cr, cc, br, g_ptr, g_slots, c_ptr, c_slots, b_ptr, b_slots)) return CADNIP_BADARG;
    if (h->d_f2tab) (void)hipFree(h->d_f2tab);
    h->d_f2tab = nullptr;
    HIP_TRY(hipMalloc((void**)&h->d_f2tab, T.data.size() * sizeof(u16)));
    HIP_TRY(hipMemcpy(h->d_f2tab, T.data.data(), T.data.size() * sizeof(u16), hipMemcpyHostToDevice));
    for (int i = 0; i < T_NTAB; ++i) h->f2off[i] = T.off[i];
    h->f2len = (int)T.data.size();
    h->fused2_dirty = false;
  }
  ProfScope ps(h, "fused2_newton");
  const LUProgram& P = h->lu;
  F2Args f;
  f.n_blk = 0;
  for (auto& b : h->blocks) {
    if (b.count == 0) continue;
    f.blk[f.n_blk++] = F2Block{b.d_nodes, b.d_ipar, b.d_par, b.type, b.count, b.n_par, b.g_base, b.c_base, b.b_base};
  }
  for (int i = 0; i < f.n_blk; ++i)
    for (int j = i + 1; j < f.n_blk; ++j)
      if ((f.blk[j].type == CADNIP_DEV_MOS1) > (f.blk[i].type == CADNIP_DEV_MOS1)) { F2Block tmp = f.blk[i]; f.blk[i] = f.blk[j]; f.blk[j] = tmp; }
  f.wave = h->d_wave;
  f.tab = h->d_f2tab;
  for (int i = 0; i < T_NTAB; ++i) f.off[i] = h->f2off[i];
  f.tab_len = h->f2len;
  f.n = h->n; f.nnz = h->nnz; f.nnz_lu = P.nnz_lu;
  f.n_lev = (int)P.lev_ptr.size() - 1; f.n_fwd_lev = (int)P.fwd_lev_ptr.size() - 1; f.n_bwd_lev = (int)P.bwd_lev_ptr.size() - 1;
  f.rounds = rounds; f.B = h->B; f.t = t;
  f.prof = h->d_f2prof;
  const size_t tab_dbl = ((size_t)h->f2len * 2 + 7) / 8;
  const size_t per = (size_t)P.nnz_lu + 4 * (size_t)h->n;
  const size_t lds_cap = 160 * 1024;
  // waves (= instances) per workgroup: 8 when they fit and there are enough instances to fill the chip, else 4, 2, 1
  int wpb = 8;
  while (wpb > 1 && ((tab_dbl + wpb * per) * 8 > lds_cap || (h->B + wpb - 1) / wpb < 256)) wpb >>= 1;
  size_t shmem = (tab_dbl + wpb * per) * 8;
  if (shmem > lds_cap) return CADNIP_BADARG;
  int grid = (h->B + wpb - 1) / wpb;
#define LAUNCH(W)                                                                                                      \
  do {                                                                                                                 \
    if (shmem > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_fused2<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
    hipLaunchKernelGGL(k_fused2<W>, dim3(grid), dim3(64 * W), shmem, h->stream, f);                                    \
  } while (0)
  if (wpb == 8) LAUNCH(8); else if (wpb == 4) LAUNCH(4); else if (wpb == 2) LAUNCH(2); else LAUNCH(1);
#undef LAUNCH
  HIP_TRY(hipGetLastError());
  return CADNIP_OK;
}

}  // namespace cadnip
