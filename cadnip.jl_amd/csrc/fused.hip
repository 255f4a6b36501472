// fused.hip -- one launch per Newton round: the whole hot path of one sweep instance in one wave.
//
// For DFF-class circuits (n ~ 10^2, nnz ~ 10^3) a Newton iteration moves a few tens of KB per
// instance; seven dependent launches per iteration round-trip the slot buffer, G, C and the LU
// factors through HBM and pay seven launch boundaries.  Here one 64-lane wave owns one sweep
// instance for `rounds` consecutive Newton iterations:
//
//   stamp (every device type) -> slots in LDS -> G, C, b by slot->nz gather (LDS) -> residual
//   -> J = G + a0*C loaded into the LU array (LDS, overlays the dead slot buffer) -> level-scheduled
//   refactor + triangular solves (LDS) -> Newton update + step controller (tran_ctrl.hpp)
//
// Only the instance's vectors (u, du, delta, history) and parameter block are read from HBM.
// The arithmetic -- every loop, every summation order -- is the per-op kernels' (kernels.hip), so
// the two paths produce identical results; the per-op path stays as the drop-in ABI.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "devices.hpp"
#include "internal.hpp"
#include "tran_ctrl.hpp"

namespace cadnip {

struct FusedBlock {
  const int* nodes; const int* ipar; const double* par;
  int type, count, n_par, g_base, c_base, b_base;
};

struct FusedArgs {
  FusedBlock blk[CADNIP_DEV_NTYPES];
  int n_blk;
  const double* wave;
  const int *g_ptr, *g_slots, *c_ptr, *c_slots, *b_ptr, *b_slots;
  const unsigned char* diag_flag;
  const int *rowptr, *colidx;
  const int *load_dst, *ent_pos, *ent_diag, *ent_ptr, *term_a, *term_b, *lev_ptr;
  const int *lu_rowptr, *lu_col, *lu_diag, *rperm, *cperm, *fwd_rows, *fwd_lev_ptr, *bwd_rows, *bwd_lev_ptr;
  int n, nnz, ns, ns_g, ns_c, nnz_lu, n_lev, n_fwd_lev, n_bwd_lev, lu_off, rounds;
  double srcFact, gshunt;
  TranArgs t;
};

__device__ __forceinline__ void dispatch_stamp(int type, const DevCtx& d, const double* u, const SlotOut& s, double* lw) {
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

__global__ void __launch_bounds__(64) k_fused_newton(FusedArgs f) {
  extern __shared__ double sm[];
  const int inst = blockIdx.x, tid = threadIdx.x, n = f.n, nnz = f.nnz;
  const TranArgs& a = f.t;
  double* S = sm;                 // [ns]   slot buffer; dead after the gather, then reused for lu | y
  double* Gs = sm + f.ns;         // [nnz]
  double* Cs = Gs + nnz;          // [nnz]
  double* bs = Cs + nnz;          // [n]
  double* rs = bs + n;            // [n]
  double* lu = sm + f.lu_off;     // [nnz_lu]  (lu_off = 0 when it fits into the slot buffer)
  double* y = lu + f.nnz_lu;      // [n]
  double* u = a.u + (size_t)inst * n;
  const double* du = a.du + (size_t)inst * n;
  double* delta = a.delta + (size_t)inst * n;
  double* lw = a.limit_w + (size_t)inst * n;
  for (int round = 0; round < f.rounds; ++round) {
    if (a.status[inst] != 0) break;
    const double tcur = a.tcur[inst], a0 = a.gamma[inst];
    // ---- stamp: every device of this instance
    for (int bi = 0; bi < f.n_blk; ++bi) {
      const FusedBlock& B = f.blk[bi];
      for (int dev = tid; dev < B.count; dev += 64) {
        DevCtx d{B.nodes, B.ipar, B.par + (size_t)inst * B.n_par * B.count, f.wave, B.count, dev, tcur, 1, 0};
        SlotOut s{S + B.g_base, S + f.ns_g + B.c_base, S + f.ns_g + f.ns_c + B.b_base, B.count, dev};
        dispatch_stamp(B.type, d, u, s, lw);
      }
    }
    __syncthreads();
    // ---- gather slots -> G, C (LDS), b (LDS)
    for (int e = tid; e < nnz; e += 64) {
      double acc = 0.0;
      for (int p = f.g_ptr[e]; p < f.g_ptr[e + 1]; ++p) acc += S[f.g_slots[p]];
      if (f.gshunt != 0.0 && f.diag_flag[e]) acc += f.gshunt;
      Gs[e] = acc;
      const double* Sc = S + f.ns_g;
      acc = 0.0;
      for (int p = f.c_ptr[e]; p < f.c_ptr[e + 1]; ++p) acc += Sc[f.c_slots[p]];
      Cs[e] = acc;
    }
    for (int i = tid; i < n; i += 64) {
      const double* Sb = S + f.ns_g + f.ns_c;
      double acc = 0.0;
      for (int p = f.b_ptr[i]; p < f.b_ptr[i + 1]; ++p) acc += Sb[f.b_slots[p]];
      if (f.srcFact < 1.0) acc *= f.srcFact;
      bs[i] = acc;
    }
    __syncthreads();
    // ---- residual r = C du + G u - b
    for (int i = tid; i < n; i += 64) {
      double accC = 0.0, accG = 0.0;
      for (int p = f.rowptr[i]; p < f.rowptr[i + 1]; ++p) { int j = f.colidx[p]; accC += Cs[p] * du[j]; accG += Gs[p] * u[j]; }
      rs[i] = (accC + accG) - bs[i];
    }
    // ---- J = G + a0 C into the LU array, refactor, solve
    for (int p = tid; p < f.nnz_lu; p += 64) lu[p] = 0.0;
    __syncthreads();
    for (int k = tid; k < nnz; k += 64) lu[f.load_dst[k]] = Gs[k] + a0 * Cs[k];
    __syncthreads();
    for (int lev = 0; lev < f.n_lev; ++lev) {
      const int e1 = f.lev_ptr[lev + 1];
      for (int e = f.lev_ptr[lev] + tid; e < e1; e += 64) {
        const int pos = f.ent_pos[e];
        double acc = lu[pos];
        const int t1 = f.ent_ptr[e + 1];
        for (int t = f.ent_ptr[e]; t < t1; ++t) acc -= lu[f.term_a[t]] * lu[f.term_b[t]];
        const int dg = f.ent_diag[e];
        if (dg >= 0) acc /= lu[dg];
        lu[pos] = acc;
      }
      __syncthreads();
    }
    int bad = 0;
    for (int i = tid; i < n; i += 64) { double dd = lu[f.lu_diag[i]]; if (dd == 0.0 || !isfinite(dd)) bad = 1; }
    if (bad) atomicOr(&a.flags[inst], 1);
    for (int i = tid; i < n; i += 64) y[i] = rs[f.rperm[i]];
    __syncthreads();
    for (int lev = 0; lev < f.n_fwd_lev; ++lev) {
      const int r1 = f.fwd_lev_ptr[lev + 1];
      for (int r = f.fwd_lev_ptr[lev] + tid; r < r1; r += 64) {
        const int i = f.fwd_rows[r];
        double acc = y[i];
        const int p1 = f.lu_diag[i];
        for (int p = f.lu_rowptr[i]; p < p1; ++p) acc -= lu[p] * y[f.lu_col[p]];
        y[i] = acc;
      }
      __syncthreads();
    }
    for (int lev = 0; lev < f.n_bwd_lev; ++lev) {
      const int r1 = f.bwd_lev_ptr[lev + 1];
      for (int r = f.bwd_lev_ptr[lev] + tid; r < r1; r += 64) {
        const int i = f.bwd_rows[r];
        double acc = y[i];
        const int dp = f.lu_diag[i], p1 = f.lu_rowptr[i + 1];
        for (int p = dp + 1; p < p1; ++p) acc -= lu[p] * y[f.lu_col[p]];
        y[i] = acc / lu[dp];
      }
      __syncthreads();
    }
    for (int i = tid; i < n; i += 64) delta[f.cperm[i]] = y[i];
    __syncthreads();
    // ---- Newton update + step controller
    tran_update_body(a, inst, tid);
    __syncthreads();
  }
}

int launch_fused_rounds(CadnipHandle* h, const TranArgs& t, int rounds) {
  if (!h->analyzed) return CADNIP_NOTREADY;
  ProfScope ps(h, "fused_newton");
  const LUProgram& P = h->lu;
  FusedArgs f;
  f.n_blk = 0;
  for (auto& b : h->blocks) {
    if (b.count == 0) continue;
    f.blk[f.n_blk++] = FusedBlock{b.d_nodes, b.d_ipar, b.d_par, b.type, b.count, b.n_par, b.g_base, b.c_base, b.b_base};
  }
  // heavier device types first: their lanes start first inside the stamp phase
  for (int i = 0; i < f.n_blk; ++i)
    for (int j = i + 1; j < f.n_blk; ++j)
      if ((f.blk[j].type == CADNIP_DEV_MOS1) > (f.blk[i].type == CADNIP_DEV_MOS1)) { FusedBlock tmp = f.blk[i]; f.blk[i] = f.blk[j]; f.blk[j] = tmp; }
  f.wave = h->d_wave;
  f.g_ptr = h->d_g_ptr; f.g_slots = h->d_g_slots; f.c_ptr = h->d_c_ptr; f.c_slots = h->d_c_slots; f.b_ptr = h->d_b_ptr; f.b_slots = h->d_b_slots;
  f.diag_flag = h->d_diag_flag; f.rowptr = h->d_rowptr; f.colidx = h->d_colidx;
  f.load_dst = h->d_load_dst; f.ent_pos = h->d_ent_pos; f.ent_diag = h->d_ent_diag; f.ent_ptr = h->d_ent_ptr; f.term_a = h->d_term_a; f.term_b = h->d_term_b;
  f.lev_ptr = h->d_lev_ptr; f.lu_rowptr = h->d_lu_rowptr; f.lu_col = h->d_lu_col; f.lu_diag = h->d_lu_diag; f.rperm = h->d_rperm; f.cperm = h->d_cperm;
  f.fwd_rows = h->d_fwd_rows; f.fwd_lev_ptr = h->d_fwd_lev_ptr; f.bwd_rows = h->d_bwd_rows; f.bwd_lev_ptr = h->d_bwd_lev_ptr;
  f.n = h->n; f.nnz = h->nnz; f.ns = h->ns; f.ns_g = h->ns_g; f.ns_c = h->ns_c; f.nnz_lu = P.nnz_lu;
  f.n_lev = (int)P.lev_ptr.size() - 1; f.n_fwd_lev = (int)P.fwd_lev_ptr.size() - 1; f.n_bwd_lev = (int)P.bwd_lev_ptr.size() - 1;
  f.rounds = rounds; f.srcFact = h->spec.srcFact; f.gshunt = h->spec.gshunt; f.t = t;
  size_t base = (size_t)h->ns + 2 * (size_t)h->nnz + 2 * (size_t)h->n;
  size_t need_lu = (size_t)P.nnz_lu + h->n;
  size_t total;
  if (need_lu <= (size_t)h->ns) { f.lu_off = 0; total = base; }
  else { f.lu_off = (int)base; total = base + need_lu; }
  size_t shmem = total * sizeof(double);
  if (shmem > 160 * 1024) return CADNIP_BADARG;   // circuit too large for the LDS-resident path
  if (shmem > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_fused_newton, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(k_fused_newton, dim3(h->B), dim3(64), shmem, h->stream, f);
  HIP_TRY(hipGetLastError());
  return CADNIP_OK;
}

}  // namespace cadnip
