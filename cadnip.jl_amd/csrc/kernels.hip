// kernels.hip -- gfx950 kernels of the unfused hot path: one stamping kernel per device type,
// slot->nz segmented gather (assemble), residual, Jacobian, and the batched sparse LU.
//
// Parallel axes: (sweep instance, device) for stamping, (sweep instance, nz / row) for the
// assembly-type kernels, one workgroup per sweep instance for the LU whose working set lives
// in LDS.  All per-instance arrays are instance-major; inside an instance the device SoA
// layout makes consecutive lanes touch consecutive addresses.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdio.h>
#include <string.h>
#include "devices.hpp"
#include "internal.hpp"

namespace cadnip {

static thread_local std::string g_last_error;
void set_last_error(const char* what, hipError_t e) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  fprintf(stderr, "[cadnip_hip] HIP error: %s\n", g_last_error.c_str());
}

ProfScope::ProfScope(CadnipHandle* hh, const char* name) : h(hh), idx(-1) {
  if (!h->prof_on) return;
  for (size_t i = 0; i < h->prof.size(); ++i)
    if (strcmp(h->prof[i].name, name) == 0) { idx = (int)i; break; }
  if (idx < 0) { h->prof.push_back(ProfEntry{name, 0.0, 0}); idx = (int)h->prof.size() - 1; }
  (void)hipEventRecord(h->ev0, h->stream);
}
ProfScope::~ProfScope() {
  if (!h->prof_on || idx < 0) return;
  (void)hipEventRecord(h->ev1, h->stream);
  (void)hipEventSynchronize(h->ev1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, h->ev0, h->ev1);
  h->prof[idx].ms += ms;
  h->prof[idx].calls += 1;
}

// ------------------------------------------------------------------------------------------
// stamping: one kernel per device type
// ------------------------------------------------------------------------------------------
struct StampArgs {
  const int* nodes; const int* ipar; const double* par; const double* wave;
  const double* u; const double* t; const int* active;
  double* S; double* limit_w;
  int B, count, n, n_par, ns, ns_g, ns_c, g_base, c_base, b_base, mode, initjct;
};

template <int TYPE>
__global__ void __launch_bounds__(256) k_stamp(StampArgs a) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= a.B * a.count) return;
  int inst = tid / a.count, dev = tid - inst * a.count;
  if (!a.active[inst]) return;
  DevCtx d{a.nodes, a.ipar, a.par + (size_t)inst * a.n_par * a.count, a.wave, a.count, dev, a.t[inst], a.mode, a.initjct};
  double* S = a.S + (size_t)inst * a.ns;
  SlotOut s{S + a.g_base, S + a.ns_g + a.c_base, S + a.ns_g + a.ns_c + a.b_base, a.count, dev};
  const double* u = a.u + (size_t)inst * a.n;
  double* lw = a.limit_w + (size_t)inst * a.n;
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
  else if (TYPE == CADNIP_DEV_MOS1) stamp_mos1(d, u, s, lw);
  else if (TYPE == CADNIP_DEV_BVSOURCE) stamp_bvsource(d, u, s, lw);
  else if (TYPE == CADNIP_DEV_BISOURCE) stamp_bisource(d, u, s, lw);
  else if (TYPE == CADNIP_DEV_VA) stamp_va(d, u, s, lw);
}

template <int TYPE>
static void launch_stamp_t(const StampArgs& a, hipStream_t st) {
  int total = a.B * a.count;
  int bs = (TYPE == CADNIP_DEV_MOS1) ? 64 : 256;
  hipLaunchKernelGGL(k_stamp<TYPE>, dim3((total + bs - 1) / bs), dim3(bs), 0, st, a);
}

// ------------------------------------------------------------------------------------------
// assemble: G[nz] = sum of its G slots (COO order), likewise C and b; then the post-stamp steps
// of fast_rebuild! (precompile.jl:508-534): deferred b is the gather itself, srcFact, gshunt.
// ------------------------------------------------------------------------------------------
struct AsmArgs {
  const double* S; const int* g_ptr; const int* g_slots; const int* c_ptr; const int* c_slots; const int* b_ptr; const int* b_slots;
  const unsigned char* diag_flag; const int* active;
  int* nonfinite;   // [B] set to 1 when an assembled G / C / b value of the instance is not finite (cadnip_rebuild -> CADNIP_NONFINITE)
  double* G; double* C; double* b;
  int B, n, nnz, ns, ns_g, ns_c; const double* srcFact; const double* gshunt;   // [B] each
};

// Entries that gather more than LONG_LIST slots (the supply rails of a large circuit: G[vdd,vdd] of the c6288 multiplier
// sums 20 k stamps) are left to k_assemble_long / k_residual_long, one workgroup each: a single thread walking such a list
// made the whole kernel take 6 ms.  The DFF has none.
#define LONG_LIST 512

__device__ __forceinline__ double block_sum_256(double v, double* red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (t < s) red[t] += red[t + s]; __syncthreads(); }
  return red[0];
}

// one workgroup per (instance, long entry); `long_e` indexes the combined [G nz | C nz | b rows] space of k_assemble
__global__ void __launch_bounds__(256) k_assemble_long(AsmArgs a, const int* long_e, int n_long) {
  __shared__ double red[256];
  const int inst = blockIdx.x / n_long, e = long_e[blockIdx.x % n_long];
  if (!a.active[inst]) return;
  const double* S = a.S + (size_t)inst * a.ns;
  const int* ptr; const int* slots; const double* Sx; int k; double* out;
  if (e < a.nnz) { ptr = a.g_ptr; slots = a.g_slots; Sx = S; k = e; out = a.G + (size_t)inst * a.nnz + k; }
  else if (e < 2 * a.nnz) { ptr = a.c_ptr; slots = a.c_slots; Sx = S + a.ns_g; k = e - a.nnz; out = a.C + (size_t)inst * a.nnz + k; }
  else { ptr = a.b_ptr; slots = a.b_slots; Sx = S + a.ns_g + a.ns_c; k = e - 2 * a.nnz; out = a.b + (size_t)inst * a.n + k; }
  double acc = 0.0;
  for (int p = ptr[k] + threadIdx.x; p < ptr[k + 1]; p += 256) acc += Sx[slots[p]];
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) {
    const double gsh = a.gshunt[inst], sf = a.srcFact[inst];
    if (e < a.nnz && gsh != 0.0 && a.diag_flag[e]) acc += gsh;
    if (e >= 2 * a.nnz && sf < 1.0) acc *= sf;
    if (!isfinite(acc)) a.nonfinite[inst] = 1;
    *out = acc;
  }
}

__global__ void __launch_bounds__(256) k_assemble(AsmArgs a) {
  int per = 2 * a.nnz + a.n;
  long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)a.B * per) return;
  int inst = (int)(tid / per), e = (int)(tid - (long)inst * per);
  if (!a.active[inst]) return;
  const double* S = a.S + (size_t)inst * a.ns;
  if (e < a.nnz) {
    if (a.g_ptr[e + 1] - a.g_ptr[e] > LONG_LIST) return;
    double acc = 0.0;
    for (int p = a.g_ptr[e]; p < a.g_ptr[e + 1]; ++p) acc += S[a.g_slots[p]];
    const double gsh = a.gshunt[inst];
    if (gsh != 0.0 && a.diag_flag[e]) acc += gsh;
    if (!isfinite(acc)) a.nonfinite[inst] = 1;
    a.G[(size_t)inst * a.nnz + e] = acc;
  } else if (e < 2 * a.nnz) {
    int k = e - a.nnz;
    const double* Sc = S + a.ns_g;
    if (a.c_ptr[k + 1] - a.c_ptr[k] > LONG_LIST) return;
    double acc = 0.0;
    for (int p = a.c_ptr[k]; p < a.c_ptr[k + 1]; ++p) acc += Sc[a.c_slots[p]];
    if (!isfinite(acc)) a.nonfinite[inst] = 1;
    a.C[(size_t)inst * a.nnz + k] = acc;
  } else {
    int i = e - 2 * a.nnz;
    const double* Sb = S + a.ns_g + a.ns_c;
    if (a.b_ptr[i + 1] - a.b_ptr[i] > LONG_LIST) return;
    double acc = 0.0;
    for (int p = a.b_ptr[i]; p < a.b_ptr[i + 1]; ++p) acc += Sb[a.b_slots[p]];
    const double sf = a.srcFact[inst];
    if (sf < 1.0) acc *= sf;
    if (!isfinite(acc)) a.nonfinite[inst] = 1;
    a.b[(size_t)inst * a.n + i] = acc;
  }
}

// resid = C*du + G*u - b   (precompile.jl:546-557), CSR row gather
struct ResArgs {
  const double* G; const double* C; const double* b; const double* u; const double* du; const int* rowptr; const int* colidx;
  const int* active; double* r; int B, n, nnz;
};
__global__ void __launch_bounds__(256) k_residual(ResArgs a) {
  long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)a.B * a.n) return;
  int inst = (int)(tid / a.n), i = (int)(tid - (long)inst * a.n);
  if (!a.active[inst]) return;
  const double* G = a.G + (size_t)inst * a.nnz;
  const double* C = a.C + (size_t)inst * a.nnz;
  const double* u = a.u + (size_t)inst * a.n;
  const double* du = a.du + (size_t)inst * a.n;
  if (a.rowptr[i + 1] - a.rowptr[i] > LONG_LIST) return;      // k_residual_long
  double accC = 0.0, accG = 0.0;
  for (int p = a.rowptr[i]; p < a.rowptr[i + 1]; ++p) { int j = a.colidx[p]; accC += C[p] * du[j]; accG += G[p] * u[j]; }
  a.r[(size_t)inst * a.n + i] = (accC + accG) - a.b[(size_t)inst * a.n + i];
}
__global__ void __launch_bounds__(256) k_residual_long(ResArgs a, const int* long_rows, int n_long) {
  __shared__ double red[256];
  const int inst = blockIdx.x / n_long, i = long_rows[blockIdx.x % n_long];
  if (!a.active[inst]) return;
  const double* G = a.G + (size_t)inst * a.nnz;
  const double* C = a.C + (size_t)inst * a.nnz;
  const double* u = a.u + (size_t)inst * a.n;
  const double* du = a.du + (size_t)inst * a.n;
  double acc = 0.0;
  for (int p = a.rowptr[i] + threadIdx.x; p < a.rowptr[i + 1]; p += 256) { int j = a.colidx[p]; acc += C[p] * du[j] + G[p] * u[j]; }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) a.r[(size_t)inst * a.n + i] = acc - a.b[(size_t)inst * a.n + i];
}

// J = G + gamma*C   (precompile.jl:580-582)
__global__ void __launch_bounds__(256) k_jacobian(const double* G, const double* C, const double* gamma, const int* active, double* J, int B, int nnz) {
  long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)B * nnz) return;
  int inst = (int)(tid / nnz);
  if (!active[inst]) return;
  J[tid] = G[tid] + gamma[inst] * C[tid];
}

// ------------------------------------------------------------------------------------------
// batched sparse LU: one workgroup per sweep instance, LU values resident in LDS.
// ------------------------------------------------------------------------------------------
struct LUArgs {
  const double* J; const double* G; const double* C; const double* gamma;   // fuse: J := G + gamma*C
  double* LU; const double* rhs; double* x; double* tmp;
  const int* active; int* flags;
  const int* load_dst; const int* ent_pos; const int* ent_diag; const int* ent_ptr; const int* term_a; const int* term_b; const int* lev_ptr;
  const int* lu_rowptr; const int* lu_col; const int* lu_diag; const int* rperm; const int* cperm;
  const int* fwd_rows; const int* fwd_lev_ptr; const int* bwd_rows; const int* bwd_lev_ptr;
  int n, nnz, nnz_lu, n_lev, n_fwd_lev, n_bwd_lev;
  int do_factor, do_solve, fuse, use_lds, write_back;
};

__global__ void k_lu(LUArgs a) {
  extern __shared__ double smem[];
  const int inst = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  if (!a.active[inst]) return;
  double* lu = a.use_lds ? smem : a.LU + (size_t)inst * a.nnz_lu;
  double* y = a.use_lds ? smem + a.nnz_lu : a.tmp + (size_t)inst * a.n;
  if (a.do_factor) {
    for (int p = tid; p < a.nnz_lu; p += nt) lu[p] = 0.0;
    __syncthreads();
    if (a.fuse) {
      const double* G = a.G + (size_t)inst * a.nnz;
      const double* C = a.C + (size_t)inst * a.nnz;
      const double gam = a.gamma[inst];
      for (int k = tid; k < a.nnz; k += nt) lu[a.load_dst[k]] = G[k] + gam * C[k];
    } else {
      const double* J = a.J + (size_t)inst * a.nnz;
      for (int k = tid; k < a.nnz; k += nt) lu[a.load_dst[k]] = J[k];
    }
    __syncthreads();
    for (int lev = 0; lev < a.n_lev; ++lev) {
      const int e1 = a.lev_ptr[lev + 1];
      for (int e = a.lev_ptr[lev] + tid; e < e1; e += nt) {
        const int pos = a.ent_pos[e];
        double acc = lu[pos];
        const int t1 = a.ent_ptr[e + 1];
        // four terms in flight: with the factors in HBM every term is two dependent memory latencies (indices, then
        // values); one term at a time made an entry with k terms cost 2 k latencies.  Same summation order.
        int t = a.ent_ptr[e];
        for (; t + 4 <= t1; t += 4) {
          int ia[4], ib[4];
          double va[4], vb[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { ia[q] = a.term_a[t + q]; ib[q] = a.term_b[t + q]; }
#pragma unroll
          for (int q = 0; q < 4; ++q) { va[q] = lu[ia[q]]; vb[q] = lu[ib[q]]; }
#pragma unroll
          for (int q = 0; q < 4; ++q) acc -= va[q] * vb[q];
        }
        for (; t < t1; ++t) acc -= lu[a.term_a[t]] * lu[a.term_b[t]];
        const int dg = a.ent_diag[e];
        if (dg >= 0) acc /= lu[dg];
        lu[pos] = acc;
      }
      __syncthreads();
    }
    int bad = 0;
    for (int i = tid; i < a.n; i += nt) { double dd = lu[a.lu_diag[i]]; if (dd == 0.0 || !isfinite(dd)) bad = 1; }
    if (bad) atomicOr(&a.flags[inst], 1);
    if (a.use_lds && a.write_back) {
      double* out = a.LU + (size_t)inst * a.nnz_lu;
      for (int p = tid; p < a.nnz_lu; p += nt) out[p] = lu[p];
    }
  }
  if (a.do_solve) {
    if (!a.do_factor) lu = a.LU + (size_t)inst * a.nnz_lu;   // read factors in place
    const double* rhs = a.rhs + (size_t)inst * a.n;
    for (int i = tid; i < a.n; i += nt) y[i] = rhs[a.rperm[i]];
    __syncthreads();
    for (int lev = 0; lev < a.n_fwd_lev; ++lev) {
      const int r1 = a.fwd_lev_ptr[lev + 1];
      for (int r = a.fwd_lev_ptr[lev] + tid; r < r1; r += nt) {
        const int i = a.fwd_rows[r];
        double acc = y[i];
        const int p1 = a.lu_diag[i];
        int p = a.lu_rowptr[i];
        for (; p + 4 <= p1; p += 4) {
          int jc[4];
          double lv[4], yv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { jc[q] = a.lu_col[p + q]; lv[q] = lu[p + q]; }
#pragma unroll
          for (int q = 0; q < 4; ++q) yv[q] = y[jc[q]];
#pragma unroll
          for (int q = 0; q < 4; ++q) acc -= lv[q] * yv[q];
        }
        for (; p < p1; ++p) acc -= lu[p] * y[a.lu_col[p]];
        y[i] = acc;
      }
      __syncthreads();
    }
    for (int lev = 0; lev < a.n_bwd_lev; ++lev) {
      const int r1 = a.bwd_lev_ptr[lev + 1];
      for (int r = a.bwd_lev_ptr[lev] + tid; r < r1; r += nt) {
        const int i = a.bwd_rows[r];
        double acc = y[i];
        const int dp = a.lu_diag[i], p1 = a.lu_rowptr[i + 1];
        const double piv = lu[dp];
        int p = dp + 1;
        for (; p + 4 <= p1; p += 4) {
          int jc[4];
          double lv[4], yv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { jc[q] = a.lu_col[p + q]; lv[q] = lu[p + q]; }
#pragma unroll
          for (int q = 0; q < 4; ++q) yv[q] = y[jc[q]];
#pragma unroll
          for (int q = 0; q < 4; ++q) acc -= lv[q] * yv[q];
        }
        for (; p < p1; ++p) acc -= lu[p] * y[a.lu_col[p]];
        y[i] = acc / piv;
      }
      __syncthreads();
    }
    double* x = a.x + (size_t)inst * a.n;
    for (int i = tid; i < a.n; i += nt) x[a.cperm[i]] = y[i];
  }
}

__global__ void __launch_bounds__(256) k_calib_copy_f64(const double* __restrict__ a, double* __restrict__ b, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}

__global__ void k_negate(double* x, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = -x[i];
}
int launch_negate(CadnipHandle* h, double* d_x, long n) {
  if (n <= 0) return CADNIP_OK;
  hipLaunchKernelGGL(k_negate, dim3((unsigned)std::min<long>((n + 255) / 256, 4096)), dim3(256), 0, h->stream, d_x, n);
  HIP_TRY(hipGetLastError());
  return CADNIP_OK;
}

int launch_calib_copy(CadnipHandle* h, long n, int reps) {
  double *a = nullptr, *b = nullptr;
  HIP_TRY(hipMalloc((void**)&a, n * sizeof(double)));
  HIP_TRY(hipMalloc((void**)&b, n * sizeof(double)));
  HIP_TRY(hipMemsetAsync(a, 0, n * sizeof(double), h->stream));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_calib_copy_f64, dim3(2048), dim3(256), 0, h->stream, a, b, n);
  HIP_TRY(hipStreamSynchronize(h->stream));
  (void)hipFree(a); (void)hipFree(b);
  return CADNIP_OK;
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
int launch_rebuild(CadnipHandle* h) {
  for (auto& blk : h->blocks) {
    if (blk.count == 0) continue;
    StampArgs a{blk.d_nodes, blk.d_ipar, blk.d_par, h->d_wave, h->d_u, h->d_t, h->d_active, h->d_S, h->d_limit_w,
                h->B, blk.count, h->n, blk.n_par, h->ns, h->ns_g, h->ns_c, blk.g_base, blk.c_base, blk.b_base, h->spec.mode, h->initjct};
    switch (blk.type) {
#define CASE(T, NAME) case T: { ProfScope ps(h, NAME); launch_stamp_t<T>(a, h->stream); } break;
      CASE(CADNIP_DEV_RESISTOR, "stamp_resistor") CASE(CADNIP_DEV_CAPACITOR, "stamp_capacitor")
      CASE(CADNIP_DEV_INDUCTOR, "stamp_inductor") CASE(CADNIP_DEV_VSOURCE, "stamp_vsource")
      CASE(CADNIP_DEV_ISOURCE, "stamp_isource") CASE(CADNIP_DEV_VCVS, "stamp_vcvs") CASE(CADNIP_DEV_VCCS, "stamp_vccs")
      CASE(CADNIP_DEV_CCVS, "stamp_ccvs") CASE(CADNIP_DEV_CCCS, "stamp_cccs") CASE(CADNIP_DEV_DIODE, "stamp_diode")
      CASE(CADNIP_DEV_DIODECAP, "stamp_diodecap") CASE(CADNIP_DEV_SIMPLEMOS, "stamp_simplemos")
      CASE(CADNIP_DEV_MOS1, "stamp_mos1") CASE(CADNIP_DEV_BVSOURCE, "stamp_bvsource") CASE(CADNIP_DEV_BISOURCE, "stamp_bisource")
      CASE(CADNIP_DEV_VA, "stamp_va")
#undef CASE
      default: return CADNIP_BADARG;
    }
  }
  {
    ProfScope ps(h, "assemble");
    AsmArgs a{h->d_S, h->d_g_ptr, h->d_g_slots, h->d_c_ptr, h->d_c_slots, h->d_b_ptr, h->d_b_slots, h->d_diag_flag, h->d_active, h->d_nonfinite,
              h->d_G, h->d_C, h->d_b, h->B, h->n, h->nnz, h->ns, h->ns_g, h->ns_c, h->d_srcfact, h->d_gshunt};
    long total = (long)h->B * (2L * h->nnz + h->n);
    hipLaunchKernelGGL(k_assemble, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, a);
    if (h->n_long_asm > 0) hipLaunchKernelGGL(k_assemble_long, dim3((unsigned)(h->B * h->n_long_asm)), dim3(256), 0, h->stream, a, h->d_long_asm, h->n_long_asm);
  }
  HIP_TRY(hipGetLastError());
  return CADNIP_OK;
}

int launch_residual(CadnipHandle* h, const double* d_du) {
  ProfScope ps(h, "residual");
  ResArgs a{h->d_G, h->d_C, h->d_b, h->d_u, d_du, h->d_rowptr, h->d_colidx, h->d_active, h->d_resid, h->B, h->n, h->nnz};
  long total = (long)h->B * h->n;
  hipLaunchKernelGGL(k_residual, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, a);
  if (h->n_long_rows > 0) hipLaunchKernelGGL(k_residual_long, dim3((unsigned)(h->B * h->n_long_rows)), dim3(256), 0, h->stream, a, h->d_long_rows, h->n_long_rows);
  HIP_TRY(hipGetLastError());
  return CADNIP_OK;
}

int launch_jacobian(CadnipHandle* h) {
  ProfScope ps(h, "jacobian");
  long total = (long)h->B * h->nnz;
  hipLaunchKernelGGL(k_jacobian, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, h->d_G, h->d_C, h->d_gamma, h->d_active, h->d_J, h->B, h->nnz);
  HIP_TRY(hipGetLastError());
  return CADNIP_OK;
}

static int launch_lu(CadnipHandle* h, const char* name, int do_factor, int do_solve, bool fuse, const double* d_rhs, double* d_x) {
  if (!h->analyzed) return CADNIP_NOTREADY;
  ProfScope ps(h, name);
  const LUProgram& P = h->lu;
  size_t lds = ((size_t)P.nnz_lu + h->n) * sizeof(double);
  int use_lds = lds <= 150 * 1024;
  LUArgs a{h->d_J, h->d_G, h->d_C, h->d_gamma, h->d_LU, d_rhs, d_x, h->d_tmp, h->d_active, h->d_flags,
           h->d_load_dst, h->d_ent_pos, h->d_ent_diag, h->d_ent_ptr, h->d_term_a, h->d_term_b, h->d_lev_ptr,
           h->d_lu_rowptr, h->d_lu_col, h->d_lu_diag, h->d_rperm, h->d_cperm, h->d_fwd_rows, h->d_fwd_lev_ptr, h->d_bwd_rows, h->d_bwd_lev_ptr,
           h->n, h->nnz, P.nnz_lu, (int)P.lev_ptr.size() - 1, (int)P.fwd_lev_ptr.size() - 1, (int)P.bwd_lev_ptr.size() - 1,
           do_factor, do_solve, fuse ? 1 : 0, use_lds, 1};
  if (!do_factor) { a.use_lds = 0; }   // solve-only reads the stored factors; y lives in d_tmp
  size_t shmem = a.use_lds ? lds : 0;
  if (shmem > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_lu, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  // one workgroup per instance; its size follows the work per dependency level: a wave for the small systems of a sweep
  // (the DFF: 186 computed entries in 14 levels), 16 waves for a single large circuit whose factors live in HBM
  // (c6288: 79 k entries, 352 k multiply-adds in 237 levels -- with one wave the refactorisation took 9.3 ms)
  int threads = !a.use_lds ? 1024 : (P.nnz_lu >= 8192 ? 256 : 64);
  if (const char* e = getenv("CADNIP_LU_THREADS")) { const int t = atoi(e); if (t >= 64 && t <= 1024 && t % 64 == 0) threads = t; }   // diagnostic
  hipLaunchKernelGGL(k_lu, dim3(h->B), dim3(threads), shmem, h->stream, a);
  HIP_TRY(hipGetLastError());
  return CADNIP_OK;
}

int launch_factor(CadnipHandle* h, bool fuse) { return launch_lu(h, "lu_factor", 1, 0, fuse, nullptr, nullptr); }
int launch_solve(CadnipHandle* h, const double* d_rhs, double* d_x) { return launch_lu(h, "lu_solve", 0, 1, false, d_rhs, d_x); }
int launch_factor_solve(CadnipHandle* h, bool fuse, const double* d_rhs, double* d_x) { return launch_lu(h, "lu_factor_solve", 1, 1, fuse, d_rhs, d_x); }

}  // namespace cadnip
