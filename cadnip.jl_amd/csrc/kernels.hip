// kernels.hip -- gfx950 kernels of the unfused hot path behind the stamping kernels (stamp_csr.hip): residual,
// Jacobian, and the batched sparse LU.
//
// Parallel axes: (sweep instance, row / nz) for residual and Jacobian, one workgroup per sweep instance for the LU
// whose working set lives in LDS.  All per-instance arrays are instance-major; inside an instance the device SoA
// layout makes consecutive lanes touch consecutive addresses.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "devices.hpp"
#include "internal.hpp"
#include "tran_ctrl.hpp"

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

// Rows / entries whose lists are longer than LONG_LIST get a workgroup each (k_residual_long): a single thread walking the
// 20 k entries of a supply-rail row of the c6288 multiplier made the whole kernel take 6 ms.  The DFF has none.
#define LONG_LIST 512

__device__ __forceinline__ double block_sum_256(double v, double* red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (t < s) red[t] += red[t + s]; __syncthreads(); }
  return red[0];
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
// Small systems (G, C, u and du of an instance fit the LDS of a workgroup): one workgroup per instance streams the instance's G, C, u, du with
// coalesced loads into LDS, then every thread sums one row out of LDS -- same row order, same additions as k_residual -- and the result leaves
// with coalesced stores.  (k_residual's threads walk their rows in HBM: neighbouring threads read addresses a row length apart, u[j] is a
// gather; 2.4 TB/s algorithmic at B = 8192.)
__global__ void __launch_bounds__(256) k_residual_lds(ResArgs a) {
  extern __shared__ double rl[];
  const int inst = blockIdx.x, t = threadIdx.x;
  if (!a.active[inst]) return;
  double *G = rl, *C = rl + a.nnz, *u = C + a.nnz, *du = u + a.n;
  const double* Gg = a.G + (size_t)inst * a.nnz;
  const double* Cg = a.C + (size_t)inst * a.nnz;
  for (int p = t; p < a.nnz; p += 256) { G[p] = Gg[p]; C[p] = Cg[p]; }
  for (int i = t; i < a.n; i += 256) { u[i] = a.u[(size_t)inst * a.n + i]; du[i] = a.du[(size_t)inst * a.n + i]; }
  __syncthreads();
  for (int i = t; i < a.n; i += 256) {
    double accC = 0.0, accG = 0.0;
    for (int p = a.rowptr[i]; p < a.rowptr[i + 1]; ++p) { const int j = a.colidx[p]; accC += C[p] * du[j]; accG += G[p] * u[j]; }
    a.r[(size_t)inst * a.n + i] = (accC + accG) - a.b[(size_t)inst * a.n + i];
  }
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

// Zero-fill of device words on the handle's stream, as a kernel of our own.  (A stream memset is implemented by the runtime with blit
// kernels of its own; under `rocprofv3 --pmc`, which serialises and re-queues the application's kernels, such a fill was observed to land
// AFTER the kernels queued behind it -- cadnip_dc_run then met the gamma = 1e9 left by the pivot analysis and every instance failed.  Kernels
// of one stream stay ordered among themselves.)
__global__ void __launch_bounds__(256) k_zero_words(unsigned* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
int dev_zero_async(CadnipHandle* h, void* p, size_t bytes) {
  if (!p || bytes == 0) return CADNIP_OK;
  if (bytes & 3) return CADNIP_BADARG;
  const size_t n = bytes / 4;
  const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(k_zero_words, dim3(grid), dim3(256), 0, h->stream, (unsigned*)p, n);
  return CADNIP_OK;
}

// Word copy as a kernel: how small host-pointer transfers travel between the mapped pinned staging area and device memory (api.hip).
__global__ void __launch_bounds__(256) k_copy_words(unsigned* dst, const unsigned* src, size_t n, int to_host) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
  if (to_host) __threadfence_system();
}
int dev_copy_async(CadnipHandle* h, void* dst, const void* src, size_t bytes, bool to_host) {
  if (bytes == 0) return CADNIP_OK;
  if (bytes & 3) return CADNIP_BADARG;
  const size_t n = bytes / 4;
  const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 1024);
  hipLaunchKernelGGL(k_copy_words, dim3(grid), dim3(256), 0, h->stream, (unsigned*)dst, (const unsigned*)src, n, to_host ? 1 : 0);
  return CADNIP_OK;
}

// Several small transfers / clears in ONE launch (cadnip_newton_step: four uploads and two clears, four downloads): segment blockIdx.y
// copies src -> dst word by word, or clears dst when src is null.
__global__ void __launch_bounds__(256) k_multi_words(MultiCopy m, int to_host) {
  const MultiCopy::Seg sg = m.seg[blockIdx.y];
  if (sg.src) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < sg.words; i += (size_t)gridDim.x * 256) sg.dst[i] = sg.src[i]; }
  else { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < sg.words; i += (size_t)gridDim.x * 256) sg.dst[i] = 0u; }
  if (to_host) __threadfence_system();
}
int dev_multi_async(CadnipHandle* h, const MultiCopy& m, bool to_host) {
  if (m.n <= 0) return CADNIP_OK;
  size_t most = 0;
  for (int k = 0; k < m.n; ++k) most = std::max(most, m.seg[k].words);
  const unsigned grid = (unsigned)std::min<size_t>(std::max<size_t>((most + 255) / 256, 1), 256);
  hipLaunchKernelGGL(k_multi_words, dim3(grid, (unsigned)m.n), dim3(256), 0, h->stream, m, to_host ? 1 : 0);
  return CADNIP_OK;
}

// ||x||_2 of every instance's vector (cadnip_newton_step's residual norm): one wave per instance
__global__ void __launch_bounds__(64) k_norm2(const double* x, int n, double* out) {
  const double* v = x + (size_t)blockIdx.x * n;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += v[i] * v[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[blockIdx.x] = sqrt(s);
}
int launch_norm2(CadnipHandle* h, const double* d_x, double* d_out) {
  hipLaunchKernelGGL(k_norm2, dim3(h->B), dim3(64), 0, h->stream, d_x, h->n, d_out);
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
int launch_residual(CadnipHandle* h, const double* d_du) {
  ProfScope ps(h, "residual");
  ResArgs a{h->d_G, h->d_C, h->d_b, h->d_u, d_du, h->d_rowptr, h->d_colidx, h->d_active, h->d_resid, h->B, h->n, h->nnz};
  long total = (long)h->B * h->n;
  const size_t lds = ((size_t)2 * h->nnz + 2 * h->n) * sizeof(double);
  if (lds <= 96 * 1024 && h->n_long_rows == 0 && !getenv("CADNIP_RESIDUAL_ROWS")) {
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_residual_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_residual_lds, dim3((unsigned)h->B), dim3(256), lds, h->stream, a);
    HIP_TRY(hipGetLastError());
    return CADNIP_OK;
  }
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
int launch_factor_solve_f2(CadnipHandle* h, const double* d_rhs, double* d_x);   // lu_f2.hip
int launch_factor_solve(CadnipHandle* h, bool fuse, const double* d_rhs, double* d_x) {
  // refactor of G + gamma C followed by the solve: the entry-program kernel (lu_f2.hip) when the circuit's tables fit into
  // LDS; CADNIP_LU_PLAIN=1 forces the level-by-level kernel below (diagnostic: the two must agree)
  if (fuse && !getenv("CADNIP_LU_PLAIN")) {
    const int rc = launch_factor_solve_f2(h, d_rhs, d_x);
    if (rc != 1) return rc;
  }
  return launch_lu(h, "lu_factor_solve", 1, 1, fuse, d_rhs, d_x);
}

}  // namespace cadnip
