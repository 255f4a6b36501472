// lat.hip -- single-wave latency microbenchmarks on gfx950: what one wave alone on a SIMD pays per dependent operation.
// Build: hipcc --offload-arch=gfx950 -O3 -o lat lat.hip ; run on the GPU box.  Numbers feed DESIGN.md (team kernel).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#define N 512
__device__ __forceinline__ double fast_div(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);
}
template <int CHAINS> __global__ void k_fma(double* out, long long* cyc, double a, double b) {
  double x[CHAINS];
  for (int c = 0; c < CHAINS; ++c) x[c] = a + c + threadIdx.x;
  long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = fma(x[c], b, a);
  }
  long long t1 = clock64();
  double s = 0; for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[threadIdx.x] = s; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_div(double* out, long long* cyc, double a, double b, int which) {
  double x = a + threadIdx.x;
  long long t0 = clock64();
  if (which == 0) { for (int i = 0; i < N; ++i) x = x / b + a; }
  else if (which == 1) { for (int i = 0; i < N; ++i) x = fast_div(x, b) + a; }
  else if (which == 2) { for (int i = 0; i < N; ++i) x = sqrt(x) + a; }
  else if (which == 3) { for (int i = 0; i < N; ++i) x = exp(x * 1e-3) + a; }
  else if (which == 4) { for (int i = 0; i < N; ++i) x = log(x) + a; }
  else { for (int i = 0; i < N; ++i) x = pow(x, 0.33) + a; }
  long long t1 = clock64();
  out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_lds(double* out, long long* cyc, int which) {
  __shared__ int idx[1024];
  __shared__ double val[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) { idx[i] = (i * 17 + 5) & 1023; val[i] = 0.0; }
  __syncthreads();
  int p = threadIdx.x; double acc = 0;
  long long t0 = clock64();
  if (which == 0) { for (int i = 0; i < N; ++i) p = idx[p]; }                          // dependent LDS read chain
  else if (which == 1) { for (int i = 0; i < N; ++i) { atomicAdd(&val[(p + i) & 1023], 1.0); } }   // independent LDS fp64 atomics (issue rate)
  else if (which == 2) { for (int i = 0; i < N; ++i) { atomicAdd(&val[p], 1.0); acc += val[(p + 64) & 1023]; p = (p + 1) & 1023; } }   // atomic then read
  else if (which == 3) { for (int i = 0; i < N; ++i) { __syncthreads(); } }
  else if (which == 4) { for (int i = 0; i < N; ++i) { val[p] = acc; __syncthreads(); acc += val[(p + 64) & 1023]; } }   // write -> barrier -> read (a level of the LU)
  long long t1 = clock64();
  out[threadIdx.x] = p + acc + val[threadIdx.x]; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_sload(double* out, long long* cyc, const int* __restrict__ tab) {
  typedef const __attribute__((address_space(4))) int* CP;
  CP t = (CP)tab;
  int p = 0;
  long long t0 = clock64();
  for (int i = 0; i < N; ++i) p = t[p];        // dependent scalar loads (constant cache)
  long long t1 = clock64();
  out[threadIdx.x] = p; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_gload(double* out, long long* cyc, const int* tab) {
  int p = threadIdx.x & 63;
  long long t0 = clock64();
  for (int i = 0; i < N; ++i) p = tab[p];      // dependent vector loads (L1 / L2 hits)
  long long t1 = clock64();
  out[threadIdx.x] = p; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_readlane(double* out, long long* cyc, double a) {
  double x = a + threadIdx.x;
  long long t0 = clock64();
  for (int i = 0; i < N; ++i) {
    int lo = __builtin_amdgcn_readlane(__double2loint(x), i & 63), hi = __builtin_amdgcn_readlane(__double2hiint(x), i & 63);
    x = fma(__hiloint2double(hi, lo), 0.5, x);
  }
  long long t1 = clock64();
  out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; long long* cyc; int* tab;
  hipMalloc(&out, 4096 * 8); hipMalloc(&cyc, 64); hipMalloc(&tab, 4096 * 4);
  int h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (i * 17 + 5) & 1023;
  hipMemcpy(tab, h, sizeof(h), hipMemcpyHostToDevice);
  long long c;
  auto rd = [&](const char* what, double per) { hipDeviceSynchronize(); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-58s %8.1f cycles per %s\n", what, (double)c / N, per ? "iteration" : "iteration"); };
  for (int rep = 0; rep < 2; ++rep) {
    k_fma<1><<<1, 64>>>(out, cyc, 1.0, 0.999); rd("fp64 fma, 1 dependent chain", 1);
    k_fma<2><<<1, 64>>>(out, cyc, 1.0, 0.999); rd("fp64 fma, 2 independent chains (per iteration = 2 fma)", 1);
    k_fma<4><<<1, 64>>>(out, cyc, 1.0, 0.999); rd("fp64 fma, 4 independent chains (per iteration = 4 fma)", 1);
    k_fma<8><<<1, 64>>>(out, cyc, 1.0, 0.999); rd("fp64 fma, 8 independent chains (per iteration = 8 fma)", 1);
    k_fma<1><<<1, 128>>>(out, cyc, 1.0, 0.999); rd("fp64 fma, 1 chain, 2 waves in the workgroup", 1);
    k_fma<1><<<1, 512>>>(out, cyc, 1.0, 0.999); rd("fp64 fma, 1 chain, 8 waves (2 per SIMD)", 1);
    k_div<<<1, 64>>>(out, cyc, 1.0, 1.7, 0); rd("x / b + a  (IEEE division), dependent", 1);
    k_div<<<1, 64>>>(out, cyc, 1.0, 1.7, 1); rd("fast_div(x, b) + a, dependent", 1);
    k_div<<<1, 64>>>(out, cyc, 1.0, 1.7, 2); rd("sqrt(x) + a, dependent", 1);
    k_div<<<1, 64>>>(out, cyc, 1.0, 1.7, 3); rd("exp(x) + a, dependent", 1);
    k_div<<<1, 64>>>(out, cyc, 1.0, 1.7, 4); rd("log(x) + a, dependent", 1);
    k_div<<<1, 64>>>(out, cyc, 1.0, 1.7, 5); rd("pow(x, .33) + a, dependent", 1);
    k_lds<<<1, 64>>>(out, cyc, 0); rd("LDS read -> address -> LDS read (dependent chain)", 1);
    k_lds<<<1, 64>>>(out, cyc, 1); rd("LDS fp64 atomic add, independent addresses (issue)", 1);
    k_lds<<<1, 64>>>(out, cyc, 2); rd("LDS fp64 atomic add + independent read", 1);
    k_lds<<<1, 256>>>(out, cyc, 3); rd("__syncthreads, 4 waves", 1);
    k_lds<<<1, 64>>>(out, cyc, 3); rd("__syncthreads, 1 wave", 1);
    k_lds<<<1, 256>>>(out, cyc, 4); rd("LDS write -> __syncthreads (4 waves) -> LDS read", 1);
    k_sload<<<1, 64>>>(out, cyc, tab); rd("scalar load -> address -> scalar load (constant cache)", 1);
    k_gload<<<1, 64>>>(out, cyc, tab); rd("global load -> address -> global load (L1 hit)", 1);
    k_readlane<<<1, 64>>>(out, cyc, 1.0); rd("2 x v_readlane + fma, dependent", 1);
  }
  return 0;
}
