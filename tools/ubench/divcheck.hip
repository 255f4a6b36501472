// divcheck.hip -- is fast_div (tran_ctrl.hpp: v_rcp_f64 + two Newton steps + one residual correction) the correctly rounded quotient
// for normal-range operands?  Compares it bit for bit with IEEE division over 2^28 random pairs drawn from several magnitude ranges.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ double fast_div(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);
}
__device__ uint64_t rng(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
__global__ void k(unsigned long long* bad, unsigned long long* worst, int spread) {
  uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
  unsigned long long nb = 0, w = 0;
  for (int i = 0; i < 4096; ++i) {
    uint64_t x = rng(s), y = rng(s);
    // mantissas random, exponents within +-spread of 1.0, random signs
    int ea = 1023 + (int)(rng(s) % (2 * spread + 1)) - spread, eb = 1023 + (int)(rng(s) % (2 * spread + 1)) - spread;
    double a = __longlong_as_double((long long)((x & 0x800FFFFFFFFFFFFFull) | ((uint64_t)ea << 52)));
    double b = __longlong_as_double((long long)((y & 0x800FFFFFFFFFFFFFull) | ((uint64_t)eb << 52)));
    double q0 = a / b, q1 = fast_div(a, b);
    long long d = __double_as_longlong(q0) - __double_as_longlong(q1);
    if (d < 0) d = -d;
    if (d) { ++nb; if ((unsigned long long)d > w) w = d; }
  }
  atomicAdd(bad, nb); atomicMax(worst, w);
}
int main() {
  unsigned long long *bad, *worst, h[2];
  hipMalloc(&bad, 8); hipMalloc(&worst, 8);
  for (int spread : {0, 3, 40, 300, 500}) {
    hipMemset(bad, 0, 8); hipMemset(worst, 0, 8);
    k<<<1024, 64>>>(bad, worst, spread);
    hipDeviceSynchronize();
    hipMemcpy(&h[0], bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&h[1], worst, 8, hipMemcpyDeviceToHost);
    printf("exponent spread +-%3d: %llu of %llu quotients differ from IEEE division, worst %llu ulp\n", spread, h[0], 1024ull * 64 * 4096, h[1]);
  }
  return 0;
}
