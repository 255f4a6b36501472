// fused2_v0.hip -- instantiates k_fused2<WPB, DC, 0> (fused2_kernel.hpp) for WPB = 1, 2, 4, 8 and both modes.
#include "fused2_kernel.hpp"

namespace cadnip {

template <int W, bool D>
static int f2_launch_one(int grid, size_t shmem, hipStream_t stream, const F2Args& f) {
  if (shmem > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_fused2<W, D, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL((k_fused2<W, D, 0>), dim3(grid), dim3(64 * W), shmem, stream, f);
  return CADNIP_OK;
}

template <> int f2_launch_variant<0>(int wpb, bool dc, int grid, size_t shmem, hipStream_t stream, const F2Args& f) {
  if (dc) return wpb == 8 ? f2_launch_one<8, true>(grid, shmem, stream, f) : wpb == 4 ? f2_launch_one<4, true>(grid, shmem, stream, f)
               : wpb == 2 ? f2_launch_one<2, true>(grid, shmem, stream, f) : f2_launch_one<1, true>(grid, shmem, stream, f);
  return wpb == 8 ? f2_launch_one<8, false>(grid, shmem, stream, f) : wpb == 4 ? f2_launch_one<4, false>(grid, shmem, stream, f)
       : wpb == 2 ? f2_launch_one<2, false>(grid, shmem, stream, f) : f2_launch_one<1, false>(grid, shmem, stream, f);
}

// the trace counters are per translation unit (devices.hpp): read the ones of this variant, the benchmark's
#ifdef CADNIP_TRACE
int trace_read(unsigned long long* sum, unsigned long long* cnt, int reset) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(sum, HIP_SYMBOL(g_trace_sum), 64 * sizeof(unsigned long long)));
  HIP_TRY(hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_trace_cnt), 64 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[64] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_sum), z, sizeof(z)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_cnt), z, sizeof(z)));
  }
  return CADNIP_OK;
}
#endif

}  // namespace cadnip

#ifdef CADNIP_TRACE
// diagnostic library only: not part of include/cadnip_hip.h
extern "C" int cadnip_debug_trace(unsigned long long* sum, unsigned long long* cnt, int reset) { return cadnip::trace_read(sum, cnt, reset); }
#endif
