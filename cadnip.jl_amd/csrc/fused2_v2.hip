// fused2_v2.hip -- instantiates k_fused2<WPB, DC, 2> (fused2_kernel.hpp) for WPB = 1, 2, 4, 8 and both modes.
#include "fused2_kernel.hpp"

namespace cadnip {

template <int W, bool D>
static int f2_launch_one(int grid, size_t shmem, hipStream_t stream, const F2Args& f) {
  if (shmem > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_fused2<W, D, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL((k_fused2<W, D, 2>), dim3(grid), dim3(64 * W), shmem, stream, f);
  return CADNIP_OK;
}

template <> int f2_launch_variant<2>(int wpb, bool dc, int grid, size_t shmem, hipStream_t stream, const F2Args& f) {
  if (dc) return wpb == 8 ? f2_launch_one<8, true>(grid, shmem, stream, f) : wpb == 4 ? f2_launch_one<4, true>(grid, shmem, stream, f)
               : wpb == 2 ? f2_launch_one<2, true>(grid, shmem, stream, f) : f2_launch_one<1, true>(grid, shmem, stream, f);
  return wpb == 8 ? f2_launch_one<8, false>(grid, shmem, stream, f) : wpb == 4 ? f2_launch_one<4, false>(grid, shmem, stream, f)
       : wpb == 2 ? f2_launch_one<2, false>(grid, shmem, stream, f) : f2_launch_one<1, false>(grid, shmem, stream, f);
}

}  // namespace cadnip
