// fused_team.hip -- instantiates k_fteam<NW> (fused_team_kernel.hpp): the fused Newton kernel with a team of NW waves per sweep instance.
#include "fused_team_kernel.hpp"

namespace cadnip {

template <int NW>
static int fteam_launch_one(int grid, size_t shmem, hipStream_t stream, const F2Args& f) {
  if (shmem > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_fteam<NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL((k_fteam<NW, false>), dim3(grid), dim3(64 * NW), shmem, stream, f);
  return CADNIP_OK;
}

int fteam_launch(int nw, int grid, size_t shmem, hipStream_t stream, const F2Args& f) {
  return nw == 4 ? fteam_launch_one<4>(grid, shmem, stream, f) : fteam_launch_one<2>(grid, shmem, stream, f);
}

int fteam_launch_step(int grid, size_t shmem, hipStream_t stream, const F2Args& f) {
  if (shmem > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_fteam<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL((k_fteam<4, true>), dim3(grid), dim3(256), shmem, stream, f);
  return CADNIP_OK;
}

#ifdef CADNIP_TRACE
int trace_read_team(unsigned long long* sum, unsigned long long* cnt, int reset, int wave) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(sum, HIP_SYMBOL(g_trace_sum), 64 * sizeof(unsigned long long)));
  HIP_TRY(hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_trace_cnt), 64 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[64] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_sum), z, sizeof(z)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_cnt), z, sizeof(z)));
  }
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_wave), &wave, sizeof(int)));
  return CADNIP_OK;
}
#endif

}  // namespace cadnip

#ifdef CADNIP_TRACE
// diagnostic library only: cycle timeline of wave `wave` of the first team (tools/trace_fused2.py --team)
extern "C" int cadnip_debug_trace_team(unsigned long long* sum, unsigned long long* cnt, int reset, int wave) { return cadnip::trace_read_team(sum, cnt, reset, wave); }
#endif
