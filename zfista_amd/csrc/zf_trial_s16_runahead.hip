// chains of 16: the run-ahead full chain (zf_runahead_kernel): consecutive full-chain passes launched alternately
// on two streams, pass p + 1 running while pass p is finalised
#include "zf_trial_launch.h"

void zf_launch_s16_runahead(const zf_trial_sel& v, int grid, hipStream_t st, const zf_step_args& a) {
    if (v.res) return zf_launch_res_runahead(v, grid, st, a);
    // (no box variant: zf_solver_create does not enable run-ahead passes for clipped problems)
    if (v.nest && v.nt) hipLaunchKernelGGL((zf_runahead_kernel<true, false, true>), dim3(grid), dim3(ZF_BLOCK), 0, st, a);
    else if (v.nest) hipLaunchKernelGGL((zf_runahead_kernel<true, false, false>), dim3(grid), dim3(ZF_BLOCK), 0, st, a);
    else if (v.nt) hipLaunchKernelGGL((zf_runahead_kernel<false, false, true>), dim3(grid), dim3(ZF_BLOCK), 0, st, a);
    else hipLaunchKernelGGL((zf_runahead_kernel<false, false, false>), dim3(grid), dim3(ZF_BLOCK), 0, st, a);
}

// workgroups of the run-ahead kernel the device holds at once (0: could not be determined): two passes in flight never
// wait for a slot that only the other can free while every pass fits
int zf_runahead_capacity(const zf_trial_sel& v) {
    if (v.res) return zf_res_runahead_capacity(v);
    // (asked once per variant and process: every solver of a one-round size asks at creation)
    static int cache[4] = {-1, -1, -1, -1};
    const int slot = (v.nest ? 2 : 0) | (v.nt ? 1 : 0);
    if (cache[slot] >= 0) return cache[slot];
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    hipError_t e = hipErrorUnknown;
    if (v.nest && v.nt) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, zf_runahead_kernel<true, false, true>, ZF_BLOCK, 0);
    else if (v.nest) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, zf_runahead_kernel<true, false, false>, ZF_BLOCK, 0);
    else if (v.nt) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, zf_runahead_kernel<false, false, true>, ZF_BLOCK, 0);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, zf_runahead_kernel<false, false, false>, ZF_BLOCK, 0);
    if (e != hipSuccess) return 0;
    cache[slot] = per_cu * prop.multiProcessorCount;
    return cache[slot];
}
