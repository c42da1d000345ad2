// chains of 16: the run-ahead full chain (zf_runahead_kernel): consecutive full-chain passes launched alternately
// on two streams, pass p + 1 running while pass p is finalised
#include "zf_trial_launch.h"

void zf_launch_s16_runahead(const zf_trial_sel& v, int grid, hipStream_t st, const zf_step_args& a) {
#define CALL(N, B, T) hipLaunchKernelGGL((zf_runahead_kernel<N, B, T>), dim3(grid), dim3(ZF_BLOCK), 0, st, a)
    ZF_SEL_NBT(v, CALL);
#undef CALL
}

// workgroups of the run-ahead kernel the device holds at once (0: could not be determined): two passes in flight never
// wait for a slot that only the other can free while every pass fits
int zf_runahead_capacity(const zf_trial_sel& v) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    hipError_t e = hipErrorUnknown;
#define CALL(N, B, T) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, zf_runahead_kernel<N, B, T>, ZF_BLOCK, 0)
    ZF_SEL_NBT(v, CALL);
#undef CALL
    if (e != hipSuccess) return 0;
    return per_cu * prop.multiProcessorCount;
}
