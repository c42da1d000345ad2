// chains of 16: the persistent multi-pass kernel of the full chain (zf_persist_kernel)
#include "zf_trial_launch.h"

void zf_launch_s16_persist(const zf_trial_sel& v, int grid, hipStream_t st, const zf_step_args& a, int npass, unsigned spin_limit) {
#define CALL(N, B, T) hipLaunchKernelGGL((zf_persist_kernel<N, B, T>), dim3(grid), dim3(ZF_BLOCK), 0, st, a, npass, spin_limit)
    ZF_SEL_NBT(v, CALL);
#undef CALL
}

int zf_persist_capacity(const zf_trial_sel& v) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    hipError_t e = hipErrorUnknown;
#define CALL(N, B, T) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, zf_persist_kernel<N, B, T>, ZF_BLOCK, 0)
    ZF_SEL_NBT(v, CALL);
#undef CALL
    if (e != hipSuccess) return 0;
    return per_cu * prop.multiProcessorCount;
}
