// ZF_ACCEPT_RESOLVED solvers, chains of 16: the full chain per pass, as a pass ahead and as a run-ahead pass
#include "zf_trial_launch.h"

void zf_launch_res_full(const zf_trial_sel& v, bool ahead, int grid, hipStream_t st, const zf_step_args& a) {
#define PLAIN(N, B) ZF_LAUNCH_TRIAL_RES(N, B, 16, false, 0, 0, false)
#define AHEAD(N, B) ZF_LAUNCH_TRIAL_RES(N, B, 16, false, 0, 0, true)
    if (ahead) ZF_SEL_NB(v, AHEAD);
    else ZF_SEL_NB(v, PLAIN);
#undef PLAIN
#undef AHEAD
}

void zf_launch_res_runahead(const zf_trial_sel& v, int grid, hipStream_t st, const zf_step_args& a) {
    if (v.nest) hipLaunchKernelGGL((zf_runahead_kernel<true, false, true, true>), dim3(grid), dim3(ZF_BLOCK), 0, st, a);
    else hipLaunchKernelGGL((zf_runahead_kernel<false, false, true, true>), dim3(grid), dim3(ZF_BLOCK), 0, st, a);
}

int zf_res_runahead_capacity(const zf_trial_sel& v) {
    static int cache[2] = {-1, -1};
    const int slot = v.nest ? 1 : 0;
    if (cache[slot] >= 0) return cache[slot];
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    hipError_t e;
    if (v.nest) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, zf_runahead_kernel<true, false, true, true>, ZF_BLOCK, 0);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, zf_runahead_kernel<false, false, true, true>, ZF_BLOCK, 0);
    if (e != hipSuccess) return 0;
    cache[slot] = per_cu * prop.multiProcessorCount;
    return cache[slot];
}
