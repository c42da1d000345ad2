// chains of 16, PART 0: the full chain
#include "zf_trial_launch.h"


void zf_launch_s16_full(const zf_trial_sel& v, int grid, hipStream_t st, const zf_step_args& a) {
    if (v.res) return zf_launch_res_full(v, false, grid, st, a);
#define CALL(N, B, T) ZF_LAUNCH_TRIAL(true, N, B, T, 16, false, 0, 0)
    ZF_SEL_NBT(v, CALL);
#undef CALL
}
