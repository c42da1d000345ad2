// chains of 16, PART 1: up to 8 fresh trials behind the lagging iterations (the 8-trial bodies), materialise-only passes
#include "zf_trial_launch.h"

void zf_launch_s16_short(const zf_trial_sel& v, int grid, hipStream_t st, const zf_step_args& a) {
    if (v.res) return zf_launch_res_short(v, grid, st, a);
#define CALL(N, B, T) ZF_LAUNCH_TRIAL(true, N, B, T, 16, false, 1, 0)
    ZF_SEL_NBT(v, CALL);
#undef CALL
}
