// chains of 16, PART 2: the general 16-trial body (any number of lagging iterations, 9 .. 15 fresh trials)
#include "zf_trial_launch.h"

void zf_launch_s16_general(const zf_trial_sel& v, int grid, hipStream_t st, const zf_step_args& a) {
    if (v.res) return zf_launch_res_general(v, grid, st, a);
#define CALL(N, B, T) ZF_LAUNCH_TRIAL(true, N, B, T, 16, false, 2, 0)
    ZF_SEL_NBT(v, CALL);
#undef CALL
}
