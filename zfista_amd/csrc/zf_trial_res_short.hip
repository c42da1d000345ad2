// ZF_ACCEPT_RESOLVED solvers, chains of 16, PART 1: the 8-trial bodies (up to 8 fresh trials behind lagging iterations)
#include "zf_trial_launch.h"

void zf_launch_res_short(const zf_trial_sel& v, int grid, hipStream_t st, const zf_step_args& a) {
#define CALL(N, B) ZF_LAUNCH_TRIAL_RES(N, B, 16, false, 1, 0, false)
    ZF_SEL_NB(v, CALL);
#undef CALL
}
