// ZF_ACCEPT_RESOLVED solvers, chains of 16, PART 2 (the general body), and single trials (sub_iters 1; return_all)
#include "zf_trial_launch.h"

void zf_launch_res_general(const zf_trial_sel& v, int grid, hipStream_t st, const zf_step_args& a) {
#define CALL(N, B) ZF_LAUNCH_TRIAL_RES(N, B, 16, false, 2, 0, false)
    ZF_SEL_NB(v, CALL);
#undef CALL
}

void zf_launch_res_single(const zf_trial_sel& v, bool hist, int grid, hipStream_t st, const zf_step_args& a) {
#define PLAIN(N, B) ZF_LAUNCH_TRIAL_RES(N, B, 1, false, 0, 0, false)
#define HIST(N, B) ZF_LAUNCH_TRIAL_RES(N, B, 1, true, 0, 0, false)
    if (hist) ZF_SEL_NB(v, HIST);
    else ZF_SEL_NB(v, PLAIN);
#undef PLAIN
#undef HIST
}
