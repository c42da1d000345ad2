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

int zf_ra_op_res(const zf_trial_sel& v, int len, int grid, hipStream_t st, const zf_step_args& a) {
    if (len != ZF_MAX_SUB) return -1;
#define FULL(N, B) return zf_ra_kernel_op<N, B, true, true, 0>(grid, st, a)
    ZF_SEL_NB(v, FULL);
#undef FULL
    return -1;
}
