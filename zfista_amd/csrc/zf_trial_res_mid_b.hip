// ZF_ACCEPT_RESOLVED solvers, chains of 16, PART 3: mid chains of 13 .. 15 trials, per pass and as passes ahead
#include "zf_trial_launch.h"

bool zf_launch_res_mid_a(bool nest, bool ahead, int len, int grid, hipStream_t st, const zf_step_args& a);

bool zf_launch_res_mid(const zf_trial_sel& v, bool ahead, int len, int grid, hipStream_t st, const zf_step_args& a) {
    if (!zf_have_s16_mid(v, len)) return false;
    if (len <= 12) return zf_launch_res_mid_a(v.nest, ahead, len, grid, st, a);
#define MID(LEN)                                                                     \
    case LEN:                                                                        \
        if (v.nest && ahead) ZF_LAUNCH_TRIAL_RES(true, false, 16, false, 3, LEN, true);     \
        else if (v.nest) ZF_LAUNCH_TRIAL_RES(true, false, 16, false, 3, LEN, false);        \
        else if (ahead) ZF_LAUNCH_TRIAL_RES(false, false, 16, false, 3, LEN, true);         \
        else ZF_LAUNCH_TRIAL_RES(false, false, 16, false, 3, LEN, false);                   \
        return true;
    switch (len) {
        MID(13)
        MID(14)
        MID(15)
    }
#undef MID
    return false;
}
