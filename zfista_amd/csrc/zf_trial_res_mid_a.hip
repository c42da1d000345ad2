// ZF_ACCEPT_RESOLVED solvers, chains of 16, PART 3: mid chains of 9 .. 12 trials, per pass and as passes ahead
#include "zf_trial_launch.h"

bool zf_launch_res_mid_a(bool nest, bool ahead, int len, int grid, hipStream_t st, const zf_step_args& a) {
#define MID(LEN)                                                                     \
    case LEN:                                                                        \
        if (nest && ahead) ZF_LAUNCH_TRIAL_RES(true, false, 16, false, 3, LEN, true);       \
        else if (nest) ZF_LAUNCH_TRIAL_RES(true, false, 16, false, 3, LEN, false);          \
        else if (ahead) ZF_LAUNCH_TRIAL_RES(false, false, 16, false, 3, LEN, true);         \
        else ZF_LAUNCH_TRIAL_RES(false, false, 16, false, 3, LEN, false);                   \
        return true;
    switch (len) {
        MID(9)
        MID(10)
        MID(11)
        MID(12)
    }
#undef MID
    return false;
}
