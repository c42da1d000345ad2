// chains of 16, PART 3: branch-free mid chains of 9 .. 12 trials (no box, nontemporal policy; other variants take the
// general body)
#include "zf_trial_launch.h"

bool zf_launch_s16_mid_a(bool nest, int len, int grid, hipStream_t st, const zf_step_args& a) {
#define MID(LEN)                                                          \
    case LEN:                                                             \
        if (nest) ZF_LAUNCH_TRIAL(true, true, false, true, 16, false, 3, LEN);  \
        else ZF_LAUNCH_TRIAL(true, false, false, true, 16, false, 3, LEN);      \
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
