// chains of 16, PART 3: branch-free mid chains of 13 .. 15 trials (no box, nontemporal policy)
#include "zf_trial_launch.h"

bool zf_launch_s16_mid_a(bool nest, int len, int grid, hipStream_t st, const zf_step_args& a);

bool zf_have_s16_mid(const zf_trial_sel& v, int len) {
    return !v.box && v.nt && len >= ZF_MID_MIN && len <= ZF_MID_MAX;
}

bool zf_launch_s16_mid(const zf_trial_sel& v, int len, int grid, hipStream_t st, const zf_step_args& a) {
    if (!zf_have_s16_mid(v, len)) return false;
    if (v.res) return zf_launch_res_mid(v, false, len, grid, st, a);
    if (len <= 12) return zf_launch_s16_mid_a(v.nest, len, grid, st, a);
#define MID(LEN)                                                          \
    case LEN:                                                             \
        if (v.nest) ZF_LAUNCH_TRIAL(true, true, false, true, 16, false, 3, LEN);  \
        else ZF_LAUNCH_TRIAL(true, false, false, true, 16, false, 3, LEN);        \
        return true;
    switch (len) {
        MID(13)
        MID(14)
        MID(15)
    }
#undef MID
    return false;
}
