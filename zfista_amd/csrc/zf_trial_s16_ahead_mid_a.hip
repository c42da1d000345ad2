// chains of 16, passes AHEAD at kernel granularity: branch-free mid chains of 9 .. 12 trials (no box, nontemporal policy)
#include "zf_trial_launch.h"

bool zf_launch_s16_ahead_mid_a(bool nest, int len, int grid, hipStream_t st, const zf_step_args& a) {
#define MID(LEN)                                                                                                                   \
    case LEN:                                                                                                                      \
        if (nest) hipLaunchKernelGGL((zf_trial_kernel<true, true, false, true, 16, false, 3, LEN, true>), dim3(grid), dim3(ZF_BLOCK), 0, st, a);  \
        else hipLaunchKernelGGL((zf_trial_kernel<true, false, false, true, 16, false, 3, LEN, true>), dim3(grid), dim3(ZF_BLOCK), 0, st, a);      \
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
