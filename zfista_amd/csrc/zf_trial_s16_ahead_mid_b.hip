// chains of 16, passes AHEAD at kernel granularity: branch-free mid chains of 13 .. 15 trials (no box, nontemporal policy)
#include "zf_trial_launch.h"

bool zf_launch_s16_ahead_mid_a(bool nest, int len, int grid, hipStream_t st, const zf_step_args& a);

// (the same variants as the per-pass mid chains: zf_have_s16_mid)
bool zf_launch_s16_ahead_mid(const zf_trial_sel& v, int len, int grid, hipStream_t st, const zf_step_args& a) {
    if (!zf_have_s16_mid(v, len)) return false;
    if (v.res) return zf_launch_res_mid(v, true, len, grid, st, a);
    if (len <= 12) return zf_launch_s16_ahead_mid_a(v.nest, len, grid, st, a);
#define MID(LEN)                                                                                                                   \
    case LEN:                                                                                                                      \
        if (v.nest) hipLaunchKernelGGL((zf_trial_kernel<true, true, false, true, 16, false, 3, LEN, true>), dim3(grid), dim3(ZF_BLOCK), 0, st, a); \
        else hipLaunchKernelGGL((zf_trial_kernel<true, false, false, true, 16, false, 3, LEN, true>), dim3(grid), dim3(ZF_BLOCK), 0, st, a);     \
        return true;
    switch (len) {
        MID(13)
        MID(14)
        MID(15)
    }
#undef MID
    return false;
}
