// chains of 16, passes AHEAD at kernel granularity: the full chain (PART 0) on the head the host expects, rows stored
// plainly (zf_trial_kernel<..., AHEAD>; nontemporal policy only), and the stand-alone finalisation of such a pass
#include "zf_trial_launch.h"

void zf_launch_s16_ahead_full(const zf_trial_sel& v, int grid, hipStream_t st, const zf_step_args& a) {
    if (v.res) return zf_launch_res_full(v, true, grid, st, a);
#define AH(N, B) hipLaunchKernelGGL((zf_trial_kernel<true, N, B, true, 16, false, 0, 0, true>), dim3(grid), dim3(ZF_BLOCK), 0, st, a)
    if (v.nest && v.box) AH(true, true);
    else if (v.nest) AH(true, false);
    else if (v.box) AH(false, true);
    else AH(false, false);
#undef AH
}

// rows -> packs (-> decide, unsharded) of a pass ahead: a.fin_ng workgroups when the rows are grouped, one otherwise
void zf_launch_s16_tail(hipStream_t st, const zf_step_args& a) {
    const int wgs = a.fin_gsz > 1 ? a.fin_ng : 1;
    hipLaunchKernelGGL(zf_tail_kernel<16>, dim3(wgs), dim3(128), 0, st, a);
}
