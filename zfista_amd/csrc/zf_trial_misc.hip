// history-recording kernels (streaming return_all) and the trial kernel with the gradient vector in HBM (least squares)
#include "zf_trial_launch.h"

void zf_launch_hist(const zf_trial_sel& v, bool grad_inline, int S, int part, int grid, hipStream_t st, const zf_step_args& a) {
    if (v.res) {   // (separable problem, single trials only: zf_solver_create)
        if (grad_inline && S == 1 && part == 0) zf_launch_res_single(v, true, grid, st, a);
        return;
    }
#define H(GI, SS, PART)                                                          \
    do {                                                                         \
        if (v.nest && v.box) ZF_LAUNCH_TRIAL(GI, true, true, true, SS, true, PART, 0);        \
        else if (v.nest) ZF_LAUNCH_TRIAL(GI, true, false, true, SS, true, PART, 0);           \
        else if (v.box) ZF_LAUNCH_TRIAL(GI, false, true, true, SS, true, PART, 0);            \
        else ZF_LAUNCH_TRIAL(GI, false, false, true, SS, true, PART, 0);                      \
    } while (0)
    if (grad_inline && S == 8 && part == 0) H(true, 8, 0);
    else if (grad_inline && S == 8) H(true, 8, 1);
    else if (grad_inline && part == 0) H(true, 1, 0);
    else if (!grad_inline && part == 0) H(false, 1, 0);
#undef H
}

void zf_launch_vec(const zf_trial_sel& v, int grid, hipStream_t st, const zf_step_args& a) {
#define CALL(N, B, T) ZF_LAUNCH_TRIAL(false, N, B, T, 1, false, 0, 0)
    ZF_SEL_NBT(v, CALL);
#undef CALL
}
