// chains of 8 / 4 / 2 and single trials of the separable problem (explicit sub_iters; S = 8 also serves return_all)
#include "zf_trial_launch.h"

void zf_launch_chain(const zf_trial_sel& v, int S, int part, int grid, hipStream_t st, const zf_step_args& a) {
    if (v.res) {   // (single trials only: zf_solver_create)
        if (S == 1 && part == 0) zf_launch_res_single(v, false, grid, st, a);
        return;
    }
#define CALL8_0(N, B, T) ZF_LAUNCH_TRIAL(true, N, B, T, 8, false, 0, 0)
#define CALL8_1(N, B, T) ZF_LAUNCH_TRIAL(true, N, B, T, 8, false, 1, 0)
#define CALL4_0(N, B, T) ZF_LAUNCH_TRIAL(true, N, B, T, 4, false, 0, 0)
#define CALL4_1(N, B, T) ZF_LAUNCH_TRIAL(true, N, B, T, 4, false, 1, 0)
#define CALL2_0(N, B, T) ZF_LAUNCH_TRIAL(true, N, B, T, 2, false, 0, 0)
#define CALL2_1(N, B, T) ZF_LAUNCH_TRIAL(true, N, B, T, 2, false, 1, 0)
#define CALL1_0(N, B, T) ZF_LAUNCH_TRIAL(true, N, B, T, 1, false, 0, 0)
    if (S == 8 && part == 0) ZF_SEL_NBT(v, CALL8_0);
    else if (S == 8) ZF_SEL_NBT(v, CALL8_1);
    else if (S == 4 && part == 0) ZF_SEL_NBT(v, CALL4_0);
    else if (S == 4) ZF_SEL_NBT(v, CALL4_1);
    else if (S == 2 && part == 0) ZF_SEL_NBT(v, CALL2_0);
    else if (S == 2) ZF_SEL_NBT(v, CALL2_1);
    else if (part == 0) ZF_SEL_NBT(v, CALL1_0);
}
