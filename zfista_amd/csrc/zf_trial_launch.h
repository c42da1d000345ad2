// zf_trial_launch.h - host-side launchers of the shape-specific trial kernels.
//
// zf_trial_kernel has ~130 instantiations (chain length x shape x momentum x box x store policy x history); in one
// translation unit they compiled for 2.5 minutes on one core.  Each group of kernels now lives in a translation unit
// of its own (zf_trial_*.hip) behind one plain function; zf_solver.hip only dispatches.
#pragma once
#include "zf_kernels_step.h"

struct zf_trial_sel {
    bool nest, box, nt;
    bool res = false;   // ZF_ACCEPT_RESOLVED: the kernels that accumulate f(x+) - f(y) (zf_elem_diag<..., RES>; nontemporal
                        // policy, chains of 16 and single trials only - zf_solver_create holds such a solver to that)
};

// chains of 16 (the default): the full chain (PART 0), the short bodies (PART 1), the general body (PART 2),
// the branch-free mid chains (PART 3, `len` trials; false: no such kernel for this variant - the caller launches the
// general body instead)
void zf_launch_s16_full(const zf_trial_sel& v, int grid, hipStream_t st, const zf_step_args& a);
void zf_launch_s16_short(const zf_trial_sel& v, int grid, hipStream_t st, const zf_step_args& a);
void zf_launch_s16_general(const zf_trial_sel& v, int grid, hipStream_t st, const zf_step_args& a);
bool zf_launch_s16_mid(const zf_trial_sel& v, int len, int grid, hipStream_t st, const zf_step_args& a);
bool zf_have_s16_mid(const zf_trial_sel& v, int len);
// passes AHEAD at kernel granularity (nontemporal policy): the full chain and the mid chains on the head the host expects,
// rows stored plainly; zf_launch_s16_tail = rows -> packs (-> decide) of such a pass (zf_tail_kernel)
void zf_launch_s16_ahead_full(const zf_trial_sel& v, int grid, hipStream_t st, const zf_step_args& a);
bool zf_launch_s16_ahead_mid(const zf_trial_sel& v, int len, int grid, hipStream_t st, const zf_step_args& a);
void zf_launch_s16_tail(hipStream_t st, const zf_step_args& a);
// the run-ahead kernels (zf_runahead_kernel): len = 16 the full chain (every variant), ZF_MID_MIN .. ZF_MID_MAX a mid chain
// (no box, nontemporal policy - where the per-pass mid chains exist); false / 0: no such kernel.  The capacity is the
// number of workgroups of that kernel the device holds at once.
bool zf_launch_s16_runahead(const zf_trial_sel& v, int len, int grid, hipStream_t st, const zf_step_args& a);
int zf_runahead_capacity(const zf_trial_sel& v, int len);
// ONE run-ahead kernel: launched (grid > 0: returns 1), or asked of the runtime how many of its workgroups the device
// holds at once (grid == 0: returns that number, cached per kernel and process; 0 when it cannot be determined)
template <bool N, bool B, bool NT, bool RES, int L>
inline int zf_ra_kernel_op(int grid, hipStream_t st, const zf_step_args& a) {
    if (grid > 0) {
        hipLaunchKernelGGL((zf_runahead_kernel<N, B, NT, RES, L>), dim3(grid), dim3(ZF_BLOCK), 0, st, a);
        return 1;
    }
    static int cached = -1;
    if (cached >= 0) return cached;
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, zf_runahead_kernel<N, B, NT, RES, L>, ZF_BLOCK, 0) != hipSuccess) return 0;
    cached = per_cu * prop.multiProcessorCount;
    return cached;
}
// (by translation unit; -1: not one of its kernels)
int zf_ra_op_mid_a(const zf_trial_sel& v, int len, int grid, hipStream_t st, const zf_step_args& a);
int zf_ra_op_mid_b(const zf_trial_sel& v, int len, int grid, hipStream_t st, const zf_step_args& a);
int zf_ra_op_res(const zf_trial_sel& v, int len, int grid, hipStream_t st, const zf_step_args& a);
int zf_ra_op_res_mid(const zf_trial_sel& v, int len, int grid, hipStream_t st, const zf_step_args& a);
// chains of 8 / 4 / 2 (part 0: full chain, part 1: every other shape) and single trials (S = 1, part 0)
void zf_launch_chain(const zf_trial_sel& v, int S, int part, int grid, hipStream_t st, const zf_step_args& a);
// history-recording kernels (streaming return_all; nontemporal policy only): separable S = 8 / 1, gradient vector S = 1
void zf_launch_hist(const zf_trial_sel& v, bool grad_inline, int S, int part, int grid, hipStream_t st, const zf_step_args& a);
// gradient vector read from HBM (least squares), S = 1
void zf_launch_vec(const zf_trial_sel& v, int grid, hipStream_t st, const zf_step_args& a);

// CALL(NEST, BOX, NT) for the variant `v`
#define ZF_SEL_NBT(v, CALL)                                  \
    do {                                                     \
        if ((v).nest && (v).box && (v).nt) { CALL(true, true, true); }          \
        else if ((v).nest && (v).box) { CALL(true, true, false); }              \
        else if ((v).nest && (v).nt) { CALL(true, false, true); }               \
        else if ((v).nest) { CALL(true, false, false); }                        \
        else if ((v).box && (v).nt) { CALL(false, true, true); }                \
        else if ((v).box) { CALL(false, true, false); }                         \
        else if ((v).nt) { CALL(false, false, true); }                          \
        else { CALL(false, false, false); }                                     \
    } while (0)
#define ZF_LAUNCH_TRIAL(GI, N, B, T, S, HIST, PART, L) \
    hipLaunchKernelGGL((zf_trial_kernel<GI, N, B, T, S, HIST, PART, L>), dim3(grid), dim3(ZF_BLOCK), 0, st, a)
// the same kernel of a ZF_ACCEPT_RESOLVED solver (separable problem, nontemporal policy); AH: the pass-ahead variant
#define ZF_LAUNCH_TRIAL_RES(N, B, S, HIST, PART, L, AH) \
    hipLaunchKernelGGL((zf_trial_kernel<true, N, B, true, S, HIST, PART, L, AH, true>), dim3(grid), dim3(ZF_BLOCK), 0, st, a)
#define ZF_SEL_NB(v, CALL)                                \
    do {                                                  \
        if ((v).nest && (v).box) { CALL(true, true); }    \
        else if ((v).nest) { CALL(true, false); }         \
        else if ((v).box) { CALL(false, true); }          \
        else { CALL(false, false); }                      \
    } while (0)

// ZF_ACCEPT_RESOLVED solvers (v.res): every launcher above forwards to these (zf_trial_res_*.hip)
void zf_launch_res_full(const zf_trial_sel& v, bool ahead, int grid, hipStream_t st, const zf_step_args& a);
void zf_launch_res_short(const zf_trial_sel& v, int grid, hipStream_t st, const zf_step_args& a);
void zf_launch_res_general(const zf_trial_sel& v, int grid, hipStream_t st, const zf_step_args& a);
bool zf_launch_res_mid(const zf_trial_sel& v, bool ahead, int len, int grid, hipStream_t st, const zf_step_args& a);
void zf_launch_res_single(const zf_trial_sel& v, bool hist, int grid, hipStream_t st, const zf_step_args& a);
