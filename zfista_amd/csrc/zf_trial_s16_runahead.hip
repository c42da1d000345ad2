// chains of 16: the run-ahead passes (zf_runahead_kernel): consecutive passes the host predicts exactly, launched alternately
// on two streams, pass p + 1 running while pass p is finalised.  Here: the full chain of every variant and the dispatch
// over the translation units of the mid chains.
#include "zf_trial_launch.h"

static int zf_ra_op(const zf_trial_sel& v, int len, int grid, hipStream_t st, const zf_step_args& a) {
    if (v.res) return len == ZF_MAX_SUB ? zf_ra_op_res(v, len, grid, st, a) : zf_ra_op_res_mid(v, len, grid, st, a);
    if (len == ZF_MAX_SUB) {
#define FULL(N, B, T) return zf_ra_kernel_op<N, B, T, false, 0>(grid, st, a)
        ZF_SEL_NBT(v, FULL);
#undef FULL
    }
    if (v.box || !v.nt || len < ZF_MID_MIN || len > ZF_MID_MAX) return -1;
    return len <= 12 ? zf_ra_op_mid_a(v, len, grid, st, a) : zf_ra_op_mid_b(v, len, grid, st, a);
}

bool zf_launch_s16_runahead(const zf_trial_sel& v, int len, int grid, hipStream_t st, const zf_step_args& a) {
    return grid > 0 && zf_ra_op(v, len, grid, st, a) == 1;
}

int zf_runahead_capacity(const zf_trial_sel& v, int len) {
    const int c = zf_ra_op(v, len, 0, nullptr, zf_step_args{});
    return c > 0 ? c : 0;
}
