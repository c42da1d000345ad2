// chains of 16: run-ahead mid chains of 9 .. 15 trials of a ZF_ACCEPT_RESOLVED solver (no box, nontemporal policy)
#include "zf_trial_launch.h"

int zf_ra_op_res_mid(const zf_trial_sel& v, int len, int grid, hipStream_t st, const zf_step_args& a) {
    switch (len) {
    case 9:
        return v.nest ? zf_ra_kernel_op<true, false, true, true, 9>(grid, st, a) : zf_ra_kernel_op<false, false, true, true, 9>(grid, st, a);
    case 10:
        return v.nest ? zf_ra_kernel_op<true, false, true, true, 10>(grid, st, a) : zf_ra_kernel_op<false, false, true, true, 10>(grid, st, a);
    case 11:
        return v.nest ? zf_ra_kernel_op<true, false, true, true, 11>(grid, st, a) : zf_ra_kernel_op<false, false, true, true, 11>(grid, st, a);
    case 12:
        return v.nest ? zf_ra_kernel_op<true, false, true, true, 12>(grid, st, a) : zf_ra_kernel_op<false, false, true, true, 12>(grid, st, a);
    case 13:
        return v.nest ? zf_ra_kernel_op<true, false, true, true, 13>(grid, st, a) : zf_ra_kernel_op<false, false, true, true, 13>(grid, st, a);
    case 14:
        return v.nest ? zf_ra_kernel_op<true, false, true, true, 14>(grid, st, a) : zf_ra_kernel_op<false, false, true, true, 14>(grid, st, a);
    case 15:
        return v.nest ? zf_ra_kernel_op<true, false, true, true, 15>(grid, st, a) : zf_ra_kernel_op<false, false, true, true, 15>(grid, st, a);
    }
    return -1;
}
