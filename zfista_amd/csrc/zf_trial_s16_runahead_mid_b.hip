// chains of 16: run-ahead mid chains of 13 .. 15 trials (no box, nontemporal policy)
#include "zf_trial_launch.h"

int zf_ra_op_mid_b(const zf_trial_sel& v, int len, int grid, hipStream_t st, const zf_step_args& a) {
    switch (len) {
    case 13:
        return v.nest ? zf_ra_kernel_op<true, false, true, false, 13>(grid, st, a) : zf_ra_kernel_op<false, false, true, false, 13>(grid, st, a);
    case 14:
        return v.nest ? zf_ra_kernel_op<true, false, true, false, 14>(grid, st, a) : zf_ra_kernel_op<false, false, true, false, 14>(grid, st, a);
    case 15:
        return v.nest ? zf_ra_kernel_op<true, false, true, false, 15>(grid, st, a) : zf_ra_kernel_op<false, false, true, false, 15>(grid, st, a);
    }
    return -1;
}
