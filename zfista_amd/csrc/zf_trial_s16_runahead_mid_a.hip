// chains of 16: run-ahead mid chains of 9 .. 12 trials (no box, nontemporal policy)
#include "zf_trial_launch.h"

int zf_ra_op_mid_a(const zf_trial_sel& v, int len, int grid, hipStream_t st, const zf_step_args& a) {
    switch (len) {
    case 9:
        return v.nest ? zf_ra_kernel_op<true, false, true, false, 9>(grid, st, a) : zf_ra_kernel_op<false, false, true, false, 9>(grid, st, a);
    case 10:
        return v.nest ? zf_ra_kernel_op<true, false, true, false, 10>(grid, st, a) : zf_ra_kernel_op<false, false, true, false, 10>(grid, st, a);
    case 11:
        return v.nest ? zf_ra_kernel_op<true, false, true, false, 11>(grid, st, a) : zf_ra_kernel_op<false, false, true, false, 11>(grid, st, a);
    case 12:
        return v.nest ? zf_ra_kernel_op<true, false, true, false, 12>(grid, st, a) : zf_ra_kernel_op<false, false, true, false, 12>(grid, st, a);
    }
    return -1;
}
