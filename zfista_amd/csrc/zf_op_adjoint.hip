// zf_op_adjoint.hip - instantiations of zf_op_adjoint_kernel (W B r)
#include "zf_kernels_op.h"

template <int K, int TY, bool SEP>
static int adjoint_wgs(int tiles) {   // (zf_op_apply.hip: the tiles, or what the device holds of this kernel at once)
    static int cache = -1;
    if (!zf_op_geo<K, TY>::WALK || !zf_op_persist()) return tiles;
    const int r = zf_op_resident(reinterpret_cast<const void*>(zf_op_adjoint_kernel<K, TY, SEP>), &cache);
    return (r > 0 && tiles > r) ? r : tiles;
}

template <int K>
static void launch_adjoint_k(const zf_op_plan& pl, hipStream_t st, const zf_op_args& P0, const double* r, double* grad, double two_scale,
                             const zf_op_fuse& F) {
    zf_op_args P = P0;
    P.tiles = pl.grid;
#define GO(TY, SEP) hipLaunchKernelGGL((zf_op_adjoint_kernel<K, TY, SEP>), dim3(adjoint_wgs<K, TY, SEP>(pl.grid)), dim3(ZF_BLOCK), 0, st, P, r, grad, two_scale, F)
    if (pl.ty == 32 && pl.sep) GO(32, true);
    else if (pl.ty == 32) GO(32, false);
    else if (pl.ty == 16 && pl.sep) GO(16, true);
    else if (pl.ty == 16) GO(16, false);
    else if (pl.sep) GO(8, true);
    else GO(8, false);
#undef GO
}

void zf_launch_op_adjoint(const zf_op_plan& pl, hipStream_t st, const zf_op_args& P, const double* r, double* grad, double two_scale,
                          const zf_op_fuse& F) {
    switch (pl.K) {
        case 3: return launch_adjoint_k<3>(pl, st, P, r, grad, two_scale, F);
        case 5: return launch_adjoint_k<5>(pl, st, P, r, grad, two_scale, F);
        case 7: return launch_adjoint_k<7>(pl, st, P, r, grad, two_scale, F);
        case 9: return launch_adjoint_k<9>(pl, st, P, r, grad, two_scale, F);
        case 11: return launch_adjoint_k<11>(pl, st, P, r, grad, two_scale, F);
        case 13: return launch_adjoint_k<13>(pl, st, P, r, grad, two_scale, F);
        default: return launch_adjoint_k<15>(pl, st, P, r, grad, two_scale, F);
    }
}
