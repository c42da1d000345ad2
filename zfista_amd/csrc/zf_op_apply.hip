// zf_op_apply.hip - instantiations of zf_op_apply_kernel (B W^-1 x) and the plan of an operator problem
#include <math.h>
#include <stdlib.h>

#include "zf_kernels_op.h"

zf_op_plan zf_op_make_plan(int64_t h, int64_t w, int k, bool separable) {
    zf_op_plan pl;
    pl.K = k < 3 ? 3 : k;
    const int64_t tx = (w + ZF_OP_TX - 1) / ZF_OP_TX;
    const int64_t tall = tx * ((h + 31) / 32);
    pl.ty = tall >= 256 ? 32 : 8;   // (measured, round 5: 1024 x 1024 = 512 tall tiles runs 20.6 k it/s on 64 x 32 tiles, 15.5 k on 64 x 8; 256 x 256 the other way round)
    if (const char* e = getenv("ZF_OP_TY")) {   // (experiments: 8 / 16 / 32 rows per tile)
        const int v = atoi(e);
        if (v == 8 || v == 16 || v == 32) pl.ty = v;
    }
    pl.sep = separable;
    pl.grid = (int)(tx * ((h + pl.ty - 1) / pl.ty));
    return pl;
}

// k = u v^T ?  Pivot on the largest entry: u = its column, v = its row / the pivot.
bool zf_op_factor_rank1(const double* taps, int k, double* u, double* v) {
    int pi = 0, pj = 0;
    double big = 0.0;
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j)
            if (fabs(taps[i * k + j]) > big) {
                big = fabs(taps[i * k + j]);
                pi = i;
                pj = j;
            }
    if (!(big > 0.0) || !isfinite(big)) return false;
    for (int i = 0; i < k; ++i) u[i] = taps[i * k + pj];
    for (int j = 0; j < k; ++j) v[j] = taps[pi * k + j] / taps[pi * k + pj];
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j)
            if (!(fabs(taps[i * k + j] - u[i] * v[j]) <= 1e-14 * big)) return false;
    return true;
}

// workgroups of a launch: the tiles of the image, or - more tiles than the device holds workgroups of this kernel at once - that
// many (a multiple of 8: a workgroup's tiles stay on its XCD); the kernels walk their tiles (ZF_OP_PERSIST=0: a workgroup per tile)
bool zf_op_persist() {
    static const bool on = [] {
        const char* e = getenv("ZF_OP_PERSIST");
        return e ? atoi(e) != 0 : true;
    }();
    return on;
}
int zf_op_resident(const void* kernel, int* cache) {
    if (*cache >= 0) return *cache;
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, ZF_BLOCK, 0) != hipSuccess) return 0;
    *cache = (per_cu * prop.multiProcessorCount) & ~7;
    return *cache;
}
template <int K, int TY, bool SEP>
static int apply_wgs(int tiles) {
    static int cache = -1;
    if (!zf_op_geo<K, TY>::WALK || !zf_op_persist()) return tiles;
    const int r = zf_op_resident(reinterpret_cast<const void*>(zf_op_apply_kernel<K, TY, SEP>), &cache);
    return (r > 0 && tiles > r) ? r : tiles;
}

template <int K>
static void launch_apply_k(const zf_op_plan& pl, hipStream_t st, const zf_op_args& P0, const double* x0, const double* x1, const double* x2,
                           double* s0, double* s1, double* s2, int slot, const zf_op_fuse& F) {
    zf_op_args P = P0;
    P.tiles = pl.grid;
#define GO(TY, SEP) hipLaunchKernelGGL((zf_op_apply_kernel<K, TY, SEP>), dim3(apply_wgs<K, TY, SEP>(pl.grid)), dim3(ZF_BLOCK), 0, st, P, x0, x1, x2, s0, s1, s2, slot, F)
    if (pl.ty == 32 && pl.sep) GO(32, true);
    else if (pl.ty == 32) GO(32, false);
    else if (pl.ty == 16 && pl.sep) GO(16, true);
    else if (pl.ty == 16) GO(16, false);
    else if (pl.sep) GO(8, true);
    else GO(8, false);
#undef GO
}

void zf_launch_op_apply(const zf_op_plan& pl, hipStream_t st, const zf_op_args& P, const double* x0, const double* x1, const double* x2,
                        double* s0, double* s1, double* s2, int slot, const zf_op_fuse& F) {
    switch (pl.K) {
        case 3: return launch_apply_k<3>(pl, st, P, x0, x1, x2, s0, s1, s2, slot, F);
        case 5: return launch_apply_k<5>(pl, st, P, x0, x1, x2, s0, s1, s2, slot, F);
        case 7: return launch_apply_k<7>(pl, st, P, x0, x1, x2, s0, s1, s2, slot, F);
        case 9: return launch_apply_k<9>(pl, st, P, x0, x1, x2, s0, s1, s2, slot, F);
        case 11: return launch_apply_k<11>(pl, st, P, x0, x1, x2, s0, s1, s2, slot, F);
        case 13: return launch_apply_k<13>(pl, st, P, x0, x1, x2, s0, s1, s2, slot, F);
        default: return launch_apply_k<15>(pl, st, P, x0, x1, x2, s0, s1, s2, slot, F);
    }
}
