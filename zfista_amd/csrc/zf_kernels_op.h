// zf_kernels_op.h - operator-form least squares: A = B W^-1 with B a K x K correlation with symmetric
// (edge-including mirror) boundary and W one orthonormal Haar level - the image-deblurring LASSO of the
// reference's examples/cameraman.ipynb (cells 8-11):
//     f(x)     = scale |B W^-1 x - b|^2                 (notebook: scale = 1)
//     jac_f(x) = 2 scale W B (B W^-1 x - b)             (the notebook applies B itself as the adjoint)
//     g(x)     = lam |x|_1,  prox = soft threshold
// x holds the Haar coefficients [cA, cH, cV, cD] (pywt.dwt2 layout, each (H/2) x (W/2), row-major), b the
// observed H x W image.  The operator replaces the two GEMV sweeps of the dense least-squares path
// (zf_kernels_gemv.h); everything else of a trial - the residual at y by linearity from the cached A x_k,
// A x_{k-1}, the fused prox step with the gradient vector in HBM, the decide pass - is that path's.
//
// Both kernels work on 32 x 8 output tiles (one per workgroup: 256 workgroups at 256 x 256) staged in LDS
// with a halo of K / 2; the inverse Haar level is folded into the tile load of zf_op_apply_kernel and the
// forward level into the epilogue of zf_op_adjoint_kernel, so the image never exists in memory.
// Arithmetic: the Haar sums in NumPy's left-to-right order ((a + b) + c) + d, then / 2; the correlation
// accumulates its K^2 products row by row with plain multiply-adds (-ffp-contract=off: no FMA) - SciPy's
// own summation order is not specified, parity is to the solver's 1e-10, not bit for bit.
#pragma once
#include "zf_common.h"
#include "zf_decide.h"

constexpr int ZF_OP_TX = 32, ZF_OP_TY = 8;      // output tile of a workgroup
constexpr int ZF_OP_MAXK = 15;                  // largest supported kernel size (odd)
constexpr int ZF_OP_HALO = ZF_OP_MAXK / 2;
constexpr int ZF_OP_LW = ZF_OP_TX + 2 * ZF_OP_HALO, ZF_OP_LH = ZF_OP_TY + 2 * ZF_OP_HALO;

struct zf_op_args {
    const zf_control* ctl;   // NULL: no early exit, slot ignored (evaluation outside the solver loop)
    int H, W, K;             // image size (even), kernel size (odd, <= ZF_OP_MAXK)
    const double* taps;      // K x K, row-major
};

// scipy.signal.correlate2d(..., boundary="symm"): the image mirrored about its edges, edge sample included
__device__ __forceinline__ int zf_op_reflect(int i, int n) {
    if (i < 0) i = -i - 1;
    if (i >= n) i = 2 * n - 1 - i;
    return i;
}

// pixel (iy, ix) of W^-1 x
__device__ __forceinline__ double zf_op_idwt_pixel(const double* __restrict__ x, int iy, int ix, int h, int w) {
    const int64_t q = (int64_t)h * w, at = (int64_t)(iy >> 1) * w + (ix >> 1);
    const double cA = x[at], cH = x[q + at], cV = x[2 * q + at], cD = x[3 * q + at];
    double v;
    if ((iy & 1) == 0) v = (ix & 1) == 0 ? ((cA + cH) + cV) + cD : ((cA + cH) - cV) - cD;
    else v = (ix & 1) == 0 ? ((cA - cH) + cV) - cD : ((cA - cH) - cV) + cD;
    return v / 2;
}

// K x K correlation of the LDS tile at output (ty, tx): rows top to bottom, taps left to right
__device__ __forceinline__ double zf_op_correlate(const double* tile, const double* taps_lds, int K, int ty, int tx) {
    const int off = ZF_OP_HALO - K / 2;
    double acc = 0.0;
    for (int u = 0; u < K; ++u) {
        const double* row = tile + (ty + off + u) * ZF_OP_LW + tx + off;
        for (int v = 0; v < K; ++v) acc = acc + row[v] * taps_lds[u * K + v];
    }
    return acc;
}

// s = B W^-1 x.   SRC_RING: x = ring buffer (cur + slot) % 3 and s likewise (the trial's x+: slot 1); else x / s as given.
__global__ __launch_bounds__(ZF_BLOCK) void zf_op_apply_kernel(zf_op_args P, const double* __restrict__ x0,
                                                               const double* __restrict__ x1,
                                                               const double* __restrict__ x2, double* s0, double* s1,
                                                               double* s2, int slot) {
    __shared__ double tile[ZF_OP_LH * ZF_OP_LW];
    __shared__ double taps[ZF_OP_MAXK * ZF_OP_MAXK];
    int idx = 0;
    if (P.ctl) {
        if (P.ctl->status != ZF_RUNNING) return;
        idx = (P.ctl->cur + slot) % 3;
    }
    const double* __restrict__ x = idx == 0 ? x0 : idx == 1 ? x1 : x2;
    double* __restrict__ s = idx == 0 ? s0 : idx == 1 ? s1 : s2;
    const int K = P.K, half = K / 2, off = ZF_OP_HALO - half;
    const int tiles_x = (P.W + ZF_OP_TX - 1) / ZF_OP_TX;
    const int oy0 = ((int)blockIdx.x / tiles_x) * ZF_OP_TY, ox0 = ((int)blockIdx.x % tiles_x) * ZF_OP_TX;
    for (int k = threadIdx.x; k < K * K; k += ZF_BLOCK) taps[k] = P.taps[k];
    const int lw = ZF_OP_TX + 2 * half, lh = ZF_OP_TY + 2 * half;
    for (int k = threadIdx.x; k < lw * lh; k += ZF_BLOCK) {
        const int ly = k / lw, lx = k % lw;
        const int iy = zf_op_reflect(oy0 + ly - half, P.H), ix = zf_op_reflect(ox0 + lx - half, P.W);
        tile[(ly + off) * ZF_OP_LW + lx + off] = zf_op_idwt_pixel(x, iy, ix, P.H / 2, P.W / 2);
    }
    __syncthreads();
    const int ty = threadIdx.x / ZF_OP_TX, tx = threadIdx.x % ZF_OP_TX;
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy < P.H && ox < P.W) s[(int64_t)oy * P.W + ox] = zf_op_correlate(tile, taps, K, ty, tx);
}

// grad = 2 scale W (B r):  r an H x W image (the residual at y); skipped unless ctl->need_grad.
__global__ __launch_bounds__(ZF_BLOCK) void zf_op_adjoint_kernel(zf_op_args P, const double* __restrict__ r,
                                                                 double* __restrict__ grad, double two_scale) {
    __shared__ double tile[ZF_OP_LH * ZF_OP_LW];
    __shared__ double taps[ZF_OP_MAXK * ZF_OP_MAXK];
    __shared__ double blurred[ZF_OP_TY * ZF_OP_TX];
    if (P.ctl && (P.ctl->status != ZF_RUNNING || !P.ctl->need_grad)) return;
    const int K = P.K, half = K / 2, off = ZF_OP_HALO - half;
    const int tiles_x = (P.W + ZF_OP_TX - 1) / ZF_OP_TX;
    const int oy0 = ((int)blockIdx.x / tiles_x) * ZF_OP_TY, ox0 = ((int)blockIdx.x % tiles_x) * ZF_OP_TX;
    for (int k = threadIdx.x; k < K * K; k += ZF_BLOCK) taps[k] = P.taps[k];
    const int lw = ZF_OP_TX + 2 * half, lh = ZF_OP_TY + 2 * half;
    for (int k = threadIdx.x; k < lw * lh; k += ZF_BLOCK) {
        const int ly = k / lw, lx = k % lw;
        const int iy = zf_op_reflect(oy0 + ly - half, P.H), ix = zf_op_reflect(ox0 + lx - half, P.W);
        tile[(ly + off) * ZF_OP_LW + lx + off] = r[(int64_t)iy * P.W + ix];
    }
    __syncthreads();
    const int ty = threadIdx.x / ZF_OP_TX, tx = threadIdx.x % ZF_OP_TX;
    blurred[ty * ZF_OP_TX + tx] = (oy0 + ty < P.H && ox0 + tx < P.W) ? zf_op_correlate(tile, taps, K, ty, tx) : 0.0;
    __syncthreads();
    // one Haar level of the tile: thread t < 64 owns the 2 x 2 block (t / 16, t % 16)
    if (threadIdx.x < (ZF_OP_TY / 2) * (ZF_OP_TX / 2)) {
        const int by = threadIdx.x / (ZF_OP_TX / 2), bx = threadIdx.x % (ZF_OP_TX / 2);
        const int py = oy0 + 2 * by, px = ox0 + 2 * bx;
        if (py < P.H && px < P.W) {
            const double a = blurred[(2 * by) * ZF_OP_TX + 2 * bx], b = blurred[(2 * by) * ZF_OP_TX + 2 * bx + 1];
            const double c = blurred[(2 * by + 1) * ZF_OP_TX + 2 * bx], d = blurred[(2 * by + 1) * ZF_OP_TX + 2 * bx + 1];
            const int h = P.H / 2, w = P.W / 2;
            const int64_t q = (int64_t)h * w, at = (int64_t)(py >> 1) * w + (px >> 1);
            grad[at] = two_scale * ((((a + b) + c) + d) / 2);
            grad[q + at] = two_scale * ((((a + b) - c) - d) / 2);
            grad[2 * q + at] = two_scale * ((((a - b) + c) - d) / 2);
            grad[3 * q + at] = two_scale * ((((a - b) - c) + d) / 2);
        }
    }
}
