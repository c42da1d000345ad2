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

// The fused forms (inside the solver loop: three launches per trial instead of six).
// zf_op_adjoint_kernel<true>: the residual at y is formed in the tile load - r = (s_k + beta (s_k - s_{k-1})) - b by
// linearity from the cached B W^-1 x_k, B W^-1 x_{k-1} - and the workgroup's share of |r|^2 (its own tile, no halo)
// goes to part_y.  zf_op_apply_kernel<true>: the workgroup's share of |s+ - b|^2 goes to part_x, and the LAST workgroup
// to arrive adds both in workgroup order, adds the partials of the prox step, builds the pack and runs the decide
// pass (model value, acceptance, lr decay, termination, buffer hand-over, trace row:
// proximal_gradient.py:149-155,:298-307,:510,:525,:539) - the tail of zf_ls_small_rows_kernel for this operator.
struct zf_op_fuse {
    const double* b;          // observed image
    const double* sk[3];      // ring of B W^-1 x (zf_solver::sring)
    double scale, lam;
    int nesterov;
    double* part_y;           // [workgroups] shares of |r(y)|^2   (written when need_grad)
    double* part_x;           // [workgroups] shares of |s+ - b|^2
    unsigned* cnt;            // arrival counter of zf_op_apply_kernel<true> (zero between launches)
    const double* blk_part;   // the prox step's partials, quantity-major [6][grid_step]
    int grid_step;
    double* ls_scal;          // [0] f(y) [1] f(x+)
    double* pack;
    zf_control* ctl_rw;
    double* trace;
    const double* beta_ring;
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

// s = B W^-1 x.   Inside the loop (ctl given): x = ring buffer (cur + slot) % 3 and s likewise (the trial's x+: slot 1);
// else x / s as given.  FUSED: see zf_op_fuse.
template <bool FUSED>
__global__ __launch_bounds__(ZF_BLOCK) void zf_op_apply_kernel(zf_op_args P, const double* __restrict__ x0,
                                                               const double* __restrict__ x1,
                                                               const double* __restrict__ x2, double* s0, double* s1,
                                                               double* s2, int slot, zf_op_fuse F) {
    __shared__ double tile[ZF_OP_LH * ZF_OP_LW];
    __shared__ double taps[ZF_OP_MAXK * ZF_OP_MAXK];
    __shared__ double s_w[ZF_WAVES];
    __shared__ double s_pack[ZF_PACK_LEN];
    __shared__ zf_trial_eval s_pre[ZF_MAX_SUB_ITERS];
    __shared__ int s_last;
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
    double sq = 0.0;
    if (oy < P.H && ox < P.W) {
        const double v = zf_op_correlate(tile, taps, K, ty, tx);
        s[(int64_t)oy * P.W + ox] = v;
        if constexpr (FUSED) {
            const double rv = v - F.b[(int64_t)oy * P.W + ox];
            sq = rv * rv;
        }
    }
    if constexpr (FUSED) {
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        sq = zf_wave_sum(sq);
        if (lane == 0) s_w[wave] = sq;
        __syncthreads();
        if (tid == 0) {
            double t = s_w[0];
            for (int w = 1; w < ZF_WAVES; ++w) t += s_w[w];
            zf_publish(F.part_x + blockIdx.x, t);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned tk = __hip_atomic_fetch_add(F.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (tk == (unsigned)(gridDim.x - 1));
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(F.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
            }
            s_last = last;
        }
        __syncthreads();
        if (!s_last || wave != 0) return;
        // last arriver, wave 0: f(y), f(x+) from the workgroup shares (chunks of 64 workgroups in order, within a chunk
        // the fixed shuffle tree), the prox step's partials, the pack, the decide pass
        double fy = 0.0, fx = 0.0;
        for (int g0 = 0; g0 < (int)gridDim.x; g0 += 64) {
            const int g = g0 + lane;
            const bool in = g < (int)gridDim.x;
            fx += zf_wave_sum(in ? zf_consume(F.part_x + g) : 0.0);
            fy += zf_wave_sum(in ? F.part_y[g] : 0.0);
        }
        const double nx = sqrt(fx), ny = sqrt(fy);
        const double f_x = F.scale * (nx * nx), f_y = F.scale * (ny * ny);     // np.linalg.norm(.) ** 2
        double dot = 0.0, ss = 0.0, l1 = 0.0, mx = 0.0;
        const int64_t G = F.grid_step;
        for (int64_t g0 = 0; g0 < G; g0 += 64) {
            const int64_t g = g0 + lane;
            const bool in = g < G;
            dot += zf_wave_sum(in ? F.blk_part[1 * G + g] : 0.0);
            ss += zf_wave_sum(in ? F.blk_part[2 * G + g] : 0.0);
            l1 += zf_wave_sum(in ? F.blk_part[3 * G + g] : 0.0);
            mx = fmax(mx, zf_wave_max(in ? F.blk_part[5 * G + g] : 0.0));
        }
        double pk[ZF_PACK_LEN];
        // (a rejected trial leaves y as it is: the shares of |r(y)|^2 - and f(y) - are those of the trial before)
        pk[ZF_PK_FY] = __shfl(f_y, 0, 64);
        pk[ZF_PK_DOT] = __shfl(dot, 0, 64);
        pk[ZF_PK_SS] = __shfl(ss, 0, 64);
        pk[ZF_PK_GX] = F.lam * __shfl(l1, 0, 64);
        pk[ZF_PK_FX] = __shfl(f_x, 0, 64);
        pk[ZF_PK_ERR] = __shfl(mx, 0, 64);
        pk[6] = 0.0;
        pk[7] = 0.0;
        if (lane == 0) {
            F.ls_scal[0] = pk[ZF_PK_FY];
            F.ls_scal[1] = pk[ZF_PK_FX];
#pragma unroll
            for (int k = 0; k < ZF_PACK_LEN; ++k) {
                F.pack[k] = pk[k];
                s_pack[k] = pk[k];
            }
        }
        zf_decide_pass_wave(F.ctl_rw, s_pack, pk, F.trace, F.beta_ring, lane, 64, s_pre);
    }
}

// grad = 2 scale W (B r):  r an H x W image (the residual at y); skipped unless ctl->need_grad.  FUSED: r is formed in
// the tile load from the ring of B W^-1 x (zf_op_fuse), `r` is not read.
template <bool FUSED>
__global__ __launch_bounds__(ZF_BLOCK) void zf_op_adjoint_kernel(zf_op_args P, const double* __restrict__ r,
                                                                 double* __restrict__ grad, double two_scale, zf_op_fuse F) {
    __shared__ double tile[ZF_OP_LH * ZF_OP_LW];
    __shared__ double taps[ZF_OP_MAXK * ZF_OP_MAXK];
    __shared__ double blurred[ZF_OP_TY * ZF_OP_TX];
    __shared__ double s_w[ZF_WAVES];
    if (P.ctl && (P.ctl->status != ZF_RUNNING || !P.ctl->need_grad)) return;
    const double* __restrict__ sk = nullptr;
    const double* __restrict__ so = nullptr;
    double beta = 0.0;
    if constexpr (FUSED) {
        const int cur = P.ctl->cur;
        sk = F.sk[cur];
        so = F.sk[(cur + 2) % 3];
        beta = F.nesterov ? P.ctl->beta_next : 0.0;
    }
    const int K = P.K, half = K / 2, off = ZF_OP_HALO - half;
    const int tiles_x = (P.W + ZF_OP_TX - 1) / ZF_OP_TX;
    const int oy0 = ((int)blockIdx.x / tiles_x) * ZF_OP_TY, ox0 = ((int)blockIdx.x % tiles_x) * ZF_OP_TX;
    for (int k = threadIdx.x; k < K * K; k += ZF_BLOCK) taps[k] = P.taps[k];
    const int lw = ZF_OP_TX + 2 * half, lh = ZF_OP_TY + 2 * half;
    for (int k = threadIdx.x; k < lw * lh; k += ZF_BLOCK) {
        const int ly = k / lw, lx = k % lw;
        const int iy = zf_op_reflect(oy0 + ly - half, P.H), ix = zf_op_reflect(ox0 + lx - half, P.W);
        const int64_t at = (int64_t)iy * P.W + ix;
        double rv;
        if constexpr (FUSED) {
            double ay = sk[at];
            if (F.nesterov) ay = ay + beta * (ay - so[at]);   // B W^-1 y by linearity (zf_resid_y_kernel's expression)
            rv = ay - F.b[at];
        } else {
            rv = r[at];
        }
        tile[(ly + off) * ZF_OP_LW + lx + off] = rv;
    }
    __syncthreads();
    const int ty = threadIdx.x / ZF_OP_TX, tx = threadIdx.x % ZF_OP_TX;
    blurred[ty * ZF_OP_TX + tx] = (oy0 + ty < P.H && ox0 + tx < P.W) ? zf_op_correlate(tile, taps, K, ty, tx) : 0.0;
    if constexpr (FUSED) {   // this workgroup's share of |r|^2: its own pixels (the halo belongs to the neighbours)
        double sq = 0.0;
        if (oy0 + ty < P.H && ox0 + tx < P.W) {
            const double rv = tile[(ty + ZF_OP_HALO) * ZF_OP_LW + tx + ZF_OP_HALO];
            sq = rv * rv;
        }
        sq = zf_wave_sum(sq);
        if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = sq;
    }
    __syncthreads();
    if constexpr (FUSED) {
        if (threadIdx.x == 0) {
            double t = s_w[0];
            for (int w = 1; w < ZF_WAVES; ++w) t += s_w[w];
            F.part_y[blockIdx.x] = t;
        }
    }
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
