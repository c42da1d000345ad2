// zf_kernels_step.h - the fused trial kernels of the single-objective path.
//
// One HBM pass per line-search trial.  For the separable problem (P-diag) the
// pass reads x_k, x_{k-1}, d, c and writes x+ : 40 B per element, which is the
// algorithmic minimum; y_k, grad f(y_k) and all six reductions
// live in registers.  Reference sites carried by the element body:
//   y  = x_k + beta (x_k - x_{k-1})          proximal_gradient.py:534
//   v  = y - lr * grad f(y)                  proximal_gradient.py:148
//   x+ = prox_{lr g}(v)                      proximal_gradient.py:148 (callback :13)
//   <grad f(y), x+ - y>, |x+ - y|^2, g(x+)   proximal_gradient.py:150-152
//   f(y), f(x+)                              proximal_gradient.py:140,295
//   max |x+ - y|                             proximal_gradient.py:510
// Element arithmetic is written in NumPy's evaluation order and compiled with
// -ffp-contract=off, so every x+ is bit-identical to the NumPy expression; only
// the summation order of the reductions differs.
#pragma once
#include "zf_common.h"

// number of per-block partial quantities a trial kernel emits
// [0] f(y) raw  [1] dot  [2] ss  [3] |x+|_1  [4] f(x+) raw  [5] max
constexpr int ZF_NPART = 6;

struct zf_elem_acc {
    double fy, dot, ss, l1, fx, mx;
};

// --- element bodies --------------------------------------------------------
template <bool NESTEROV, bool BOX>
__device__ __forceinline__ double zf_elem_diag(double xk, double xo, double d, double c, double beta,
                                               double lr, double tau, double lo, double hi,
                                               zf_elem_acc& a) {
    double y = xk;
    if (NESTEROV) y = xk + beta * (xk - xo);
    const double r = y - c;
    const double grad = d * r;          // jac_f = d * (y - c)
    a.fy += d * (r * r);                // f = 0.5 * sum(d * (r*r))
    const double v = y - lr * grad;
    double xn = zf_soft_threshold(v, tau);
    if (BOX) xn = zf_clip(xn, lo, hi);
    const double dx = xn - y;
    a.dot += grad * dx;
    a.ss += dx * dx;
    a.l1 += fabs(xn);
    const double rn = xn - c;
    a.fx += d * (rn * rn);
    a.mx = fmax(a.mx, fabs(dx));
    return xn;
}

template <bool NESTEROV, bool BOX>
__device__ __forceinline__ double zf_elem_vec(double xk, double xo, double grad, double beta, double lr,
                                              double tau, double lo, double hi, zf_elem_acc& a) {
    double y = xk;
    if (NESTEROV) y = xk + beta * (xk - xo);
    const double v = y - lr * grad;
    double xn = zf_soft_threshold(v, tau);
    if (BOX) xn = zf_clip(xn, lo, hi);
    const double dx = xn - y;
    a.dot += grad * dx;
    a.ss += dx * dx;
    a.l1 += fabs(xn);
    a.mx = fmax(a.mx, fabs(dx));
    return xn;
}

struct zf_step_args {
    const zf_control* ctl;
    const double* beta_ring;  // ZF_RING momentum factors, indexed by accepted count
    double* xb[3];            // x ring
    const double* p0;         // diag: d        vec: grad
    const double* p1;         // diag: c        vec: unused
    double lam, lo, hi;
    int64_t n;
    double* partials;         // ZF_NPART x gridDim.x, quantity-major
};

// GRAD_INLINE: true  -> separable quadratic, gradient computed from d, c
//              false -> gradient vector read from HBM (least squares)
template <bool GRAD_INLINE, bool NESTEROV, bool BOX>
__global__ __launch_bounds__(ZF_BLOCK) void zf_trial_kernel(zf_step_args A) {
    __shared__ double lds[ZF_WAVES * ZF_NPART];
    // wave-uniform control reads (scalar loads); written by the previous decide kernel
    const int status = A.ctl->status;
    if (status != ZF_RUNNING) return;
    const int cur = A.ctl->cur;
    const double lr = A.ctl->lr;
    const double beta = NESTEROV ? A.beta_ring[A.ctl->nit % ZF_RING] : 0.0;
    const double tau = A.lam * lr;   // oracle: soft_threshold(x, lam * weight)
    const double* __restrict__ xk = A.xb[cur];
    const double* __restrict__ xo = A.xb[(cur + 2) % 3];
    double* __restrict__ xn = A.xb[(cur + 1) % 3];
    const double* __restrict__ p0 = A.p0;
    const double* __restrict__ p1 = A.p1;
    const int64_t n = A.n;

    zf_elem_acc acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const int64_t n2 = n >> 1;  // double2 units
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    const double2* __restrict__ xk2 = reinterpret_cast<const double2*>(xk);
    const double2* __restrict__ xo2 = reinterpret_cast<const double2*>(xo);
    const double2* __restrict__ p02 = reinterpret_cast<const double2*>(p0);
    const double2* __restrict__ p12 = reinterpret_cast<const double2*>(p1);
    double2* __restrict__ xn2 = reinterpret_cast<double2*>(xn);

    int64_t i = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x;
    // two 16-byte units per thread per trip: 8 independent 16-B loads in flight
    for (; i + stride < n2; i += 2 * stride) {
        const int64_t j = i + stride;
        double2 a0 = xk2[i], a1 = xk2[j];
        double2 o0 = a0, o1 = a1;
        if (NESTEROV) { o0 = xo2[i]; o1 = xo2[j]; }
        double2 q0 = p02[i], q1 = p02[j];
        double2 c0 = q0, c1 = q1;
        if (GRAD_INLINE) { c0 = p12[i]; c1 = p12[j]; }
        double2 r0, r1;
        if (GRAD_INLINE) {
            r0.x = zf_elem_diag<NESTEROV, BOX>(a0.x, o0.x, q0.x, c0.x, beta, lr, tau, A.lo, A.hi, acc);
            r0.y = zf_elem_diag<NESTEROV, BOX>(a0.y, o0.y, q0.y, c0.y, beta, lr, tau, A.lo, A.hi, acc);
            r1.x = zf_elem_diag<NESTEROV, BOX>(a1.x, o1.x, q1.x, c1.x, beta, lr, tau, A.lo, A.hi, acc);
            r1.y = zf_elem_diag<NESTEROV, BOX>(a1.y, o1.y, q1.y, c1.y, beta, lr, tau, A.lo, A.hi, acc);
        } else {
            r0.x = zf_elem_vec<NESTEROV, BOX>(a0.x, o0.x, q0.x, beta, lr, tau, A.lo, A.hi, acc);
            r0.y = zf_elem_vec<NESTEROV, BOX>(a0.y, o0.y, q0.y, beta, lr, tau, A.lo, A.hi, acc);
            r1.x = zf_elem_vec<NESTEROV, BOX>(a1.x, o1.x, q1.x, beta, lr, tau, A.lo, A.hi, acc);
            r1.y = zf_elem_vec<NESTEROV, BOX>(a1.y, o1.y, q1.y, beta, lr, tau, A.lo, A.hi, acc);
        }
        xn2[i] = r0;
        xn2[j] = r1;
    }
    for (; i < n2; i += stride) {
        double2 a0 = xk2[i];
        double2 o0 = a0;
        if (NESTEROV) o0 = xo2[i];
        double2 q0 = p02[i];
        double2 c0 = q0;
        if (GRAD_INLINE) c0 = p12[i];
        double2 r0;
        if (GRAD_INLINE) {
            r0.x = zf_elem_diag<NESTEROV, BOX>(a0.x, o0.x, q0.x, c0.x, beta, lr, tau, A.lo, A.hi, acc);
            r0.y = zf_elem_diag<NESTEROV, BOX>(a0.y, o0.y, q0.y, c0.y, beta, lr, tau, A.lo, A.hi, acc);
        } else {
            r0.x = zf_elem_vec<NESTEROV, BOX>(a0.x, o0.x, q0.x, beta, lr, tau, A.lo, A.hi, acc);
            r0.y = zf_elem_vec<NESTEROV, BOX>(a0.y, o0.y, q0.y, beta, lr, tau, A.lo, A.hi, acc);
        }
        xn2[i] = r0;
    }
    // odd tail element
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t t = n - 1;
        const double xo_t = NESTEROV ? xo[t] : xk[t];
        if (GRAD_INLINE)
            xn[t] = zf_elem_diag<NESTEROV, BOX>(xk[t], xo_t, p0[t], p1[t], beta, lr, tau, A.lo, A.hi, acc);
        else
            xn[t] = zf_elem_vec<NESTEROV, BOX>(xk[t], xo_t, p0[t], beta, lr, tau, A.lo, A.hi, acc);
    }

    const double sums[5] = {acc.fy, acc.dot, acc.ss, acc.l1, acc.fx};
    const double maxs[1] = {acc.mx};
    double out = 0.0;
    zf_block_reduce<5, 1, ZF_WAVES>(sums, maxs, lds, out);
    if (threadIdx.x < ZF_NPART) A.partials[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = out;
}

// --- f(x), g(x) at a point (initial F(x0), proximal_gradient.py:466,472) -------
// partials: [0] f raw sum  [1] |x|_1  [2] box violations
template <bool GRAD_INLINE, bool BOX>
__global__ __launch_bounds__(ZF_BLOCK) void zf_eval_kernel(const double* __restrict__ x,
                                                           const double* __restrict__ d,
                                                           const double* __restrict__ c, double lo,
                                                           double hi, int64_t n, double* partials) {
    __shared__ double lds[ZF_WAVES * 3];
    double f = 0.0, l1 = 0.0, viol = 0.0;
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; i < n; i += stride) {
        const double xv = x[i];
        if (GRAD_INLINE) {
            const double r = xv - c[i];
            f += d[i] * (r * r);
        }
        l1 += fabs(xv);
        if (BOX) viol += (xv < lo || xv > hi) ? 1.0 : 0.0;
    }
    const double sums[3] = {f, l1, viol};
    const double maxs[1] = {0.0};
    double out = 0.0;
    zf_block_reduce<3, 0, ZF_WAVES>(sums, maxs, lds, out);
    if (threadIdx.x < 3) partials[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = out;
}

// --- fixed-order reduction of per-block partials (one block) ---------------
// out[k] = scale[k] * reduce_k(partials[k][0..nblocks))   k < nq; quantity
// `max_index` (or -1) is a max, the rest are sums.
constexpr int ZF_FIN_BLOCK = 256;
__device__ __forceinline__ void zf_finalize_partials(const double* __restrict__ partials, int nblocks,
                                                     int nq, int max_index, double* lds /* waves*8 */,
                                                     double* totals /* lds, 8 */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int W = ZF_FIN_BLOCK / 64;
    for (int k = 0; k < nq; ++k) {
        const bool is_max = (k == max_index);
        double v = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += ZF_FIN_BLOCK) {
            const double p = partials[(int64_t)k * nblocks + b];
            v = is_max ? fmax(v, p) : v + p;
        }
        v = is_max ? zf_wave_max(v) : zf_wave_sum(v);
        if (lane == 0) lds[wave * 8 + k] = v;
    }
    __syncthreads();
    if (threadIdx.x < nq) {
        const int k = threadIdx.x;
        double v = lds[k];
        for (int w = 1; w < W; ++w) v = (k == max_index) ? fmax(v, lds[w * 8 + k]) : v + lds[w * 8 + k];
        totals[k] = v;
    }
    __syncthreads();
}
