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
#include "zf_decide.h"

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

// ---------------------------------------------------------------------------
// launch geometry of the trial kernel (measured on MI355X, tools/tune_trial.hip,
// n = 1e8: 0.66 ms = 6.05 TB/s for the streaming part; a capped grid-stride loop
// ran 0.87 ms, persistent tile walks 0.67-0.71 ms depending on the device):
//   * one STREAMING workgroup per tile of ZF_TILE_UNITS = 4 x 256 consecutive
//     16-byte units (16 KiB of every stream); the dispatcher hands out tiles in
//     order, so the resident workgroups sweep each array as one contiguous
//     window (DRAM locality);
//   * all 16 loads of a thread are issued before the first use;
//   * d, c are read once per trial and x+ is written once: nontemporal (nt)
//     loads / stores keep them from displacing x_k in the caches (+8 %).
// Reduction: every workgroup stores its six partials with plain stores and retires
// (no per-workgroup hand-off); a second, tiny launch - zf_finalize_kernel, ZF_FIN_WGS
// workgroups of 1024 threads, every load independent - adds them in index order, its
// last-arriving workgroup (one ticket per finalize workgroup, guide Guideline 16 R1
// counter form) builds the scalar pack and runs the decide step.  Measured per step in
// loops of 20 launches (tools/tune_trial.hip, one box): streaming + finalize 0.675 ms
// (n = 1e8) / 0.069 ms (1e7); a fused single launch with per-workgroup tickets and a
// two-level in-launch reduction 0.686 / 0.078 ms at one tile per workgroup, and
// device-dependent (0.66 - 0.71) with several tiles per workgroup; the first version's
// single-workgroup finalize with dependent load rounds 0.849 / 0.098 ms.
// ---------------------------------------------------------------------------
constexpr int ZF_TILE_U = 4;
constexpr int ZF_TILE_UNITS = ZF_TILE_U * ZF_BLOCK;   // 16-byte units per tile
constexpr int ZF_MAX_TILES_PER_WG = 8;                // upper bound of zf_step_args.tiles_per_wg
constexpr int ZF_FIN_WGS = 48;                        // workgroups of the finalize kernel
constexpr int ZF_FIN_THREADS = 1024;

typedef double zf_d2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ zf_d2 zf_ld2(const zf_d2* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
template <bool NT> __device__ __forceinline__ void zf_st2(zf_d2* p, zf_d2 v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
__device__ __forceinline__ void zf_publish(double* p, double v) {   // 8-byte write-through store
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double zf_consume(const double* p) {     // sc1 load, bypasses this CU's L1
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct zf_step_args {
    const zf_control* ctl;
    double* xb[3];            // x ring
    const double* p0;         // diag: d        vec: grad
    const double* p1;         // diag: c        vec: unused
    double lam, lo, hi;
    int64_t n;
    int tiles_per_wg;         // interleaved tiles per workgroup (1 .. ZF_MAX_TILES_PER_WG)
    double* blk_part;         // ZF_NPART x gridDim.x per-workgroup partials, quantity-major
};

struct zf_finalize_args {
    const double* blk_part;   // ZF_NPART x nblocks (written by the trial kernel)
    int nblocks;
    double* slice_part;       // ZF_NPART x ZF_FIN_WGS
    unsigned* cnt;            // arrival counter of the finalize workgroups (zero between launches)
    double scale[ZF_NPART];   // pack[k] = scale[k] * total[k]
    const double* f_y_ext;    // least squares: f(y), f(x+) come from the GEMV side (else NULL)
    const double* f_x_ext;
    int contribute_f;         // sharded least squares: only rank 0 contributes the replicated f values
    double* pack;             // local pack out (ZF_PACK_LEN)
    zf_control* ctl;          // read for the early exit; written when `decide`
    int decide;               // unsharded x: run the decide step here
    double* trace;
    const double* beta_ring;
};

// Second launch of a step.  Every workgroup adds its slice of the per-workgroup partials
// (thread t takes workgroups t, t+1024, ... of the slice: index order; six independent loads per
// index), publishes the slice totals write-through and takes a ticket; the last arriver adds
// the ZF_FIN_WGS slices in slice order, builds the pack and (decide) runs the decide step:
// model value, acceptance, lr decay, failure, termination, buffer rotation, trace row
// (proximal_gradient.py:149-155,:298-307,:510,:525,:539).  Deterministic: no float atomics,
// sums in index order.
__global__ __launch_bounds__(ZF_FIN_THREADS) void zf_finalize_kernel(zf_finalize_args F) {
    __shared__ double lds[(ZF_FIN_THREADS / 64) * 8];
    __shared__ int s_last;
    if (F.ctl->status != ZF_RUNNING) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = ZF_FIN_THREADS / 64;
    const int per = (F.nblocks + (int)gridDim.x - 1) / (int)gridDim.x;
    const int b0 = blockIdx.x * per;
    int b1 = b0 + per;
    if (b1 > F.nblocks) b1 = F.nblocks;
    double v[ZF_NPART];
#pragma unroll
    for (int k = 0; k < ZF_NPART; ++k) v[k] = 0.0;
    for (int b = b0 + threadIdx.x; b < b1; b += ZF_FIN_THREADS) {
        double p[ZF_NPART];
#pragma unroll
        for (int k = 0; k < ZF_NPART; ++k) p[k] = F.blk_part[(int64_t)k * F.nblocks + b];
#pragma unroll
        for (int k = 0; k < ZF_NPART - 1; ++k) v[k] += p[k];
        v[ZF_NPART - 1] = fmax(v[ZF_NPART - 1], p[ZF_NPART - 1]);
    }
#pragma unroll
    for (int k = 0; k < ZF_NPART; ++k) {
        const double r = (k == ZF_NPART - 1) ? zf_wave_max(v[k]) : zf_wave_sum(v[k]);
        if (lane == 0) lds[wave * 8 + k] = r;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        if (threadIdx.x < ZF_NPART) {
            const int k = threadIdx.x;
            double r = lds[k];
            for (int w = 1; w < NW; ++w) r = (k == ZF_NPART - 1) ? fmax(r, lds[w * 8 + k]) : r + lds[w * 8 + k];
            zf_publish(F.slice_part + (int64_t)k * ZF_FIN_WGS + blockIdx.x, r);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(F.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (t == (unsigned)(gridDim.x - 1));
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(F.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_last = last;
        }
    }
    __syncthreads();
    if (!s_last || wave != 0) return;
    // last arriver, wave 0: lane q holds slice q (six independent sc1 loads: one round trip),
    // then the fixed shuffle tree adds the slices in slice order
    double tot[ZF_NPART];
#pragma unroll
    for (int k = 0; k < ZF_NPART; ++k)
        tot[k] = (lane < (int)gridDim.x) ? zf_consume(F.slice_part + (int64_t)k * ZF_FIN_WGS + lane) : 0.0;
#pragma unroll
    for (int k = 0; k < ZF_NPART; ++k) tot[k] = (k == ZF_NPART - 1) ? zf_wave_max(tot[k]) : zf_wave_sum(tot[k]);
    if (lane != 0) return;
    double* pack = F.pack;
    pack[ZF_PK_FY] = F.f_y_ext ? (F.contribute_f ? *F.f_y_ext : 0.0) : F.scale[0] * tot[0];
    pack[ZF_PK_DOT] = tot[1];
    pack[ZF_PK_SS] = tot[2];
    pack[ZF_PK_GX] = F.scale[3] * tot[3];
    pack[ZF_PK_FX] = F.f_x_ext ? (F.contribute_f ? *F.f_x_ext : 0.0) : F.scale[4] * tot[4];
    pack[ZF_PK_ERR] = tot[5];
    pack[6] = 0.0;
    pack[7] = 0.0;
    if (F.decide) zf_decide_step(F.ctl, pack, F.trace, F.beta_ring);
}

// GRAD_INLINE: true  -> separable quadratic, gradient computed from d, c
//              false -> gradient vector read from HBM (least squares)
// NT: nontemporal policy for the once-touched streams (p0, p1 loads, x+ stores)
template <bool GRAD_INLINE, bool NESTEROV, bool BOX, bool NT>
__global__ __launch_bounds__(ZF_BLOCK) void zf_trial_kernel(zf_step_args A) {
    __shared__ double lds[ZF_WAVES * 8 + 8];
    // wave-uniform control reads (scalar loads); written by the previous launch's decide step
    const int status = A.ctl->status;
    if (status != ZF_RUNNING) return;
    const int cur = A.ctl->cur;
    const double lr = A.ctl->lr;
    const double beta = NESTEROV ? A.ctl->beta_next : 0.0;   // same cache line as status / cur / lr
    const double tau = A.lam * lr;   // oracle: soft_threshold(x, lam * weight)
    const double* __restrict__ xk = A.xb[cur];
    const double* __restrict__ xo = A.xb[(cur + 2) % 3];
    double* __restrict__ xn = A.xb[(cur + 1) % 3];
    const double* __restrict__ p0 = A.p0;
    const double* __restrict__ p1 = A.p1;
    const int64_t n = A.n;

    zf_elem_acc acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const int64_t n2 = n >> 1;  // 16-byte units
    const zf_d2* __restrict__ xk2 = reinterpret_cast<const zf_d2*>(xk);
    const zf_d2* __restrict__ xo2 = reinterpret_cast<const zf_d2*>(xo);
    const zf_d2* __restrict__ p02 = reinterpret_cast<const zf_d2*>(p0);
    const zf_d2* __restrict__ p12 = reinterpret_cast<const zf_d2*>(p1);
    zf_d2* __restrict__ xn2 = reinterpret_cast<zf_d2*>(xn);

    // Workgroup b owns tiles b, b + G, b + 2G, ... (G = gridDim.x, A.tiles_per_wg of them):
    // interleaved, so that at any time the resident workgroups still cover one contiguous window
    // of every stream (consecutive tiles per workgroup measured 4-8 % slower), while the
    // per-workgroup costs (block reduction, partial stores) are paid once per T tiles.  Which T
    // wins is device-dependent (T = 4: -1 % on some MI355X boxes, +7 % on others, against T = 1),
    // so the solver measures it once at initialisation (zf_solver_autotune); T = 1 otherwise.
    const int64_t ntiles = (n2 + ZF_TILE_UNITS - 1) / ZF_TILE_UNITS;
    const int64_t G = gridDim.x;
    zf_d2 r[ZF_TILE_U];
    for (int t = 0; t < A.tiles_per_wg; ++t) {
        const int64_t tile = (int64_t)t * G + blockIdx.x;
        if (tile >= ntiles) break;
        const int64_t base = tile * ZF_TILE_UNITS + threadIdx.x;
        if ((tile + 1) * ZF_TILE_UNITS <= n2) {   // full tile (workgroup-uniform)
            zf_d2 a[ZF_TILE_U], o[ZF_TILE_U], q[ZF_TILE_U], cc[ZF_TILE_U];
#pragma unroll
            for (int u = 0; u < ZF_TILE_U; ++u) {
                const int64_t i = base + u * ZF_BLOCK;
                a[u] = xk2[i];
                o[u] = a[u];
                if (NESTEROV) o[u] = xo2[i];
                q[u] = zf_ld2<NT>(p02 + i);
                cc[u] = q[u];
                if (GRAD_INLINE) cc[u] = zf_ld2<NT>(p12 + i);
            }
            // keep all 16 loads of the tile in flight: without this fence the scheduler sinks the
            // last four below the first arithmetic to save registers
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < ZF_TILE_U; ++u) {
                if (GRAD_INLINE) {
                    r[u].x = zf_elem_diag<NESTEROV, BOX>(a[u].x, o[u].x, q[u].x, cc[u].x, beta, lr, tau, A.lo, A.hi, acc);
                    r[u].y = zf_elem_diag<NESTEROV, BOX>(a[u].y, o[u].y, q[u].y, cc[u].y, beta, lr, tau, A.lo, A.hi, acc);
                } else {
                    r[u].x = zf_elem_vec<NESTEROV, BOX>(a[u].x, o[u].x, q[u].x, beta, lr, tau, A.lo, A.hi, acc);
                    r[u].y = zf_elem_vec<NESTEROV, BOX>(a[u].y, o[u].y, q[u].y, beta, lr, tau, A.lo, A.hi, acc);
                }
            }
#pragma unroll
            for (int u = 0; u < ZF_TILE_U; ++u) zf_st2<NT>(xn2 + base + u * ZF_BLOCK, r[u]);
        } else {
            // ragged last tile of the vector: stored at once (one workgroup per launch)
            for (int u = 0; u < ZF_TILE_U; ++u) {
                const int64_t i = base + u * ZF_BLOCK;
                if (i < n2) {
                    const zf_d2 a0 = xk2[i];
                    const zf_d2 o0 = NESTEROV ? xo2[i] : a0;
                    const zf_d2 q0 = p02[i];
                    const zf_d2 c0 = GRAD_INLINE ? p12[i] : q0;
                    zf_d2 r0;
                    if (GRAD_INLINE) {
                        r0.x = zf_elem_diag<NESTEROV, BOX>(a0.x, o0.x, q0.x, c0.x, beta, lr, tau, A.lo, A.hi, acc);
                        r0.y = zf_elem_diag<NESTEROV, BOX>(a0.y, o0.y, q0.y, c0.y, beta, lr, tau, A.lo, A.hi, acc);
                    } else {
                        r0.x = zf_elem_vec<NESTEROV, BOX>(a0.x, o0.x, q0.x, beta, lr, tau, A.lo, A.hi, acc);
                        r0.y = zf_elem_vec<NESTEROV, BOX>(a0.y, o0.y, q0.y, beta, lr, tau, A.lo, A.hi, acc);
                    }
                    xn2[i] = r0;
                }
            }
        }
    }
    // odd trailing element (n odd): workgroup 0
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
    double mine = 0.0;
    zf_block_reduce<5, 1, ZF_WAVES>(sums, maxs, lds, mine);
    if (threadIdx.x < ZF_NPART) A.blk_part[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = mine;
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
