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
// The reduction is finished inside the same launch (no finalize kernel, no
// extra launch boundary): every workgroup publishes its six partials
// write-through (sc1) and takes a ticket on its group's counter; the group's
// last arriver adds the members' partials (fixed shuffle tree = index order),
// publishes the group partial and takes a ticket on the launch counter; the
// launch's last arriver adds the group partials in group order, builds the scalar
// pack and (unsharded x) runs the decide step.  Wave 0 does the publish/ticket
// BEFORE it stores its own x+ tile: `s_waitcnt vmcnt(0)` waits for every earlier
// store of the wave in issue order, and the 8-byte publishes complete in ~1 us
// while HBM stores take several.  Hand-off protocol: guide section 6
// Guideline 16 (R1, counter form) / MI355X_MICROARCH "Valid forms": sc1 stores,
// the storing wave drains vmcnt, one relaxed agent-scope add per workgroup, the
// reducer takes one agent-scope acquire and reads with sc1 loads.  Sums are
// added in workgroup / group index order, never arrival order: deterministic.
// Alternatives measured and rejected (tools/tune_trial.hip, n = 1e8 / 1e7):
// separate one-workgroup finalize kernel +21 us +2 launch gaps; reducer
// workgroups appended to the grid that poll the counters: same at 1e8, +10 us
// at 1e7 (they are dispatched last and start late); persistent tile walk:
// device-dependent, up to 9 % slower than one workgroup per tile.
// ---------------------------------------------------------------------------
constexpr int ZF_TILE_U = 4;
constexpr int ZF_TILE_UNITS = ZF_TILE_U * ZF_BLOCK;   // 16-byte units per tile
constexpr int ZF_MAX_TILES_PER_WG = 8;                // upper bound of zf_step_args.tiles_per_wg
constexpr int ZF_GROUP = 64;                          // workgroups per reduction group

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

struct zf_reduce_ws {
    double* blk_part;     // ZF_NPART x nblocks, quantity-major
    double* grp_part;     // ZF_NPART x ngroups
    unsigned* grp_cnt;    // ngroups arrival counters (zero between launches)
    unsigned* top_cnt;    // 1
    double* totals;       // ZF_NPART launch totals (raw sums / max)
};

struct zf_tail_args {
    double scale[ZF_NPART];   // pack[k] = scale[k] * total[k]
    double* pack;             // local pack out (ZF_PACK_LEN), may be NULL
    zf_control* ctl_rw;       // decide in-launch when non-NULL (unsharded separable problem)
    double* trace;
};

struct zf_step_args {
    const zf_control* ctl;
    const double* beta_ring;  // ZF_RING momentum factors, indexed by accepted count
    double* xb[3];            // x ring
    const double* p0;         // diag: d        vec: grad
    const double* p1;         // diag: c        vec: unused
    double lam, lo, hi;
    int64_t n;
    int tiles_per_wg;         // interleaved tiles per workgroup (1 .. ZF_MAX_TILES_PER_WG)
    zf_reduce_ws ws;
    zf_tail_args tail;
};

// Elect-and-reduce tail, part 1 (wave 0 only; `mine` = this workgroup's total of
// quantity threadIdx.x, threads < ZF_NPART): publish, drain, ticket.  Sets *s_flag
// (LDS) to 1 in the workgroup whose ticket is the group's last.  Must run BEFORE
// the wave issues its x+ stores.
__device__ __forceinline__ void zf_publish_and_ticket(const zf_reduce_ws& W, double mine, int* s_flag) {
    const int nblocks = gridDim.x;
    const int b = blockIdx.x;
    const int g = b / ZF_GROUP;
    const int g0 = g * ZF_GROUP;
    const int gsize = (nblocks - g0 < ZF_GROUP) ? (nblocks - g0) : ZF_GROUP;
    if (threadIdx.x < ZF_NPART) zf_publish(W.blk_part + (int64_t)threadIdx.x * nblocks + b, mine);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(W.grp_cnt + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == (unsigned)(gsize - 1));
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *s_flag = last;
    }
}

// part 2 (whole workgroup, after a barrier that follows part 1; only the group's
// last workgroup calls it).  Returns true in every thread of the launch's last
// workgroup, with totals[] (LDS) filled.
__device__ __forceinline__ bool zf_group_and_launch_reduce(const zf_reduce_ws& W, double* lds /* >= 40 */,
                                                           double* totals /* LDS, 8 */, int* s_flag) {
    const int nblocks = gridDim.x;
    const int g = blockIdx.x / ZF_GROUP;
    const int ngroups = (nblocks + ZF_GROUP - 1) / ZF_GROUP;
    const int g0 = g * ZF_GROUP;
    const int gsize = (nblocks - g0 < ZF_GROUP) ? (nblocks - g0) : ZF_GROUP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // group reducer: lane l holds workgroup g0+l; fixed shuffle tree = index order.
    // wave w reduces quantities w and w+4 (both loads in flight before the first use)
    {
        const int k0 = wave, k1 = wave + ZF_WAVES;
        double v0 = 0.0, v1 = 0.0;
        if (lane < gsize) {
            v0 = zf_consume(W.blk_part + (int64_t)k0 * nblocks + g0 + lane);
            if (k1 < ZF_NPART) v1 = zf_consume(W.blk_part + (int64_t)k1 * nblocks + g0 + lane);
        }
        v0 = zf_wave_sum(v0);                                            // k0 < 4: always a sum
        v1 = (k1 == ZF_NPART - 1) ? zf_wave_max(v1) : zf_wave_sum(v1);
        if (lane == 0) {
            zf_publish(W.grp_part + (int64_t)k0 * ngroups + g, v0);
            if (k1 < ZF_NPART) zf_publish(W.grp_part + (int64_t)k1 * ngroups + g, v1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(W.grp_cnt + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
        const unsigned t = __hip_atomic_fetch_add(W.top_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == (unsigned)(ngroups - 1));
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *s_flag = last;
    }
    __syncthreads();
    if (!*s_flag) return false;
    // last workgroup of the launch: add the group partials in group order; the six
    // loads of one group index are independent and issued together
    double v[ZF_NPART];
#pragma unroll
    for (int k = 0; k < ZF_NPART; ++k) v[k] = 0.0;
    for (int q = threadIdx.x; q < ngroups; q += ZF_BLOCK) {
        double p[ZF_NPART];
#pragma unroll
        for (int k = 0; k < ZF_NPART; ++k) p[k] = zf_consume(W.grp_part + (int64_t)k * ngroups + q);
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
    if (threadIdx.x < ZF_NPART) {
        const int k = threadIdx.x;
        double r = lds[k];
        for (int w = 1; w < ZF_WAVES; ++w) r = (k == ZF_NPART - 1) ? fmax(r, lds[w * 8 + k]) : r + lds[w * 8 + k];
        totals[k] = r;
        W.totals[k] = r;
    }
    if (threadIdx.x == 0) __hip_atomic_store(W.top_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return true;
}

// GRAD_INLINE: true  -> separable quadratic, gradient computed from d, c
//              false -> gradient vector read from HBM (least squares)
// NT: nontemporal policy for the once-touched streams (p0, p1 loads, x+ stores)
template <bool GRAD_INLINE, bool NESTEROV, bool BOX, bool NT>
__global__ __launch_bounds__(ZF_BLOCK) void zf_trial_kernel(zf_step_args A) {
    __shared__ double lds[ZF_WAVES * 8 + 8];
    __shared__ double totals[8];
    __shared__ int s_flag;
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
    // per-workgroup costs - block reduction, publish, ticket - are paid once per T tiles.
    // Which T wins is device-dependent (T = 4: -1 % on some MI355X boxes, +7 % on others, against
    // T = 1), so the solver measures it once at initialisation (zf_solver_autotune).
    const int64_t ntiles = (n2 + ZF_TILE_UNITS - 1) / ZF_TILE_UNITS;
    const int64_t G = gridDim.x;
    zf_d2 r[ZF_TILE_U];
    int64_t deferred_base = -1;   // the last full tile's x+ stays in registers until after the ticket
    for (int t = 0; t < A.tiles_per_wg; ++t) {
        const int64_t tile = (int64_t)t * G + blockIdx.x;
        if (tile >= ntiles) break;
        if (deferred_base >= 0) {
#pragma unroll
            for (int u = 0; u < ZF_TILE_U; ++u) zf_st2<NT>(xn2 + deferred_base + u * ZF_BLOCK, r[u]);
            deferred_base = -1;
        }
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
            deferred_base = base;
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
    // wave 0 publishes the partials and takes the ticket while its last x+ tile is still in
    // registers; waves 1-3 stream theirs out meanwhile
    if (threadIdx.x < 64) zf_publish_and_ticket(A.ws, mine, &s_flag);
    if (deferred_base >= 0) {
#pragma unroll
        for (int u = 0; u < ZF_TILE_U; ++u) zf_st2<NT>(xn2 + deferred_base + u * ZF_BLOCK, r[u]);
    }
    __syncthreads();
    if (!s_flag) return;
    if (!zf_group_and_launch_reduce(A.ws, lds, totals, &s_flag)) return;
    // last workgroup of the launch
    if (threadIdx.x == 0 && A.tail.pack) {
        double* pack = A.tail.pack;
        pack[ZF_PK_FY] = A.tail.scale[0] * totals[0];
        pack[ZF_PK_DOT] = totals[1];
        pack[ZF_PK_SS] = totals[2];
        pack[ZF_PK_GX] = A.tail.scale[3] * totals[3];
        pack[ZF_PK_FX] = A.tail.scale[4] * totals[4];
        pack[ZF_PK_ERR] = totals[5];
        pack[6] = 0.0;
        pack[7] = 0.0;
        if (A.tail.ctl_rw) zf_decide_step(A.tail.ctl_rw, pack, A.tail.trace, A.beta_ring);
    }
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
