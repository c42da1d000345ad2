// zf_kernels_ls_small.h - dense least squares whose matrix is small enough to be launch-bound
// (BASELINE cfg1: A 512 x 1024 = 4 MiB, cache-resident): the seven launches of a trial
// (residual, column sweep, combine, prox step, row sweep, residual, finalize + decide:
// 36 us per iteration, 1.5 - 2 us of it per kernel boundary) become TWO:
//
//   zf_ls_small_step_kernel   one workgroup per panel of 32 columns:
//       r = A y - b by linearity from the cached A x_k, A x_{k-1} (every workgroup, m <= 4096 values
//       into LDS; workgroup 0 also leaves f(y)), grad_j = 2 scale sum_i A_ij r_i over ALL rows for its
//       columns (16 row groups x 16 column pairs, 16-byte loads, row groups added in order), then
//       the prox step of exactly those columns (zf_elem_vec: proximal_gradient.py:534,:148,:150-152,
//       :510) and its four partial sums
//   zf_ls_small_rows_kernel   one wave per row: s+ = A x+; workgroup sums of (s+ - b)^2; the LAST
//       ARRIVER adds them in index order -> f(x+) (:295), adds the step kernel's partials -> pack,
//       and runs the decide pass (:149-155,:298-307,:525,:539)
//
// A kernel boundary between the two is the grid-wide dependency (every row needs all of x+); a
// grid barrier inside one launch would cost more than the boundary (MI355X_MICROARCH.md: 4 - 7 us
// against 1.5 us).  Same rings, control block and decide logic as the general path
// (zf_kernels_gemv.h + zf_kernels_step.h); only the summation orders of grad, f(y), f(x+) differ.
#pragma once
#include "zf_kernels_gemv.h"
#include "zf_kernels_step.h"

constexpr int LS_SMALL_COLS = 32;        // columns per workgroup of the step kernel
constexpr int LS_SMALL_MAX_M = 4096;     // rows (r lives in LDS)
constexpr int64_t LS_SMALL_MAX_ELEMS = int64_t(1) << 22;   // 32 MiB of fp64: beyond that the sweeps are HBM-bound

struct zf_ls_small_args {
    zf_control* ctl;
    const double* beta_ring;
    double* xb[3];
    zf_ring3 sring;
    const double* A;
    const double* b;
    int64_t m, n;
    double scale, lam, lo, hi;
    double* ls_scal;       // [0] f(y)  [1] f(x+)
    double* blk_part;      // ZF_NPART x grid_step partials of the step kernel
    int grid_step;
    double* row_part;      // workgroup sums of (s+ - b)^2, one per workgroup of the rows kernel
    unsigned* cnt;         // arrival counter of the rows kernel (zero between launches)
    double* pack;
    double* trace;
    double* hist;          // streaming return_all (or NULL)
    int64_t hist_cap, hist_stride;
    int* pass_log;         // timing only (else NULL): slot pass_slot receives 1 when the step kernel ran a trial
    int pass_slot;
    int pass_tag;          // (launch number & 0x7fff) << 16
};

template <bool NESTEROV, bool BOX>
__global__ __launch_bounds__(ZF_BLOCK) void zf_ls_small_step_kernel(zf_ls_small_args P) {
    __shared__ double s_r[LS_SMALL_MAX_M];
    __shared__ double s_part[16][LS_SMALL_COLS];
    __shared__ double s_red[ZF_WAVES];
    const zf_control* ctl = P.ctl;
    if (ctl->status != ZF_RUNNING) return;
    const int tid = threadIdx.x;
    if (P.pass_log && blockIdx.x == 0 && tid == 0) P.pass_log[P.pass_slot] = P.pass_tag | 1;   // (one trial, nothing replayed)
    const int cur = ctl->cur, prev = ctl->prev;
    const double beta = NESTEROV ? ctl->beta_next : 0.0;
    const double lr = ctl->lr;
    const int64_t m = P.m, n = P.n;
    // r = A y - b,  A y = s_k + beta (s_k - s_{k-1})
    const double* __restrict__ sk = P.sring.p[cur];
    const double* __restrict__ so = P.sring.p[(cur + 2) % 3];
    double rr = 0.0;
    for (int64_t i = tid; i < m; i += ZF_BLOCK) {
        double ay = sk[i];
        if (NESTEROV) ay = ay + beta * (ay - so[i]);
        const double rv = ay - P.b[i];
        s_r[i] = rv;
        rr += rv * rv;
    }
    if (blockIdx.x == 0) {   // f(y) = scale (sqrt(sum r^2))^2, one writer
        rr = zf_wave_sum(rr);
        if ((tid & 63) == 0) s_red[tid >> 6] = rr;
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) {
        double t = s_red[0];
        for (int w = 1; w < ZF_WAVES; ++w) t += s_red[w];
        const double nrm = sqrt(t);
        P.ls_scal[0] = P.scale * (nrm * nrm);
    }
    // column sweep of this workgroup's panel: thread = (column pair p, row group rg)
    const int p = tid & 15, rg = tid >> 4;
    const int64_t j0 = (int64_t)blockIdx.x * LS_SMALL_COLS + 2 * p;
    zf_row2 acc = {0.0, 0.0};
    if (j0 < n) {
        const double* __restrict__ Ap = P.A + j0;
        int64_t i = rg;
        for (; i + 16 * 7 < m; i += 16 * 8) {   // eight rows of this group in flight
            zf_row2 a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const zf_row2*>(Ap + (i + 16 * u) * n);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc.x += a[u].x * s_r[i + 16 * u];
                acc.y += a[u].y * s_r[i + 16 * u];
            }
        }
        for (; i < m; i += 16) {
            const zf_row2 a = *reinterpret_cast<const zf_row2*>(Ap + i * n);
            acc.x += a.x * s_r[i];
            acc.y += a.y * s_r[i];
        }
    }
    s_part[rg][2 * p] = acc.x;
    s_part[rg][2 * p + 1] = acc.y;
    __syncthreads();
    // the prox step of the panel's columns: lanes 0 .. 31 of wave 0
    zf_elem_acc ea = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (tid < 64) {
        const int64_t j = (int64_t)blockIdx.x * LS_SMALL_COLS + tid;
        if (tid < LS_SMALL_COLS && j < n) {
            double t = s_part[0][tid];
#pragma unroll
            for (int g = 1; g < 16; ++g) t += s_part[g][tid];   // row groups in order
            const double grad = (2 * P.scale) * t;
            int first, second;
            zf_free_bufs(cur, prev, 3, &first, &second);
            const double xk = P.xb[cur][j];
            const double xo = NESTEROV ? P.xb[prev][j] : xk;
            const double xn = zf_elem_vec<NESTEROV, BOX>(xk, xo, grad, beta, lr, P.lam * lr, P.lo, P.hi, ea);
            P.xb[first][j] = xn;
            if (P.hist) P.hist[((ctl->nit + 1) % P.hist_cap) * P.hist_stride + j] = xn;
        }
        const double dot = zf_wave_sum(ea.dot), ss = zf_wave_sum(ea.ss), l1 = zf_wave_sum(ea.l1);
        const double mx = zf_wave_max(ea.mx);
        if (tid == 0) {
            const int64_t G = P.grid_step;
            P.blk_part[0 * G + blockIdx.x] = 0.0;
            P.blk_part[1 * G + blockIdx.x] = dot;
            P.blk_part[2 * G + blockIdx.x] = ss;
            P.blk_part[3 * G + blockIdx.x] = l1;
            P.blk_part[4 * G + blockIdx.x] = 0.0;
            P.blk_part[5 * G + blockIdx.x] = mx;
        }
    }
}

__global__ __launch_bounds__(ZF_BLOCK) void zf_ls_small_rows_kernel(zf_ls_small_args P) {
    __shared__ double s_w[ZF_WAVES];
    __shared__ zf_trial_eval s_pre[ZF_MAX_SUB_ITERS];
    __shared__ double s_pack[ZF_PACK_LEN];
    __shared__ int s_last;
    zf_control* ctl = P.ctl;
    if (ctl->status != ZF_RUNNING) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cur = ctl->cur, prev = ctl->prev;
    int first, second;
    zf_free_bufs(cur, prev, 3, &first, &second);
    const double* __restrict__ x = P.xb[first];            // x+ of the step kernel
    double* __restrict__ s_out = P.sring.p[(cur + 1) % 3];  // A x+ (slot 1 of the s ring, as zf_gemv_rows_kernel)
    const int64_t m = P.m, n = P.n;
    const int64_t row = (int64_t)blockIdx.x * ZF_WAVES + wave;
    double sq = 0.0;
    if (row < m) {
        const zf_row2* __restrict__ Ar = reinterpret_cast<const zf_row2*>(P.A + row * n);
        const zf_row2* __restrict__ xv = reinterpret_cast<const zf_row2*>(x);
        double acc = 0.0;
        int64_t j = lane;
        for (; j + 64 * 7 < n / 2; j += 64 * 8) {   // eight 16-byte loads of the row (and of x+) in flight
            zf_row2 a[8], xx[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = Ar[j + 64 * u];
                xx[u] = xv[j + 64 * u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += a[u].x * xx[u].x + a[u].y * xx[u].y;
        }
        for (; j < n / 2; j += 64) {
            const zf_row2 a = Ar[j], xx = xv[j];
            acc += a.x * xx.x + a.y * xx.y;
        }
        acc = zf_wave_sum(acc);
        if (lane == 0) {
            s_out[row] = acc;
            const double rv = acc - P.b[row];
            sq = rv * rv;
        }
    }
    if (lane == 0) s_w[wave] = sq;
    __syncthreads();
    if (tid == 0) {
        double t = s_w[0];
        for (int w = 1; w < ZF_WAVES; ++w) t += s_w[w];
        zf_publish(P.row_part + blockIdx.x, t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tk = __hip_atomic_fetch_add(P.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (tk == (unsigned)(gridDim.x - 1));
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(P.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last || wave != 0) return;
    // last arriver, wave 0: f(x+) from the workgroup sums (index order), the step kernel's partials, decide
    double fx = 0.0;
    for (int g0 = 0; g0 < (int)gridDim.x; g0 += 64) {
        const int g = g0 + lane;
        const double v = g < (int)gridDim.x ? zf_consume(P.row_part + g) : 0.0;
        fx += zf_wave_sum(v);   // (chunks of 64 workgroups in order; within a chunk the fixed shuffle tree)
    }
    const double nrm = sqrt(fx);
    const double f_x = P.scale * (nrm * nrm);
    double dot = 0.0, ss = 0.0, l1 = 0.0, mx = 0.0;
    const int64_t G = P.grid_step;
    for (int64_t g0 = 0; g0 < G; g0 += 64) {
        const int64_t g = g0 + lane;
        const bool in = g < G;
        dot += zf_wave_sum(in ? P.blk_part[1 * G + g] : 0.0);
        ss += zf_wave_sum(in ? P.blk_part[2 * G + g] : 0.0);
        l1 += zf_wave_sum(in ? P.blk_part[3 * G + g] : 0.0);
        mx = fmax(mx, zf_wave_max(in ? P.blk_part[5 * G + g] : 0.0));
    }
    double pk[ZF_PACK_LEN];
    pk[ZF_PK_FY] = P.ls_scal[0];
    pk[ZF_PK_DOT] = __shfl(dot, 0, 64);
    pk[ZF_PK_SS] = __shfl(ss, 0, 64);
    pk[ZF_PK_GX] = P.lam * __shfl(l1, 0, 64);
    pk[ZF_PK_FX] = __shfl(f_x, 0, 64);
    pk[ZF_PK_ERR] = __shfl(mx, 0, 64);
    pk[6] = 0.0;
    pk[7] = 0.0;
    if (lane == 0) {
        P.ls_scal[1] = pk[ZF_PK_FX];
#pragma unroll
        for (int k = 0; k < ZF_PACK_LEN; ++k) {
            P.pack[k] = pk[k];
            s_pack[k] = pk[k];
        }
    }
    zf_decide_pass_wave(ctl, s_pack, pk, P.trace, P.beta_ring, lane, 64, s_pre);
}
