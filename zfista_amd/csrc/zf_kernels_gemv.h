// zf_kernels_gemv.h - dense fp64 least-squares kernels (K14):
//   f(x) = scale |Ax - b|^2,  grad f(y) = 2 scale A^T (A y - b)
// (tests/test_proximal_gradient.py:49-57 with scale = 1/6; BASELINE cfg1/cfg3
// with scale = 1/2).  A is row-major m_rows x n in HBM and is streamed exactly
// twice per accepted trial: once by rows for A x+ and once by columns for
// A^T r.  A y is obtained by linearity from the cached A x_k, A x_{k-1}.
// Both sweeps are HBM-bound (0.25 flop/B); they use 16-B loads (V = 2; V = 1
// is the scalar form for odd n, where rows are not 16-B aligned), one row
// panel or column panel per workgroup and fixed-order reductions.
#pragma once
#include "zf_common.h"

struct zf_ring3 {
    double* p[3];
};

template <int V> struct zf_vec;
template <> struct zf_vec<1> { using type = double; };
template <> struct zf_vec<2> { using type = double2; };
template <int V> __device__ __forceinline__ void zf_fma_acc(double (&acc)[V], typename zf_vec<V>::type a, double s);
template <> __device__ __forceinline__ void zf_fma_acc<1>(double (&acc)[1], double a, double s) { acc[0] += a * s; }
template <> __device__ __forceinline__ void zf_fma_acc<2>(double (&acc)[2], double2 a, double s) {
    acc[0] += a.x * s;
    acc[1] += a.y * s;
}
template <int V> __device__ __forceinline__ double zf_dot_v(typename zf_vec<V>::type a, typename zf_vec<V>::type b);
template <> __device__ __forceinline__ double zf_dot_v<1>(double a, double b) { return a * b; }
template <> __device__ __forceinline__ double zf_dot_v<2>(double2 a, double2 b) { return a.x * b.x + a.y * b.y; }

// ---- s_out = A x_in  (row dot products) ------------------------------------
// One workgroup owns GEMV_ROWS consecutive rows; threads sweep the columns in
// V-wide units so every wave load is contiguous per row; x (n doubles) is
// re-read by every workgroup from L2 / Infinity Cache.
// slot: -1 -> x_in = xr.p[0], s_out = sr.p[0] (plain call)
//       >=0 -> relative ring slot: index (ctl->cur + slot) % 3 on both rings.
// Measured on 16384 x 65536 (rocprofv3, one sweep = 8.59 GB): 2 rows per workgroup 1.390 ms,
// 4: 1.418, 8: 1.415, 16: 1.484 (rocBLAS gemv on the same matrix: 1.672 ms).
#ifndef ZF_GEMV_ROWS
#define ZF_GEMV_ROWS 2
#endif
#ifndef ZF_GEMV_NT
#define ZF_GEMV_NT 1   // A is read once per sweep: nontemporal loads keep x / r cache-resident
#endif
constexpr int GEMV_ROWS = ZF_GEMV_ROWS;
template <typename T> __device__ __forceinline__ T zf_ld_stream(const T* p) {
#if ZF_GEMV_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
typedef double zf_row2 __attribute__((ext_vector_type(2)));
template <int V> struct zf_rowvec;
template <> struct zf_rowvec<1> { using type = double; };
template <> struct zf_rowvec<2> { using type = zf_row2; };
__device__ __forceinline__ double zf_dot_row(double a, double b) { return a * b; }
__device__ __forceinline__ double zf_dot_row(zf_row2 a, double2 b) { return a.x * b.x + a.y * b.y; }
template <int V>
__global__ __launch_bounds__(ZF_BLOCK) void zf_gemv_rows_kernel(const zf_control* ctl,
                                                                const double* __restrict__ A,
                                                                zf_ring3 xr, zf_ring3 sr, int slot,
                                                                int64_t m_rows, int64_t n) {
    using VT = typename zf_vec<V>::type;
    __shared__ double lds[ZF_WAVES * GEMV_ROWS];
    int idx = 0;
    if (slot >= 0) {
        if (ctl->status != ZF_RUNNING) return;
        idx = (ctl->cur + slot) % 3;
    }
    const double* __restrict__ x = xr.p[idx];
    double* __restrict__ s = sr.p[idx];
    const int64_t nv = n / V;
    const VT* __restrict__ xv = reinterpret_cast<const VT*>(x);
    for (int64_t row0 = (int64_t)blockIdx.x * GEMV_ROWS; row0 < m_rows;
         row0 += (int64_t)gridDim.x * GEMV_ROWS) {
        double acc[GEMV_ROWS];
#pragma unroll
        for (int r = 0; r < GEMV_ROWS; ++r) acc[r] = 0.0;
        for (int64_t j = threadIdx.x; j < nv; j += ZF_BLOCK) {
            const VT xj = xv[j];
#pragma unroll
            for (int r = 0; r < GEMV_ROWS; ++r) {
                if (row0 + r < m_rows) {
                    using RT = typename zf_rowvec<V>::type;
                    const RT a = zf_ld_stream(reinterpret_cast<const RT*>(A + (row0 + r) * n) + j);
                    acc[r] += zf_dot_row(a, xj);
                }
            }
        }
        const double maxs[1] = {0.0};
        double out = 0.0;
        zf_block_reduce<GEMV_ROWS, 0, ZF_WAVES>(acc, maxs, lds, out);
        if (threadIdx.x < GEMV_ROWS && row0 + threadIdx.x < m_rows) s[row0 + threadIdx.x] = out;
        __syncthreads();
    }
}

// ---- residual at y by linearity + f(y) -------------------------------------
// r = (s_k + beta (s_k - s_{k-1})) - b ;  f(y) = scale * (sqrt(sum r^2))^2
// One workgroup (m_rows is at most a few 10^4).  Skipped unless ctl->need_grad.
constexpr int RESID_BLOCK = 1024;
__device__ __forceinline__ double zf_block_sum_1024(double acc, double* lds) {
    acc = zf_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    double t = lds[0];
    for (int w = 1; w < RESID_BLOCK / 64; ++w) t += lds[w];
    return t;
}

__global__ __launch_bounds__(RESID_BLOCK) void zf_resid_y_kernel(const zf_control* ctl,
                                                                 const double* beta_ring, zf_ring3 sr,
                                                                 const double* __restrict__ b,
                                                                 double* __restrict__ r, double scale,
                                                                 int64_t m_rows, double* f_out,
                                                                 int nesterov) {
    __shared__ double lds[RESID_BLOCK / 64];
    if (ctl->status != ZF_RUNNING || !ctl->need_grad) return;
    const int cur = ctl->cur;
    const double beta = nesterov ? ctl->beta_next : 0.0;
    (void)beta_ring;
    const double* __restrict__ sk = sr.p[cur];
    const double* __restrict__ so = sr.p[(cur + 2) % 3];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < m_rows; i += RESID_BLOCK) {
        double ay = sk[i];
        if (nesterov) ay = ay + beta * (ay - so[i]);
        const double rv = ay - b[i];
        r[i] = rv;
        acc += rv * rv;
    }
    const double t = zf_block_sum_1024(acc, lds);
    if (threadIdx.x == 0) {
        const double nrm = sqrt(t);
        *f_out = scale * (nrm * nrm);
    }
}

// f(x) = scale * (sqrt(sum (s - b)^2))^2 for a computed s = A x
// slot as in zf_gemv_rows_kernel.
__global__ __launch_bounds__(RESID_BLOCK) void zf_resid_x_kernel(const zf_control* ctl, zf_ring3 sr,
                                                                 int slot, const double* __restrict__ b,
                                                                 double scale, int64_t m_rows,
                                                                 double* f_out) {
    __shared__ double lds[RESID_BLOCK / 64];
    int idx = 0;
    if (slot >= 0) {
        if (ctl->status != ZF_RUNNING) return;
        idx = (ctl->cur + slot) % 3;
    }
    const double* __restrict__ s = sr.p[idx];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < m_rows; i += RESID_BLOCK) {
        const double rv = s[i] - b[i];
        acc += rv * rv;
    }
    const double t = zf_block_sum_1024(acc, lds);
    if (threadIdx.x == 0) {
        const double nrm = sqrt(t);
        *f_out = scale * (nrm * nrm);
    }
}

// The same for LONG residuals (the operator problem: m = n pixels) in two launches: every workgroup the share of its
// contiguous chunk -> partials[blockIdx.x]; then one workgroup adds the shares in order.  (One workgroup alone took 8 ms
// for 1.7e7 pixels at the initialisation of a 4096 x 4096 deblurring solve.)
__global__ __launch_bounds__(ZF_BLOCK) void zf_resid_x_wide_kernel(const double* __restrict__ s, const double* __restrict__ b,
                                                                   int64_t m_rows, double* __restrict__ partials) {
    __shared__ double lds[ZF_WAVES];
    const int64_t per = (m_rows + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < m_rows ? lo + per : m_rows;
    double acc = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += ZF_BLOCK) {
        const double rv = s[i] - b[i];
        acc += rv * rv;
    }
    acc = zf_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = lds[0];
        for (int w = 1; w < ZF_WAVES; ++w) t += lds[w];
        partials[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(RESID_BLOCK) void zf_resid_x_finish_kernel(const double* __restrict__ partials, int count, double scale,
                                                                       double* f_out) {
    __shared__ double lds[RESID_BLOCK / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += RESID_BLOCK) acc += partials[i];
    const double t = zf_block_sum_1024(acc, lds);
    if (threadIdx.x == 0) {
        const double nrm = sqrt(t);
        *f_out = scale * (nrm * nrm);
    }
}

// s <- s - b (residual in place; operator evaluation outside the solver loop)
__global__ __launch_bounds__(ZF_BLOCK) void zf_axmb_kernel(double* __restrict__ s,
                                                           const double* __restrict__ b, int64_t m_rows) {
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; i < m_rows; i += stride) s[i] = s[i] - b[i];
}

// ---- grad = 2 scale A^T r  (column sums) ------------------------------------
// Grid (column panels of 256*V, row slices).  Each thread owns V adjacent
// columns (one V*8-byte load per row) and walks its row slice; slice partials
// go to a slab [slices][n] that a second kernel adds in slice order (no
// atomics: bit-reproducible).
constexpr int GEMVT_UNROLL = 8;
template <int V>
__global__ __launch_bounds__(ZF_BLOCK) void zf_gemvT_partial_kernel(const zf_control* ctl,
                                                                    const double* __restrict__ A,
                                                                    const double* __restrict__ r,
                                                                    double* __restrict__ slab,
                                                                    int64_t m_rows, int64_t n,
                                                                    int64_t rows_per_slice) {
    using VT = typename zf_vec<V>::type;
    if (ctl && (ctl->status != ZF_RUNNING || !ctl->need_grad)) return;   // ctl == NULL: plain call
    const int64_t colv = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x;  // V-wide unit
    const int64_t nv = n / V;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_slice;
    int64_t r1 = r0 + rows_per_slice;
    if (r1 > m_rows) r1 = m_rows;
    if (colv >= nv) return;
    double acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.0;
    const VT* __restrict__ Ac = reinterpret_cast<const VT*>(A) + colv;
    int64_t i = r0;
    for (; i + GEMVT_UNROLL <= r1; i += GEMVT_UNROLL) {
        VT a[GEMVT_UNROLL];
#pragma unroll
        for (int u = 0; u < GEMVT_UNROLL; ++u) a[u] = Ac[(i + u) * nv];
#pragma unroll
        for (int u = 0; u < GEMVT_UNROLL; ++u) zf_fma_acc<V>(acc, a[u], r[i + u]);
    }
    for (; i < r1; ++i) zf_fma_acc<V>(acc, Ac[i * nv], r[i]);
    double* out = slab + (int64_t)blockIdx.y * n + colv * V;
#pragma unroll
    for (int v = 0; v < V; ++v) out[v] = acc[v];
}

// ---- the same column sweep on the matrix cores (v_mfma_f64_16x16x4_f64) ---------------------
// A wave owns a 32-column panel and takes 4 rows per step.  B operand: lane l holds
// B[k = l>>4][col = l&15]; with one 16-B load a lane fetches columns 2c, 2c+1 (c = l&15) of row
// i0 + (l>>4), which feeds two MFMAs (even / odd columns of the panel).  A operand: lane l holds
// A[row = l&15][k = l>>4] = r[i0 + (l>>4)] (all 16 rows identical, so every row of D is the
// wanted 16-column strip).  C/D: col = l&15, row = (l>>4) + 4*reg -> row 0 is lanes 0..15,
// register 0.  Each matrix element is used once, so an MFMA consumes exactly the 64 elements a
// v_fma_f64 wave instruction does: measured 6.27 TB/s against 6.26 TB/s for the VALU form on
// 16384 x 65536 (tools/tune_gemv.hip) - both at the HBM roof, the matrix pipe ~2 % busy.
typedef double zf_f64x2 __attribute__((ext_vector_type(2)));
typedef double zf_f64x4 __attribute__((ext_vector_type(4)));
constexpr int GEMVT_MFMA_UNROLL = 4;
constexpr int GEMVT_MFMA_COLS = 128;   // columns per workgroup: 4 waves x 32
__global__ __launch_bounds__(ZF_BLOCK) void zf_gemvT_partial_mfma_kernel(const zf_control* ctl,
                                                                         const double* __restrict__ A,
                                                                         const double* __restrict__ r,
                                                                         double* __restrict__ slab,
                                                                         int64_t m_rows, int64_t n,
                                                                         int64_t rows_per_slice) {
    if (ctl && (ctl->status != ZF_RUNNING || !ctl->need_grad)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t j0 = ((int64_t)blockIdx.x * ZF_WAVES + wave) * 32;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_slice;
    int64_t r1 = r0 + rows_per_slice;
    if (r1 > m_rows) r1 = m_rows;
    if (j0 >= n) return;
    const int kq = lane >> 4, c = lane & 15;
    const double* __restrict__ Ap = A + j0 + 2 * c;
    zf_f64x4 acc_e = {0.0, 0.0, 0.0, 0.0}, acc_o = {0.0, 0.0, 0.0, 0.0};
    int64_t i = r0;
    for (; i + 4 * GEMVT_MFMA_UNROLL <= r1; i += 4 * GEMVT_MFMA_UNROLL) {
        zf_f64x2 b[GEMVT_MFMA_UNROLL];
        double rv[GEMVT_MFMA_UNROLL];
#pragma unroll
        for (int u = 0; u < GEMVT_MFMA_UNROLL; ++u) {
            b[u] = zf_ld_stream(reinterpret_cast<const zf_f64x2*>(Ap + (i + 4 * u + kq) * n));
            rv[u] = r[i + 4 * u + kq];
        }
#pragma unroll
        for (int u = 0; u < GEMVT_MFMA_UNROLL; ++u) {
            acc_e = __builtin_amdgcn_mfma_f64_16x16x4f64(rv[u], b[u].x, acc_e, 0, 0, 0);
            acc_o = __builtin_amdgcn_mfma_f64_16x16x4f64(rv[u], b[u].y, acc_o, 0, 0, 0);
        }
    }
    if (kq == 0) {
        double te = acc_e[0], to = acc_o[0];
        for (; i < r1; ++i) {   // fewer than 16 rows left in the slice
            const zf_f64x2 a = *reinterpret_cast<const zf_f64x2*>(Ap + i * n);
            const double x = r[i];
            te += a.x * x;
            to += a.y * x;
        }
        zf_f64x2 o;
        o.x = te;
        o.y = to;
        *reinterpret_cast<zf_f64x2*>(slab + (int64_t)blockIdx.y * n + j0 + 2 * c) = o;
    }
}

__global__ __launch_bounds__(ZF_BLOCK) void zf_gemvT_combine_kernel(const zf_control* ctl,
                                                                    const double* __restrict__ slab,
                                                                    double* __restrict__ grad,
                                                                    double two_scale, int64_t n,
                                                                    int slices) {
    if (ctl && (ctl->status != ZF_RUNNING || !ctl->need_grad)) return;
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; j < n; j += stride) {
        // slices are added in slice order; eight loads are in flight at a time (a rolled loop
        // paid one full load latency per slice: 16 us at 64 slices)
        double t = slab[j];
        int s = 1;
        for (; s + 8 <= slices; s += 8) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = slab[(int64_t)(s + k) * n + j];
#pragma unroll
            for (int k = 0; k < 8; ++k) t += v[k];
        }
        for (; s < slices; ++s) t += slab[(int64_t)s * n + j];
        grad[j] = two_scale * t;
    }
}
