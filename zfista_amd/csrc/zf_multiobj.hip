// zf_multiobj.hip - device side of the multi-objective trial (m >= 2 objectives).
//
// The dual of the scalarised subproblem is minimised on the host by SciPy, as in
// the reference (zfista/proximal_gradient.py:179-205); everything O(n) runs here:
//   prepare   J = jac_f(y), f(y)                       problems.py:193-205 (JOS1), :312-328 (FDS)
//   dual_eval v = y - lr (w @ J); p = prox(lr w, v);   proximal_gradient.py:162-173
//             g_i(p), |p - v|^2, |w @ J|^2, J (p - y)  -> 2m+2 scalars, nothing written
//   recover   x+ = prox(lr w*, y - lr w* @ J), max|x+ - y|       :206, :510
//   eval_F    f(x), g(x)                               :279, :295   (g: problems.py:101-117)
//   commit    x_old <- x_k <- x+ ; y = x_k + beta (x_k - x_old)  :534-538
// g / prox are the shifted-l1 + box family shared by every reference problem
// (problems.py:101-138), including the quirk that shift 0 is (net) ignored in the
// prox (:129).  x, y, J stay resident in HBM between calls.
#include <vector>

#include <atomic>

#include "zf_common.h"
#include "zf_dual_native.h"

namespace {

constexpr int MO_MAX_M = 8;
constexpr int MO_GRID_MAX = 1024;

struct mo_g {             // g / prox descriptor, passed by value
    int m;
    int has_l1;
    int has_box;
    double ratio[MO_MAX_M];
    double shift[MO_MAX_M];
    double lo, hi;
    const double* lo_v;   // per-coordinate bounds (device, n each) or NULL: bounds given as arrays
    const double* hi_v;   // (zfista/problems.py:69-70 allows either)
};

// prox_wsum_g(weight, x), problems.py:126-138; coef = weight * l1_ratios
__device__ __forceinline__ double mo_prox(const mo_g& G, const double* coef, double tail_sum, double x, int64_t j) {
    if (G.has_l1) {
        // stage 0: prox_lasso(x + sum(coef[1:]) - s0 + s0, coef[0])
        x = zf_soft_threshold(x + tail_sum - G.shift[0] + G.shift[0], coef[0]);
        for (int i = 1; i < G.m; ++i) x = zf_soft_threshold(x - coef[i] - G.shift[i], coef[i]) + G.shift[i];
    }
    if (G.has_box) x = zf_clip(x, G.lo_v ? G.lo_v[j] : G.lo, G.hi_v ? G.hi_v[j] : G.hi);
    return x;
}

struct mo_w {
    double w[MO_MAX_M];      // dual weights
    double coef[MO_MAX_M];   // (lr * w) * l1_ratios
    double tail_sum;         // sum(coef[1:])
    double lr;
};

// ---- dual evaluation: reads J (m x n), y ; writes nothing --------------------------
// partials (quantity-major): [0..m) sum|p - s_i|, [m] |p-v|^2, [m+1] |wJ|^2, [m+2 .. 2m+2) J_i.(p-y)
template <int M>
__global__ __launch_bounds__(ZF_BLOCK) void k_dual_eval(const double* __restrict__ J, const double* __restrict__ y,
                                                        mo_g G, mo_w W, int64_t n, double* partials) {
    constexpr int NQ = 2 * M + 2;
    __shared__ double lds[ZF_WAVES * NQ];
    double acc[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) acc[k] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; j < n; j += stride) {
        double Jc[M];
        double wJ = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            Jc[i] = J[(int64_t)i * n + j];
            wJ += W.w[i] * Jc[i];                       // weight @ jac_f_yk
        }
        const double yj = y[j];
        const double v = yj - W.lr * wJ;
        const double p = mo_prox(G, W.coef, W.tail_sum, v, j);
#pragma unroll
        for (int i = 0; i < M; ++i) acc[i] += fabs(p - G.shift[i]);
        const double dv = p - v;
        acc[M] += dv * dv;
        acc[M + 1] += wJ * wJ;
        const double dy = p - yj;
#pragma unroll
        for (int i = 0; i < M; ++i) acc[M + 2 + i] += Jc[i] * dy;
    }
    const double maxs[1] = {0.0};
    double out = 0.0;
    zf_block_reduce<NQ, 0, ZF_WAVES>(acc, maxs, lds, out);
    if (threadIdx.x < NQ) partials[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = out;
}

// ---- primal recovery: x+ = prox(lr w, y - lr w@J); partial: [0] max|x+ - y| ------------------
template <int M>
__global__ __launch_bounds__(ZF_BLOCK) void k_recover(const double* __restrict__ J, const double* __restrict__ y,
                                                      double* __restrict__ xn, mo_g G, mo_w W, int64_t n,
                                                      double* partials) {
    __shared__ double lds[ZF_WAVES];
    double mx = 0.0;
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; j < n; j += stride) {
        double wJ = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) wJ += W.w[i] * J[(int64_t)i * n + j];
        const double yj = y[j];
        const double p = mo_prox(G, W.coef, W.tail_sum, yj - W.lr * wJ, j);
        xn[j] = p;
        mx = fmax(mx, fabs(p - yj));
    }
    const double sums[1] = {0.0};
    const double maxs[1] = {mx};
    double out = 0.0;
    zf_block_reduce<0 + 1, 1, ZF_WAVES>(sums, maxs, lds, out);   // [0] dummy sum, [1] max
    if (threadIdx.x == 1) partials[blockIdx.x] = out;
}

// ---- g(x): partials [0..m) sum|x - s_i|, [m] box violations -----------------------------------
template <int M>
__global__ __launch_bounds__(ZF_BLOCK) void k_g_terms(const double* __restrict__ x, mo_g G, int64_t n,
                                                      double* partials) {
    __shared__ double lds[ZF_WAVES * (M + 1)];
    double acc[M + 1];
#pragma unroll
    for (int k = 0; k <= M; ++k) acc[k] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; j < n; j += stride) {
        const double xv = x[j];
#pragma unroll
        for (int i = 0; i < M; ++i) acc[i] += fabs(xv - G.shift[i]);
        if (G.has_box) acc[M] += (xv < (G.lo_v ? G.lo_v[j] : G.lo) || xv > (G.hi_v ? G.hi_v[j] : G.hi)) ? 1.0 : 0.0;
    }
    const double maxs[1] = {0.0};
    double out = 0.0;
    zf_block_reduce<M + 1, 0, ZF_WAVES>(acc, maxs, lds, out);
    if (threadIdx.x <= M) partials[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = out;
}

// ---- JOS1 (problems.py:193-205): sums [0] x^2, [1] (x-2)^2 ; J rows 2x/n, 2(x-2)/n -----------------
__global__ __launch_bounds__(ZF_BLOCK) void k_jos1_sums(const double* __restrict__ x, int64_t n, double* partials) {
    __shared__ double lds[ZF_WAVES * 2];
    double acc[2] = {0.0, 0.0};
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; j < n; j += stride) {
        const double xv = x[j], t = xv - 2;
        acc[0] += xv * xv;
        acc[1] += t * t;
    }
    const double maxs[1] = {0.0};
    double out = 0.0;
    zf_block_reduce<2, 0, ZF_WAVES>(acc, maxs, lds, out);
    if (threadIdx.x < 2) partials[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = out;
}
// n: local length (J rows are n apart); ng: n_features of the whole problem (the 1/n factors)
__global__ __launch_bounds__(ZF_BLOCK) void k_jos1_jac(const double* __restrict__ x, double* __restrict__ J, int64_t n,
                                                       int64_t ng) {
    const double dn = (double)ng;
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; j < n; j += stride) {
        const double xv = x[j];
        J[j] = 2 * xv / dn;               // 2 * x / n
        J[n + j] = 2 * (xv - 2) / dn;     // 2 * (x - 2) / n
    }
}

// ---- FDS (problems.py:309-328): sums [0] i (x-i)^4, [1] x, [2] x^2, [3] i(n-i+1) e^{-x} -------------
// off: global index of local element 0 (x is a contiguous block of the decision vector)
__global__ __launch_bounds__(ZF_BLOCK) void k_fds_sums(const double* __restrict__ x, int64_t n, int64_t ng, int64_t off,
                                                       double* partials) {
    __shared__ double lds[ZF_WAVES * 4];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; j < n; j += stride) {
        const double xv = x[j];
        const int64_t gj = off + j;
        const double idx = (double)(gj + 1);
        const double conv = (double)((gj + 1) * (ng - gj));   // one_to_n * one_to_n[::-1]
        const double t = xv - idx, t2 = t * t;
        acc[0] += idx * (t2 * t2);
        acc[1] += xv;
        acc[2] += xv * xv;
        acc[3] += conv * exp(-xv);
    }
    const double maxs[1] = {0.0};
    double out = 0.0;
    zf_block_reduce<4, 0, ZF_WAVES>(acc, maxs, lds, out);
    if (threadIdx.x < 4) partials[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = out;
}
// e_mean = exp(sum(x)/n) read from totals[1] (device), so no host round trip between the two passes
__global__ __launch_bounds__(ZF_BLOCK) void k_fds_jac(const double* __restrict__ x, double* __restrict__ J, int64_t n,
                                                      int64_t ng, int64_t off, const double* __restrict__ totals) {
    const double dn = (double)ng;
    const double e_mean = exp(totals[1] / dn);
    const double c1 = 4 / (dn * dn);            // 4 / n**2
    const double den = dn * (dn + 1);           // n (n + 1)
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; j < n; j += stride) {
        const double xv = x[j];
        const int64_t gj = off + j;
        const double idx = (double)(gj + 1);
        const double conv = (double)((gj + 1) * (ng - gj));
        const double t = xv - idx;
        J[j] = c1 * idx * (t * t * t);                  // 4 / n**2 * idx * (x - idx)**3
        J[n + j] = e_mean / dn + 2 * xv;                // exp(sum/n)/n + 2 x
        J[2 * n + j] = -conv * exp(-xv) / den;          // -conv * exp(-x) / (n (n + 1))
    }
}

__global__ __launch_bounds__(ZF_BLOCK) void k_prox_only(double* x, mo_g G, mo_w W, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; j < n; j += stride)
        x[j] = mo_prox(G, W.coef, W.tail_sum, x[j], j);
}

__global__ __launch_bounds__(ZF_BLOCK) void k_commit(double* __restrict__ y, const double* __restrict__ xk,
                                                     const double* __restrict__ xo, double beta, int nesterov,
                                                     int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; j < n; j += stride)
        y[j] = nesterov ? xk[j] + beta * (xk[j] - xo[j]) : xk[j];
}

// one-block fixed-order reduce of `nq` quantities; quantity `max_index` (or -1) is a max
// One wave per quantity (nq <= 16 waves): every lane's loads are independent, so the whole
// reduce is one memory round trip instead of nq dependent ones (it sits on the host's
// critical path once per dual evaluation).
constexpr int MO_RED_WAVES = 16;
__global__ __launch_bounds__(64 * MO_RED_WAVES) void k_mo_reduce(const double* __restrict__ partials, int nblocks,
                                                                 int nq, int max_index, double* out) {
    const int lane = threadIdx.x & 63;
    for (int k = threadIdx.x >> 6; k < nq; k += MO_RED_WAVES) {   // (nq = 2m + 2 <= 18)
        const bool is_max = (k == max_index);
        double v = 0.0;
        for (int b = lane; b < nblocks; b += 64) {
            const double p = partials[(int64_t)k * nblocks + b];
            v = is_max ? fmax(v, p) : v + p;
        }
        v = is_max ? zf_wave_max(v) : zf_wave_sum(v);
        if (lane == 0) out[k] = v;
    }
}


// the gathered totals of all ranks (rank-major, `count` each) added in RANK ORDER - entry max_index a maximum -
// so that every rank continues with bitwise-identical scalars (C3)
__global__ void k_mo_combine_ranks(const double* __restrict__ gathered, int world, int count, int max_index,
                                   double* __restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= count) return;
    double v = gathered[q];
    for (int r = 1; r < world; ++r) {
        const double p = gathered[(int64_t)r * count + q];
        v = (q == max_index) ? fmax(v, p) : v + p;
    }
    out[q] = v;
}

// f(y) from the raw sums, on the device (zf_mo_prepare_async: no host round trip between the sums
// of f(y), the Jacobian kernel that needs sum(y), and the dual search that needs f(y))
__host__ __device__ __attribute__((always_inline)) inline void mo_f_from_sums(int kind, double dn, const double* t, double* f_out) {
    if (kind == ZF_MO_JOS1) {
        const double n0 = sqrt(t[0]), n1 = sqrt(t[1]);
        f_out[0] = n0 * n0 / dn;   // np.linalg.norm(x) ** 2 / n
        f_out[1] = n1 * n1 / dn;
    } else {
        const double nx = sqrt(t[2]);
        f_out[0] = t[0] / (dn * dn);                    // inner(idx, (x-idx)**4) / n**2
        f_out[1] = exp(t[1] / dn) + nx * nx;            // exp(x.sum()/n) + norm(x)**2
        f_out[2] = t[3] / (dn * (dn + 1));              // inner(conv, exp(-x)) / (n (n+1))
    }
}
__global__ void k_f_from_sums(int kind, double dn, const double* __restrict__ t, double* __restrict__ f_out) {
    if (threadIdx.x || blockIdx.x) return;
    mo_f_from_sums(kind, dn, t, f_out);
}

// ---- the whole dual search of one trial in ONE persistent kernel (dual_solver="device") ---------
// Grid: one workgroup of MO_SOLVE_TPB threads per CU (all co-resident).  Every workgroup copies its
// share of (J, y) into LDS once - up to ~156 KB of the CU's 160 KB: n = 1e6, m = 3 needs 125 KB per
// workgroup, so a dual evaluation then touches no global memory at all; elements beyond the LDS
// capacity are streamed from L2 / HBM per evaluation.
// Per batch of points the zf_dual::machine asks for:
//   every workgroup: sums of the batch over its elements -> workgroup totals published write-through
//   -> one ticket; the LAST ARRIVER adds the workgroup totals in index order (deterministic, no float
//   atomics), publishes the batch totals and bumps a generation word; every workgroup picks them up,
//   composes D(w), grad D(w) (proximal_gradient.py:161-177) and lets lane 0 advance its own copy of
//   the machine - all copies see the same numbers in the same order and stay in lock step, so no
//   state is ever broadcast.  At the end the primal recovery x+ = prox(lr w*, y - lr w* @ J) (:206)
//   and max|x+ - y| (:510) are fused in.  One launch, one result read-back per trial.
typedef unsigned long long mo_u64;
constexpr int MO_SOLVE_TPB = 512;
constexpr int MO_SOLVE_LDS_BYTES = 156 * 1024;   // dynamic LDS for the resident elements (static LDS: ~3 KB for m <= 3;
                                                 // larger m: what the kernel's own static part leaves of the CU's 160 KB)
// hand-over records per parity: batches of NB points x NQP sums, at most 2 x (2 m + 2 + m^2) for m = 8 with the exact Hessian
constexpr int MO_REC_CAP = 2 * (2 * MO_MAX_M + 2 + MO_MAX_M * MO_MAX_M + 2);
constexpr int MO_SOLVE_WAVES = MO_SOLVE_TPB / 64;
constexpr unsigned MO_SPIN_LIMIT = 1u << 24;   // polls (each >= ~100 ns): a stuck grid gives up after seconds (default of
                                               // mo_solve_args.spin_limit; ZF_MO_SPIN_LIMIT in the environment overrides it)

struct mo_solve_result {
    double w[MO_MAX_M];
    double fun, err;
    int64_t nit, evals, batches;
    int32_t ok;        // 1 solved, 0 not attempted (non-finite start), -1 the grid-wide wait timed out
    int32_t reserved;
    int64_t cyc_total, cyc_eval, cyc_combine, cyc_step;   // workgroup 0, shader-clock cycles (diagnostics)
    // F(x+) of the recovered point (tail_kind != 0): raw sums of f (problem-specific, see
    // mo_builtin_f) and of g: [0..m) sum |x+ - s_i|, [m] box violations
    double f_sums[4], g_sums[MO_MAX_M + 1];
    int32_t has_F, reserved2;
    double f_y[MO_MAX_M];   // f(y) the search used (computed on the device after zf_mo_prepare_async)
    // the acceptance test of the trial (:298-303), taken by the kernel on F(x+) formed on the device (the
    // built-in problems): what lets the NEXT trial be launched before this record is read (gate)
    double f_x[MO_MAX_M], g_x[MO_MAX_M];
    int32_t accepted;   // 1: F(x+) - F(x_k) <= model + tol for all objectives (or decay_rate == 1); 0: not / not decidable here
    int32_t skipped;    // 1: the launch found its gate closed (the trial before it was not accepted) and did nothing
    // the record lives in pinned host memory and is written by the kernel itself; `seq` = the launch
    // number, stored last behind a system-scope fence: the host waits for it instead of a copy + stream sync
    unsigned long long seq;
};

struct mo_solve_args {
    const double* J;
    const double* y;
    double* xn;
    mo_g G;
    int64_t n;
    double lr;
    double f_y[MO_MAX_M], F_old[MO_MAX_M];
    const double* f_y_dev;   // f(y) left on the device by zf_mo_prepare_async (else NULL: f_y[] above)
    int deprecated, has_w0;
    double w0[MO_MAX_M];
    double tol;
    int64_t max_iter;
    int resident_rows;    // elements per thread kept in LDS (rows of MO_SOLVE_TPB elements)
    int tail_kind;        // ZF_MO_JOS1 / ZF_MO_FDS: also evaluate f(x+), g(x+) (:295); 0: g only
    mo_u64* partials;     // [2][MAXB * NQ][gridDim.x] self-validating 16-byte records (mo_grid_combine)
    mo_u64* totals;       // [2][MAXB * NQ] records
    unsigned nonce;       // launch number: part of the records' key (stale records of earlier launches fail the check)
    mo_solve_result* out;
    // fused outer iteration (zf_mo_set_fused): the kernel forms its own inputs first -
    int prep_kind;        // ZF_MO_JOS1 / ZF_MO_FDS: f(y) and J = jac_f(y) are computed here (problems.py:193-205,
                          // :312-328: what zf_mo_prepare_async enqueues as four launches); 0: they are given
    int make_y;           // 1: y = x_k + beta (x_k - x_{k-1}) first (:534) - the commit of the previous
    int nesterov;         //    accepted trial, deferred into this launch (zf_mo_commit); 0: y is given
    double beta;
    const double* xk;
    const double* xo;
    double* y_w;          // writable y, J (= y, J above) and f(y) on the device: everything formed here is also
    double* J_w;          //   stored, so every other entry point finds the state zf_mo_prepare_async leaves
    double* f_y_w;
    // trials launched ahead (zf_mo_trial_launch): the acceptance decision of this trial goes to *accept_out
    // and F(x+) to F_new_out; a launch with a gate exits at once unless *gate == 1 and takes F(x_k) from
    // F_old_dev - the outputs of the launch before it, which the host has not read yet
    const int* gate;
    const double* F_old_dev;
    int* accept_out;
    double* F_new_out;
    // warm start of a trial launched ahead (:192-205, :286-288): the weights the trial before it ended with, on the
    // device (w_new_out of that launch); NULL: w0[] above / the uniform start
    const double* w0_dev;
    double* w_new_out;
    int decay_is_one;
    double accept_tol;
    unsigned spin_limit;  // polls a grid-wide wait may take before the launch gives up (ok = -1)
    int force_timeout;    // test hook (zf_mo_debug_force_timeout): the first hand-over reports a timeout
};

template <int M>
__device__ __forceinline__ double mo_prox_t(const mo_g& G, const double* coef, double tail_sum, double x, int64_t j) {
    if (G.has_l1) {
        x = zf_soft_threshold(x + tail_sum - G.shift[0] + G.shift[0], coef[0]);
#pragma unroll
        for (int i = 1; i < M; ++i) x = zf_soft_threshold(x - coef[i] - G.shift[i], coef[i]) + G.shift[i];
    }
    if (G.has_box) x = zf_clip(x, G.lo_v ? G.lo_v[j] : G.lo, G.hi_v ? G.hi_v[j] : G.hi);
    return x;
}

// one element's contribution to the 2M+2 sums of a dual evaluation (k_dual_eval's arithmetic)
template <int M>
__device__ __forceinline__ void mo_dual_terms(const mo_g& G, const double (&w)[M], const double (&coef)[M],
                                              double tail_sum, double lr, const double (&Jc)[M], double yj, int64_t j,
                                              double (&acc)[2 * M + 2]) {
    double wJ = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) wJ += w[i] * Jc[i];
    const double v = yj - lr * wJ;
    const double p = mo_prox_t<M>(G, coef, tail_sum, v, j);
#pragma unroll
    for (int i = 0; i < M; ++i) acc[i] += fabs(p - G.shift[i]);
    const double dv = p - v;
    acc[M] += dv * dv;
    acc[M + 1] += wJ * wJ;
    const double dy = p - yj;
#pragma unroll
    for (int i = 0; i < M; ++i) acc[M + 2 + i] += Jc[i] * dy;
}

// prox_wsum_g with its derivatives (forward mode through the composed soft-thresholds and the clip,
// problems.py:126-138): dv = d prox / d x, dc[k] = d prox / d coef_k.  The prox is piecewise linear in
// (x, coef): on the piece the point sits in these are exact (0 / +-1 patterns).
template <int M>
__device__ __forceinline__ double mo_prox_ad(const mo_g& G, const double (&coef)[M], double tail_sum, double x, int64_t j,
                                             double& dv, double (&dc)[M]) {
    dv = 1.0;
#pragma unroll
    for (int k = 0; k < M; ++k) dc[k] = 0.0;
    if (G.has_l1) {
        {   // stage 0: soft-threshold of x + sum(coef[1:]) at coef[0]
            const double u = x + tail_sum - G.shift[0] + G.shift[0];
            const double act = fabs(u) > coef[0] ? 1.0 : 0.0;
            const double sg = u < 0.0 ? -1.0 : 1.0;
            x = zf_soft_threshold(u, coef[0]);
            dv = act;
            dc[0] = -sg * act;
#pragma unroll
            for (int k = 1; k < M; ++k) dc[k] = act;
        }
#pragma unroll
        for (int i = 1; i < M; ++i) {   // stage i: soft-threshold of x - coef[i] - shift[i] at coef[i], shifted back
            const double u = x - coef[i] - G.shift[i];
            const double act = fabs(u) > coef[i] ? 1.0 : 0.0;
            const double sg = u < 0.0 ? -1.0 : 1.0;
            x = zf_soft_threshold(u, coef[i]) + G.shift[i];
            dv *= act;
#pragma unroll
            for (int k = 0; k < M; ++k) dc[k] = act * (dc[k] - (k == i ? 1.0 : 0.0));
            dc[i] -= sg * act;
        }
    }
    if (G.has_box) {
        const double lo = G.lo_v ? G.lo_v[j] : G.lo, hi = G.hi_v ? G.hi_v[j] : G.hi;
        const double inside = (x >= lo && x <= hi) ? 1.0 : 0.0;
        x = zf_clip(x, lo, hi);
        dv *= inside;
#pragma unroll
        for (int k = 0; k < M; ++k) dc[k] *= inside;
    }
    return x;
}

// mo_dual_terms plus the M x M sums of the generalised HESSIAN of the dual (acc[2M + 2 + i M + k]):
//   jac_i = - ratio_i sum_j |p_j - s_i| - sum_j J_ij (p_j - y_j) + const            (:173-177)
//   d p_j / d w_k = lr (- P_j J_kj + ratio_k c_kj)      (v = y - lr w@J, coef_k = lr w_k ratio_k)
//   H_ik = d jac_i / d w_k = lr sum_j (ratio_i sign(p_j - s_i) + J_ij) (P_j J_kj - ratio_k c_kj)
// (the factor lr is applied to the totals)
template <int M, int NQP>
__device__ __forceinline__ void mo_dual_terms_h(const mo_g& G, const double (&w)[M], const double (&coef)[M],
                                                double tail_sum, double lr, const double (&Jc)[M], double yj, int64_t j,
                                                double (&acc)[NQP]) {
    double wJ = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) wJ += w[i] * Jc[i];
    const double v = yj - lr * wJ;
    double P, c[M];
    const double p = mo_prox_ad<M>(G, coef, tail_sum, v, j, P, c);
    double a[M], b[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const double e = p - G.shift[i];
        acc[i] += fabs(e);
        const double sg = e > 0.0 ? 1.0 : (e < 0.0 ? -1.0 : 0.0);
        // (the Hessian only steers the Newton model: its sums take fused multiply-adds; the 2m + 2 sums of the
        //  reference's formulas below keep their NumPy rounding)
        a[i] = G.has_l1 ? __builtin_fma(G.ratio[i], sg, Jc[i]) : Jc[i];
        b[i] = G.has_l1 ? __builtin_fma(-G.ratio[i], c[i], P * Jc[i]) : P * Jc[i];
    }
    const double dv = p - v;
    acc[M] += dv * dv;
    acc[M + 1] += wJ * wJ;
    const double dy = p - yj;
#pragma unroll
    for (int i = 0; i < M; ++i) acc[M + 2 + i] += Jc[i] * dy;
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int k = 0; k < M; ++k) acc[2 * M + 2 + i * M + k] = __builtin_fma(a[i], b[k], acc[2 * M + 2 + i * M + k]);
}

// the Hessian sums alone, over J and y in global memory (zf_mo_dual_hessian: the check of the formulas
// above against differences of the gradient; the search itself takes them inside k_dual_solve)
template <int M>
__global__ __launch_bounds__(ZF_BLOCK) void k_dual_hessian(const double* __restrict__ J, const double* __restrict__ y,
                                                           mo_g G, mo_w W, int64_t n, double* partials) {
    constexpr int NQP = 2 * M + 2 + M * M;
    __shared__ double lds[ZF_WAVES * M * M];
    double acc[NQP], w[M], coef[M];
#pragma unroll
    for (int k = 0; k < NQP; ++k) acc[k] = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        w[i] = W.w[i];
        coef[i] = W.coef[i];
    }
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; j < n; j += stride) {
        double Jc[M];
#pragma unroll
        for (int i = 0; i < M; ++i) Jc[i] = J[(int64_t)i * n + j];
        mo_dual_terms_h<M, NQP>(G, w, coef, W.tail_sum, W.lr, Jc, y[j], j, acc);
    }
    double hs[M * M];
#pragma unroll
    for (int q = 0; q < M * M; ++q) hs[q] = acc[2 * M + 2 + q];
    const double maxs[1] = {0.0};
    double out = 0.0;
    zf_block_reduce<M * M, 0, ZF_WAVES>(hs, maxs, lds, out);
    if (threadIdx.x < M * M) partials[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = out;
}

// Grid-wide hand-over of `count` doubles per workgroup WITHOUT atomics, flags or fences: every value
// travels as a self-validating 16-byte record {bits(v), bits(v) ^ key}, key unique per launch and
// batch.  Workgroup 0 polls the records of all workgroups (a torn or stale record fails the check and
// is simply read again), adds them in workgroup-index order (sums; quantity max_index, if >= 0, a
// maximum: deterministic, no float atomics) and publishes the totals the same way; every workgroup
// polls the totals, each thread its own.  Two dependent memory round trips (records visible to workgroup 0,
// totals visible to all) instead of the six of publish / ticket / gather / publish / flag / read.
// Records of batch e + 2 reuse the slots of batch e, which nobody can still need: a workgroup
// publishes e + 2 only after it has read the totals of e + 1, which exist only after every workgroup
// published e + 1, i.e. after every workgroup read the totals of e.
// On return lds_tot[0..count) holds the grid totals in every workgroup; false = the wait timed out.
__device__ __forceinline__ mo_u64 mo_key(unsigned nonce, unsigned epoch) {
    return ((mo_u64)nonce * 0x9E3779B97F4A7C15ull) ^ ((mo_u64)(epoch + 1u) * 0xC2B2AE3D27D4EB4Full) ^ 0x5851F42D4C957F2Dull;
}
__device__ __forceinline__ void mo_put(mo_u64* rec, double v, mo_u64 key) {
    const mo_u64 lo = (mo_u64)__double_as_longlong(v);
    __hip_atomic_store(rec, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(rec + 1, lo ^ key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool mo_get(const mo_u64* rec, mo_u64 key, double* v) {
    const mo_u64 lo = __hip_atomic_load(rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const mo_u64 hi = __hip_atomic_load(rec + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *v = __longlong_as_double((long long)lo);
    return (lo ^ hi) == key;
}

// QW: quantities per wave of the reducer = ceil(largest count / waves)
// LAST: the hand-over whose totals only workgroup 0 needs (the trial's result record): every other workgroup leaves behind
// its records, workgroup 0 keeps the totals it has just added up - one trip to memory instead of two at the end of the kernel
template <int QW, bool LAST = false>
__device__ __forceinline__ bool mo_grid_combine(const double* my_vals /* lds, count */, int count, int max_index,
                                                mo_u64* partials /* [count][G] records */, mo_u64* totals /* [count + 1] */,
                                                unsigned nonce, unsigned epoch, double* lds_tot, int* lds_flag,
                                                unsigned spin_limit) {
    const int G = (int)gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const mo_u64 key = mo_key(nonce, epoch);
    if (tid < count) mo_put(partials + 2 * ((int64_t)tid * G + blockIdx.x), my_vals[tid], key);
    if (tid == 0) *lds_flag = 1;
    __syncthreads();
    if (blockIdx.x == 0) {
        // the reducer: wave w takes quantities w, w + WAVES, ...; lane l takes workgroups l, l + 64, ...
        constexpr int GL = 4;   // workgroups per lane (<= 256 workgroups)
        double pv[QW][GL];
        // (one 4-bit group per quantity of this wave: a single 32-bit mask held QW x GL <= 32 records - enough for
        //  m <= 4, silently wrong from m = 5 on, where a batch hands over 78 .. 166 values)
        unsigned pend[QW];
        unsigned pending = 0;
#pragma unroll
        for (int a = 0; a < QW; ++a) {
            pend[a] = 0;
#pragma unroll
            for (int c = 0; c < GL; ++c) {
                pv[a][c] = 0.0;
                if (wave + a * MO_SOLVE_WAVES < count && lane + 64 * c < G) pend[a] |= 1u << c;
            }
            pending |= pend[a];
        }
        unsigned spins = 0;
        while (pending) {
            // all records of this lane in flight at once, then the checks: loading one record, checking it
            // and only then loading the next made a poll QW x GL dependent round trips to memory
            mo_u64 rlo[QW][GL], rhi[QW][GL];
#pragma unroll
            for (int a = 0; a < QW; ++a)
#pragma unroll
                for (int c = 0; c < GL; ++c) {
                    int q = wave + a * MO_SOLVE_WAVES, g = lane + 64 * c;
                    q = q < count ? q : 0;   // (out-of-range slots re-read a valid record; their pending bit is clear)
                    g = g < G ? g : 0;
                    const mo_u64* r = partials + 2 * ((int64_t)q * G + g);
                    rlo[a][c] = __hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    rhi[a][c] = __hip_atomic_load(r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            pending = 0;
#pragma unroll
            for (int a = 0; a < QW; ++a) {
#pragma unroll
                for (int c = 0; c < GL; ++c)
                    if ((pend[a] & (1u << c)) && (rlo[a][c] ^ rhi[a][c]) == key) {
                        pv[a][c] = __longlong_as_double((long long)rlo[a][c]);
                        pend[a] &= ~(1u << c);
                    }
                pending |= pend[a];
            }
            if (pending) {
                __builtin_amdgcn_s_sleep(2);   // (do not hammer the memory system while the others still compute)
                if (++spins > spin_limit) {
                    *lds_flag = 0;
                    break;
                }
            }
        }
        // Every wave of the reducer must have all of its records before ANY total goes out (sums over records that never
        // arrived are not totals): the give-up flag settles at the barrier.  Then the totals - or, in slot 0, an ABORT
        // record (the complemented key): every workgroup leaves the search at once instead of spinning to its own limit.
        __syncthreads();
        const bool all_ok = *lds_flag != 0;
#pragma unroll
        for (int a = 0; a < QW; ++a) {
            const int q = wave + a * MO_SOLVE_WAVES;
            const bool is_max = (q == max_index);
            double v = pv[a][0];
#pragma unroll
            for (int c = 1; c < GL; ++c) v = is_max ? fmax(v, pv[a][c]) : v + pv[a][c];
            v = is_max ? zf_wave_max(v) : zf_wave_sum(v);
            if (LAST) {
                if (lane == 0 && q < count) lds_tot[q] = v;
            } else if (all_ok && lane == 0 && q < count) {
                mo_put(totals + 2 * q, v, key);
            }
        }
        if (!LAST && !all_ok && tid == 0) mo_put(totals, 0.0, ~key);
    }
    if constexpr (LAST) {
        __syncthreads();
        return *lds_flag != 0;
    }
    // Every workgroup: thread t < count polls ITS total - self-validating like the records, so no "totals are out" word in
    // front of them and no second trip to memory behind it (round 5: one dependent trip fewer per hand-over, and the
    // reducer no longer waits for its stores).  Slot 0 is read along with it: the reducer's ABORT.
    if (tid < count) {
        unsigned spins = 0;
        double v = 0.0;
        for (;;) {
            const mo_u64 lo = __hip_atomic_load(totals + 2 * tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const mo_u64 hi = __hip_atomic_load(totals + 2 * tid + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const mo_u64 lo0 = __hip_atomic_load(totals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const mo_u64 hi0 = __hip_atomic_load(totals + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((lo ^ hi) == key) {
                v = __longlong_as_double((long long)lo);
                break;
            }
            if ((lo0 ^ hi0) == ~key || ++spins > spin_limit) {
                *lds_flag = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        lds_tot[tid] = v;
    }
    __syncthreads();
    return *lds_flag != 0;
}

// One step of the solver's state machine, by the lanes of ONE wave on identical register copies:
// LDS -> registers -> advance -> LDS (constant indices throughout: ~80 registers, no scratch, no
// dependent LDS round trips in the dense helpers; the lanes differ only inside the simplex QP).
// (A first version kept the resident elements in 64 registers per thread: together with this step
// the kernel needed > 256 VGPRs and spilled; the elements now live in LDS.)
template <int M>
__device__ __forceinline__ void mo_machine_step(zf_dual::machine<M, true>* s_mach, const double* s_fun,
                                                const double (*s_jac)[M], const double* s_hess) {
    constexpr int NB = zf_dual::machine<M, true>::NB;
    zf_dual::machine<M, true> mach = *s_mach;
    // (the values of the batch are read where they are used, at constant indices, straight from LDS:
    //  a register copy of all NB x (M + 1) of them beside the machine spilled)
    mach.advance(*reinterpret_cast<const double(*)[NB]>(s_fun), *reinterpret_cast<const double(*)[NB][M]>(s_jac), s_hess);
    *s_mach = mach;
}

template <int M>
__global__ __launch_bounds__(MO_SOLVE_TPB) void k_dual_solve(mo_solve_args A) {
    constexpr int NQ = 2 * M + 2;
    extern __shared__ double s_data[];   // [(M + 1) x ER x MO_SOLVE_TPB]: J rows, then y
    // The solver's state machine rests in LDS between batches; thread 0 advances it on a register
    // copy (constant indices throughout: no scratch, no dependent LDS round trips inside the
    // dense helpers) - its ~100 registers are then live only inside that step, not across the
    // evaluation loops that hold the resident elements.
    // the machine in its exact-Hessian mode (m >= 3): every point comes with the Hessian of the dual there,
    // accumulated in the same pass as the gradient (mo_dual_terms_h) - no curvature probes, a Newton
    // iteration is ONE batch.  NQP sums per point: the 2m + 2 of the reference's formulas + m x m (+ padding)
    using mach_t = zf_dual::machine<M, true>;
    constexpr int NB = mach_t::NB;
    constexpr bool XH = mach_t::XH;
    constexpr int NQP = (NQ + (XH ? M * M : 0) + 1) & ~1;
    constexpr int REC_CAP = MO_REC_CAP;   // records per parity (the host allocates 2 x this)
    static_assert(NB * NQP <= REC_CAP, "hand-over buffers too small");
    __shared__ mach_t s_mach;
    __shared__ double s_hess[NB * M * M];
    __shared__ double s_w[NB][M], s_coef[NB][M], s_tail[NB];
    __shared__ double s_red[MO_SOLVE_WAVES * NB * NQP];
    __shared__ double s_mine[NB * NQP + 2], s_tot[NB * NQP + 2];
    __shared__ double s_fun[NB], s_jac[NB][M];
    __shared__ int s_flag;
    __shared__ double s_fy[M];   // f(y): given (host value / zf_mo_prepare_async) or formed by the prologue below
    __shared__ double s_Fold[M];
    __shared__ double s_prep[4];   // m = 2: the sums of f(y) waiting for the first hand-over of the search
    int prep_merge = 0;
    const int tid = threadIdx.x;
    if (A.gate && *A.gate != 1) {   // launched ahead of a trial that was then not accepted: nothing may be touched
        if (blockIdx.x == 0 && tid == 0) {
            if (A.accept_out) *A.accept_out = 0;
            A.out->skipped = 1;
            A.out->accepted = 0;
            A.out->ok = 0;
            __threadfence_system();
            __hip_atomic_store(&A.out->seq, (unsigned long long)A.nonce, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    if (A.force_timeout) {   // test hook (zf_mo_debug_force_timeout): the record of a launch whose grid-wide wait gave up
        if (blockIdx.x == 0 && tid == 0) {
            if (A.accept_out) *A.accept_out = 0;
            A.out->skipped = 0;
            A.out->accepted = 0;
            A.out->ok = -1;
            __threadfence_system();
            __hip_atomic_store(&A.out->seq, (unsigned long long)A.nonce, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    if (tid < M) s_Fold[tid] = A.F_old_dev ? A.F_old_dev[tid] : A.F_old[tid];
    const int64_t n = A.n;
    const int64_t stride = (int64_t)gridDim.x * MO_SOLVE_TPB;
    const int64_t j0 = (int64_t)blockIdx.x * MO_SOLVE_TPB + tid;

    // this workgroup's first ER rows of elements stay in LDS for the whole search; element (e, tid)
    // is j = j0 + e * stride and sits at [row e][tid] of every plane: conflict-free
    const int ER = A.resident_rows;
    unsigned epoch = 0;
    int timed_out = 0;
    const int64_t c_begin = clock64();
    if (A.prep_kind == 0) {
        for (int e = 0; e < ER; ++e) {
            const int64_t j = j0 + e * stride;
            s_data[((int64_t)M * ER + e) * MO_SOLVE_TPB + tid] = j < n ? A.y[j] : 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) s_data[((int64_t)i * ER + e) * MO_SOLVE_TPB + tid] = j < n ? A.J[(int64_t)i * n + j] : 0.0;
        }
        if (tid < M) s_fy[tid] = A.f_y_dev ? A.f_y_dev[tid] : A.f_y[tid];
    } else {
        // Fused outer iteration: this launch forms its own inputs.  Pass 1 - y (the deferred commit of the
        // previous accepted trial, k_commit's expression) and the raw sums of f(y) (k_jos1_sums /
        // k_fds_sums' terms); one grid-wide hand-over; pass 2 - the rows of J = jac_f(y) (k_jos1_jac /
        // k_fds_jac's expressions; FDS needs sum(y) first) straight into LDS.  y, J and f(y) are also
        // stored, so the other entry points find what zf_mo_prepare_async + zf_mo_commit would have left.
        const double dn = (double)n;
        constexpr int NP = (M == 2) ? 2 : 4;
        double ps[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) ps[q] = 0.0;
        auto pass1 = [&](int64_t j) -> double {
            double yj;
            if (A.make_y) {
                const double xv = A.xk[j];
                yj = A.nesterov ? xv + A.beta * (xv - A.xo[j]) : xv;
                A.y_w[j] = yj;
            } else {
                yj = A.y[j];
            }
            if constexpr (M == 2) {            // JOS1
                const double t = yj - 2;
                ps[0] += yj * yj;
                ps[1] += t * t;
            } else {                           // FDS (x unsharded here: global index = j)
                const double idx = (double)(j + 1);
                const double conv = (double)((j + 1) * (n - j));
                const double t = yj - idx, t2 = t * t;
                ps[0] += idx * (t2 * t2);
                ps[1] += yj;
                ps[2] += yj * yj;
                ps[3] += conv * exp(-yj);
            }
            return yj;
        };
        for (int e = 0; e < ER; ++e) {
            const int64_t j = j0 + e * stride;
            s_data[((int64_t)M * ER + e) * MO_SOLVE_TPB + tid] = j < n ? pass1(j) : 0.0;
        }
        for (int64_t j = j0 + ER * stride; j < n; j += stride) (void)pass1(j);
        {
            const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const double v = zf_wave_sum(ps[q]);
                if (lane == 0) s_red[wave * NP + q] = v;
            }
            __syncthreads();
            if (tid < NP) {
                double v = s_red[tid];
#pragma unroll
                for (int wv = 1; wv < MO_SOLVE_WAVES; ++wv) v += s_red[wv * NP + tid];
                s_mine[tid] = v;
            }
            __syncthreads();
        }
        if constexpr (M == 2) {
            // JOS1: the rows of J need none of the sums (2 y / n, 2 (y - 2) / n) - the two sums of f(y) ride
            // along with the first batch of the search instead of a hand-over of their own
            if (tid < NP) s_prep[tid] = s_mine[tid];
            prep_merge = 1;
        } else {
            if (!mo_grid_combine<1>(s_mine, NP, -1, A.partials, A.totals, A.nonce, epoch, s_tot, &s_flag, A.spin_limit)) timed_out = 1;
            epoch += 1;
        }
        if (M != 2 && tid == 0) {
            double t4[4] = {0.0, 0.0, 0.0, 0.0}, fy[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < NP; ++q) t4[q] = s_tot[q];
            mo_f_from_sums(M == 2 ? ZF_MO_JOS1 : ZF_MO_FDS, dn, t4, fy);
#pragma unroll
            for (int i = 0; i < M; ++i) {
                s_fy[i] = fy[i];
                if (blockIdx.x == 0) A.f_y_w[i] = fy[i];
            }
        }
        const double e_mean = (M == 2) ? 0.0 : exp(s_tot[1] / dn);
        const double c1 = 4 / (dn * dn), den = dn * (dn + 1);
        auto pass2 = [&](double yj, int64_t j, double (&Jc)[M]) {
            if constexpr (M == 2) {
                Jc[0] = 2 * yj / dn;
                Jc[1] = 2 * (yj - 2) / dn;
            } else {
                const double idx = (double)(j + 1);
                const double conv = (double)((j + 1) * (n - j));
                const double t = yj - idx;
                Jc[0] = c1 * idx * (t * t * t);
                Jc[1] = e_mean / dn + 2 * yj;
                Jc[M - 1] = -conv * exp(-yj) / den;
            }
#pragma unroll
            for (int i = 0; i < M; ++i) A.J_w[(int64_t)i * n + j] = Jc[i];
        };
        for (int e = 0; e < ER; ++e) {
            const int64_t j = j0 + e * stride;
            double Jc[M];
#pragma unroll
            for (int i = 0; i < M; ++i) Jc[i] = 0.0;
            if (j < n) pass2(s_data[((int64_t)M * ER + e) * MO_SOLVE_TPB + tid], j, Jc);
#pragma unroll
            for (int i = 0; i < M; ++i) s_data[((int64_t)i * ER + e) * MO_SOLVE_TPB + tid] = Jc[i];
        }
        for (int64_t j = j0 + ER * stride; j < n; j += stride) {   // (streamed part: read back from y, J below)
            double Jc[M];
            pass2(A.make_y ? A.y_w[j] : A.y[j], j, Jc);
        }
    }
    if (tid == 0) {
        mach_t mach;
        double w_start[M];
#pragma unroll
        for (int i = 0; i < M; ++i) w_start[i] = A.w0_dev ? A.w0_dev[i] : A.w0[i];
        mach.start((A.has_w0 || A.w0_dev) ? w_start : nullptr, A.tol, (long)A.max_iter);
        s_mach = mach;
    }
    __syncthreads();

    int64_t evals = 0, batches = 0;
    int64_t c_eval = 0, c_comb = 0, c_step = 0;
    while (!timed_out && !s_mach.done()) {
        const int64_t c0 = clock64();
        const int npts = s_mach.npts;
        if (tid < npts) {   // weights of point `tid` (proximal_gradient.py:164, problems.py:127)
            double tail = 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const double wi = s_mach.pts[tid][i];
                s_w[tid][i] = wi;
                s_coef[tid][i] = A.G.has_l1 ? (A.lr * wi) * A.G.ratio[i] : 0.0;
            }
#pragma unroll
            for (int i = 1; i < M; ++i) tail += s_coef[tid][i];
            s_tail[tid] = tail;
        }
        __syncthreads();
        {
            // the points of the batch in groups of GP: one pass over the workgroup's elements per group
            // (an element is read from LDS once per group; GP x NQ running sums per thread - all NB points
            // at once held 120 VGPRs of sums and weights, more than the kernel can spare), the sums of a
            // group with ONE transposing butterfly per wave (zf_wave_reduce_multi: the pairing of a
            // shuffle tree per quantity), then the wave totals in wave order
            constexpr int GP = XH ? 1 : 2;   // (with the m x m Hessian sums a point is 18 sums: one point per pass)
            constexpr int NV = NB * NQP;
            constexpr int GV = GP * NQP;
            constexpr int H = (GV % 8 == 0) ? 3 : (GV % 4 == 0) ? 2 : (GV % 2 == 0) ? 1 : 0;
            const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
            for (int kb = 0; kb < NB; kb += GP) {
                if (kb < npts) {
                    double w[GP][M], coef[GP][M], tail[GP], acc[GV];
#pragma unroll
                    for (int k = 0; k < GP; ++k) {
                        const int kk = (kb + k < NB) ? kb + k : NB - 1;
#pragma unroll
                        for (int i = 0; i < M; ++i) {
                            w[k][i] = s_w[kk][i];
                            coef[k][i] = s_coef[kk][i];
                        }
                        tail[k] = s_tail[kk];
                    }
#pragma unroll
                    for (int q = 0; q < GV; ++q) acc[q] = 0.0;
                    auto element = [&](const double (&Jc)[M], double yj, int64_t j) {
#pragma unroll
                        for (int k = 0; k < GP; ++k) {
                            if (kb + k < npts) {
                                double a8[NQP];
#pragma unroll
                                for (int q = 0; q < NQP; ++q) a8[q] = acc[k * NQP + q];
                                // the Hessian sums only for point 0 of a batch: the machine uses the Hessian of a
                                // line-search point only if the FIRST step length is accepted (a shorter step is
                                // re-evaluated, zf_dual::machine::advance_pick), and they cost 1.4 x the rest
                                if (XH && kb + k == 0) {
                                    mo_dual_terms_h<M, NQP>(A.G, w[k], coef[k], tail[k], A.lr, Jc, yj, j, a8);
                                } else {
                                    double a0[NQ];
#pragma unroll
                                    for (int q = 0; q < NQ; ++q) a0[q] = a8[q];
                                    mo_dual_terms<M>(A.G, w[k], coef[k], tail[k], A.lr, Jc, yj, j, a0);
#pragma unroll
                                    for (int q = 0; q < NQ; ++q) a8[q] = a0[q];
                                }
#pragma unroll
                                for (int q = 0; q < NQP; ++q) acc[k * NQP + q] = a8[q];
                            }
                        }
                    };
                    for (int e = 0; e < ER; ++e) {
                        const int64_t j = j0 + e * stride;
                        if (j >= n) break;
                        double Jc[M];
#pragma unroll
                        for (int i = 0; i < M; ++i) Jc[i] = s_data[((int64_t)i * ER + e) * MO_SOLVE_TPB + tid];
                        element(Jc, s_data[((int64_t)M * ER + e) * MO_SOLVE_TPB + tid], j);
                    }
                    for (int64_t j = j0 + ER * stride; j < n; j += stride) {   // beyond the LDS-resident part
                        double Jc[M];
#pragma unroll
                        for (int i = 0; i < M; ++i) Jc[i] = A.J[(int64_t)i * n + j];
                        element(Jc, A.y[j], j);
                    }
                    zf_wave_reduce_multi<GV, H, false>(acc, lane);
                    if ((lane & ((64 >> H) - 1)) == 0) {
#pragma unroll
                        for (int q = 0; q < (GV >> H); ++q) {
                            const int idx = kb * NQP + zf_wave_reduce_multi_index<GV, H>(q, lane);
                            if (idx < NV) s_red[wave * NV + idx] = acc[q];
                        }
                    }
                }
            }
            __syncthreads();
            if (tid < NV) {
                double v = s_red[tid];
#pragma unroll
                for (int wv = 1; wv < MO_SOLVE_WAVES; ++wv) v += s_red[wv * NV + tid];
                s_mine[tid] = v;
            }
            __syncthreads();
        }
        const int parity = (int)(epoch & 1u);
        const int extra = (prep_merge && batches == 0) ? 2 : 0;
        const int cnt = npts * NQP + extra;
        if (extra) {
            if (tid < 2) s_mine[npts * NQP + tid] = s_prep[tid];
            __syncthreads();
        }
        const int64_t c1 = clock64();
        c_eval += c1 - c0;
        if (!mo_grid_combine<(NB * NQP + 2 + MO_SOLVE_WAVES - 1) / MO_SOLVE_WAVES>(s_mine, cnt, -1, A.partials + 2 * ((int64_t)parity * REC_CAP * gridDim.x),
                             A.totals + 2 * (parity * REC_CAP), A.nonce, epoch, s_tot, &s_flag, A.spin_limit)) {
            timed_out = 1;
            break;
        }
        epoch += 1;
        if (extra) {   // f(y) of JOS1 from the two sums that came with this hand-over (problems.py:193-197)
            if (tid == 0) {
                const double t4[4] = {s_tot[npts * NQP], s_tot[npts * NQP + 1], 0.0, 0.0};
                double fy[3] = {0.0, 0.0, 0.0};
                mo_f_from_sums(ZF_MO_JOS1, (double)n, t4, fy);
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    s_fy[i] = fy[i < 3 ? i : 0];
                    if (blockIdx.x == 0) A.f_y_w[i] = fy[i < 3 ? i : 0];
                }
            }
            __syncthreads();
        }
        batches += 1;
        evals += npts;
        const int64_t c2 = clock64();
        c_comb += c2 - c1;
        if (tid < npts) {   // D(w), grad D(w) of point `tid` from the grid totals (:165-177)
            const double* t = s_tot + tid * NQP;
            double g_p[M], inner = 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                g_p[i] = A.G.has_l1 ? A.G.ratio[i] * t[i] : 0.0;
                inner += s_w[tid][i] * g_p[i];
            }
            const double n_pv = sqrt(t[M]), n_wJ = sqrt(t[M + 1]);
            double f = -inner - n_pv * n_pv / 2 / A.lr + A.lr / 2 * (n_wJ * n_wJ);
            double corr = 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                double jv = -g_p[i] - t[M + 2 + i];
                if (!A.deprecated) {
                    const double dF = s_Fold[i] - s_fy[i];
                    corr += s_w[tid][i] * dF;
                    jv += dF;
                }
                s_jac[tid][i] = jv;
            }
            if (!A.deprecated) f += corr;
            s_fun[tid] = f;
            if constexpr (XH) {   // the Hessian of the dual at point `tid` (symmetrised by the machine)
#pragma unroll
                for (int q = 0; q < M * M; ++q) s_hess[tid * M * M + q] = A.lr * t[NQ + q];
            }
        }
        __syncthreads();
        // wave 0, all 64 lanes on identical copies of the state (same LDS words in, same words out):
        // the support enumeration of the simplex QP runs lane-parallel (zf_dual::machine::simplex_qp)
        if (tid < 64) mo_machine_step<M>(&s_mach, s_fun, s_jac, s_hess);   // (inlined: see above)
        __syncthreads();
        c_step += clock64() - c2;
    }

    // primal recovery with the dual solution (:206), max|x+ - y| (:510) and - same pass - the sums
    // of F(x+) = f(x+) + g(x+) (:295; problems.py:101-117,193-205,312-328): the trial's acceptance
    // test then needs nothing but this kernel's result record
    constexpr int NT = 1 + 4 + M + 1;   // [0] max|x+ - y|  [1..5) f sums  [5..5+M) sum|x+ - s_i|  [5+M] violations
    double tl[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) tl[q] = 0.0;
    const bool solved = !timed_out && s_mach.ok;
    const int tail_kind = A.tail_kind;
    if (solved) {
        double w[M], coef[M], tail = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            w[i] = s_mach.w[i];
            coef[i] = A.G.has_l1 ? (A.lr * w[i]) * A.G.ratio[i] : 0.0;
        }
#pragma unroll
        for (int i = 1; i < M; ++i) tail += coef[i];
        auto one = [&](const double (&Jc)[M], double yj, int64_t j) {
            double wJ = 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) wJ += w[i] * Jc[i];
            const double p = mo_prox_t<M>(A.G, coef, tail, yj - A.lr * wJ, j);
            A.xn[j] = p;
            tl[0] = fmax(tl[0], fabs(p - yj));
            // (terms first, then unconditional accumulation at constant indices: accumulating inside
            //  the branches made the compiler merge them into a dynamically indexed stack slot)
            double f1 = 0.0, f2 = 0.0, f3 = 0.0, f4 = 0.0;
            if (tail_kind == ZF_MO_JOS1) {           // k_jos1_sums
                const double t = p - 2;
                f1 = p * p;
                f2 = t * t;
            } else if (tail_kind == ZF_MO_FDS) {     // k_fds_sums (x unsharded here: global index = j)
                const double idx = (double)(j + 1);
                const double conv = (double)((j + 1) * (n - j));
                const double t = p - idx, t2 = t * t;
                f1 = idx * (t2 * t2);
                f2 = p;
                f3 = p * p;
                f4 = conv * exp(-p);
            }
            tl[1] += f1;
            tl[2] += f2;
            tl[3] += f3;
            tl[4] += f4;
#pragma unroll
            for (int i = 0; i < M; ++i) tl[5 + i] += fabs(p - A.G.shift[i]);   // k_g_terms
            if (A.G.has_box)
                tl[5 + M] += (p < (A.G.lo_v ? A.G.lo_v[j] : A.G.lo) || p > (A.G.hi_v ? A.G.hi_v[j] : A.G.hi)) ? 1.0 : 0.0;
        };
        for (int e = 0; e < ER; ++e) {
            const int64_t j = j0 + e * stride;
            if (j >= n) break;
            double Jc[M];
#pragma unroll
            for (int i = 0; i < M; ++i) Jc[i] = s_data[((int64_t)i * ER + e) * MO_SOLVE_TPB + tid];
            one(Jc, s_data[((int64_t)M * ER + e) * MO_SOLVE_TPB + tid], j);
        }
        for (int64_t j = j0 + ER * stride; j < n; j += stride) {
            double Jc[M];
#pragma unroll
            for (int i = 0; i < M; ++i) Jc[i] = A.J[(int64_t)i * n + j];
            one(Jc, A.y[j], j);
        }
    }
    if (solved) {
        // workgroup totals: wave butterflies, then the wave values in wave order ([0] is a maximum)
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const double v = (q == 0) ? zf_wave_max(tl[q]) : zf_wave_sum(tl[q]);
            if (lane == 0) s_red[wave * NT + q] = v;
        }
        __syncthreads();
        if (tid < NT) {
            double v = s_red[tid];
#pragma unroll
            for (int wv = 1; wv < MO_SOLVE_WAVES; ++wv) v = (tid == 0) ? fmax(v, s_red[wv * NT + tid]) : v + s_red[wv * NT + tid];
            s_mine[tid] = v;
        }
        __syncthreads();
        const int parity = (int)(epoch & 1u);
        if (!mo_grid_combine<(NT + MO_SOLVE_WAVES - 1) / MO_SOLVE_WAVES, true>(s_mine, NT, 0, A.partials + 2 * ((int64_t)parity * REC_CAP * gridDim.x),
                             A.totals + 2 * (parity * REC_CAP), A.nonce, epoch, s_tot, &s_flag, A.spin_limit))
            timed_out = 1;
    }
    if (blockIdx.x == 0 && tid == 0) {
        mo_solve_result r;
#pragma unroll
        for (int i = 0; i < MO_MAX_M; ++i) r.w[i] = (i < M) ? s_mach.w[i < M ? i : 0] : 0.0;
        r.fun = s_mach.fun;
        r.err = (solved && !timed_out) ? s_tot[0] : 0.0;
        r.has_F = (solved && !timed_out) ? 1 : 0;
        r.reserved2 = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) r.f_sums[q] = r.has_F ? s_tot[1 + q] : 0.0;
#pragma unroll
        for (int q = 0; q < MO_MAX_M + 1; ++q) r.g_sums[q] = (q <= M && r.has_F) ? s_tot[5 + (q <= M ? q : 0)] : 0.0;
#pragma unroll
        for (int i = 0; i < MO_MAX_M; ++i) r.f_y[i] = (i < M) ? s_fy[i < M ? i : 0] : 0.0;
        r.nit = s_mach.nit;
        r.evals = evals;
        r.batches = batches;
        r.ok = timed_out ? -1 : s_mach.ok;
        r.reserved = 0;
        r.cyc_total = clock64() - c_begin;
        r.cyc_eval = c_eval;
        r.cyc_combine = c_comb;
        r.cyc_step = c_step;
        // F(x+) and the acceptance test (:295, :298-303) from the sums of this same kernel
        int accepted = 0;
#pragma unroll
        for (int i = 0; i < MO_MAX_M; ++i) r.f_x[i] = r.g_x[i] = 0.0;
        if (r.has_F && tail_kind != 0 && r.ok == 1) {
            double fx[3] = {0.0, 0.0, 0.0};
            const double t4[4] = {s_tot[1], s_tot[2], s_tot[3], s_tot[4]};   // (not &r.f_sums: that sends r to scratch)
            mo_f_from_sums(M == 2 ? ZF_MO_JOS1 : ZF_MO_FDS, (double)n, t4, fx);   // (tail_kind != 0: JOS1 has m = 2, FDS m = 3)
            accepted = 1;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const double gx = (A.G.has_box && s_tot[5 + M] > 0.0) ? INFINITY : (A.G.has_l1 ? A.G.ratio[i] * s_tot[5 + i] : 0.0);
                r.f_x[i] = fx[i];
                r.g_x[i] = gx;
                const double F_new = fx[i] + gx;
                if (A.F_new_out) A.F_new_out[i] = F_new;
                if (A.w_new_out) A.w_new_out[i] = s_mach.w[i];
                const double lhs = A.deprecated ? fx[i] - s_fy[i] : F_new - s_Fold[i];
                if (!(lhs <= -s_mach.fun + A.accept_tol)) accepted = 0;   // (fun of the trial = - dual value, :207)
            }
            if (A.decay_is_one) accepted = 1;
        }
        r.accepted = accepted;
        r.skipped = 0;
        if (A.accept_out) *A.accept_out = accepted;
        r.seq = 0;
        *A.out = r;
        __threadfence_system();
        __hip_atomic_store(&A.out->seq, (unsigned long long)A.nonce, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace

// ---------------------------------------------------------------------------
struct zf_mo {
    int kind, m;
    int64_t n;
    mo_g G;
    hipStream_t stream;
    double* buf = nullptr;     // 3 x-buffers + y + J
    double* xb[3] = {nullptr, nullptr, nullptr};
    int cur = 0;               // xb[cur] = x_k, xb[(cur+2)%3] = x_{k-1}, xb[(cur+1)%3] = x+
    double* y = nullptr;
    double* J = nullptr;
    double* partials = nullptr;
    double* totals = nullptr;  // device, 32
    double* h_totals = nullptr;  // pinned host mirror of totals (DMA target, no staging copy)
    int grid = 1;
    double f_y[MO_MAX_M];
    // x sharded over ranks (contiguous blocks): n is the local length
    int64_t n_global = 0;      // n_features of the whole problem
    int64_t offset = 0;        // global index of local element 0
    double* bounds_v = nullptr;             // 2 n: per-coordinate lower, upper bounds (optional)
    zf_mo_exchange_fn exchange = nullptr;   // combines raw totals over the ranks, in place
    void* exchange_ctx = nullptr;
    // ... or, instead of the callback: a communicator of the library (zf_mo_set_comm) - the totals of every
    // reduction are all-gathered on the stream and added in rank order on the device; the library's own dual
    // search then exchanges once per BATCH of its state machine, not once per evaluation
    zf_comm* comm = nullptr;
    int comm_world = 1;
    double* cm_partials = nullptr;   // MO_CM_CAP x grid: block partials of every point of a batch
    double* cm_totals = nullptr;     // MO_CM_CAP: this rank's totals of the batch
    double* cm_gathered = nullptr;   // world x MO_CM_CAP, rank-major
    double* h_cm = nullptr;          // pinned host mirror of the combined totals
    // the same search driven FROM THE DEVICE (zf_mo_solve_dual_stream): the state machine lives in device memory and is
    // advanced by a one-wave kernel between stream-ordered all-gathers - no host round trip per batch
    void* ds_state = nullptr;        // device: zf_dual::machine<m> (the largest, m = 8, fits)
    void* h_ds_state = nullptr;      // pinned host mirror (upload of the started machine, download of the finished one)
    double* ds_consts = nullptr;     // device: f_y[MO_MAX_M], F_old[MO_MAX_M] of the trial
    double* h_ds_consts = nullptr;   // pinned staging of the same
    int64_t ds_rounds = 0;           // batches enqueued so far (diagnostics / tests)
    int64_t n_exchanges = 0;         // collectives issued so far (diagnostics / tests)
    // device-side dual search (k_dual_solve): workspace, allocated at first use
    unsigned long long* solve_partials = nullptr;
    unsigned long long* solve_totals = nullptr;
    unsigned solve_nonce = 0;
    mo_solve_result* solve_out = nullptr;      // device view of h_solve_out (two records: launches alternate)
    mo_solve_result* h_solve_out = nullptr;    // pinned host memory the kernel writes directly
    int* accept_dev = nullptr;                 // [2] acceptance decision of the launch of either parity
    double* F_dev = nullptr;                   // [2][MO_MAX_M] F(x+) of the launch of either parity; [2][MO_MAX_M] behind them: its weights
    unsigned last_nonce[2] = {0, 0};           // launch number whose record each slot is waiting for / holds
    int last_slot = 0;                         // slot of the most recent launch (zf_mo_solve_stats)
    int solve_grid = 0;
    size_t solve_lds_set = (size_t)-1;         // dynamic LDS size last registered for k_dual_solve
    int64_t solve_static_lds = -1;             // static LDS of k_dual_solve<m> (hipFuncGetAttributes, once)
    double* f_y_dev = nullptr;                 // f(y) of zf_mo_prepare_async (device, MO_MAX_M)
    bool f_y_on_device = false;
    // fused outer iteration (zf_mo_set_fused): zf_mo_commit and zf_mo_prepare_async only RECORD what is
    // due; the next zf_mo_solve_dual_device forms y, f(y) and J inside its one kernel.  Every other
    // entry point first brings the buffers up to date (mo_flush), so the deferral is invisible.
    bool fused = false;
    bool y_pending = false;      // y = x_k + beta (x_k - x_{k-1}) not formed yet
    bool prep_pending = false;   // f(y), J not formed yet
    double pend_beta = 0.0;
    int pend_nesterov = 0;
    // the grid-wide waits of k_dual_solve need every workgroup resident at once
    bool solve_unavailable = false;   // the occupancy query says the grid cannot be co-resident: never launched
    struct { bool fused, make_y; double beta; int nesterov; } launched[2] = {};   // what each slot's launch was to form
    unsigned spin_limit = MO_SPIN_LIMIT;
    int force_timeouts = 0;           // test hook: this many of the next launches report a timeout
};

struct zf_comm;
extern "C" int zf_comm_all_gather(zf_comm* c, const double* send_dev, double* recv_dev, int64_t count, void* stream);
extern "C" int zf_comm_info(zf_comm* c, int32_t* rank, int32_t* world);
constexpr int MO_CM_CAP = (zf_dual::MAXM + 1) * (2 * MO_MAX_M + 2);   // totals of one batch: <= m + 1 points x (2m + 2)

static int mo_prepare_async_now(zf_mo* s);
// pending work of the fused mode, done the unfused way (k_commit; the four launches of prepare_async)
static int mo_flush(zf_mo* s) {
    if (s->y_pending) {
        s->y_pending = false;
        hipLaunchKernelGGL(k_commit, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, s->y, s->xb[s->cur],
                           s->xb[(s->cur + 2) % 3], s->pend_beta, s->pend_nesterov, s->n);
        ZF_HIP(hipGetLastError());
    }
    if (s->prep_pending) {
        s->prep_pending = false;
        return mo_prepare_async_now(s);
    }
    return ZF_OK;
}


// launch KERNEL<m> for m = 2 .. MO_MAX_M
#define MO_LAUNCH(KERNEL, m, ...)                                                                              \
    do {                                                                                                        \
        switch (m) {                                                                                            \
            case 2: hipLaunchKernelGGL(KERNEL<2>, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, __VA_ARGS__); break; \
            case 3: hipLaunchKernelGGL(KERNEL<3>, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, __VA_ARGS__); break; \
            case 4: hipLaunchKernelGGL(KERNEL<4>, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, __VA_ARGS__); break; \
            case 5: hipLaunchKernelGGL(KERNEL<5>, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, __VA_ARGS__); break; \
            case 6: hipLaunchKernelGGL(KERNEL<6>, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, __VA_ARGS__); break; \
            case 7: hipLaunchKernelGGL(KERNEL<7>, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, __VA_ARGS__); break; \
            default: hipLaunchKernelGGL(KERNEL<8>, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, __VA_ARGS__); break; \
        }                                                                                                       \
    } while (0)

static int mo_reduce_to_host(zf_mo* s, int nq, int max_index, double* host) {
    hipLaunchKernelGGL(k_mo_reduce, dim3(1), dim3(64 * MO_RED_WAVES), 0, s->stream, s->partials, s->grid, nq, max_index,
                       s->totals);
    ZF_HIP(hipGetLastError());
    if (s->comm) {
        // the library's communicator: all-gather on the stream, rank-ordered sum on the device (the totals stay
        // there for the kernels that read them: FDS needs the global sum x), one copy to the host
        if (int rc = zf_comm_all_gather(s->comm, s->totals, s->cm_gathered, nq, s->stream)) return rc;
        s->n_exchanges += 1;
        hipLaunchKernelGGL(k_mo_combine_ranks, dim3(1), dim3(64), 0, s->stream, s->cm_gathered, s->comm_world, nq, max_index,
                           s->totals);
        ZF_HIP(hipGetLastError());
    }
    ZF_HIP(hipMemcpyAsync(s->h_totals, s->totals, sizeof(double) * nq, hipMemcpyDeviceToHost, s->stream));
    ZF_HIP(hipStreamSynchronize(s->stream));
    memcpy(host, s->h_totals, sizeof(double) * nq);
    if (s->exchange) {
        // C3: every rank contributes its raw totals; the callee adds them in rank order
        // (max for quantity max_index), so all ranks continue with bitwise-identical scalars
        if (s->exchange(s->exchange_ctx, host, nq, max_index) != 0)
            return zf_fail(ZF_ERR_STATE, "zf_mo: the exchange callback failed");
        // kernels that read the totals on the device (FDS: sum x) need the global values too
        memcpy(s->h_totals, host, sizeof(double) * nq);
        ZF_HIP(hipMemcpyAsync(s->totals, s->h_totals, sizeof(double) * nq, hipMemcpyHostToDevice, s->stream));
    }
    return ZF_OK;
}

static double* mo_which(zf_mo* s, int which) {
    switch (which) {
        case 0: return s->xb[s->cur];
        case 1: return s->y;
        case 2: return s->xb[(s->cur + 1) % 3];
        case 3: return s->xb[(s->cur + 2) % 3];
        default: return nullptr;
    }
}

static void mo_fill_w(const zf_mo* s, double lr, const double* w, mo_w* W) {
    W->lr = lr;
    W->tail_sum = 0.0;
    for (int i = 0; i < MO_MAX_M; ++i) {
        W->w[i] = i < s->m ? w[i] : 0.0;
        // coef = weight * l1_ratios with weight = lr * w   (proximal_gradient.py:164, problems.py:127)
        W->coef[i] = (i < s->m && s->G.has_l1) ? (lr * w[i]) * s->G.ratio[i] : 0.0;
    }
    for (int i = 1; i < s->m; ++i) W->tail_sum += W->coef[i];   // np.sum(coef[1:])
}

extern "C" int zf_mo_create(zf_mo** out, int32_t kind, int32_t m, int64_t n, const double* l1_ratios,
                            const double* l1_shifts, double box_lo, double box_hi, void* stream) {
    ZF_REQUIRE(out && n >= 1, "zf_mo_create: bad argument");
    ZF_REQUIRE(m >= 2 && m <= MO_MAX_M, "zf_mo_create: 2 <= n_objectives <= 8 supported");
    ZF_REQUIRE(kind == ZF_MO_GENERIC || (kind == ZF_MO_JOS1 && m == 2) || (kind == ZF_MO_FDS && m == 3),
               "zf_mo_create: kind / n_objectives mismatch");
    zf_mo* s = new (std::nothrow) zf_mo();
    if (!s) return zf_fail(ZF_ERR_ARG, "zf_mo_create: out of host memory");
    s->kind = kind;
    s->m = m;
    s->n = n;
    s->n_global = n;
    s->stream = (hipStream_t)stream;
    memset(&s->G, 0, sizeof(s->G));
    s->G.m = m;
    s->G.has_l1 = l1_ratios != nullptr;
    for (int i = 0; i < m; ++i) {
        s->G.ratio[i] = l1_ratios ? l1_ratios[i] : 0.0;
        s->G.shift[i] = l1_shifts ? l1_shifts[i] : 0.0;
    }
    s->G.has_box = !(box_lo == -INFINITY && box_hi == INFINITY);
    s->G.lo = box_lo;
    s->G.hi = box_hi;
    const int64_t n_pad = (n + 63) & ~int64_t(63);
    hipError_t e = hipMalloc(&s->buf, sizeof(double) * n_pad * (4 + m));
    if (e == hipSuccess) e = hipMalloc(&s->partials, sizeof(double) * 32 * MO_GRID_MAX);
    if (e == hipSuccess) e = hipMalloc(&s->totals, sizeof(double) * 32);
    if (e == hipSuccess) e = hipHostMalloc((void**)&s->h_totals, sizeof(double) * 32, hipHostMallocDefault);
    if (e != hipSuccess) {
        if (s->buf) (void)hipFree(s->buf);
        if (s->partials) (void)hipFree(s->partials);
        delete s;
        return zf_fail(ZF_ERR_HIP, "zf_mo_create: %s", hipGetErrorString(e));
    }
    for (int k = 0; k < 3; ++k) s->xb[k] = s->buf + k * n_pad;
    s->y = s->buf + 3 * n_pad;
    s->J = s->buf + 4 * n_pad;   // rows are n apart (not n_pad): J[i*n + j]
    int64_t g = (n + ZF_BLOCK - 1) / ZF_BLOCK;
    s->grid = (int)(g > MO_GRID_MAX ? MO_GRID_MAX : (g < 1 ? 1 : g));
    *out = s;
    return ZF_OK;
}

// x is one contiguous block [offset, offset + n) of a decision vector of n_global entries that is
// split over ranks.  `fn(ctx, vals, count, max_index)` is called once per reduction with this
// rank's raw totals and must replace them by the combination over all ranks (sums added in
// rank order; entry max_index, if >= 0, is a maximum); nonzero return = failure.
extern "C" int zf_mo_set_shard(zf_mo* s, int64_t n_global, int64_t offset, zf_mo_exchange_fn fn, void* ctx) {
    ZF_REQUIRE(s && fn, "zf_mo_set_shard: null argument");
    ZF_REQUIRE(offset >= 0 && offset + s->n <= n_global, "zf_mo_set_shard: block outside the vector");
    s->n_global = n_global;
    s->offset = offset;
    s->exchange = fn;
    s->exchange_ctx = ctx;
    return ZF_OK;
}

// The same sharding with a communicator of the library (zf_comm_create / zf_comm_create_local_group) in place of
// the callback: every reduction all-gathers its raw totals on the stream and adds them in rank order on the
// device (no host code per exchange), and zf_mo_solve_dual exchanges once per BATCH of its search - the start
// point with its m curvature probes, the two step lengths of a line search - instead of once per evaluation.
// The communicator must outlive the engine; rank / world are its own.
extern "C" int zf_mo_set_comm(zf_mo* s, zf_comm* comm, int64_t n_global, int64_t offset) {
    ZF_REQUIRE(s && comm, "zf_mo_set_comm: null argument");
    ZF_REQUIRE(offset >= 0 && offset + s->n <= n_global, "zf_mo_set_comm: block outside the vector");
    ZF_REQUIRE(!s->exchange, "zf_mo_set_comm: an exchange callback is already set (zf_mo_set_shard)");
    int32_t rank = 0, world = 1;
    if (int rc = zf_comm_info(comm, &rank, &world)) return rc;
    if (!s->cm_partials) {
        ZF_HIP(hipMalloc(&s->cm_partials, sizeof(double) * MO_CM_CAP * MO_GRID_MAX));
        ZF_HIP(hipMalloc(&s->cm_totals, sizeof(double) * MO_CM_CAP));
        ZF_HIP(hipHostMalloc((void**)&s->h_cm, sizeof(double) * MO_CM_CAP, hipHostMallocDefault));
    }
    if (s->cm_gathered) (void)hipFree(s->cm_gathered);
    s->cm_gathered = nullptr;
    ZF_HIP(hipMalloc(&s->cm_gathered, sizeof(double) * MO_CM_CAP * world));
    s->comm = comm;
    s->comm_world = world;
    s->n_global = n_global;
    s->offset = offset;
    return ZF_OK;
}
extern "C" int zf_mo_exchange_count(zf_mo* s, int64_t* count) {
    ZF_REQUIRE(s && count, "zf_mo_exchange_count: null argument");
    *count = s->n_exchanges;
    return ZF_OK;
}

// Per-coordinate box (arrays of n; this rank's block when sharded): replaces the scalar bounds.
extern "C" int zf_mo_set_bounds(zf_mo* s, const double* lo_host, const double* hi_host) {
    ZF_REQUIRE(s && lo_host && hi_host, "zf_mo_set_bounds: null argument");
    if (!s->bounds_v) ZF_HIP(hipMalloc(&s->bounds_v, sizeof(double) * 2 * s->n));
    ZF_HIP(hipMemcpyAsync(s->bounds_v, lo_host, sizeof(double) * s->n, hipMemcpyHostToDevice, s->stream));
    ZF_HIP(hipMemcpyAsync(s->bounds_v + s->n, hi_host, sizeof(double) * s->n, hipMemcpyHostToDevice, s->stream));
    ZF_HIP(hipStreamSynchronize(s->stream));
    s->G.has_box = 1;
    s->G.lo_v = s->bounds_v;
    s->G.hi_v = s->bounds_v + s->n;
    return ZF_OK;
}

extern "C" int zf_mo_destroy(zf_mo* s) {
    if (!s) return ZF_OK;
    (void)hipStreamSynchronize(s->stream);
    if (s->bounds_v) (void)hipFree(s->bounds_v);
    if (s->solve_partials) (void)hipFree(s->solve_partials);
    if (s->solve_totals) (void)hipFree(s->solve_totals);
    if (s->f_y_dev) (void)hipFree(s->f_y_dev);
    if (s->h_solve_out) (void)hipHostFree(s->h_solve_out);
    if (s->accept_dev) (void)hipFree(s->accept_dev);
    if (s->F_dev) (void)hipFree(s->F_dev);
    if (s->cm_partials) (void)hipFree(s->cm_partials);
    if (s->cm_totals) (void)hipFree(s->cm_totals);
    if (s->cm_gathered) (void)hipFree(s->cm_gathered);
    if (s->h_cm) (void)hipHostFree(s->h_cm);
    if (s->ds_state) (void)hipFree(s->ds_state);
    if (s->h_ds_state) (void)hipHostFree(s->h_ds_state);
    if (s->ds_consts) (void)hipFree(s->ds_consts);
    if (s->h_ds_consts) (void)hipHostFree(s->h_ds_consts);
    (void)hipFree(s->buf);
    (void)hipFree(s->partials);
    (void)hipFree(s->totals);
    (void)hipHostFree(s->h_totals);
    delete s;
    return ZF_OK;
}

// x_k = x_{k-1} = y = x0   (proximal_gradient.py:463-465)
extern "C" int zf_mo_set_x0(zf_mo* s, const double* x0_host) {
    ZF_REQUIRE(s && x0_host, "zf_mo_set_x0: null argument");
    s->y_pending = s->prep_pending = false;
    s->cur = 0;
    const size_t bytes = sizeof(double) * s->n;
    ZF_HIP(hipMemcpyAsync(s->xb[0], x0_host, bytes, hipMemcpyHostToDevice, s->stream));
    ZF_HIP(hipMemcpyAsync(s->xb[2], s->xb[0], bytes, hipMemcpyDeviceToDevice, s->stream));
    ZF_HIP(hipMemcpyAsync(s->y, s->xb[0], bytes, hipMemcpyDeviceToDevice, s->stream));
    ZF_HIP(hipStreamSynchronize(s->stream));
    return ZF_OK;
}

// f from the raw sums of k_jos1_sums / k_fds_sums (also produced by the tail of k_dual_solve)
static int mo_builtin_f(zf_mo* s, const double* x, double* f_out) {
    const double dn = (double)s->n_global;
    double t[4];
    if (s->kind == ZF_MO_JOS1) {
        hipLaunchKernelGGL(k_jos1_sums, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, x, s->n, s->partials);
        int rc = mo_reduce_to_host(s, 2, -1, t);
        if (rc) return rc;
        mo_f_from_sums(s->kind, dn, t, f_out);
        return ZF_OK;
    }
    if (s->kind == ZF_MO_FDS) {
        hipLaunchKernelGGL(k_fds_sums, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, x, s->n, s->n_global, s->offset,
                           s->partials);
        int rc = mo_reduce_to_host(s, 4, -1, t);
        if (rc) return rc;
        mo_f_from_sums(s->kind, dn, t, f_out);
        return ZF_OK;
    }
    return zf_fail(ZF_ERR_STATE, "zf_mo: f is a host callback for this problem kind");
}

static int mo_g_values(zf_mo* s, const double* x, double* g_out) {
    double t[MO_MAX_M + 1];
    const int m = s->m;
    MO_LAUNCH(k_g_terms, m, x, s->G, s->n, s->partials);
    int rc = mo_reduce_to_host(s, m + 1, -1, t);
    if (rc) return rc;
    for (int i = 0; i < m; ++i) {
        if (s->G.has_box && t[m] > 0.0) g_out[i] = INFINITY;             // problems.py:104-106
        else g_out[i] = s->G.has_l1 ? s->G.ratio[i] * t[i] : 0.0;         // :112-117
    }
    return ZF_OK;
}

// f(x), g(x) at which = 0: x_k, 1: y, 2: x+ ; f_out may be NULL (generic kind: host callback)
extern "C" int zf_mo_eval_F(zf_mo* s, int32_t which, double* f_out, double* g_out) {
    ZF_REQUIRE(s && g_out, "zf_mo_eval_F: null argument");
    if (int rc = mo_flush(s)) return rc;
    const double* x = mo_which(s, which);
    ZF_REQUIRE(x, "zf_mo_eval_F: bad point selector");
    if (f_out) {
        int rc = mo_builtin_f(s, x, f_out);
        if (rc) return rc;
    }
    return mo_g_values(s, x, g_out);
}

// J = jac_f(y), f_y = f(y) for the built-in problems
extern "C" int zf_mo_prepare(zf_mo* s, double* f_y_out) {
    ZF_REQUIRE(s && f_y_out, "zf_mo_prepare: null argument");
    s->prep_pending = false;   // (formed right here)
    int rc = mo_flush(s);
    if (rc) return rc;
    rc = mo_builtin_f(s, s->y, f_y_out);   // leaves the raw sums in s->totals (FDS needs sum x)
    if (rc) return rc;
    s->f_y_on_device = false;
    if (s->kind == ZF_MO_JOS1)
        hipLaunchKernelGGL(k_jos1_jac, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, s->y, s->J, s->n, s->n_global);
    else
        hipLaunchKernelGGL(k_fds_jac, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, s->y, s->J, s->n, s->n_global,
                           s->offset, s->totals);
    ZF_HIP(hipGetLastError());
    return ZF_OK;
}

// The same without a host round trip (unsharded built-in problems): the sums of f(y) are reduced on
// the device, f(y) is formed there (k_f_from_sums) and stays there for zf_mo_solve_dual_device, which
// reports it back with its result; the Jacobian kernel reads sum(y) from the same device totals.
// Everything is enqueued on the stream; nothing is synchronised.
extern "C" int zf_mo_prepare_async(zf_mo* s) {
    ZF_REQUIRE(s, "zf_mo_prepare_async: null argument");
    ZF_REQUIRE(s->kind == ZF_MO_JOS1 || s->kind == ZF_MO_FDS, "zf_mo_prepare_async: f is a host callback for this kind");
    ZF_REQUIRE(!s->exchange && !s->comm, "zf_mo_prepare_async: x is sharded over ranks (use zf_mo_prepare)");
    if (!s->f_y_dev) ZF_HIP(hipMalloc(&s->f_y_dev, sizeof(double) * MO_MAX_M));
    if (s->fused && s->m <= 3) {   // formed by the next zf_mo_solve_dual_device (or by mo_flush)
        s->prep_pending = true;
        s->f_y_on_device = true;
        return ZF_OK;
    }
    if (int rc = mo_flush(s)) return rc;
    return mo_prepare_async_now(s);
}

// Fused outer iteration on / off.  On: zf_mo_commit and zf_mo_prepare_async defer their work into the
// next zf_mo_solve_dual_device, whose one kernel then forms y (:534), f(y) and J = jac_f(y) itself -
// one launch and one read-back per trial instead of six launches.  Results of every entry point are
// unchanged (deferred work is done the unfused way whenever something else needs it).
extern "C" int zf_mo_set_fused(zf_mo* s, int32_t on) {
    ZF_REQUIRE(s, "zf_mo_set_fused: null argument");
    if (!on)
        if (int rc = mo_flush(s)) return rc;
    s->fused = on != 0;
    return ZF_OK;
}

static int mo_prepare_async_now(zf_mo* s) {
    int nq = 2;
    if (s->kind == ZF_MO_JOS1) {
        hipLaunchKernelGGL(k_jos1_sums, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, s->y, s->n, s->partials);
    } else {
        nq = 4;
        hipLaunchKernelGGL(k_fds_sums, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, s->y, s->n, s->n_global, s->offset,
                           s->partials);
    }
    hipLaunchKernelGGL(k_mo_reduce, dim3(1), dim3(64 * MO_RED_WAVES), 0, s->stream, s->partials, s->grid, nq, -1, s->totals);
    hipLaunchKernelGGL(k_f_from_sums, dim3(1), dim3(64), 0, s->stream, s->kind, (double)s->n_global, s->totals, s->f_y_dev);
    if (s->kind == ZF_MO_JOS1)
        hipLaunchKernelGGL(k_jos1_jac, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, s->y, s->J, s->n, s->n_global);
    else
        hipLaunchKernelGGL(k_fds_jac, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, s->y, s->J, s->n, s->n_global,
                           s->offset, s->totals);
    ZF_HIP(hipGetLastError());
    s->f_y_on_device = true;
    return ZF_OK;
}

// f(y) of the last zf_mo_prepare_async (synchronises the stream; only needed when the device search
// was not attempted and the host continues)
extern "C" int zf_mo_get_f_y(zf_mo* s, double* f_y_out) {
    ZF_REQUIRE(s && f_y_out && s->f_y_dev && s->f_y_on_device, "zf_mo_get_f_y: no zf_mo_prepare_async result");
    if (int rc = mo_flush(s)) return rc;
    ZF_HIP(hipMemcpyAsync(s->h_totals, s->f_y_dev, sizeof(double) * s->m, hipMemcpyDeviceToHost, s->stream));
    ZF_HIP(hipStreamSynchronize(s->stream));
    memcpy(f_y_out, s->h_totals, sizeof(double) * s->m);
    return ZF_OK;
}

// generic kind: J (m x n, row-major) computed by a host callback
extern "C" int zf_mo_set_jac(zf_mo* s, const double* J_host) {
    ZF_REQUIRE(s && J_host, "zf_mo_set_jac: null argument");
    s->prep_pending = false;
    if (int rc = mo_flush(s)) return rc;
    ZF_HIP(hipMemcpyAsync(s->J, J_host, sizeof(double) * s->m * s->n, hipMemcpyHostToDevice, s->stream));
    ZF_HIP(hipStreamSynchronize(s->stream));
    return ZF_OK;
}

// out: [0..m) g_i(p)  [m] |p-v|^2  [m+1] |w@J|^2  [m+2 .. 2m+2) J_i . (p - y)
extern "C" int zf_mo_dual_eval(zf_mo* s, double lr, const double* w_host, double* out) {
    ZF_REQUIRE(s && w_host && out, "zf_mo_dual_eval: null argument");
    if (int rc = mo_flush(s)) return rc;
    mo_w W;
    mo_fill_w(s, lr, w_host, &W);
    const int m = s->m;
    MO_LAUNCH(k_dual_eval, m, s->J, s->y, s->G, W, s->n, s->partials);
    int rc = mo_reduce_to_host(s, 2 * m + 2, -1, out);
    if (rc) return rc;
    for (int i = 0; i < m; ++i) out[i] = s->G.has_l1 ? s->G.ratio[i] * out[i] : 0.0;
    return ZF_OK;
}

// H_out (m x m, row-major): the generalised Hessian of the dual at w - d jac_i / d w_k on the quadratic piece w
// sits in (what the device-side search feeds its Newton model with; m <= 5: the reduction carries m x m sums)
extern "C" int zf_mo_dual_hessian(zf_mo* s, double lr, const double* w_host, double* H_out) {
    ZF_REQUIRE(s && w_host && H_out, "zf_mo_dual_hessian: null argument");
    ZF_REQUIRE(s->m <= 5, "zf_mo_dual_hessian: m <= 5");
    if (int rc = mo_flush(s)) return rc;
    mo_w W;
    mo_fill_w(s, lr, w_host, &W);
    const int m = s->m;
    switch (m) {
        case 2: hipLaunchKernelGGL(k_dual_hessian<2>, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, s->J, s->y, s->G, W, s->n, s->partials); break;
        case 3: hipLaunchKernelGGL(k_dual_hessian<3>, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, s->J, s->y, s->G, W, s->n, s->partials); break;
        case 4: hipLaunchKernelGGL(k_dual_hessian<4>, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, s->J, s->y, s->G, W, s->n, s->partials); break;
        default: hipLaunchKernelGGL(k_dual_hessian<5>, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, s->J, s->y, s->G, W, s->n, s->partials); break;
    }
    ZF_HIP(hipGetLastError());
    int rc = mo_reduce_to_host(s, m * m, -1, H_out);
    if (rc) return rc;
    for (int q = 0; q < m * m; ++q) H_out[q] *= lr;
    return ZF_OK;
}

// ---- the dual of one trial solved inside the library (ZF_DUAL_SOLVER=native) -------------------
// D(w) and its gradient from one fused evaluation, as zfista/proximal_gradient.py:161-177 composes
// them:  fun = -<w, g(p)> - |p - v|^2 / 2 / lr + lr / 2 |w@J|^2 (+ <w, F_old - f_y>),
//        jac = -g(p) - J (p - y) (+ F_old - f_y).
namespace {
struct mo_dual_ctx {
    zf_mo* s;
    double lr;
    const double* f_y;
    const double* F_old;
    int deprecated;
};
int mo_dual_fn(void* vctx, const double* w, double* fun, double* jac) {
    mo_dual_ctx* c = static_cast<mo_dual_ctx*>(vctx);
    const int m = c->s->m;
    double out[2 * MO_MAX_M + 2];
    int rc = zf_mo_dual_eval(c->s, c->lr, w, out);
    if (rc) return rc;
    const double* g_p = out;
    const double ss_pv = out[m], ss_wJ = out[m + 1];
    const double* dots = out + m + 2;
    double inner = 0.0;
    for (int i = 0; i < m; ++i) inner += w[i] * g_p[i];
    const double n_pv = sqrt(ss_pv), n_wJ = sqrt(ss_wJ);
    double f = -inner - n_pv * n_pv / 2 / c->lr + c->lr / 2 * (n_wJ * n_wJ);
    for (int i = 0; i < m; ++i) jac[i] = -g_p[i] - dots[i];
    if (!c->deprecated) {
        double corr = 0.0;
        for (int i = 0; i < m; ++i) {
            const double dF = c->F_old[i] - c->f_y[i];
            corr += w[i] * dF;
            jac[i] += dF;
        }
        f += corr;
    }
    *fun = f;
    return 0;
}
}  // namespace

// The same search with x sharded over the ranks of a library communicator: the points of a batch are evaluated
// back to back (one k_dual_eval + one reduce each, nothing synchronised in between), their 2m + 2 totals travel
// in ONE all-gather, are added in rank order on the device and come to the host in ONE copy - all ranks advance
// identical copies of the state machine on identical numbers.
namespace {
template <int M>
int mo_solve_dual_batched(zf_mo* s, double lr, const double* f_y, const double* F_old, int deprecated, const double* w0,
                          double tol, long max_iter, double* w_out, double* fun_out, long* nit_out, int* ok_out,
                          int64_t* evals) {
    constexpr int NQ = 2 * M + 2;
    using mach_t = zf_dual::machine<M>;
    mach_t S;
    S.start(w0, tol, max_iter);
    double funs[mach_t::NB] = {0}, jacs[mach_t::NB][M] = {{0}};
    static_assert(mach_t::NB * NQ <= MO_CM_CAP, "batch buffers too small");
    while (!S.done()) {
        const int npts = S.npts;
        for (int k = 0; k < npts; ++k) {
            mo_w W;
            mo_fill_w(s, lr, S.pts[k], &W);
            hipLaunchKernelGGL(k_dual_eval<M>, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, s->J, s->y, s->G, W, s->n,
                               s->cm_partials + (int64_t)k * NQ * s->grid);
        }
        // (quantity-major partials of consecutive points are one [npts * NQ][grid] array: one reduce launch)
        hipLaunchKernelGGL(k_mo_reduce, dim3(1), dim3(64 * MO_RED_WAVES), 0, s->stream, s->cm_partials, s->grid, npts * NQ, -1,
                           s->cm_totals);
        ZF_HIP(hipGetLastError());
        if (int rc = zf_comm_all_gather(s->comm, s->cm_totals, s->cm_gathered, npts * NQ, s->stream)) return rc;
        s->n_exchanges += 1;
        hipLaunchKernelGGL(k_mo_combine_ranks, dim3((npts * NQ + 63) / 64), dim3(64), 0, s->stream, s->cm_gathered,
                           s->comm_world, npts * NQ, -1, s->cm_totals);
        ZF_HIP(hipGetLastError());
        ZF_HIP(hipMemcpyAsync(s->h_cm, s->cm_totals, sizeof(double) * npts * NQ, hipMemcpyDeviceToHost, s->stream));
        ZF_HIP(hipStreamSynchronize(s->stream));
        *evals += npts;
        for (int k = 0; k < npts; ++k) {   // D(w), grad D(w) as mo_dual_fn composes them (:165-177)
            const double* t = s->h_cm + k * NQ;
            const double* w = S.pts[k];
            double g_p[M], inner = 0.0;
            for (int i = 0; i < M; ++i) {
                g_p[i] = s->G.has_l1 ? s->G.ratio[i] * t[i] : 0.0;
                inner += w[i] * g_p[i];
            }
            const double n_pv = sqrt(t[M]), n_wJ = sqrt(t[M + 1]);
            double f = -inner - n_pv * n_pv / 2 / lr + lr / 2 * (n_wJ * n_wJ);
            for (int i = 0; i < M; ++i) jacs[k][i] = -g_p[i] - t[M + 2 + i];
            if (!deprecated) {
                double corr = 0.0;
                for (int i = 0; i < M; ++i) {
                    const double dF = F_old[i] - f_y[i];
                    corr += w[i] * dF;
                    jacs[k][i] += dF;
                }
                f += corr;
            }
            funs[k] = f;
        }
        S.advance(funs, jacs);
    }
    *ok_out = S.ok;
    for (int i = 0; i < M; ++i) w_out[i] = S.w[i];
    *fun_out = S.fun;
    *nit_out = S.nit;
    return ZF_OK;
}
}  // namespace

// ---- the dual search on a SHARDED x without a host round trip per batch (dual_solver="device" through a library
// communicator).  The persistent kernel of the single-rank device search (k_dual_solve) cannot exchange between ranks
// from inside a launch; the host-driven search above synchronises with the host once per batch (~5 per trial).  Here the
// state machine (zf_dual::machine<M>, the probing mode the host loop runs) lives in DEVICE memory; a round is
//     k_ds_eval  (every point of the pending batch against this rank's share of J, y -> block partials)
//     k_mo_reduce -> zf_comm_all_gather (ONE collective per batch) -> k_ds_advance (one wave: totals added in rank
//     order, D(w) and grad D(w) composed as :165-177, the machine advanced, the next batch left in its place)
// - all stream-ordered; ROUNDS of them are enqueued back to back (a round after the search has finished exits at once:
// the machine says so) and the host looks once per chunk.  Every rank advances an identical copy on identical numbers.
namespace {
constexpr int MO_DS_CHUNK = 6;   // rounds enqueued between two looks of the host (a trial of cfg4 takes 4 - 5)

template <int M>
__global__ __launch_bounds__(ZF_BLOCK) void k_ds_eval(const double* __restrict__ J, const double* __restrict__ y, mo_g G,
                                                      const zf_dual::machine<M>* mach, double lr, int64_t n, double* partials) {
    constexpr int NQ = 2 * M + 2;
    using mach_t = zf_dual::machine<M>;
    __shared__ double lds[ZF_WAVES * NQ];
    if (mach->phase == mach_t::P_DONE) return;
    const int npts = mach->npts;
    __shared__ mo_w sW;   // (in LDS: as a local its arrays - handed to mo_prox by address - went to scratch memory)
    for (int k = 0; k < npts; ++k) {
        if (threadIdx.x == 0) {   // mo_fill_w on the device: weight = lr w, coef = weight * l1_ratios (proximal_gradient.py:164, problems.py:127)
            sW.lr = lr;
            double tail = 0.0;
#pragma unroll
            for (int i = 0; i < MO_MAX_M; ++i) {
                const double wi = i < M ? mach->pts[k][i < M ? i : 0] : 0.0;
                sW.w[i] = wi;
                sW.coef[i] = (i < M && G.has_l1) ? (lr * wi) * G.ratio[i] : 0.0;
                if (i >= 1 && i < M) tail += sW.coef[i];
            }
            sW.tail_sum = tail;
        }
        __syncthreads();
        const mo_w& W = sW;
        double acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
        const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
        for (int64_t j = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; j < n; j += stride) {   // (k_dual_eval's element body)
            double Jc[M];
            double wJ = 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                Jc[i] = J[(int64_t)i * n + j];
                wJ += W.w[i] * Jc[i];
            }
            const double yj = y[j];
            const double v = yj - W.lr * wJ;
            const double p = mo_prox(G, W.coef, W.tail_sum, v, j);
#pragma unroll
            for (int i = 0; i < M; ++i) acc[i] += fabs(p - G.shift[i]);
            const double dv = p - v;
            acc[M] += dv * dv;
            acc[M + 1] += wJ * wJ;
            const double dy = p - yj;
#pragma unroll
            for (int i = 0; i < M; ++i) acc[M + 2 + i] += Jc[i] * dy;
        }
        const double maxs[1] = {0.0};
        double out = 0.0;
        zf_block_reduce<NQ, 0, ZF_WAVES>(acc, maxs, lds, out);
        if (threadIdx.x < NQ) partials[(int64_t)(k * NQ + threadIdx.x) * gridDim.x + blockIdx.x] = out;
        __syncthreads();   // (lds is reused by the next point)
    }
}

struct mo_ds_consts {
    double lr;
    int deprecated, has_l1, world, pad;
    double ratio[MO_MAX_M];
};

template <int M>
__global__ __launch_bounds__(64) void k_ds_advance(zf_dual::machine<M>* g_mach, const double* __restrict__ gathered,
                                                   const double* __restrict__ consts /* f_y, F_old */, mo_ds_consts C,
                                                   long long* evals) {
    constexpr int NQ = 2 * M + 2;
    using mach_t = zf_dual::machine<M>;
    constexpr int NB = mach_t::NB;
    __shared__ double s_tot[NB * NQ];
    __shared__ double s_fun[NB];
    __shared__ double s_jac[NB][M];
    __shared__ mach_t s_mach;
    const int lane = threadIdx.x;
    if (g_mach->phase == mach_t::P_DONE) return;
    const int npts = g_mach->npts;
    for (int q = lane; q < npts * NQ; q += 64) {   // this batch's totals of all ranks, added in RANK ORDER
        double v = gathered[q];
        for (int r = 1; r < C.world; ++r) v += gathered[(int64_t)r * (NB * NQ) + q];
        s_tot[q] = v;
    }
    __syncthreads();
    if (lane < NB) {   // D(w), grad D(w) of point `lane` as mo_dual_fn composes them (:165-177)
        double f = 0.0, jac[M];
#pragma unroll
        for (int i = 0; i < M; ++i) jac[i] = 0.0;
        if (lane < npts) {
            const double* t = s_tot + lane * NQ;
            double g_p[M], inner = 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                g_p[i] = C.has_l1 ? C.ratio[i] * t[i] : 0.0;
                inner += g_mach->pts[lane][i] * g_p[i];
            }
            const double n_pv = sqrt(t[M]), n_wJ = sqrt(t[M + 1]);
            f = -inner - n_pv * n_pv / 2 / C.lr + C.lr / 2 * (n_wJ * n_wJ);
#pragma unroll
            for (int i = 0; i < M; ++i) jac[i] = -g_p[i] - t[M + 2 + i];
            if (!C.deprecated) {
                double corr = 0.0;
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    const double dF = consts[MO_MAX_M + i] - consts[i];
                    corr += g_mach->pts[lane][i] * dF;
                    jac[i] += dF;
                }
                f += corr;
            }
        }
        s_fun[lane] = f;
#pragma unroll
        for (int i = 0; i < M; ++i) s_jac[lane][i] = jac[i];
    }
    // (the whole wave steps the machine: the KKT systems of the simplex QP are solved lane-parallel on the device)
    if (lane == 0) s_mach = *g_mach;
    __syncthreads();
    {
        mach_t mach = s_mach;
        mach.advance(*reinterpret_cast<const double(*)[NB]>(s_fun), *reinterpret_cast<const double(*)[NB][M]>(s_jac));
        if (lane == 0) {
            *g_mach = mach;
            *evals += npts;
        }
    }
}

template <int M>
int mo_solve_dual_stream(zf_mo* s, double lr, const double* f_y, const double* F_old, int deprecated, const double* w0, double tol,
                         long max_iter, double* w_out, double* fun_out, long* nit_out, int* ok_out, int64_t* evals) {
    constexpr int NQ = 2 * M + 2;
    using mach_t = zf_dual::machine<M>;
    constexpr int NB = mach_t::NB;
    static_assert(NB * NQ <= MO_CM_CAP, "batch buffers too small");
    static_assert(sizeof(mach_t) + 16 <= 8192, "the device blob holds the machine and the evaluation counter");
    mach_t* hm = static_cast<mach_t*>(s->h_ds_state);
    long long* h_evals = reinterpret_cast<long long*>(static_cast<char*>(s->h_ds_state) + 8192 - 16);
    memset(s->h_ds_state, 0, 8192);
    hm->start(w0, tol, max_iter);
    *h_evals = 0;
    for (int i = 0; i < MO_MAX_M; ++i) {
        s->h_ds_consts[i] = i < M ? f_y[i] : 0.0;
        s->h_ds_consts[MO_MAX_M + i] = i < M ? F_old[i] : 0.0;
    }
    ZF_HIP(hipMemcpyAsync(s->ds_state, s->h_ds_state, 8192, hipMemcpyHostToDevice, s->stream));
    ZF_HIP(hipMemcpyAsync(s->ds_consts, s->h_ds_consts, sizeof(double) * 2 * MO_MAX_M, hipMemcpyHostToDevice, s->stream));
    mo_ds_consts C;
    C.lr = lr;
    C.deprecated = deprecated;
    C.has_l1 = s->G.has_l1;
    C.world = s->comm_world;
    C.pad = 0;
    for (int i = 0; i < MO_MAX_M; ++i) C.ratio[i] = s->G.ratio[i];
    mach_t* dm = static_cast<mach_t*>(s->ds_state);
    long long* d_evals = reinterpret_cast<long long*>(static_cast<char*>(s->ds_state) + 8192 - 16);
    // a search needs a handful of batches; max_iter Newton iterations bound it (2 batches each + the start)
    const long cap = 2 * max_iter + 8;
    for (long done_rounds = 0; done_rounds < cap;) {
        for (int r = 0; r < MO_DS_CHUNK; ++r) {
            hipLaunchKernelGGL(k_ds_eval<M>, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, s->J, s->y, s->G, dm, lr, s->n, s->cm_partials);
            hipLaunchKernelGGL(k_mo_reduce, dim3(1), dim3(64 * MO_RED_WAVES), 0, s->stream, s->cm_partials, s->grid, NB * NQ, -1,
                               s->cm_totals);
            ZF_HIP(hipGetLastError());
            if (int rc = zf_comm_all_gather(s->comm, s->cm_totals, s->cm_gathered, NB * NQ, s->stream)) return rc;
            s->n_exchanges += 1;
            hipLaunchKernelGGL(k_ds_advance<M>, dim3(1), dim3(64), 0, s->stream, dm, s->cm_gathered, s->ds_consts, C, d_evals);
            ZF_HIP(hipGetLastError());
        }
        done_rounds += MO_DS_CHUNK;
        s->ds_rounds += MO_DS_CHUNK;
        ZF_HIP(hipMemcpyAsync(s->h_ds_state, s->ds_state, 8192, hipMemcpyDeviceToHost, s->stream));
        ZF_HIP(hipStreamSynchronize(s->stream));
        if (hm->done()) break;
    }
    *evals += (int64_t)*h_evals;
    if (!hm->done()) return zf_fail(ZF_ERR_STATE, "zf_mo_solve_dual_stream: the search did not finish within its round budget%s");
    *ok_out = hm->ok;
    for (int i = 0; i < M; ++i) w_out[i] = hm->w[i];
    *fun_out = hm->fun;
    *nit_out = hm->nit;
    return ZF_OK;
}
}  // namespace

// dual_solver="device" with x sharded over a library communicator (zf_mo_set_comm): the search of zf_mo_solve_dual with
// the state machine on the device - ONE collective per batch, no host synchronisation per batch (MO_DS_CHUNK batches
// per look).  Same arguments and results as zf_mo_solve_dual; x+ is NOT recovered here (zf_mo_recover follows).
extern "C" int zf_mo_solve_dual_stream(zf_mo* s, double lr, const double* f_y, const double* F_old, int32_t deprecated,
                                       const double* w0, double tol, int64_t max_iter, double* w_out, double* fun_out,
                                       int64_t* nit_out, int32_t* ok_out, int64_t* evals_out) {
    ZF_REQUIRE(s && f_y && F_old && w_out && fun_out && nit_out && ok_out, "zf_mo_solve_dual_stream: null argument");
    ZF_REQUIRE(lr > 0.0 && max_iter >= 1, "zf_mo_solve_dual_stream: lr must be > 0 and max_iter >= 1");
    ZF_REQUIRE(s->comm, "zf_mo_solve_dual_stream: needs a communicator (zf_mo_set_comm); single-rank solves take zf_mo_solve_dual_device");
    if (int rc = mo_flush(s)) return rc;
    if (!s->ds_state) {
        ZF_HIP(hipMalloc(&s->ds_state, 8192));
        ZF_HIP(hipHostMalloc(&s->h_ds_state, 8192, hipHostMallocDefault));
        ZF_HIP(hipMalloc(&s->ds_consts, sizeof(double) * 2 * MO_MAX_M));
        ZF_HIP(hipHostMalloc((void**)&s->h_ds_consts, sizeof(double) * 2 * MO_MAX_M, hipHostMallocDefault));
    }
    long nit = 0;
    int ok = 1;
    double fun = 0.0;
    int64_t evals = 0;
    int rc = ZF_ERR_ARG;
    switch (s->m) {
#define ZF_MO_STREAM(MM) case MM: rc = mo_solve_dual_stream<MM>(s, lr, f_y, F_old, (int)deprecated, w0, tol, (long)max_iter, w_out, &fun, &nit, &ok, &evals); break;
        ZF_MO_STREAM(2) ZF_MO_STREAM(3) ZF_MO_STREAM(4) ZF_MO_STREAM(5) ZF_MO_STREAM(6) ZF_MO_STREAM(7) ZF_MO_STREAM(8)
#undef ZF_MO_STREAM
    }
    if (evals_out) *evals_out = evals;
    if (rc) return rc;
    *fun_out = fun;
    *nit_out = nit;
    *ok_out = ok;
    return ZF_OK;
}

// w0 may be NULL (uniform start).  *ok_out = 0: the start point is not finite (e.g. F(x_k) = inf
// outside the box) - nothing was solved, use the reference's SciPy calls.  *evals_out counts the
// dual evaluations spent.
extern "C" int zf_mo_solve_dual(zf_mo* s, double lr, const double* f_y, const double* F_old, int32_t deprecated,
                                const double* w0, double tol, int64_t max_iter, double* w_out, double* fun_out,
                                int64_t* nit_out, int32_t* ok_out, int64_t* evals_out) {
    ZF_REQUIRE(s && f_y && F_old && w_out && fun_out && nit_out && ok_out, "zf_mo_solve_dual: null argument");
    ZF_REQUIRE(lr > 0.0 && max_iter >= 1, "zf_mo_solve_dual: lr must be > 0 and max_iter >= 1");
    if (int rc = mo_flush(s)) return rc;
    if (s->comm) {   // sharded over a library communicator: one exchange per batch of the search
        long nit = 0;
        int ok = 1;
        double fun = 0.0;
        int64_t evals = 0;
        int rc = ZF_ERR_ARG;
        switch (s->m) {
#define ZF_MO_BATCHED(MM) case MM: rc = mo_solve_dual_batched<MM>(s, lr, f_y, F_old, (int)deprecated, w0, tol, (long)max_iter, w_out, &fun, &nit, &ok, &evals); break;
            ZF_MO_BATCHED(2) ZF_MO_BATCHED(3) ZF_MO_BATCHED(4) ZF_MO_BATCHED(5) ZF_MO_BATCHED(6) ZF_MO_BATCHED(7) ZF_MO_BATCHED(8)
#undef ZF_MO_BATCHED
        }
        if (evals_out) *evals_out = evals;
        if (rc) return rc;
        *fun_out = fun;
        *nit_out = nit;
        *ok_out = ok;
        return ZF_OK;
    }
    mo_dual_ctx ctx = {s, lr, f_y, F_old, (int)deprecated};
    zf_dual::evaluator E = {mo_dual_fn, &ctx, 0};
    long nit = 0;
    int ok = 1;
    double fun = 0.0;
    const int rc = zf_dual::solve(E, s->m, w0, tol, (long)max_iter, w_out, &fun, &nit, &ok);
    if (evals_out) *evals_out = E.evals;
    if (rc) return ZF_ERR_STATE;   // (the failing evaluation left its message)
    *fun_out = fun;
    *nit_out = nit;
    *ok_out = ok;
    return ZF_OK;
}

// The same search inside ONE persistent kernel (k_dual_solve), primal recovery included: on return
// x+ is in its buffer and *err_out = max|x+ - y| (what zf_mo_recover would have produced).  One
// launch and one 128-byte read-back per trial; no host round trip per dual evaluation.
// f_x_out / g_x_out (m each, may be NULL): f(x+), g(x+) from sums taken in the same pass (:295) - f_x_out[0]
// is NaN when f is a host callback (ZF_MO_GENERIC).  f_y may be NULL after zf_mo_prepare_async (f(y) is
// then read on the device); f_y_out (m, may be NULL) receives the f(y) the search used, also when it
// was not attempted.
// *ok_out = 0: not attempted (non-finite start, sharded x, unsupported m) - use zf_mo_solve_dual /
// the reference's calls and zf_mo_recover instead.
// ---- launch / wait halves of a device trial ----------------------------------------------------
namespace {
struct mo_launch_opts {
    int gated = 0;          // launched ahead: runs only if the launch before it accepted its trial, F(x_k) from there
    int decay_is_one = 0;   // decay_rate == 1: every trial is accepted (:299)
    int warm = 0;           // warm_start: a gated trial starts its search from the weights of the trial before it
    double accept_tol = 0.0;
};

// launches one trial; *slot_out = the record slot (0 / 1) to wait on; *launched = 0: not attempted here
// (sharded x, m > 3: the caller's host loop), nothing was enqueued
int mo_trial_launch(zf_mo* s, double lr, const double* f_y, const double* F_old, int32_t deprecated, const double* w0,
                    double tol, int64_t max_iter, const mo_launch_opts& L, int* slot_out, int* launched) {
    *launched = 0;
    // fused outer iteration: y (deferred commit), f(y) and J are formed by this launch itself
    const bool fuse = s->fused && s->prep_pending && !f_y && !s->exchange && !s->comm && s->m <= 3 &&
                      (s->kind == ZF_MO_JOS1 || s->kind == ZF_MO_FDS);
    if (!fuse)
        if (int rc = mo_flush(s)) return rc;
    if (s->exchange || s->comm) return ZF_OK;   // sharded x: every evaluation needs an exchange
    if (s->solve_unavailable) return ZF_OK;      // (the grid cannot be co-resident on this device: the caller's host loop)
    if (!s->solve_partials) {
        int dev = 0, cus = 0;
        ZF_HIP(hipGetDevice(&dev));
        ZF_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        int64_t g = (s->n + MO_SOLVE_TPB - 1) / MO_SOLVE_TPB;
        if (g > cus) g = cus;      // one workgroup per CU: the whole grid is resident (grid-wide waits)
        if (g < 1) g = 1;
        s->solve_grid = (int)g;
        if (const char* sl = getenv("ZF_MO_SPIN_LIMIT")) {
            const long long v = atoll(sl);
            if (v > 0) s->spin_limit = v > 0x7fffffffLL ? 0x7fffffffu : (unsigned)v;
        }
        // the hand-over records in UNCACHED device memory: every access is an agent-scope atomic that has to
        // reach memory anyway (the 8 XCDs' L2s are not coherent with each other); without the L2 in the way a
        // hand-over is ~0.5 us shorter (cfg4 9 700 -> 10 150-10 300 it/s, same box, A/B)
        ZF_HIP(hipExtMallocWithFlags((void**)&s->solve_partials, 16 * 2 * (size_t)MO_REC_CAP * s->solve_grid,
                                     hipDeviceMallocUncached));
        ZF_HIP(hipExtMallocWithFlags((void**)&s->solve_totals, 16 * 2 * (size_t)MO_REC_CAP, hipDeviceMallocUncached));
        ZF_HIP(hipMemsetAsync(s->solve_partials, 0, 16 * 2 * (size_t)MO_REC_CAP * s->solve_grid, s->stream));
        ZF_HIP(hipMemsetAsync(s->solve_totals, 0, 16 * 2 * (size_t)MO_REC_CAP, s->stream));
        // the result records: pinned host memory the kernel writes directly (fine-grained, device-visible)
        ZF_HIP(hipHostMalloc((void**)&s->h_solve_out, 2 * sizeof(mo_solve_result), hipHostMallocMapped));
        memset(s->h_solve_out, 0, 2 * sizeof(mo_solve_result));
        ZF_HIP(hipHostGetDevicePointer((void**)&s->solve_out, s->h_solve_out, 0));
        ZF_HIP(hipMalloc(&s->accept_dev, 2 * sizeof(int)));
        ZF_HIP(hipMalloc(&s->F_dev, 4 * MO_MAX_M * sizeof(double)));
        ZF_HIP(hipMemsetAsync(s->accept_dev, 0, 2 * sizeof(int), s->stream));
        ZF_HIP(hipMemsetAsync(s->F_dev, 0, 4 * MO_MAX_M * sizeof(double), s->stream));
    }
    mo_solve_args A;
    memset(&A, 0, sizeof(A));
    A.J = s->J;
    A.y = s->y;
    A.xn = s->xb[(s->cur + 1) % 3];
    A.G = s->G;
    A.n = s->n;
    A.lr = lr;
    for (int i = 0; i < s->m; ++i) {
        A.f_y[i] = f_y ? f_y[i] : 0.0;
        A.F_old[i] = F_old ? F_old[i] : 0.0;
        if (w0) A.w0[i] = w0[i];
    }
    A.f_y_dev = f_y ? nullptr : s->f_y_dev;
    A.deprecated = deprecated;
    A.has_w0 = w0 != nullptr;
    A.tol = tol;
    A.max_iter = max_iter;
    A.tail_kind = (s->kind == ZF_MO_JOS1 || s->kind == ZF_MO_FDS) ? s->kind : 0;
    A.partials = s->solve_partials;
    A.totals = s->solve_totals;
    A.nonce = ++s->solve_nonce;
    const int slot = (int)(A.nonce & 1u);
    A.out = s->solve_out + slot;
    A.accept_out = s->accept_dev + slot;
    A.F_new_out = s->F_dev + slot * MO_MAX_M;
    A.w_new_out = s->F_dev + (2 + slot) * MO_MAX_M;
    if (L.gated && L.warm) A.w0_dev = s->F_dev + (2 + (1 - slot)) * MO_MAX_M;   // the weights of the trial launched before
    A.decay_is_one = L.decay_is_one;
    A.accept_tol = L.accept_tol;
    if (L.gated) {   // the launch before this one has the other parity
        A.gate = s->accept_dev + (1 - slot);
        A.F_old_dev = s->F_dev + (1 - slot) * MO_MAX_M;
    }
    if (fuse) {
        A.prep_kind = s->kind;
        A.make_y = s->y_pending ? 1 : 0;
        A.nesterov = s->pend_nesterov;
        A.beta = s->pend_beta;
        A.xk = s->xb[s->cur];
        A.xo = s->xb[(s->cur + 2) % 3];
        A.y_w = s->y;
        A.J_w = s->J;
        A.f_y_w = s->f_y_dev;
        s->y_pending = s->prep_pending = false;   // (this launch leaves y, J and f(y) behind in their buffers)
    }
    const dim3 grid(s->solve_grid), block(MO_SOLVE_TPB);
    // rows of MO_SOLVE_TPB elements a workgroup owns / can keep in LDS
    const int64_t per_wg = (s->n + (int64_t)s->solve_grid * MO_SOLVE_TPB - 1) / ((int64_t)s->solve_grid * MO_SOLVE_TPB);
    // (round 4: the search for every m the engine takes, 2 .. 8.  m <= 3 fit 256 VGPRs without scratch; from m = 4 on the
    //  solver step of the one deciding wave - the lane-parallel KKT systems of the simplex QP, (m + 1) x (m + 1) each -
    //  spills 270 B .. 2 KB per thread to scratch, outside the element loops for m <= 6: tests/test_abi.py)
    const void* fn = nullptr;
#define MO_PICK(M_) case M_: fn = (const void*)k_dual_solve<M_>; break;
    switch (s->m) {
        MO_PICK(2) MO_PICK(3) MO_PICK(4) MO_PICK(5) MO_PICK(6) MO_PICK(7) MO_PICK(8)
        default: return ZF_OK;
    }
#undef MO_PICK
    if (s->solve_static_lds < 0) {
        hipFuncAttributes fa;
        ZF_HIP(hipFuncGetAttributes(&fa, fn));
        s->solve_static_lds = (int64_t)fa.sharedSizeBytes;
    }
    int64_t lds_budget = (int64_t)160 * 1024 - s->solve_static_lds - 512;
    if (lds_budget > MO_SOLVE_LDS_BYTES) lds_budget = MO_SOLVE_LDS_BYTES;
    const int64_t cap = lds_budget / ((int64_t)(s->m + 1) * MO_SOLVE_TPB * sizeof(double));
    A.resident_rows = (int)(per_wg < cap ? per_wg : cap);
    const size_t lds = (size_t)A.resident_rows * (s->m + 1) * MO_SOLVE_TPB * sizeof(double);
    const bool set_attr = s->solve_lds_set != lds;   // (driver calls of several microseconds: once, not per trial)
    s->solve_lds_set = lds;
    A.spin_limit = s->spin_limit;
    A.force_timeout = 0;
    if (s->force_timeouts > 0) {
        s->force_timeouts -= 1;
        A.force_timeout = 1;
    }
    if (set_attr) {
        ZF_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        // Every workgroup spins on grid-wide hand-overs: the whole grid has to be resident at once.  What the
        // device can hold of THIS kernel (registers, LDS) is checked here, once per LDS size; what other work
        // on the device takes away at run time cannot be known in advance - then a wait gives up after
        // spin_limit polls, the record says ok = -1 and the caller continues with its host loop.
        int per_cu = 0, dev = 0, cus = 0;
        ZF_HIP(hipGetDevice(&dev));
        ZF_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        ZF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, MO_SOLVE_TPB, lds));
        if ((int64_t)per_cu * cus < (int64_t)s->solve_grid) {
            s->solve_unavailable = true;
            s->solve_lds_set = (size_t)-1;
            if (fuse) s->y_pending = A.make_y != 0, s->prep_pending = true;   // (nothing was formed: still due)
            return ZF_OK;
        }
    }
    {
        void* kargs[] = {(void*)&A};
        ZF_HIP(hipLaunchKernel(fn, grid, block, kargs, lds, s->stream));
    }
    ZF_HIP(hipGetLastError());
    s->last_nonce[slot] = A.nonce;
    s->last_slot = slot;
    s->launched[slot] = {fuse, fuse && A.make_y != 0, A.beta, A.nesterov};
    *slot_out = slot;
    *launched = 1;
    return ZF_OK;
}

// waits for the record of slot `slot` (the kernel stores `seq` last): no copy, no stream synchronisation.
// Should the launch die without writing it, the stream goes idle: checked now and then, reported as an error.
int mo_trial_wait(zf_mo* s, int slot, const mo_solve_result** out) {
    volatile unsigned long long* seq = &s->h_solve_out[slot].seq;
    const unsigned long long want = (unsigned long long)s->last_nonce[slot];
    unsigned spins = 0;
    while (*seq != want) {
        if ((++spins & 0xFFFu) == 0) {
            const hipError_t q = hipStreamQuery(s->stream);
            if (q == hipSuccess) {
                if (*seq == want) break;
                return zf_fail(ZF_ERR_STATE, "zf_mo: the trial kernel ended without a result record%s");
            }
            if (q != hipErrorNotReady) return zf_fail(ZF_ERR_HIP, "zf_mo: %s", hipGetErrorString(q));
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    *out = &s->h_solve_out[slot];
    return ZF_OK;
}

// unpacks a record the way zf_mo_solve_dual_device reports it
int mo_trial_unpack(zf_mo* s, const mo_solve_result& r, double* w_out, double* fun_out, int64_t* nit_out, int32_t* ok_out,
                    int64_t* evals_out, double* err_out, double* f_x_out, double* g_x_out, double* f_y_out) {
    *ok_out = 0;
    if (evals_out) *evals_out = r.skipped ? 0 : r.evals;
    if (r.skipped) return ZF_OK;
    if (f_y_out)
        for (int i = 0; i < s->m; ++i) f_y_out[i] = r.f_y[i];
    if (r.ok < 0) {   // a grid-wide wait gave up (another kernel occupying the CUs?): nothing of this trial is usable
        *ok_out = -1;
        return ZF_OK;
    }
    if (r.ok == 0) return ZF_OK;
    for (int i = 0; i < s->m; ++i) w_out[i] = r.w[i];
    *fun_out = r.fun;
    *nit_out = r.nit;
    *err_out = r.err;
    *ok_out = 1;
    // F(x+) from the sums of the same kernel (:295): g always, f for the built-in problems
    if (g_x_out)
        for (int i = 0; i < s->m; ++i) {
            if (s->G.has_box && r.g_sums[s->m] > 0.0) g_x_out[i] = INFINITY;      // problems.py:104-106
            else g_x_out[i] = s->G.has_l1 ? s->G.ratio[i] * r.g_sums[i] : 0.0;     // :112-117
        }
    if (f_x_out) {
        if (s->kind == ZF_MO_JOS1 || s->kind == ZF_MO_FDS) mo_f_from_sums(s->kind, (double)s->n_global, r.f_sums, f_x_out);
        else f_x_out[0] = NAN;   // f is a host callback for this kind
    }
    return ZF_OK;
}
}  // namespace

extern "C" int zf_mo_solve_dual_device(zf_mo* s, double lr, const double* f_y, const double* F_old, int32_t deprecated,
                                       const double* w0, double tol, int64_t max_iter, double* w_out, double* fun_out,
                                       int64_t* nit_out, int32_t* ok_out, int64_t* evals_out, double* err_out,
                                       double* f_x_out, double* g_x_out, double* f_y_out) {
    ZF_REQUIRE(s && F_old && w_out && fun_out && nit_out && ok_out && err_out, "zf_mo_solve_dual_device: null argument");
    ZF_REQUIRE(f_y || s->f_y_on_device, "zf_mo_solve_dual_device: f_y is NULL and no zf_mo_prepare_async result exists");
    ZF_REQUIRE(lr > 0.0 && max_iter >= 1, "zf_mo_solve_dual_device: lr must be > 0 and max_iter >= 1");
    *ok_out = 0;
    if (evals_out) *evals_out = 0;
    int slot = 0, launched = 0;
    int rc = mo_trial_launch(s, lr, f_y, F_old, deprecated, w0, tol, max_iter, mo_launch_opts(), &slot, &launched);
    if (rc || !launched) return rc;
    const mo_solve_result* r = nullptr;
    if ((rc = mo_trial_wait(s, slot, &r))) return rc;
    return mo_trial_unpack(s, *r, w_out, fun_out, nit_out, ok_out, evals_out, err_out, f_x_out, g_x_out, f_y_out);
}

// Trials launched AHEAD of their predecessor's result (the outer loop without the host in its critical
// path).  zf_mo_trial_launch enqueues one trial and returns a ticket; zf_mo_trial_wait returns its result
// and the acceptance decision (:298-303) the kernel took on F(x+) = f(x+) + g(x+) formed on the device.
// A trial launched with gated = 1 - after zf_mo_commit + zf_mo_prepare_async in fused mode, BEFORE the
// result of the trial before it is known - runs only if that trial was accepted (else it exits at once,
// *skipped_out = 1, having touched nothing: call zf_mo_uncommit and retry with a smaller step) and takes
// F(x_k) from that trial's F(x+) on the device (F_old may be NULL).  At most one trial may be in flight
// ahead of the one being waited for.
extern "C" int zf_mo_trial_launch(zf_mo* s, double lr, const double* F_old, int32_t deprecated, const double* w0,
                                  double tol, int64_t max_iter, double accept_tol, int32_t decay_is_one, int32_t gated,
                                  int32_t* ticket_out) {
    ZF_REQUIRE(s && ticket_out, "zf_mo_trial_launch: null argument");
    ZF_REQUIRE(F_old || gated, "zf_mo_trial_launch: F_old may be NULL only for a gated trial");
    ZF_REQUIRE(s->f_y_on_device, "zf_mo_trial_launch: call zf_mo_prepare_async first");
    ZF_REQUIRE(lr > 0.0 && max_iter >= 1, "zf_mo_trial_launch: lr must be > 0 and max_iter >= 1");
    mo_launch_opts L;
    L.gated = gated != 0;
    L.warm = gated == 2;   // gated AND warm-started from the weights the trial before it ends with (on the device)
    L.decay_is_one = decay_is_one != 0;
    L.accept_tol = accept_tol;
    int slot = 0, launched = 0;
    int rc = mo_trial_launch(s, lr, nullptr, F_old, deprecated, w0, tol, max_iter, L, &slot, &launched);
    if (rc) return rc;
    *ticket_out = launched ? slot : -1;   // -1: not a device trial (sharded x, m > 3)
    return ZF_OK;
}

extern "C" int zf_mo_trial_wait(zf_mo* s, int32_t ticket, double* w_out, double* fun_out, int64_t* nit_out,
                                int32_t* ok_out, int64_t* evals_out, double* err_out, double* f_x_out, double* g_x_out,
                                double* f_y_out, int32_t* accepted_out, int32_t* skipped_out) {
    ZF_REQUIRE(s && w_out && fun_out && nit_out && ok_out && err_out && accepted_out && skipped_out &&
                   (ticket == 0 || ticket == 1) && s->h_solve_out,
               "zf_mo_trial_wait: bad argument");
    const mo_solve_result* r = nullptr;
    int rc = mo_trial_wait(s, ticket, &r);
    if (rc) return rc;
    *accepted_out = r->accepted;
    *skipped_out = r->skipped;
    rc = mo_trial_unpack(s, *r, w_out, fun_out, nit_out, ok_out, evals_out, err_out, nullptr, nullptr, f_y_out);
    if (rc || !*ok_out) return rc;
    for (int i = 0; i < s->m; ++i) {   // the values the kernel decided on
        if (f_x_out) f_x_out[i] = r->f_x[i];
        if (g_x_out) g_x_out[i] = r->g_x[i];
    }
    return ZF_OK;
}

// After the device trial `ticket` gave up (*ok_out = -1; and after zf_mo_uncommit, if a trial had been launched
// ahead of it): whatever its prologue was to form - y = x_k + beta (x_k - x_{k-1}), f(y), J = jac_f(y) - is not to
// be trusted (the sums behind f(y), J never completed) and is due again; the next entry point that needs them
// forms them the unfused way.
extern "C" int zf_mo_invalidate_prepare(zf_mo* s, int32_t ticket) {
    ZF_REQUIRE(s && ticket >= -1 && ticket <= 1, "zf_mo_invalidate_prepare: bad argument");
    if (ticket < 0) ticket = s->last_slot;   // the most recent launch (zf_mo_solve_dual_device)
    if (s->kind != ZF_MO_JOS1 && s->kind != ZF_MO_FDS) return ZF_OK;   // (f, J are the host's for this kind)
    if (!s->launched[ticket].fused) return ZF_OK;                      // (the launch formed nothing itself)
    if (!s->f_y_dev) ZF_HIP(hipMalloc(&s->f_y_dev, sizeof(double) * MO_MAX_M));
    if (s->launched[ticket].make_y) {
        s->y_pending = true;
        s->pend_beta = s->launched[ticket].beta;
        s->pend_nesterov = s->launched[ticket].nesterov;
    }
    s->prep_pending = true;
    s->f_y_on_device = true;
    return ZF_OK;
}

// test hook: the next `launches` device trials report a timeout at their first grid-wide hand-over
extern "C" int zf_mo_debug_force_timeout(zf_mo* s, int32_t launches) {
    ZF_REQUIRE(s && launches >= 0, "zf_mo_debug_force_timeout: bad argument");
    s->force_timeouts = launches;
    return ZF_OK;
}

// undo of a zf_mo_commit (+ zf_mo_prepare_async) whose gated trial was skipped: x_k, x_{k-1}, y, J and
// f(y) are those of the rejected trial again
extern "C" int zf_mo_uncommit(zf_mo* s) {
    ZF_REQUIRE(s, "zf_mo_uncommit: null argument");
    s->cur = (s->cur + 2) % 3;
    s->y_pending = s->prep_pending = false;
    return ZF_OK;
}

// diagnostics of the last zf_mo_solve_dual_device call: [0] batches, [1] evaluations, [2..5] shader-clock
// cycles of workgroup 0: whole kernel, evaluation loops, grid-wide hand-overs, solver steps
extern "C" int zf_mo_solve_stats(zf_mo* s, int64_t* out, int64_t count) {
    ZF_REQUIRE(s && out && count >= 6, "zf_mo_solve_stats: needs a buffer of >= 6 entries");
    ZF_REQUIRE(s->h_solve_out, "zf_mo_solve_stats: no device solve has run");
    const mo_solve_result& r = s->h_solve_out[s->last_slot];
    out[0] = r.batches, out[1] = r.evals, out[2] = r.cyc_total, out[3] = r.cyc_eval, out[4] = r.cyc_combine,
    out[5] = r.cyc_step;
    return ZF_OK;
}

// x+ = prox(lr w, y - lr w@J) ; *err_out = max|x+ - y|
extern "C" int zf_mo_recover(zf_mo* s, double lr, const double* w_host, double* err_out) {
    ZF_REQUIRE(s && w_host && err_out, "zf_mo_recover: null argument");
    if (int rc = mo_flush(s)) return rc;
    mo_w W;
    mo_fill_w(s, lr, w_host, &W);
    double* xn = s->xb[(s->cur + 1) % 3];
    const int m = s->m;
    MO_LAUNCH(k_recover, m, s->J, s->y, xn, s->G, W, s->n, s->partials);
    return mo_reduce_to_host(s, 1, 0, err_out);
}

// accept x+: x_{k-1} <- x_k <- x+ ; y = x_k + beta (x_k - x_{k-1})  (or y = x_k)   :530-538
extern "C" int zf_mo_commit(zf_mo* s, double beta, int32_t nesterov) {
    ZF_REQUIRE(s, "zf_mo_commit: null argument");
    if (int rc = mo_flush(s)) return rc;   // (a commit still pending belongs to the OLD x_k, x_{k-1})
    if (s->fused) {
        s->cur = (s->cur + 1) % 3;
        s->y_pending = true;
        s->pend_beta = beta;
        s->pend_nesterov = (int)nesterov;
        return ZF_OK;
    }
    s->cur = (s->cur + 1) % 3;
    hipLaunchKernelGGL(k_commit, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, s->y, s->xb[s->cur],
                       s->xb[(s->cur + 2) % 3], beta, (int)nesterov, s->n);
    ZF_HIP(hipGetLastError());
    return ZF_OK;
}

// which = 0: x_k, 1: y, 2: x+, 3: x_{k-1}
extern "C" int zf_mo_get(zf_mo* s, int32_t which, double* host, int64_t count) {
    ZF_REQUIRE(s && host, "zf_mo_get: null argument");
    ZF_REQUIRE(count >= s->n, "zf_mo_get: the host buffer holds fewer than n doubles");
    if (int rc = mo_flush(s)) return rc;
    const double* x = mo_which(s, which);
    ZF_REQUIRE(x, "zf_mo_get: bad point selector");
    ZF_HIP(hipMemcpyAsync(host, x, sizeof(double) * s->n, hipMemcpyDeviceToHost, s->stream));
    ZF_HIP(hipStreamSynchronize(s->stream));
    return ZF_OK;
}

extern "C" int zf_mo_get_jac(zf_mo* s, double* J_host, int64_t count) {
    ZF_REQUIRE(s && J_host, "zf_mo_get_jac: null argument");
    ZF_REQUIRE(count >= (int64_t)s->m * s->n, "zf_mo_get_jac: the host buffer holds fewer than m * n doubles");
    if (int rc = mo_flush(s)) return rc;
    ZF_HIP(hipMemcpyAsync(J_host, s->J, sizeof(double) * s->m * s->n, hipMemcpyDeviceToHost, s->stream));
    ZF_HIP(hipStreamSynchronize(s->stream));
    return ZF_OK;
}

// upload a host point into slot which = 1 (y) or 2 (x+): used when a host callback produced it
extern "C" int zf_mo_put(zf_mo* s, int32_t which, const double* host) {
    ZF_REQUIRE(s && host && (which == 1 || which == 2 || which == 0), "zf_mo_put: bad argument");
    if (int rc = mo_flush(s)) return rc;
    ZF_HIP(hipMemcpyAsync(mo_which(s, which), host, sizeof(double) * s->n, hipMemcpyHostToDevice, s->stream));
    ZF_HIP(hipStreamSynchronize(s->stream));
    return ZF_OK;
}

// prox_wsum_g(weight, x) at a host point (callback contract of Problem.prox_wsum_g)
extern "C" int zf_mo_prox_host(zf_mo* s, const double* weight_host, const double* x_host, double* out_host) {
    ZF_REQUIRE(s && weight_host && x_host && out_host, "zf_mo_prox_host: null argument");
    mo_w W;
    mo_fill_w(s, 1.0, weight_host, &W);                  // coef = weight * l1_ratios
    double* tmp_in = s->xb[(s->cur + 1) % 3];            // x+ slot is scratch between trials
    ZF_HIP(hipMemcpyAsync(tmp_in, x_host, sizeof(double) * s->n, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(k_prox_only, dim3(s->grid), dim3(ZF_BLOCK), 0, s->stream, tmp_in, s->G, W, s->n);
    ZF_HIP(hipGetLastError());
    ZF_HIP(hipMemcpyAsync(out_host, tmp_in, sizeof(double) * s->n, hipMemcpyDeviceToHost, s->stream));
    ZF_HIP(hipStreamSynchronize(s->stream));
    return ZF_OK;
}

// ---- generic kind: terms that follow a host prox callback ---------------------------------
// p (host) = prox_wsum_g(lr w, v) was computed by the user's callback; returns
// out[0..m) = J_i . (p - y)   and   out[m] = |p - (y - lr w@J)|^2   (proximal_gradient.py:168,173)
namespace {
template <int M>
__global__ __launch_bounds__(ZF_BLOCK) void k_post_terms(const double* __restrict__ J, const double* __restrict__ y,
                                                         const double* __restrict__ p, mo_w W, int64_t n,
                                                         double* partials) {
    __shared__ double lds[ZF_WAVES * (M + 1)];
    double acc[M + 1];
#pragma unroll
    for (int k = 0; k <= M; ++k) acc[k] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * ZF_BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * ZF_BLOCK + threadIdx.x; j < n; j += stride) {
        double Jc[M];
        double wJ = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            Jc[i] = J[(int64_t)i * n + j];
            wJ += W.w[i] * Jc[i];
        }
        const double yj = y[j], pj = p[j];
        const double v = yj - W.lr * wJ;
        const double dy = pj - yj, dv = pj - v;
#pragma unroll
        for (int i = 0; i < M; ++i) acc[i] += Jc[i] * dy;
        acc[M] += dv * dv;
    }
    const double maxs[1] = {0.0};
    double out = 0.0;
    zf_block_reduce<M + 1, 0, ZF_WAVES>(acc, maxs, lds, out);
    if (threadIdx.x <= M) partials[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = out;
}
}  // namespace

extern "C" int zf_mo_post_terms(zf_mo* s, double lr, const double* w_host, const double* p_host, double* out) {
    ZF_REQUIRE(s && w_host && p_host && out, "zf_mo_post_terms: null argument");
    if (int rc = mo_flush(s)) return rc;
    mo_w W;
    mo_fill_w(s, lr, w_host, &W);
    double* pd = s->xb[(s->cur + 1) % 3];
    ZF_HIP(hipMemcpyAsync(pd, p_host, sizeof(double) * s->n, hipMemcpyHostToDevice, s->stream));
    const int m = s->m;
    MO_LAUNCH(k_post_terms, m, s->J, s->y, pd, W, s->n, s->partials);
    return mo_reduce_to_host(s, m + 1, -1, out);
}
