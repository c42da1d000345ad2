// zf_dual_native.h - the library's own solver of the m-dimensional dual of a multi-objective
// trial.
//
// The dual  D(w) = -(min over x of the scalarised model)  of zfista/proximal_gradient.py:161-177
// is convex, C^1 and - for the shifted-l1 + box family of zfista/problems.py:101-138 - piecewise
// quadratic on the unit simplex.  The reference minimises it with scipy.optimize (:179-205);
// this is the opt-in replacement (dual_solver="native" / "device"), the C++ counterpart of
// zfista_amd/multiobjective.py::solve_dual_native (kept there for opaque Python callbacks):
//   m = 2 : the derivative along w = (s, 1 - s) is monotone and piecewise linear: a bracketing
//           root finder (Illinois-modified regula falsi, exact on a linear piece);
//   m >= 3: projected Newton on the simplex - gradient from one fused evaluation, curvature
//           from m further evaluations along the feasible directions e_i - w, the m-variable
//           QP on the simplex solved exactly by support enumeration, Armijo backtracking.
//
// The algorithm is written as a STATE MACHINE (zf_dual::machine): `start()` / `advance()` hand
// out a batch of points to evaluate and consume the values, with no call-backs and no recursion,
// so the very same code runs
//   * on the host, driven by zf_dual::solve() with one kernel launch per evaluation
//     (dual_solver="native": zf_mo_solve_dual), and
//   * on the device, inside ONE persistent kernel per trial whose workgroups evaluate the batch on
//     register-resident data, combine the sums through a last-arriver reduction and let one lane
//     advance the machine (dual_solver="device": zf_mo_solve_dual_device, zf_multiobj.hip).
// Points that do not depend on each other are requested together - the m curvature probes of a
// Newton step, the first two step lengths of its line search - because on the device a batch
// costs one grid-wide hand-over whatever its size.  The machine is a template on m with constant
// indices throughout: on the device it lives in registers (no scratch, no LDS round trips).
#pragma once
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define ZF_DHD __host__ __device__
#define ZF_DHD_INLINE __host__ __device__ __attribute__((always_inline))   // (a device call needs a stack frame)
#else
#define ZF_DHD
#define ZF_DHD_INLINE
#endif

namespace zf_dual {

constexpr int MAXM = 8;
constexpr int MAXB = MAXM;      // points per batch (the m curvature probes are the largest request)
constexpr int LS_BATCH = 2;     // step lengths tried together in the line search: t, t / 2

// callable: evaluates fun and jac[m] at w[m]; returns 0 on success
struct evaluator {
    int (*fn)(void* ctx, const double* w, double* fun, double* jac);
    void* ctx;
    int evals;
    int eval(const double* w, double* fun, double* jac) {
        ++evals;
        return fn(ctx, w, fun, jac);
    }
};

// M = number of objectives, a compile-time constant: every loop below has a constant trip count and
// every array index is a constant after unrolling, so on the device the whole object lives in the
// registers of the one lane that advances it (dynamically indexed members would be sent to scratch
// or LDS: measured 80 us per Newton step for m = 3 against ~2 us of arithmetic).
// state of the m = 2 root finder; empty for m >= 3 (the machine is copied LDS -> registers -> LDS around
// every step on the device: fields that are never used would still occupy registers all along)
template <bool ON>
struct bracket_state {
    double a, b, pa, pb, s, fs;
    int side;
};
template <>
struct bracket_state<false> {};

// EXACT_H (m >= 3): the caller hands in, with the value and the gradient of every point, the generalised
// HESSIAN of the dual there (the dual of the shifted-l1 + box family is piecewise quadratic: its Hessian on
// the piece a point sits in comes out of the same pass over the data as the gradient, zf_multiobj.hip).
// The Newton model then needs no finite-difference probes: a Newton iteration is ONE batch (its line search,
// whose accepted point brings its own Hessian) instead of two, and m evaluations fewer.
template <int M, bool EXACT_H = false>
struct machine {
    static_assert(M >= 2 && M <= MAXM, "2 <= m <= 8");
    static constexpr bool XH = EXACT_H && M >= 3;
    enum { P_INIT, P_ENDS, P_BRACKET, P_FINAL, P_CURV, P_LS, P_REEVAL, P_DONE };
    static constexpr int N = M + 1;   // KKT systems of the simplex QP
    // largest batch: the start point together with its M curvature probes (m >= 3; the probes do not
    // depend on the values at the start point, and on the device a batch costs one grid-wide hand-over
    // whatever its size)
    static constexpr int NB = (M >= 3 && !XH) ? M + 1 : LS_BATCH;
    // problem
    double tol;
    long max_iter;
    // request: npts points to evaluate
    int npts;
    double pts[NB][M];
    // result (valid once done)
    int phase, ok;
    long nit;
    double w[M], fun;
    // state
    double grad[M], h, d[M], step, slope, t_base, t_acc;
    bracket_state<M == 2> br;     // m = 2 bracket
    double T[M][M];               // column i = e_i - w (directions of the curvature probes)

    ZF_DHD bool done() const { return phase == P_DONE; }

    // ---- tiny dense helpers (constant indices only) ----------------------------------------
    // solve K x = rhs (N x N) by Gaussian elimination; the pivot of column c is brought up by
    // compare-and-swap against every row below (the largest entry ends in row c); false if singular
    ZF_DHD_INLINE static bool solve_kkt(double (&K)[N][N], double (&rhs)[N], double (&sol)[N]) {
        bool okay = true;
#pragma unroll
        for (int c = 0; c < N; ++c) {
#pragma unroll
            for (int r = c + 1; r < N; ++r) {
                const bool sw = fabs(K[r][c]) > fabs(K[c][c]);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const double x = K[c][j], y = K[r][j];
                    K[c][j] = sw ? y : x;
                    K[r][j] = sw ? x : y;
                }
                const double x = rhs[c], y = rhs[r];
                rhs[c] = sw ? y : x;
                rhs[r] = sw ? x : y;
            }
            if (fabs(K[c][c]) < 1e-300) okay = false;
            const double piv = okay ? K[c][c] : 1.0;
#pragma unroll
            for (int r = c + 1; r < N; ++r) {
                const double f = K[r][c] / piv;
#pragma unroll
                for (int j = c; j < N; ++j) K[r][j] -= f * K[c][j];
                rhs[r] -= f * rhs[c];
            }
        }
#pragma unroll
        for (int r = N - 1; r >= 0; --r) {
            double t = rhs[r];
#pragma unroll
            for (int j = r + 1; j < N; ++j) t -= K[r][j] * sol[j];
            sol[r] = t / (okay ? K[r][r] : 1.0);
        }
        return okay;
    }

    // smallest eigenvalue of the symmetric matrix Q (cyclic Jacobi on a copy)
    ZF_DHD_INLINE static double min_eigenvalue(const double (&Q)[M][M]) {
        double A[M][M];
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) A[i][j] = Q[i][j];
        for (int sweep = 0; sweep < 60; ++sweep) {
            // converged when the off-diagonal part is at rounding level of the diagonal (in floating
            // point it never gets smaller: an absolute test alone runs all 60 sweeps - microseconds
            // on the host, 150 000 cycles per Newton step on one GPU lane)
            double off = 0.0, dia = 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                dia += A[i][i] * A[i][i];
#pragma unroll
                for (int j = i + 1; j < M; ++j) off += A[i][j] * A[i][j];
            }
            if (off <= 1e-30 * dia || off < 1e-300) break;
#pragma unroll
            for (int p = 0; p < M; ++p)
#pragma unroll
                for (int r = p + 1; r < M; ++r) {
                    const bool rot = A[p][r] != 0.0;
                    const double apr = rot ? A[p][r] : 1.0;
                    const double theta = (A[r][r] - A[p][p]) / (2.0 * apr);
                    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double c = rot ? 1.0 / sqrt(t * t + 1.0) : 1.0, sn = rot ? t * c : 0.0;
#pragma unroll
                    for (int k = 0; k < M; ++k) {
                        const double akp = A[k][p], akq = A[k][r];
                        A[k][p] = c * akp - sn * akq;
                        A[k][r] = sn * akp + c * akq;
                    }
#pragma unroll
                    for (int k = 0; k < M; ++k) {
                        const double apk = A[p][k], aqk = A[r][k];
                        A[p][k] = c * apk - sn * aqk;
                        A[r][k] = sn * apk + c * aqk;
                    }
                }
        }
        double lo = A[0][0];
#pragma unroll
        for (int i = 1; i < M; ++i) lo = A[i][i] < lo ? A[i][i] : lo;
        return lo;
    }

    // w_new = argmin over the unit simplex of q.w + 1/2 w'Qw by enumerating supports (KKT check).
    // For a support S the KKT system is written at full size - rows outside S pin w_i = 0 - so
    // that every index is a constant.  qp_candidate examines ONE support: false if it yields no
    // KKT point, else the point and its model value.
    ZF_DHD_INLINE static bool qp_candidate(const double (&q)[M], const double (&Q)[M][M], const int mask,
                                           double (&wq)[M], double& val) {
        double K[N][N], rhs[N], sol[N];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const bool in_i = (mask >> i) & 1;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const bool in_j = (mask >> j) & 1;
                K[i][j] = in_i ? (in_j ? Q[i][j] : 0.0) : (i == j ? 1.0 : 0.0);
            }
            K[i][M] = in_i ? 1.0 : 0.0;
            K[M][i] = in_i ? 1.0 : 0.0;
            rhs[i] = in_i ? -q[i] : 0.0;
        }
        K[M][M] = 0.0;
        rhs[M] = 1.0;
        bool good = solve_kkt(K, rhs, sol);
#pragma unroll
        for (int i = 0; i < M; ++i)
            if (((mask >> i) & 1) && sol[i] < -1e-14) good = false;   // infeasible
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            wq[i] = (((mask >> i) & 1) && sol[i] > 0.0) ? sol[i] : 0.0;
            sum += wq[i];
        }
        if (!(sum > 0.0)) good = false;
        const double inv = good ? sum : 1.0;
#pragma unroll
        for (int i = 0; i < M; ++i) wq[i] /= inv;
        const double mu = sol[M];
        double red[M], redmax = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double t = q[i] + mu;
#pragma unroll
            for (int j = 0; j < M; ++j) t += Q[i][j] * wq[j];
            red[i] = t;
            redmax = fabs(t) > redmax ? fabs(t) : redmax;
        }
#pragma unroll
        for (int i = 0; i < M; ++i)
            if (!((mask >> i) & 1) && red[i] < -1e-10 * (1.0 + redmax)) good = false;   // not a KKT point
        val = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < M; ++j) t += Q[i][j] * wq[j];
            val += q[i] * wq[i] + 0.5 * wq[i] * t;
        }
        return good;
    }

#if defined(__HIP_DEVICE_COMPILE__)
    ZF_DHD_INLINE static double lane_value(double v, int k) {   // v of lane k (k wave-uniform), in every lane
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
        return __hiloint2double(hi, lo);
    }
#endif

    ZF_DHD_INLINE static void simplex_qp(const double (&q)[M], const double (&Q)[M][M], double (&w_new)[M]) {
        double best_val = INFINITY;
        bool have = false;
        constexpr int NSUP = (1 << M) - 1;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZF_QP_SEQ)
        // On the device the machine is advanced by ALL lanes of one wave on identical copies of the state
        // (mo_machine_step): lane l examines support l + 1 - the 2^M - 1 KKT systems are solved side by
        // side instead of one after the other (each is a few hundred dependent instructions) - and every
        // lane then walks the candidates in support order, exactly the host loop's selection.
        const int lane = (int)(threadIdx.x & 63u);
        for (int base = 0; base < NSUP; base += 64) {
            const int mask = base + lane + 1;
            const bool mine = mask <= NSUP;
            double wq[M], val;
            const bool good = qp_candidate(q, Q, mine ? mask : 1, wq, val) && mine;
            const int good_i = good ? 1 : 0;
            // (the walk runs on wave-uniform values - scalar registers - and only the winner's point is
            //  fetched: keeping a running w_new in vector registers beside wq spilled)
            int best_k = -1;
#pragma unroll
            for (int k = 0; k < (NSUP < 64 ? NSUP : 64); ++k) {
                if (base + k < NSUP) {
                    const bool good_k = __builtin_amdgcn_readlane(good_i, k) != 0;
                    const double val_k = lane_value(val, k);
                    if (good_k && val_k < best_val) {
                        best_val = val_k;
                        best_k = k;
                    }
                }
            }
            if (best_k >= 0) {
                have = true;
#pragma unroll
                for (int i = 0; i < M; ++i) w_new[i] = lane_value(wq[i], best_k);
            }
        }
#else
        for (int mask = 1; mask <= NSUP; ++mask) {
            double wq[M], val;
            if (!qp_candidate(q, Q, mask, wq, val)) continue;
            if (val < best_val) {
                best_val = val;
                have = true;
#pragma unroll
                for (int i = 0; i < M; ++i) w_new[i] = wq[i];
            }
        }
#endif
        if (!have) {   // numerically degenerate: the best vertex
            int v = 0;
            double bv = INFINITY;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const double t = q[i] + 0.5 * Q[i][i];
                if (t < bv) {
                    bv = t;
                    v = i;
                }
            }
#pragma unroll
            for (int i = 0; i < M; ++i) w_new[i] = (i == v) ? 1.0 : 0.0;
        }
    }

    // ---- requests ----------------------------------------------------------------------------
    // OFF = 1: the probes follow the point w itself in the same batch (the start, P_INIT)
    template <int OFF = 0>
    ZF_DHD_INLINE void request_curvature() {
        // curvature on the tangent space: (grad(w + h (e_i - w)) - grad(w)) / h = H (e_i - w)
#pragma unroll
        for (int r = 0; r < M; ++r)
#pragma unroll
            for (int i = 0; i < M; ++i) T[r][i] = (r == i ? 1.0 : 0.0) - w[r];   // column i = e_i - w
        npts = M + OFF;
        if constexpr (OFF == 1) {
#pragma unroll
            for (int r = 0; r < M; ++r) pts[0][r] = w[r];
        }
        if constexpr (OFF + M <= NB) {
#pragma unroll
            for (int i = 0; i < M; ++i)
#pragma unroll
                for (int r = 0; r < M; ++r) pts[OFF + i][r] = w[r] + h * T[r][i];
        }
        phase = OFF == 1 ? P_INIT : P_CURV;
    }
    ZF_DHD_INLINE void request_line_search(double t0) {
        t_base = t0;
        npts = LS_BATCH;
        double t = t0;
#pragma unroll
        for (int k = 0; k < LS_BATCH; ++k) {
#pragma unroll
            for (int i = 0; i < M; ++i) pts[k][i] = w[i] + t * d[i];
            t *= 0.5;
        }
        phase = P_LS;
    }
    // Acceptance of a step length t along d in the line search.  Armijo on the VALUES - or, where the values no
    // longer resolve the decrease, on the DERIVATIVE along d.  Near the optimum the dual is a difference of sums
    // of n terms: its value carries ~1e-12 |D| of rounding noise while a Newton step of length 1e-7 lowers it by
    // ~1e-8 - Armijo then rejects steps that the gradient (accurate to ~1e-13 of its own scale, and what the KKT
    // conditions are made of) shows to be right, and the search stalls five orders of magnitude short of what
    // SciPy's gradient-driven trust-constr reaches on the same problem (G11 quad3: measured).  The approximate
    // Wolfe conditions of Hager & Zhang (SIAM J. Optim. 16, 2005): phi(t) <= phi(0) + eps |phi(0)| and
    // sigma phi'(0) <= phi'(t) <= (2 delta - 1) phi'(0) - the slope along d has shrunk, without overshooting.
    // The value guard of the Wolfe branch (round 4; made to BIND in round 5 - as first written it held for every input:
    // under the slope window pred = 1/2 t (phi'(0) + phi'(t)) < 0, so |df - pred| > 1/2 |df| whenever df > 0).
    // The dual is convex along d, so with CONSISTENT values and gradients phi' is nondecreasing on [0, t] and
    //     phi(t) - phi(0) = int_0^t phi' <= t max(phi'(t), 0).
    // * df within the rounding noise of the values (1e-10 |D|): the values say nothing, the slopes decide.
    // * noise < df <= 2 t max(phi'(t), 0): a genuine rise - the step overshot a kink, the slope window alone would let
    //   it pass (|phi'(t)| < |phi'(0)| is all it asks) - REJECTED: the step is halved.
    // * df beyond that bound: the values CONTRADICT the gradients.  The reference's dual value and gradient (:161-177)
    //   are consistent only if prox_wsum_g is the exact prox of sum_i w_i g_i - and zfista/problems.py:126-138 COMPOSES
    //   soft-thresholds, which for several shifted l1 terms with different shifts is not that prox: along a Newton
    //   direction the value may rise by 4.5e-4 t while the gradient field says - 1.2e-6 t (measured: five quadratics +
    //   shifted l1, n = 2e4).  SciPy's trust-constr, which the reference calls with jac=True, converges to the zero of
    //   the GRADIENT field (KKT gap 1e-8 there); a value guard kept this search at a gap of 0.08.  The guard is waived
    //   and the slopes decide, as they do for SciPy.
    ZF_DHD_INLINE bool ls_accept(double f_t, const double (&g_t)[M], double t) const {
        if (f_t <= fun + 1e-4 * t * slope + 1e-15 * fabs(fun)) return true;
        double gbar = 0.0, dphi = 0.0;   // (mean-free, like `slope`: see newton_model)
#pragma unroll
        for (int i = 0; i < M; ++i) gbar += g_t[i];
        gbar /= M;
#pragma unroll
        for (int i = 0; i < M; ++i) dphi += (g_t[i] - gbar) * d[i];
        const double df = f_t - fun, noise = 1e-10 * fabs(fun);
        const double convex_bound = 2.0 * t * (dphi > 0.0 ? dphi : 0.0) + noise;
        const bool value_ok = df <= noise || df > convex_bound;
        return value_ok && dphi >= 0.9 * slope && dphi <= -(1.0 - 2e-4) * slope;
    }
    ZF_DHD_INLINE void request_bracket_point() {
        if constexpr (M == 2) {
            br.s = (br.a * br.pb - br.b * br.pa) / (br.pb - br.pa);   // secant point of the bracket
            if (!(br.a < br.s && br.s < br.b)) br.s = 0.5 * (br.a + br.b);
            npts = 1;
            pts[0][0] = br.s;
            pts[0][1] = 1.0 - br.s;
            phase = P_BRACKET;
        }
    }
    ZF_DHD_INLINE void finish(long nit_value) {
        nit = nit_value;
        npts = 0;
        phase = P_DONE;
    }
    // after an accepted Newton step: stop, or probe the curvature at the new point
    ZF_DHD_INLINE void after_step() {
        if (t_acc * step <= tol) return finish(nit > max_iter ? max_iter : nit);
        // the finite-difference step follows the Newton step: curvature of the quadratic piece
        // the iterate sits in
        h = 0.1 * t_acc * step;
        h = h < 1e-7 ? 1e-7 : (h > 1e-5 ? 1e-5 : h);
        nit += 1;
        if (nit > max_iter) return finish(max_iter);
        request_curvature();
    }

    // The Newton model from the gradients at the M probes jacs[OFF .. OFF + M) (finite-difference
    // curvature on the tangent space), the exact simplex QP, and the request for its line search.
    template <int OFF>
    ZF_DHD_INLINE void newton_step(const double (&jacs)[NB][M]) {
        double HT[M][M];
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int r = 0; r < M; ++r) HT[r][i] = (jacs[(OFF + i) % NB][r] - grad[r]) / h;
        newton_model(HT);
    }
    // exact-Hessian mode: H (m x m, flattened) is the generalised Hessian of the dual at w
    ZF_DHD_INLINE void newton_from_hessian(const double (&H)[M][M]) {
        double HT[M][M];
#pragma unroll
        for (int r = 0; r < M; ++r)
#pragma unroll
            for (int i = 0; i < M; ++i) T[r][i] = (r == i ? 1.0 : 0.0) - w[r];   // column i = e_i - w
#pragma unroll
        for (int r = 0; r < M; ++r)
#pragma unroll
            for (int i = 0; i < M; ++i) {
                double t = 0.0;
#pragma unroll
                for (int q2 = 0; q2 < M; ++q2) t += H[r][q2] * T[q2][i];
                HT[r][i] = t;                                                      // H (e_i - w)
            }
        newton_model(HT);
    }
    // HT = H T (columns: the curvature along e_i - w) -> the model on the tangent space, the QP, the line search
    ZF_DHD_INLINE void newton_model(const double (&HT)[M][M]) {
        double Q[M][M], q[M], w_new[M];
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) {
                double t = 0.0;
#pragma unroll
                for (int r = 0; r < M; ++r) t += T[r][i] * HT[r][j];
                Q[i][j] = t;
            }
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = i + 1; j < M; ++j) Q[i][j] = Q[j][i] = 0.5 * (Q[i][j] + Q[j][i]);
        const double ev = min_eigenvalue(Q);
        if (ev < 0.0)   // keep the model convex against finite-difference noise
#pragma unroll
            for (int i = 0; i < M; ++i) Q[i][i] += 1e-12 - ev;
        // The model is q . d + 1/2 d'Qd in the STEP d = w' - w; written in w' (what simplex_qp solves) its linear
        // term is q - Q w.  Analytically Q w = T'H (T w) = 0, but a probed Q satisfies that only to ~1e-8 |Q|:
        // leaving the term out moved every Newton point by ~1e-9 and the search stalled at a gradient residual
        // 1e5 x the one an exact tangent-space Newton step reaches (measured on G11's quad3 problems).  Both are
        // scaled to O(1) entries: the KKT systems couple Q with the constraint row of ones.
        double qs = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) qs = fabs(Q[i][j]) > qs ? fabs(Q[i][j]) : qs;
        qs = qs > 0.0 ? 1.0 / qs : 1.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double t = 0.0;
#pragma unroll
            for (int r = 0; r < M; ++r) t += T[r][i] * grad[r];
#pragma unroll
            for (int j = 0; j < M; ++j) t -= Q[i][j] * w[j];
            q[i] = t * qs;
            w_new[i] = 0.0;
        }
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) Q[i][j] *= qs;
        simplex_qp(q, Q, w_new);
        // slope = grad . d with the mean of the gradient taken out first: d sums to zero only up to rounding, and
        // the common part of the gradient (F_old - f_y and friends: 1e5 where the differences that matter are
        // 1e-4) times that rounding residue is as large as the true slope of the last Newton steps
        double gbar = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) gbar += grad[i];
        gbar /= M;
        step = 0.0, slope = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            d[i] = w_new[i] - w[i];
            step = fabs(d[i]) > step ? fabs(d[i]) : step;
            slope += (grad[i] - gbar) * d[i];
        }
        // stop at `tol` in w, or when the predicted decrease grad . d is below what the GRADIENT resolves: its
        // components are sums of n terms with ~4e-16 (|D| + |grad|) of rounding noise each, so along a step of
        // length `step` nothing below noise x step is information.  (Round 2 compared the slope with the noise of
        // the VALUE, 4e-16 |D|, and stopped at |dw| ~ 1e-7 where SciPy's gradient-driven search reaches 1e-13.)
        double gmax = fabs(fun) > 1.0 ? fabs(fun) : 1.0;
#pragma unroll
        for (int i = 0; i < M; ++i) gmax = fabs(grad[i]) > gmax ? fabs(grad[i]) : gmax;
        if (step <= tol || slope >= -4e-16 * gmax * step)
            return finish(nit > max_iter ? max_iter : nit);
        return request_line_search(1.0);
    }

    // ---- the machine -----------------------------------------------------------------------------
    // w0 may be NULL (uniform start)
    ZF_DHD_INLINE void start(const double* w0, double tol_, long max_iter_) {
        tol = tol_;
        max_iter = max_iter_;
        ok = 1;
        nit = 0;
        h = 1e-5;
        fun = 0.0;
        step = slope = t_base = t_acc = 0.0;
        if constexpr (M == 2) {
            br.a = br.b = br.pa = br.pb = br.s = br.fs = 0.0;
            br.side = 0;
        }
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            w[i] = w0 ? (w0[i] > 0.0 ? w0[i] : 0.0) : 1.0 / M;
            grad[i] = d[i] = 0.0;
            sum += w[i];
        }
#pragma unroll
        for (int i = 0; i < M; ++i) w[i] /= sum;
#pragma unroll
        for (int k = 0; k < NB; ++k)
#pragma unroll
            for (int i = 0; i < M; ++i) pts[k][i] = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r)
#pragma unroll
            for (int i = 0; i < M; ++i) T[r][i] = 0.0;
        npts = 1;
#pragma unroll
        for (int i = 0; i < M; ++i) pts[0][i] = w[i];
        phase = P_INIT;
        if constexpr (M >= 3 && !XH) request_curvature<1>();   // the start point and its curvature probes together
    }

    // exact-Hessian mode, first half of advance(): consumes the values of the batch; returns 1 / 2 with the
    // Hessian at the (new) w in Hn when a Newton model is due, 0 when a request was issued or the search ended
    ZF_DHD_INLINE int advance_pick(const double (&funs)[NB], const double (&jacs)[NB][M], const double* hess,
                                   double (&Hn)[M][M]) {
#pragma unroll
        for (int r = 0; r < M; ++r)
#pragma unroll
            for (int q2 = 0; q2 < M; ++q2) Hn[r][q2] = 0.0;
        if (phase == P_INIT || phase == P_REEVAL) {
            if (phase == P_INIT) {
                bool finite = isfinite(funs[0]);
#pragma unroll
                for (int i = 0; i < M; ++i) finite = finite && isfinite(jacs[0][i]);
                if (!finite) {   // e.g. F(x_k) = inf outside the box: not attempted
                    ok = 0;
                    finish(0);
                    return 0;
                }
                nit = 1;
            }
            const int todo = phase == P_INIT ? 1 : 2;
            fun = funs[0];
#pragma unroll
            for (int i = 0; i < M; ++i) grad[i] = jacs[0][i];
#pragma unroll
            for (int r = 0; r < M; ++r)
#pragma unroll
                for (int q2 = 0; q2 < M; ++q2) Hn[r][q2] = hess[r * M + q2];
            return todo;
        }
        if (phase != P_LS) return 0;
        double t = t_base;
        bool picked = false;
        double f_pick = 0.0, g_pick[M], p_pick[M];
#pragma unroll
        for (int i = 0; i < M; ++i) g_pick[i] = p_pick[i] = 0.0;
#pragma unroll
        for (int k = 0; k < LS_BATCH; ++k) {
            const bool take = !picked && (ls_accept(funs[k], jacs[k], t) || t < 1e-10);
            if (take) {
                picked = true;
                t_acc = t;
                f_pick = funs[k];
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    g_pick[i] = jacs[k][i];
                    p_pick[i] = pts[k][i];
                }
#pragma unroll
                for (int r = 0; r < M; ++r)
#pragma unroll
                    for (int q2 = 0; q2 < M; ++q2) Hn[r][q2] = hess[(k * M + r) * M + q2];
            }
            if (!picked) t *= 0.5;
        }
        if (!picked) {
            request_line_search(t);   // (t is already halved past the batch)
            return 0;
        }
        bool moved = false;
        double wq[M], sum = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            wq[i] = p_pick[i] > 0.0 ? p_pick[i] : 0.0;
            sum += wq[i];
        }
#pragma unroll
        for (int i = 0; i < M; ++i) {
            wq[i] /= sum;
            if (wq[i] != p_pick[i]) moved = true;
            w[i] = wq[i];
        }
        if (t_acc < 1.0 || moved) {
            npts = 1;
#pragma unroll
            for (int i = 0; i < M; ++i) pts[0][i] = w[i];
            phase = P_REEVAL;
            return 0;
        }
        fun = f_pick;
#pragma unroll
        for (int i = 0; i < M; ++i) grad[i] = g_pick[i];
        return 2;
    }

    // funs[k], jacs[k][0..M): the dual and its gradient at pts[k], k < npts
    // hess (exact-Hessian mode, else ignored / NULL): [NB][M][M] flattened, the Hessian of the dual at pts[k]
    ZF_DHD_INLINE void advance(const double (&funs)[NB], const double (&jacs)[NB][M], const double* hess = nullptr) {
        if constexpr (XH) {
            // exact-Hessian mode: the cases below only pick the point and its Hessian; the Newton model is
            // built at ONE place (three inlined copies of the QP and the eigenvalue sweep cost the device
            // kernel its register budget)
            double Hn[M][M];
            const int todo = advance_pick(funs, jacs, hess, Hn);   // 0: request issued / done, 1: first model, 2: after a step
            if (todo == 2) {
                if (t_acc * step <= tol) return finish(nit > max_iter ? max_iter : nit);
                nit += 1;
                if (nit > max_iter) return finish(max_iter);
            }
            if (todo) newton_from_hessian(Hn);
            return;
        }
        switch (phase) {
        case P_INIT: {
            bool finite = isfinite(funs[0]);
#pragma unroll
            for (int i = 0; i < M; ++i) finite = finite && isfinite(jacs[0][i]);
            if (!finite) {   // e.g. F(x_k) = inf outside the box: not attempted
                ok = 0;
                return finish(0);
            }
            fun = funs[0];
#pragma unroll
            for (int i = 0; i < M; ++i) grad[i] = jacs[0][i];
            if constexpr (M == 2) {   // both ends of the segment at once
                npts = 2;
                pts[0][0] = 0.0, pts[0][1] = 1.0;
                pts[1][0] = 1.0, pts[1][1] = 0.0;
                phase = P_ENDS;
                return;
            } else {
                nit = 1;
                return newton_step<1>(jacs);   // (m >= 3: the probes came with the start point)
            }
        }
        case P_ENDS: {
            if constexpr (M == 2) {
                br.pa = jacs[0][0] - jacs[0][1];
                br.pb = jacs[1][0] - jacs[1][1];
                if (br.pa >= 0.0) {
                    w[0] = 0.0, w[1] = 1.0, fun = funs[0];
                    return finish(1);
                }
                if (br.pb <= 0.0) {
                    w[0] = 1.0, w[1] = 0.0, fun = funs[1];
                    return finish(2);
                }
                br.a = 0.0, br.b = 1.0, br.side = 0, br.fs = funs[0];
                nit = 3;
                return request_bracket_point();
            }
            return;
        }
        case P_BRACKET: {
            if constexpr (M == 2) {
                br.fs = funs[0];
                const double ps = jacs[0][0] - jacs[0][1];
                const bool stop = (ps == 0.0 || (br.b - br.a) <= tol);
                if (!stop) {
                    // the new point replaces the end of its sign; Illinois: an end that stays for the
                    // second time in a row has its derivative halved.  (Written with selects: the
                    // compiler turned the branchy form into a dynamically indexed stack slot.)
                    const bool neg = ps < 0.0;
                    const double pa_kept = (br.side == 1) ? br.pa * 0.5 : br.pa;
                    const double pb_kept = (br.side == -1) ? br.pb * 0.5 : br.pb;
                    br.a = neg ? br.s : br.a;
                    br.b = neg ? br.b : br.s;
                    br.pa = neg ? ps : pa_kept;
                    br.pb = neg ? pb_kept : ps;
                    br.side = neg ? -1 : 1;
                    if ((br.b - br.a) <= tol) {
                        br.s = 0.5 * (br.a + br.b);
                        npts = 1;
                        pts[0][0] = br.s, pts[0][1] = 1.0 - br.s;
                        phase = P_FINAL;
                        return;
                    }
                    nit += 1;
                    if (nit < max_iter + 3) return request_bracket_point();
                }
                w[0] = br.s, w[1] = 1.0 - br.s, fun = br.fs;
                return finish(nit);
            }
            return;
        }
        case P_FINAL: {
            if constexpr (M == 2) {
                w[0] = br.s, w[1] = 1.0 - br.s, fun = funs[0];
                return finish(nit);
            }
            return;
        }
        case P_CURV:
            return newton_step<0>(jacs);
        case P_LS: {
            double t = t_base;
            bool picked = false;
            double f_pick = 0.0, g_pick[M], p_pick[M];
#pragma unroll
            for (int i = 0; i < M; ++i) g_pick[i] = p_pick[i] = 0.0;
#pragma unroll
            for (int k = 0; k < LS_BATCH; ++k) {
                const bool take = !picked && (ls_accept(funs[k], jacs[k], t) || t < 1e-10);
                if (take) {
                    picked = true;
                    t_acc = t;
                    f_pick = funs[k];
#pragma unroll
                    for (int i = 0; i < M; ++i) {
                        g_pick[i] = jacs[k][i];
                        p_pick[i] = pts[k][i];
                    }
                }
                if (!picked) t *= 0.5;
            }
            if (!picked) return request_line_search(t);   // (t is already halved past the batch)
            bool moved = false;
            double wq[M], sum = 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                wq[i] = p_pick[i] > 0.0 ? p_pick[i] : 0.0;
                sum += wq[i];
            }
#pragma unroll
            for (int i = 0; i < M; ++i) {
                wq[i] /= sum;
                if (wq[i] != p_pick[i]) moved = true;
                w[i] = wq[i];
            }
            if (t_acc < 1.0 || moved) {
                npts = 1;
#pragma unroll
                for (int i = 0; i < M; ++i) pts[0][i] = w[i];
                phase = P_REEVAL;
                return;
            }
            fun = f_pick;
#pragma unroll
            for (int i = 0; i < M; ++i) grad[i] = g_pick[i];
            return after_step();
        }
        case P_REEVAL: {
            fun = funs[0];
#pragma unroll
            for (int i = 0; i < M; ++i) grad[i] = jacs[0][i];
            return after_step();
        }
        default:
            return;
        }
    }
};

template <int M>
inline int solve_t(evaluator& E, const double* w0, double tol, long max_iter, double* w, double* fun_out,
                   long* nit_out, int* ok) {
    machine<M> S;
    S.start(w0, tol, max_iter);
    double funs[machine<M>::NB] = {0}, jacs[machine<M>::NB][M] = {{0}};
    while (!S.done()) {
        for (int k = 0; k < S.npts; ++k) {
            double jac[MAXM];
            if (E.eval(S.pts[k], &funs[k], jac)) return -1;
            for (int i = 0; i < M; ++i) jacs[k][i] = jac[i];
        }
        S.advance(funs, jacs);
    }
    *ok = S.ok;
    for (int i = 0; i < M; ++i) w[i] = S.w[i];
    *fun_out = S.fun;
    *nit_out = S.nit;
    return 0;
}

// Host driver: one evaluator call per requested point.  *ok = 0 when the start point is not
// finite (caller falls back to the reference's SciPy calls, e.g. F(x_k) = inf).  Returns 0 unless
// an evaluation failed.
inline int solve(evaluator& E, int m, const double* w0, double tol, long max_iter, double* w, double* fun_out,
                 long* nit_out, int* ok) {
    switch (m) {
        case 2: return solve_t<2>(E, w0, tol, max_iter, w, fun_out, nit_out, ok);
        case 3: return solve_t<3>(E, w0, tol, max_iter, w, fun_out, nit_out, ok);
        case 4: return solve_t<4>(E, w0, tol, max_iter, w, fun_out, nit_out, ok);
        case 5: return solve_t<5>(E, w0, tol, max_iter, w, fun_out, nit_out, ok);
        case 6: return solve_t<6>(E, w0, tol, max_iter, w, fun_out, nit_out, ok);
        case 7: return solve_t<7>(E, w0, tol, max_iter, w, fun_out, nit_out, ok);
        case 8: return solve_t<8>(E, w0, tol, max_iter, w, fun_out, nit_out, ok);
        default: return -1;
    }
}

// (tests) smallest eigenvalue of a symmetric m x m matrix through the machine's helper
template <int M>
inline double min_eigenvalue_t(const double Qin[MAXM][MAXM]) {
    double Q[M][M];
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j) Q[i][j] = Qin[i][j];
    return machine<M>::min_eigenvalue(Q);
}
inline double min_eigenvalue(int m, const double Qin[MAXM][MAXM]) {
    switch (m) {
        case 2: return min_eigenvalue_t<2>(Qin);
        case 3: return min_eigenvalue_t<3>(Qin);
        case 4: return min_eigenvalue_t<4>(Qin);
        case 5: return min_eigenvalue_t<5>(Qin);
        case 6: return min_eigenvalue_t<6>(Qin);
        case 7: return min_eigenvalue_t<7>(Qin);
        default: return min_eigenvalue_t<8>(Qin);
    }
}

}  // namespace zf_dual
