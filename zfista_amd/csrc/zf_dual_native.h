// zf_dual_native.h - the library's own solver of the m-dimensional dual of a multi-objective
// trial, host C++ driving the fused dual-evaluation kernel.
//
// The dual  D(w) = -(min over x of the scalarised model)  of zfista/proximal_gradient.py:161-177
// is convex, C^1 and - for the shifted-l1 + box family of zfista/problems.py:101-138 - piecewise
// quadratic on the unit simplex.  The reference minimises it with scipy.optimize (:179-205);
// this is the opt-in replacement (ZF_DUAL_SOLVER=native), the C++ counterpart of
// zfista_amd/multiobjective.py::solve_dual_native (kept there for opaque Python callbacks):
//   m = 2 : the derivative along w = (s, 1 - s) is monotone and piecewise linear: a bracketing
//           root finder (Illinois-modified regula falsi, exact on a linear piece);
//   m >= 3: projected Newton on the simplex - gradient from one fused evaluation, curvature
//           from m further evaluations along the feasible directions e_i - w, the m-variable
//           QP on the simplex solved exactly by support enumeration, Armijo backtracking.
// One evaluation = one k_dual_eval launch + one small reduce + a pinned 2m+2-double read-back;
// no Python, NumPy or SciPy call in between.
#pragma once
#include <math.h>
#include <string.h>

namespace zf_dual {

constexpr int MAXM = 8;

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

// ---- tiny dense helpers ---------------------------------------------------------------------
// solve K x = rhs (k <= MAXM + 1) by Gaussian elimination with partial pivoting; false if singular
inline bool solve_small(int k, double K[MAXM + 1][MAXM + 1], double* rhs, double* x) {
    int piv[MAXM + 1];
    for (int i = 0; i < k; ++i) piv[i] = i;
    for (int c = 0; c < k; ++c) {
        int p = c;
        for (int r = c + 1; r < k; ++r)
            if (fabs(K[r][c]) > fabs(K[p][c])) p = r;
        if (fabs(K[p][c]) < 1e-300) return false;
        if (p != c) {
            for (int j = 0; j < k; ++j) {
                const double t = K[c][j];
                K[c][j] = K[p][j];
                K[p][j] = t;
            }
            const double t = rhs[c];
            rhs[c] = rhs[p];
            rhs[p] = t;
        }
        for (int r = c + 1; r < k; ++r) {
            const double f = K[r][c] / K[c][c];
            if (f == 0.0) continue;
            for (int j = c; j < k; ++j) K[r][j] -= f * K[c][j];
            rhs[r] -= f * rhs[c];
        }
    }
    for (int r = k - 1; r >= 0; --r) {
        double t = rhs[r];
        for (int j = r + 1; j < k; ++j) t -= K[r][j] * x[j];
        x[r] = t / K[r][r];
    }
    return true;
}

// smallest eigenvalue of a symmetric m x m matrix (cyclic Jacobi; m <= MAXM)
inline double min_eigenvalue(int m, const double Qin[MAXM][MAXM]) {
    double A[MAXM][MAXM];
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) A[i][j] = Qin[i][j];
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int i = 0; i < m; ++i)
            for (int j = i + 1; j < m; ++j) off += A[i][j] * A[i][j];
        if (off < 1e-300) break;
        for (int p = 0; p < m; ++p)
            for (int q = p + 1; q < m; ++q) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < m; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < m; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
            }
    }
    double lo = A[0][0];
    for (int i = 1; i < m; ++i) lo = A[i][i] < lo ? A[i][i] : lo;
    return lo;
}

// argmin over the unit simplex of q.w + 1/2 w'Qw, m <= MAXM, by enumerating supports (KKT check)
inline void simplex_qp(int m, const double* q, const double Q[MAXM][MAXM], double* best) {
    double best_val = INFINITY;
    bool have = false;
    for (int mask = 1; mask < (1 << m); ++mask) {
        int S[MAXM], k = 0;
        for (int i = 0; i < m; ++i)
            if (mask >> i & 1) S[k++] = i;
        double K[MAXM + 1][MAXM + 1], rhs[MAXM + 1], sol[MAXM + 1];
        for (int a = 0; a < k; ++a) {
            for (int b = 0; b < k; ++b) K[a][b] = Q[S[a]][S[b]];
            K[a][k] = 1.0;
            K[k][a] = 1.0;
            rhs[a] = -q[S[a]];
        }
        K[k][k] = 0.0;
        rhs[k] = 1.0;
        if (!solve_small(k + 1, K, rhs, sol)) continue;
        bool feasible = true;
        for (int a = 0; a < k; ++a)
            if (sol[a] < -1e-14) feasible = false;
        if (!feasible) continue;
        double w[MAXM] = {0}, sum = 0.0;
        for (int a = 0; a < k; ++a) {
            w[S[a]] = sol[a] > 0.0 ? sol[a] : 0.0;
            sum += w[S[a]];
        }
        if (!(sum > 0.0)) continue;
        for (int i = 0; i < m; ++i) w[i] /= sum;
        const double mu = sol[k];
        double red[MAXM], redmax = 0.0;
        for (int i = 0; i < m; ++i) {
            double t = q[i] + mu;
            for (int j = 0; j < m; ++j) t += Q[i][j] * w[j];
            red[i] = t;
            redmax = fabs(t) > redmax ? fabs(t) : redmax;
        }
        bool kkt = true;
        for (int i = 0; i < m; ++i)
            if (!(mask >> i & 1) && red[i] < -1e-10 * (1.0 + redmax)) kkt = false;
        if (!kkt) continue;
        double val = 0.0;
        for (int i = 0; i < m; ++i) {
            double t = 0.0;
            for (int j = 0; j < m; ++j) t += Q[i][j] * w[j];
            val += q[i] * w[i] + 0.5 * w[i] * t;
        }
        if (val < best_val) {
            best_val = val;
            have = true;
            for (int i = 0; i < m; ++i) best[i] = w[i];
        }
    }
    if (!have) {   // numerically degenerate: the best vertex
        int v = 0;
        double bv = INFINITY;
        for (int i = 0; i < m; ++i) {
            const double t = q[i] + 0.5 * Q[i][i];
            if (t < bv) {
                bv = t;
                v = i;
            }
        }
        for (int i = 0; i < m; ++i) best[i] = (i == v) ? 1.0 : 0.0;
    }
}

// m = 2.  Returns 0 ok; w[2], *fun, *nit filled.
inline int solve_1d(evaluator& E, double tol, long max_iter, double* w, double* fun, long* nit) {
    double jac[2], wa[2] = {0.0, 1.0}, wb[2] = {1.0, 0.0};
    double fa, fb, pa, pb;
    if (E.eval(wa, &fa, jac)) return -1;
    pa = jac[0] - jac[1];
    if (pa >= 0.0) {
        w[0] = 0.0, w[1] = 1.0, *fun = fa, *nit = 1;
        return 0;
    }
    if (E.eval(wb, &fb, jac)) return -1;
    pb = jac[0] - jac[1];
    if (pb <= 0.0) {
        w[0] = 1.0, w[1] = 0.0, *fun = fb, *nit = 2;
        return 0;
    }
    double a = 0.0, b = 1.0, s = 0.5, fs = fa;
    int side = 0;
    long it = 2;
    for (it = 3; it < max_iter + 3; ++it) {
        s = (a * pb - b * pa) / (pb - pa);   // secant point of the bracket
        if (!(a < s && s < b)) s = 0.5 * (a + b);
        double ws[2] = {s, 1.0 - s}, ps;
        if (E.eval(ws, &fs, jac)) return -1;
        ps = jac[0] - jac[1];
        if (ps == 0.0 || (b - a) <= tol) break;
        if (ps < 0.0) {
            a = s, pa = ps;
            if (side == -1) pb *= 0.5;   // Illinois: halve the stale end
            side = -1;
        } else {
            b = s, pb = ps;
            if (side == 1) pa *= 0.5;
            side = 1;
        }
        if ((b - a) <= tol) {
            s = 0.5 * (a + b);
            double wm[2] = {s, 1.0 - s};
            if (E.eval(wm, &fs, jac)) return -1;
            break;
        }
    }
    w[0] = s, w[1] = 1.0 - s, *fun = fs, *nit = it;
    return 0;
}

// General m.  *ok = 0 when the start point is not finite (caller falls back to the reference's
// SciPy calls, e.g. F(x_k) = inf).  Returns 0 unless an evaluation failed.
inline int solve(evaluator& E, int m, const double* w0, double tol, long max_iter, double* w, double* fun_out,
                 long* nit_out, int* ok) {
    *ok = 1;
    double sum = 0.0;
    for (int i = 0; i < m; ++i) {
        w[i] = w0 ? (w0[i] > 0.0 ? w0[i] : 0.0) : 1.0 / m;
        sum += w[i];
    }
    for (int i = 0; i < m; ++i) w[i] /= sum;
    double fun, grad[MAXM];
    if (E.eval(w, &fun, grad)) return -1;
    bool finite = isfinite(fun);
    for (int i = 0; i < m; ++i) finite = finite && isfinite(grad[i]);
    if (!finite) {
        *ok = 0;
        return 0;
    }
    if (m == 2) return solve_1d(E, tol, max_iter, w, fun_out, nit_out);
    long nit = 0;
    double h = 1e-5;
    for (nit = 1; nit <= max_iter; ++nit) {
        // curvature on the tangent space: (grad(w + h (e_i - w)) - grad(w)) / h = H (e_i - w)
        double T[MAXM][MAXM], HT[MAXM][MAXM], Q[MAXM][MAXM];
        for (int r = 0; r < m; ++r)
            for (int i = 0; i < m; ++i) T[r][i] = (r == i ? 1.0 : 0.0) - w[r];   // column i = e_i - w
        for (int i = 0; i < m; ++i) {
            double wi[MAXM], fi, gi[MAXM];
            for (int r = 0; r < m; ++r) wi[r] = w[r] + h * T[r][i];
            if (E.eval(wi, &fi, gi)) return -1;
            for (int r = 0; r < m; ++r) HT[r][i] = (gi[r] - grad[r]) / h;
        }
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) {
                double t = 0.0;
                for (int r = 0; r < m; ++r) t += T[r][i] * HT[r][j];
                Q[i][j] = t;
            }
        for (int i = 0; i < m; ++i)
            for (int j = i + 1; j < m; ++j) Q[i][j] = Q[j][i] = 0.5 * (Q[i][j] + Q[j][i]);
        const double ev = min_eigenvalue(m, Q);
        if (ev < 0.0)   // keep the model convex against finite-difference noise
            for (int i = 0; i < m; ++i) Q[i][i] += 1e-12 - ev;
        double q[MAXM], w_new[MAXM], d[MAXM];
        for (int i = 0; i < m; ++i) {
            double t = 0.0;
            for (int r = 0; r < m; ++r) t += T[r][i] * grad[r];
            q[i] = t;
        }
        simplex_qp(m, q, Q, w_new);
        double step = 0.0, slope = 0.0;
        for (int i = 0; i < m; ++i) {
            d[i] = w_new[i] - w[i];
            step = fabs(d[i]) > step ? fabs(d[i]) : step;
            slope += grad[i] * d[i];
        }
        // stop at `tol` in w, or when the model predicts no decrease resolvable in double precision
        if (step <= tol || slope >= -4e-16 * (fabs(fun) > 1.0 ? fabs(fun) : 1.0)) break;
        double t = 1.0, f_try, g_try[MAXM], wt[MAXM];
        for (;;) {
            for (int i = 0; i < m; ++i) wt[i] = w[i] + t * d[i];
            if (E.eval(wt, &f_try, g_try)) return -1;
            if (f_try <= fun + 1e-4 * t * slope + 1e-15 * fabs(fun) || t < 1e-10) break;
            t *= 0.5;
        }
        bool moved = false;
        sum = 0.0;
        double wn[MAXM];
        for (int i = 0; i < m; ++i) {
            wn[i] = wt[i] > 0.0 ? wt[i] : 0.0;
            sum += wn[i];
        }
        for (int i = 0; i < m; ++i) {
            wn[i] /= sum;
            if (wn[i] != wt[i]) moved = true;
            w[i] = wn[i];
        }
        if (t < 1.0 || moved) {
            if (E.eval(w, &fun, grad)) return -1;
        } else {
            fun = f_try;
            for (int i = 0; i < m; ++i) grad[i] = g_try[i];
        }
        if (t * step <= tol) break;
        // the finite-difference step follows the Newton step: curvature of the quadratic piece
        // the iterate sits in
        h = 0.1 * t * step;
        h = h < 1e-7 ? 1e-7 : (h > 1e-5 ? 1e-5 : h);
    }
    *fun_out = fun;
    *nit_out = nit > max_iter ? max_iter : nit;
    return 0;
}

}  // namespace zf_dual
