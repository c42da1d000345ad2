// Host-only harness around zfista_amd/csrc/zf_dual_native.h (plain C++, no HIP): minimises
//   D(w) = 1/2 w'Qw + q'w + sum_i kink_i * max(w_i - knot_i, 0)^2      over the unit simplex
// - convex, C^1, piecewise quadratic like the dual of a multi-objective trial - so that the
// library's dual solver can be checked on a machine without a GPU (tests/test_dual_native_cpp.py).
#include "../../zfista_amd/csrc/zf_dual_native.h"

namespace {
struct problem {
    int m;
    const double* Q;   // m x m row-major, symmetric positive semidefinite
    const double* q;
    const double* kink;
    const double* knot;
};
int eval(void* ctx, const double* w, double* fun, double* jac) {
    const problem* p = static_cast<const problem*>(ctx);
    double f = 0.0;
    for (int i = 0; i < p->m; ++i) {
        double t = 0.0;
        for (int j = 0; j < p->m; ++j) t += p->Q[i * p->m + j] * w[j];
        jac[i] = t + p->q[i];
        f += 0.5 * w[i] * t + p->q[i] * w[i];
        const double e = w[i] - p->knot[i];
        if (e > 0.0) {
            f += p->kink[i] * e * e;
            jac[i] += 2.0 * p->kink[i] * e;
        }
    }
    *fun = f;
    return 0;
}
}  // namespace

extern "C" int dual_native_solve(int m, const double* Q, const double* q, const double* kink, const double* knot,
                                 const double* w0, double tol, long max_iter, double* w, double* fun, long* nit,
                                 int* evals) {
    problem p = {m, Q, q, kink, knot};
    zf_dual::evaluator E = {eval, &p, 0};
    int ok = 1;
    const int rc = zf_dual::solve(E, m, w0, tol, max_iter, w, fun, nit, &ok);
    *evals = E.evals;
    return rc ? rc : (ok ? 0 : 1);
}

// The same problem through the EXACT-HESSIAN mode of the machine (zf_dual::machine<M, true>): every point
// comes with the Hessian of its quadratic piece, Q + 2 diag(kink_i [w_i > knot_i]) - no curvature probes.
namespace {
template <int M>
int solve_xh(const problem& p, const double* w0, double tol, long max_iter, double* w, double* fun, long* nit,
             int* evals, int* batches) {
    using mach = zf_dual::machine<M, true>;
    mach S;
    S.start(w0, tol, max_iter);
    double funs[mach::NB] = {0}, jacs[mach::NB][M] = {{0}}, hess[mach::NB * M * M] = {0};
    *evals = *batches = 0;
    while (!S.done()) {
        for (int k = 0; k < S.npts; ++k) {
            double jac[zf_dual::MAXM];
            eval(const_cast<problem*>(&p), S.pts[k], &funs[k], jac);
            for (int i = 0; i < M; ++i) {
                jacs[k][i] = jac[i];
                for (int j = 0; j < M; ++j)
                    hess[(k * M + i) * M + j] = p.Q[i * M + j] + ((i == j && S.pts[k][i] > p.knot[i]) ? 2.0 * p.kink[i] : 0.0);
            }
            *evals += 1;
        }
        *batches += 1;
        S.advance(funs, jacs, hess);
    }
    for (int i = 0; i < M; ++i) w[i] = S.w[i];
    *fun = S.fun;
    *nit = S.nit;
    return S.ok ? 0 : 1;
}
}  // namespace

extern "C" int dual_native_solve_exact_hessian(int m, const double* Q, const double* q, const double* kink,
                                               const double* knot, const double* w0, double tol, long max_iter,
                                               double* w, double* fun, long* nit, int* evals, int* batches) {
    problem p = {m, Q, q, kink, knot};
    switch (m) {
        case 3: return solve_xh<3>(p, w0, tol, max_iter, w, fun, nit, evals, batches);
        case 4: return solve_xh<4>(p, w0, tol, max_iter, w, fun, nit, evals, batches);
        case 5: return solve_xh<5>(p, w0, tol, max_iter, w, fun, nit, evals, batches);
        case 8: return solve_xh<8>(p, w0, tol, max_iter, w, fun, nit, evals, batches);
        default: return -1;
    }
}

extern "C" double dual_native_min_eig(int m, const double* Q) {
    double A[zf_dual::MAXM][zf_dual::MAXM];
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) A[i][j] = Q[i * m + j];
    return zf_dual::min_eigenvalue(m, A);
}

// The line-search acceptance rule alone (zf_dual::machine<3>::ls_accept) on a hand-built state: value `fun` and slope
// along `d` at the base point, value f_t and gradient g_t at step length t.  For the test of the value guard.
extern "C" int dual_native_ls_accept(double fun, double slope, const double* d, double f_t, const double* g_t, double t) {
    zf_dual::machine<3> S;
    const double w0[3] = {1.0 / 3, 1.0 / 3, 1.0 / 3};
    S.start(w0, 1e-12, 10);
    S.fun = fun;
    S.slope = slope;
    double g[3];
    for (int i = 0; i < 3; ++i) {
        S.d[i] = d[i];
        g[i] = g_t[i];
    }
    return S.ls_accept(f_t, g, t) ? 1 : 0;
}
