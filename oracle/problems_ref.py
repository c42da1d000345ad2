"""CPU oracle: NumPy restatement of the callback providers on the path.

TEST INFRASTRUCTURE ONLY (see ``oracle/cpu_ref.py``).

Restated from the reference (file:line relative to /root/reference):

* ``Problem.g``            zfista/problems.py:101-117   (shifted l1 values, box -> inf)
* ``Problem.prox_wsum_g``  zfista/problems.py:119-138   (composed soft-thresholds, box clip)
* ``JOS1.f / jac_f``       zfista/problems.py:193-205
* ``FDS.f / jac_f``        zfista/problems.py:309-328
* ``SD``, ``ZDT1``, ``TOI4``, ``TRIDIA``, ``LinearFunctionRank1`` ``f / jac_f``
                           zfista/problems.py:247-264, 368-386, 430-448, 494-514, 558-575
                           (SD pinned by tests/test_problems.py:45-74; the other four have no
                           known answers in the reference: parity unpinned, restated from the source
                           expressions and cross-checked by finite differences)
* test-LASSO closures      tests/test_proximal_gradient.py:49-61,81-97

Third-party arithmetic not under /root/reference: jaxopt (unpinned,
pyproject.toml:16).  ``jaxopt.prox.prox_lasso(x, t) = sign(x) * relu(|x| - t)``
and ``jaxopt.projection.projection_box(x, (lo, hi)) = clip(x, lo, hi)`` are
restated from their published definitions; the reference's own known answers at
that boundary (tests/test_problems.py:37-42,69-74,121-126 and the toy-LASSO
optima, tests/test_proximal_gradient.py:103-114) pin them and are checked in
``tests/test_oracle_golden.py``.

The single-objective operator families used by the benchmark configurations
(diagonal quadratic + l1, dense least squares + l1) have no class in the
reference; their NumPy expressions are *defined* here and the HIP engine
follows them term by term.
"""
from __future__ import annotations

import numpy as np


def soft_threshold(u, tau):
    """prox of tau*|.|: sign(u) * max(|u| - tau, 0)  (jaxopt.prox.prox_lasso)."""
    return np.sign(u) * np.maximum(np.abs(u) - tau, 0)


def clip_box(u, lo, hi):
    """jaxopt.projection.projection_box."""
    return np.clip(u, lo, hi)


# --------------------------------------------------------------------------
# Problem family of zfista/problems.py (multi-objective, shifted l1 + box)
# --------------------------------------------------------------------------
class ProblemRef:
    """g and prox_wsum_g shared by every reference problem (problems.py:63-138)."""

    def __init__(self, n_features, n_objectives, l1_ratios=None, l1_shifts=None, bounds=None):
        self.n_features = n_features
        self.n_objectives = n_objectives
        self.l1_ratios = None if l1_ratios is None else np.array(l1_ratios)
        self.l1_shifts = np.zeros(n_objectives) if l1_shifts is None else np.array(l1_shifts)
        self.bounds = bounds

    def g(self, x):
        if self.n_features != len(x):
            raise ValueError(f"len(x) should be equal to n_features, got {x}.")
        if self.bounds is not None:
            lo, hi = self.bounds
            if (x < lo).any() or (x > hi).any():
                return np.full(self.n_objectives, np.inf)
        if self.l1_ratios is None:
            return np.zeros(self.n_objectives)
        if self.n_objectives != len(self.l1_ratios):
            raise ValueError("len(l1_ratios) should be equal to n_objectives.")
        if self.n_objectives != len(self.l1_shifts):
            raise ValueError("len(l1_shifts) should be equal to n_objectives.")
        dist = np.linalg.norm(x - self.l1_shifts.reshape(-1, 1), ord=1, axis=1)
        return self.l1_ratios * dist

    def prox_wsum_g(self, weight, x):
        if self.n_features != len(x):
            raise ValueError(f"len(x) should be equal to n_features, got {x}.")
        if self.n_objectives != len(weight):
            raise ValueError("len(weight) should be equal to n_objectives.")
        if self.l1_ratios is not None:
            coef = weight * self.l1_ratios
            s = self.l1_shifts
            # stage 0 adds and subtracts s[0]: shift 0 is (net) ignored (:129)
            x = soft_threshold(x + np.sum(coef[1:]) - s[0] + s[0], coef[0])
            for i in range(1, self.n_objectives):
                x = soft_threshold(x - coef[i] - s[i], coef[i]) + s[i]
        if self.bounds is not None:
            x = clip_box(x, self.bounds[0], self.bounds[1])
        return x

    def callbacks(self):
        return self.f, self.g, self.jac_f, self.prox_wsum_g


class JOS1Ref(ProblemRef):
    def __init__(self, n_features=5, l1_ratios=None, l1_shifts=None, bounds=None):
        super().__init__(n_features, 2, l1_ratios, l1_shifts, bounds)

    def f(self, x):
        if self.n_features != len(x):
            raise ValueError(f"len(x) should be equal to n_features, got {x}.")
        n = self.n_features
        return np.array([np.linalg.norm(x) ** 2 / n, np.linalg.norm(x - 2) ** 2 / n])

    def jac_f(self, x):
        if self.n_features != len(x):
            raise ValueError(f"len(x) should be equal to n_features, got {x}.")
        n = self.n_features
        return np.vstack((2 * x / n, 2 * (x - 2) / n))


class FDSRef(ProblemRef):
    def __init__(self, n_features=10, l1_ratios=None, l1_shifts=None, bounds=None):
        super().__init__(n_features, 3, l1_ratios, l1_shifts, bounds)
        self.idx = np.arange(n_features) + 1          # one_to_n (:309)
        self.conv = self.idx * self.idx[::-1]          # i (n - i + 1) (:310)

    def f(self, x):
        if self.n_features != len(x):
            raise ValueError(f"len(x) should be equal to n_features, got {x}.")
        n = self.n_features
        f1 = np.inner(self.idx, (x - self.idx) ** 4) / n**2
        f2 = np.exp(x.sum() / n) + np.linalg.norm(x) ** 2
        f3 = np.inner(self.conv, np.exp(-x)) / (n * (n + 1))
        return np.array([f1, f2, f3])

    def jac_f(self, x):
        if self.n_features != len(x):
            raise ValueError(f"len(x) should be equal to n_features, got {x}.")
        n = self.n_features
        r1 = 4 / n**2 * self.idx * (x - self.idx) ** 3
        r2 = np.exp(x.sum() / n) / n + 2 * x
        r3 = -self.conv * np.exp(-x) / (n * (n + 1))
        return np.vstack((r1, r2, r3))


# --------------------------------------------------------------------------
# Single-objective operator families (benchmark configurations)
# --------------------------------------------------------------------------
class SDRef(ProblemRef):
    """Stadler-Dauer truss problem as the reference CODES it (zfista/problems.py:238-264):
    f_2 ends in 2 / x_4 (the docstring says x_4) and the box is (1e-6, inf)."""

    def __init__(self):
        super().__init__(4, 2, bounds=(1e-6, np.inf))

    def f(self, x):
        r2 = np.sqrt(2)
        return np.array([2 * x[0] + r2 * x[1] + r2 * x[2] + x[3],
                         2 / x[0] + 2 * r2 / x[1] + 2 * r2 / x[2] + 2 / x[3]])

    def jac_f(self, x):
        r2 = np.sqrt(2)
        return np.vstack((np.array([2, r2, r2, 1]),
                          np.array([-2 / x[0] ** 2, -2 * r2 / x[1] ** 2, -2 * r2 / x[2] ** 2, -2 / x[3] ** 2])))


class ZDT1Ref(ProblemRef):
    """zfista/problems.py:365-386.  The coded gradient of f_2 has 9 (2 - sqrt(x_1/h)) / (2 (n-1)) in
    its tail entries - the derivative (finite-difference checked in tests/test_problem_library.py);
    the class docstring's 1 - sqrt(...) is a typo.  The restatement follows the code."""

    def __init__(self, n_features=30):
        super().__init__(n_features, 2, bounds=(1e-6, np.inf))

    def _h(self, x):
        return 1 + 9 / (self.n_features - 1) * np.sum(x[1:])

    def f(self, x):
        h = self._h(x)
        return np.array([x[0], h * (1 - np.sqrt(x[0] / h))])

    def jac_f(self, x):
        n, h = self.n_features, self._h(x)
        j1 = np.zeros(n)
        j1[0] = 1
        j2 = np.full(n, 9 * (2 - np.sqrt(x[0] / h)) / 2 / (n - 1))
        j2[0] = -np.sqrt(h / x[0]) / 2
        return np.vstack((j1, j2))


class TOI4Ref(ProblemRef):
    """zfista/problems.py:415-448."""

    def __init__(self, l1_ratios=None, l1_shifts=None, bounds=None):
        super().__init__(4, 2, l1_ratios, l1_shifts, bounds)

    def f(self, x):
        return np.array([x[0] ** 2 + x[1] ** 2 + 1, 0.5 * ((x[0] - x[1]) ** 2 + (x[2] - x[3]) ** 2) + 1])

    def jac_f(self, x):
        a, b = x[0] - x[1], x[2] - x[3]
        return np.array([[2 * x[0], 2 * x[1], 0.0, 0.0], [a, -a, b, -b]])


class TRIDIARef(ProblemRef):
    """zfista/problems.py:479-514."""

    def __init__(self, l1_ratios=None, l1_shifts=None, bounds=None):
        super().__init__(3, 3, l1_ratios, l1_shifts, bounds)

    def f(self, x):
        return np.array([(2 * x[0] - 1) ** 2, 2 * (2 * x[0] - x[1]) ** 2, 3 * (2 * x[1] - x[2]) ** 2])

    def jac_f(self, x):
        return np.array([[8 * x[0] - 4, 0, 0],
                         [16 * x[0] - 8 * x[1], 4 * x[1] - 8 * x[0], 0],
                         [0, 24 * x[1] - 12 * x[2], 6 * x[2] - 12 * x[1]]], dtype=np.float64)


class LinearFunctionRank1Ref(ProblemRef):
    """zfista/problems.py:540-575: f_i = (i <j, x> - 1)^2 with j = 1..n."""

    def __init__(self, n_features=10, n_objectives=4, l1_ratios=None, l1_shifts=None, bounds=None):
        super().__init__(n_features, n_objectives, l1_ratios, l1_shifts, bounds)
        self.i = np.arange(1, n_objectives + 1)
        self.j = np.arange(1, n_features + 1)

    def f(self, x):
        return (self.i * np.inner(self.j, x) - 1) ** 2

    def jac_f(self, x):
        return 2 * self.i[:, None] * self.j * (self.i[:, None] * np.inner(self.j, x) - 1)


class DiagQuadL1Ref:
    """f(x) = 1/2 sum d_i (x_i - c_i)^2,  g(x) = lam ||x||_1   (P-diag)."""

    def __init__(self, d, c, lam):
        self.d, self.c, self.lam = np.asarray(d, float), np.asarray(c, float), float(lam)

    def f(self, x):
        r = x - self.c
        return 0.5 * np.sum(self.d * (r * r))

    def g(self, x):
        return self.lam * np.sum(np.abs(x))

    def jac_f(self, x):
        return self.d * (x - self.c)

    def f_diff(self, x_new, y):
        """f(x_new) - f(y) element by element, as a difference of squares (no cancellation of O(|f|) sums):
        1/2 d (rn^2 - r^2) = 1/2 (d dx)(rn + r).  For oracle.cpu_ref.minimize_proximal_gradient(..., f_diff=) - the
        checker of the engine's acceptance="resolved" (not part of the reference)."""
        r, rn = y - self.c, x_new - self.c
        return 0.5 * np.sum((self.d * (x_new - y)) * (rn + r))

    def prox_wsum_g(self, weight, x):
        return soft_threshold(x, self.lam * weight)

    def callbacks(self):
        return self.f, self.g, self.jac_f, self.prox_wsum_g


class LeastSquaresL1Ref:
    """f(x) = scale ||Ax - b||^2,  g(x) = lam ||x||_1   (P-lasso; scale 1/2).

    The test closures of the reference are the scale = 1/6 member
    (tests/test_proximal_gradient.py:81-97).
    """

    def __init__(self, A, b, lam, scale=0.5):
        self.A = np.asarray(A, float)
        self.b = np.asarray(b, float)
        self.lam, self.scale = float(lam), float(scale)

    def f(self, x):
        return self.scale * np.linalg.norm(self.A @ x - self.b) ** 2

    def g(self, x):
        return self.lam * np.linalg.norm(x, ord=1)

    def jac_f(self, x):
        return (2 * self.scale) * (self.A.T @ (self.A @ x - self.b))

    def prox_wsum_g(self, weight, x):
        return soft_threshold(x, self.lam * weight)

    def callbacks(self):
        return self.f, self.g, self.jac_f, self.prox_wsum_g


def stacked(problem, m):
    """The reference's m-fold duplicated test objectives
    (tests/test_proximal_gradient.py:128-149,181-202): every objective equal,
    prox uses weight.sum()."""

    def f(x):
        return np.full(m, problem.f(x))

    def g(x):
        return np.full(m, problem.g(x))

    def jac_f(x):
        return np.vstack([problem.jac_f(x)] * m)

    def prox_wsum_g(weight, x):
        return problem.prox_wsum_g(weight.sum(), x)

    return f, g, jac_f, prox_wsum_g


# --------------------------------------------------------------------------
# Synthetic inputs of SURVEY.md 8(d) (seeded; shared by tests and bench)
# --------------------------------------------------------------------------
def make_pdiag(n, seed=1, lam=0.1):
    rng = np.random.default_rng(seed)
    d = rng.uniform(0.5, 2.0, n)
    c = rng.standard_normal(n)
    return d, c, lam


def make_plasso(m_rows, n, seed=0, n_informative=20, noise=0.01, lam_frac=0.1):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m_rows, n))
    x_true = np.zeros(n)
    x_true[:n_informative] = rng.standard_normal(n_informative)
    b = A @ x_true + noise * rng.standard_normal(m_rows)
    lam = lam_frac * np.max(np.abs(A.T @ b))
    return A, b, lam


class DiagQuadMORef(ProblemRef):
    """m separable quadratics f_i(x) = 1/2 sum_j D_ij (x_j - C_ij)^2 with the shifted-l1 / box g of
    ``ProblemRef`` - a well-conditioned multi-objective family for fixtures at sizes where the
    reference's own large-n problem (FDS: f_1 ~ n^4 / 6) is dominated by rounding.
    No class of the reference: the NumPy expressions are *defined* here (like ``DiagQuadL1Ref``)
    and handed to the reference's solver as plain callbacks."""

    def __init__(self, D, C, l1_ratios=None, l1_shifts=None, bounds=None):
        D, C = np.asarray(D, float), np.asarray(C, float)
        super().__init__(D.shape[1], D.shape[0], l1_ratios, l1_shifts, bounds)
        self.D, self.C = D, C

    def f(self, x):
        r = x - self.C
        return 0.5 * np.sum(self.D * (r * r), axis=1)

    def jac_f(self, x):
        return self.D * (x - self.C)


def make_quad_mo(n, m=3, seed=5):
    """Seeded inputs of ``DiagQuadMORef``: D ~ U[0.5, 2] (L = 2), C_i ~ N(i - 1, 1)."""
    rng = np.random.default_rng(seed)
    D = rng.uniform(0.5, 2.0, (m, n))
    C = rng.standard_normal((m, n)) + np.arange(m)[:, None]
    return D, C
