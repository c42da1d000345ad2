"""CPU: the library's C++ dual solver (zfista_amd/csrc/zf_dual_native.h - plain C++, compiled here
with g++) on synthetic convex piecewise-quadratic duals, against the Python version of the same
algorithm (zfista_amd.multiobjective.solve_dual_native) and against SciPy's SLSQP on the simplex."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest
from scipy.optimize import minimize

from conftest import ROOT
from zfista_amd.multiobjective import solve_dual_native


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    out = tmp_path_factory.mktemp("dualcpp") / "libdual_native_host.so"
    subprocess.run([gxx, "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
                    os.path.join(ROOT, "tests", "cpp", "dual_native_host.cpp"), "-o", str(out)], check=True)
    L = C.CDLL(str(out))
    L.dual_native_solve.restype = C.c_int
    L.dual_native_min_eig.restype = C.c_double
    return L


def _problem(m, seed):
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((m, max(1, m - 1)))      # rank-deficient Q: flat directions like a real dual
    Q = B @ B.T + 1e-3 * np.eye(m)
    q = rng.standard_normal(m)
    kink = rng.uniform(0.0, 5.0, m)
    knot = rng.uniform(0.0, 0.6, m)

    def dual(w):
        w = np.asarray(w, float)
        e = np.maximum(w - knot, 0.0)
        return 0.5 * w @ Q @ w + q @ w + np.sum(kink * e * e), Q @ w + q + 2 * kink * e

    return Q, q, kink, knot, dual


def _solve_cpp(lib, m, Q, q, kink, knot, tol=1e-12):
    w = np.zeros(m)
    fun, nit, evals = C.c_double(), C.c_long(), C.c_int()
    p = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.c_void_p)   # noqa: E731
    Qc, qc, kc, nc = (np.ascontiguousarray(a, dtype=np.float64) for a in (Q, q, kink, knot))
    rc = lib.dual_native_solve(C.c_int(m), p(Qc), p(qc), p(kc), p(nc), None, C.c_double(tol), C.c_long(200),
                               w.ctypes.data_as(C.c_void_p), C.byref(fun), C.byref(nit), C.byref(evals))
    assert rc == 0
    return w, fun.value, nit.value, evals.value


@pytest.mark.parametrize("m", [2, 3, 4, 5, 8])
@pytest.mark.parametrize("seed", range(6))
def test_cpp_solver_against_python_and_slsqp(lib, m, seed):
    Q, q, kink, knot, dual = _problem(m, 100 * m + seed)
    w, fun, nit, evals = _solve_cpp(lib, m, Q, q, kink, knot)
    assert abs(w.sum() - 1) < 1e-12 and np.all(w >= 0)
    assert abs(fun - dual(w)[0]) <= 1e-12 * max(1, abs(fun))
    wp, fp, _ = solve_dual_native(dual, m, np.ones(m) / m, 1e-12, 200)
    np.testing.assert_allclose(w, wp, rtol=0, atol=1e-8)
    assert abs(fun - fp) <= 1e-12 * max(1.0, abs(fp))
    ref = minimize(lambda v: dual(v)[0], np.ones(m) / m, jac=lambda v: dual(v)[1], method="SLSQP",
                   bounds=[(0, None)] * m, constraints=[{"type": "eq", "fun": lambda v: v.sum() - 1}],
                   options=dict(ftol=1e-15, maxiter=500))
    assert fun <= ref.fun + 1e-9 * max(1.0, abs(ref.fun)), "the Newton solver must reach SLSQP's value or better"
    assert evals <= 40 * (m + 2)


def test_min_eigenvalue(lib):
    rng = np.random.default_rng(0)
    for m in (2, 3, 5, 8):
        A = rng.standard_normal((m, m))
        A = 0.5 * (A + A.T)
        got = lib.dual_native_min_eig(C.c_int(m), np.ascontiguousarray(A).ctypes.data_as(C.c_void_p))
        assert abs(got - np.linalg.eigvalsh(A).min()) < 1e-10


@pytest.mark.parametrize("m", [3, 4, 5, 8])
@pytest.mark.parametrize("seed", range(6))
def test_exact_hessian_mode_reaches_the_same_optimum_in_fewer_evaluations(lib, m, seed):
    """zf_dual::machine<M, true> (the mode the device kernel runs, where the Hessian of the dual's quadratic
    piece comes out of the same pass as the gradient): same optimum as the probing mode, without the m
    curvature probes per Newton iteration."""
    Q, q, kink, knot, dual = _problem(m, 100 * m + seed)
    w_probe, f_probe, _, evals_probe = _solve_cpp(lib, m, Q, q, kink, knot)
    w = np.zeros(m)
    fun, nit, evals, batches = C.c_double(), C.c_long(), C.c_int(), C.c_int()
    p = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.c_void_p)   # noqa: E731
    Qc, qc, kc, nc = (np.ascontiguousarray(a, dtype=np.float64) for a in (Q, q, kink, knot))
    lib.dual_native_solve_exact_hessian.restype = C.c_int
    rc = lib.dual_native_solve_exact_hessian(C.c_int(m), p(Qc), p(qc), p(kc), p(nc), None, C.c_double(1e-12),
                                             C.c_long(200), w.ctypes.data_as(C.c_void_p), C.byref(fun), C.byref(nit),
                                             C.byref(evals), C.byref(batches))
    assert rc == 0
    assert abs(w.sum() - 1) < 1e-12 and np.all(w >= 0)
    np.testing.assert_allclose(w, w_probe, rtol=0, atol=1e-8)
    assert abs(fun.value - f_probe) <= 1e-12 * max(1.0, abs(f_probe))
    assert evals.value < evals_probe and batches.value <= evals.value


def test_the_value_guard_of_the_slope_branch_binds(lib):
    """The advisor's finding of round 4: as first written the guard of the approximate-Wolfe branch held for every
    input.  Now: a rise of the value that the convexity of the dual allows (df <= 2 t max(phi'(t), 0): an overshoot
    across a kink, which the slope window alone lets pass) is REJECTED; a rise the gradients cannot explain (the
    reference's composed prox: value up, slope still negative) is waived, as SciPy's gradient-driven search does;
    rounding-size rises leave the decision to the slopes; Armijo steps pass as ever."""
    lib.dual_native_ls_accept.restype = C.c_int
    lib.dual_native_ls_accept.argtypes = [C.c_double, C.c_double, C.c_void_p, C.c_double, C.c_void_p, C.c_double]
    d = np.array([1.0, -0.5, -0.5])

    def accept(fun, slope, f_t, dphi, t=1.0):
        g = dphi * d / (d @ d)          # (mean-free already: (g - mean g) . d = dphi)
        return bool(lib.dual_native_ls_accept(fun, slope, d.ctypes.data_as(C.c_void_p), f_t, np.ascontiguousarray(g).ctypes.data_as(C.c_void_p), t))

    fun, slope = 100.0, -1.0
    assert accept(fun, slope, fun - 0.3, -0.2)                 # Armijo: the value fell by more than 1e-4 t |slope|
    assert accept(fun, slope, fun + 1e-9, -0.5)                # within the noise of the values: the slopes decide (window holds)
    assert not accept(fun, slope, fun + 1e-9, -0.95)           # ... and the window does not hold (slope has hardly shrunk)
    assert not accept(fun, slope, fun + 0.3, 0.6)              # a genuine overshoot: 0.3 <= 2 * 0.6 - rejected although the window holds
    assert not accept(fun, slope, fun + 1.0, 0.9)
    assert accept(fun, slope, fun + 0.3, -0.5)                 # the value rises, the slope is still negative: contradiction - waived
    assert accept(fun, slope, fun + 3.0, 0.6)                  # a rise no convex function with these slopes can show: waived
    assert not accept(fun, slope, fun + 3.0, 1.5)              # (waived, but the slope overshot the window)
