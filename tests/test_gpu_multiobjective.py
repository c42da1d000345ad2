"""GPU parity: multi-objective path (m = 2: minimize_scalar, m = 3: trust-constr)
against the golden vectors of the reference solver (G4), the reference's literal
known answers (G5) and its own m = 2 / m = 3 solver tests (closures, generic path).

Tolerances: problem callables 1e-13 (elementwise + one reduction).  Iterates with
dual_solver="scipy" (the reference's calls): max(1e-10, 10 x the reference's OWN spread when
its feature order is permuted - fixture G10): five of the eight G4 cases reproduce to 1e-13
and are held to 1e-10; where SciPy's Brent / barrier end point itself moves with the last
bits of the dual values nothing tighter is defined.  The library's own search ("native",
"device") stops at the stated 2e-5 of the reference's runs and 1e-9 of itself (host loop
against persistent kernel); at grid-wide sizes it is judged in the oracle's arithmetic too
(tests/test_gpu_mo_fullsize.py).  Equal iteration counts throughout."""
import warnings

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _cases():
    from zfista_amd.problems import FDS, JOS1
    from oracle import problems_ref as P

    return {
        "jos1_n50": (lambda: JOS1(50), lambda: P.JOS1Ref(50), dict(lr=1.0)),
        "jos1_n50_l1": (lambda: JOS1(50, l1_ratios=np.arange(1, 3) / 50, l1_shifts=[0, 1]),
                        lambda: P.JOS1Ref(50, l1_ratios=np.arange(1, 3) / 50, l1_shifts=[0, 1]), dict(lr=1.0)),
        "jos1_n1000_l1": (lambda: JOS1(1000, l1_ratios=np.arange(1, 3) / 1000, l1_shifts=[0, 1]),
                          lambda: P.JOS1Ref(1000, l1_ratios=np.arange(1, 3) / 1000, l1_shifts=[0, 1]), dict(lr=1.0)),
        "jos1_n50_box": (lambda: JOS1(50, bounds=(-1.0, 1.5)), lambda: P.JOS1Ref(50, bounds=(-1.0, 1.5)), dict(lr=1.0)),
        "fds_n10": (lambda: FDS(10), lambda: P.FDSRef(10), dict(lr=0.05)),
        "fds_n10_l1": (lambda: FDS(10, l1_ratios=np.arange(1, 4) / 10, l1_shifts=[0, 1, 2]),
                       lambda: P.FDSRef(10, l1_ratios=np.arange(1, 4) / 10, l1_shifts=[0, 1, 2]), dict(lr=0.05)),
        "fds_n100_l1": (lambda: FDS(100, l1_ratios=np.arange(1, 4) / 100, l1_shifts=[0, 1, 2]),
                        lambda: P.FDSRef(100, l1_ratios=np.arange(1, 4) / 100, l1_shifts=[0, 1, 2]), dict(lr=1e-3)),
        "fds_n10_pos": (lambda: FDS(10, bounds=(0, np.inf)), lambda: P.FDSRef(10, bounds=(0, np.inf)), dict(lr=0.05)),
    }


CASES = ["jos1_n50", "jos1_n50_l1", "jos1_n1000_l1", "jos1_n50_box", "fds_n10", "fds_n10_l1", "fds_n100_l1",
         "fds_n10_pos"]


# ---- G5: the reference's literal known answers, evaluated on the GPU ------------------------
def test_known_answers_jos1():   # /root/reference/tests/test_problems.py:8-42
    from zfista_amd.problems import JOS1

    p = JOS1()
    x = np.array([1, 2, 3, 4, 5])
    np.testing.assert_almost_equal(p.f(x), [11, 3])
    np.testing.assert_almost_equal(p.jac_f(x), [[.4, .8, 1.2, 1.6, 2.0], [-.4, 0, .4, .8, 1.2]])
    p = JOS1(l1_ratios=[0.2, 0.1], l1_shifts=[0, 1])
    np.testing.assert_almost_equal(p.g(x), [3, 1])
    np.testing.assert_almost_equal(p.prox_wsum_g(np.array([0.5, 0.5]), np.array([3, 4, 5, 6, 7])),
                                   [2.85, 3.85, 4.85, 5.85, 6.85])


def test_known_answers_fds():   # /root/reference/tests/test_problems.py:77-126
    from zfista_amd.problems import FDS

    p = FDS(n_features=5)
    x = np.array([1, 2, 3, 4, 5])
    np.testing.assert_almost_equal(p.f(x), [0.0, 75.0855369, 0.1183459])
    np.testing.assert_almost_equal(p.jac_f(x), [
        [0, 0, 0, 0, 0],
        [6.01710738, 8.01710738, 10.0171074, 12.0171074, 14.0171074],
        [-0.0613132402, -0.0360894089, -0.0149361205, -4.88417037e-03, -1.12299117e-03]])
    p = FDS(n_features=5, bounds=(0, np.inf))
    np.testing.assert_almost_equal(p.g(np.ones(5)), [0, 0, 0])
    assert np.all(np.isinf(p.g(-np.ones(5))))
    np.testing.assert_almost_equal(p.prox_wsum_g(np.ones(3) / 3, np.array([-3, -1, 0, 1, 3])), [0, 0, 0, 1, 3])
    with pytest.raises(ValueError):
        p.g(np.ones(4))
    with pytest.raises(ValueError):
        p.prox_wsum_g(np.ones(2), np.ones(5))


@pytest.mark.parametrize("tag", CASES)
def test_callables_vs_oracle(tag):
    make, make_ref, _ = _cases()[tag]
    p, r = make(), make_ref()
    rng = np.random.default_rng(3)
    n, m = p.n_features, p.n_objectives
    x = rng.uniform(-2, 2, n) if "pos" not in tag else rng.uniform(0, 2, n)
    np.testing.assert_allclose(p.f(x), r.f(x), rtol=1e-13)
    np.testing.assert_allclose(p.jac_f(x), r.jac_f(x), rtol=1e-13, atol=1e-300)
    np.testing.assert_allclose(p.g(x), r.g(x), rtol=1e-13)
    w = rng.uniform(0.1, 1.0, m)
    got, exp = p.prox_wsum_g(w, 3 * x), r.prox_wsum_g(w, 3 * x)
    np.testing.assert_allclose(got, exp, rtol=0, atol=0)   # same elementwise expression order


def _floor(tag):
    """G10: spread of the imported reference against itself under permuted feature orders."""
    import json
    import os

    from conftest import GOLDEN

    with open(os.path.join(GOLDEN, "g10_reference_noise_floor.json")) as fh:
        return json.load(fh)["cases"][tag]


@pytest.mark.parametrize("tag", CASES)
def test_subproblem_capture(tag, golden):
    """Direct _solve_subproblem capture of the reference at a fixed (lr, x_old, y)."""
    from zfista_amd.multiobjective import MoEngine, X_K, X_NEW, Y, device_dual, solve_dual

    G = golden("g4_multiobjective.npz")
    make, _, _ = _cases()[tag]
    p = make()
    eng = p._engine()
    x0, y, lr = G(f"{tag}.x0"), G(f"{tag}.sub.y"), float(G(f"{tag}.sub.lr"))
    eng.set_x0(x0)
    eng.put(Y, y)
    f0, g0 = eng.eval_F(X_K)
    f_y = eng.prepare()
    m = p.n_objectives
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        w, dual_fun, nit = solve_dual(device_dual(eng, lr, f_y, f0 + g0, False), m, np.ones(m) / m, 1e-12, 100000)
    eng.recover(lr, w)
    x = eng.get(X_NEW)
    # tolerance = max(1e-10, 10 x the reference's OWN spread of this capture when its feature order
    # is permuted, G10): five of the eight cases reproduce to 1e-13 or better, and there the engine
    # has to as well; where SciPy's Brent / barrier end point itself moves with the last bits of the
    # dual values (jos1_n50_l1: 2e-9, fds_n100_l1: 3e-8 in x+, 1e-5 in w) nothing tighter is defined
    fl = _floor(tag)
    assert rel_err(x, G(f"{tag}.sub.x")) <= max(1e-10, 10 * fl["sub_x_rel"])
    np.testing.assert_allclose(w, G(f"{tag}.sub.weight"), rtol=0, atol=max(1e-10, 10 * fl["sub_w_abs"]))
    if np.isfinite(float(G(f"{tag}.sub.fun"))):
        np.testing.assert_allclose(-dual_fun, float(G(f"{tag}.sub.fun")), rtol=max(1e-10, 10 * fl["sub_fun_rel"]),
                                   atol=1e-12)


@pytest.mark.parametrize("tag", CASES)
@pytest.mark.parametrize("nesterov", [False, True])
def test_traces_vs_golden(tag, nesterov, golden):
    from zfista_amd import minimize_proximal_gradient

    G = golden("g4_multiobjective.npz")
    make, _, kw = _cases()[tag]
    p = make()
    v = "fista" if nesterov else "ista"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize_proximal_gradient(*p.callbacks(), G(f"{tag}.x0"), nesterov=nesterov, tol=1e-5, max_iter=12,
                                         return_all=True, **kw)
    assert res.nit == int(G(f"{tag}.{v}.nit"))
    fl = _floor(tag)   # the reference's own spread over the same 12 iterations (G10), see above
    tol = max(1e-10, 10 * fl["trace_x_rel"])
    for a, b in zip(res.allvecs, G(f"{tag}.{v}.vecs")):
        assert rel_err(a, b) <= tol
    np.testing.assert_allclose(np.stack(res.allfuns), G(f"{tag}.{v}.allfuns"), rtol=max(1e-10, 10 * fl["trace_F_rel"]))
    np.testing.assert_allclose(res.allerrs, G(f"{tag}.{v}.allerrs"), rtol=1e-10, atol=max(1e-12, 10 * fl["trace_err_abs"]))


# ---- the reference's own m = 2 / m = 3 solver tests (closures -> generic path) -----------------
def _stacked_toy(l1_ratio, m):
    """tests/test_proximal_gradient.py:122-149 (m = 2) and :175-202 (m = 3)."""
    A = np.array([[-1.0], [0.0], [1.0]])
    b = np.array([-1.0, 0.0, 1.0])

    def f(x):
        val = np.linalg.norm(A @ x - b) ** 2 / 6
        return np.array([val] * m)

    def g(x):
        val = l1_ratio * np.linalg.norm(x, ord=1)
        return np.array([val] * m)

    def jac_f(x):
        grad_fi = A.T @ (A @ x - b) / 3
        return np.vstack([grad_fi] * m)

    def prox_wsum_g(weight, x):
        return np.sign(x) * np.maximum(np.abs(x) - l1_ratio * weight.sum(), 0)

    return f, g, jac_f, prox_wsum_g


@pytest.mark.parametrize("m", [2, 3])
def test_minimize_proximal_gradient_multiobjective_lasso_toy(m):
    from zfista_amd import minimize_proximal_gradient

    x0 = np.random.random(1)
    for l1_ratio, expected in [(1e-8, 1), (0.1, 0.85), (0.5, 0.25), (1, 0)]:
        cb = _stacked_toy(l1_ratio, m)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = minimize_proximal_gradient(*cb, x0)
            res_nesterov = minimize_proximal_gradient(*cb, x0, nesterov=True)
        np.testing.assert_array_almost_equal(res.x, [expected], decimal=3)
        np.testing.assert_array_almost_equal(res_nesterov.x, [expected], decimal=3)


@pytest.mark.parametrize("m", [2, 3])
def test_generic_multiobjective_golden(m, golden):
    """G1: traces of the duplicated-objective toy LASSO (reference outputs)."""
    from zfista_amd import minimize_proximal_gradient

    G = golden("g1_toy_lasso.npz")
    for li, lam in enumerate(G("lams")):
        for nest in (False, True):
            tag = f"l{li}_m{m}_{'fista' if nest else 'ista'}"
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                res = minimize_proximal_gradient(*_stacked_toy(float(lam), m), np.array([0.3]), nesterov=nest,
                                                 return_all=True)
            assert res.nit == int(G(f"{tag}.nit")), tag
            np.testing.assert_allclose(np.concatenate(res.allvecs), G(f"{tag}.vecs").ravel(), rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("tag", ["jos1_n1000_l1", "fds_n10_l1", "fds_n100_l1", "fds_n10_pos"])
def test_native_dual_solver_full_solve(tag, golden):
    """Opt-in native dual solver: same outer iteration count as the
    reference run and iterates within the accuracy of the reference's own dual solves."""
    from zfista_amd import minimize_proximal_gradient

    G = golden("g4_multiobjective.npz")
    make, _, kw = _cases()[tag]
    p = make()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize_proximal_gradient(*p.callbacks(), G(f"{tag}.x0"), nesterov=True, tol=1e-5, max_iter=12,
                                         return_all=True, dual_solver="native", **kw)
    assert res.nit == int(G(f"{tag}.fista.nit"))
    for a, b in zip(res.allvecs, G(f"{tag}.fista.vecs")):
        assert rel_err(a, b) <= 2e-5
    np.testing.assert_allclose(np.stack(res.allfuns), G(f"{tag}.fista.allfuns"), rtol=1e-5)
    assert p._engine().n_dual_evals <= 60 * res.nit


@pytest.mark.parametrize("tag", ["jos1_n50", "jos1_n1000_l1", "jos1_n50_box", "fds_n10_l1", "fds_n100_l1", "fds_n10_pos"])
def test_device_dual_solver_matches_the_host_loop(tag, golden):
    """dual_solver="device" (the whole search of a trial in one persistent kernel: register-resident
    data, last-arriver reductions, the solver's state machine on the device) against
    dual_solver="native" (the same state machine on the host, one launch per evaluation): same
    weights, model value, iteration counts and recovered x+ to rounding - the two differ only in
    the summation order of the 2m+2 sums."""
    from zfista_amd.multiobjective import X_K, X_NEW, Y

    G = golden("g4_multiobjective.npz")
    make, _, _ = _cases()[tag]
    p = make()
    m = p.n_objectives
    eng = p._engine()
    x0, y, lr = G(f"{tag}.x0"), G(f"{tag}.sub.y"), float(G(f"{tag}.sub.lr"))
    eng.set_x0(x0)
    eng.put(Y, y)
    f0, g0 = eng.eval_F(X_K)
    f_y = eng.prepare()
    host = eng.solve_dual(lr, f_y, f0 + g0, False, None, 1e-12, 100000)
    dev = eng.solve_dual_device(lr, f_y, f0 + g0, False, None, 1e-12, 100000)
    if host is None:   # non-finite start (x0 outside the box): neither is attempted
        assert dev is None
        return
    assert dev is not None
    w_h, fun_h, nit_h = host
    w_d, fun_d, nit_d, err_d, f_d, g_d, f_y_d = dev
    assert np.array_equal(f_y_d, f_y)
    x_d = eng.get(X_NEW)
    F_x = eng.eval_F(X_NEW)                      # the kernel's own f(x+), g(x+) against the separate kernels
    np.testing.assert_allclose(g_d, F_x[1], rtol=1e-12)
    np.testing.assert_allclose(f_d, F_x[0], rtol=1e-12)
    err_h = eng.recover(lr, w_h)
    x_h = eng.get(X_NEW)
    np.testing.assert_allclose(w_d, w_h, rtol=0, atol=1e-9)
    np.testing.assert_allclose(fun_d, fun_h, rtol=1e-9, atol=1e-12)
    assert rel_err(x_d, x_h) <= 1e-9
    np.testing.assert_allclose(err_d, err_h, rtol=1e-7, atol=1e-12)
    assert abs(nit_d - nit_h) <= 1


@pytest.mark.parametrize("tag", ["jos1_n1000_l1", "fds_n10_l1", "fds_n100_l1", "fds_n10_pos"])
def test_device_dual_solver_full_solve(tag, golden):
    """Full solves with dual_solver="device": same outer iteration count as the reference run,
    iterates within the accuracy of the reference's own dual solves (as for "native")."""
    from zfista_amd import minimize_proximal_gradient

    G = golden("g4_multiobjective.npz")
    make, _, kw = _cases()[tag]
    p = make()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize_proximal_gradient(*p.callbacks(), G(f"{tag}.x0"), nesterov=True, tol=1e-5, max_iter=12,
                                         return_all=True, dual_solver="device", **kw)
    assert res.nit == int(G(f"{tag}.fista.nit"))
    for a, b in zip(res.allvecs, G(f"{tag}.fista.vecs")):
        assert rel_err(a, b) <= 2e-5
    np.testing.assert_allclose(np.stack(res.allfuns), G(f"{tag}.fista.allfuns"), rtol=1e-5)


def test_device_dual_solver_cfg4_size():
    """BASELINE cfg4 size (FDS n = 1e6, m = 3): every thread of the persistent kernel keeps its 8
    elements of (J, y) in registers; result against the host loop of the same solver."""
    from zfista_amd.multiobjective import X_K, X_NEW
    from zfista_amd.problems import FDS

    n = 10**6
    p = FDS(n, l1_ratios=np.arange(1, 4) / n, l1_shifts=[0, 1, 2])
    x0 = np.random.default_rng(1).uniform(-2, 2, n)
    eng = p._engine()
    eng.set_x0(x0)
    f0, g0 = eng.eval_F(X_K)
    f_y = eng.prepare()
    lr = 1e-7
    w_h, fun_h, nit_h = eng.solve_dual(lr, f_y, f0 + g0, False, None, 1e-12, 100000)
    before = eng.n_dual_evals
    w_d, fun_d, nit_d, err_d, f_d, g_d, _ = eng.solve_dual_device(lr, f_y, f0 + g0, False, None, 1e-12, 100000)
    assert eng.n_dual_evals - before >= 2
    # f(y) formed on the device (prepare_async): the same search, f(y) to the last bits of exp()
    eng.prepare_async()
    w_a, fun_a, nit_a, err_a, _, _, f_y_a = eng.solve_dual_device(lr, None, f0 + g0, False, None, 1e-12, 100000)
    np.testing.assert_allclose(f_y_a, f_y, rtol=1e-15)
    np.testing.assert_allclose(w_a[1:], w_d[1:], rtol=0, atol=1e-9)
    x_d = eng.get(X_NEW)
    F_x = eng.eval_F(X_NEW)
    np.testing.assert_allclose(f_d, F_x[0], rtol=1e-12)
    np.testing.assert_allclose(g_d, F_x[1], rtol=1e-12)
    err_h = eng.recover(lr, w_h)
    x_h = eng.get(X_NEW)
    # the first weight is ~2e-13 against a gradient row of ~1e16 (G10, cfg4 first trial): compare the
    # STEPS, to the accuracy two summation orders of f_1 ~ 1.7e23 allow
    step_d, step_h = x_d - x0, x_h - x0
    assert rel_err(step_d, step_h) <= 0.05
    assert rel_err(x_d, x_h) <= 1e-9
    np.testing.assert_allclose(w_d[1:], w_h[1:], rtol=0, atol=1e-9)
    np.testing.assert_allclose(err_d, err_h, rtol=0.05)


def test_fds_1e6_dual_eval_consistency():
    """BASELINE cfg4 size (n = 10^6, m = 3): one dual evaluation against the oracle's
    NumPy expressions on the same J, y, w (no SciPy in the loop)."""
    from oracle import cpu_ref, problems_ref as P
    from zfista_amd.multiobjective import X_K, Y, device_dual
    from zfista_amd.problems import FDS

    n = 10**6
    ratios, shifts = np.arange(1, 4) / n, [0, 1, 2]
    p, r = FDS(n, l1_ratios=ratios, l1_shifts=shifts), P.FDSRef(n, l1_ratios=ratios, l1_shifts=shifts)
    rng = np.random.default_rng(1)
    x0 = rng.uniform(-2, 2, n)
    y = x0 + 1e-3 * rng.standard_normal(n)
    eng = p._engine()
    eng.set_x0(x0)
    eng.put(Y, y)
    f0, g0 = eng.eval_F(X_K)
    np.testing.assert_allclose(f0, r.f(x0), rtol=1e-12)
    np.testing.assert_allclose(g0, r.g(x0), rtol=1e-12)
    f_y = eng.prepare()
    np.testing.assert_allclose(f_y, r.f(y), rtol=1e-12)
    J = r.jac_f(y)
    np.testing.assert_allclose(eng.get_jac(), J, rtol=1e-12, atol=1e-300)
    lr = 1e-7
    for w in (np.ones(3) / 3, np.array([0.7, 0.2, 0.1]), np.array([0.0, 0.5, 0.5])):
        fun, jac = device_dual(eng, lr, f_y, f0 + g0, False)(w)
        efun, ejac = cpu_ref.dual_value_and_grad(w, r.g, r.prox_wsum_g, lr, y, J, r.f(y), r.f(x0) + r.g(x0))
        # fun and jac contain F(x_old) - f(y): at n = 1e6 FDS's f_1 is ~1e23, so that difference
        # carries ~1e-16 * 1e23 of cancellation noise whatever the summation order
        noise = 4e-16 * np.max(np.abs(f_y)) * 8
        np.testing.assert_allclose(fun, efun, rtol=1e-9, atol=noise)
        np.testing.assert_allclose(jac, ejac, rtol=1e-9, atol=noise)


@pytest.mark.parametrize("path", ["native", "generic"])
@pytest.mark.parametrize("cls", ["jos1", "fds"])
def test_warm_start_of_the_dual_search(cls, path):
    """warm_start=True feeds the weights of the previous trial to the next dual search
    (zfista/proximal_gradient.py:192-205,286-288): native (device engine) and generic (opaque
    callbacks) paths against the oracle, which restates the same hand-over."""
    from oracle import cpu_ref, problems_ref as P
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.problems import FDS, JOS1

    if cls == "jos1":
        n, kw = 60, dict(l1_ratios=np.arange(1, 3) / 60, l1_shifts=[0, 1])
        prob, ref, o = JOS1(n, **kw), P.JOS1Ref(n, **kw), dict(lr=1.0)
    else:
        n, kw = 10, dict(l1_ratios=np.arange(1, 4) / 10, l1_shifts=[0, 1, 2])
        prob, ref, o = FDS(n, **kw), P.FDSRef(n, **kw), dict(lr=0.05)
    x0 = np.random.default_rng(12).uniform(-2, 2, n)
    o.update(nesterov=True, tol=1e-6, max_iter=10, warm_start=True, return_all=True)
    cbs = prob.callbacks()
    if path == "generic":   # plain closures: not recognised as a device-native problem
        cbs = tuple((lambda fn: (lambda *a: fn(*a)))(fn) for fn in ref.callbacks())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize_proximal_gradient(*cbs, x0, **o)
        exp = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), x0, **o)
        cold = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), x0, **dict(o, warm_start=False))
    assert res.nit == exp.nit
    tol = 1e-7 if cls == "jos1" else 1e-6
    for a, b in zip(res.allvecs, exp.allvecs):
        assert np.linalg.norm(a - b) <= tol * max(1.0, np.linalg.norm(b))
    np.testing.assert_allclose(np.stack(res.allfuns), np.stack(exp.allfuns), rtol=1e-6)
    assert len(cold.allvecs) >= 2   # (the cold-start run exists and converges to the same point class)


@pytest.mark.parametrize("tag", ["jos1_n50_l1", "jos1_n1000_l1", "fds_n10", "fds_n10_l1", "fds_n100_l1", "fds_n10_pos"])
def test_library_dual_solver_matches_the_python_one(tag, golden):
    """zf_mo_solve_dual (C++, no Python between evaluations) against
    multiobjective.solve_dual_native (the same algorithm in Python) on the same engine state."""
    from zfista_amd.multiobjective import Y, device_dual, solve_dual_native

    G = golden("g4_multiobjective.npz")
    p = _cases()[tag][0]()
    eng = p._engine()
    eng.set_x0(G(f"{tag}.x0"))
    eng.put(Y, G(f"{tag}.sub.y"))
    f0, g0 = eng.eval_F(0)
    F_old = f0 + g0
    f_y = eng.prepare()
    lr = float(G(f"{tag}.sub.lr"))
    m = p.n_objectives
    for deprecated in (False, True):
        got = eng.solve_dual(lr, f_y, F_old, deprecated, None, 1e-12, 100000)
        exp = solve_dual_native(device_dual(eng, lr, f_y, F_old, deprecated), m, np.ones(m) / m, 1e-12, 100000)
        assert got is not None and exp is not None
        np.testing.assert_allclose(got[0], exp[0], rtol=0, atol=1e-9)
        np.testing.assert_allclose(got[1], exp[1], rtol=1e-12, atol=1e-12 * max(1.0, abs(exp[1])))
        assert abs(got[0].sum() - 1.0) < 1e-12 and np.all(got[0] >= 0)
    # not attempted when F(x_k) is not finite
    assert eng.solve_dual(lr, f_y, np.full(m, np.inf), False, None, 1e-12, 100) is None


@pytest.mark.parametrize("kind", ["jos1", "fds"])
def test_fused_outer_iteration_is_invisible(kind):
    """zf_mo_set_fused: commit() and prepare_async() deferred into the next solve_dual_device(), whose
    kernel forms y, f(y) and J itself.  (a) The deferred work done the unfused way on demand gives the
    same buffers bit for bit: y after commit(), J and f(y) after prepare_async(), read through the
    ordinary accessors.  (b) A sequence of trials with and without fusion ends in the same iterates to
    1e-12 (f(y) is summed in another order inside the kernel: last-bit differences in the dual)."""
    from zfista_amd.multiobjective import X_K, X_NEW, Y
    from zfista_amd.problems import FDS, JOS1

    n = 40000 if kind == "fds" else 100003
    rng = np.random.default_rng(11)
    if kind == "jos1":
        make = lambda: JOS1(n, l1_ratios=np.array([1.0, 2.0]) / n, l1_shifts=[0.0, 1.0])   # noqa: E731
        x0, lr = rng.uniform(-2, 4, n), 1.0
    else:
        make = lambda: FDS(n, l1_ratios=np.arange(1, 4) / n, l1_shifts=[0.0, 1.0, 2.0])   # noqa: E731
        x0, lr = rng.uniform(-2, 2, n), 1e-4
    m = 2 if kind == "jos1" else 3

    def run(fused, probe):
        eng = make()._engine()
        eng.set_x0(x0)
        eng.set_fused(fused)
        f0, g0 = eng.eval_F(X_K)
        F_old = f0 + g0
        seen = []
        for k, beta in enumerate([0.0, 0.28, 0.43, 0.53]):
            eng.prepare_async()
            if probe:   # every accessor must see up-to-date buffers while work is still deferred
                seen.append((eng.get(Y), eng.get_jac(), eng.get_f_y()))   # (this does the deferred work the unfused way)
            out = eng.solve_dual_device(lr, None, F_old, False, None, 1e-12, 100000)
            assert out is not None
            f_x, g_x = out[4], out[5]
            F_old = f_x + g_x
            eng.commit(beta, True)
        xk, y = eng.get(X_K), eng.get(Y)
        eng.close()
        return xk, y, F_old, seen

    xa, ya, Fa, _ = run(False, False)
    xb, yb, Fb, _ = run(True, False)
    xc, yc, Fc, seen_c = run(True, True)
    xd, yd, Fd, seen_d = run(False, True)
    assert rel_err(xb, xa) <= 1e-12 and rel_err(yb, ya) <= 1e-12
    np.testing.assert_allclose(Fb, Fa, rtol=1e-12)
    # with the deferred work flushed before every trial the fused mode IS the unfused sequence
    assert np.array_equal(xc, xd) and np.array_equal(yc, yd) and np.array_equal(Fc, Fd)
    for (y1, J1, f1), (y2, J2, f2) in zip(seen_c, seen_d):
        assert np.array_equal(y1, y2) and np.array_equal(J1, J2) and np.array_equal(f1, f2)
    assert len(seen_c) == 4 and seen_c[0][1].shape == (m, n)


@pytest.mark.parametrize("case", ["jos1_backtrack", "jos1_converge", "fds_backtrack", "fds_box_outside", "jos1_short",
                                  "fds_deprecated", "jos1_decay1", "jos1_warm", "fds_warm"])
def test_trials_launched_ahead_match_the_sequential_loop(case, monkeypatch):
    """dual_solver="device" launches every trial ahead of its predecessor's result (gated on the decision the
    kernel takes itself, zf_mo_trial_launch / _wait).  Against the same solve with the host reading every
    result before the next launch (ZF_MO_LAUNCH_AHEAD=0): line searches that backtrack (closed gates, undone
    commits), convergence by tol, max_iter 1 / 2 / 3, a start outside the box (the search is not attempted on
    the device: the host loop takes over), the deprecated test and decay_rate = 1."""
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.problems import FDS, JOS1

    rng = np.random.default_rng(12)
    kw = dict(nesterov=True, tol=0.0, dual_solver="device")
    if case.startswith("jos1"):
        n = 20011
        prob = lambda: JOS1(n, l1_ratios=np.array([1.0, 2.0]) / n, l1_shifts=[0.0, 1.0])   # noqa: E731
        x0 = rng.uniform(-2, 4, n)
        if case in ("jos1_backtrack", "jos1_warm"):
            # several halvings in the first line searches; warm: every search starts from the weights of the trial
            # before it (:286-288) - of the accepted trial on the device (gated = 2), of a rejected one from its record
            runs = [dict(kw, lr=64.0 * n, max_iter=15, warm_start=(case == "jos1_warm"))]
        elif case == "jos1_converge":
            runs = [dict(kw, lr=0.5 * n, tol=1e-3, max_iter=500)]
        elif case == "jos1_short":
            runs = [dict(kw, lr=0.5 * n, max_iter=k) for k in (1, 2, 3)]
        else:
            runs = [dict(kw, lr=0.25 * n, decay_rate=1, max_iter=9)]
    else:
        n = 3001
        if case == "fds_box_outside":
            prob = lambda: FDS(n, l1_ratios=np.arange(1, 4) / n, l1_shifts=[0.0, 1.0, 2.0], bounds=(-1.0, 1.0))   # noqa: E731
            x0 = rng.uniform(-2, 2, n)                             # F(x0) = inf: not attempted on the device
        else:
            prob = lambda: FDS(n, l1_ratios=np.arange(1, 4) / n, l1_shifts=[0.0, 1.0, 2.0])   # noqa: E731
            x0 = rng.uniform(-2, 2, n)
        runs = [dict(kw, lr=1.0, max_iter=10, deprecated=(case == "fds_deprecated"), warm_start=(case == "fds_warm"))]
    for opts in runs:
        out = []
        for ahead in ("1", "0"):
            monkeypatch.setenv("ZF_MO_LAUNCH_AHEAD", ahead)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                out.append(minimize_proximal_gradient(*prob().callbacks(), x0, **opts))
        a, b = out
        assert (a.nit, a.get("status"), a.success, a.message) == (b.nit, b.get("status"), b.success, b.message)
        if case == "fds_box_outside":   # F(x0) = inf: SciPy refuses the dual on both paths - reported, not raised (:493-509)
            assert a.success is False and a.message.startswith("Error: ") and np.array_equal(a.x, b.x)
            continue
        assert rel_err(a.x, b.x) <= 1e-10 and np.allclose(a.fun, b.fun, rtol=1e-10, atol=0)
        if case == "jos1_converge":
            assert a.success and a.nit < 500
        if case == "jos1_short":
            assert a.nit == opts["max_iter"]


@pytest.mark.parametrize("variant", ["l1", "l1_box", "box", "plain"])
def test_dual_hessian_against_differences_of_the_gradient(variant):
    """zf_mo_dual_hessian (the derivative of prox_wsum_g through its composed soft-thresholds and the clip,
    accumulated as m x m sums: what the device-side search builds its Newton model from) against central
    differences of the gradient of the dual (zf_mo_dual_eval), FDS m = 3 with / without l1 terms and box,
    at random weights.  The dual is piecewise quadratic: inside a piece the two agree to the accuracy of the
    differences."""
    from zfista_amd.multiobjective import X_K
    from zfista_amd.problems import FDS

    n = 400
    rng = np.random.default_rng(21)
    kw = {}
    if "l1" in variant:
        kw.update(l1_ratios=np.array([0.7, 1.3, 0.4]) / n, l1_shifts=[0.0, 0.8, -0.5])
    if "box" in variant:
        kw.update(bounds=(-1.5, 1.5))
    p = FDS(n, **kw)
    eng = p._engine()
    eng.set_x0(rng.uniform(-1.4, 1.4, n))
    eng.eval_F(X_K)
    eng.prepare()
    m, lr = 3, 0.05

    def grad(w):   # the gradient of the dual as :173-177 compose it (without the constant F_old - f_y)
        g_p, _, _, dots = eng.dual_eval(lr, w)
        return -g_p - dots

    worst = 0.0
    for _ in range(12):
        w = rng.dirichlet(np.ones(m))
        H = eng.dual_hessian(lr, w)
        h = 1e-6
        fd = np.zeros((m, m))
        for k in range(m):
            e = np.zeros(m)
            e[k] = h
            fd[:, k] = (grad(w + e) - grad(w - e)) / (2 * h)
        scale = np.abs(fd).max()
        worst = max(worst, np.abs(H - fd).max() / scale)
        assert np.abs(H - H.T).max() <= 1e-9 * scale      # symmetric, as a Hessian is
    # (a kink of some element may fall inside a difference interval: allow the odd piece change)
    assert worst <= 2e-3
    eng.close()


@pytest.mark.parametrize("at", [1, 4])
@pytest.mark.parametrize("ahead", ["1", "0"])
@pytest.mark.parametrize("kind", ["jos1", "fds"])
def test_device_trial_that_gives_up_falls_back_to_the_host_search(kind, ahead, at, monkeypatch):
    """k_dual_solve keeps its whole grid spinning on grid-wide hand-overs; when its workgroups are not all
    resident (another kernel / process / a CU mask holds CUs) a wait gives up, the record says ok = -1 and the
    solve must go on with the host-driven search - a UserWarning, not a failed solve.  The give-up is forced
    here (zf_mo_debug_force_timeout) at the first device launch (a line search that then backtracks on the
    host) and at the fourth (in the middle of a solve whose trials are all accepted, so that with trials
    launched ahead the forced launch is one that runs: a gated launch behind a rejected trial exits at its gate
    before any wait), with trials launched ahead and without.  The result equals the dual_solver="native"
    solve (the same search, host loop) with the same iteration count."""
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.multiobjective import MoEngine
    from zfista_amd.problems import FDS, JOS1

    monkeypatch.setenv("ZF_MO_LAUNCH_AHEAD", ahead)
    rng = np.random.default_rng(12)
    if kind == "jos1":
        n = 20011
        make = lambda: JOS1(n, l1_ratios=np.array([1.0, 2.0]) / n, l1_shifts=[0.0, 1.0])   # noqa: E731
        x0, kw = rng.uniform(-2, 4, n), dict(lr=(64.0 if at == 1 else 0.25) * n, max_iter=12)
    else:
        n = 3001
        make = lambda: FDS(n, l1_ratios=np.arange(1, 4) / n, l1_shifts=[0.0, 1.0, 2.0])   # noqa: E731
        x0, kw = rng.uniform(-2, 2, n), dict(lr=1.0 if at == 1 else 1e-9, max_iter=8)
    kw.update(nesterov=True, tol=0.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = minimize_proximal_gradient(*make().callbacks(), x0, dual_solver="native", **kw)
        clean = minimize_proximal_gradient(*make().callbacks(), x0, dual_solver="device", **kw)
    launches = {"n": 0}
    for name in ("trial_launch", "solve_dual_device"):
        orig = getattr(MoEngine, name)

        def counted(self, *a, _orig=orig, **k):
            launches["n"] += 1
            if launches["n"] == at:
                self.debug_force_timeout(1)
            return _orig(self, *a, **k)

        monkeypatch.setattr(MoEngine, name, counted)
    with pytest.warns(UserWarning, match="gave up waiting for its grid"):
        got = minimize_proximal_gradient(*make().callbacks(), x0, dual_solver="device", **kw)
    assert launches["n"] >= at
    assert (got.nit, got.status, got.success, got.message) == (want.nit, want.status, want.success, want.message)
    assert rel_err(got.x, want.x) <= 1e-9 and rel_err(got.x, clean.x) <= 1e-9
    np.testing.assert_allclose(got.fun, want.fun, rtol=1e-9)
