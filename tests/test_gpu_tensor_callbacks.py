"""GPU: callbacks written against device tensors (PyTorch on ROCm).  ``x0`` is a float64 CUDA
tensor, f / g / jac_f / prox_wsum_g take and return CUDA tensors, iterates stay in HBM; the
solver's own vector arithmetic (proximal_gradient.py:148, :150-152, :510, :534) runs in the
library's kernels through the C ABI (zf_dev_*).  Checked against the CPU oracle running the same
problems with NumPy callbacks: same nit, lr decisions and result shape, iterates to 1e-10."""
import warnings

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _torch_lasso(A, b, lam):
    import torch

    At, bt = torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda()

    def f(x):
        r = At @ x - bt
        return 0.5 * torch.sum(r * r)

    def g(x):
        return lam * torch.sum(torch.abs(x))

    def jac_f(x):
        return At.T @ (At @ x - bt)

    def prox(w, v):
        return torch.sign(v) * torch.clamp(torch.abs(v) - lam * w, min=0.0)

    return f, g, jac_f, prox


@pytest.mark.parametrize("kw", [
    dict(lr=1.0, nesterov=True, tol=1e-7, max_iter=60),
    dict(lr=1.0, nesterov=False, tol=0.0, max_iter=25),
    dict(lr=1.0, nesterov=True, nesterov_ratio=(0.5, 0.25), tol=0.0, max_iter=25),
    dict(lr=1e-3, nesterov=True, deprecated=True, tol=0.0, max_iter=20),
])
def test_tensor_callbacks_lasso_vs_oracle(kw):
    import torch

    from oracle import cpu_ref, problems_ref as P
    from zfista_amd import minimize_proximal_gradient

    A, b, lam = P.make_plasso(64, 128, seed=0)
    x0 = torch.zeros(128, dtype=torch.float64, device="cuda")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize_proximal_gradient(*_torch_lasso(A, b, lam), x0, return_all=True, **kw)
        exp = cpu_ref.minimize_proximal_gradient(*P.LeastSquaresL1Ref(A, b, lam).callbacks(), np.zeros(128),
                                                 return_all=True, **kw)
    assert isinstance(res.x, torch.Tensor) and res.x.is_cuda
    assert res.nit == exp.nit and res.status == exp.status and res.message == exp.message
    assert sorted(res.keys()) == sorted(k for k in exp.keys() if k not in ("alllrs", "alltrials"))
    for a, e in zip(res.allvecs, exp.allvecs):
        assert rel_err(a.cpu().numpy(), e) <= 1e-10 or np.linalg.norm(e) == 0
    np.testing.assert_allclose(res.allerrs, exp.allerrs, rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(res.allfuns, exp.allfuns, rtol=1e-10)
    np.testing.assert_allclose(res.fun, exp.fun, rtol=1e-10)


def test_tensor_callbacks_diag_vs_reference_golden(golden):
    """Elementwise torch callbacks round like NumPy: the iterates of the reference are reproduced
    bit for bit, decisions included."""
    import torch

    from oracle import problems_ref as P
    from zfista_amd import minimize_proximal_gradient

    G = golden("g3_diag_n10007.npz")
    d, c, lam = P.make_pdiag(10007, seed=1)
    dt, ct = torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda()
    cbs = (lambda x: 0.5 * torch.sum(dt * (x - ct) ** 2), lambda x: lam * torch.sum(torch.abs(x)),
           lambda x: dt * (x - ct),
           lambda w, v: torch.sign(v) * torch.clamp(torch.abs(v) - lam * w, min=0.0))
    tag = "fista_lr4_backtrack"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize_proximal_gradient(*cbs, torch.zeros(10007, dtype=torch.float64, device="cuda"), lr=4.0,
                                         nesterov=True, tol=0.0, max_iter=60, return_all=True)
    assert res.nit == int(G(f"{tag}.nit"))
    np.testing.assert_allclose(res.allerrs, G(f"{tag}.allerrs"), rtol=1e-10)
    np.testing.assert_allclose(res.allfuns, G(f"{tag}.allfuns"), rtol=1e-10)
    assert np.array_equal(res.x.cpu().numpy(), G(f"{tag}.x"))


def test_tensor_callbacks_error_and_warning_shapes(capsys):
    import torch

    from oracle import problems_ref as P
    from zfista_amd import minimize_proximal_gradient

    A, b, lam = P.make_plasso(16, 32, seed=2)
    f, g, jac_f, prox = _torch_lasso(A, b, lam)
    x0 = torch.zeros(32, dtype=torch.float64, device="cuda")
    with pytest.warns(UserWarning, match="Maximum number of iterations reached"):
        res = minimize_proximal_gradient(f, g, jac_f, prox, x0, lr=1e-3, max_iter=3)
    assert (res.status, res.success, res.nit) == (0, False, 3)
    # a line search that can never succeed: reported, not raised (proximal_gradient.py:493-509)
    res = minimize_proximal_gradient(f, g, lambda x: -1e6 * torch.ones_like(x), prox, x0, lr=1e3, max_backtrack_iter=4)
    assert "Backtracking failed" in capsys.readouterr().out
    assert res.success is False and res.message.startswith("Error: ") and res.nit == 0 and "status" not in res
    assert torch.equal(res.x, x0)
    with pytest.raises(TypeError):
        minimize_proximal_gradient(f, g, jac_f, prox, x0.float())


def test_operator_form_lasso_example_matches_numpy_callbacks():
    """The deblurring workload of the reference's cameraman notebook (operator-form LASSO: blur and
    Haar transform as callbacks) with tensor callbacks against the oracle driven by the same
    operators in NumPy / SciPy (examples/deblur_operator_lasso.py)."""
    import importlib.util
    import os

    import torch

    from conftest import ROOT
    from oracle import cpu_ref, operator_ref as O
    from zfista_amd import minimize_proximal_gradient

    spec = importlib.util.spec_from_file_location("deblur_example", os.path.join(ROOT, "examples",
                                                                                 "deblur_operator_lasso.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    size = 64
    kernel, observed, x0, _ = O.make_deblur(size)
    cbs_np = O.BlurHaarL1Ref(kernel, observed).callbacks()
    cbs_t, dwt_t, idwt_t = ex.tensor_problem(kernel, observed)
    assert np.allclose(O.idwt(x0, observed.shape), observed, atol=1e-14)        # orthonormal Haar level
    assert np.array_equal(dwt_t(torch.from_numpy(observed).cuda()).cpu().numpy(), x0)
    kw = dict(lr=1 / (2 * kernel.sum() ** 2), decay_rate=1, nesterov=True, tol=0.0, max_iter=25, return_all=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize_proximal_gradient(*cbs_t, torch.from_numpy(x0).cuda(), **kw)
        exp = cpu_ref.minimize_proximal_gradient(*cbs_np, x0, **kw)
    assert res.nit == exp.nit == 25
    for a, e in zip(res.allvecs, exp.allvecs):
        assert rel_err(a.cpu().numpy(), e) <= 1e-10
    np.testing.assert_allclose(np.asarray(res.allfuns, float).ravel(), np.asarray(exp.allfuns, float).ravel(), rtol=1e-10)


# ---------------------------------------------------------------------------------------------------
# m >= 2 on device tensors (an extension: the reference takes NumPy arrays only; the contract is its
# m-agnostic one, proximal_gradient.py:143,:467 - f, g -> (m,), jac_f -> (m, n), weight = lr * w)
# ---------------------------------------------------------------------------------------------------
def _jos1_numpy(n, ratios, shifts):
    """JOS1 with shifted-l1 terms as plain NumPy closures (zfista/problems.py:101-138,193-205)."""
    def f(x):
        return np.array([np.linalg.norm(x) ** 2 / n, np.linalg.norm(x - 2) ** 2 / n])

    def g(x):
        return np.array([r * np.linalg.norm(x - s, ord=1) for r, s in zip(ratios, shifts)])

    def jac_f(x):
        return np.vstack([2 * x / n, 2 * (x - 2) / n])

    def st(v, t):
        return np.sign(v) * np.maximum(np.abs(v) - t, 0)

    def prox(weight, x):
        coef = weight * ratios
        x = st(x + coef[1:].sum() - shifts[0] + shifts[0], coef[0])
        for c, s in zip(coef[1:], shifts[1:]):
            x = st(x - c - s, c) + s
        return x

    return f, g, jac_f, prox


def _jos1_torch(n, ratios, shifts):
    import torch

    rt = torch.tensor(ratios, dtype=torch.float64, device="cuda")

    def f(x):
        return torch.stack([torch.linalg.norm(x) ** 2 / n, torch.linalg.norm(x - 2) ** 2 / n])

    def g(x):
        return torch.stack([r * torch.sum(torch.abs(x - s)) for r, s in zip(ratios, shifts)])

    def jac_f(x):
        return torch.stack([2 * x / n, 2 * (x - 2) / n])

    def st(v, t):
        return torch.sign(v) * torch.clamp(torch.abs(v) - t, min=0)

    def prox(weight, x):
        assert weight.is_cuda and weight.shape == (2,)
        coef = weight * rt
        x = st(x + coef[1:].sum() - shifts[0] + shifts[0], coef[0])
        for c, s in zip(coef[1:], shifts[1:]):
            x = st(x - c - s, c) + s
        return x

    return f, g, jac_f, prox


@pytest.mark.parametrize("nesterov", [False, True])
def test_two_objectives_on_device_tensors_match_the_oracle(nesterov):
    """JOS1 + shifted l1 (m = 2) written against device tensors: iterates, errors and function values
    of every iteration against the oracle driven by the same problem in NumPy (SciPy's bounded Brent on
    both sides, as the reference)."""
    import torch

    from oracle import cpu_ref
    from zfista_amd import minimize_proximal_gradient

    n = 2000
    ratios, shifts = np.array([1 / n, 2 / n]), np.array([0.0, 1.0])
    x0 = np.random.default_rng(4).uniform(-2, 4, n)
    kw = dict(lr=1.0, nesterov=nesterov, tol=1e-6, max_iter=12, return_all=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize_proximal_gradient(*_jos1_torch(n, ratios, shifts), torch.from_numpy(x0).cuda(), **kw)
        exp = cpu_ref.minimize_proximal_gradient(*_jos1_numpy(n, ratios, shifts), x0, **kw)
    assert res.nit == exp.nit and res.status == exp.status and res.message == exp.message
    assert torch.is_tensor(res.x) and res.x.is_cuda
    # the reference's own reproducibility for m >= 2 (tests/golden/g10: SciPy's end point moves with
    # the last bits of the dual values) bounds what "equal" can mean: 1e-7 on this problem
    for a, e in zip(res.allvecs, exp.allvecs):
        assert rel_err(a.cpu().numpy(), e) <= 1e-7
    np.testing.assert_allclose(np.asarray(res.allfuns), np.asarray(exp.allfuns), rtol=1e-7)
    np.testing.assert_allclose(np.asarray(res.allerrs), np.asarray(exp.allerrs), rtol=1e-5, atol=1e-9)
    assert np.asarray(res.fun).shape == (2,)


@pytest.mark.parametrize("m", [2, 3])
@pytest.mark.parametrize("dual_solver", ["scipy", "native"])
def test_stacked_toy_lasso_on_device_tensors(m, dual_solver):
    """The reference's multi-objective toy test (tests/test_proximal_gradient.py:104-219: the LASSO toy
    stacked m times; known answers to 3 decimals) with callbacks on device tensors."""
    import torch

    from zfista_amd import minimize_proximal_gradient

    A = torch.tensor([[-1.0], [0.0], [1.0]], dtype=torch.float64, device="cuda")
    b = torch.tensor([-1.0, 0.0, 1.0], dtype=torch.float64, device="cuda")
    x0 = torch.tensor([0.3], dtype=torch.float64, device="cuda")
    for l1_ratio, expected in [(1e-8, 1), (0.1, 0.85), (0.5, 0.25), (1, 0)]:
        def f(x):
            return torch.stack([torch.linalg.norm(A @ x - b) ** 2 / 6] * m)

        def g(x, l1_ratio=l1_ratio):
            return torch.stack([l1_ratio * torch.sum(torch.abs(x))] * m)

        def jac_f(x):
            return torch.stack([A.T @ (A @ x - b) / 3] * m)

        def prox(weight, x, l1_ratio=l1_ratio):
            return torch.sign(x) * torch.clamp(torch.abs(x) - l1_ratio * weight.sum(), min=0)

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = minimize_proximal_gradient(f, g, jac_f, prox, x0, dual_solver=dual_solver)
            res_n = minimize_proximal_gradient(f, g, jac_f, prox, x0, nesterov=True, dual_solver=dual_solver)
        np.testing.assert_array_almost_equal(res.x.cpu().numpy(), [expected], decimal=3)
        np.testing.assert_array_almost_equal(res_n.x.cpu().numpy(), [expected], decimal=3)


def test_multiobjective_tensor_vector_kernels_against_numpy():
    """zf_dev_mo_combine / zf_dev_mo_post_terms (the solver's own expressions of :162-173) against NumPy,
    m = 1 .. 8, sizes around the block and grid boundaries."""
    import torch

    from zfista_amd.proximal_gradient import _DevOps

    ops = _DevOps()
    rng = np.random.default_rng(9)
    for m, n in [(1, 1), (2, 255), (3, 4097), (5, 100003), (8, 600001)]:
        J, y, p, w = rng.standard_normal((m, n)), rng.standard_normal(n), rng.standard_normal(n), rng.random(m)
        lr = 0.37
        Jt, yt, pt = (torch.from_numpy(a).cuda() for a in (J, y, p))
        v, ss = ops.mo_combine(yt, Jt, w, lr)
        wJ = np.zeros(n)
        for i in range(m):
            wJ = wJ + w[i] * J[i]
        assert np.array_equal(v.cpu().numpy(), y - lr * wJ)          # elementwise: bit for bit
        np.testing.assert_allclose(ss.item(), np.sum(wJ * wJ), rtol=1e-12)
        out = ops.mo_post_terms(Jt, yt, pt, v).cpu().numpy()
        np.testing.assert_allclose(out[:m], J @ (p - y), rtol=1e-11, atol=1e-11 * np.abs(J).sum(axis=1).max())
        np.testing.assert_allclose(out[m], np.sum((p - (y - lr * wJ)) ** 2), rtol=1e-12)
