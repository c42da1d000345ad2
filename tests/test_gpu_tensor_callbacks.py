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
    # two objectives are not supported on the tensor path
    with pytest.raises(NotImplementedError):
        minimize_proximal_gradient(lambda x: torch.stack([f(x), f(x)]), g, jac_f, prox, x0)
    with pytest.raises(TypeError):
        minimize_proximal_gradient(f, g, jac_f, prox, x0.float())


def test_operator_form_lasso_example_matches_numpy_callbacks():
    """The deblurring workload of the reference's cameraman notebook (operator-form LASSO: blur and
    Haar transform as callbacks) with tensor callbacks against the oracle driven by the same
    operators in NumPy / SciPy (examples/deblur_operator_lasso.py)."""
    import importlib.util
    import os

    import torch
    from scipy.signal import correlate2d

    from conftest import ROOT
    from oracle import cpu_ref
    from zfista_amd import minimize_proximal_gradient

    spec = importlib.util.spec_from_file_location("deblur_example", os.path.join(ROOT, "examples",
                                                                                 "deblur_operator_lasso.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    size = 64
    kernel = ex.gaussian_kernel()
    observed = correlate2d(ex.synthetic_image(size), kernel, mode="same", boundary="symm") \
        + np.random.default_rng(1).standard_normal((size, size)) * 1e-3
    cbs_np, dwt_np, idwt_np = ex.numpy_problem(kernel, observed)
    cbs_t, dwt_t, idwt_t = ex.tensor_problem(kernel, observed)
    x0 = dwt_np(observed)
    assert np.allclose(idwt_np(x0), observed, atol=1e-14)                       # orthonormal Haar level
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
