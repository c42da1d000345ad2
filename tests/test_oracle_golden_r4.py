"""CPU: the oracle on the operator-form LASSO callbacks (oracle/operator_ref.py) reproduces fixture G13 - the
imported reference solver on the same callbacks (tests/golden/make_golden_r4.py) - exactly."""
import warnings

import numpy as np
import pytest

from oracle import cpu_ref, operator_ref as O

CASES = {
    "n64_fista": (64, dict(nesterov=True, nesterov_ratio=(0, 0.25), tol=0.0, max_iter=60)),
    "n64_ista": (64, dict(nesterov=False, tol=0.0, max_iter=40)),
    "n64_ab": (64, dict(nesterov=True, nesterov_ratio=(0.5, 1 / 16), tol=0.0, max_iter=40)),
    "n64_tol": (64, dict(nesterov=True, nesterov_ratio=(0, 0.25), max_iter=400)),
}


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_reproduces_the_reference_on_the_operator_lasso(tag, golden):
    G = golden("g13_operator_lasso.npz")
    size, kw = CASES[tag]
    kernel, observed, x0, L = O.make_deblur(size)
    assert 1 / L == float(G(f"{tag}.lr"))
    prob = O.BlurHaarL1Ref(kernel, observed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = cpu_ref.minimize_proximal_gradient(*prob.callbacks(), x0, lr=1 / L, decay_rate=1, return_all=True, **kw)
    assert res.nit == int(G(f"{tag}.nit")) and res.status == int(G(f"{tag}.status"))
    assert np.array_equal(res.x, G(f"{tag}.x")) and np.array_equal(np.asarray(res.fun), G(f"{tag}.fun"))
    assert np.array_equal(np.concatenate([np.asarray(v).reshape(-1) for v in res.allfuns]), G(f"{tag}.allfuns"))
    assert np.array_equal(np.asarray(res.allerrs), G(f"{tag}.allerrs"))


def test_haar_level_is_orthonormal_and_the_kernel_is_the_gaussian_window():
    from scipy.signal.windows import gaussian

    img = np.random.default_rng(0).standard_normal((12, 20))
    v = O.dwt(img)
    assert np.allclose(O.idwt(v, img.shape), img, atol=1e-15) and np.isclose(np.linalg.norm(v), np.linalg.norm(img))
    assert np.allclose(O.gaussian_kernel(), np.outer(gaussian(9, 4), gaussian(9, 4)), atol=0)
