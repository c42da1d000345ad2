"""GPU parity: the operator-form LASSO of the reference's examples/cameraman.ipynb as a recognised,
device-resident problem (zfista_amd.problems.BlurHaarL1, csrc/zf_kernels_op.h) against fixture G13 - the imported
reference solver on the notebook's callbacks - and against the oracle run live.  Tolerance 1e-10 relative
(north_star); the correlation sums its 81 products in another order than SciPy, so no bit-exactness is claimed."""
import warnings

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10

CASES = {
    "n64_fista": (64, dict(nesterov=True, nesterov_ratio=(0, 0.25), tol=0.0, max_iter=60)),
    "n64_ista": (64, dict(nesterov=False, tol=0.0, max_iter=40)),
    "n64_ab": (64, dict(nesterov=True, nesterov_ratio=(0.5, 1 / 16), tol=0.0, max_iter=40)),
    "n64_tol": (64, dict(nesterov=True, nesterov_ratio=(0, 0.25), max_iter=400)),
    "n256_fista": (256, dict(nesterov=True, nesterov_ratio=(0, 0.25), tol=0.0, max_iter=30)),
}


def _problem(size):
    from oracle import operator_ref as O
    from zfista_amd.problems import BlurHaarL1

    kernel, observed, x0, L = O.make_deblur(size)
    return BlurHaarL1(kernel, observed, O.L1_RATIO), O.BlurHaarL1Ref(kernel, observed), x0, L


def test_callbacks_match_the_notebook_expressions():
    """f, g, jac_f, prox_wsum_g as plain callables (types included: arrays of one value, a (1, n) Jacobian)."""
    prob, ref, x0, L = _problem(64)
    x = x0 + np.random.default_rng(3).standard_normal(x0.size) * 0.1
    f, fr = prob.f(x), ref.f(x)
    assert f.shape == fr.shape == (1,) and abs(f[0] - fr[0]) <= 1e-12 * abs(fr[0])
    g, gr = prob.g(x), ref.g(x)
    assert g.shape == gr.shape == (1,) and abs(g[0] - gr[0]) <= 1e-13 * abs(gr[0])
    J, Jr = prob.jac_f(x), ref.jac_f(x)
    assert J.shape == Jr.shape == (1, x.size) and rel_err(J, Jr) <= 1e-13
    p, pr = prob.prox_wsum_g(1 / L, x), ref.prox_wsum_g(1 / L, x)
    assert np.array_equal(p, pr)


@pytest.mark.parametrize("tag", list(CASES))
def test_device_resident_solve_matches_the_reference_fixture(tag, golden):
    from zfista_amd import minimize_proximal_gradient

    G = golden("g13_operator_lasso.npz")
    size, kw = CASES[tag]
    prob, _, x0, L = _problem(size)
    assert 1 / L == float(G(f"{tag}.lr"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize_proximal_gradient(*prob.callbacks(), x0, lr=1 / L, decay_rate=1, return_all=True, **kw)
    assert res.nit == int(G(f"{tag}.nit")) and res.status == int(G(f"{tag}.status"))
    stride = 1 if size == 64 else 17
    assert abs(np.linalg.norm(res.x) - float(G(f"{tag}.xnorm"))) <= TOL * float(G(f"{tag}.xnorm"))
    assert np.linalg.norm(res.x[::stride] - G(f"{tag}.x")) <= TOL * float(G(f"{tag}.xnorm"))
    assert np.asarray(res.fun).shape == (1,), "fun keeps the callbacks' type: an array of one value"
    np.testing.assert_allclose(np.asarray(res.fun), G(f"{tag}.fun"), rtol=TOL)
    np.testing.assert_allclose(np.concatenate([np.asarray(v).reshape(-1) for v in res.allfuns]), G(f"{tag}.allfuns"), rtol=TOL)
    np.testing.assert_allclose(np.asarray(res.allerrs), G(f"{tag}.allerrs"), rtol=1e-9, atol=1e-16)
    assert len(res.allvecs) == res.nit + 1 and res.allvecs[0] is x0


def test_with_a_line_search_a_box_and_other_kernel_sizes():
    """Beyond the notebook's options: backtracking from lr = 40 / L (decay_rate 0.5), a box, a 5 x 5 and a 15 x 15
    kernel on a non-square image - against the oracle on the same callbacks."""
    from oracle import cpu_ref, operator_ref as O
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.problems import BlurHaarL1

    rng = np.random.default_rng(7)
    for ksize, shape in ((5, (48, 80)), (15, (64, 32)), (9, (64, 64))):
        kernel = O.gaussian_kernel(ksize, 2.0)
        kernel = kernel / kernel.sum()
        observed = rng.standard_normal(shape)
        ref = O.BlurHaarL1Ref(kernel, observed, l1_ratio=0.02)
        prob = BlurHaarL1(kernel, observed, 0.02)
        x0 = O.dwt(observed)
        kw = dict(lr=40.0, nesterov=True, tol=1e-9, max_iter=80)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = minimize_proximal_gradient(*prob.callbacks(), x0, **kw)
            exp = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), x0, **kw)
        assert res.nit == exp.nit and res.status == exp.status, (ksize, res.nit, exp.nit)
        assert rel_err(res.x, exp.x) <= TOL
        np.testing.assert_allclose(np.asarray(res.fun), np.asarray(exp.fun), rtol=TOL)


@pytest.mark.parametrize("variant", ["separable", "general_same_kernel", "general_rank_two", "k1", "k3_tall_image"])
def test_beyond_the_notebook_size_and_every_kernel_path(variant, monkeypatch):
    """Round 5: the correlation kernels were rewritten (64 x 32 tiles, register sliding windows, a separable path for
    rank-1 kernels, 64 x 8 tiles for images of few tiles).  Every path against the oracle on the same callbacks at
    sizes where the 64 x 32 tiles run (1024 x 1024: the verdict's "parity test at 1024^2"), with partial tiles at the
    right and bottom edges, a kernel that is NOT separable, the separable kernel forced through the general path
    (ZF_OP_SEPARABLE=0), and the degenerate 1 x 1 kernel (zero-padded to 3 x 3)."""
    from oracle import cpu_ref, operator_ref as O
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.problems import BlurHaarL1

    rng = np.random.default_rng(11)
    iters = 4
    if variant in ("separable", "general_same_kernel"):
        kernel, observed, x0, L = O.make_deblur(1024)
        lam = O.L1_RATIO
        if variant == "general_same_kernel":
            monkeypatch.setenv("ZF_OP_SEPARABLE", "0")
    elif variant == "general_rank_two":
        shape = (1056, 1000)          # partial tiles in both directions, 64 x 32 tiles (33 x 16 = 528 ... x 2 rows)
        k1 = O.gaussian_kernel(7, 1.5)
        kernel = k1 / k1.sum() + 0.05 * np.outer(np.arange(7) - 3, np.ones(7)) / 49     # rank 2: no separable path
        observed = rng.standard_normal(shape)
        x0, lam = O.dwt(observed), 0.02
        L = O.lipschitz(kernel)
    elif variant == "k1":
        kernel = np.array([[0.7]])
        observed = rng.standard_normal((64, 192))
        x0, lam, L = O.dwt(observed), 0.05, 2 * 0.7 ** 2
    else:
        kernel = O.gaussian_kernel(3, 1.0)
        kernel = kernel / kernel.sum()
        observed = rng.standard_normal((2048, 96))     # tall and narrow: two tile columns, the second partial
        x0, lam = O.dwt(observed), 0.02
        L = O.lipschitz(kernel)
    ref = O.BlurHaarL1Ref(kernel, observed, l1_ratio=lam)
    prob = BlurHaarL1(kernel, observed, lam)
    kw = dict(lr=1 / L, decay_rate=1, nesterov=True, tol=0.0, max_iter=iters, return_all=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize_proximal_gradient(*prob.callbacks(), x0, **kw)
        exp = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), x0, **kw)
    assert res.nit == exp.nit == iters
    for k in range(1, iters + 1):
        assert rel_err(np.asarray(res.allvecs[k]), exp.allvecs[k]) <= TOL, k
    np.testing.assert_allclose(np.concatenate([np.asarray(v).reshape(-1) for v in res.allfuns]),
                               np.concatenate([np.asarray(v).reshape(-1) for v in exp.allfuns]), rtol=TOL)
    # the callbacks outside the loop take the same kernels (zf_op_eval)
    x = x0 + 0.1 * rng.standard_normal(x0.size)
    assert rel_err(prob.jac_f(x), ref.jac_f(x)) <= 1e-12 and abs(prob.f(x)[0] - ref.f(x)[0]) <= 1e-12 * abs(ref.f(x)[0])


@pytest.mark.parametrize("case", ["fista_line_search", "ista", "box", "non_square_k5", "tall_tiles_1024"])
def test_prox_step_in_the_adjoint_kernel_equals_a_launch_of_its_own(case, monkeypatch):
    """Round 5: the prox step of a trial rides in the epilogue of the adjoint kernel (two launches per trial; images of up
    to 5 Mi pixels, no history ring).  ZF_OP_FUSE_PROX=0 is the three-launch trial: same iterates (the step is the same
    expression on the same values - only the order in which its four sums are added differs), same decisions, and both equal
    the oracle."""
    from oracle import cpu_ref, operator_ref as O
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.problems import BlurHaarL1

    rng = np.random.default_rng(11)
    ksize, shape, bounds = 9, (64, 64), None
    kw = dict(lr=40.0, nesterov=True, tol=1e-9, max_iter=60)   # rejections first: a retry runs the fused kernel again
    if case == "ista":
        kw = dict(lr=0.5, nesterov=False, tol=0.0, max_iter=40)
    elif case == "box":
        bounds = (-0.05, 0.4)
    elif case == "non_square_k5":
        ksize, shape = 5, (48, 160)
    elif case == "tall_tiles_1024":
        shape = (1024, 1024)
        kw = dict(lr=0.9, nesterov=True, tol=0.0, max_iter=12)
    kernel = O.gaussian_kernel(ksize, 2.0)
    kernel = kernel / kernel.sum()
    observed = rng.standard_normal(shape)
    prob = BlurHaarL1(kernel, observed, 0.02, bounds=bounds)
    x0 = O.dwt(observed)
    if bounds is not None:
        x0 = np.clip(x0, *bounds)
    out = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("ZF_OP_FUSE_PROX", fuse)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out[fuse] = minimize_proximal_gradient(*prob.callbacks(), x0, **kw)
    a, b = out["1"], out["0"]
    assert a.nit == b.nit and a.status == b.status
    assert rel_err(a.x, b.x) <= 1e-13
    np.testing.assert_allclose(np.asarray(a.fun), np.asarray(b.fun), rtol=1e-12)
    if case != "tall_tiles_1024" and bounds is None:   # (the oracle's SciPy correlation at 1024 x 1024 takes its time; box: no oracle callbacks)
        ref = O.BlurHaarL1Ref(kernel, observed, l1_ratio=0.02)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            exp = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), x0, **kw)
        assert a.nit == exp.nit and a.status == exp.status
        assert rel_err(a.x, exp.x) <= TOL


@pytest.mark.parametrize("ksize,general", [(9, False), (7, True)])
def test_workgroups_that_walk_their_tiles_with_partial_tiles_at_the_edges(ksize, general, monkeypatch):
    """More tiles than the device holds workgroups (1410 of 64 x 32 on a 1502 x 1898 image, sides no multiples of the tile):
    a workgroup walks several tiles, the apply kernel with the next tile's coefficients in flight.  ZF_OP_PERSIST=0 (a
    workgroup per tile) adds the same shares in the same order: bit-identical; both equal the oracle."""
    from oracle import cpu_ref, operator_ref as O
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.problems import BlurHaarL1

    rng = np.random.default_rng(23)
    kernel = O.gaussian_kernel(ksize, 2.0)
    if general:
        kernel = kernel + 0.3 * np.outer(np.arange(ksize), np.ones(ksize)) / ksize
    kernel = kernel / kernel.sum()
    observed = rng.standard_normal((1502, 1898))
    prob = BlurHaarL1(kernel, observed, 0.02)
    x0 = O.dwt(observed)
    kw = dict(lr=0.5, nesterov=True, tol=0.0, max_iter=5)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ZF_OP_PERSIST", mode)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out[mode] = minimize_proximal_gradient(*prob.callbacks(), x0, **kw)
    assert np.array_equal(out["1"].x, out["0"].x) and np.array_equal(np.asarray(out["1"].fun), np.asarray(out["0"].fun))
    ref = O.BlurHaarL1Ref(kernel, observed, l1_ratio=0.02)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), x0, **kw)
    assert out["1"].nit == exp.nit and rel_err(out["1"].x, exp.x) <= TOL
    np.testing.assert_allclose(np.asarray(out["1"].fun), np.asarray(exp.fun), rtol=TOL)


def test_independent_solves_on_streams_equal_the_solves_alone():
    """zfista_amd.replicas.solve_on_streams: the notebook's sweep pattern (cameraman.ipynb cell 11: joblib over momentum
    settings) as host threads with a HIP stream each on ONE GPU - every result must be the one the same call gives alone."""
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.replicas import solve_on_streams

    prob, _, x0, L = _problem(128)
    ratios = [(0, 0.25), (0.5, 1 / 16), (0.75, 0.25), (0.25, 1 / 64), (1 / 6, 1 / 144)]
    kws = [dict(lr=1 / L, decay_rate=1, nesterov=True, nesterov_ratio=r, return_all=(k % 2 == 0)) for k, r in enumerate(ratios)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        alone = [minimize_proximal_gradient(*prob.callbacks(), x0, **kw) for kw in kws]
    together = solve_on_streams([(prob, x0, kw) for kw in kws], streams=5)
    again = solve_on_streams([(prob.callbacks(), x0, kw) for kw in kws], streams=2)
    for a, b, c in zip(alone, together, again):
        assert a.nit == b.nit == c.nit and a.status == b.status == c.status
        assert np.array_equal(a.x, b.x) and np.array_equal(a.x, c.x) and np.array_equal(np.asarray(a.fun), np.asarray(b.fun))
        if a.allfuns is not None:
            assert np.array_equal(np.concatenate(a.allfuns), np.concatenate(b.allfuns))
