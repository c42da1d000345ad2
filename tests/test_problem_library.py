"""The remaining problem families of zfista/problems.py (SD, ZDT1, TOI4, TRIDIA,
LinearFunctionRank1): known answers the reference's tests hold (SD,
tests/test_problems.py:45-74), product == oracle on random points, finite-difference checks of
the restated Jacobians, and the oracle solver against traces of the imported reference solver
(tests/golden/g7_problem_library.npz).  f / jac_f of these families are host NumPy (n <= 30), so
everything here runs without a GPU; the GPU end-to-end runs are in test_gpu_problem_library.py."""
import warnings

import numpy as np
import pytest

from oracle import cpu_ref, problems_ref as P
from zfista_amd import problems as Z

PAIRS = {
    "sd": (Z.SD, P.SDRef, {}),
    "zdt1": (Z.ZDT1, P.ZDT1Ref, {}),
    "zdt1_n7": (Z.ZDT1, P.ZDT1Ref, dict(n_features=7)),
    "toi4": (Z.TOI4, P.TOI4Ref, {}),
    "tridia": (Z.TRIDIA, P.TRIDIARef, {}),
    "lfr1": (Z.LinearFunctionRank1, P.LinearFunctionRank1Ref, {}),
    "lfr1_6x3": (Z.LinearFunctionRank1, P.LinearFunctionRank1Ref, dict(n_features=6, n_objectives=3)),
}


def test_sd_known_answers_of_the_reference_tests():
    r2 = np.sqrt(2)
    x = np.array([1, r2, r2, 1])
    for sd in (Z.SD(), P.SDRef()):
        np.testing.assert_almost_equal(sd.f(x), [7, 8])
        np.testing.assert_almost_equal(sd.jac_f(x), [[2, r2, r2, 1], [-2, -r2, -r2, -2]])
    np.testing.assert_almost_equal(P.SDRef().g(x), [0, 0])
    np.testing.assert_almost_equal(P.SDRef().prox_wsum_g(np.array([0.5, 0.5]), x), x)


@pytest.mark.parametrize("tag", list(PAIRS))
def test_product_callables_equal_the_oracle(tag):
    zc, pc, kw = PAIRS[tag]
    z, p = zc(**kw), pc(**kw)
    assert (z.n_features, z.n_objectives, z.name) == (p.n_features, p.n_objectives, z._generate_name())
    rng = np.random.default_rng(3)
    for _ in range(5):
        x = rng.uniform(0.2, 2.0, z.n_features)
        assert np.array_equal(z.f(x), p.f(x)) and z.f(x).shape == (z.n_objectives,)
        assert np.array_equal(z.jac_f(x), p.jac_f(x)) and z.jac_f(x).shape == (z.n_objectives, z.n_features)
    with pytest.raises(ValueError, match="len\\(x\\) should be equal to n_features"):
        z.f(np.zeros(z.n_features + 1))


def _fd_jac(p, x, h=1e-6):
    J = np.zeros((p.n_objectives, p.n_features))
    for k in range(p.n_features):
        e = np.zeros(p.n_features)
        e[k] = h
        J[:, k] = (p.f(x + e) - p.f(x - e)) / (2 * h)
    return J


@pytest.mark.parametrize("tag", ["sd", "toi4", "tridia", "lfr1", "lfr1_6x3"])
def test_restated_jacobians_are_the_derivatives(tag):
    _, pc, kw = PAIRS[tag]
    p = pc(**kw)
    x = np.random.default_rng(5).uniform(0.5, 1.5, p.n_features)
    np.testing.assert_allclose(p.jac_f(x), _fd_jac(p, x), rtol=1e-6, atol=1e-6)


def test_zdt1_jacobian_follows_the_code_not_the_docstring():
    """zfista/problems.py:381-383 codes 9 (2 - sqrt(x_1/h)) / (2 (n-1)) for the tail of grad f_2 where
    the class docstring prints 1 - sqrt(...): the code is the derivative."""
    p = P.ZDT1Ref(12)
    x = np.random.default_rng(6).uniform(0.2, 0.9, 12)
    np.testing.assert_allclose(p.jac_f(x), _fd_jac(p, x), rtol=1e-6, atol=1e-7)


def test_names_follow_the_reference_scheme():   # zfista/problems.py:81-91
    assert Z.SD().name == "SD_n_4_bounds_1e-06_inf"
    assert Z.ZDT1().name == "ZDT1_n_30_bounds_1e-06_inf"
    assert Z.TOI4(l1_ratios=[0.1, 0.2], l1_shifts=[0.0, 0.5]).name == "TOI4_n_4_l1_ratios_0.1_0.2_l1_shifts_0.0_0.5"
    assert Z.TRIDIA(bounds=(-1, 2)).name == "TRIDIA_n_3_bounds_-1_2"
    assert Z.LinearFunctionRank1().name == "LinearFunctionRank1_n_10"


G7_CASES = {
    "sd": (P.SDRef, {}, dict(lr=0.5)),
    "zdt1_n30": (P.ZDT1Ref, dict(n_features=30), dict(lr=0.1)),
    "toi4": (P.TOI4Ref, {}, dict(lr=0.5)),
    "toi4_l1": (P.TOI4Ref, dict(l1_ratios=[0.1, 0.2], l1_shifts=[0.0, 0.5]), dict(lr=0.5)),
    "tridia": (P.TRIDIARef, {}, dict(lr=0.05)),
    "tridia_l1_box": (P.TRIDIARef, dict(l1_ratios=[0.1, 0.05, 0.02], l1_shifts=[0.0, 0.1, 0.2], bounds=(-0.5, 0.9)),
                      dict(lr=0.05)),
    "lfr1": (P.LinearFunctionRank1Ref, {}, dict(lr=1e-4)),
}


@pytest.mark.parametrize("tag", list(G7_CASES))
@pytest.mark.parametrize("nesterov", [False, True])
def test_oracle_reproduces_the_reference_solver_traces(tag, nesterov, golden):
    G = golden("g7_problem_library.npz")
    pc, pkw, kw = G7_CASES[tag]
    v = "fista" if nesterov else "ista"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        o = cpu_ref.minimize_proximal_gradient(*pc(**pkw).callbacks(), G(f"{tag}.x0"), nesterov=nesterov, tol=1e-6,
                                               max_iter=15, return_all=True, **kw)
    assert o.nit == int(G(f"{tag}.{v}.nit"))
    assert np.array_equal(np.stack(o.allvecs), G(f"{tag}.{v}.vecs"))
    assert np.array_equal(np.asarray(o.allerrs), G(f"{tag}.{v}.allerrs"))
