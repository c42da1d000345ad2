"""zfista_amd.metrics against the known answers of the reference's own tests
(tests/test_metrics.py:36-118 in /root/reference - reproduced here as data) and against
independent checks of the two pymoo functions it restates (non-dominated filtering, hypervolume)."""
import numpy as np
import pytest
from scipy.optimize import OptimizeResult

from zfista_amd.metrics import (calculate_metrics, extract_function_values, extract_non_dominated_points,
                                hypervolume, purity, spread_metrics)


@pytest.fixture
def three_results():
    return [
        OptimizeResult(fun=np.array([0.1, 0.2]), success=True, time=1, nit=10, nit_internal=5),
        OptimizeResult(fun=np.array([0.2, 0.1]), success=True, time=2, nit=20, nit_internal=10),
        OptimizeResult(fun=np.array([0.3, 0.3]), success=True, time=3, nit=30, nit_internal=15),
    ]


F3 = np.array([[0.1, 0.2], [0.2, 0.1], [0.3, 0.3]])


def test_reference_known_answers(three_results):
    assert np.array_equal(extract_function_values(three_results), F3)
    assert np.array_equal(extract_non_dominated_points(F3), F3[:2])
    assert purity(F3[:2], F3) == pytest.approx(2 / 3)
    gamma, delta = spread_metrics(F3[:2], F3)
    assert float(gamma) == pytest.approx(0.1) and delta == pytest.approx(0.5)
    metrics, ratios = calculate_metrics(("result", three_results))
    expected = {"Hypervolume": 0, "Gamma": 0.1, "Delta": 0, "Purity": 1.0, "Error rate": 0.0,
                "Avg computation time": 2.0, "Avg iterations": 20.0, "Avg internal iterations": 10.0}
    assert set(metrics) == set(expected) == set(ratios)
    for key, val in expected.items():
        assert metrics[key]["result"] == pytest.approx(val), key
        assert ratios[key]["result"] == pytest.approx(1), key


def test_non_dominated_filter_properties():
    rng = np.random.default_rng(0)
    F = rng.random((200, 3))
    nd = extract_non_dominated_points(F)
    for p in nd:                       # nothing dominates a kept point
        assert not np.any(np.all(F <= p, axis=1) & np.any(F < p, axis=1))
    kept = {tuple(p) for p in nd}
    for p in F:                        # every dropped point is dominated by someone
        if tuple(p) not in kept:
            assert np.any(np.all(F <= p, axis=1) & np.any(F < p, axis=1))
    assert np.array_equal(extract_non_dominated_points(np.array([[1.0, 1.0], [1.0, 1.0]])), [[1.0, 1.0], [1.0, 1.0]])


def test_hypervolume_closed_forms():
    assert hypervolume([[0.0, 0.0]], [1.0, 1.0]) == pytest.approx(1.0)
    assert hypervolume([[0.25, 0.75], [0.75, 0.25]], [1.0, 1.0]) == pytest.approx(0.75 * 0.25 * 2 - 0.25 * 0.25)
    assert hypervolume([[0.1, 0.2], [0.2, 0.1]], [0.2, 0.2]) == 0.0            # on the reference corner
    assert hypervolume([[0.0, 0.0, 0.0]], [1.0, 2.0, 3.0]) == pytest.approx(6.0)
    # two boxes in 3-D: inclusion - exclusion
    a, b, ref = np.array([0.2, 0.6, 0.1]), np.array([0.5, 0.1, 0.4]), np.ones(3)
    exp = np.prod(ref - a) + np.prod(ref - b) - np.prod(ref - np.maximum(a, b))
    assert hypervolume([a, b], ref) == pytest.approx(exp)
    assert hypervolume(np.zeros((0, 2)), [1.0, 1.0]) == 0.0


@pytest.mark.parametrize("m", [2, 3, 4])
def test_hypervolume_monte_carlo(m):
    rng = np.random.default_rng(m)
    P = rng.random((12, m)) * 0.8
    ref = np.ones(m)
    samples = rng.random((400000, m))
    dominated = np.zeros(len(samples), dtype=bool)
    for p in P:
        dominated |= np.all(samples >= p, axis=1)
    assert hypervolume(P, ref) == pytest.approx(dominated.mean(), abs(4e-3))


def test_calculate_metrics_two_solvers_and_missing_nit_internal():
    mk = lambda f, ok=True, t=1.0, nit=5: OptimizeResult(fun=np.array(f), success=ok, time=t, nit=nit)  # noqa: E731
    a = [mk([0.0, 1.0]), mk([1.0, 0.0]), mk([0.6, 0.6], ok=False)]
    b = [mk([0.5, 0.5], t=3.0, nit=9), mk([2.0, 2.0], t=5.0, nit=11)]
    metrics, ratios = calculate_metrics(("a", a), ("b", b))
    # true front: (0,1), (1,0), (.5,.5); (.6,.6) is dominated by (.5,.5)
    assert metrics["Purity"] == {"a": pytest.approx(2 / 3), "b": pytest.approx(1 / 3)}
    assert metrics["Error rate"] == {"a": pytest.approx(1 / 3), "b": 0.0}
    assert metrics["Avg iterations"]["b"] == 10.0 and metrics["Avg computation time"]["a"] == 1.0
    assert metrics["Hypervolume"]["b"] == pytest.approx(0.25) and metrics["Hypervolume"]["a"] == pytest.approx(0.16)
    assert ratios["Hypervolume"]["b"] == 1 and ratios["Hypervolume"]["a"] == pytest.approx(0.25 / 0.16)
    assert np.isnan(metrics["Avg internal iterations"]["a"])     # the solver does not report it (deviation noted)
    assert metrics["Gamma"]["b"] == np.inf                        # one common point only
