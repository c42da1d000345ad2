"""CPU: the oracle against the round-3 fixtures of the imported reference (G11, G12) - the parts that fit a CPU suite
of minutes.  (At generation time tests/golden/make_golden_r3.py asserts oracle == reference bit for bit on all of it.)"""
import json
import os
import warnings

import numpy as np

from conftest import GOLDEN, rel_err

STRIDE = 997


def _kkt_gap(w, grad):
    return float(np.dot(w, grad - grad.min()))


def test_g11_jos1_n1e6_subproblem_capture_and_first_iterations(golden):
    from oracle import cpu_ref, problems_ref as P

    G = golden("g11_multiobjective_fullsize.npz")
    M = json.load(open(os.path.join(GOLDEN, "g11_meta.json")))["cases"]["jos1_n1e6"]
    n = M["n"]
    ref = P.JOS1Ref(n, l1_ratios=np.arange(1, 3) / n, l1_shifts=[0, 1])
    x0 = np.random.default_rng(1).uniform(-2, 2, n)
    y = x0 + 0.1 * np.random.default_rng(2).standard_normal(n)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        s = cpu_ref.trial_multi(ref.f, ref.g, ref.jac_f, ref.prox_wsum_g, M["lr_sub"], x0, y, np.ones(2) / 2,
                                tol=1e-12, max_iter=100000)
    # (the same NumPy / SciPy: equal up to BLAS summation order of the host - the reference's own spread is the bound)
    fl = M["floor"]
    np.testing.assert_allclose(s.weight, G("jos1_n1e6.sub.weight"), rtol=0, atol=max(1e-12, 10 * fl["sub_w_abs"]))
    assert rel_err(s.x[::STRIDE], G("jos1_n1e6.sub.x_sample")) <= max(1e-12, 10 * fl["sub_x_rel"])
    np.testing.assert_allclose(s.fun, float(G("jos1_n1e6.sub.fun")), rtol=max(1e-12, 10 * fl["sub_fun_rel"]))
    # the recorded KKT gap of the reference's weights is what the oracle's dual gives at them
    J, f_y, F_old = ref.jac_f(y), ref.f(y), ref.f(x0) + ref.g(x0)
    _, grad = cpu_ref.dual_value_and_grad(G("jos1_n1e6.sub.weight"), ref.g, ref.prox_wsum_g, M["lr_sub"], y, J, f_y, F_old)
    np.testing.assert_allclose(grad, G("jos1_n1e6.sub.dual_grad"), rtol=1e-9)
    assert abs(_kkt_gap(G("jos1_n1e6.sub.weight"), grad) - M["sub_kkt_gap"]) <= 1e-6 * M["sub_grad_scale"]
    # three FISTA iterations (the first line search backtracks once): traces and iterates of the reference
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), x0, lr=M["lr"], nesterov=True, tol=0.0, max_iter=3,
                                               return_all=True)
    assert r.alltrials == M["alltrials"][:3]
    tol = max(1e-12, 10 * fl["trace_x_rel"])
    for k in range(4):
        assert rel_err(r.allvecs[k][::STRIDE], G("jos1_n1e6.x_samples")[k]) <= tol
    np.testing.assert_allclose(np.stack(r.allfuns), G("jos1_n1e6.allfuns")[:4], rtol=max(1e-12, 10 * fl["trace_F_rel"]))


def test_g12_first_iterations_and_recorded_rejections(golden):
    """P-diag n = 1e7: the first three iterations of the oracle against the reference's run (elementwise: exact up to
    the host's summation order in F), and the fixture's own record of where the reference rejects."""
    from oracle import cpu_ref, problems_ref as P

    G = golden("g12_noise_floor_diag_n1e7.npz")
    n = int(G("n"))
    d, c, lam = P.make_pdiag(n, seed=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = cpu_ref.minimize_proximal_gradient(*P.DiagQuadL1Ref(d, c, lam).callbacks(), np.zeros(n), lr=float(G("lr")),
                                               nesterov=True, tol=0.0, max_iter=3, return_all=True)
    np.testing.assert_allclose(r.allfuns, G("allfuns")[:4], rtol=1e-13)
    np.testing.assert_allclose(r.allerrs, G("allerrs")[:3], rtol=1e-13)
    trials = G("alltrials")
    assert (np.nonzero(trials > 1)[0] + 1).tolist() == [90, 94, 95, 97, 102] and int(G("first_rejection")) == 90
    assert float(G("alllrs")[-1]) == 0.45 * 0.5 ** 5 and int(G("rejections")) == 5
