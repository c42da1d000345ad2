import sys, time, warnings, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zfista_amd import minimize_proximal_gradient
from zfista_amd.problems import JOS1
n = 10**6
for solver in ("device", "native"):
    p = JOS1(n, l1_ratios=np.array([1.0, 2.0]) / n, l1_shifts=[0.0, 1.0])
    x0 = np.random.default_rng(1).uniform(-2, 4, n)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        minimize_proximal_gradient(*p.callbacks(), x0, lr=0.4 * n, nesterov=True, tol=0.0, max_iter=3, dual_solver=solver)
        p._engine().n_dual_evals = 0
        t0 = time.perf_counter()
        K = 200 if solver == "device" else 50
        res = minimize_proximal_gradient(*p.callbacks(), x0, lr=0.4 * n, nesterov=True, tol=0.0, max_iter=K, dual_solver=solver)
        dt = time.perf_counter() - t0
    ev = p._engine().n_dual_evals
    print(solver, "it/s", res.nit / dt, "evals/it", ev / res.nit, "evals/s", ev / dt)
    if solver == "device":
        print(p._engine().solve_stats())
