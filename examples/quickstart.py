#!/usr/bin/env python3
"""zfista_amd in five calls: the reference's API (zfista.minimize_proximal_gradient) on an MI355X.

    python examples/quickstart.py
"""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zfista_amd import minimize_proximal_gradient  # noqa: E402
from zfista_amd.problems import FDS, JOS1, DiagQuadL1, LeastSquaresL1  # noqa: E402

warnings.simplefilter("ignore")
rng = np.random.default_rng(0)

# 1. l1-regularised diagonal quadratic: device-resident, 8 FISTA iterations per pass over the data
n = 1_000_000
d, c = rng.uniform(0.5, 2.0, n), rng.standard_normal(n)
prob = DiagQuadL1(d, c, lam=0.1)
res = minimize_proximal_gradient(*prob.callbacks(), np.zeros(n), lr=0.45, nesterov=True, tol=1e-8)
x_star = np.sign(c) * np.maximum(np.abs(c) - 0.1 / d, 0)
print(f"diag n={n}: nit={res.nit} |x - x*|_inf={np.max(np.abs(res.x - x_star)):.1e} ({res.message})")

# 2. dense LASSO 1/2 |Ax - b|^2 + lam |x|_1 with backtracking from lr = 1
A = rng.standard_normal((512, 1024))
xt = np.zeros(1024)
xt[:20] = rng.standard_normal(20)
b = A @ xt + 0.01 * rng.standard_normal(512)
lam = 0.1 * np.max(np.abs(A.T @ b))
res = minimize_proximal_gradient(*LeastSquaresL1(A, b, lam).callbacks(), np.zeros(1024), nesterov=True, tol=1e-6)
print(f"lasso 512x1024: nit={res.nit} nonzeros={int(np.sum(res.x != 0))} F={res.fun:.6f}")

# 3. the same LASSO through arbitrary NumPy callbacks (solver arithmetic still on the GPU)
f = lambda x: 0.5 * np.sum((A @ x - b) ** 2)                                # noqa: E731
g = lambda x: lam * np.sum(np.abs(x))                                       # noqa: E731
jac_f = lambda x: A.T @ (A @ x - b)                                         # noqa: E731
prox = lambda w, x: np.sign(x) * np.maximum(np.abs(x) - lam * w, 0)         # noqa: E731
res2 = minimize_proximal_gradient(f, g, jac_f, prox, np.zeros(1024), nesterov=True, tol=1e-6)
print(f"same problem, opaque callbacks: nit={res2.nit} |dx|={np.linalg.norm(res2.x - res.x):.1e}")

# 4. multi-objective problems of zfista/problems.py (weak Pareto points from random starts)
jos = JOS1(n_features=1000, l1_ratios=[1e-3, 2e-3], l1_shifts=[0, 1])
res = jos.minimize_proximal_gradient(rng.uniform(-2, 4, 1000), nesterov=True)
print(f"JOS1 n=1000: nit={res.nit} F={np.round(res.fun, 6)}")
fds = FDS(n_features=10)
res = fds.minimize_proximal_gradient(rng.uniform(-2, 2, 10), nesterov=True, lr=0.05, max_iter=200)
print(f"FDS n=10: nit={res.nit} F={np.round(res.fun, 4)} success={res.success}")
