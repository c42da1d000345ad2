#!/usr/bin/env python3
"""Generic path (opaque NumPy callbacks, solver arithmetic on the GPU) against the CPU oracle on
the same callbacks: what a user of the reference's closure style sees for small problems."""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref, problems_ref as P  # noqa: E402
from zfista_amd import minimize_proximal_gradient  # noqa: E402

warnings.simplefilter("ignore")
for (m, n) in ((512, 1024), (2048, 8192)):
    A, b, lam = P.make_plasso(m, n, seed=0)
    cb = P.LeastSquaresL1Ref(A, b, lam).callbacks()   # plain closures: not recognised as a native operator
    cbs = tuple((lambda fn: (lambda *a: fn(*a)))(fn) for fn in cb)
    kw = dict(lr=1e-4 if n > 2000 else 2 ** -11, nesterov=True, tol=0.0, max_iter=100)
    minimize_proximal_gradient(*cbs, np.zeros(n), **dict(kw, max_iter=3))
    t0 = time.perf_counter(); r = minimize_proximal_gradient(*cbs, np.zeros(n), **kw); t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter(); e = cpu_ref.minimize_proximal_gradient(*cbs, np.zeros(n), **kw); t_cpu = time.perf_counter() - t0
    rel = np.linalg.norm(r.x - e.x) / max(np.linalg.norm(e.x), 1e-300)
    print(f"LASSO {m}x{n} NumPy callbacks: generic path {100 / t_gpu:8.1f} it/s, CPU oracle {100 / t_cpu:8.1f} it/s, rel-err {rel:.1e}")
