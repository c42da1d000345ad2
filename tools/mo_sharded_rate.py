#!/usr/bin/env python3
"""cfg4 (FDS m = 3, n = 1e6 + shifted l1, lr = 1e-7) with x sharded over a library communicator - here ONE rank through a real
RCCL communicator (world 1: every exchange is an ncclAllGather) - with the host-driven search (dual_solver="native": one host
synchronisation per batch) and the device-driven one (dual_solver="device": zf_mo_solve_dual_stream), beside the unsharded
device search (one persistent kernel per trial).      tools/mo_sharded_rate.py [iterations]"""
import json
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zfista_amd.comm import LibComm  # noqa: E402
from zfista_amd.problems import FDS  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n = 10**6
x0 = np.random.default_rng(1).uniform(-2, 2, n)
mk = lambda g: FDS(n, l1_ratios=np.arange(1, 4) / n, l1_shifts=[0, 1, 2], group=g)   # noqa: E731
out = {"workload": "cfg4 FDS m=3 n=1e6 + l1, lr=1e-7", "iterations": K}
comm = LibComm(0, 1, LibComm.new_unique_id())
for tag, group, solver in (("sharded_world1_native", comm, "native"), ("sharded_world1_device_driven", comm, "device"),
                           ("unsharded_device", None, "device")):
    prob = mk(group)
    kw = dict(lr=1e-7, nesterov=True, tol=0.0, dual_solver=solver)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        prob.minimize_proximal_gradient(x0, max_iter=5, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = prob.minimize_proximal_gradient(x0, max_iter=K, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    eng = prob._engine()
    out[tag] = {"it_per_s": K / dt, "us_per_iteration": dt / K * 1e6, "dual_search_trials": res.get("dual_search_trials"),
                "dual_evaluations_per_iteration": eng.n_dual_evals / (K + 5), "collectives_per_iteration": eng.exchange_count() / (K + 5) if group else 0,
                "fun": [float(v) for v in np.asarray(res.fun)]}
comm.close()
print(json.dumps(out))
