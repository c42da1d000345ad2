#!/usr/bin/env python3
"""Clean-regime rate of the separable solve by size, with and without the persistent multi-pass kernel
(ZF_PERSIST=0), same process, alternating: n, it/s, microseconds per pass.   tools/r4_sizes.py [K] [W]"""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import LAM, LR, make_inputs  # noqa: E402
from zfista_amd import _lib  # noqa: E402
from zfista_amd.problems import DiagQuadL1  # noqa: E402
from zfista_amd.proximal_gradient import NativeRun  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sizes = [int(float(a)) for a in sys.argv[3:]] or [30000, 100000, 1000000, 2000000, 5000000, 10000000, 20000000]


def block(prob, x0, o):
    run = NativeRun(prob, x0, dict(o, max_iter=W))
    S = run.sub_iters
    while run.status == _lib.ZF_RUNNING:
        run.advance((W - run.nit_seen + S - 1) // S)
    run.set_max_iter(W + K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while run.status == _lib.ZF_RUNNING:
        run.enqueue_only((W + K - run.nit_seen + S - 1) // S)
        run.collect()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert run.nit_seen == W + K
    pc = run.solver.persist_counts()
    steps = run.solver.launch_counts()
    run.solver.close()
    return dt, pc, steps


for n in sizes:
    d, c = make_inputs(n, 1, "cuda")
    prob = DiagQuadL1(d, c, LAM)
    x0 = torch.zeros(n, dtype=torch.float64, device="cuda")
    o = dict(lr=LR, tol=0.0, tol_internal=1e-12, max_iter=W, max_backtrack_iter=100, decay_rate=0.5, nesterov=True,
             nesterov_ratio=(0, 0.25), deprecated=False, sub_iters=16)
    res = {"persist": [], "per_pass": []}
    info = {}
    for rep in range(40):
        for mode in ("persist", "per_pass"):
            os.environ["ZF_PERSIST"] = "0" if mode == "per_pass" else "1"
            dt, pc, steps = block(prob, x0, o)
            res[mode].append(dt)
            info[mode] = (pc, steps)
    os.environ.pop("ZF_PERSIST", None)
    passes = (K + 15) // 16
    out = dict(n=n, K=K, W=W, passes=passes)
    for mode in res:
        med = statistics.median(res[mode][5:])
        out[mode] = dict(it_per_s=K / med, us_per_pass=med / passes * 1e6, persist_counts=info[mode][0], steps_kernels=info[mode][1])
    print(json.dumps(out), flush=True)
