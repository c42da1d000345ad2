#!/usr/bin/env python3
"""What a poll and a chunk of passes cost the host (P-diag, chains of 16): an idle zf_solver_poll against a bare
stream synchronise, and the round trip of 1 and of 8 passes + poll -> time per pass and fixed cost per chunk."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from bench import LAM, LR, make_inputs
from zfista_amd import _lib
from zfista_amd.problems import DiagQuadL1
from zfista_amd.proximal_gradient import NativeRun
for n in (10**5, 10**7, 10**5):
    d, c = make_inputs(n, 1, "cuda")
    o = dict(lr=LR, tol=0.0, tol_internal=1e-12, max_iter=16 * 4000, max_backtrack_iter=100, decay_rate=0.5, nesterov=True,
             nesterov_ratio=(0, 0.25), deprecated=False)
    run = NativeRun(DiagQuadL1(d, c, LAM), torch.zeros(n, dtype=torch.float64, device="cuda"), o, timing=False)
    run.advance(1)
    for _ in range(50):
        run.solver.poll()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        run.solver.poll()
    t_poll = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(200):
        torch.cuda.synchronize()
    t_sync = (time.perf_counter() - t0) / 200
    # one pass + poll round trip, against 8 passes + poll
    def rt(steps, reps=50):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run.enqueue_only(steps)
            run.collect()
            best = min(best, time.perf_counter() - t0)
        return best
    r1, r8 = rt(1), rt(8)
    # what the HOST spends enqueuing a pass (returns before the device has run it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.enqueue_only(16)
    t_enq = (time.perf_counter() - t0) / 16
    run.collect()
    print(f"n={n:.0e} idle poll {t_poll*1e6:.1f} us, idle synchronize {t_sync*1e6:.1f} us, 1 pass + poll {r1*1e6:.1f} us, 8 passes + poll {r8*1e6:.1f} us -> per pass {(r8-r1)/7*1e6:.1f} us, fixed {(r1-(r8-r1)/7)*1e6:.1f} us; host time to enqueue a pass {t_enq*1e6:.1f} us")
    run.solver.close()
