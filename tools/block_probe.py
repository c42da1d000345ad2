"""Where the time of one timed block (K iterations + one look at the device) goes at sizes where the kernels are short.

    python tools/block_probe.py --n 1000000 --steps 64 [--events]

Times, per block of K accepted iterations (median over --reps fresh extensions of one solve): the enqueue call on the host,
the collect call (poll: DMA + synchronize + bookkeeping), the two device synchronisations bench.py brackets a block with.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--reps", type=int, default=60)
    ap.add_argument("--events", action="store_true", help="kernel events on, as bench.py's default")
    ap.add_argument("--acceptance", default=None)
    args = ap.parse_args()
    from bench import LAM, LR, make_inputs
    from zfista_amd import _lib
    from zfista_amd.problems import DiagQuadL1
    from zfista_amd.proximal_gradient import NativeRun

    n, K = args.n, args.steps
    d, c = make_inputs(n, seed=1, device="cuda")
    prob = DiagQuadL1(d, c, LAM)
    opts = dict(lr=LR, tol=0.0, tol_internal=1e-12, max_iter=16, max_backtrack_iter=100, decay_rate=0.5, nesterov=True,
                nesterov_ratio=(0, 0.25), deprecated=False, sub_iters=16, acceptance=args.acceptance)
    x0 = torch.zeros(n, dtype=torch.float64, device="cuda")
    rows = []
    polls = []
    run = None
    for rep in range(args.reps):
        if run is None or run.nit_seen + K > 80:   # stay in front of the noise floor: a fresh solve every block
            if run is not None:
                run.solver.close()
            run = NativeRun(prob, x0, opts, timing=args.events)
            S = run.sub_iters
            while run.status == _lib.ZF_RUNNING:
                run.advance(1)
        nit0 = run.nit_seen
        run.set_max_iter(nit0 + K)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run.enqueue_only((K + S - 1) // S)
        t1 = time.perf_counter()
        run.solver.poll()            # (what collect() starts with: the library call alone)
        tp = time.perf_counter()
        run.collect()                # (a second poll on an idle stream + the host's bookkeeping)
        t2 = time.perf_counter()
        polls.append((tp - t1, t2 - tp))
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        assert run.nit_seen - nit0 == K and run.status == _lib.ZF_MAXITER, (run.nit_seen - nit0, run.status)
        rows.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
    a = np.array(rows[5:]) * 1e6
    med = np.median(a, axis=0)
    pm = np.median(np.array(polls[5:]) * 1e6, axis=0)
    print(json.dumps(dict(first_poll_us=round(float(pm[0]), 1), idle_poll_and_bookkeeping_us=round(float(pm[1]), 1))), file=sys.stderr)
    print(json.dumps(dict(n=n, K=K, events=bool(args.events), passes=(K + 15) // 16, enqueue_us=round(float(med[0]), 1),
                          collect_us=round(float(med[1]), 1), sync_us=round(float(med[2]), 1), block_us=round(float(med[3]), 1),
                          us_per_pass=round(float(med[3]) / ((K + 15) // 16), 1), it_per_s=round(K / med[3] * 1e6))))


if __name__ == "__main__":
    main()
