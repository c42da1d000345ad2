#!/usr/bin/env python3
"""What the per-pass exchange of a sharded solve costs, measured on ONE GPU: the same solve (P-diag, chains of 16,
clean regime) three ways -
  plain      unsharded: a pass is one launch, the decide pass runs inside it;
  rccl_1     the multi-rank step sequence over a 1-rank RCCL communicator of the library: the trial launch leaves
             the local packs, ncclAllGather (a real RCCL call, world 1) gathers them, zf_decide_kernel decides;
  threads_W  W rank threads on this GPU through the in-process communicator group (n / W elements each): the
             multi-rank sequence with a host barrier per exchange - an upper bound, the ranks share one device.
Prints one JSON object: microseconds per pass for each and the differences.
    tools/exchange_overhead.py [n] [passes] [W]"""
import json
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import LAM, LR, make_inputs  # noqa: E402
from zfista_amd import _lib  # noqa: E402
from zfista_amd.comm import LibComm  # noqa: E402
from zfista_amd.problems import DiagQuadL1  # noqa: E402
from zfista_amd.proximal_gradient import NativeRun  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W = int(sys.argv[3]) if len(sys.argv) > 3 else 8
K = 16 * (passes + 1)   # (+ 1: an untimed first pass and poll, after which the host launches one kernel per pass)
o = dict(lr=LR, tol=0.0, tol_internal=1e-12, max_iter=K, max_backtrack_iter=100, decay_rate=0.5, nesterov=True,
         nesterov_ratio=(0, 0.25), deprecated=False)


def solve(prob, x0, sync, reps=12):
    best = 1e9
    for _ in range(reps):
        run = NativeRun(prob, x0, o)
        run.advance(1)
        sync()
        t0 = time.perf_counter()
        run.enqueue_only(passes)
        run.collect()
        sync()
        best = min(best, time.perf_counter() - t0)
        assert run.nit_seen == K, run.nit_seen
        run.solver.close()
    return best / passes * 1e6


d, c = make_inputs(n, 1, "cuda")
x0 = torch.zeros(n, dtype=torch.float64, device="cuda")
out = dict(n=n, passes=passes, iterations=K)
out["plain_us_per_pass"] = solve(DiagQuadL1(d, c, LAM), x0, torch.cuda.synchronize)
comm = LibComm(0, 1, LibComm.new_unique_id())
out["rccl_1_us_per_pass"] = solve(DiagQuadL1(d, c, LAM, group=comm), x0, torch.cuda.synchronize)
out["rccl_1_minus_plain_us"] = out["rccl_1_us_per_pass"] - out["plain_us_per_pass"]
comm.close()

comms = LibComm.local_group(W, cap_doubles=4096)
bar = threading.Barrier(W)
res = [0.0] * W


def rank_main(r):
    lo, hi = r * n // W, (r + 1) * n // W
    with torch.cuda.stream(torch.cuda.Stream()):
        prob = DiagQuadL1(d[lo:hi].clone(), c[lo:hi].clone(), LAM, group=comms[r])

        def sync():
            torch.cuda.synchronize()
            bar.wait()

        res[r] = solve(prob, x0[lo:hi].clone(), sync)


ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(W)]
for t in ts:
    t.start()
for t in ts:
    t.join()
out[f"threads_{W}_us_per_pass"] = max(res)
out[f"threads_{W}_minus_plain_us"] = max(res) - out["plain_us_per_pass"]
out["note"] = ("per pass of 16 iterations over the same n elements in total; rccl_1 - plain = one all-gather of 16 x 64 B + the "
               "decide kernel + two kernel boundaries; the thread ranks share ONE device (their kernels interleave) and meet at a "
               "host barrier per exchange")
print(json.dumps(out))
