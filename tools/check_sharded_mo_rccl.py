#!/usr/bin/env python3
"""Sharded multi-objective solve over a real torch.distributed NCCL (= RCCL) group.  Launch with
torch.distributed.run; with one rank it exercises the RCCL exchange path on a one-GPU box:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 tools/check_sharded_mo_rccl.py
"""
import os
import sys
import warnings

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zfista_amd.problems import JOS1  # noqa: E402

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0))))
n = 100003
x0 = np.random.default_rng(1).uniform(-1, 1, n)
kw = dict(lr=1.0, nesterov=True, tol=1e-9, max_iter=6)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    prob = JOS1(n, l1_ratios=np.arange(1, 3) / n, l1_shifts=[0, 1], group=dist.group.WORLD)
    lo, hi = prob.shard_bounds()
    res = prob.minimize_proximal_gradient(x0[lo:hi], **kw)
    if world == 1:
        full = JOS1(n, l1_ratios=np.arange(1, 3) / n, l1_shifts=[0, 1]).minimize_proximal_gradient(x0, **kw)
        assert res.nit == full.nit and np.array_equal(res.x, full.x), "one-rank group must reproduce the unsharded solve"
print(f"rank {rank}/{world}: nit={res.nit} fun={res.fun} exchanges={prob._engine().n_exchanges} ok")
dist.barrier()
dist.destroy_process_group()
