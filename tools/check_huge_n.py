#!/usr/bin/env python3
"""n beyond 2^31 elements on one GPU (default 2.5e9: 20 GB per vector, ~140 GB in all): the 64-bit
indexing of the fused kernel.  A strided sample of the iterate - including the very last elements -
must equal the element recursion of the oracle (P-diag is separable: every element evolves on its
own once the accept / reject decisions are known; lr = 0.45 accepts every trial)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import problems_ref as P  # noqa: E402
from zfista_amd import _lib  # noqa: E402
from zfista_amd.engine import momentum_factors  # noqa: E402
from zfista_amd.problems import DiagQuadL1  # noqa: E402
from zfista_amd.proximal_gradient import NativeRun  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_500_000_000
K = 16
gen = torch.Generator(device="cuda").manual_seed(3)
d = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) * 1.5 + 0.5
c = torch.randn(n, dtype=torch.float64, device="cuda", generator=gen)
lam, lr = 0.1, 0.45
o = dict(lr=lr, tol=0.0, tol_internal=1e-12, max_iter=K, max_backtrack_iter=100, decay_rate=0.5, nesterov=True,
         nesterov_ratio=(0, 0.25), deprecated=False)
run = NativeRun(DiagQuadL1(d, c, lam), torch.zeros(n, dtype=torch.float64, device="cuda"), o)
t0 = time.time()
rows = []
while run.status == _lib.ZF_RUNNING:
    rows.append(run.advance(4))
rows = np.concatenate(rows)
dt = time.time() - t0
assert run.nit_seen == K and np.all(rows[:, _lib.TR_TRIALS] == 1) and np.all(rows[:, _lib.TR_LR] == lr)
xptr = run.solver.x_dev_ptr()
idx = torch.cat([torch.arange(0, n, 100_000_007, device="cuda"), torch.arange(n - 5, n, device="cuda"),
                 torch.tensor([2**31 - 1, 2**31, 2**31 + 1, 2**32 - 1, 2**32, 2**32 + 1], device="cuda")])
idx = idx[idx < n]
# view the solver's x_k without copying 20 GB to the host
import ctypes
xs = torch.empty(idx.numel(), dtype=torch.float64, device="cuda")
full = torch.empty(0, dtype=torch.float64, device="cuda")
# torch has no from-pointer constructor: gather through the library's D2D copy, one element each
lib = _lib.load()
for k, i in enumerate(idx.tolist()):
    _lib.check(lib.zf_memcpy_d2d(ctypes.c_void_p(xs.data_ptr() + 8 * k), ctypes.c_void_p(xptr + 8 * i), 8, None))
torch.cuda.synchronize()
got = xs.cpu().numpy()
ds, cs = d[idx].cpu().numpy(), c[idx].cpu().numpy()
betas = np.concatenate([[0.0], momentum_factors(K, (0, 0.25))[0]])
xk = np.zeros_like(ds)
xo = xk.copy()
for k in range(K):
    y = xk + betas[k] * (xk - xo)
    xn = P.soft_threshold(y - lr * (ds * (y - cs)), lam * lr)
    xo, xk = xk, xn
assert np.array_equal(got, xk), (got, xk)
print(f"n = {n:.3e}: {K} iterations in {dt:.2f} s ({K / dt:.0f} it/s), sample of {idx.numel()} elements incl. the last five "
      f"and the 2^31 / 2^32 boundaries equals the oracle recursion bit for bit; F = {rows[-1, _lib.TR_F]:.6e}")
