"""Soak: repeated solves + one long solve on the GPU (leaks, stability); run by hand on the GPU box."""
import os
import sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_inputs, LAM, LR
from zfista_amd import _lib
from zfista_amd.problems import DiagQuadL1, LeastSquaresL1
from zfista_amd.proximal_gradient import NativeRun
o = dict(lr=LR, tol=0.0, tol_internal=1e-12, max_iter=300, max_backtrack_iter=100, decay_rate=0.5, nesterov=True,
         nesterov_ratio=(0, 0.25), deprecated=False)
n = 10**7
d, c = make_inputs(n, 1, "cuda")
prob = DiagQuadL1(d, c, LAM)
free0 = torch.cuda.mem_get_info()[0]
t0 = time.time()
for k in range(40):
    run = NativeRun(prob, torch.zeros(n, dtype=torch.float64, device="cuda"), o)
    while run.status == _lib.ZF_RUNNING:
        run.advance(16)
    assert run.nit_seen == 300
    run.solver.close()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print(f"40 diag solves of 300 iterations at n=1e7: {time.time()-t0:.1f} s; device memory drift {(free0-free1)/2**20:.1f} MiB")
A = torch.randn(2048, 4096, dtype=torch.float64, device="cuda"); b = torch.randn(2048, dtype=torch.float64, device="cuda")
pl = LeastSquaresL1(A, b, 5.0)
for k in range(40):
    run = NativeRun(pl, torch.zeros(4096, dtype=torch.float64, device="cuda"), dict(o, lr=1e-4, max_iter=200))
    while run.status == _lib.ZF_RUNNING:
        run.advance(64)
    run.solver.close()
torch.cuda.synchronize()
free2 = torch.cuda.mem_get_info()[0]
print(f"40 lasso solves: device memory drift {(free1-free2)/2**20:.1f} MiB")
# one long solve at the headline size
n = 10**8
d, c = make_inputs(n, 1, "cuda")
run = NativeRun(DiagQuadL1(d, c, LAM), torch.zeros(n, dtype=torch.float64, device="cuda"), dict(o, max_iter=3000))
t0 = time.time(); passes = 0
while run.status == _lib.ZF_RUNNING:
    run.advance(32); passes += 32
torch.cuda.synchronize()
ctl = run.solver.ctl
print(f"n=1e8, 3000 iterations: {time.time()-t0:.2f} s, nit={ctl.nit} status={ctl.status} lr={ctl.lr:.3g} trials={ctl.total_trials} err={ctl.err:.3g}")
