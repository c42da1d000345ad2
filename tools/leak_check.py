import sys
sys.path.insert(0, "/root/repo")
import torch
from bench import make_inputs, LAM, LR
from zfista_amd import _lib
from zfista_amd.problems import DiagQuadL1, JOS1
from zfista_amd.proximal_gradient import NativeRun
import numpy as np
o = dict(lr=LR, tol=0.0, tol_internal=1e-12, max_iter=50, max_backtrack_iter=100, decay_rate=0.5, nesterov=True,
         nesterov_ratio=(0, 0.25), deprecated=False)
n = 10**7
d, c = make_inputs(n, 1, "cuda")
prob = DiagQuadL1(d, c, LAM)
x0 = torch.zeros(n, dtype=torch.float64, device="cuda")
def once():
    run = NativeRun(prob, x0, o)
    while run.status == _lib.ZF_RUNNING:
        run.advance(16)
    run.solver.close()
once(); torch.cuda.synchronize()
base = torch.cuda.mem_get_info()[0]
for rnd in range(4):
    for k in range(25):
        once()
    torch.cuda.synchronize()
    print(f"after {(rnd+1)*25} more solves: drift {(base - torch.cuda.mem_get_info()[0])/2**20:.1f} MiB")
import warnings
warnings.simplefilter("ignore")
p = JOS1(100000, l1_ratios=[1e-5, 2e-5], l1_shifts=[0, 1])
xx = np.random.default_rng(0).uniform(-1, 1, 100000)
p.minimize_proximal_gradient(xx, max_iter=3); torch.cuda.synchronize()
base = torch.cuda.mem_get_info()[0]
for k in range(30):
    q = JOS1(100000, l1_ratios=[1e-5, 2e-5], l1_shifts=[0, 1]); q.minimize_proximal_gradient(xx, max_iter=3); q._engine().close()
torch.cuda.synchronize()
print(f"30 multi-objective engines: drift {(base - torch.cuda.mem_get_info()[0])/2**20:.1f} MiB")
