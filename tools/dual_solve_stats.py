import sys, numpy as np, warnings
sys.path.insert(0, '.')
from zfista_amd.multiobjective import X_K
from zfista_amd.problems import FDS
n = 10**6
p = FDS(n, l1_ratios=np.arange(1, 4) / n, l1_shifts=[0, 1, 2])
x0 = np.random.default_rng(1).uniform(-2, 2, n)
eng = p._engine(); eng.set_x0(x0)
f0, g0 = eng.eval_F(X_K); f_y = eng.prepare()
for lr in (1e-7, 1e-9, 1e-7):
    out = eng.solve_dual_device(lr, f_y, f0 + g0, False, None, 1e-12, 100000)
    print(lr, out[2], eng.solve_stats())
