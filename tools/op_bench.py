#!/usr/bin/env python3
"""Operator-form LASSO (zfista_amd.problems.BlurHaarL1: blur o inverse Haar, the reference notebook's problem) beyond the
notebook's size: the device-resident solve alone, timed over --iters iterations after a warm-up.

    python tools/op_bench.py --size 4096 [--iters 200] [--check 3]

Prints one JSON line: it/s, ms per iteration, the algorithmic HBM bytes of an iteration (two operator applications of
16 B per pixel, the prox step's 32 B per element, the residual / linearity traffic) and the fraction of the 8 TB/s peak
they amount to; --check K compares K iterations with the CPU oracle on the same callbacks (slow beyond 1024 x 1024)."""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--check", type=int, default=0)
    ap.add_argument("--k", type=int, default=0, help="another blur size (odd, 3 .. 15): a normalised Gaussian window, lr = 1/2")
    ap.add_argument("--general", action="store_true", help="with --k: a rank-2 kernel (the general correlation path)")
    a = ap.parse_args()
    import torch

    from oracle.operator_ref import L1_RATIO, BlurHaarL1Ref, make_deblur
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.problems import BlurHaarL1

    kernel, observed, x0, L = make_deblur(a.size)   # (the 9 x 9 Gaussian window of the notebook)
    if a.k:
        from oracle.operator_ref import gaussian_kernel

        kernel = gaussian_kernel(a.k, 2.0)
        if a.general:
            kernel = kernel + 0.3 * np.outer(np.arange(a.k), np.ones(a.k)) / a.k
        kernel = kernel / kernel.sum()
        L = 2.0
    kw = dict(lr=1 / L, decay_rate=1, nesterov=True, tol=0.0)
    native = BlurHaarL1(kernel, observed, L1_RATIO)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        minimize_proximal_gradient(*native.callbacks(), x0, max_iter=5, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = minimize_proximal_gradient(*native.callbacks(), x0, max_iter=a.iters, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    n = a.size * a.size
    # per iteration: adjoint kernel reads s_k, s_{k-1}, b (24 B) and writes grad (8 B); prox step reads x_k, x_{k-1}, grad (24 B),
    # writes x+ (8 B); apply kernel reads x+ (8 B) and b (8 B), writes s+ (8 B)
    bytes_iter = 88 * n
    out = {"workload": f"operator-form LASSO, {a.size} x {a.size}, {kernel.shape[0]} x {kernel.shape[0]} {'rank-2 kernel' if a.general else 'Gaussian window'}, FISTA, lr = 1/L, decay_rate = 1",
           "n": n, "iterations": a.iters, "it_per_s": a.iters / dt, "ms_per_iteration": dt / a.iters * 1e3,
           "algorithmic_bytes_per_iteration": bytes_iter,
           "hbm_fraction_of_8TBps": bytes_iter / (dt / a.iters) / 8e12, "F_final": float(np.asarray(res.fun).reshape(-1)[0])}
    if a.check:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            from oracle import cpu_ref

            exp = cpu_ref.minimize_proximal_gradient(*BlurHaarL1Ref(kernel, observed).callbacks(), x0, max_iter=a.check, **kw)
            got = minimize_proximal_gradient(*native.callbacks(), x0, max_iter=a.check, **kw)
        out["rel_err_vs_cpu_oracle"] = float(np.linalg.norm(got.x - exp.x) / np.linalg.norm(exp.x))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
