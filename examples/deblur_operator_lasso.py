#!/usr/bin/env python3
"""Operator-form LASSO on the GPU: the image-deblurring workload of the reference's
``examples/cameraman.ipynb`` (cells 6-11: 9x9 Gaussian blur, symmetric boundary, one Haar level,
f = |B W^-1 x - b|^2, g = l1 |x|_1, lr = 1/L, decay_rate = 1, FISTA) on a synthetic 256 x 256 image
(scikit-image / PyWavelets are not in this image), three ways: as the recognised DEVICE-RESIDENT problem
``zfista_amd.problems.BlurHaarL1`` (library kernels for B W^-1 and W B, no host synchronisation per
iteration), through tensor callbacks (blur and Haar written with torch ops, three scalars per trial cross
PCIe), and - as the checker - the notebook's NumPy / SciPy callbacks under the CPU oracle.

    python examples/deblur_operator_lasso.py [--size 256] [--iters 200] [--cpu-iters 20]

The reference records 7.7 it/s for this problem (517 iterations in 66.7 s, 8 loky workers,
``examples/data/cameraman_ab.csv:3``).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle.operator_ref import L1_RATIO, BlurHaarL1Ref, make_deblur   # the notebook's callbacks in NumPy / SciPy (the checker)


# ---- the same operators on device tensors -----------------------------------------------------
def tensor_problem(kernel, observed):
    import torch
    import torch.nn.functional as F

    kt = torch.from_numpy(kernel).cuda()[None, None]
    obs = torch.from_numpy(observed).cuda()
    h = observed.shape[0] // 2

    def dwt(img):
        a, b, c, d = img[0::2, 0::2], img[0::2, 1::2], img[1::2, 0::2], img[1::2, 1::2]
        return torch.stack([(a + b + c + d) / 2, (a + b - c - d) / 2, (a - b + c - d) / 2, (a - b - c + d) / 2]).flatten()

    def idwt(vec):
        cA, cH, cV, cD = vec.reshape(4, h, h)
        img = torch.empty((2 * h, 2 * h), dtype=vec.dtype, device=vec.device)
        img[0::2, 0::2] = (cA + cH + cV + cD) / 2
        img[0::2, 1::2] = (cA + cH - cV - cD) / 2
        img[1::2, 0::2] = (cA - cH + cV - cD) / 2
        img[1::2, 1::2] = (cA - cH - cV + cD) / 2
        return img

    def blur(img):   # correlate2d(mode="same", boundary="symm"): edge-including mirror padding
        p = torch.cat([img[:, :4].flip(1), img, img[:, -4:].flip(1)], dim=1)
        p = torch.cat([p[:4].flip(0), p, p[-4:].flip(0)], dim=0)
        return F.conv2d(p[None, None], kt)[0, 0]

    f = lambda x: torch.sum((blur(idwt(x)) - obs) ** 2)                                           # noqa: E731
    jac_f = lambda x: 2 * dwt(blur(blur(idwt(x)) - obs))                                          # noqa: E731
    g = lambda x: L1_RATIO * torch.sum(torch.abs(x))                                               # noqa: E731
    prox = lambda w, x: torch.where(torch.abs(x) <= L1_RATIO * w, torch.zeros_like(x),           # noqa: E731
                                    x - L1_RATIO * w * torch.sign(x))
    return (f, g, jac_f, prox), dwt, idwt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--cpu-iters", type=int, default=20)
    a = ap.parse_args()
    import torch

    from oracle import cpu_ref
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.problems import BlurHaarL1

    kernel, observed, x0, L = make_deblur(a.size)
    kw = dict(lr=1 / L, decay_rate=1, nesterov=True, tol=0.0)
    cbs_np = BlurHaarL1Ref(kernel, observed).callbacks()
    cbs_t, _, _ = tensor_problem(kernel, observed)
    native = BlurHaarL1(kernel, observed, L1_RATIO)
    x0_t = torch.from_numpy(x0).cuda()

    def timed(cbs, start, iters):
        minimize_proximal_gradient(*cbs, start, max_iter=3, **kw)          # warm-up (kernel caches)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = minimize_proximal_gradient(*cbs, start, max_iter=iters, **kw)
        torch.cuda.synchronize()
        return res, time.perf_counter() - t0

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res_n, t_native = timed(native.callbacks(), x0, a.iters)
        res_long, t_long = timed(native.callbacks(), x0, 20 * a.iters)
        res_t, t_tensor = timed(cbs_t, x0_t, a.iters)
        t0 = time.perf_counter()
        exp = cpu_ref.minimize_proximal_gradient(*cbs_np, x0, max_iter=a.cpu_iters, **kw)
        t_cpu = time.perf_counter() - t0
        chk_n = minimize_proximal_gradient(*native.callbacks(), x0, max_iter=a.cpu_iters, **kw)
        chk_t = minimize_proximal_gradient(*cbs_t, x0_t, max_iter=a.cpu_iters, **kw)
    rel_n = float(np.linalg.norm(chk_n.x - exp.x) / np.linalg.norm(exp.x))
    rel_t = float(np.linalg.norm(chk_t.x.cpu().numpy() - exp.x) / np.linalg.norm(exp.x))
    print(json.dumps({
        "workload": f"operator-form LASSO (9x9 Gaussian blur, 1 Haar level), n = {a.size * a.size}, FISTA (0, 1/4), "
                    "lr = 1/L, decay_rate = 1",
        "gpu_device_resident_it_per_s": a.iters / t_native, "gpu_iterations": a.iters,
        "gpu_device_resident_it_per_s_long_solve": 20 * a.iters / t_long, "gpu_iterations_long_solve": 20 * a.iters,
        "gpu_tensor_callbacks_it_per_s": a.iters / t_tensor,
        "cpu_numpy_oracle_it_per_s": a.cpu_iters / t_cpu, "cpu_iterations": a.cpu_iters,
        "rel_err_device_resident_vs_cpu": rel_n, "rel_err_tensor_callbacks_vs_cpu": rel_t,
        "reference_recorded_it_per_s": 7.7,
        "F_final_device_resident": float(np.asarray(res_n.fun).reshape(-1)[0]),
        "F_final_tensor_callbacks": float(np.asarray(res_t.fun).reshape(-1)[0]),
    }))


if __name__ == "__main__":
    main()
