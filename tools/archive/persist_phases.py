#!/usr/bin/env python3
"""Where a pass of the persistent kernel spends its time (debug build: make EXTRA=-DZF_PERSIST_DEBUG LIB=... OBJDIR=...;
ZF_LIB_PATH=that library).  Every workgroup stamps the phases of every pass (100 MHz wall clock): start, control
block fetched, body done, tail done, barrier passed.   tools/persist_phases.py [n] [passes]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import LAM, LR, make_inputs  # noqa: E402
from zfista_amd import _lib  # noqa: E402
from zfista_amd.problems import DiagQuadL1  # noqa: E402
from zfista_amd.proximal_gradient import NativeRun  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**7
P = int(sys.argv[2]) if len(sys.argv) > 2 else 6
W = 16
d, c = make_inputs(n, 1, "cuda")
prob = DiagQuadL1(d, c, LAM)
x0 = torch.zeros(n, dtype=torch.float64, device="cuda")
o = dict(lr=LR, tol=0.0, tol_internal=1e-12, max_iter=W, max_backtrack_iter=100, decay_rate=0.5, nesterov=True,
         nesterov_ratio=(0, 0.25), deprecated=False, sub_iters=16)
buf = torch.zeros(max(P, 8) * 512 * 8, dtype=torch.int64, device="cuda")
os.environ["ZF_PERSIST_DBG"] = str(buf.data_ptr())
os.environ.setdefault("ZF_PERSIST", "1")
for rep in range(3):
    buf.zero_()
    run = NativeRun(prob, x0, dict(o, max_iter=W))
    while run.status == _lib.ZF_RUNNING:
        run.advance(1)
    run.set_max_iter(W + 16 * P + 32)
    run.enqueue_only(P)
    run.collect()
    G = run.solver.launch_counts()
    run.solver.close()
if os.environ.get("ZF_PERSIST") == "0":
    # per-pass launches: the full-chain kernel stamps slot pass_seq % 8 (start, body done, tail done)
    raw = buf.cpu().numpy()
    G = int(raw[6::8].max())
    t = raw[:8 * G * 8].reshape(8, G, 8).astype(np.float64) / 100.0
    for slot in range(8):
        if not (t[slot, :, 0] > 0).all():
            continue
        s0, b, tl = t[slot, :, 0], t[slot, :, 2], t[slot, :, 3]
        base = s0.min()
        hw = raw[:8 * G * 8].reshape(8, G, 8)[slot, :, 7]
        cu = ((hw >> 32) & 0xF) * 65536 + (hw & 0xFF00)          # XCC, SE / SH / CU of HW_ID
        ncu = len(set(cu.tolist()))
        overlap = 0
        for key in set(cu.tolist()):
            idx = np.where(cu == key)[0]
            iv = sorted((s0[i], tl[i]) for i in idx)
            overlap += sum(1 for a, b_ in zip(iv, iv[1:]) if b_[0] < a[1])
        per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
        print(f"         CUs used {ncu}, workgroups per CU min {per_cu.min()} max {per_cu.max()}, pairs on one CU that overlap in time: {overlap}; "
              f"starts in the first 5 us: {(s0 - base < 5).sum()}, later than 30 us: {(s0 - base > 30).sum()}")
        print(f"slot {slot}: start spread {s0.max() - base:5.1f} | body mean {(b - s0).mean():7.1f} (min {(b - s0).min():.1f} max {(b - s0).max():.1f}) | "
              f"last body end {b.max() - base:7.1f} | last tail end {tl.max() - base:7.1f} (+{tl.max() - b.max():5.1f})")
    sys.exit(0)
raw = buf.cpu().numpy()
G = int(raw[6])
t = raw[:P * G * 8].reshape(P, G, 8)
P = int((t[:, :, 0] > 0).all(axis=1).sum())          # passes every workgroup ran
t = t[:P].astype(np.float64) / 100.0   # microseconds
t0 = t[0, :, 0].min()
print(f"n = {n}, grid = {G}, passes = {P}; times in us since the first workgroup started")
for p in range(P):
    s, c_, b, tl, go, dec = (t[p, :, k] - t0 for k in range(6))
    dec_wg = int(np.argmax(t[p, :, 5]))
    print(f"pass {p}: start {s.min():8.1f}..{s.max():8.1f} | ctl fetched +{(c_ - s).mean():5.1f} | body {(b - c_).mean():7.1f} (min {(b - c_).min():.1f} max {(b - c_).max():.1f}) "
          f"| last body end {b.max():8.1f} | decider wg {dec_wg}: tail done {tl[dec_wg]:8.1f} (+{tl[dec_wg] - b.max():5.1f} after the last body) "
          f"| barrier passed {go.min():8.1f}..{go.max():8.1f} (+{go.max() - tl[dec_wg]:5.1f})")
