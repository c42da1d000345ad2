"""Test-only stand-in for ``zfista_amd.engine.DeviceSolver`` (no GPU needed).

The O(n) work of one trial is done with the ORACLE's NumPy expressions
(oracle/problems_ref.py); the control logic is the product's own decide step,
run on the host through the C ABI (``zf_decide_host`` - the very function the
decide kernel executes).  Used by the ``-m "not gpu"`` tests to exercise the
host driver (``NativeRun`` / ``_solve_native``), the result assembly and the
world_size-2 gloo exchange.  Never imported by the product.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from oracle import problems_ref as P
from zfista_amd import _lib
from zfista_amd.engine import gather_packs


class FakeProblem:
    """Duck-typed NativeProblem for P-diag: data stays in host NumPy arrays."""

    def __init__(self, d, c, lam, group=None, world=1, rank=0):
        self.d, self.c, self.lam = np.asarray(d, float), np.asarray(c, float), float(lam)
        self.n_features = self.d.size
        self.group, self.world, self.rank = group, world, rank

    def _descriptor(self):
        return dict(kind=_lib.ZF_PROBLEM_DIAG_QUAD_L1, world=self.world, rank=self.rank,
                    n=self.n_features, lam=self.lam), ()


class FakeSolver:
    """``options["sub_iters"] = S > 1`` makes a pass chain S fresh trials the way the temporally
    blocked kernel does (csrc/zf_kernels_step.h): lagging iterations are replayed first (no
    sums), same buffer hand-over, only the last two iterates of a chain are kept."""

    def __init__(self, fields, options, problem, x0):
        self.lib = _lib.load()
        self.p = problem
        self.world, self.group = fields["world"], problem.group
        self.n = fields["n"]
        self.sub_iters = sub = max(1, int(options.get("sub_iters", 0) or 1))
        ring = 4 if sub > 1 else 3
        c = _lib.Control()
        c.lr, c.tol, c.tol_internal = options["lr"], options["tol"], options["tol_internal"]
        c.decay_rate, c.max_iter = options["decay_rate"], options["max_iter"]
        c.max_backtrack = options["max_backtrack_iter"]
        c.status = _lib.ZF_BACKTRACK_FAILED if c.max_backtrack == 0 else _lib.ZF_RUNNING
        c.nesterov, c.deprecated = options["nesterov"], options["deprecated"]
        c.need_grad, c.world, c.cur = 1, self.world, 0
        c.ring_size, c.sub_iters, c.prev = ring, sub, ring - 1
        c.lag, c.pend_status = 0, 0
        self.ctl = c
        self.trace = np.zeros((_lib.ZF_RING, _lib.ZF_TRACE_COLS))
        self.beta = np.zeros(_lib.ZF_RING)
        x0 = np.asarray(x0, float)
        self.xb = [x0.copy() for _ in range(ring)]
        self.passes = 0
        ref = P.DiagQuadL1Ref(problem.d, problem.c, problem.lam)
        packs = self._exchange(np.array([ref.f(x0), ref.g(x0), 0, 0, 0, 0, 0, 0], float), 1)
        f = sum(packs[r * 8 + 0] for r in range(self.world))
        g = sum(packs[r * 8 + 1] for r in range(self.world))
        c.f_x, c.g_x, c.F_old = f, g, f + g

    def _exchange(self, pack_local, npacks):
        if self.world == 1:
            return pack_local.copy()
        import torch

        local = torch.from_numpy(np.ascontiguousarray(pack_local))
        allp = torch.zeros(_lib.ZF_PACK_LEN * npacks * self.world, dtype=torch.float64)
        gather_packs(allp, local, self.group)
        return allp.numpy()

    def set_beta(self, first, betas):
        for k, b in enumerate(betas):
            self.beta[(first + k) % _lib.ZF_RING] = b

    def set_max_iter(self, max_iter):
        c = self.ctl
        c.max_iter = max_iter
        if c.status == _lib.ZF_MAXITER and c.nit < max_iter:
            c.status = _lib.ZF_RUNNING
        if c.pend_status == _lib.ZF_MAXITER and c.nit < max_iter:
            c.pend_status = 0

    def _fresh_len(self):   # csrc/zf_decide.h: zf_fresh_len
        c, S = self.ctl, self.sub_iters
        if c.pend_status != 0:
            return 0
        left = c.max_iter - c.nit
        n = S // 2 if S >= 16 else S
        if c.lag == 0:
            if left >= 2 * S or left == S:
                return S
            n = (left + 1) // 2 if left > S else left
        return max(1, min(n, 2 * S - 1 - c.lag, left))

    def _chain_packs(self):
        """One pass: replay of the lagging iterations, then the fresh trials, from the stored
        (x_{nit-lag}, x_{nit-lag-1}); returns sub x 8 packs (fresh trials only)."""
        c, p = self.ctl, self.p
        ref = P.DiagQuadL1Ref(p.d, p.c, p.lam)
        lag, nf = c.lag, self._fresh_len()
        base = c.nit - lag
        xk, xo = self.xb[c.cur], self.xb[c.prev]
        for i in range(lag):
            beta = self.beta[(base + i) % _lib.ZF_RING] if c.nesterov else 0.0
            y = xk + beta * (xk - xo) if c.nesterov else xk
            xn = ref.prox_wsum_g(c.lag_lr[i], y - c.lag_lr[i] * ref.jac_f(y))
            xo, xk = xk, xn
        packs = np.zeros((self.sub_iters, _lib.ZF_PACK_LEN))
        for j in range(nf):
            lr = c.lr
            beta = self.beta[(c.nit + j) % _lib.ZF_RING] if c.nesterov else 0.0
            y = xk + beta * (xk - xo) if c.nesterov else xk
            grad = ref.jac_f(y)
            xn = ref.prox_wsum_g(lr, y - lr * grad)
            dx = xn - y
            packs[j] = [ref.f(y), grad @ dx, np.sum(dx * dx), ref.g(xn), ref.f(xn),
                        np.max(np.abs(dx)) if dx.size else 0.0, 0.0, 0.0]
            xo, xk = xk, xn
        free = [i for i in range(c.ring_size) if i not in (c.cur, c.prev)]
        if lag + nf == 1:
            self.xb[free[0]] = xk
        elif lag + nf >= 2:
            self.xb[free[0]], self.xb[free[1]] = xo, xk
        return packs.reshape(-1)

    def flush(self):
        c = self.ctl
        if c.status == _lib.ZF_RUNNING and c.lag > 0 and c.pend_status == 0:
            c.pend_status = _lib.ZF_PEND_FLUSH
        self.enqueue(1)

    def enqueue(self, steps):
        for _ in range(steps):
            running = self.ctl.status == _lib.ZF_RUNNING
            pack = self._chain_packs() if running else np.zeros(_lib.ZF_PACK_LEN * self.sub_iters)
            packs = np.ascontiguousarray(self._exchange(pack, self.sub_iters))
            if running:
                self.passes += 1
                _lib.check(self.lib.zf_decide_host(C.byref(self.ctl), C.sizeof(self.ctl), C.c_void_p(_lib.ptr(packs)),
                                                   C.c_void_p(_lib.ptr(self.trace))))

    def poll(self):
        return self.ctl, self.trace

    def get_x(self):
        return self.xb[self.ctl.cur].copy()

    def close(self):
        pass
