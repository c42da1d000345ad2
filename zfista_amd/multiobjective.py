"""Multi-objective trial (m >= 2 objectives): zfista/proximal_gradient.py:159-209.

The m-dimensional dual on the unit simplex is minimised on the host with the
same SciPy calls and options as the reference (`minimize_scalar` for m = 2,
`trust-constr` + BFGS for m >= 3; :179-205).  Each dual evaluation is ONE fused
HIP kernel over (J, y) that writes nothing and returns 2m+2 scalars
(``zf_mo_dual_eval``); primal recovery, F evaluations, the Jacobians of the
built-in problems and the momentum update are HIP kernels too
(``csrc/zf_multiobj.hip``).  x_k, x_{k-1}, y, x+ and J stay resident in HBM.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from warnings import warn

import numpy as np
from scipy.optimize import BFGS, Bounds, LinearConstraint, OptimizeResult, minimize, minimize_scalar

from . import _lib
from .engine import momentum_factors

X_K, Y, X_NEW, X_OLD = 0, 1, 2, 3


def combine_totals(vals, max_index, group):
    """C3: the one exchange of a sharded multi-objective reduction.  Every rank's raw
    totals are all-gathered and combined in RANK ORDER (sums added rank 0, 1, ...; entry
    ``max_index`` a maximum), so all ranks continue with bitwise-identical scalars and the host
    dual solver takes identical steps everywhere.  ``group``: a ``torch.distributed`` process
    group (gloo: CPU tensors; nccl = RCCL: device tensors) or any object with
    ``all_gather_host(np.ndarray) -> [np.ndarray per rank]`` (tests)."""
    vals = np.asarray(vals, dtype=np.float64)
    if hasattr(group, "all_gather_host"):
        parts = group.all_gather_host(vals.copy())
    elif hasattr(group, "handle") and hasattr(group, "all_gather"):   # zfista_amd.comm.LibComm (RCCL in the library)
        import torch

        mine = torch.from_numpy(vals.copy()).cuda()
        out = torch.empty(group.world * vals.size, dtype=torch.float64, device=mine.device)
        group.all_gather(mine, out)
        parts = list(out.cpu().numpy().reshape(group.world, vals.size))
    else:
        import torch
        import torch.distributed as dist

        world = dist.get_world_size(group)
        on_gpu = dist.get_backend(group) == "nccl"
        mine = torch.from_numpy(vals.copy())
        if on_gpu:
            mine = mine.cuda()
        out = torch.empty(world * vals.size, dtype=torch.float64, device=mine.device)
        dist.all_gather_into_tensor(out, mine, group=group)
        parts = list(out.cpu().numpy().reshape(world, vals.size))
    total = np.array(parts[0], dtype=np.float64)
    for p in parts[1:]:
        for k in range(total.size):
            total[k] = max(total[k], p[k]) if k == max_index else total[k] + p[k]
    return total


_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.c_int32)


class MoEngine:
    """ctypes wrapper of one ``zf_mo`` object.  ``group`` / ``n_global`` / ``offset``: x is this
    rank's contiguous block of a decision vector sharded over the ranks of ``group``."""

    def __init__(self, kind, m, n, l1_ratios=None, l1_shifts=None, bounds=None, group=None, n_global=None,
                 offset=0):
        self.lib = _lib.require_gpu()
        self.kind, self.m, self.n = kind, int(m), int(n)
        self.group = group
        ratios = None if l1_ratios is None else np.ascontiguousarray(l1_ratios, dtype=np.float64)
        shifts = np.zeros(m) if l1_shifts is None else np.ascontiguousarray(l1_shifts, dtype=np.float64)
        if ratios is not None and ratios.size != m:
            raise ValueError("len(l1_ratios) should be equal to n_objectives.")   # problems.py:108-109
        if shifts.size != m:
            raise ValueError("len(l1_shifts) should be equal to n_objectives.")   # problems.py:110-111
        lo, hi = (-np.inf, np.inf) if bounds is None else (float(bounds[0]), float(bounds[1]))
        try:
            import torch

            stream = torch.cuda.current_stream().cuda_stream
        except Exception:
            stream = None
        h = C.c_void_p()
        _lib.check(self.lib.zf_mo_create(
            C.byref(h), kind, self.m, self.n,
            C.c_void_p(_lib.ptr(ratios)) if ratios is not None else None,
            C.c_void_p(_lib.ptr(shifts)), lo, hi, C.c_void_p(stream)), "zf_mo_create")
        self.h = h
        self.n_dual_evals = 0
        self.n_exchanges = 0
        self.device_search_lost = False   # a device trial gave up on its grid-wide wait: host loop from then on
        self._cb = None
        self._exchange_error = None
        from .comm import LibComm

        self.libcomm = False   # exchanges through the library's communicator (zf_mo_set_comm)
        if isinstance(group, LibComm) and os.environ.get("ZF_MO_COMM", "lib") == "lib":
            # the library's own communicator: every reduction exchanges on the stream (all-gather + rank-ordered
            # sum on the device), zf_mo_solve_dual once per batch of its search - no Python per exchange
            _lib.check(self.lib.zf_mo_set_comm(h, group.handle, int(n_global), int(offset)), "zf_mo_set_comm")
            self.libcomm = True
        elif group is not None:
            def exchange(_ctx, vals, count, max_index):
                try:
                    arr = np.ctypeslib.as_array(vals, shape=(count,))
                    arr[:] = combine_totals(arr, int(max_index), group)
                    self.n_exchanges += 1
                    return 0
                except Exception as exc:   # never let an exception cross the C boundary
                    self._exchange_error = exc
                    return 1

            self._cb = _EXCHANGE_FN(exchange)   # keep alive as long as the engine
            _lib.check(self.lib.zf_mo_set_shard(h, int(n_global), int(offset), self._cb, None), "zf_mo_set_shard")

    def _check(self, rc, what):
        """_lib.check, but an exception raised inside the exchange callback (it cannot cross the
        C boundary) is re-raised here as the cause."""
        if rc != _lib.ZF_OK and self._exchange_error is not None:
            exc, self._exchange_error = self._exchange_error, None
            raise _lib.ZfError(f"{what}: the exchange between ranks failed") from exc
        _lib.check(rc, what)

    def exchange_count(self):
        """Collectives issued so far: by the library over its communicator, or through the Python callback."""
        c = C.c_int64(0)
        _lib.check(self.lib.zf_mo_exchange_count(self.h, C.byref(c)), "zf_mo_exchange_count")
        return int(c.value) + self.n_exchanges

    def set_bounds(self, lo, hi):
        lo = np.ascontiguousarray(lo, dtype=np.float64)
        hi = np.ascontiguousarray(hi, dtype=np.float64)
        if lo.size != self.n or hi.size != self.n:
            raise ValueError("bounds arrays must have n_features entries")
        _lib.check(self.lib.zf_mo_set_bounds(self.h, C.c_void_p(_lib.ptr(lo)), C.c_void_p(_lib.ptr(hi))),
                   "zf_mo_set_bounds")

    def _vec(self, a):
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
        if a.size != self.n:
            raise ValueError(f"len(x) should be equal to n_features, got {a}.")
        return a

    def set_x0(self, x0):
        x0 = self._vec(x0)
        _lib.check(self.lib.zf_mo_set_x0(self.h, C.c_void_p(_lib.ptr(x0))), "zf_mo_set_x0")

    def eval_F(self, which, builtin_f=True):
        f = np.zeros(self.m)
        g = np.zeros(self.m)
        self._check(self.lib.zf_mo_eval_F(self.h, which, C.c_void_p(_lib.ptr(f)) if builtin_f else None,
                                         C.c_void_p(_lib.ptr(g))), "zf_mo_eval_F")
        return (f if builtin_f else None), g

    def prepare(self):
        f_y = np.zeros(self.m)
        self._check(self.lib.zf_mo_prepare(self.h, C.c_void_p(_lib.ptr(f_y))), "zf_mo_prepare")
        return f_y

    def prepare_async(self):
        """prepare() without the host round trip: f(y) stays on the device for solve_dual_device."""
        self._check(self.lib.zf_mo_prepare_async(self.h), "zf_mo_prepare_async")

    def set_fused(self, on):
        """Fused outer iteration (zf_mo_set_fused): commit() and prepare_async() are deferred into the
        next solve_dual_device(), whose one kernel forms y, f(y) and J itself."""
        self._check(self.lib.zf_mo_set_fused(self.h, int(bool(on))), "zf_mo_set_fused")

    def get_f_y(self):
        f_y = np.zeros(self.m)
        self._check(self.lib.zf_mo_get_f_y(self.h, C.c_void_p(_lib.ptr(f_y))), "zf_mo_get_f_y")
        return f_y

    def set_jac(self, J):
        J = np.ascontiguousarray(np.asarray(J, dtype=np.float64))
        if J.shape != (self.m, self.n):
            raise ValueError(f"jac_f must return shape ({self.m}, {self.n}), got {J.shape}")
        _lib.check(self.lib.zf_mo_set_jac(self.h, C.c_void_p(_lib.ptr(J))), "zf_mo_set_jac")

    def dual_eval(self, lr, w):
        # (called 10 - 1e5 times per outer iteration: buffers and their ctypes pointers are kept)
        bufs = self.__dict__.get("_dual_bufs")
        if bufs is None:
            wb, ob = np.zeros(self.m), np.zeros(2 * self.m + 2)
            bufs = self._dual_bufs = (wb, ob, C.c_void_p(_lib.ptr(wb)), C.c_void_p(_lib.ptr(ob)))
        wb, ob, wp, op_ = bufs
        wb[:] = w
        rc = self.lib.zf_mo_dual_eval(self.h, float(lr), wp, op_)
        if rc != _lib.ZF_OK:
            self._check(rc, "zf_mo_dual_eval")
        self.n_dual_evals += 1
        m = self.m
        out = ob.copy()
        return out[:m], out[m], out[m + 1], out[m + 2:]

    def dual_hessian(self, lr, w):
        """The generalised Hessian of the dual at w (m x m), as the device-side search uses it."""
        w = np.ascontiguousarray(w, dtype=np.float64)
        H = np.zeros((self.m, self.m))
        self._check(self.lib.zf_mo_dual_hessian(self.h, float(lr), C.c_void_p(_lib.ptr(w)), C.c_void_p(_lib.ptr(H))),
                    "zf_mo_dual_hessian")
        return H

    def solve_dual(self, lr, f_y, F_old, deprecated, w0, tol, max_iter):
        """The whole dual search of a trial inside the library (zf_mo_solve_dual; opt-in,
        ZF_DUAL_SOLVER=native).  Returns (weight, fun, nit) or None when it was not attempted
        (non-finite start: the caller uses the reference's SciPy calls)."""
        f_y = np.ascontiguousarray(f_y, dtype=np.float64)
        F_old = np.ascontiguousarray(F_old, dtype=np.float64)
        w0 = None if w0 is None else np.ascontiguousarray(w0, dtype=np.float64)
        w = np.zeros(self.m)
        fun, nit, ok, evals = C.c_double(0.0), C.c_int64(0), C.c_int32(0), C.c_int64(0)
        rc = self.lib.zf_mo_solve_dual(self.h, float(lr), C.c_void_p(_lib.ptr(f_y)), C.c_void_p(_lib.ptr(F_old)),
                                       int(bool(deprecated)), None if w0 is None else C.c_void_p(_lib.ptr(w0)),
                                       float(tol), int(max_iter), C.c_void_p(_lib.ptr(w)), C.byref(fun),
                                       C.byref(nit), C.byref(ok), C.byref(evals))
        self.n_dual_evals += int(evals.value)
        self._check(rc, "zf_mo_solve_dual")
        if not ok.value:
            return None
        return w, np.float64(fun.value), int(nit.value)

    def solve_dual_stream(self, lr, f_y, F_old, deprecated, w0, tol, max_iter):
        """The dual search of a trial for an x sharded over a library communicator, driven from the device
        (zf_mo_solve_dual_stream): the state machine lives in device memory, a batch is evaluation -> reduce -> ONE
        all-gather -> a one-wave kernel that advances the machine; no host synchronisation per batch.  Returns
        (weight, fun, nit) or None when it was not attempted (non-finite start)."""
        f_y = np.ascontiguousarray(f_y, dtype=np.float64)
        F_old = np.ascontiguousarray(F_old, dtype=np.float64)
        w0 = None if w0 is None else np.ascontiguousarray(w0, dtype=np.float64)
        w = np.zeros(self.m)
        fun, nit, ok, evals = C.c_double(0.0), C.c_int64(0), C.c_int32(0), C.c_int64(0)
        rc = self.lib.zf_mo_solve_dual_stream(self.h, float(lr), C.c_void_p(_lib.ptr(f_y)), C.c_void_p(_lib.ptr(F_old)),
                                              int(bool(deprecated)), None if w0 is None else C.c_void_p(_lib.ptr(w0)),
                                              float(tol), int(max_iter), C.c_void_p(_lib.ptr(w)), C.byref(fun),
                                              C.byref(nit), C.byref(ok), C.byref(evals))
        self.n_dual_evals += int(evals.value)
        self._check(rc, "zf_mo_solve_dual_stream")
        if not ok.value:
            return None
        return w, np.float64(fun.value), int(nit.value)

    def solve_dual_device(self, lr, f_y, F_old, deprecated, w0, tol, max_iter):
        """The dual search AND the primal recovery of a trial in one persistent kernel
        (zf_mo_solve_dual_device, dual_solver="device").  ``f_y`` None: take f(y) from the device
        (after prepare_async()).  Returns (weight, fun, nit, err, f_x, g_x, f_y)
        with x+ left in its buffer (f_x is None when f is a host callback), or None when it was not attempted (non-finite start, sharded x,
        m > 8): the caller continues with solve_dual() / the reference's calls and recover()."""
        # (once per trial on the solver's critical path: buffers and their ctypes pointers are kept)
        st = self.__dict__.get("_solve_bufs")
        if st is None:
            arrs = [np.zeros(self.m) for _ in range(7)]   # f_y, F_old, w0, w, f_x, g_x, f_y_used
            scal = (C.c_double(0.0), C.c_int64(0), C.c_int32(0), C.c_int64(0), C.c_double(0.0))
            st = self._solve_bufs = (arrs, [C.c_void_p(_lib.ptr(a)) for a in arrs], scal, [C.byref(v) for v in scal])
        (b_fy, b_Fold, b_w0, w, f_x, g_x, f_y_used), ptrs, (fun, nit, ok, evals, err), refs = st
        if self.device_search_lost:
            return None
        if f_y is not None:
            b_fy[:] = f_y
        b_Fold[:] = F_old
        if w0 is not None:
            b_w0[:] = w0
        rc = self.lib.zf_mo_solve_dual_device(self.h, float(lr), None if f_y is None else ptrs[0], ptrs[1],
                                              int(bool(deprecated)), None if w0 is None else ptrs[2],
                                              float(tol), int(max_iter), ptrs[3], refs[0], refs[1], refs[2], refs[3],
                                              refs[4], ptrs[4], ptrs[5], ptrs[6])
        self.n_dual_evals += int(evals.value)
        if rc != _lib.ZF_OK:
            self._check(rc, "zf_mo_solve_dual_device")
        if ok.value < 0:
            self.device_timed_out()
            return None
        if not ok.value:
            return None
        return (w.copy(), np.float64(fun.value), int(nit.value), np.float64(err.value),
                None if np.isnan(f_x[0]) else f_x.copy(), g_x.copy(), f_y_used.copy())

    def device_timed_out(self, ticket=None):
        """A device trial gave up on a grid-wide wait (its workgroups were not all resident: another kernel,
        process or a CU mask holds part of the GPU).  Nothing of that trial is used; f(y), J are formed again
        the unfused way, and this engine keeps to the host loop from now on (a second attempt would wait out
        the same seconds).  Call after zf_mo_uncommit when a trial was launched ahead."""
        self.device_search_lost = True
        if ticket is None:   # (solve_dual_device: the launch just made)
            ticket = -1
        _lib.check(self.lib.zf_mo_invalidate_prepare(self.h, int(ticket)), "zf_mo_invalidate_prepare")
        warn("the device-side dual search gave up waiting for its grid (is another kernel occupying the GPU?); "
             "continuing with the host-driven search", UserWarning, stacklevel=3)

    def debug_force_timeout(self, launches=1):
        _lib.check(self.lib.zf_mo_debug_force_timeout(self.h, int(launches)), "zf_mo_debug_force_timeout")

    # -- trials launched ahead of their predecessor's result (zf_mo_trial_launch / _wait) -------------
    def trial_launch(self, lr, F_old, deprecated, w0, tol, max_iter, accept_tol, decay_is_one, gated):
        """Enqueue one trial (after prepare_async()); returns a ticket, or None when this problem has no
        device trial (sharded x, m > 8).  ``gated``: the trial runs only if the one launched before it
        turns out accepted, and takes F(x_k) from that trial's F(x+) on the device (``F_old`` None)."""
        st = self.__dict__.get("_launch_bufs")
        if st is None:
            arrs = [np.zeros(self.m) for _ in range(2)]   # F_old, w0
            st = self._launch_bufs = (arrs, [C.c_void_p(_lib.ptr(a)) for a in arrs], C.c_int32(0))
        (b_Fold, b_w0), ptrs, ticket = st
        if self.device_search_lost:
            return None
        if F_old is not None:
            b_Fold[:] = F_old
        if w0 is not None:
            b_w0[:] = w0
        rc = self.lib.zf_mo_trial_launch(self.h, float(lr), None if F_old is None else ptrs[0], int(bool(deprecated)),
                                         None if w0 is None else ptrs[1], float(tol), int(max_iter), float(accept_tol),
                                         int(bool(decay_is_one)), int(gated), C.byref(ticket))
        if rc != _lib.ZF_OK:
            self._check(rc, "zf_mo_trial_launch")
        return None if ticket.value < 0 else int(ticket.value)

    def trial_wait(self, ticket):
        """Result of a launched trial: None if it was skipped (its gate was closed), else
        (weight, fun, nit, err, f_x, g_x, f_y, accepted) - or ("not attempted", f_y) when the search did not
        run (non-finite start), ("timed out", None) when a grid-wide wait gave up: the caller continues with the
        host path (after device_timed_out())."""
        st = self.__dict__.get("_wait_bufs")
        if st is None:
            arrs = [np.zeros(self.m) for _ in range(4)]   # w, f_x, g_x, f_y
            scal = (C.c_double(0.0), C.c_int64(0), C.c_int32(0), C.c_int64(0), C.c_double(0.0), C.c_int32(0), C.c_int32(0))
            st = self._wait_bufs = (arrs, [C.c_void_p(_lib.ptr(a)) for a in arrs], scal, [C.byref(v) for v in scal])
        (w, f_x, g_x, f_y), ptrs, (fun, nit, ok, evals, err, accepted, skipped), refs = st
        rc = self.lib.zf_mo_trial_wait(self.h, int(ticket), ptrs[0], refs[0], refs[1], refs[2], refs[3], refs[4],
                                       ptrs[1], ptrs[2], ptrs[3], refs[5], refs[6])
        if rc != _lib.ZF_OK:
            self._check(rc, "zf_mo_trial_wait")
        if skipped.value:
            return None
        if ok.value < 0:
            return ("timed out", None)
        self.n_dual_evals += int(evals.value)
        if not ok.value:
            return ("not attempted", f_y.copy())
        return (w.copy(), np.float64(fun.value), int(nit.value), np.float64(err.value), f_x.copy(), g_x.copy(),
                f_y.copy(), bool(accepted.value))

    def uncommit(self):
        _lib.check(self.lib.zf_mo_uncommit(self.h), "zf_mo_uncommit")

    def solve_stats(self):
        """Diagnostics of the last solve_dual_device(): batches, evaluations and the shader-clock
        cycles workgroup 0 spent in total / evaluating / in grid-wide hand-overs / advancing the solver."""
        out = np.zeros(6, dtype=np.int64)
        _lib.check(self.lib.zf_mo_solve_stats(self.h, C.c_void_p(_lib.ptr(out)), out.size), "zf_mo_solve_stats")
        return dict(zip(("batches", "evals", "cyc_total", "cyc_eval", "cyc_combine", "cyc_step"), map(int, out)))

    def recover(self, lr, w):
        w = np.ascontiguousarray(w, dtype=np.float64)
        err = C.c_double(0.0)
        self._check(self.lib.zf_mo_recover(self.h, float(lr), C.c_void_p(_lib.ptr(w)), C.byref(err)), "zf_mo_recover")
        return np.float64(err.value)

    def commit(self, beta, nesterov):
        _lib.check(self.lib.zf_mo_commit(self.h, float(beta), int(bool(nesterov))), "zf_mo_commit")

    def get(self, which):
        out = np.empty(self.n)
        _lib.check(self.lib.zf_mo_get(self.h, which, C.c_void_p(_lib.ptr(out)), out.size), "zf_mo_get")
        return out

    def put(self, which, x):
        x = self._vec(x)
        _lib.check(self.lib.zf_mo_put(self.h, which, C.c_void_p(_lib.ptr(x))), "zf_mo_put")

    def get_jac(self):
        out = np.empty((self.m, self.n))
        _lib.check(self.lib.zf_mo_get_jac(self.h, C.c_void_p(_lib.ptr(out)), out.size), "zf_mo_get_jac")
        return out

    def prox_host(self, weight, x):
        weight = np.ascontiguousarray(weight, dtype=np.float64)
        if weight.size != self.m:
            raise ValueError("len(weight) should be equal to n_objectives.")   # problems.py:124-125
        x = self._vec(x)
        out = np.empty(self.n)
        _lib.check(self.lib.zf_mo_prox_host(self.h, C.c_void_p(_lib.ptr(weight)), C.c_void_p(_lib.ptr(x)),
                                            C.c_void_p(_lib.ptr(out))), "zf_mo_prox_host")
        return out

    def post_terms(self, lr, w, p):
        w = np.ascontiguousarray(w, dtype=np.float64)
        p = self._vec(p)
        out = np.zeros(self.m + 1)
        self._check(self.lib.zf_mo_post_terms(self.h, float(lr), C.c_void_p(_lib.ptr(w)), C.c_void_p(_lib.ptr(p)),
                                             C.c_void_p(_lib.ptr(out))), "zf_mo_post_terms")
        return out[:self.m], out[self.m]

    def close(self):
        if getattr(self, "h", None):
            self.lib.zf_mo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve_dual(dual, m, w0, tol, max_iter, solver="scipy"):
    """The reference's two SciPy calls (:179-205), or - ``solver`` "native" / "device" - the
    library's own search driven from Python (opaque callbacks).  Returns (weight, fun, nit)."""
    if solver in ("native", "device"):
        out = solve_dual_native(dual, m, w0, tol, max_iter)
        if out is not None:
            return out
    if m == 2:
        sol = minimize_scalar(lambda s: dual(np.array([s, 1 - s]))[0], bounds=(0, 1),
                              options={"maxiter": max_iter, "xatol": tol})
        if not sol.success:
            warn(sol.message, stacklevel=2)
        return np.array([sol.x, 1 - sol.x]), sol.fun, sol.nit
    sol = minimize(fun=dual, x0=w0, method="trust-constr", jac=True, hess=BFGS(),
                   bounds=Bounds(lb=0, ub=np.inf), constraints=LinearConstraint(np.ones(m), lb=1, ub=1),
                   options={"gtol": tol, "xtol": tol, "barrier_tol": tol, "maxiter": max_iter})
    if not sol.success:
        warn(sol.message, stacklevel=2)
    return sol.x, sol.fun, sol.nit


# ---------------------------------------------------------------------------
# native dual solver: opt-in replacement of the two SciPy calls
# ---------------------------------------------------------------------------
# selected per call: minimize_proximal_gradient(..., dual_solver="scipy" | "native" | "device");
# "scipy" = the reference's calls (parity default)


def _simplex_qp(q, Q):
    """argmin_{w in simplex} q.w + 1/2 w'Qw for tiny m by enumerating supports (KKT check)."""
    m = q.size
    best, best_val = None, np.inf
    for mask in range(1, 1 << m):
        S = [i for i in range(m) if mask >> i & 1]
        k = len(S)
        K = np.zeros((k + 1, k + 1))
        K[:k, :k] = Q[np.ix_(S, S)]
        K[:k, k] = 1.0
        K[k, :k] = 1.0
        rhs = np.concatenate([-q[S], [1.0]])
        try:
            sol = np.linalg.solve(K, rhs)
        except np.linalg.LinAlgError:
            sol = np.linalg.lstsq(K, rhs, rcond=None)[0]
        wS, mu = sol[:k], sol[k]
        if np.any(wS < -1e-14):
            continue
        w = np.zeros(m)
        w[S] = np.maximum(wS, 0.0)
        w /= w.sum()
        red = q + Q @ w + mu          # reduced costs; must be >= 0 off the support
        if np.any(red[[i for i in range(m) if i not in S]] < -1e-10 * (1 + np.abs(red).max())):
            continue
        val = q @ w + 0.5 * w @ Q @ w
        if val < best_val:
            best, best_val = w, val
    if best is None:                  # numerically degenerate: fall back to the best vertex
        v = np.argmin(q + 0.5 * np.diag(Q))
        best = np.zeros(m)
        best[v] = 1.0
    return best


def _solve_dual_1d(dual, tol, max_iter):
    """m = 2: the dual restricted to w = (s, 1 - s) is convex C^1 on [0, 1]; its derivative
    phi(s) = dD/dw_0 - dD/dw_1 is monotone and piecewise linear, so a bracketing root finder
    (Illinois-modified regula falsi, exact on a linear piece) places the minimiser to `tol`
    in s - something a comparison-based search on D itself (minimize_scalar) cannot do below
    sqrt(eps)."""

    def phi(s):
        f, g = dual(np.array([s, 1.0 - s]))
        return f, g[0] - g[1]

    fa, pa = phi(0.0)
    if pa >= 0.0:
        return np.array([0.0, 1.0]), fa, 1
    fb, pb = phi(1.0)
    if pb <= 0.0:
        return np.array([1.0, 0.0]), fb, 2
    a, b = 0.0, 1.0
    side = 0
    s, fs = 0.5, None
    nit = 2
    for nit in range(3, int(max_iter) + 3):
        s = (a * pb - b * pa) / (pb - pa)          # secant point of the bracket
        if not (a < s < b):
            s = 0.5 * (a + b)
        fs, ps = phi(s)
        if ps == 0.0 or (b - a) <= tol:
            break
        if ps < 0.0:
            a, pa = s, ps
            if side == -1:
                pb *= 0.5                            # Illinois: halve the stale end
            side = -1
        else:
            b, pb = s, ps
            if side == 1:
                pa *= 0.5
            side = 1
        if (b - a) <= tol:
            s = 0.5 * (a + b)
            fs, _ = phi(s)
            break
    return np.array([s, 1.0 - s]), fs, nit


def solve_dual_native(dual, m, w0, tol, max_iter):
    """Minimise the convex C^1 (piecewise quadratic for the l1 + box family) dual over the
    unit simplex by a projected Newton method: gradient from one fused evaluation,
    curvature from m further gradient evaluations along the feasible directions e_i - w,
    the m-variable QP on the simplex solved exactly, Armijo backtracking.  Typically
    (m + 2) evaluations per iteration and 3-8 iterations, against 30 - 1e5 evaluations of
    trust-constr at the same tolerance.  Returns (weight, fun, n_iterations)."""
    w = np.ones(m) / m if w0 is None else np.clip(np.asarray(w0, float), 0, None)
    w = w / w.sum()
    fun, grad = dual(w)
    if not (np.isfinite(fun) and np.all(np.isfinite(grad))):
        return None   # e.g. F(x_k) = inf (x_k outside the box): leave it to the reference's calls
    if m == 2:
        return _solve_dual_1d(dual, tol, max_iter)
    nit = 0
    h = 1e-5
    for nit in range(1, int(max_iter) + 1):
        # curvature on the tangent space: (grad(w + h (e_i - w)) - grad(w)) / h = H (e_i - w)
        T = np.eye(m) - w[:, None]                  # column i = e_i - w
        HT = np.empty((m, m))
        for i in range(m):
            _, gi = dual(w + h * T[:, i])
            HT[:, i] = (gi - grad) / h
        Q = T.T @ HT
        Q = 0.5 * (Q + Q.T)
        ev = np.linalg.eigvalsh(Q)
        if ev.min() < 0:                            # keep the model convex against FD noise
            Q = Q + (1e-12 - ev.min()) * np.eye(m)
        # model in w' (sum w' = 1, w' - w = T w'):  grad.T w' + 1/2 w'Qw'
        # (scaled to O(1) entries: the KKT systems couple Q with the constraint row of ones - left at |Q| ~ 1e5 they
        #  are conditioned like |Q|^2 and the Newton point comes out with an absolute error of 1e-9 in w)
        scale = max(np.max(np.abs(Q)), 1e-300)
        # The model is q . d + 1/2 d'Qd in the STEP d = w' - w.  Written in w' it reads (q - Q w) . w' + 1/2 w'Qw';
        # analytically Q w = T'H(T w) = 0, but the probed Q satisfies that only to ~1e-8 |Q|, and leaving the term
        # out shifted every Newton point by ~1e-9: the search stalled at a gradient residual of 1e-4 where one
        # exact tangent-space Newton step reaches 3e-11 (measured on G11's quad3).
        w_new = _simplex_qp((T.T @ grad - Q @ w) / scale, Q / scale)
        d = w_new - w
        step = np.max(np.abs(d))
        slope = (grad - grad.mean()) @ d    # (mean-free: d sums to zero only up to rounding, the common part of grad is 1e5)
        # stop at `tol` in w, or when the predicted decrease grad . d is below what the gradient resolves
        # (its components carry ~4e-16 (|D| + |grad|) of rounding noise: zf_dual::machine::newton_model)
        if step <= tol or slope >= -4e-16 * max(1.0, abs(fun), np.max(np.abs(grad))) * step:
            break
        t = 1.0
        while True:
            f_try, g_try = dual(w + t * d)
            if f_try <= fun + 1e-4 * t * slope + 1e-15 * abs(fun) or t < 1e-10:
                break
            # where the values no longer resolve the decrease (~1e-12 |D| of rounding noise near the optimum) the
            # derivative along d decides: the approximate Wolfe conditions of Hager & Zhang (zf_dual::machine::ls_accept)
            dphi = (g_try - g_try.mean()) @ d
            # (the value guard: within the noise of the values the slopes decide; a rise the convexity of the dual allows -
            #  df <= 2 t max(phi'(t), 0) - is an overshoot and is rejected; a rise beyond that CONTRADICTS the gradients -
            #  the reference's composed prox is not the exact prox of several shifted l1 terms, so its dual value and
            #  gradient are not consistent there - and the slopes decide, as they do for SciPy: zf_dual::machine::ls_accept)
            df, noise = f_try - fun, 1e-10 * abs(fun)
            value_ok = df <= noise or df > 2.0 * t * max(dphi, 0.0) + noise
            if value_ok and 0.9 * slope <= dphi <= -(1.0 - 2e-4) * slope:
                break
            t *= 0.5
        w_prev = w
        w = w + t * d
        w = np.clip(w, 0.0, None)
        w /= w.sum()
        if t < 1.0 or not np.array_equal(w, w_prev + t * d):
            fun, grad = dual(w)
        else:
            fun, grad = f_try, g_try
        if t * step <= tol:
            break
        # finite-difference step follows the Newton step so the curvature is that of the
        # quadratic piece the iterate sits in (the dual is piecewise quadratic)
        h = float(np.clip(0.1 * t * step, 1e-7, 1e-5))
    return w, fun, nit


def device_dual(eng, lr, f_y, F_old, deprecated):
    """(:161-177) with every O(n) term taken from one zf_mo_dual_eval launch."""

    def dual(w):
        g_p, ss_pv, ss_wJ, dots = eng.dual_eval(lr, w)
        fun = -np.inner(w, g_p) - np.sqrt(ss_pv) ** 2 / 2 / lr + lr / 2 * np.sqrt(ss_wJ) ** 2
        jac = -g_p - dots
        if not deprecated:
            fun += np.inner(w, F_old - f_y)
            jac += F_old - f_y
        return fun, jac

    return dual


# ---------------------------------------------------------------------------
# native path: JOS1 / FDS bound methods -> everything but the dual search on the GPU
# ---------------------------------------------------------------------------
def solve_native(problem, x0, o):
    """Outer loop (:463-555) for a recognised multi-objective Problem."""
    from .proximal_gradient import (_MSG_BACKTRACK, _MSG_MAXITER, _MSG_OK, _print_header, _print_row)

    t0 = time.time()
    eng = problem._engine()
    eng.set_x0(x0)
    m = problem.n_objectives
    res = OptimizeResult(x0=x0, tol=o["tol"], tol_internal=o["tol_internal"],
                         nesterov=o["nesterov"], nesterov_ratio=o["nesterov_ratio"])
    if o["verbose"]:
        _print_header()
    dual_solver = o.get("dual_solver", "scipy")
    use_native = dual_solver in ("native", "device")
    host_f = bool(getattr(problem, "_host_f", False))   # f / jac_f are host NumPy (n <= 30 families)

    def eval_F(which):
        if not host_f:
            return eng.eval_F(which)
        _, g_val = eng.eval_F(which, builtin_f=False)
        return np.asarray(problem.f(eng.get(which)), dtype=np.float64), g_val

    # dual_solver="device" on a built-in problem: f(y), J are enqueued without a host round trip and
    # f(y) comes back with the result of the trial's one kernel
    lazy_f_y = dual_solver == "device" and not host_f and eng.group is None and m <= 3
    # ... and in the same kernel: the momentum update of the previous iteration, f(y) and J (one launch
    # and one read-back per trial; zf_mo_set_fused)
    eng.set_fused(lazy_f_y)

    def prepare():
        if lazy_f_y:
            eng.prepare_async()
            return None
        if not host_f:
            return eng.prepare()
        y = eng.get(Y)
        eng.set_jac(problem.jac_f(y))
        return np.asarray(problem.f(y), dtype=np.float64)

    f0, g0 = eval_F(X_K)
    F_old = f0 + g0                     # F(x_k); cached between iterations instead of recomputed (:279)
    lr = o["lr"]
    t_state = None
    betas = []
    nit_done = trials_done = 0
    w_last = None
    # which search produced the weights of how many trials (reported with the result when another search than the
    # reference's was asked for: a requested "device" that was not attempted shows up as "native" / "scipy" counts)
    searches = {"device": 0, "native": 0, "scipy": 0}
    if (lazy_f_y and not o["return_all"] and o["max_iter"] >= 1
            and os.environ.get("ZF_MO_LAUNCH_AHEAD", "1") != "0"):
        out = _solve_native_ahead(eng, o, m, F_old, res, t0, searches)
        if isinstance(out, _HandOver):
            # a device trial gave up in the middle of the solve: the loop below continues from the state it
            # left - x_k, x_{k-1}, y in the engine, the line search of iteration nit_done + 1 under way
            nit_done, trials_done, lr, F_old, t_state, betas, w_last = out
        elif out is not None:
            if dual_solver != "scipy":
                out[0]["dual_search_trials"] = dict(searches)
            return out
        else:
            eng.set_x0(x0)              # (no device trial for this problem: the loop below, from the start)
            eng.set_fused(lazy_f_y)
    w0 = np.ones(m) / m
    if o["warm_start"] and w_last is not None:
        # handed over in the middle of a solve: the reference goes on from the weights of the last trial (:286-288) -
        # the one just rejected inside this line search, else the one the previous iteration was accepted with
        w0 = np.asarray(w_last, dtype=np.float64)
    allvecs = allfuns = allerrs = None
    if o["return_all"]:
        allvecs, allfuns, allerrs = [x0], [f0 + g0], []
    status = _lib.ZF_MAXITER
    nit = nit_done
    for nit in range(nit_done + 1, o["max_iter"] + 1):
        try:
            f_y = prepare()             # f(y_k), J = jac_f(y_k): once per line search (y_k is fixed)
            accepted = False
            first, trials_done = trials_done, 0
            for _ in range(first, o["max_backtrack_iter"]):
                out = err = F_dev = None
                if dual_solver == "device":   # search + recovery + F(x+) in one persistent kernel
                    out = eng.solve_dual_device(lr, f_y, F_old, o["deprecated"], w0, o["tol_internal"],
                                                o["max_iter_internal"])
                    if out is not None:
                        out, err, F_dev, f_y = out[:3], out[3], out[4:6], out[6]
                        searches["device"] += 1
                if f_y is None:   # the device search was not attempted (non-finite start): fetch f(y)
                    f_y = eng.get_f_y()
                if out is None and dual_solver == "device" and eng.libcomm:
                    # x sharded over a library communicator: the same search driven from the device, one collective
                    # per batch and no host round trip per batch (the persistent kernel is single-rank)
                    out = eng.solve_dual_stream(lr, f_y, F_old, o["deprecated"], w0, o["tol_internal"],
                                                o["max_iter_internal"])
                    if out is not None:
                        searches["device"] += 1
                if out is None and use_native:   # the library's own dual solver: no Python between the evaluations
                    out = eng.solve_dual(lr, f_y, F_old, o["deprecated"], w0, o["tol_internal"],
                                         o["max_iter_internal"])
                    if out is not None:
                        searches["native"] += 1
                if out is None:
                    dual = device_dual(eng, lr, f_y, F_old, o["deprecated"])
                    out = solve_dual(dual, m, w0, o["tol_internal"], o["max_iter_internal"], dual_solver)
                    searches["native" if dual_solver in ("native", "device") else "scipy"] += 1
                weight, dual_fun, nit_int = out
                if err is None:
                    err = eng.recover(lr, weight)      # x+ and max|x+ - y|   (:206, :510)
                fun = -dual_fun                         # (:207)
                if F_dev is not None and F_dev[0] is not None and not host_f:
                    f_x, g_x = F_dev                    # sums taken by the same kernel
                else:
                    f_x, g_x = eval_F(X_NEW)
                F_new = f_x + g_x                       # (:295)
                if o["warm_start"]:
                    w0 = weight
                if o["decay_rate"] == 1:
                    accepted = True
                elif o["deprecated"]:
                    accepted = bool(np.all(f_x - f_y <= fun + o["tol_internal"]))
                else:
                    accepted = bool(np.all(F_new - F_old <= fun + o["tol_internal"]))
                if accepted:
                    break
                lr *= o["decay_rate"]
            if not accepted:
                raise RuntimeError(_MSG_BACKTRACK)
        except Exception as exc:   # :493-509
            print(f"An error occurred: {exc}")
            bad = OptimizeResult()
            bad.update(success=False, message=f"Error: {str(exc)}", x=eng.get(X_K), fun=F_old, nit=nit - 1,
                       time=time.time() - t0, allvecs=allvecs, allfuns=allfuns, allerrs=allerrs)
            if dual_solver != "scipy":
                bad["dual_search_trials"] = dict(searches)
            return bad, _lib.ZF_BACKTRACK_FAILED
        if o["verbose"]:
            _print_row(nit, nit_int, err, fun, lr)
        beta = 0.0
        if o["nesterov"]:
            if not betas:   # (the recursion in blocks: same NumPy scalar expressions, fewer Python calls)
                b, t_state = momentum_factors(64, o["nesterov_ratio"], t_state)
                betas = list(b[::-1])
            beta = betas.pop()
        eng.commit(beta, o["nesterov"])     # x_{k-1} <- x_k <- x+ ; y_{k+1}
        F_old = F_new
        if o["return_all"]:
            allvecs.append(eng.get(X_K))
            allfuns.append(F_new)
            allerrs.append(err)
        if err < o["tol"]:   # :525 (the momentum update above does not touch x_k)
            res.status, res.message, res.success = 1, _MSG_OK, True
            status = _lib.ZF_CONVERGED
            break
    if status == _lib.ZF_MAXITER:
        res.status, res.message, res.success = 0, _MSG_MAXITER, False
    res.update(x=eng.get(X_K), fun=F_old, nit=nit, allvecs=allvecs, allfuns=allfuns, allerrs=allerrs,
               time=time.time() - t0)
    if dual_solver != "scipy":
        res["dual_search_trials"] = dict(searches)   # (extension field: absent with the reference's own search)
    return res, status


class _HandOver(tuple):
    """(nit_done, trials_done, lr, F(x_k), t_state, betas, last weights): where _solve_native_ahead left a solve whose
    device trial gave up; solve_native's sequential loop continues from it (the weights of the last completed trial
    are what a ``warm_start`` search starts from)."""


def _solve_native_ahead(eng, o, m, F_old, res, t0, searches):
    """The outer loop (:474-538) with every trial launched AHEAD of its predecessor's result
    (dual_solver="device", built-in problems): while the host reads the record of trial k and does its
    bookkeeping, the kernel of trial k + 1 - enqueued under the assumption that k is accepted, gated on
    the decision the kernel of k takes itself (:298-303) - is already running.  A rejected trial closes the
    gate: the trial launched ahead exits without touching anything, the speculative commit is undone and
    the line search continues with the smaller step.  Returns (result, status), or None when the problem
    has no device trial / the search cannot run on the device (the caller's loop takes over)."""
    from .proximal_gradient import (_MSG_BACKTRACK, _MSG_MAXITER, _MSG_OK, _print_row)

    lr = o["lr"]
    tol_i, max_i, dep = o["tol_internal"], o["max_iter_internal"], o["deprecated"]
    decay_one = o["decay_rate"] == 1
    warm = bool(o["warm_start"])
    t_state, betas = None, []

    def next_beta():
        nonlocal t_state, betas
        if not o["nesterov"]:
            return 0.0
        if not betas:
            b, t_state = momentum_factors(64, o["nesterov_ratio"], t_state)
            betas = list(b[::-1])
        return betas.pop()

    eng.prepare_async()
    ticket = eng.trial_launch(lr, F_old, dep, None, tol_i, max_i, tol_i, decay_one, False)
    if ticket is None:
        return None
    status = _lib.ZF_MAXITER
    nit = 0
    F_k = F_old
    w_last = None
    for nit in range(1, o["max_iter"] + 1):
        trials = 0
        ahead = None
        while True:
            trials += 1
            if ahead is None and nit < o["max_iter"]:
                # assume this trial is accepted: its commit and the next trial, before its result is known
                beta = next_beta()
                eng.commit(beta, o["nesterov"])
                eng.prepare_async()
                # (warm_start: its search starts from the weights this trial ends with - taken on the device, gated = 2)
                ahead = (eng.trial_launch(lr, None, dep, None, tol_i, max_i, tol_i, decay_one, 2 if warm else 1), beta)
            out = eng.trial_wait(ticket)
            if out is not None and isinstance(out[0], str) and out[0] == "timed out":
                # a grid-wide wait of the kernel gave up (its workgroups were not all resident): its record says
                # "not accepted", so the trial launched ahead found its gate closed.  Nothing of this trial is
                # used; the sequential loop continues this line search with the host-driven search.
                if ahead is not None:
                    eng.trial_wait(ahead[0])
                    eng.uncommit()
                    if o["nesterov"]:
                        betas.append(ahead[1])
                eng.device_timed_out(ticket)
                return _HandOver((nit - 1, trials - 1, lr, F_k, t_state, betas, w_last))
            if out is None or isinstance(out[0], str):
                # the search did not run on the device (non-finite dual values, e.g. F(x_0) = inf outside the
                # box): its record says "not accepted", so a trial launched ahead found its gate closed
                if ahead is not None:
                    eng.trial_wait(ahead[0])
                    eng.uncommit()
                if out is not None and nit == 1 and trials == 1:
                    return None                               # from the start: the caller's loop (host search)
                raise RuntimeError("the device-side trial was not attempted in the middle of a solve")
            weight, dual_fun, nit_int, err, f_x, g_x, f_y, accepted = out
            searches["device"] += 1
            w_last = weight
            fun = -dual_fun                                   # (:207)
            F_new = f_x + g_x                                 # (:295) formed and tested on the device
            if accepted:
                break
            # rejected: the trial launched ahead found its gate closed
            if ahead is not None:
                skipped = eng.trial_wait(ahead[0])
                assert skipped is None, "a gated trial ran although its predecessor was rejected"
                eng.uncommit()
                if o["nesterov"]:
                    betas.append(ahead[1])                    # (the factor belongs to the next ACCEPTED iteration)
                ahead = None
            if trials >= o["max_backtrack_iter"]:
                print(f"An error occurred: {_MSG_BACKTRACK}")
                bad = OptimizeResult()
                bad.update(success=False, message=f"Error: {_MSG_BACKTRACK}", x=eng.get(X_K), fun=F_k, nit=nit - 1,
                           time=time.time() - t0, allvecs=None, allfuns=None, allerrs=None)
                return bad, _lib.ZF_BACKTRACK_FAILED
            lr *= o["decay_rate"]
            # (warm_start: the retry starts from the weights of the trial just rejected, :286-288)
            ticket = eng.trial_launch(lr, F_k, dep, weight if warm else None, tol_i, max_i, tol_i, decay_one, False)
        if o["verbose"]:
            _print_row(nit, nit_int, err, fun, lr)
        F_k = F_new
        if ahead is None:                                     # the last iteration max_iter allows: commit for x_k
            eng.commit(next_beta(), o["nesterov"])
        else:
            ticket = ahead[0]
        if err < o["tol"]:   # :525
            res.status, res.message, res.success = 1, _MSG_OK, True
            status = _lib.ZF_CONVERGED
            if ahead is not None:
                eng.trial_wait(ahead[0])                      # (runs to its end: it only wrote y, J and the x+ slot)
            break
    if status == _lib.ZF_MAXITER:
        res.status, res.message, res.success = 0, _MSG_MAXITER, False
    res.update(x=eng.get(X_K), fun=F_k, nit=nit, allvecs=None, allfuns=None, allerrs=None, time=time.time() - t0)
    return res, status


# ---------------------------------------------------------------------------
# generic path: opaque host callbacks f, g, jac_f, prox_wsum_g with m >= 2
# ---------------------------------------------------------------------------
def trial_generic(ops, f, g, jac_f, prox, lr, x_old, y, w0, tol, max_iter, deprecated, solver="scipy"):
    """(:140-209) for m >= 2 with opaque callbacks.  Returns (x, fun, nit, weight, err).

    J lives on the GPU for the duration of the trial; w@J, the norms and J@(p-y) are
    HIP reductions; the opaque prox / g callbacks run where the user wrote them."""
    f_y = f(y)
    F_old = f(x_old) + g(x_old)
    J = np.asarray(jac_f(y), dtype=np.float64)
    m, n = J.shape
    eng = MoEngine(_lib.ZF_MO_GENERIC, m, n)   # no l1 / box: the device prox is the identity
    try:
        eng.put(Y, y)
        eng.set_jac(J)

        def dual(w):
            # identity device prox: p_dev = v = y - lr w@J ; gives |w@J|^2 and v itself
            _, _, ss_wJ, _ = eng.dual_eval(lr, w)
            eng.recover(lr, w)                       # x+ slot := v
            v = eng.get(X_NEW)
            p = prox(lr * w, v)                      # user's prox   (:164)
            g_p = g(p)                               # user's g      (:165)
            dots, ss_pv = eng.post_terms(lr, w, p)   # J@(p - y), |p - v|^2 on the GPU
            fun = -np.inner(w, g_p) - np.sqrt(ss_pv) ** 2 / 2 / lr + lr / 2 * np.sqrt(ss_wJ) ** 2
            jac = -g_p - dots
            if not deprecated:
                fun += np.inner(w, F_old - f_y)
                jac += F_old - f_y
            return fun, jac

        weight, dual_fun, nit_int = solve_dual(dual, m, w0, tol, max_iter, solver)
        eng.recover(lr, weight)
        x_new = prox(lr * weight, eng.get(X_NEW))    # (:206)
        _, _, err = ops.model_terms(np.zeros(n), x_new, y)
        return x_new, -dual_fun, nit_int, weight, err
    finally:
        eng.close()
