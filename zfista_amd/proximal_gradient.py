"""``minimize_proximal_gradient`` - drop-in for zfista's solver entry point.

Same signature, keyword defaults, result fields, warnings and error behaviour
as ``zfista.minimize_proximal_gradient`` (zfista/proximal_gradient.py:311-555).
Two execution paths, both on the GPU (there is no host fallback):

* native   - the four callbacks are the bound methods of one
             ``zfista_amd.problems.NativeProblem``: the iteration runs
             device-resident (fused trial kernel + decide kernel, see
             ``csrc/zf_solver.hip``); the host polls once per chunk of trials.
* generic  - arbitrary Python callables: the callbacks run where the user wrote
             them (host), the solver's own vector arithmetic (:148, :150-152,
             :510, :534, and the dual terms :162-173) runs in HIP kernels through
             ``zf_host_*``.
* tensor   - ``x0`` is a float64 CUDA tensor and the callbacks are written against
             device tensors (e.g. PyTorch on ROCm): iterates never leave HBM, the
             solver's vector arithmetic runs through ``zf_dev_*`` on the caller's
             stream, three scalars per trial cross PCIe (single objective).
"""
from __future__ import annotations

import collections.abc as _abc
import ctypes as C
import time
from warnings import warn

import numpy as np
from scipy.optimize import OptimizeResult

from . import _lib
from .engine import DeviceSolver, momentum_factors
from .problems import match_native, match_native_multi

_MSG_OK = "Optimization terminated successfully"   # proximal_gradient.py:527
_MSG_MAXITER = "Maximum number of iterations reached"   # :541
_MSG_BACKTRACK = "Backtracking failed to find a suitable stepsize."   # :307
_MSG_DEPRECATED = (
    "Using the deprecated option is not mathematically proven to converge. "
    "Please consider using the recommended condition instead."
)   # :446-447
class IterateHistory(_abc.Sequence):
    """``allvecs`` of a device-resident solve: x_0 ... x_nit, as the reference's list is
    (zfista/proximal_gradient.py:471,521-522), but kept where the kernels wrote it - a ring in HBM
    (``zf_solver_set_history``) - and brought to the host iterate by iterate on access.  Iterates
    the ring could not hold any longer were moved to host memory during the solve.  Indexing /
    iterating yield NumPy arrays; ``device(k)`` is the iterate as a CUDA tensor view (no copy)
    while it is still in the ring."""

    def __init__(self, x0, n, length, ring, cap, stride, host):
        self._x0, self._n, self._len = x0, n, length
        self._ring, self._cap, self._stride, self._host = ring, cap, stride, host

    def __len__(self):
        return self._len

    def device(self, k):
        if k in self._host or self._ring is None:
            return None
        return self._ring[(k % self._cap) * self._stride:(k % self._cap) * self._stride + self._n]

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(self._len))]
        if k < 0:
            k += self._len
        if not 0 <= k < self._len:
            raise IndexError("iterate index out of range")
        if k == 0:
            return self._x0          # the caller's own object, as in the reference (:471)
        if k not in self._host:
            self._host[k] = self.device(k).cpu().numpy()
        return self._host[k]


_HEADER = ["niter", "nit internal", "max(abs(xk - yk)))", "subprob func", "learning rate"]   # :24-30
_WIDTHS = [7, 7, 13, 13, 10]


def _print_header():
    fmt = "|" + "|".join(f"{{:^{w}}}" for w in _WIDTHS) + "|"
    print(fmt.format(*_HEADER))
    print(fmt.format(*["-" * w for w in _WIDTHS]))


def _print_row(nit, nit_internal, err, fun, lr):
    # The reference formats five columns from four values and raises IndexError
    # (:511-520); this prints the row the header promises (documented deviation).
    print(f"|{nit:^7}|{nit_internal:^7}|{err:^+13.4e}|{fun:^+13.4e}|{lr:^10.2e}|")


def minimize_proximal_gradient(
    f, g, jac_f, prox_wsum_g, x0,
    lr=1, tol=1e-5, tol_internal=1e-12, max_iter=1000000, max_iter_internal=100000,
    max_backtrack_iter=100, warm_start=False, decay_rate=0.5, nesterov=False,
    nesterov_ratio=(0, 0.25), return_all=False, verbose=False, deprecated=False,
    *, dual_solver=None, sub_iters=None, acceptance=None,
):
    """Minimise F = f + g by the (accelerated) proximal gradient method on MI355X.

    Parameters, returned ``OptimizeResult`` fields and messages are those of
    zfista/proximal_gradient.py:332-443.  ``x0`` may be a NumPy array (or, on the
    native path, a float64 CUDA tensor holding this rank's shard).

    Keyword-only extensions (not in the reference; the defaults reproduce it):

    dual_solver : {"scipy", "native", "device"}, m >= 2 only.  "scipy" (default) minimises the dual
        of every trial with the reference's two SciPy calls (:179-205).  "native" uses the library's
        own simplex Newton / bracketing solver (host loop, one kernel per evaluation); "device"
        runs that same search inside one persistent kernel per trial (recognised problems only).
        The keyword decides.  Only when it is not given, the environment variable ZF_DUAL_SOLVER may
        override the default (for running an unmodified caller against another search); the result then
        says so in an extra field ``dual_solver`` - numerics never change silently with the environment.
    sub_iters : {1, 2, 4, 8, 16}, separable single-objective problems only: iterations chained per
        pass over the data (temporal blocking).  Results do not depend on it.  Default 16 (8 with
        ``return_all``).
    acceptance : {"reference", "resolved"}.  "reference" (default) evaluates the sufficient-decrease test as
        zfista/proximal_gradient.py:303 writes it, ``F(x+) - F(x_k) <= fun + tol_internal`` - two differences of
        O(|F|) numbers, below double-precision resolution once ``|x+ - y|^2 << ulp(F)``: at n = 1e8 trials are then
        rejected by rounding noise (and a long solve ends in "Backtracking failed") exactly as the reference's would.
        "resolved" (separable single-objective problems, ``DiagQuadL1``; other problems raise) evaluates the same
        inequality with F(x_k) and g(x+) cancelled and ``f(x+) - f(y)`` accumulated element by element:
        ``[f(x+) - f(y)] - <grad f(y), x+ - y> - |x+ - y|^2 / 2 / lr <= tol_internal``.  Where the reference's
        evaluation resolves the test both take the same decisions; iterates of accepted trials are the same
        arithmetic.  The result says which one ran (field ``acceptance``).  Only when the keyword is not given,
        ZF_ACCEPT in the environment may override the default.
    """
    if deprecated:
        warn(_MSG_DEPRECATED, stacklevel=2)
    from_env = False
    if dual_solver is None:
        import os

        dual_solver = os.environ.get("ZF_DUAL_SOLVER")
        from_env = dual_solver is not None
        dual_solver = dual_solver or "scipy"
    if dual_solver not in ("scipy", "native", "device"):
        raise ValueError(f"dual_solver must be 'scipy', 'native' or 'device', got {dual_solver!r}")
    accept_from_env = False
    if acceptance is None:
        import os

        acceptance = os.environ.get("ZF_ACCEPT")
        accept_from_env = acceptance is not None
        acceptance = acceptance or "reference"
    if acceptance not in ("reference", "resolved"):
        raise ValueError(f"acceptance must be 'reference' or 'resolved', got {acceptance!r}")
    opts = dict(
        dual_solver=dual_solver, sub_iters=int(sub_iters or 0), acceptance=acceptance,
        lr=lr, tol=tol, tol_internal=tol_internal, max_iter=max_iter,
        max_iter_internal=max_iter_internal, max_backtrack_iter=max_backtrack_iter,
        warm_start=warm_start, decay_rate=decay_rate, nesterov=nesterov,
        nesterov_ratio=nesterov_ratio, return_all=return_all, verbose=verbose,
        deprecated=deprecated,
    )
    native = match_native(f, g, jac_f, prox_wsum_g)
    if native is not None and not (lr > 0 and decay_rate > 0 and native.lam >= 0):
        native = None   # the fused kernels assume a threshold lam * lr >= 0; the callback path does not
    native_multi = match_native_multi(f, g, jac_f, prox_wsum_g)
    if acceptance == "resolved" and not (native is not None and getattr(native, "separable", False)):
        if accept_from_env:
            opts["acceptance"] = acceptance = "reference"   # (the environment asks for what this problem has not: the reference's test runs)
        else:
            raise ValueError("acceptance='resolved' needs a separable native problem (zfista_amd.problems.DiagQuadL1): the "
                             "element-wise difference f(x+) - f(y) is formed inside its fused kernels")
    if native is not None:
        res, status = _solve_native(native, x0, opts)
    elif _is_device_tensor(x0) and native_multi is None:
        res, status = _solve_tensor(f, g, jac_f, prox_wsum_g, x0, opts)
    elif native_multi is not None:
        from . import multiobjective

        res, status = multiobjective.solve_native(native_multi, x0, opts)
    else:
        res, status = _solve_generic(f, g, jac_f, prox_wsum_g, x0, opts)
    if from_env:
        res["dual_solver"] = f"{dual_solver} (from the environment: ZF_DUAL_SOLVER)"
    if acceptance != "reference":
        res["acceptance"] = acceptance + (" (from the environment: ZF_ACCEPT)" if accept_from_env else "")
    overrides = _lib.env_overrides()
    if overrides:
        # run-time switches that select other kernels, launch geometries or numerics paths: a result produced under
        # one says so (an extra field; absent when the defaults ran)
        res["overrides"] = overrides
    if status == _lib.ZF_MAXITER:
        warn(res.message, stacklevel=2)   # :543
    return res


# ---------------------------------------------------------------------------
# native path: device-resident iteration
# ---------------------------------------------------------------------------
class NativeRun:
    """A device-resident solve that can be advanced in chunks (used by bench.py
    to time exactly K iterations with the inputs already resident in HBM)."""

    def __init__(self, problem, x0, opts, timing=False, solver_factory=None, _snapshot=None):
        self.problem = problem
        self.opts = opts
        fields, keep = problem._descriptor()
        options = dict(
            lr=float(opts["lr"]), tol=float(opts["tol"]), tol_internal=float(opts["tol_internal"]),
            decay_rate=float(opts["decay_rate"]), max_iter=int(opts["max_iter"]),
            max_backtrack_iter=int(opts["max_backtrack_iter"]),
            nesterov=int(bool(opts["nesterov"])), deprecated=int(bool(opts["deprecated"])),
            accept_mode=(_lib.ZF_ACCEPT_RESOLVED if opts.get("acceptance") == "resolved" else _lib.ZF_ACCEPT_REFERENCE),
            # return_all records every iterate into a ring in HBM as the trial computes it
            # (zf_solver_set_history): recording kernels exist for chains of 8 and of 1
            sub_iters=(int(opts.get("sub_iters", 0) or 0) if not opts.get("return_all")
                       else (1 if (int(opts.get("sub_iters", 0) or 0) in (1, 2, 4) or solver_factory is not None
                                   or opts.get("acceptance") == "resolved")   # (its recording kernels: single trials)
                             else 8)),   # (test stand-ins have no history ring: one iterate per pass)
        )
        if solver_factory is not None:
            # test seam: a stand-in with DeviceSolver's interface (tests/fake_engine.py)
            self.solver = solver_factory(fields, options, problem, x0)
        else:
            import torch

            if isinstance(x0, torch.Tensor):
                x0_dev = x0.to(device="cuda", dtype=torch.float64).contiguous()
            else:
                x0_dev = torch.from_numpy(np.ascontiguousarray(np.asarray(x0, dtype=np.float64))).cuda()
            if x0_dev.numel() != problem.n_features:
                raise ValueError(f"len(x) should be equal to n_features, got {x0}.")
            self.solver = DeviceSolver(fields, options, keepalive=keep, group=problem.group, timing=timing)
            if opts.get("return_all"):
                self._init_history(x0, x0_dev, int(opts["max_iter"]), opts.get("history_slots"))
            if _snapshot is None:
                self.solver.init(x0_dev.data_ptr())
            else:
                xp = torch.from_numpy(np.ascontiguousarray(_snapshot["x_prev"], dtype=np.float64)).cuda()
                ctl = _lib.Control.from_buffer_copy(np.ascontiguousarray(_snapshot["control"]).tobytes())
                self.solver.restore(x0_dev.data_ptr(), xp.data_ptr(), ctl)
            self._x0_dev = x0_dev
        # iterations one pass may accept (temporal blocking of separable f; csrc/zf_kernels_step.h)
        self.sub_iters = int(getattr(self.solver, "sub_iters", 1))
        self.ratio = opts["nesterov_ratio"]
        self._t_state = None
        self._beta_filled = 0     # momentum factors uploaded for accepted counts < this
        self.nit_seen = 0
        self.status = _lib.ZF_RUNNING
        ctl, _ = self.solver.poll()
        self.F0 = ctl.F_old
        self.status = ctl.status
        if _snapshot is not None:
            # the momentum recursion is replayed up to the saved iteration count (scalar work); the
            # factors from there on are uploaded by the next advance()
            self.nit_seen = int(ctl.nit)
            if opts["nesterov"] and self.nit_seen > 0:
                if self.nit_seen > 1:
                    _, self._t_state = momentum_factors(self.nit_seen - 1, self.ratio, None)
                self._beta_filled = self.nit_seen
                self._fill_beta(self.nit_seen + 1)

    # -- streaming return_all -------------------------------------------------------------------
    def _init_history(self, x0, x0_dev, max_iter, slots=None):
        """A ring of iterates in HBM, MODEST by default: what the solve can need, at most 256 MiB (and never more than
        a quarter of the free device memory) - but always the 2 S + 2 slots two chains of a pass take (n = 1e8: 18
        slots = 14 GB).  What the ring cannot keep is moved to the host in blocks between chunks of passes
        (PCIe-bound: 0.8 GB per iterate at n = 1e8).  ``history_slots`` asks for a larger ring (solves whose whole
        history should stay on the device; the default max_iter of 1e6 must not make every small ``return_all``
        solve - or every worker sharing a GPU - allocate gigabytes up front)."""
        import torch

        n = x0_dev.numel()
        stride = (n + 63) // 64 * 64
        S = int(self.solver.sub_iters)
        free, _ = torch.cuda.mem_get_info()
        budget = min(free // 4, 256 << 20) // (8 * stride)
        cap = int(slots) if slots else min(max_iter + 1, budget)
        cap = max(cap, 2 * S + 2)
        self._hist = torch.empty(cap * stride, dtype=torch.float64, device=x0_dev.device)
        self._hist[:n].copy_(x0_dev)
        self._hist_x0, self._hist_n, self._hist_cap, self._hist_stride = x0, n, cap, stride
        self._hist_host = {}          # iterates already moved to host memory
        self._hist_saved = 0          # iterations <= this are safe (x0, or on the host)
        self.solver.set_history(self._hist.data_ptr(), cap, stride)

    def _history_room(self):
        """Passes that may be enqueued before a slot still needed would be overwritten; iterates
        are moved to the host first when there is no room for a single pass."""
        S = self.sub_iters
        room = (self._hist_saved + self._hist_cap - 1 - self.nit_seen) // S
        if room < 1:
            for k in range(self._hist_saved + 1, self.nit_seen + 1):   # accepted, final, still in the ring
                lo = (k % self._hist_cap) * self._hist_stride
                self._hist_host[k] = self._hist[lo:lo + self._hist_n].cpu().numpy()
            self._hist_saved = self.nit_seen
            room = (self._hist_cap - 1) // S
        return room

    def history(self):
        """x_0 ... x_nit of the solve so far (IterateHistory), after collect()."""
        return IterateHistory(self._hist_x0, self._hist_n, self.nit_seen + 1, self._hist, self._hist_cap,
                              self._hist_stride, self._hist_host)

    def snapshot(self):
        """The state of the solve after the last advance(): x_k, x_{k-1} and the control block
        (host arrays; ``np.savez(path, **state)`` makes it a checkpoint file).  Iterations that
        were accepted by a chain that broke before its end are materialised first (one
        replay-only pass, zf_solver_flush), so the saved iterates are x_k, x_{k-1} of ``nit``."""
        ctl, _ = self.solver.poll()
        if ctl.status == _lib.ZF_RUNNING and ctl.lag > 0:
            self.solver.flush()
            ctl, _ = self.solver.poll()
        # (the launch geometry decides the order of the reduced sums, hence knife-edge accept / reject decisions: it travels
        #  with the state so that a resume under another rule - another build of the library - can say so)
        return dict(x=self.solver.get_x(), x_prev=self.solver.get_x_prev(),
                    control=np.frombuffer(bytes(ctl), dtype=np.uint8).copy(),
                    tiles_per_wg=np.int64(getattr(self.solver, "tiles_per_wg", 0) or 0))

    @classmethod
    def from_snapshot(cls, problem, state, opts, timing=False):
        """Continue a solve from ``snapshot()`` (possibly in another process, with another
        max_iter or chain length).  The continuation is bit-identical to the uninterrupted solve.
        ``return_all`` is refused: the iterates before the snapshot are not part of it, and a history
        whose first ``nit`` entries are uninitialised memory is worse than none."""
        if opts.get("return_all"):
            raise ValueError("return_all is not available for a solve resumed from a snapshot (the iterates before "
                             "the snapshot are not part of the saved state)")
        run = cls(problem, np.asarray(state["x"]), opts, timing=timing, _snapshot=state)
        saved = int(np.asarray(state["tiles_per_wg"])) if "tiles_per_wg" in state else 0
        now = int(getattr(run.solver, "tiles_per_wg", 0) or 0)
        if saved and now and saved != now:
            warn(f"the snapshot was taken with {saved} tiles per workgroup, this build runs the problem with {now}: the resumed "
                 "solve computes the same iterates, but its sums are added in another order and accept / reject decisions at "
                 "the resolution limit of the acceptance test may differ from the uninterrupted solve", stacklevel=2)
        return run

    def _fill_beta(self, upto):
        """Upload momentum factors for accepted-iteration counts < upto."""
        if not self.opts["nesterov"]:
            return
        # refill as far ahead as the ring allows - the slots of the ZF_MAX_LAG counts behind
        # nit_seen may still be replayed (zf_control.lag), the one of nit_seen + ZF_RING - ZF_MAX_LAG
        # would alias the oldest of them - bounded by what the solve can still use: one upload
        # per ~1000 iterations, and none between set_max_iter() and the passes that follow it
        ahead = self.nit_seen + _lib.ZF_RING - 1 - _lib.ZF_MAX_LAG
        upto = max(upto, min(ahead, int(self.opts["max_iter"]) + 1))
        if upto <= self._beta_filled:
            return
        first = self._beta_filled
        count = upto - first
        betas = np.zeros(count)
        lo = 0
        if first == 0:
            betas[0] = 0.0   # y_1 = x_0 (:465)
            lo = 1
        if count - lo > 0:
            vals, self._t_state = momentum_factors(count - lo, self.ratio, self._t_state)
            betas[lo:] = vals
        self.solver.set_beta(first, betas)
        self._beta_filled = upto

    def advance(self, steps):
        """Enqueue ``steps`` passes (each examines up to ``sub_iters`` trials), then
        synchronise.  Returns the trace rows of the iterations accepted meanwhile
        (array [k, ZF_TRACE_COLS])."""
        self.enqueue_only(steps)
        return self.collect()

    def enqueue_only(self, steps):
        # the trace / momentum rings hold ZF_RING iterations: never run further ahead of the host
        steps = int(min(steps, (_lib.ZF_RING - 1 - _lib.ZF_MAX_LAG) // self.sub_iters))
        if getattr(self, "_hist", None) is not None:
            steps = int(min(steps, self._history_room()))
        # + 1: the decide step of the last trial resolves the factor of the trial after it
        self._fill_beta(self.nit_seen + steps * self.sub_iters + 1)
        self.solver.enqueue(steps)

    def set_max_iter(self, max_iter):
        """Raise (or lower) max_iter of the live solve; a run stopped by it resumes (:539)."""
        self.solver.set_max_iter(int(max_iter))
        self.opts = dict(self.opts, max_iter=int(max_iter))
        ctl, _ = self.solver.poll()
        self.status = int(ctl.status)
        self._fill_beta(self.nit_seen + 1)   # momentum factors for the extended range, as far as the ring allows

    def collect(self):
        ctl, trace = self.solver.poll()
        idx = np.arange(self.nit_seen, ctl.nit) % _lib.ZF_RING
        rows = trace[idx].copy()
        self.nit_seen = int(ctl.nit)
        self.status = int(ctl.status)
        return rows


def _solve_native(problem, x0, opts, solver_factory=None):
    t0 = time.time()
    res = OptimizeResult(
        x0=x0, tol=opts["tol"], tol_internal=opts["tol_internal"],
        nesterov=opts["nesterov"], nesterov_ratio=opts["nesterov_ratio"],
    )
    if opts["verbose"]:
        _print_header()
    run = NativeRun(problem, x0, opts, solver_factory=solver_factory)
    return_all, verbose = opts["return_all"], opts["verbose"]
    allvecs = allfuns = allerrs = None
    if return_all:
        allfuns = [np.float64(run.F0)]
        allerrs = []
    streaming = return_all and getattr(run, "_hist", None) is not None   # (test stand-ins have no ring)
    if return_all and not streaming:
        allvecs = [x0]
    chunk = 1
    last_lr = float(opts["lr"])
    while run.status == _lib.ZF_RUNNING:
        before = run.nit_seen
        rows = run.advance(chunk)
        for k, row in enumerate(rows):
            if verbose:
                _print_row(before + k + 1, 1, row[_lib.TR_ERR], row[_lib.TR_FUN], row[_lib.TR_LR])
            if return_all:
                allfuns.append(np.float64(row[_lib.TR_F]))
                allerrs.append(np.float64(row[_lib.TR_ERR]))
            last_lr = row[_lib.TR_LR]
        if return_all and not streaming and len(rows):
            allvecs.append(run.solver.get_x())   # chunk == 1: exactly this iterate
        if streaming or not return_all:
            chunk = min(chunk * 2, 256)
    if streaming:
        allvecs = run.history()
    ctl = run.solver.ctl
    x = run.solver.get_x()
    F = np.float64(ctl.F_old)
    if getattr(problem, "array_valued", False):
        # the problem's f and g return arrays of one value (m = 1 by :143, e.g. the notebook's deblurring callbacks):
        # fun and allfuns carry that shape, as the reference's f(xk) + g(xk) would (:523, :547)
        F = np.array([F])
        if allfuns is not None:
            allfuns = [np.array([v]) for v in allfuns]
    if run.status == _lib.ZF_BACKTRACK_FAILED:
        # proximal_gradient.py:493-509: reported, not raised
        print(f"An error occurred: {_MSG_BACKTRACK}")
        err = OptimizeResult()
        err.update(success=False, message=f"Error: {_MSG_BACKTRACK}", x=x, fun=F, nit=int(ctl.nit),
                   time=time.time() - t0, allvecs=allvecs, allfuns=allfuns, allerrs=allerrs)
        run.solver.close()
        return err, run.status
    if run.status == _lib.ZF_CONVERGED:
        res.status, res.message, res.success = 1, _MSG_OK, True
    else:
        res.status, res.message, res.success = 0, _MSG_MAXITER, False
    res.update(x=x, fun=F, nit=int(ctl.nit), allvecs=allvecs, allfuns=allfuns, allerrs=allerrs,
               time=time.time() - t0)
    report = getattr(run.solver, "ahead_report", None)
    if report is not None:
        rep = report()
        if rep["timeouts"]:
            # passes that ran ahead of their predecessor's decision waited in vain (the device was shared): the solver
            # went back to one launch per pass - same result, and the result says so
            res["runahead"] = (f"switched off after {rep['timeouts']} wait(s) of run-ahead passes gave up "
                               f"({rep['void']} void passes)")
    run.solver.close()
    return res, run.status


# ---------------------------------------------------------------------------
# generic path: opaque callbacks, solver arithmetic in HIP kernels
# ---------------------------------------------------------------------------
class _VecOps:
    """The solver's own vector expressions, evaluated on the GPU (zf_host_*)."""

    def __init__(self):
        self.lib = _lib.require_gpu()

    @staticmethod
    def _h(a):
        return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))

    def grad_step(self, y, jac, lr):   # :148  y - lr * jac.flatten()
        y, jac = self._h(y), self._h(jac)
        out = np.empty_like(y)
        _lib.check(self.lib.zf_host_grad_step(C.c_void_p(_lib.ptr(out)), C.c_void_p(_lib.ptr(y)),
                                              C.c_void_p(_lib.ptr(jac)), float(lr), y.size))
        return out

    def model_terms(self, jac, x, y):   # :150-152, :510
        jac, x, y = self._h(jac), self._h(x), self._h(y)
        out = np.zeros(3)
        _lib.check(self.lib.zf_host_model_terms(C.c_void_p(_lib.ptr(jac)), C.c_void_p(_lib.ptr(x)),
                                                C.c_void_p(_lib.ptr(y)), x.size, C.c_void_p(_lib.ptr(out))))
        return np.float64(out[0]), np.float64(out[1]), np.float64(out[2])

    def momentum(self, x, x_old, beta):   # :534
        x, x_old = self._h(x), self._h(x_old)
        out = np.empty_like(x)
        _lib.check(self.lib.zf_host_momentum(C.c_void_p(_lib.ptr(out)), C.c_void_p(_lib.ptr(x)),
                                             C.c_void_p(_lib.ptr(x_old)), float(beta), x.size))
        return out


def _is_device_tensor(x):
    return type(x).__module__.split(".")[0] == "torch" and getattr(x, "is_cuda", False)


class _DevOps:
    """The solver's vector expressions on device tensors (zf_dev_*), on torch's current stream."""

    def __init__(self):
        import torch

        self.torch = torch
        self.lib = _lib.require_gpu()

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def _v(self, t):
        t = t.reshape(-1)
        if t.dtype != self.torch.float64 or not t.is_cuda:
            raise TypeError("tensor callbacks must return float64 CUDA tensors")
        return t.contiguous()

    def grad_step(self, y, jac, lr):   # :148
        y, jac = self._v(y), self._v(jac)
        out = self.torch.empty_like(y)
        _lib.check(self.lib.zf_dev_grad_step(C.c_void_p(out.data_ptr()), C.c_void_p(y.data_ptr()),
                                             C.c_void_p(jac.data_ptr()), float(lr), y.numel(), self._stream()))
        return out

    def model_terms(self, jac, x, y):   # :150-152, :510
        jac, x, y = self._v(jac), self._v(x), self._v(y)
        out = np.zeros(3)
        _lib.check(self.lib.zf_dev_model_terms(C.c_void_p(jac.data_ptr()), C.c_void_p(x.data_ptr()),
                                               C.c_void_p(y.data_ptr()), x.numel(), C.c_void_p(_lib.ptr(out)),
                                               self._stream()))
        return np.float64(out[0]), np.float64(out[1]), np.float64(out[2])

    def model_terms_dev(self, jac, x, y):
        """The same three scalars left on the device (a 3-vector tensor), nothing synchronised."""
        jac, x, y = self._v(jac), self._v(x), self._v(y)
        out = self.torch.empty(3, dtype=self.torch.float64, device=x.device)
        _lib.check(self.lib.zf_dev_model_terms_async(C.c_void_p(jac.data_ptr()), C.c_void_p(x.data_ptr()),
                                                     C.c_void_p(y.data_ptr()), x.numel(), C.c_void_p(out.data_ptr()),
                                                     self._stream()))
        return out

    def momentum(self, x, x_old, beta):   # :534
        x, x_old = self._v(x), self._v(x_old)
        out = self.torch.empty_like(x)
        _lib.check(self.lib.zf_dev_momentum(C.c_void_p(out.data_ptr()), C.c_void_p(x.data_ptr()),
                                            C.c_void_p(x_old.data_ptr()), float(beta), x.numel(), self._stream()))
        return out

    # -- m >= 2 (:162-173): the two vector expressions around the caller's prox and g --------------
    def _jac(self, J, n):
        if J.dtype != self.torch.float64 or not J.is_cuda or J.dim() != 2 or J.shape[1] != n:
            raise TypeError(f"jac_f must return a float64 CUDA tensor of shape (m, {n})")
        return J.contiguous()

    def mo_combine(self, y, J, w, lr):
        """v = y - lr (w @ J) and |w @ J|^2 (a 1-element device tensor); w is a host array."""
        y = self._v(y)
        J = self._jac(J, y.numel())
        w = np.ascontiguousarray(w, dtype=np.float64)
        v = self.torch.empty_like(y)
        ss = self.torch.empty(1, dtype=self.torch.float64, device=y.device)
        _lib.check(self.lib.zf_dev_mo_combine(C.c_void_p(v.data_ptr()), C.c_void_p(y.data_ptr()),
                                              C.c_void_p(J.data_ptr()), C.c_void_p(_lib.ptr(w)), float(lr),
                                              int(J.shape[0]), y.numel(), C.c_void_p(ss.data_ptr()), self._stream()))
        return v, ss

    def mo_post_terms(self, J, y, p, v):
        """(J_i . (p - y))_i and |p - v|^2 as one (m + 1)-element device tensor."""
        y, p, v = self._v(y), self._v(p), self._v(v)
        J = self._jac(J, y.numel())
        m = int(J.shape[0])
        out = self.torch.empty(m + 1, dtype=self.torch.float64, device=y.device)
        _lib.check(self.lib.zf_dev_mo_post_terms(C.c_void_p(J.data_ptr()), C.c_void_p(y.data_ptr()),
                                                 C.c_void_p(p.data_ptr()), C.c_void_p(v.data_ptr()), m, y.numel(),
                                                 C.c_void_p(out.data_ptr()), self._stream()))
        return out


def _scalar(v):
    """A callback's objective value as a NumPy float (0-dim / 1-element tensors allowed)."""
    if hasattr(v, "numel"):
        if v.numel() != 1:
            raise TypeError(f"expected one objective value, got a tensor of {v.numel()} (m >= 2 problems return (m,) "
                            "tensors from f AND g: _solve_tensor_multi)")
        return np.float64(v.item())
    return np.float64(v)


def _values(v, m):
    """A callback's (m,) objective values as a host array (device tensors, NumPy arrays and lists allowed)."""
    if hasattr(v, "detach"):
        v = v.detach().reshape(-1).cpu().numpy()
    v = np.asarray(v, dtype=np.float64).reshape(-1)
    if v.size != m:
        raise ValueError(f"expected {m} objective values, got {v.size}")
    return v


# ---------------------------------------------------------------------------
# callback paths: ONE outer loop (proximal_gradient.py:463-554), three providers of its trial
# ---------------------------------------------------------------------------
class _TrialResult:
    """What one line-search trial hands to the loop: the trial point, the model value (:149-155 / :207), the
    inner iteration count, the dual weights (m >= 2), max|x+ - y| (:510), F(x+) (:295) and f(x+) - f(y) for the
    deprecated acceptance test (:301)."""

    def __init__(self, x, fun, nit_int, weight, err, F_new, df):
        self.x, self.fun, self.nit_int, self.weight, self.err, self.F_new, self.df = x, fun, nit_int, weight, err, F_new, df
        self.f_new = None   # (tensor paths: f(x+), carried to the next line search as f(x_k))


def _solve_callbacks(P, x0, o):
    """The loop of proximal_gradient.py:463-554 for callbacks, written once.  ``P`` provides what differs between
    opaque NumPy callbacks (_GenericOps), callbacks on device tensors (_TensorOps) and their m >= 2 form
    (_TensorMultiOps): ``start`` (F(x0)), ``begin`` (what a line search fixes: F(x_k) and, for the tensor paths,
    f(y), jac_f(y)), ``trial`` (one trial at a step size), ``momentum`` (:534) and the values reported with
    results (``F_record``, ``F_final``, ``F_error``)."""
    t0 = time.time()
    res = OptimizeResult(x0=x0, tol=o["tol"], tol_internal=o["tol_internal"],
                         nesterov=o["nesterov"], nesterov_ratio=o["nesterov_ratio"])
    if o["verbose"]:
        _print_header()
    x_old = x_cur = y = x0
    F_old = P.start(x0)
    w0 = np.ones(P.m) / P.m if P.m > 1 else None
    lr = o["lr"]
    allvecs = allfuns = allerrs = None
    if o["return_all"]:
        allvecs, allfuns, allerrs = [x0], [F_old], []
    t_state = None
    status = _lib.ZF_MAXITER
    nit = 0
    for nit in range(1, o["max_iter"] + 1):
        try:
            ls = P.begin(x_old, y, F_old)                      # :279 (and :140, :142 where they are cached)
            accepted = False
            for _ in range(o["max_backtrack_iter"]):
                tr = P.trial(ls, lr, x_old, y, w0)
                if w0 is not None and o["warm_start"]:
                    w0 = tr.weight
                if o["decay_rate"] == 1:                       # :298
                    accepted = True
                elif o["deprecated"]:                          # :301
                    accepted = bool(np.all(tr.df() <= tr.fun + o["tol_internal"]))
                else:                                          # :303
                    accepted = bool(np.all(tr.F_new - ls.F_old <= tr.fun + o["tol_internal"]))
                if accepted:
                    break
                lr *= o["decay_rate"]                          # :305
            if not accepted:
                raise RuntimeError(_MSG_BACKTRACK)             # :306-307
        except Exception as exc:   # :493-509: reported, not raised
            print(f"An error occurred: {exc}")
            bad = OptimizeResult()
            bad.update(success=False, message=f"Error: {str(exc)}", x=x_old, fun=P.F_error(x_old, F_old),
                       nit=nit - 1, time=time.time() - t0,
                       allvecs=allvecs, allfuns=allfuns, allerrs=allerrs)
            return bad, _lib.ZF_BACKTRACK_FAILED
        x_cur = tr.x
        if o["verbose"]:
            _print_row(nit, tr.nit_int, tr.err, tr.fun, lr)
        if o["return_all"]:
            allvecs.append(x_cur)
            allfuns.append(P.F_record(tr))
            allerrs.append(tr.err)
        F_old = tr.F_new
        P.accepted(tr)
        if tr.err < o["tol"]:   # :525
            res.status, res.message, res.success = 1, _MSG_OK, True
            status = _lib.ZF_CONVERGED
            break
        if o["nesterov"]:       # :531-535
            beta, t_state = momentum_factors(1, o["nesterov_ratio"], t_state)
            y = P.momentum(x_cur, x_old, beta[0])
        else:
            y = x_cur
        x_old = x_cur
    if status == _lib.ZF_MAXITER:
        res.status, res.message, res.success = 0, _MSG_MAXITER, False
    res.update(x=x_cur, fun=P.F_final(x_cur, F_old), nit=nit, allvecs=allvecs, allfuns=allfuns, allerrs=allerrs,
               time=time.time() - t0)
    return res, status


class _LineSearch:
    __slots__ = ("F_old", "f_y", "jac")

    def __init__(self, F_old, f_y=None, jac=None):
        self.F_old, self.f_y, self.jac = F_old, f_y, jac


class _GenericOps:
    """Opaque NumPy callbacks: they are called where and as often as the reference calls them (:140-142, :279,
    :295, :301, :501, :523, :547); the solver's own vector expressions run in HIP kernels (zf_host_*)."""

    def __init__(self, f, g, jac_f, prox, o):
        self.f, self.g, self.jac_f, self.prox, self.o = f, g, jac_f, prox, o
        self.ops = _VecOps()
        self.m = 1

    def start(self, x0):
        f0 = self.f(x0)                                         # :466
        self.m = _objectives(f0)
        # g(x0) is called before the loop only for the record (:472); every line search re-evaluates F(x_k) (:279), so
        # nothing else needs the value - a callback that counts its calls sees exactly the reference's
        return f0 + self.g(x0) if self.o["return_all"] else None

    def begin(self, x_old, y, F_old):
        return _LineSearch(self.f(x_old) + self.g(x_old))      # :279

    def trial(self, ls, lr, x_old, y, w0):
        o = self.o
        if self.m == 1:
            x, fun, nit_int, weight, err = _trial_generic_single(self.ops, self.f, self.g, self.jac_f, self.prox, lr,
                                                                 x_old, y, o["deprecated"])
        else:
            from . import multiobjective

            x, fun, nit_int, weight, err = multiobjective.trial_generic(
                self.ops, self.f, self.g, self.jac_f, self.prox, lr, x_old, y, w0, o["tol_internal"],
                o["max_iter_internal"], o["deprecated"], o.get("dual_solver", "scipy"))
        F_new = self.f(x) + self.g(x)                           # :295
        return _TrialResult(x, fun, nit_int, weight, err, F_new, lambda: self.f(x) - self.f(y))   # :301

    def accepted(self, tr):
        pass

    def momentum(self, x, x_old, beta):
        return self.ops.momentum(x, x_old, beta)

    def F_record(self, tr):
        return self.f(tr.x) + self.g(tr.x)                      # :523

    def F_final(self, x, F_old):
        return self.f(x) + self.g(x)                            # :547

    def F_error(self, x_old, F_old):
        return self.f(x_old) + self.g(x_old)                    # :501


class _TensorOps:
    """Callbacks on device tensors, m = 1.  f(y) and jac_f(y) are evaluated once per line search (y is fixed
    while lr shrinks) and F(x_k) is the value obtained when x_k was accepted - the reference re-evaluates both
    per trial (:140-142, :279), which returns the same numbers for deterministic callbacks.  Iterates never leave
    HBM; one transfer of five scalars per trial crosses PCIe."""

    m = 1

    def __init__(self, f, g, jac_f, prox, o, f0):
        import torch

        self.torch = torch
        self.f, self.g, self.jac_f, self.prox, self.o = f, g, jac_f, prox, o
        self.ops = _DevOps()
        self.f0 = f0
        self.f_old = None        # f(x_k) of the accepted iterate
        self.x_k = None

    def start(self, x0):
        self.f_old = _scalar(self.f0)
        self.x_k = x0
        return self.f_old + _scalar(self.g(x0))

    def begin(self, x_old, y, F_old):
        f_y = self.f_old if y is x_old else _scalar(self.f(y))   # :140
        return _LineSearch(F_old, f_y, self.jac_f(y))             # :142

    def trial(self, ls, lr, x_old, y, w0):
        torch, ops, o = self.torch, self.ops, self.o
        x = self.prox(lr, ops.grad_step(y, ls.jac, lr))                  # :148
        terms = ops.model_terms_dev(ls.jac, x, y)                        # :150-152, :510 (device)
        g_val, f_val = self.g(x), self.f(x)
        if _is_device_tensor(g_val) and _is_device_tensor(f_val) and g_val.numel() == 1 == f_val.numel():
            # one transfer for all five scalars of the trial
            five = torch.cat([terms, g_val.reshape(1).to(torch.float64),
                              f_val.reshape(1).to(torch.float64)]).cpu().numpy()
            dot, ss, err, g_new, f_new = (np.float64(v) for v in five)
        else:
            dot, ss, err = (np.float64(v) for v in terms.cpu().numpy())
            g_new, f_new = _scalar(g_val), _scalar(f_val)
        fun = np.float64(dot + g_new + np.sqrt(ss) ** 2 / 2 / lr)       # :149-152
        if not o["deprecated"]:
            fun = fun + (ls.f_y - ls.F_old)                              # :155
        tr = _TrialResult(x, fun, 1, None, err, f_new + g_new, lambda: f_new - ls.f_y)   # :295, :301
        tr.f_new = f_new
        return tr

    def accepted(self, tr):
        self.f_old = tr.f_new

    def momentum(self, x, x_old, beta):
        return self.ops.momentum(x, x_old, beta)

    def F_record(self, tr):
        return tr.F_new

    def F_final(self, x, F_old):
        return F_old

    def F_error(self, x_old, F_old):
        return F_old


class _TensorMultiOps(_TensorOps):
    """The m >= 2 trial (:161-209) for callbacks on device tensors: f, g return (m,) tensors, jac_f an (m, n)
    tensor, prox_wsum_g(weight, x) receives the weight lr * w as an (m,) device tensor.  Iterates, J and every
    O(n) expression stay in HBM; the solver's own vector expressions (v = y - lr w@J, |w@J|^2, J(p - y),
    |p - v|^2, max|x+ - y|) are HIP kernels (zf_dev_mo_*), the m-dimensional dual search runs on the host exactly
    as for NumPy callbacks (``dual_solver=``), and one transfer of 2m + 2 scalars per dual evaluation crosses PCIe."""

    def __init__(self, f, g, jac_f, prox, o, f0, m, x0):
        super().__init__(f, g, jac_f, prox, o, f0)
        self.m = m
        self.device = x0.device
        solver = o.get("dual_solver", "scipy")
        # (the persistent-kernel search needs the library's own prox: built-in problems)
        self.solver = "native" if solver == "device" else solver

    def start(self, x0):
        self.f_old = _values(self.f0, self.m)
        return self.f_old + _values(self.g(x0), self.m)

    def begin(self, x_old, y, F_old):
        f_y = self.f_old if y is x_old else _values(self.f(y), self.m)   # :140
        return _LineSearch(F_old, f_y, self.jac_f(y))                     # :142

    def trial(self, ls, lr, x_old, y, w0):
        from . import multiobjective

        torch, ops, o, m = self.torch, self.ops, self.o, self.m
        J, f_y, F_old = ls.jac, ls.f_y, ls.F_old

        def dual(w):                                                     # _dual_minimized_fun_jac, :162-177
            v, ss_wJ = ops.mo_combine(y, J, w, lr)
            p = self.prox(torch.as_tensor(lr * w, dtype=torch.float64, device=self.device), v)   # :164
            g_p = self.g(p)                                                                        # :165
            post = ops.mo_post_terms(J, y, p, v)
            if _is_device_tensor(g_p):
                host = torch.cat([ss_wJ, post, g_p.reshape(-1).to(torch.float64)]).cpu().numpy()
                g_pv = host[m + 2:]
            else:
                host = torch.cat([ss_wJ, post]).cpu().numpy()
                g_pv = _values(g_p, m)
            ss_w, dots, ss_pv = np.float64(host[0]), host[1:m + 1], np.float64(host[m + 1])
            fun = -np.inner(w, g_pv) - np.sqrt(ss_pv) ** 2 / 2 / lr + lr / 2 * np.sqrt(ss_w) ** 2
            jac = -g_pv - dots
            if not o["deprecated"]:
                fun += np.inner(w, F_old - f_y)
                jac = jac + (F_old - f_y)
            return fun, jac

        weight, dual_fun, nit_int = multiobjective.solve_dual(dual, m, w0, o["tol_internal"], o["max_iter_internal"],
                                                              self.solver)
        v, _ = ops.mo_combine(y, J, weight, lr)
        x = self.prox(torch.as_tensor(lr * weight, dtype=torch.float64, device=self.device), v)   # :206
        err = np.float64(ops.model_terms_dev(v, x, y).cpu().numpy()[2])                            # :510
        f_new, g_new = _values(self.f(x), m), _values(self.g(x), m)
        tr = _TrialResult(x, -dual_fun, nit_int, weight, err, f_new + g_new, lambda: f_new - f_y)   # :207, :295, :301
        tr.f_new = f_new
        return tr


def _solve_tensor(f, g, jac_f, prox, x0, o):
    """Callbacks on device tensors: the shared loop with the tensor providers above."""
    import torch

    if x0.dtype != torch.float64:
        raise TypeError("x0 must be a float64 tensor (the reference is float64 throughout)")
    f0 = f(x0)
    m = int(f0.numel()) if hasattr(f0, "numel") else (f0.shape[0] if isinstance(f0, np.ndarray) else 1)   # :143,:467
    P = _TensorMultiOps(f, g, jac_f, prox, o, f0, m, x0) if m > 1 else _TensorOps(f, g, jac_f, prox, o, f0)
    return _solve_callbacks(P, x0, o)


def _objectives(value):
    return value.shape[0] if isinstance(value, np.ndarray) else 1   # :143,:467


def _trial_generic_single(ops, f, g, jac_f, prox, lr, x_old, y, deprecated):
    """(:140-157) with the vector arithmetic on the device."""
    f_y = f(y)
    F_old = f(x_old) + g(x_old)
    jac = jac_f(y)
    x_new = prox(lr, ops.grad_step(y, jac, lr))
    dot, ss, err = ops.model_terms(jac, x_new, y)
    fun = float(dot + g(x_new) + np.sqrt(ss) ** 2 / 2 / lr)
    if not deprecated:
        fun += f_y - F_old
    return x_new, fun, 1, None, err


def _solve_generic(f, g, jac_f, prox, x0, o):
    """Opaque NumPy callbacks: the shared loop with _GenericOps."""
    return _solve_callbacks(_GenericOps(f, g, jac_f, prox, o), x0, o)
