"""Host driver of the device-resident solver (``zf_solver`` in libzfista_hip.so).

PyTorch is plumbing here: it owns device allocations handed in by the caller,
names the HIP stream the kernels are enqueued on, and (for a sharded decision
vector) carries the per-trial scalar packs between ranks with
``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box).  All
solver arithmetic happens in the library's HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib


def momentum_factors(count: int, ratio, state=None):
    """The next ``count`` momentum factors of zfista/proximal_gradient.py:531-535.

    ``t_next = sqrt(t^2 - a t + b) + 1/2``, ``beta = (t - 1) / t_next`` evaluated
    with the same NumPy scalar expressions as the reference, so the factors the
    trial kernel multiplies with are bit-identical to the reference's.
    Returns (betas, state) where state carries ``t`` between calls.
    """
    a, b = ratio
    t_prev = 1 if state is None else state
    out = np.empty(count, dtype=np.float64)
    for j in range(count):
        t_next = np.sqrt(t_prev**2 - a * t_prev + b) + 0.5
        out[j] = (t_prev - 1) / t_next
        t_prev = t_next
    return out, t_prev


def gather_packs(pack_all, pack_local, group):
    """C1: the one collective of a sharded trial - a packed
    all-gather of ZF_PACK_LEN doubles (64 B) per rank, rank-major.  Every rank
    then adds the packs in rank order inside the decide step, so all ranks take
    bitwise-identical branch decisions.  Runs on RCCL for CUDA tensors (backend
    "nccl") and on gloo for the CPU tests."""
    import torch.distributed as dist

    if pack_local.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal of the multi-process path on one GPU (several ranks cannot share a device
        # under RCCL): stage the few hundred bytes through the host
        host = pack_all.cpu()
        dist.all_gather_into_tensor(host, pack_local.cpu(), group=group)
        pack_all.copy_(host)
        return
    dist.all_gather_into_tensor(pack_all, pack_local, group=group)


class DeviceSolver:
    """One ``zf_solver``: a problem descriptor bound to device buffers + options.

    Parameters
    ----------
    desc_fields : dict of ``zf_problem_desc`` fields (device pointers as ints)
    keepalive   : objects (torch tensors) that own the memory behind those pointers
    group       : ``torch.distributed`` process group when x is sharded, else None
    """

    def __init__(self, desc_fields, options, keepalive=(), group=None, stream=None, timing=False):
        self.lib = _lib.require_gpu()
        self._keep = list(keepalive)
        self.group = group
        self.world = int(desc_fields.get("world", 1))
        self.rank = int(desc_fields.get("rank", 0))
        self.n = int(desc_fields["n"])
        self.kind = int(desc_fields["kind"])
        self.m_rows = int(desc_fields.get("m_rows", 0) or 0)
        self.row_sharded = bool(desc_fields.get("row_sharded", 0))
        if stream is None:
            import torch

            stream = torch.cuda.current_stream().cuda_stream
        self.stream = stream
        d = _lib.ProblemDesc()
        for k, v in desc_fields.items():
            setattr(d, k, v)
        o = _lib.Options()
        for k, v in options.items():
            setattr(o, k, v)
        self.nesterov = bool(options.get("nesterov", 0))
        h = C.c_void_p()
        _lib.check(self.lib.zf_solver_create(C.byref(h), C.byref(d), C.byref(o), C.c_void_p(stream)),
                   "zf_solver_create")
        self.handle = h
        sub = C.c_int32(1)
        _lib.check(self.lib.zf_solver_sub_iters(h, C.byref(sub)), "zf_solver_sub_iters")
        self.sub_iters = int(sub.value)   # iterations one pass may accept (temporal blocking)
        self.ctl = _lib.Control()
        self.trace = np.zeros((_lib.ZF_RING, _lib.ZF_TRACE_COLS), dtype=np.float64)
        self._pack_local = self._pack_all = None
        if timing:
            _lib.check(self.lib.zf_solver_set_timing(self.handle, 1))
        # Sharded x.  Preferred: the library's own RCCL communicator (zfista_amd.comm.LibComm, given
        # as `group` or derived from an nccl process group): the exchanges of a pass are issued by
        # zf_solver_enqueue_steps itself, no host code per pass.  Otherwise (gloo rehearsals, thread
        # stand-ins, ZF_COMM=torch) the host runs trial -> torch.distributed all-gather -> decide.
        # ZF_FORCE_SPLIT=1 runs that sequence even for one rank.
        from .comm import LibComm

        self.comm = None
        if isinstance(group, LibComm):
            self.comm = group
        elif self.world > 1 and group is not None and os.environ.get("ZF_COMM", "lib") == "lib":
            import torch.distributed as dist

            try:   # only the QUESTION "is this an nccl process group" may fail quietly (the tests' in-process
                   # stand-ins are no process groups at all)
                is_nccl = dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl"
            except (RuntimeError, ValueError, TypeError, AttributeError):
                is_nccl = False
            if is_nccl:
                # collective (an object broadcast + ncclCommInitRank + an all-reduce of "did it work"): either every
                # rank gets a communicator or none does (None: the torch.distributed sequence below, with a warning) -
                # one rank on one sequence and the others issuing RCCL all-gathers inside zf_solver_enqueue_steps
                # would be a hang
                self.comm = LibComm.from_group(group)
        if self.comm is not None:
            _lib.check(self.lib.zf_solver_set_comm(self.handle, self.comm.handle), "zf_solver_set_comm")
        self.split = self.comm is None and (self.world > 1 or os.environ.get("ZF_FORCE_SPLIT") == "1")
        if self.split:
            self._wrap_packs()

    # -- sharded x: the per-trial exchange -------------------------------------
    def _wrap_packs(self):
        """torch-owned pack buffers the collective runs on; the library writes /
        reads them in place (zf_solver_set_pack_buffers)."""
        import torch

        plen = _lib.ZF_PACK_LEN * self.sub_iters   # one pack per sub-iteration of a pass
        self._pack_local = torch.zeros(plen, dtype=torch.float64, device="cuda")
        self._pack_all = torch.zeros(plen * self.world, dtype=torch.float64, device="cuda")
        _lib.check(self.lib.zf_solver_set_pack_buffers(
            self.handle, C.c_void_p(self._pack_local.data_ptr()), C.c_void_p(self._pack_all.data_ptr())))
        self._s_part = self._s_all = None
        if self.kind == _lib.ZF_PROBLEM_LEAST_SQUARES_L1 and self.world > 1:
            # C2: the m-vector A_p x_p of every rank (column blocks) or the n-vector
            # A_p^T r_p (row blocks), gathered once per trial
            length = self.n if self.row_sharded else self.m_rows
            self._s_part = torch.zeros(length, dtype=torch.float64, device="cuda")
            self._s_all = torch.zeros(length * self.world, dtype=torch.float64, device="cuda")
            _lib.check(self.lib.zf_solver_set_svec_buffers(
                self.handle, C.c_void_p(self._s_part.data_ptr()), C.c_void_p(self._s_all.data_ptr())))

    def _gather(self):
        gather_packs(self._pack_all, self._pack_local, self.group)

    def _gather_svec(self):
        if self._s_part is not None:
            gather_packs(self._s_all, self._s_part, self.group)   # same rank-major all-gather

    def trial_finish(self):
        _lib.check(self.lib.zf_solver_enqueue_trial_finish(self.handle), "enqueue_trial_finish")

    def init_finish(self):
        _lib.check(self.lib.zf_solver_enqueue_init_finish(self.handle), "enqueue_init_finish")

    # the two halves of a sharded step, and of the sharded initialisation; the
    # caller runs the exchange between them (enqueue()/init() do exactly that)
    def enqueue_trial(self):
        _lib.check(self.lib.zf_solver_enqueue_trial(self.handle), "enqueue_trial")

    def enqueue_decide(self):
        _lib.check(self.lib.zf_solver_enqueue_decide(self.handle), "enqueue_decide")

    def init_begin(self, x0_dev_ptr: int):
        _lib.check(self.lib.zf_solver_enqueue_init(self.handle, C.c_void_p(x0_dev_ptr)), "init")

    def init_commit(self):
        _lib.check(self.lib.zf_solver_enqueue_init_commit(self.handle), "init_commit")

    # -- life cycle ---------------------------------------------------------------
    def init(self, x0_dev_ptr: int):
        if self.comm is not None:   # init + exchanges + commit inside the library
            _lib.check(self.lib.zf_solver_enqueue_init_all(self.handle, C.c_void_p(x0_dev_ptr)), "init_all")
            t = C.c_int32(1)
            _lib.check(self.lib.zf_solver_autotune(self.handle, C.byref(t)), "geometry")
            self.tiles_per_wg = int(t.value)
            return
        self.init_begin(x0_dev_ptr)
        if self.split:
            if not self.row_sharded:   # (row blocks: A_p x0 is local, nothing to exchange at the start)
                self._gather_svec()
            self.init_finish()
            self._gather()
        self.init_commit()
        # launch geometry: a function of n only (it fixes the rounding of the reduced sums)
        t = C.c_int32(1)
        _lib.check(self.lib.zf_solver_autotune(self.handle, C.byref(t)), "geometry")
        self.tiles_per_wg = int(t.value)

    def set_beta(self, first: int, betas: np.ndarray):
        betas = np.ascontiguousarray(betas, dtype=np.float64)
        _lib.check(self.lib.zf_solver_set_beta(self.handle, first, C.c_void_p(_lib.ptr(betas)), betas.size),
                   "set_beta")

    def enqueue(self, steps: int):
        """Enqueue ``steps`` passes (each = up to ``sub_iters`` line-search trials); no host
        synchronisation."""
        if not self.split:
            _lib.check(self.lib.zf_solver_enqueue_steps(self.handle, steps), "enqueue_steps")
            return
        for _ in range(steps):
            self.enqueue_trial()
            self._gather_svec()
            self.trial_finish()
            self._gather()
            self.enqueue_decide()

    def set_history(self, ring_dev_ptr: int, cap_slots: int, stride: int):
        """Streaming return_all: every trial stores its iterate into slot (k + 1) % cap_slots."""
        _lib.check(self.lib.zf_solver_set_history(self.handle, C.c_void_p(ring_dev_ptr), int(cap_slots), int(stride)),
                   "set_history")

    def flush(self):
        """One replay-only pass that stores x_k, x_{k-1} when iterates lag behind the accepted
        count (zf_control.lag); no-op otherwise."""
        _lib.check(self.lib.zf_solver_flush(self.handle), "flush")
        self.enqueue(1)

    def get_x_prev(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.float64)
        _lib.check(self.lib.zf_solver_get_x_prev(self.handle, C.c_void_p(_lib.ptr(out)), out.size), "get_x_prev")
        return out

    def restore(self, xk_dev_ptr: int, xprev_dev_ptr: int, ctl):
        """Instead of init(): continue from a saved (x_k, x_{k-1}, control block)."""
        _lib.check(self.lib.zf_solver_restore(self.handle, C.c_void_p(xk_dev_ptr), C.c_void_p(xprev_dev_ptr),
                                              C.byref(ctl), C.sizeof(ctl)), "restore")
        # launch geometry: a function of n only (it fixes the rounding of the reduced sums)
        t = C.c_int32(1)
        _lib.check(self.lib.zf_solver_autotune(self.handle, C.byref(t)), "geometry")
        self.tiles_per_wg = int(t.value)

    def set_max_iter(self, max_iter: int):
        _lib.check(self.lib.zf_solver_set_max_iter(self.handle, int(max_iter)), "set_max_iter")

    def poll(self):
        """Synchronise and fetch the control block + trace ring."""
        _lib.check(self.lib.zf_solver_poll(self.handle, C.byref(self.ctl), C.sizeof(self.ctl),
                                           C.c_void_p(_lib.ptr(self.trace)), self.trace.nbytes), "poll")
        return self.ctl, self.trace

    def get_x(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.float64)
        _lib.check(self.lib.zf_solver_get_x(self.handle, C.c_void_p(_lib.ptr(out)), out.size), "get_x")
        return out

    def x_dev_ptr(self) -> int:
        p = C.c_void_p()
        _lib.check(self.lib.zf_solver_x_dev(self.handle, C.byref(p)))
        return p.value

    def trial_kernel_ms(self):
        ms, cnt = C.c_double(0.0), C.c_int64(0)
        _lib.check(self.lib.zf_solver_trial_kernel_ms(self.handle, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def pass_stats(self):
        """(mean ms, count) of full-chain passes and of all other passes since the last call."""
        out = np.zeros(4)
        _lib.check(self.lib.zf_solver_pass_stats(self.handle, C.c_void_p(_lib.ptr(out)), out.size))
        return (out[0], int(out[1])), (out[2], int(out[3]))

    def launch_counts(self):
        """(trial steps issued, shape-specific trial kernels launched for them) since the solver was created."""
        out = np.zeros(2, dtype=np.int64)
        _lib.check(self.lib.zf_solver_launch_counts(self.handle, C.c_void_p(_lib.ptr(out)), out.size))
        return int(out[0]), int(out[1])

    def runahead_counts(self):
        """(run-ahead passes launched, those of them launched while their predecessor was still in flight) since the
        solver was created (zf_runahead_kernel: consecutive full chains on two streams)."""
        out = np.zeros(6, dtype=np.int64)
        _lib.check(self.lib.zf_solver_launch_counts(self.handle, C.c_void_p(_lib.ptr(out)), out.size))
        return int(out[4]), int(out[5])

    def ahead_report(self):
        """What the passes that ran ahead of their predecessor's decision report, as of the last poll: a dict with
        ``runahead`` (run-ahead passes launched), ``runahead_overlapped`` (of them behind a pass still in flight),
        ``timeouts`` (waits that gave up: the device did not hold two passes of this solver at once), ``void`` (run-ahead
        passes that did not count), ``runahead_off`` (the solver stopped launching them after a wait gave up),
        ``ahead`` / ``ahead_void`` (passes ahead at kernel granularity: sharded solves through the library's
        communicator)."""
        out = np.zeros(11, dtype=np.int64)
        _lib.check(self.lib.zf_solver_launch_counts(self.handle, C.c_void_p(_lib.ptr(out)), out.size))
        return dict(runahead=int(out[4]), runahead_overlapped=int(out[5]), timeouts=int(out[6]), void=int(out[7]),
                    ahead=int(out[8]), ahead_void=int(out[9]), runahead_off=bool(out[10]))

    def pass_stats_ex(self):
        """pass_stats() plus (fresh trials, replayed iterations) the other passes carried in total."""
        out = np.zeros(6)
        _lib.check(self.lib.zf_solver_pass_stats_ex(self.handle, C.c_void_p(_lib.ptr(out)), out.size))
        return (out[0], int(out[1])), (out[2], int(out[3])), (int(out[4]), int(out[5]))

    def pass_records(self, cap=65536):
        """[(lagging iterations, fresh trials, 0, ms)] of every timed launch that ran a pass since the last call (timing on;
        passes that ran ahead and turned out void are left out)."""
        out = np.zeros(2 * cap)
        cnt = C.c_int64(0)
        _lib.check(self.lib.zf_solver_pass_records(self.handle, C.c_void_p(_lib.ptr(out)), cap, C.byref(cnt)))
        rec = []
        for k in range(cnt.value):
            shape = int(out[2 * k])
            rec.append(((shape >> 5) & 31, shape & 31, (shape >> 10) & 63, float(out[2 * k + 1])))
        return rec

    def exchange_stats(self):
        """(mean ms, count) of the per-pass pack exchanges the library issued since the last call (timing on)."""
        out = np.zeros(2)
        _lib.check(self.lib.zf_solver_exchange_stats(self.handle, C.c_void_p(_lib.ptr(out)), out.size))
        return out[0], int(out[1])

    def close(self):
        if getattr(self, "handle", None):
            self.lib.zf_solver_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
