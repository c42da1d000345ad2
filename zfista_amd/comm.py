"""RCCL communicator of the library (``zf_comm``, csrc/zf_comm.hip) for a decision vector sharded
over the GPUs of one node - one process per GPU.

``LibComm`` owns one ``ncclComm_t`` created INSIDE libzfista_hip.so.  A device solver that has one
attached (``zf_solver_set_comm``) issues the per-pass exchanges itself, on its own stream, between
the trial and the decide kernel: a multi-rank pass is one C call with no host code per pass, exactly
like the single-GPU case.  The 128-byte unique id travels over whatever host channel is at hand:

* ``LibComm.from_group(group)`` - an initialised ``torch.distributed`` process group (the id is
  broadcast through it once; torch is not involved afterwards);
* ``LibComm(rank, world, unique_id)`` - the caller ships ``LibComm.new_unique_id()`` itself
  (a file, MPI, a socket ...).

Any ``NativeProblem(..., group=comm)`` accepts a ``LibComm`` in place of a process group.
"""
from __future__ import annotations

import ctypes as C

from . import _lib

_CACHE = {}


class LibComm:
    def __init__(self, rank: int, world: int, unique_id: bytes):
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of LibComm.new_unique_id()")
        self.lib = _lib.require_gpu()
        self.rank, self.world = int(rank), int(world)
        buf = (C.c_char * 128).from_buffer_copy(unique_id)
        h = C.c_void_p()
        _lib.check(self.lib.zf_comm_create(C.byref(h), self.rank, self.world, C.cast(buf, C.c_void_p)), "zf_comm_create")
        self.handle = h

    @classmethod
    def local_group(cls, world: int, cap_doubles: int = 1 << 20):
        """``world`` communicators of an in-process group, one per host THREAD playing a rank on this
        one GPU (each thread runs its solver on its own stream).  For tests: the library's multi-rank
        step sequence and buffer layouts without a second device (zf_comm_create_local_group)."""
        lib = _lib.require_gpu()
        arr = (C.c_void_p * world)()
        _lib.check(lib.zf_comm_create_local_group(arr, int(world), int(cap_doubles)), "zf_comm_create_local_group")
        out = []
        for r in range(world):
            c = cls.__new__(cls)
            c.lib, c.rank, c.world, c.handle = lib, r, int(world), C.c_void_p(arr[r])
            out.append(c)
        return out

    @staticmethod
    def new_unique_id() -> bytes:
        lib = _lib.require_gpu()
        buf = (C.c_char * 128)()
        _lib.check(lib.zf_comm_unique_id(C.cast(buf, C.c_void_p)), "zf_comm_unique_id")
        return bytes(buf.raw)

    @classmethod
    def from_group(cls, group=None):
        """One communicator per torch.distributed group and process (cached).  Collective."""
        import torch.distributed as dist

        # keyed on the group OBJECT (kept alive beside its communicator, so its id cannot be reused) and on
        # (rank, world): after destroy_process_group + a new init the default group is another object and the
        # stale communicator of the old world is never handed out
        pg = group if group is not None else dist.group.WORLD
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        key = (id(pg), rank, world)
        hit = _CACHE.get(key)
        if hit is not None and hit[1] is pg:
            return hit[0]   # (None: the group has no library communicator - decided once, by all ranks)
        err, uid = None, None
        if rank == 0:
            try:
                uid = cls.new_unique_id()
            except Exception as exc:   # noqa: BLE001 - the other ranks wait in the broadcast below: tell them
                err = exc
        box = [uid]
        src = dist.get_global_rank(group, 0) if group is not None and group is not dist.group.WORLD else 0
        dist.broadcast_object_list(box, src=src, group=group)
        # ncclCommInitRank is collective; so is the verdict on it: if any rank could not create its communicator
        # (librccl not loadable, an init error) EVERY rank gives its own up and the caller's exchanges go through
        # torch.distributed on all ranks alike - never one rank on one sequence and the rest on the other
        import torch

        comm = None
        if box[0] is not None:
            try:
                comm = cls(rank, world, box[0])
            except Exception as exc:   # noqa: BLE001 - reported below, on every rank
                err = exc
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            if comm is not None:
                comm.close()
                comm = None
            import warnings

            warnings.warn("zfista_amd: the library's RCCL communicator could not be created on every rank"
                          + (f" (this rank: {err})" if err is not None else "")
                          + "; all ranks use torch.distributed all-gathers for the per-pass exchange", RuntimeWarning,
                          stacklevel=2)
        _CACHE[key] = (comm, pg)
        return comm

    def all_gather(self, send, recv, stream=None):
        """recv (world x count, rank-major) <- send of every rank (float64 CUDA tensors)."""
        import torch

        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        _lib.check(self.lib.zf_comm_all_gather(self.handle, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()),
                                               send.numel(), C.c_void_p(st)), "zf_comm_all_gather")

    def describe(self) -> dict:
        """What the communicator says about itself (``zf_comm_describe``): the ranks RCCL reports, the library file the
        RCCL symbols came from, its version, the all-gathers issued so far."""
        d = _lib.CommDesc()
        _lib.check(self.lib.zf_comm_describe(self.handle, C.byref(d), C.sizeof(d)), "zf_comm_describe")
        return {"via": "zf_comm (RCCL inside libzfista_hip.so)" if d.kind == 0 else "zf_comm in-process thread-rank group (no RCCL)",
                "world": int(d.world), "rank": int(d.rank),
                "rank_count_seen": int(d.nccl_count) if d.kind == 0 else int(d.world),
                "rccl_user_rank": int(d.nccl_user_rank), "rccl_device": int(d.nccl_device),
                "rccl_version": int(d.rccl_version), "library": d.library.decode(errors="replace"),
                "all_gathers_issued": int(d.all_gathers)}

    def close(self):
        if getattr(self, "handle", None):
            self.lib.zf_comm_destroy(self.handle)
            self.handle = None


def rank_world(group):
    """(rank, world) of a torch.distributed group, a LibComm, or an in-process stand-in (tests)."""
    if group is None:
        return 0, 1
    if isinstance(group, LibComm) or hasattr(group, "all_gather_host"):
        return group.rank, group.world
    import torch.distributed as dist

    return dist.get_rank(group), dist.get_world_size(group)
