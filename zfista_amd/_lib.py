"""ctypes binding of ``csrc/libzfista_hip.so`` (C ABI: ``include/zfista_hip.h``).

The library is the product: there is no CPU fallback.  If the shared object is
missing or no GPU is usable, everything that needs it raises ``HipUnavailable``
with the reason - callers never silently continue on the host.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# ZF_LIB_PATH: load another build of the same sources (kernel-variant experiments, tools/)
LIB_PATH = os.environ.get("ZF_LIB_PATH") or os.path.join(CSRC, "libzfista_hip.so")

ZF_OK = 0
ZF_RUNNING, ZF_CONVERGED, ZF_MAXITER, ZF_BACKTRACK_FAILED = 0, 1, 2, 3
ZF_PROBLEM_DIAG_QUAD_L1, ZF_PROBLEM_LEAST_SQUARES_L1, ZF_PROBLEM_BLUR_HAAR_L1 = 1, 2, 3
ZF_MO_GENERIC, ZF_MO_JOS1, ZF_MO_FDS = 0, 1, 2
ZF_PACK_LEN, ZF_TRACE_COLS, ZF_RING = 8, 8, 1024
ZF_MAX_SUB_ITERS = 16
ZF_MAX_LAG = 2 * ZF_MAX_SUB_ITERS - 2
ZF_PEND_FLUSH = -1
TR_ERR, TR_F, TR_LR, TR_FUN, TR_TRIALS, TR_FX, TR_GX, TR_FY = range(8)
PK_FY, PK_DOT, PK_SS, PK_GX, PK_FX, PK_ERR = range(6)
PK_DF = 7   # ZF_ACCEPT_RESOLVED: f(x+) - f(y), accumulated element by element
ZF_ACCEPT_REFERENCE, ZF_ACCEPT_RESOLVED = 0, 1


class HipUnavailable(RuntimeError):
    """The HIP engine cannot be used (library not built, or no GPU)."""


class ZfError(RuntimeError):
    """A libzfista_hip call returned a negative status."""


class Control(C.Structure):
    """Mirror of ``zf_control`` (include/zfista_hip.h)."""

    _fields_ = [
        ("lr", C.c_double), ("F_old", C.c_double), ("f_x", C.c_double), ("g_x", C.c_double),
        ("err", C.c_double), ("fun", C.c_double), ("tol", C.c_double),
        ("tol_internal", C.c_double), ("decay_rate", C.c_double), ("f_y", C.c_double),
        ("nit", C.c_int64), ("max_iter", C.c_int64), ("trial", C.c_int64),
        ("max_backtrack", C.c_int64), ("total_trials", C.c_int64),
        ("status", C.c_int32), ("cur", C.c_int32), ("nesterov", C.c_int32),
        ("deprecated", C.c_int32), ("need_grad", C.c_int32), ("world", C.c_int32),
        ("accept_mode", C.c_int32), ("reserved0", C.c_int32),
        ("beta_next", C.c_double),
        ("ring_size", C.c_int32), ("sub_iters", C.c_int32), ("prev", C.c_int32),
        ("lag", C.c_int32), ("pend_status", C.c_int32), ("pass_seq", C.c_int32),
        ("lag_lr", C.c_double * ZF_MAX_LAG),
    ]


class ProblemDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("world", C.c_int32), ("rank", C.c_int32), ("row_sharded", C.c_int32),
        ("n", C.c_int64), ("m_rows", C.c_int64),
        ("d", C.c_void_p), ("c", C.c_void_p), ("A", C.c_void_p), ("b", C.c_void_p),
        ("scale", C.c_double), ("lam", C.c_double), ("box_lo", C.c_double), ("box_hi", C.c_double),
        ("op_h", C.c_int64), ("op_w", C.c_int64), ("op_taps", C.c_void_p), ("op_k", C.c_int32), ("op_reserved", C.c_int32),
    ]


class Options(C.Structure):
    _fields_ = [
        ("lr", C.c_double), ("tol", C.c_double), ("tol_internal", C.c_double),
        ("decay_rate", C.c_double), ("max_iter", C.c_int64), ("max_backtrack_iter", C.c_int64),
        ("nesterov", C.c_int32), ("deprecated", C.c_int32), ("sub_iters", C.c_int32), ("accept_mode", C.c_int32),
    ]


class CommDesc(C.Structure):
    """Mirror of ``zf_comm_desc``."""

    _fields_ = [
        ("rank", C.c_int32), ("world", C.c_int32), ("kind", C.c_int32), ("nccl_count", C.c_int32),
        ("nccl_user_rank", C.c_int32), ("nccl_device", C.c_int32), ("rccl_version", C.c_int32), ("reserved", C.c_int32),
        ("all_gathers", C.c_int64), ("library", C.c_char * 256),
    ]


_P = C.c_void_p
_D = C.POINTER(C.c_double)
# name -> (restype, argtypes); every symbol include/zfista_hip.h declares
SIGNATURES = {
    "zf_abi_version": (C.c_int, []),
    "zf_last_error": (C.c_char_p, []),
    "zf_sizeof_control": (C.c_int64, []),
    "zf_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "zf_set_device": (C.c_int, [C.c_int]),
    "zf_malloc": (C.c_int, [C.POINTER(_P), C.c_int64]),
    "zf_free": (C.c_int, [_P]),
    "zf_memcpy_h2d": (C.c_int, [_P, _P, C.c_int64, _P]),
    "zf_memcpy_d2h": (C.c_int, [_P, _P, C.c_int64, _P]),
    "zf_memcpy_d2d": (C.c_int, [_P, _P, C.c_int64, _P]),
    "zf_stream_synchronize": (C.c_int, [_P]),
    "zf_shutdown": (C.c_int, []),
    "zf_comm_unique_id": (C.c_int, [_P]),
    "zf_comm_create": (C.c_int, [C.POINTER(_P), C.c_int32, C.c_int32, _P]),
    "zf_comm_destroy": (C.c_int, [_P]),
    "zf_comm_create_local_group": (C.c_int, [C.POINTER(_P), C.c_int32, C.c_int64]),
    "zf_comm_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "zf_comm_describe": (C.c_int, [_P, C.POINTER(CommDesc), C.c_int64]),
    "zf_comm_all_gather": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "zf_solver_set_comm": (C.c_int, [_P, _P]),
    "zf_solver_enqueue_init_all": (C.c_int, [_P, _P]),
    "zf_decide_host": (C.c_int, [C.POINTER(Control), C.c_int64, _P, _P]),
    "zf_solver_create": (C.c_int, [C.POINTER(_P), C.POINTER(ProblemDesc), C.POINTER(Options), _P]),
    "zf_solver_destroy": (C.c_int, [_P]),
    "zf_solver_enqueue_init": (C.c_int, [_P, _P]),
    "zf_solver_enqueue_init_commit": (C.c_int, [_P]),
    "zf_solver_set_beta": (C.c_int, [_P, C.c_int64, _P, C.c_int64]),
    "zf_solver_enqueue_steps": (C.c_int, [_P, C.c_int64]),
    "zf_solver_autotune": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "zf_solver_flush": (C.c_int, [_P]),
    "zf_solver_enqueue_trial": (C.c_int, [_P]),
    "zf_solver_enqueue_decide": (C.c_int, [_P]),
    "zf_solver_sub_iters": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "zf_solver_set_history": (C.c_int, [_P, _P, C.c_int64, C.c_int64]),
    "zf_solver_set_max_iter": (C.c_int, [_P, C.c_int64]),
    "zf_solver_pack_ptrs": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P)]),
    "zf_solver_set_pack_buffers": (C.c_int, [_P, _P, _P]),
    "zf_solver_svec_ptrs": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P)]),
    "zf_solver_set_svec_buffers": (C.c_int, [_P, _P, _P]),
    "zf_solver_enqueue_trial_finish": (C.c_int, [_P]),
    "zf_solver_enqueue_init_finish": (C.c_int, [_P]),
    "zf_solver_poll": (C.c_int, [_P, C.POINTER(Control), C.c_int64, _P, C.c_int64]),
    "zf_solver_x_dev": (C.c_int, [_P, C.POINTER(_P)]),
    "zf_solver_get_x": (C.c_int, [_P, _P, C.c_int64]),
    "zf_solver_get_x_prev": (C.c_int, [_P, _P, C.c_int64]),
    "zf_solver_restore": (C.c_int, [_P, _P, _P, C.POINTER(Control), C.c_int64]),
    "zf_solver_trial_kernel_ms": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "zf_solver_set_timing": (C.c_int, [_P, C.c_int32]),
    "zf_solver_pass_stats": (C.c_int, [_P, _P, C.c_int64]),
    "zf_solver_pass_stats_ex": (C.c_int, [_P, _P, C.c_int64]),
    "zf_solver_exchange_stats": (C.c_int, [_P, _P, C.c_int64]),
    "zf_solver_pass_records": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "zf_solver_launch_counts": (C.c_int, [_P, _P, C.c_int64]),
    "zf_host_grad_step": (C.c_int, [_P, _P, _P, C.c_double, C.c_int64]),
    "zf_host_model_terms": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "zf_host_momentum": (C.c_int, [_P, _P, _P, C.c_double, C.c_int64]),
    "zf_dev_grad_step": (C.c_int, [_P, _P, _P, C.c_double, C.c_int64, _P]),
    "zf_dev_model_terms": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P]),
    "zf_dev_model_terms_async": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P]),
    "zf_dev_momentum": (C.c_int, [_P, _P, _P, C.c_double, C.c_int64, _P]),
    "zf_dev_mo_combine": (C.c_int, [_P, _P, _P, _P, C.c_double, C.c_int32, C.c_int64, _P, _P]),
    "zf_dev_mo_post_terms": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int64, _P, _P]),
    "zf_eval_diag_l1": (C.c_int, [_P, _P, _P, C.c_double, C.c_int64, _P, _P]),
    "zf_host_prox_l1_box": (C.c_int, [_P, _P, C.c_double, C.c_double, C.c_double, C.c_int64]),
    "zf_host_asum": (C.c_int, [_P, C.c_int64, C.POINTER(C.c_double)]),
    "zf_host_diag_grad": (C.c_int, [_P, _P, _P, _P, C.c_int64]),
    "zf_mo_create": (C.c_int, [C.POINTER(_P), C.c_int32, C.c_int32, C.c_int64, _P, _P, C.c_double, C.c_double, _P]),
    "zf_mo_set_shard": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P]),
    "zf_mo_set_comm": (C.c_int, [_P, _P, C.c_int64, C.c_int64]),
    "zf_mo_exchange_count": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "zf_mo_set_bounds": (C.c_int, [_P, _P, _P]),
    "zf_mo_destroy": (C.c_int, [_P]),
    "zf_mo_set_x0": (C.c_int, [_P, _P]),
    "zf_mo_eval_F": (C.c_int, [_P, C.c_int32, _P, _P]),
    "zf_mo_prepare": (C.c_int, [_P, _P]),
    "zf_mo_set_jac": (C.c_int, [_P, _P]),
    "zf_mo_dual_eval": (C.c_int, [_P, C.c_double, _P, _P]),
    "zf_mo_dual_hessian": (C.c_int, [_P, C.c_double, _P, _P]),
    "zf_mo_solve_dual": (C.c_int, [_P, C.c_double, _P, _P, C.c_int32, _P, C.c_double, C.c_int64, _P,
                                  C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int64)]),
    "zf_mo_solve_dual_stream": (C.c_int, [_P, C.c_double, _P, _P, C.c_int32, _P, C.c_double, C.c_int64, _P,
                                         C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int64)]),
    "zf_mo_solve_dual_device": (C.c_int, [_P, C.c_double, _P, _P, C.c_int32, _P, C.c_double, C.c_int64, _P,
                                         C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int64), C.POINTER(C.c_double), _P, _P, _P]),
    "zf_mo_prepare_async": (C.c_int, [_P]),
    "zf_mo_set_fused": (C.c_int, [_P, C.c_int32]),
    "zf_mo_trial_launch": (C.c_int, [_P, C.c_double, _P, C.c_int32, _P, C.c_double, C.c_int64, C.c_double, C.c_int32,
                                     C.c_int32, _P]),
    "zf_mo_trial_wait": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "zf_mo_uncommit": (C.c_int, [_P]),
    "zf_mo_invalidate_prepare": (C.c_int, [_P, C.c_int32]),
    "zf_mo_debug_force_timeout": (C.c_int, [_P, C.c_int32]),
    "zf_mo_get_f_y": (C.c_int, [_P, _P]),
    "zf_mo_solve_stats": (C.c_int, [_P, _P, C.c_int64]),
    "zf_mo_recover": (C.c_int, [_P, C.c_double, _P, _P]),
    "zf_mo_commit": (C.c_int, [_P, C.c_double, C.c_int32]),
    "zf_mo_get": (C.c_int, [_P, C.c_int32, _P, C.c_int64]),
    "zf_mo_put": (C.c_int, [_P, C.c_int32, _P]),
    "zf_mo_get_jac": (C.c_int, [_P, _P, C.c_int64]),
    "zf_mo_prox_host": (C.c_int, [_P, _P, _P, _P]),
    "zf_mo_post_terms": (C.c_int, [_P, C.c_double, _P, _P, _P]),
    "zf_op_eval": (C.c_int, [_P, C.c_int32, _P, C.c_int64, C.c_int64, C.c_double, _P, C.POINTER(C.c_double), _P]),
    "zf_ls_eval": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_double, _P, C.POINTER(C.c_double), _P]),
}

_lib = None


def build(verbose: bool = False) -> str:
    """Compile the library in-tree with hipcc for gfx950 (cross-compiles on a CPU box)."""
    out = subprocess.run(["make", "-C", CSRC], capture_output=True, text=True)
    if out.returncode != 0:
        raise HipUnavailable(f"building libzfista_hip.so failed:\n{out.stdout}\n{out.stderr}")
    if verbose:
        print(out.stdout)
    return LIB_PATH


def load():
    """Load the shared library and declare every signature.  Raises HipUnavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipUnavailable(
            f"{LIB_PATH} is not built; run `make -C {CSRC}` (or __graft_entry__.build()). "
            "zfista_amd has no CPU fallback."
        )
    # torch bundles its own libamdhip64; import it first so both resolve to ONE
    # HIP runtime (soname libamdhip64.so.7) inside this process.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is plumbing, not a requirement of the library
        pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:
        raise HipUnavailable(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.zf_sizeof_control() != C.sizeof(Control):
        raise HipUnavailable("zf_control layout mismatch between library and Python mirror")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != ZF_OK:
        msg = load().zf_last_error().decode(errors="replace")
        raise ZfError(f"{what or 'libzfista_hip'} failed ({rc}): {msg}")


def require_gpu():
    """The product path: library + at least one GPU, or a loud failure."""
    lib = load()
    n = C.c_int(0)
    rc = lib.zf_device_count(C.byref(n))
    if rc != ZF_OK or n.value < 1:
        raise HipUnavailable(
            "no usable GPU: " + lib.zf_last_error().decode(errors="replace")
            + " - zfista_amd runs its solver arithmetic on MI355X only (no CPU fallback)."
        )
    return lib


# Every ZF_* environment variable product code reads.  Each changes which kernels run, their launch geometry or a
# numerics path - results that were produced under one say so (``OptimizeResult.overrides``, bench.py's
# ``config.overrides``): numerics never change silently with the environment.
ENV_SWITCHES = ("ZF_FIN_KERNEL", "ZF_SPECULATE", "ZF_NT", "ZF_LS_SMALL", "ZF_GEMV_MFMA", "ZF_TILES_PER_WG", "ZF_SUB_ITERS",
                "ZF_COMM", "ZF_MO_COMM", "ZF_MO_LAUNCH_AHEAD", "ZF_MO_SPIN_LIMIT", "ZF_RCCL_LIB", "ZF_DUAL_SOLVER",
                "ZF_FORCE_SPLIT", "ZF_LIB_PATH", "ZF_MID_CHAINS", "ZF_BENCH_BACKEND",
                "ZF_RUNAHEAD", "ZF_RUNAHEAD_SPIN_LIMIT", "ZF_SHORT_VIA_GENERAL", "ZF_PASS_SEQ_START", "ZF_AHEAD",
                "ZF_AHEAD_UNSHARDED", "ZF_ACCEPT", "ZF_OP_SEPARABLE", "ZF_OP_TY", "ZF_OP_XCD_BANDS", "ZF_OP_FUSE_PROX", "ZF_OP_PERSIST", "ZF_RUNAHEAD_SHARDED", "ZF_RAS_LAST_INLINE")


def env_overrides() -> dict:
    """{name: value} of every ZF_* switch set in this process's environment (empty: the defaults ran)."""
    return {k: os.environ[k] for k in ENV_SWITCHES if k in os.environ}


def ptr(a: np.ndarray) -> int:
    return a.ctypes.data
