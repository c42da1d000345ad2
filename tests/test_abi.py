"""CPU: the C-ABI shared library loads without a GPU and exports exactly the
symbols include/zfista_hip.h declares; struct mirrors agree; the product fails
loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from zfista_amd import _lib


def _declared():
    src = open(os.path.join(ROOT, "include", "zfista_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zf_[A-Za-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in zfista_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    assert lib.zf_abi_version() == 6


def test_struct_mirrors():
    lib = _lib.load()
    assert lib.zf_sizeof_control() == C.sizeof(_lib.Control) == 424
    assert C.sizeof(_lib.ProblemDesc) == 128
    assert C.sizeof(_lib.Options) == 64
    assert C.sizeof(_lib.CommDesc) == 296


def test_error_reporting_is_c_style():
    lib = _lib.load()
    rc = lib.zf_decide_host(None, 0, None, None)
    assert rc == -2 and b"zf_decide_host" in lib.zf_last_error()
    with pytest.raises(_lib.ZfError):
        _lib.check(rc, "zf_decide_host")


def test_short_output_buffers_are_refused():
    """ABI 4: entry points that write a struct (or a library-defined amount of data) into caller
    memory take the capacity of the buffer and refuse - ZF_ERR_ARG, nothing written - a buffer that
    is too small.  Round 2's documented ctypes stub allocated 176 bytes for a zf_control that had grown
    to 416: zf_solver_poll overran it and the heap blew up tests later.  No GPU needed: the size checks
    come before anything dereferences the solver handle or touches a device."""
    lib = _lib.load()
    assert lib.zf_sizeof_control() > 176
    small = (C.c_byte * 176)()
    packs = np.zeros(_lib.ZF_PACK_LEN)
    trace = np.zeros((_lib.ZF_RING, _lib.ZF_TRACE_COLS))

    def raw(fn, argtypes, *args):
        saved = fn.argtypes
        fn.argtypes = argtypes
        try:
            return fn(*args)
        finally:
            fn.argtypes = saved

    P, I = C.c_void_p, C.c_int64
    # zf_decide_host with a 176-byte control block: refused before it is read or written
    rc = raw(lib.zf_decide_host, [P, I, P, P], C.addressof(small), 176, _lib.ptr(packs), _lib.ptr(trace))
    assert rc == -2 and b"zf_sizeof_control" in lib.zf_last_error()
    assert bytes(small) == bytes(176) and not trace.any()
    # zf_solver_poll: 176 "caller" bytes at the head of a guard region that must stay untouched
    guard = (C.c_byte * 1024)()
    # a non-null "handle": a patterned region larger than any solver object - the checks must come before anything is
    # read or WRITTEN through it (a write would land here instead of in somebody's heap, and is asserted on below)
    dummy = (C.c_ubyte * (1 << 20))()
    C.memset(dummy, 0xA5, len(dummy))
    rc = raw(lib.zf_solver_poll, [P, P, I, P, I], C.addressof(dummy), C.addressof(guard), 176, None, 0)
    assert rc == -2 and b"ctl_bytes" in lib.zf_last_error()
    ctl = _lib.Control()
    rc = raw(lib.zf_solver_poll, [P, P, I, P, I], C.addressof(dummy), C.addressof(ctl), C.sizeof(ctl),
             _lib.ptr(trace), 8 * 100)
    assert rc == -2 and b"trace_bytes" in lib.zf_last_error()
    assert bytes(guard) == bytes(1024) and not trace.any()
    # a saved control block of another size is refused by the restore path as well
    rc = raw(lib.zf_solver_restore, [P, P, P, P, I], C.addressof(dummy), C.addressof(dummy), C.addressof(dummy),
             C.addressof(small), 176)
    assert rc == -2 and b"saved_bytes" in lib.zf_last_error()
    out = np.zeros(6)
    rc = raw(lib.zf_solver_pass_stats_ex, [P, P, I], C.addressof(dummy), _lib.ptr(out), 4)
    assert rc == -2 and not out.any()
    rc = raw(lib.zf_solver_pass_stats, [P, P, I], C.addressof(dummy), _lib.ptr(out), 3)
    assert rc == -2 and not out.any()
    iout = np.zeros(6, dtype=np.int64)
    rc = raw(lib.zf_solver_launch_counts, [P, P, I], C.addressof(dummy), _lib.ptr(iout), 1)
    assert rc == -2 and not iout.any()
    rc = raw(lib.zf_mo_solve_stats, [P, P, I], C.addressof(dummy), _lib.ptr(iout), 5)
    assert rc == -2 and not iout.any()
    # ABI 5: the communicator's self-description, the exchange timing and the per-launch records
    desc = _lib.CommDesc()
    rc = raw(lib.zf_comm_describe, [P, P, I], C.addressof(dummy), C.addressof(desc), C.sizeof(desc) - 8)
    assert rc == -2 and b"out_bytes" in lib.zf_last_error() and bytes(desc) == bytes(C.sizeof(desc))
    rc = raw(lib.zf_solver_exchange_stats, [P, P, I], C.addressof(dummy), _lib.ptr(out), 1)
    assert rc == -2 and not out.any()
    cnt = C.c_int64(7)
    rc = raw(lib.zf_solver_pass_records, [P, P, I, P], C.addressof(dummy), None, 4, C.addressof(cnt))
    assert rc == -2 and cnt.value == 7
    rc = raw(lib.zf_solver_launch_counts, [P, P, I], C.addressof(dummy), _lib.ptr(iout), 0)
    assert rc == -2 and not iout.any()
    assert (np.frombuffer(dummy, dtype=np.uint8) == 0xA5).all(), "an entry point wrote through the handle before checking its arguments"


def test_product_fails_loudly_without_gpu():
    lib = _lib.load()
    n = C.c_int(0)
    lib.zf_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is present")
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.problems import DiagQuadL1

    with pytest.raises(_lib.HipUnavailable):
        DiagQuadL1(np.ones(4), np.zeros(4), 0.1)
    cb = (lambda x: 0.5 * float(x @ x), lambda x: 0.0, lambda x: x, lambda w, x: x)
    with pytest.raises(_lib.HipUnavailable):
        minimize_proximal_gradient(*cb, np.ones(3))


def test_kernels_use_no_scratch_memory(tmp_path):
    """Register-resident by construction: a compiler decision to spill the reduction arrays of
    the fused kernels to scratch once cost 17 KiB of extra HBM writes per workgroup (found with
    rocprofv3 --pmc WRITE_SIZE).  Compiles the device code of every source to assembly (no GPU
    needed) and checks that no kernel has a private segment."""
    import re
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    import glob
    from concurrent.futures import ThreadPoolExecutor

    sources = sorted(glob.glob(os.path.join(_lib.CSRC, "*.hip")))
    assert len(sources) >= 10

    def to_asm(src):
        out = tmp_path / (os.path.basename(src) + ".s")
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                        "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", src, "-o", str(out)],
                       check=True, capture_output=True)
        return out.read_text()

    with ThreadPoolExecutor(max_workers=4) as pool:
        texts = list(pool.map(to_asm, sources))
    seen = ahead = 0
    for text in texts:
        for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
            seen += 1
            name = m.group(1)
            size = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(2)).group(1))
            wide = re.search(r"k_dual_solveILi(\d+)E", name)
            if wide and int(wide.group(1)) >= 4:
                # the device-side dual search for m = 4 .. 8 (round 4): the solver step of the ONE deciding wave - 2^m - 1
                # KKT systems of size m + 1, lane-parallel - does not fit 256 VGPRs beside the resident state; it spills
                # a bounded amount, and never inside the element loops of the evaluations (loop depth >= 2)
                assert size <= 4096, f"{name} uses {size} B of scratch per thread"
                code = re.search(re.escape(name) + r":.*?\.end_amdhsa_kernel", text, re.S).group(0)
                depth = 0
                for line in code.splitlines():
                    lab = re.match(r"^\.LBB\d+_\d+:(.*)", line)
                    if lab:
                        d = re.search(r"Depth=(\d+)", lab.group(1))
                        depth = int(d.group(1)) if d else 0
                    assert not ("scratch_" in line and depth >= 2), f"{name}: scratch access inside an element loop: {line.strip()}"
                continue
            assert "zf_persist_kernel" not in name, "the multi-pass kernel was withdrawn from the product (tools/archive/)"
            if "k_ds_advance" in name:
                # the one-wave kernel that advances the dual state machine of a sharded device search between two
                # all-gathers (zf_mo_solve_dual_stream): the solver step of m >= 4 spills as in k_dual_solve; it runs
                # once per batch on 64 lanes - no element loop in it at all
                assert size <= 8192, f"{name} uses {size} B of scratch per thread"
                continue
            if "zf_runahead_kernel" not in name:
                assert size == 0, f"{name} uses {size} B of scratch per thread"
                continue
            # The run-ahead chains (full, mid, clipped) are held to two waves per SIMD (256 VGPRs) and allocate what the
            # per-pass kernels of the same chains do: no scratch (round 5: the momentum factors in scalar registers).
            runahead = True
            ahead += 1
            if runahead:
                # Its iterate traffic must be agent-coherent (sc1): it reads what a kernel still running on another XCD has
                # just stored.  The GPU tests catch a dropped modifier (built with nontemporal iterate traffic 10 of the 13
                # tests of tests/test_gpu_runahead.py fail: profiles/r04_runahead_without_sc1.txt); this catches it where no
                # GPU is: the iterate streams of the DMA pipeline load with sc1 (d, c: nt), every 16-byte iterate store is
                # sc1, none is nontemporal.
                body = re.search(re.escape(name) + r":.*?\.end_amdhsa_kernel", text, re.S).group(0)
                dma_sc1 = len(re.findall(r"global_load_lds_dwordx4 [^\n]*\bsc1\b", body))
                dma_nt = len(re.findall(r"global_load_lds_dwordx4 [^\n]*\bnt\b", body))
                dma_all = len(re.findall(r"global_load_lds_dwordx4 ", body))
                # (the once-read streams d, c: nontemporal, or plain in the variant without the nontemporal policy; as many DMA
                #  instructions as the iterate streams with momentum - x_k, x_{k-1} - twice as many without)
                assert dma_sc1 > 0 and dma_nt in (0, dma_all - dma_sc1) and dma_all - dma_sc1 in (dma_sc1, 2 * dma_sc1), (name, dma_sc1, dma_nt, dma_all)
                assert len(re.findall(r"global_store_dwordx4 [^\n]*\bsc1\b", body)) >= 2, name
                assert not re.findall(r"global_store_dwordx4 [^\n]*\bnt\b", body), name
            assert size == 0, f"{name} uses {size} B of scratch per thread"
            code = re.search(re.escape(name) + r":.*?\.end_amdhsa_kernel", text, re.S).group(0)
            depth = 0
            for line in code.splitlines():
                lab = re.match(r"^\.LBB\d+_\d+:(.*)", line)
                if lab:
                    d = re.search(r"Depth=(\d+)", lab.group(1))
                    depth = int(d.group(1)) if d else 0
                elif re.match(r"^\.LBB\d+_\d+:", line) or re.match(r"^; %bb\.", line):
                    depth = 0
                assert not ("scratch_" in line and depth >= (1 if runahead else 2)), f"{name}: scratch access inside a loop: {line.strip()}"
    # (run-ahead kernels: the full chain in 8 variants + 4 of ZF_ACCEPT_RESOLVED solvers; mid chains of 9 .. 15 trials with and
    #  without momentum, 14 + 14)
    assert seen >= 100 and ahead == 40, (seen, ahead)
