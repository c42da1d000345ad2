#!/usr/bin/env python3
"""bench.py - FISTA iterations/sec of the fused grad+prox+momentum step (P-diag).

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1
the driver launches one rank per GPU with torch.distributed.run.  A "step" is one
FISTA iteration (one accepted line-search trial) over this rank's 10^8-element
shard of the decision vector.  Inputs are generated on the device and are resident in
HBM before the timed region; exactly W iterations are run untimed, then exactly K timed
(max_iter is raised from W to W + K; the device stops on it).  One launch ("pass") of the
fused kernel carries a chain of up to S = 8 iterations (temporal blocking), so
K iterations take about K / S passes.  Rank 0 prints ONE JSON line.

value = (N * K) / t : iterations per second in units of one 10^8-element shard.
At N = 1 that is exactly BASELINE.json's metric (FISTA it/s at n = 10^8); for
N > 1 (weak scaling: n = N x 10^8) the full-problem rate K / t is reported beside
it as ``config.iters_per_sec_full_problem``.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PER_GPU = 10**8
ALG_BYTES_PER_ELEM = 40          # per ITERATION: read x_k, x_{k-1}, d, c ; write x+   
PASS_BYTES_PER_ELEM = 48         # per PASS with S > 1: the same four reads, two iterates written
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
LR, LAM = 0.45, 0.1


def make_inputs(n, seed, device):
    import torch

    gen = torch.Generator(device=device).manual_seed(seed)
    d = torch.rand(n, dtype=torch.float64, device=device, generator=gen) * 1.5 + 0.5   # U[0.5, 2]
    c = torch.randn(n, dtype=torch.float64, device=device, generator=gen)
    return d, c


def measured_traffic(n, sub_iters):
    """HBM bytes per trial-kernel launch from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate
    runs of this same command, gfx950 corrections applied).  None when no profile
    for this n and chain length is committed."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as fh:
            prof = json.load(fh)
    except OSError:
        return None
    if prof.get("n") != n or prof.get("sub_iters", 1) != sub_iters:
        return None
    return prof["hbm_bytes_per_launch"]


def cpu_baseline(d, c, sample_n=10**7, iters=30):
    """The oracle (NumPy restatement of the reference path) on the host cores,
    on a bounded sample of the same workload.  Reported, never a target."""
    from oracle import cpu_ref, problems_ref as P

    sample_n = min(sample_n, d.numel())
    ds = d[:sample_n].cpu().numpy()
    cs = c[:sample_n].cpu().numpy()
    ref = P.DiagQuadL1Ref(ds, cs, LAM)
    x0 = np.zeros(sample_n)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.perf_counter()
        res = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), x0, lr=LR, nesterov=True, tol=0.0,
                                                 max_iter=iters)
        dt = time.perf_counter() - t0
    assert res.nit == iters
    its = iters / dt
    return {
        "value": its * sample_n / N_PER_GPU,
        "unit": "iterations/s (n=1e8)",
        "cores": 1,
        "kind": "port",
        "sample": f"first {sample_n:.0e} of 1e8 elements, {iters} FISTA iterations in {dt:.1f} s "
                  f"({its:.3f} it/s at n={sample_n:.0e}), scaled by n to 1e8; elementwise NumPy is "
                  f"single-threaded (host has {os.cpu_count()} cores)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n", "--elements", dest="n", type=int, default=N_PER_GPU, help="elements per GPU (default 1e8)")
    ap.add_argument("--total-n", type=int, default=0,
                    help="strong scaling: total length of x, split evenly over the GPUs (e.g. 800000000, "
                         "BASELINE cfg5); overrides --n")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket every trial kernel with HIP events (roofline becomes null); "
                         "shows what the per-kernel timing itself costs")
    args = ap.parse_args()

    # Only the JSON line may reach stdout: libraries (RCCL prints "Hostname : ..." banners at
    # communicator creation) are pointed at stderr for the whole run.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # ZF_BENCH_BACKEND=gloo rehearses the multi-process path with all ranks on ONE GPU (RCCL needs
    # a device per rank); the driver's runs use nccl = RCCL, one GPU per rank
    backend = os.environ.get("ZF_BENCH_BACKEND", "nccl")
    device = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    group = None
    use_pg = world > 1 or (os.environ.get("ZF_FORCE_SPLIT") == "1" and "RANK" in os.environ)
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD

    from zfista_amd import _lib
    from zfista_amd.problems import DiagQuadL1
    from zfista_amd.proximal_gradient import NativeRun

    n = args.n
    if args.total_n:
        n = args.total_n * (rank + 1) // world - args.total_n * rank // world   # this rank's block
    K, W = args.steps, args.warmup
    d, c = make_inputs(n, seed=1 + rank, device="cuda")
    prob = DiagQuadL1(d, c, LAM, group=group)
    opts = dict(lr=LR, tol=0.0, tol_internal=1e-12, max_iter=max(W, 1), max_backtrack_iter=100, decay_rate=0.5,
                nesterov=True, nesterov_ratio=(0, 0.25), deprecated=False)
    run = NativeRun(prob, torch.zeros(n, dtype=torch.float64, device="cuda"), opts, timing=not args.no_kernel_events)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    S = run.sub_iters
    while W > 0 and run.status == _lib.ZF_RUNNING:   # warm-up: exactly W iterations, then MAXITER
        run.advance((W - run.nit_seen + S - 1) // S)
    run.solver.trial_kernel_ms()          # reset the event window after warm-up
    nit0 = run.nit_seen
    run.set_max_iter(nit0 + K)
    sync_all()
    t0 = time.perf_counter()
    while run.status == _lib.ZF_RUNNING:   # a broken chain (rejected trial, lr halves :305) costs extra passes
        # exactly the passes the remaining iterations need if no chain breaks (so that every
        # launch counted below is a real pass); broken chains cost further rounds
        run.enqueue_only((K - (run.nit_seen - nit0) + S - 1) // S)
        run.collect()
    sync_all()
    dt = time.perf_counter() - t0
    assert run.status == _lib.ZF_MAXITER and run.nit_seen - nit0 == K, \
        f"expected {K} accepted iterations, got {run.nit_seen - nit0} (status {run.status})"
    ker_ms, ker_n = run.solver.trial_kernel_ms()
    if args.no_kernel_events:
        ker_ms, ker_n = float("nan"), None
    iters_per_pass = None if ker_n is None else K / ker_n

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        # roofline.achieved, as SURVEY 8d defines it: ALGORITHMIC bytes of the iterations a launch
        # completes (40 B per element and iteration) over the launch duration.  A pass that chains
        # S iterations moves fewer bytes than that, so the figure can exceed the HBM peak; the
        # bytes the pass really moves (48 B per element, PMC-checked) are reported beside it.
        if iters_per_pass is None:
            iters_per_pass = float("nan")
        alg_bytes = ALG_BYTES_PER_ELEM * n * iters_per_pass
        achieved = alg_bytes / (ker_ms * 1e-3) / 1e9
        pass_bytes = (PASS_BYTES_PER_ELEM if S > 1 else ALG_BYTES_PER_ELEM) * n
        line = {
            "metric": f"fista_iterations_per_sec_n{args.total_n:.0e}_total" if args.total_n
                      else "fista_iterations_per_sec_n1e8_per_gpu_shard",
            # weak: iterations/s in units of one 1e8-element shard, summed over the GPUs;
            # strong (--total-n): iterations/s of the one fixed-size problem
            "value": (K / dt) if args.total_n else world * K / dt * (n / N_PER_GPU),
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": dt / K * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.total_n else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "P-diag l1-regularised diagonal quadratic, fused grad+soft-threshold+momentum, "
                            f"n={n:.0e} per GPU, FISTA (a,b)=(0,0.25), lr=0.45, lam=0.1",
                "n_per_gpu": n,
                "n_total": args.total_n if args.total_n else n * world,
                "iters_per_sec_full_problem": K / dt,
                "temporal_blocking_chain": S,
                "passes": ker_n,
                "iterations_per_pass": iters_per_pass,
                "tiles_per_workgroup_autotuned": getattr(run.solver, "tiles_per_wg", None),
                "parallelism": f"x sharded over {world} GPU(s); per-pass scalar pack all-gather"
                               if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(n, S),
                "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                  "separate passes, bytes per launch)",
                "kernel": f"zf_trial_kernel<grad inline, nesterov, nt, S={S}>",
                "kernel_avg_ms": ker_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
                "algorithmic_bytes_note": "40 B x n x iterations completed per launch (SURVEY 8d); "
                                          "frac > 1 = the chain avoids HBM traffic the one-iteration pass needs",
                "hbm_bytes_per_launch_model": pass_bytes,
                "hbm_achieved": pass_bytes / (ker_ms * 1e-3) / 1e9,
                "hbm_frac": pass_bytes / (ker_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(d, c)
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    run.solver.close()
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
