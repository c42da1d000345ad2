#!/usr/bin/env python3
"""bench.py - FISTA iterations/sec of the fused grad+prox+momentum step (P-diag).

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1
the driver launches one rank per GPU with torch.distributed.run.  A "step" is one
FISTA iteration (one accepted line-search trial) over this rank's 10^8-element
shard of the decision vector.  Inputs are generated on the device and are resident in
HBM before the timed region.

One timed BLOCK = a fresh solve from x0 = 0: exactly W iterations untimed, then exactly K
iterations timed between barrier + synchronize on both sides (max_iter is raised from W to
W + K on the live solve; the device stops on it).  A block lasts a few milliseconds, in which
the GPU still runs at its boost clock; a long solve does not (the kernel is power-limited,
DESIGN.md 4.1).  So the block is REPEATED - same workload, same iterations 1..W+K, back to back -
until at least ``--min-seconds`` (default 0.6 s) of timed work has run (at most ``--max-blocks``),
and ``value`` / ``ms_per_step`` are the MEDIAN block: the sustained rate on a warm device.  The
first and last block are reported beside it.

One launch ("pass") of the fused kernel carries a chain of up to S = 16 iterations (temporal
blocking), so K iterations take about K / S passes.  Rank 0 prints ONE JSON line.

value = (N * K) / t : iterations per second in units of one 10^8-element shard.
At N = 1 that is exactly BASELINE.json's metric (FISTA it/s at n = 10^8); for
N > 1 (weak scaling: n = N x 10^8) the full-problem rate K / t is reported beside
it as ``config.iters_per_sec_full_problem``.

roofline (dominant kernel: zf_trial_kernel, full-chain passes only, HIP events on the solver's
stream), two roofs side by side: ``hbm`` = HBM bytes one pass MOVES (48 B x n for a chain: four
streams read, two iterates written; 40 B x n for S = 1; checked against the PMC counters in
profiles/) / mean duration / 8 TB/s, and ``fp64_valu`` = fp64 VALU issue slots the chain needs
(instructions per element and trial from the ISA, profiles/r02_isa_mix_*.json) / slots available
in that duration at 2.4 GHz.  ``bound`` names the larger fraction and ``achieved`` / ``peak`` /
``unit`` / ``frac`` repeat that roof (<= 1); the chain of 16 is bound by the fp64 vector pipe, the
chains of 8 and shorter by HBM.  SURVEY 8d's per-iteration figure (40 B per
element and ITERATION) divided by the same duration is ``equivalent_one_iteration_GBps``: what a
one-iteration-per-pass kernel would have to sustain to match - it exceeds the HBM peak because
the chain avoids that traffic, it is not a bandwidth.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
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
# fp64 vector roof: 256 CUs x 4 SIMDs x 16 lanes per cycle (a wave64 fp64 instruction issues in 4
# cycles; 78.6 TFLOP/s = this x 2 flops per FMA x 2.4 GHz)
FP64_LANES_PER_CYCLE = 256 * 4 * 16
CLOCK_GHZ = 2.4
LR, LAM = 0.45, 0.1
PMC_PROFILE = "r02_s{S}_pmc_traffic.json"          # round-2 sets (chains of 1 / 8; the full chain of 16 before round 3)
ISA_PROFILE = "r02_isa_mix_trial_kernel_s{S}.json"
# round 3: one PMC set per FLAG SET of this file, summarised per shape-specific kernel (tools/profile_r3.sh):
# the line names the kernel that dominates its timed region and reads THAT kernel's counters
# round 4: re-collected with the round's kernels (tools/profile_r4.sh); mid chains are keyed by their length
# (round 5: "general" names the general body - part2, it read part1's counters, the short bodies'; mid chains are looked
#  up by their length and report no counters when that length was not profiled)
PMC_R3 = {"full": ("r05_pmc_defaults_k100_w10.json", "part0"), "mid": ("r05_pmc_driver_k20_w5.json", "part3_L{len}"),
          "general": ("r05_pmc_defaults_k100_w10.json", "part2")}
MID_MIN, MID_MAX = 9, 15   # trials of the branch-free mid chains (csrc/zf_kernels_step.h: ZF_MID_MIN .. ZF_MID_MAX), one kernel per length


def make_inputs(n, seed, device):
    import torch

    gen = torch.Generator(device=device).manual_seed(seed)
    d = torch.rand(n, dtype=torch.float64, device=device, generator=gen) * 1.5 + 0.5   # U[0.5, 2]
    c = torch.randn(n, dtype=torch.float64, device=device, generator=gen)
    return d, c


def _profile(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            return json.load(fh)
    except OSError:
        return None


def measured_traffic(n, sub_iters):
    """HBM bytes per full-chain launch from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE collected in separate runs of this same command, gfx950 corrections applied;
    tools/profile_bench.sh).  None when no profile for this n and chain length is committed."""
    prof = _profile(PMC_PROFILE.format(S=sub_iters))
    if not prof or prof.get("n") != n or prof.get("sub_iters", 1) != sub_iters:
        return None
    return prof["hbm_bytes_per_launch"]


def measured_clock_ghz(n, sub_iters):
    """Median engine clock under the full-chain kernel from the committed GRBM_GUI_ACTIVE pass of the
    same command (tools/profile_bench.sh step 5), or None."""
    prof = _profile(PMC_PROFILE.format(S=sub_iters))
    if not prof or prof.get("n") != n or "engine_clock_GHz" not in prof:
        return None
    return prof["engine_clock_GHz"]["median"]


def kernel_profile(n, sub_iters, kind, mid_len=0):
    """The committed PMC summary of the kernel this line is about - the full chain (PART 0) under the default
    flags, the branch-free 10-trial chain (PART 3) or the general 16-trial body (PART 2) under the driver's - or
    None (another n / chain length)."""
    if sub_iters != 16:
        return None
    name, part = PMC_R3[kind]
    part = part.format(len=mid_len)
    prof = _profile(name)
    if not prof or prof.get("n") != n or part not in prof.get("kernels", {}):
        return None
    k = dict(prof["kernels"][part])
    k["file"], k["part"] = "profiles/" + name, part
    k["trials_per_pass"] = prof.get("trials_per_pass", {}).get(part)
    return k


def valu_per_element_trial(sub_iters):
    prof = _profile(ISA_PROFILE.format(S=sub_iters))
    return None if not prof else prof.get("valu_per_element_trial")


def cpu_baseline(d, c, iters=3):
    """The oracle (NumPy restatement of the reference path) on the host cores, on the FULL
    vectors of this workload (n = 1e8, 3 iterations).  Reported, never a target."""
    from oracle import cpu_ref, problems_ref as P

    n = d.numel()
    ds, cs = d.cpu().numpy(), c.cpu().numpy()
    ref = P.DiagQuadL1Ref(ds, cs, LAM)
    x0 = np.zeros(n)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.perf_counter()
        res = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), x0, lr=LR, nesterov=True, tol=0.0,
                                                 max_iter=iters)
        dt = time.perf_counter() - t0
    assert res.nit == iters
    return {
        "value": iters / dt * (n / N_PER_GPU),
        "unit": "iterations/s (n=1e8)",
        "cores": 1,
        "kind": "port",
        "sample": f"the same d, c (all {n:.0e} elements), first {iters} FISTA iterations from x0 = 0 in {dt:.1f} s; "
                  f"elementwise NumPy is single-threaded (host has {os.cpu_count()} cores)",
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
    ap.add_argument("--min-seconds", type=float, default=0.6,
                    help="repeat the W + K block until this much timed work has run (sustained clocks)")
    ap.add_argument("--max-blocks", type=int, default=400)
    ap.add_argument("--sub-iters", type=int, default=0, help="chain length S: 1 / 2 / 4 / 8 / 16 (0 = library default)")
    ap.add_argument("--libcomm", action="store_true",
                    help="N = 1 only: run the sharded step sequence over a 1-rank RCCL communicator created "
                         "inside the library (zf_comm): what the per-pass exchange costs without a second GPU")
    ap.add_argument("--acceptance", choices=("reference", "resolved"), default="reference",
                    help="how the sufficient-decrease test is evaluated (zfista_amd.minimize_proximal_gradient's keyword): "
                         "'reference' = zfista/proximal_gradient.py:303 as written (the metric's configuration); 'resolved' = the "
                         "same inequality with f(x+) - f(y) accumulated element by element - no rounding-noise rejections")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket every trial kernel with HIP events (roofline becomes null); "
                         "shows what the per-kernel timing itself costs")
    ap.add_argument("--no-regimes", action="store_true",
                    help="skip the brief second measurement of the other regime (config.iterations_per_sec_by_regime)")
    ap.add_argument("--thread-ranks", type=int, default=0,
                    help="DRY RUN of the N > 1 code path on ONE GPU: this many rank THREADS in this process, one stream "
                         "and one solver each, x sharded over them through the library's own communicator "
                         "(zf_comm_create_local_group - the all-gather RCCL serves on a real node).  Same step "
                         "sequence, same JSON line (n_gpus stays 1, config.thread_ranks says how many); use a "
                         "smaller --n (8 x 1e7 fits the time of a test).  Not a scaling measurement.")
    args = ap.parse_args()

    # Only the JSON line may reach stdout: libraries (RCCL prints "Hostname : ..." banners at
    # communicator creation) are pointed at stderr for the whole run.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    if args.thread_ranks > 1:
        return thread_rank_main(args, real_stdout)
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

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(dt):
        if world == 1:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if args.libcomm and world == 1:
        from zfista_amd.comm import LibComm

        group = LibComm(0, 1, LibComm.new_unique_id())
    line = rank_body(args, rank, world, group, barrier, max_over_ranks)
    if rank == 0:
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


def thread_rank_main(args, real_stdout):
    """--thread-ranks N: the N > 1 code path of this file with N rank threads on one GPU (see the option)."""
    import threading

    import torch

    from zfista_amd.comm import LibComm

    world = args.thread_ranks
    torch.cuda.set_device(0)
    comms = LibComm.local_group(world, cap_doubles=4096)
    bar = threading.Barrier(world)
    slots = [0.0] * world
    out, errs = [None] * world, []

    def barrier():
        bar.wait()

    def make_max(rank):
        def max_over_ranks(dt):
            slots[rank] = dt
            bar.wait()
            m = max(slots)
            bar.wait()
            return m
        return max_over_ranks

    def rank_main(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                out[r] = rank_body(args, r, world, comms[r], barrier, make_max(r), thread_ranks=world)
        except Exception as exc:   # pragma: no cover - reported below
            errs.append(exc)
            bar.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errs:
        raise errs[0]
    os.write(real_stdout, (json.dumps(out[0]) + "\n").encode())
    for c_ in comms:
        c_.close()


def rank_body(args, rank, world, group, barrier, max_over_ranks, thread_ranks=0):
    """What one rank does: warm-up, the timed blocks, and (rank 0) the JSON line.  `barrier` /
    `max_over_ranks` are the two rank-to-rank operations of the contract (torch.distributed between
    processes, a thread barrier in the dry run); the per-pass exchange is the library's."""
    import torch

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
                nesterov=True, nesterov_ratio=(0, 0.25), deprecated=False, sub_iters=args.sub_iters,
                acceptance=args.acceptance)
    x0 = torch.zeros(n, dtype=torch.float64, device="cuda")
    timing = not args.no_kernel_events

    def sync_all():
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()

    def measure(K, W, min_seconds, max_blocks, acceptance=None):
        """Blocks of W untimed + K timed iterations (a fresh solve each) until min_seconds are timed."""
        o = dict(opts, max_iter=max(W, 1))
        if acceptance:
            o["acceptance"] = acceptance
        run = NativeRun(prob, x0, o, timing=timing)
        S = run.sub_iters
        m = dict(blocks=[], timed=0.0, full_ms=0.0, full_n=0, part_ms=0.0, part_n=0, part_fresh=0, part_lag=0, S=S,
                 tiles=None)
        while True:
            while W > 0 and run.status == _lib.ZF_RUNNING:   # warm-up: exactly W iterations, then MAXITER
                run.advance((W - run.nit_seen + S - 1) // S)
            if timing:
                run.solver.pass_stats()                      # reset the event window after warm-up
            nit0 = run.nit_seen
            run.set_max_iter(nit0 + K)
            sync_all()
            t0 = time.perf_counter()
            trials0 = int(run.solver.ctl.total_trials) - run.nit_seen
            while run.status == _lib.ZF_RUNNING:
                # the passes the remaining iterations need if no chain breaks; a rejected trial (lr halves, :305)
                # costs further passes: once this block has seen rejections, two spare passes ride along with every
                # round (a pass enqueued behind the end of the solve exits at once) instead of a poll round trip each
                spare = 2 if int(run.solver.ctl.total_trials) - run.nit_seen > trials0 else 0
                run.enqueue_only((K - (run.nit_seen - nit0) + S - 1) // S + spare)
                run.collect()
            sync_all()
            dt = time.perf_counter() - t0
            assert run.status == _lib.ZF_MAXITER and run.nit_seen - nit0 == K, \
                f"expected {K} accepted iterations, got {run.nit_seen - nit0} (status {run.status})"
            if timing:
                (fm, fn), (pm, pn), (pf, pl) = run.solver.pass_stats_ex()
                m["full_ms"] += fm * fn
                m["full_n"] += fn
                m["part_ms"] += pm * pn
                m["part_n"] += pn
                m["part_fresh"] += pf
                m["part_lag"] += pl
            rep_ = run.solver.ahead_report()
            m["ra_passes"] = m.get("ra_passes", 0) + rep_["runahead"]
            m["ra_ahead"] = m.get("ra_ahead", 0) + rep_["runahead_overlapped"]
            for key in ("timeouts", "void", "ahead", "ahead_void"):
                m["rep_" + key] = m.get("rep_" + key, 0) + rep_[key]
            m["ra_off"] = m.get("ra_off", False) or rep_["runahead_off"]
            note_comm(run)
            dt = max_over_ranks(dt)
            m["blocks"].append(dt)
            m["timed"] += dt
            m["tiles"] = getattr(run.solver, "tiles_per_wg", None)
            run.solver.close()
            if m["timed"] >= min_seconds or len(m["blocks"]) >= max_blocks:   # (dt is the max over ranks: same decision everywhere)
                break
            run = NativeRun(prob, x0, o, timing=timing)   # the same workload again, on a warmer device
        return m

    comm_seen = {}

    def note_comm(run):
        """What carried the per-pass exchange of this run (rank 0 reports it: config.rccl)."""
        sv = run.solver
        if getattr(sv, "comm", None) is not None:
            comm_seen.update(sv.comm.describe())
            if timing:
                ms, cnt = sv.exchange_stats()
                comm_seen["exchange_ms_per_pass"] = (comm_seen.get("exchange_ms_per_pass", 0.0) * comm_seen.get("exchanges_timed", 0)
                                                     + ms * cnt) / max(comm_seen.get("exchanges_timed", 0) + cnt, 1)
                comm_seen["exchanges_timed"] = comm_seen.get("exchanges_timed", 0) + cnt
        elif getattr(sv, "split", False):
            comm_seen.update({"via": "torch.distributed fallback (host-driven trial / all-gather / decide per pass)",
                              "world": sv.world, "rank": sv.rank, "rank_count_seen": sv.world, "library": None})

    M = measure(K, W, args.min_seconds, args.max_blocks)
    S, blocks, timed, tiles = M["S"], M["blocks"], M["timed"], M["tiles"]
    full_ms, full_n, part_ms, part_n = M["full_ms"], M["full_n"], M["part_ms"], M["part_n"]
    part_fresh, part_lag = M["part_fresh"], M["part_lag"]
    dt = statistics.median(blocks)
    passes = full_n + part_n
    # The two regimes of this workload side by side, whatever the flags: 20 iterations from iteration 5 stay clear of
    # the resolution limit of the acceptance test (no trial is rejected); 100 iterations from iteration 10 cross it
    # at iteration ~89 and pay for two or three rounding-noise rejections.  The other one is measured
    # briefly after the main measurement (every rank takes part: the exchanges are collective).
    regimes = {}
    # (round 4: K = 24 / 28 / 30 beside K = 20 - tails of 12 + 12, 14 + 14, 15 + 15 trials: every tail length has a
    #  branch-free kernel of its own now, the line no longer has one privileged block length)
    for (k2, w2), tag in (((20, 5), "clean_regime_K20_W5"), ((24, 5), "clean_regime_K24_W5"), ((28, 5), "clean_regime_K28_W5"),
                          ((30, 5), "clean_regime_K30_W5"), ((100, 10), "across_the_noise_floor_K100_W10")):
        if (k2, w2) == (K, W):
            regimes[tag] = (world * K / dt * (n / N_PER_GPU)) if not args.total_n else K / dt
        elif not args.no_regimes:
            m2 = measure(k2, w2, 0.2, 60)
            d2 = statistics.median(m2["blocks"])
            regimes[tag] = (world * k2 / d2 * (n / N_PER_GPU)) if not args.total_n else k2 / d2

    # ... and the noisy regime once more with the acceptance test resolved below ulp(F) (acceptance="resolved": the same
    # inequality, f(x+) - f(y) accumulated element by element - no trial of this workload is rejected, 7 passes instead of 9)
    if not args.no_regimes and args.acceptance == "reference":
        m3 = measure(100, 10, 0.2, 60, acceptance="resolved")
        d3 = statistics.median(m3["blocks"])
        regimes["across_the_noise_floor_K100_W10_acceptance_resolved"] = (world * 100 / d3 * (n / N_PER_GPU)) if not args.total_n else 100 / d3
    line = None
    if rank == 0:
        n_gpus = 1 if thread_ranks else world
        line = {
            "metric": f"fista_iterations_per_sec_n{args.total_n:.0e}_total" if args.total_n
                      else "fista_iterations_per_sec_n1e8_per_gpu_shard",
            # weak: iterations/s in units of one 1e8-element shard, summed over the GPUs;
            # strong (--total-n): iterations/s of the one fixed-size problem
            "value": (K / dt) if args.total_n else world * K / dt * (n / N_PER_GPU),
            "unit": "iterations/s",
            "n_gpus": n_gpus,
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
                "timing": f"median of {len(blocks)} back-to-back blocks of W={W} untimed + K={K} timed iterations "
                          f"(fresh solve each), {timed:.2f} s timed in total",
                "blocks": len(blocks),
                "ms_per_step_first_block": blocks[0] / K * 1e3,
                "ms_per_step_median_block": dt / K * 1e3,
                "ms_per_step_last_block": blocks[-1] / K * 1e3,
                "ms_per_step_min_block": min(blocks) / K * 1e3,
                "temporal_blocking_chain": S,
                "passes_per_block": passes / len(blocks) if timing else None,
                "full_chain_passes": full_n if timing else None,
                "other_passes": part_n if timing else None,
                "tiles_per_workgroup": tiles,
                "iterations_per_sec_by_regime": regimes,
                "thread_ranks": thread_ranks or None,
                # what carried the per-pass exchange, as the communicator itself reports it (zf_comm_describe: RCCL's own
                # ncclCommCount / ncclCommUserRank / ncclCommCuDevice, the librccl file, its version) and what one exchange
                # cost on rank 0's stream (HIP events: my packs ready -> gathered packs here); None on one GPU without one
                "rccl": comm_seen or None,
                "overrides": _lib.env_overrides(),
                # consecutive full chains of a one-round grid (n <= ~2.5e7) on two streams, pass p + 1 running while pass p
                # is finalised; at the headline size the grid is four rounds deep and none are launched
                "runahead": {"passes": M.get("ra_passes", 0), "launched_behind_a_pass_in_flight": M.get("ra_ahead", 0),
                             # what the device reported (zf_solver_launch_counts [6..10]): waits that gave up - the device
                             # did not hold two passes at once - and passes that turned out void
                             "waits_that_gave_up": M.get("rep_timeouts", 0), "void_passes": M.get("rep_void", 0),
                             "switched_off": M.get("ra_off", False)},
                # passes AHEAD of their predecessor's decision at kernel granularity (sharded solves through the library's
                # communicator): trial kernels back to back, finalisation / all-gather / decide on a second stream
                "passes_ahead": {"launched": M.get("rep_ahead", 0), "void": M.get("rep_ahead_void", 0)},
                "acceptance": args.acceptance,
                "parallelism": (f"DRY RUN: x sharded over {world} rank threads on ONE GPU (in-process communicator group of the "
                                "library); the N > 1 step sequence, not a scaling measurement") if thread_ranks else
                               f"x sharded over {world} GPU(s); per-pass scalar pack all-gather (RCCL inside the library)"
                               if world > 1 else ("single GPU, 1-rank RCCL all-gather per pass" if args.libcomm
                                                  else "single GPU"),
            },
        }
        if timing and (full_n or part_n):
            # the dominant kernel: the full-chain passes (every pass when S = 1) when they hold most of the
            # kernel time, else the other passes - K < 2 S iterations are shared by two passes of about K / 2
            # trials each (zf_fresh_len) and no full chain runs at all
            on_full = full_n > 0 and full_ms >= part_ms
            ker_ms = full_ms / full_n if on_full else part_ms / part_n
            ker_ms_events = ker_ms
            ra_line = on_full and M.get("ra_ahead", 0) > 0
            if ra_line:
                # Run-ahead passes overlap: the event interval of a pass includes its wait for the pass before it (and two
                # passes share the CUs while they overlap), so no interval measured around ONE launch is the cost of a pass.
                # What a pass costs is what the block delivers: the median block / its passes (first pass, last
                # finalisation and the poll included) - the figure the fractions below are computed from.
                ker_ms = dt * 1e3 / max(passes / max(len(blocks), 1), 1.0)
            trials = float(S) if on_full else part_fresh / part_n      # fresh trials per pass
            replays = 0.0 if on_full else part_lag / part_n             # replayed iterations per pass
            pass_bytes = (PASS_BYTES_PER_ELEM if S > 1 else ALG_BYTES_PER_ELEM) * n
            mid = (not on_full) and S == 16 and abs(trials - round(trials)) < 1e-9 and MID_MIN <= round(trials) <= MID_MAX and replays == 0
            mid_len = int(round(trials)) if mid else 0
            kp = kernel_profile(n, S, "full" if on_full else ("mid" if mid else "general"), mid_len)   # THIS kernel's counters (None: not profiled)
            traffic = kp.get("hbm_bytes_per_launch") if kp else measured_traffic(n, S)
            traffic_file = kp["file"] + " [" + kp["part"] + "]" if kp else "profiles/" + PMC_PROFILE.format(S=S)
            achieved = pass_bytes / (ker_ms * 1e-3) / 1e9
            hbm_frac = achieved / HBM_PEAK_GBS
            vpe = valu_per_element_trial(S)
            valu_ms = valu_frac = None
            clk = (kp.get("engine_clock_GHz") or {}).get("median") if kp else measured_clock_ghz(n, S)
            # a replayed iteration is the iterate arithmetic alone: 11 fp64 operations + the sign copy
            inst_per_elem = None if not vpe else vpe * trials + 12.0 * replays
            inst_source = "profiles/" + ISA_PROFILE.format(S=S) + " (ISA: instructions per element and trial x trials per pass)"
            if kp and kp.get("valu_lane_instructions_per_element") and (
                    on_full or (kp.get("trials_per_pass") and abs(kp["trials_per_pass"] - trials) < 0.01 and replays == 0)):
                # measured for exactly this pass shape: SQ_INSTS_VALU per launch x 64 lanes / n
                inst_per_elem = kp["valu_lane_instructions_per_element"]
                vpe = vpe or inst_per_elem / max(trials, 1.0)
                inst_source = kp["file"] + " [" + kp["part"] + "]: SQ_INSTS_VALU per launch x 64 / n (measured for this pass shape)"
            if vpe:
                valu_ms = inst_per_elem * n / (FP64_LANES_PER_CYCLE * CLOCK_GHZ * 1e9) * 1e3
                valu_frac = valu_ms / ker_ms
            bound = "hbm" if valu_frac is None or hbm_frac >= valu_frac else "fp64_valu"
            hbm = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac,
                   "bytes_per_launch": pass_bytes,
                   "bytes_note": "HBM bytes a full-chain pass moves: 4 streams read + 2 iterates written = 48 B per "
                                 "element (S > 1), 40 B for S = 1"}
            valu = None if vpe is None else {
                "achieved": inst_per_elem * n / (ker_ms * 1e-3) / 1e12,
                "peak": FP64_LANES_PER_CYCLE * CLOCK_GHZ * 1e9 / 1e12,
                "unit": "T lane-instructions/s",
                "frac": valu_frac,
                "valu_instructions_per_element_trial": vpe,
                "valu_instructions_per_element_and_pass": inst_per_elem,
                "valu_busy_fraction_measured": kp.get("valu_busy_fraction") if kp else None,
                "source": inst_source,
                "min_ms_at_2.4GHz": valu_ms,
                "engine_clock_GHz_under_this_kernel": clk,
                "frac_at_that_clock": None if clk is None else valu_frac * CLOCK_GHZ / clk,
                "note": "fp64 VALU issue slots (256 CUs x 4 SIMDs x 16 lanes per cycle x 2.4 GHz; a wave64 fp64 "
                        "instruction of any kind - add, mul, fma, max - takes one slot of 4 cycles) used by the chain / "
                        "available in the measured duration at the NOMINAL clock; the clock drops under this load "
                        "(power limit; median clock from the GRBM_GUI_ACTIVE pass in profiles/r02_s*_pmc_traffic.json), "
                        "so the fraction of the slots at the running clock - frac_at_that_clock - is higher",
            }
            top = hbm if bound == "hbm" else valu
            line["roofline"] = {
                "bound": bound,
                "achieved": top["achieved"],
                "peak": top["peak"],
                "unit": top["unit"],
                "frac": top["frac"],
                "traffic": traffic,
                "traffic_source": f"{traffic_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bytes per launch "
                                  "of the kernel named below)" if traffic else None,
                "kernel": ((f"zf_runahead_kernel<nesterov, nt> (full-chain passes of {S} trials, run-ahead)" if ra_line else
                            f"zf_trial_kernel<grad inline, nesterov, nt, S={S}{', AHEAD' if M.get('rep_ahead', 0) else ''}> (full-chain passes)")
                           + (" [ZF_ACCEPT_RESOLVED instantiation]" if args.acceptance == "resolved" else "")) if on_full else
                          (f"zf_trial_kernel<grad inline, nesterov, nt, S={S}, PART 3, L={mid_len}> (branch-free chain of {mid_len} trials: "
                           f"the K = {K} timed iterations are shared by passes of {mid_len}; no full chain runs)" if mid else
                           f"zf_trial_kernel<grad inline, nesterov, nt, S={S}> (general body: passes of {trials:.1f} fresh "
                           f"trials + {replays:.1f} replayed iterations on average; no full chain dominates K = {K})"),
                "kernel_avg_ms": ker_ms,
                "kernel_avg_ms_note": ("run-ahead passes overlap (a launch's own event interval - kernel_event_interval_ms - includes its "
                                       "wait for the pass before it): kernel_avg_ms is what the timed block DELIVERS per pass, median "
                                       "block / passes, first pass, last finalisation and poll included") if ra_line else None,
                "kernel_event_interval_ms": ker_ms_events if ra_line else None,
                "kernel_launches_timed": full_n if on_full else part_n,
                "trials_per_pass": trials,
                "replayed_iterations_per_pass": replays,
                "hbm": hbm,
                "fp64_valu": valu,
                "other_passes_avg_ms": part_ms / part_n if part_n else None,
                "full_chain_passes_avg_ms": full_ms / full_n if full_n else None,
                "equivalent_one_iteration_GBps": ALG_BYTES_PER_ELEM * n * trials / (ker_ms * 1e-3) / 1e9,
                "equivalent_note": "40 B x n x trials per pass (SURVEY 8d's per-iteration bytes x iterations per pass) / duration: "
                                   "not a bandwidth - the chain does not move those bytes",
            }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(d, c)
    return line


if __name__ == "__main__":
    main()
