"""GPU: bench.py keeps the driver's contract - exactly one JSON line on stdout with the agreed
keys, exactly K timed iterations, a roofline block and (N = 1) a CPU baseline block."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "2000000", "--steps", "21",
                          "--warmup", "5", "--min-seconds", "0.05", *extra], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must hold exactly one line, got {len(lines)}"
    return json.loads(lines[0])


def test_bench_json_line():
    d = _run()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert (d["n_gpus"], d["steps"], d["warmup"], d["scaling"], d["dtype"], d["data"]) == (1, 21, 5, "weak", "f64",
                                                                                         "synthetic")
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "fp64_valu") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    top = r[r["bound"]]
    assert (r["achieved"], r["peak"], r["unit"], r["frac"]) == (top["achieved"], top["peak"], top["unit"], top["frac"])
    # fractions of roofs: bytes the pass really moves (issue slots it really needs) over its duration
    h, v = r["hbm"], r["fp64_valu"]
    assert h["unit"] == "GB/s" and h["peak"] == 8000.0 and 0 < h["frac"] <= 1.0
    assert h["achieved"] == h["bytes_per_launch"] / (r["kernel_avg_ms"] * 1e-3) / 1e9 and h["bytes_per_launch"] == 48 * 2000000
    assert v is None or 0 < v["frac"] <= 1.0
    assert r["frac"] == max(h["frac"], v["frac"] if v else 0.0) and r["equivalent_one_iteration_GBps"] > h["achieved"]
    # 21 iterations with chains of up to 16: fewer than two full chains' worth, so two passes share them
    # (11 + 10 fresh trials, zf_fresh_len) and the roofline describes those passes; the block is repeated
    cfg = d["config"]
    assert cfg["passes_per_block"] >= 2 and cfg["blocks"] >= 2 and cfg["temporal_blocking_chain"] == 16
    assert cfg["full_chain_passes"] == 0 and cfg["other_passes"] >= 2 * cfg["blocks"]
    assert abs(r["trials_per_pass"] - 10.5) < 1e-9 and r["replayed_iterations_per_pass"] == 0
    assert ("general body" in r["kernel"] or "PART 3" in r["kernel"]) and r["kernel_launches_timed"] == cfg["other_passes"]
    assert cfg["ms_per_step_min_block"] <= cfg["ms_per_step_median_block"] == d["ms_per_step"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    assert abs(d["ms_per_step"] * d["steps"] / 1e3 - d["steps"] / d["value"] * (2000000 / 1e8)) < 1e-9


def test_bench_line_says_when_passes_ran_ahead():
    """64 iterations from iteration 16 at n = 2e6: four full chains in a row on a one-round grid - the second to
    fourth are launched on the other stream while their predecessor is finalised, and the line says so."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "2000000", "--steps", "64", "--warmup", "16",
                          "--min-seconds", "0.05", "--no-cpu-baseline", "--no-regimes"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][0])
    ra = d["config"]["runahead"]
    assert ra["passes"] >= 4 * d["config"]["blocks"] and ra["launched_behind_a_pass_in_flight"] >= 3 * d["config"]["blocks"]
    r = d["roofline"]
    assert "run-ahead" in r["kernel_avg_ms_note"] and d["config"]["full_chain_passes"] == 4 * d["config"]["blocks"]
    # the cost of a pass on such a line is what the block delivers per pass, not an interval around one launch
    assert abs(r["kernel_avg_ms"] - d["ms_per_step"] * 64 / 4) < 1e-9 and r["kernel_event_interval_ms"] > 0
    assert r["hbm"]["achieved"] == r["hbm"]["bytes_per_launch"] / (r["kernel_avg_ms"] * 1e-3) / 1e9


def test_bench_flags():
    d = _run("--no-cpu-baseline", "--no-kernel-events")
    assert "cpu_baseline" not in d and d["steps"] == 21
    assert d["config"]["rccl"] is None and d["config"]["overrides"] == {}
    # (21 iterations = two passes of 11 + 10 trials: no two full chains in a row, nothing runs ahead)
    assert d["config"]["runahead"] == {"passes": 0, "launched_behind_a_pass_in_flight": 0, "waits_that_gave_up": 0, "void_passes": 0,
                                       "switched_off": False}
    assert d["config"]["passes_ahead"] == {"launched": 0, "void": 0} and d["config"]["acceptance"] == "reference"
    regimes = d["config"]["iterations_per_sec_by_regime"]
    assert {"clean_regime_K20_W5", "clean_regime_K24_W5", "clean_regime_K28_W5", "clean_regime_K30_W5",
            "across_the_noise_floor_K100_W10", "across_the_noise_floor_K100_W10_acceptance_resolved"} <= set(regimes)
    assert all(v > 0 for v in regimes.values())
    # the per-pass exchange through a REAL (1-rank) RCCL communicator: the line says what RCCL itself reports
    d = _run("--no-cpu-baseline", "--no-regimes", "--libcomm")
    r = d["config"]["rccl"]
    assert r["via"].startswith("zf_comm (RCCL") and (r["world"], r["rank_count_seen"], r["rccl_user_rank"]) == (1, 1, 0)
    assert "rccl" in r["library"].lower() and r["exchanges_timed"] > 0 and 0.0 < r["exchange_ms_per_pass"] < 5.0
    # (round 5) through a communicator the exactly predicted passes run AHEAD of their predecessor's decision: the line counts them
    assert d["config"]["passes_ahead"]["launched"] >= 1 and d["config"]["passes_ahead"]["void"] == 0
    # the acceptance test resolved below ulp(F): no trial of this workload is ever rejected, a K = 100 block is seven passes
    d = _run("--no-cpu-baseline", "--no-regimes", "--steps", "100", "--warmup", "10", "--acceptance", "resolved")
    assert d["config"]["acceptance"] == "resolved" and abs(d["config"]["passes_per_block"] - 7.0) < 1e-9
    assert "ZF_ACCEPT_RESOLVED" in d["roofline"]["kernel"]
    d = _run("--no-cpu-baseline", "--total-n", "3000000")
    assert d["scaling"] == "strong" and d["config"]["n_total"] == 3000000


def test_bench_two_ranks_share_the_gpu_over_gloo():
    """The N > 1 flow of bench.py end to end - one process per rank, sharded solver, per-pass pack
    exchange, barrier-bracketed timing, max over ranks, one JSON line from rank 0 - rehearsed with
    two ranks on this one GPU (ZF_BENCH_BACKEND=gloo; the driver's runs use RCCL, one GPU per rank)."""
    env = dict(os.environ, ZF_BENCH_BACKEND="gloo")
    # --standalone: the agent's own store (bound to port 0 and kept open) is the rendezvous - no picked port
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1",
                          "--nnodes=1", "--nproc-per-node", "2", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--elements", "2000000", "--steps", "24", "--warmup", "8",
                          "--min-seconds", "0.05"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert (d["n_gpus"], d["steps"], d["scaling"]) == (2, 24, "weak") and "cpu_baseline" not in d
    assert d["config"]["n_total"] == 4000000 and d["config"]["passes_per_block"] >= 2
    # whole-job aggregate: two shards' worth of iterations per unit time
    assert abs(d["value"] - 2 * 24 / (d["ms_per_step"] * 24 / 1e3) * (2000000 / 1e8)) < 1e-6 * d["value"]


def test_bench_n_gt_1_code_path_with_eight_thread_ranks():
    """What the driver executes on a whole node - bench.py's N > 1 path: x sharded over the ranks, the library's own
    communicator issuing the per-pass pack exchange, zf_decide_kernel on the gathered packs, barrier-bracketed
    timing, max over ranks, ONE JSON line from rank 0 - as a dry run on this one GPU: 8 rank THREADS x n = 1e7
    (BASELINE cfg5's layout at an eighth of its size) through the in-process communicator group behind the same
    zf_comm_all_gather RCCL serves (--thread-ranks).  The JSON contract must hold; n_gpus stays 1."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--thread-ranks", "8", "--n", "10000000",
                          "--steps", "20", "--warmup", "5", "--min-seconds", "0.2", "--no-regimes"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert (d["n_gpus"], d["steps"], d["warmup"], d["scaling"], d["dtype"]) == (1, 20, 5, "weak", "f64")
    cfg = d["config"]
    assert cfg["thread_ranks"] == 8 and cfg["n_total"] == 80000000 and cfg["n_per_gpu"] == 10000000
    # what carried the per-pass exchange, as the communicator reports it, and what one exchange cost on rank 0's stream
    assert cfg["rccl"]["world"] == 8 and cfg["rccl"]["rank_count_seen"] == 8 and "thread-rank" in cfg["rccl"]["via"]
    assert cfg["rccl"]["exchanges_timed"] > 0 and cfg["rccl"]["exchange_ms_per_pass"] > 0.0
    assert cfg["overrides"] == {}
    assert "DRY RUN" in cfg["parallelism"] and "cpu_baseline" not in d
    assert cfg["passes_per_block"] >= 2 and cfg["temporal_blocking_chain"] == 16
    # whole-job aggregate in units of one 1e8-element shard: 8 ranks x 0.1 shard each
    assert abs(d["value"] - 8 * 20 / (d["ms_per_step"] * 20 / 1e3) * 0.1) < 1e-6 * d["value"]
    r = d["roofline"]
    assert 0 < r["frac"] <= 1.0 and r["kernel_launches_timed"] >= 2
