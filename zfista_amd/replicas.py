"""Replicas: independent solves spread over the GPUs of one node, one worker process per GPU.

The reference's only parallelism is ``joblib.Parallel(n_jobs=-1)`` over independent start points
(benchmarks/benchmark.py:325,341,360; examples/cameraman.ipynb:347).  For opaque Python callbacks
that is also the only multi-GPU mode there is ("replicas only" - nothing of a solve is
shared between devices, so there is no collective).  ``solve_replicas`` keeps that shape:

    results = solve_replicas(make_problem, starts, gpus=8, **solver_kwargs)

``make_problem()`` is called once in every worker (after the worker has bound its GPU) and must
return an object with ``minimize_proximal_gradient(x0, **kw)`` (a ``zfista_amd.problems`` class) or a
4-tuple of callbacks; start ``i`` goes to worker ``i % workers``.  Results come back in the order of
``starts`` (``allvecs`` as plain lists of arrays, so that they pickle).  Workers are spawned (never
forked: a forked child must not inherit an initialised HIP runtime) and see their GPU as device 0
through ``HIP_VISIBLE_DEVICES``; with fewer GPUs than workers the workers share devices round-robin.
"""
from __future__ import annotations

import multiprocessing as mp
import os
import traceback
import warnings


class ProblemRecipe:
    """Picklable ``make_problem`` for a ``zfista_amd.problems`` multi-objective problem: the class name
    and constructor arguments (the problem object itself owns GPU memory of the parent process)."""

    def __init__(self, problem):
        import inspect

        import numpy as np

        self.cls = type(problem).__name__
        self.kw = {}
        if "n_features" in inspect.signature(type(problem).__init__).parameters:
            self.kw["n_features"] = problem.n_features
        if getattr(problem, "l1_ratios", None) is not None:
            self.kw.update(l1_ratios=np.asarray(problem.l1_ratios), l1_shifts=np.asarray(problem.l1_shifts))
        if getattr(problem, "bounds", None) is not None:
            self.kw["bounds"] = problem.bounds

    def __call__(self):
        from . import problems as Z

        return getattr(Z, self.cls)(**self.kw)


def _worker(rank, device, make_problem, jobs, kw, quiet, conn):
    try:
        os.environ["HIP_VISIBLE_DEVICES"] = str(device)   # before anything initialises HIP in this process
        from .proximal_gradient import minimize_proximal_gradient

        prob = make_problem()
        out = []
        for idx, x0 in jobs:
            with warnings.catch_warnings():
                if quiet:
                    warnings.simplefilter("ignore")
                if hasattr(prob, "minimize_proximal_gradient"):
                    res = prob.minimize_proximal_gradient(x0, **kw)
                else:
                    res = minimize_proximal_gradient(*prob, x0, **kw)
            if res.get("allvecs") is not None:
                res["allvecs"] = [v for v in res["allvecs"]]   # a lazy device history does not pickle
            out.append((idx, res))
        conn.send(("ok", out))
    except Exception:   # reported to the parent, which raises
        conn.send(("error", f"replica worker {rank} (GPU {device}):\n{traceback.format_exc()}"))
    finally:
        conn.close()


def visible_gpus():
    """Devices this process may use, as HIP_VISIBLE_DEVICES indices (no HIP initialisation here)."""
    env = os.environ.get("HIP_VISIBLE_DEVICES")
    if env:
        return [d for d in env.split(",") if d != ""]
    import torch

    return [str(i) for i in range(max(torch.cuda.device_count(), 1))]


def solve_replicas(make_problem, starts, gpus=None, workers=None, quiet=True, **solver_kwargs):
    """Solve from every start point; one worker process per GPU (``workers`` may exceed the GPU
    count: they then share devices).  Returns the list of OptimizeResults in the order of ``starts``."""
    starts = list(starts)
    devices = visible_gpus()
    if gpus is not None:
        devices = devices[:max(1, int(gpus))]
    n_workers = min(len(starts), int(workers) if workers else len(devices))
    if n_workers < 1:
        return []
    ctx = mp.get_context("spawn")
    procs = []
    for r in range(n_workers):
        jobs = [(i, starts[i]) for i in range(r, len(starts), n_workers)]
        parent, child = ctx.Pipe(duplex=False)
        p = ctx.Process(target=_worker, args=(r, devices[r % len(devices)], make_problem, jobs, solver_kwargs, quiet, child))
        p.start()
        child.close()
        procs.append((p, parent))
    results, errors = [None] * len(starts), []
    for p, conn in procs:
        try:
            status, payload = conn.recv()
        except EOFError:
            status, payload = "error", f"replica worker exited without a result (exit code {p.exitcode})"
        if status == "ok":
            for idx, res in payload:
                results[idx] = res
        else:
            errors.append(payload)
        p.join()
    if errors:
        raise RuntimeError("\n".join(errors))
    return results


def solve_on_streams(jobs, streams=None, quiet=True):
    """Independent solves on ONE GPU at the same time: each job runs in a host thread of its own, on a HIP stream of its
    own, so that solves too small to fill the device - a 256 x 256 deblurring problem is two or three launches of 128 workgroups
    per iteration on a 256-CU chip - overlap instead of queueing behind each other.  The pattern of the reference's
    notebook (``examples/cameraman.ipynb`` cell 11: ``joblib.Parallel`` over 15 momentum settings of one problem) without
    processes: the problem's device arrays are shared, nothing is pickled.

    ``jobs``: iterable of ``(callbacks, x0, kwargs)`` - ``callbacks`` the 4-tuple ``(f, g, jac_f, prox_wsum_g)`` (or an
    object with ``callbacks()``); ``streams``: how many run at once (default: all jobs, at most 16).  Returns the results
    in the order of ``jobs``; an exception in a job is raised here.  Solves are independent: every result equals the one
    the same call gives alone."""
    import threading

    import torch

    from .proximal_gradient import minimize_proximal_gradient

    jobs = list(jobs)
    width = max(1, min(len(jobs), int(streams) if streams else 16))
    results, errors = [None] * len(jobs), []
    lock = threading.Lock()
    todo = list(range(len(jobs)))
    device = torch.cuda.current_device()

    def worker():
        torch.cuda.set_device(device)
        with torch.cuda.stream(torch.cuda.Stream()):
            while True:
                with lock:
                    if not todo or errors:
                        return
                    i = todo.pop(0)
                cbs, x0, kw = jobs[i]
                if hasattr(cbs, "callbacks"):
                    cbs = cbs.callbacks()
                try:
                    with warnings.catch_warnings():
                        if quiet:
                            warnings.simplefilter("ignore")
                        results[i] = minimize_proximal_gradient(*cbs, x0, **kw)
                except Exception:   # reported to the caller, which raises
                    with lock:
                        errors.append(f"job {i}:\n{traceback.format_exc()}")
                    return
            torch.cuda.current_stream().synchronize()

    threads = [threading.Thread(target=worker) for _ in range(width)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise RuntimeError("\n".join(errors))
    return results
