"""Pareto-front metrics over lists of solver results - the counterpart of
``zfista/metrics.py``, without pymoo.

These are post-processing of a handful of m-vectors on the host (the reference does the same
with pymoo); nothing here touches the iteration path.  Restated from the definitions the
reference implements:

* non-dominated filtering        zfista/metrics.py:28-44   (pymoo ``NonDominatedSorting``,
                                 first front, ascending index order)
* purity                         :47-63    |front ∩ front_true| / |front_true|
* spread metrics Gamma, Delta    :66-100   (Custodio et al. 2011), per objective, maximised
* hypervolume                    :139      pymoo ``Hypervolume(pf=front_true)``: the volume
                                 dominated by the front inside the box whose far corner is
                                 the nadir of ``front_true`` (no normalisation)
* the metric / ratio tables      :103-199

pymoo itself is not available in this image; its two functions are pinned by the known
answers of the reference's own tests (tests/test_metrics.py:36-118), reproduced in
tests/test_metrics.py, and the hypervolume additionally by closed-form and Monte-Carlo checks.
Deviation: ``res.nit_internal`` is read with a NaN default - the reference's solver never sets
it, so ``calculate_metrics`` on real results raises AttributeError there (:159).
"""
from __future__ import annotations

import numpy as np


def extract_function_values(res):
    """Objective vectors of a list of ``OptimizeResult`` as an (N, m) array."""
    return np.vstack([r.fun for r in res])


def _non_dominated_mask(F):
    F = np.asarray(F, dtype=np.float64)
    n = F.shape[0]
    keep = np.ones(n, dtype=bool)
    for i in range(n):
        # j dominates i: no worse in every objective and better in at least one
        le = np.all(F <= F[i], axis=1)
        lt = np.any(F < F[i], axis=1)
        if np.any(le & lt):
            keep[i] = False
    return keep


def extract_non_dominated_points(F):
    """Rows of F no other row dominates (minimisation), in their original order."""
    F = np.asarray(F)
    return F[_non_dominated_mask(F)]


def purity(front, front_true):
    return len(front) / len(front_true)


def spread_metrics(front, front_true):
    """(Gamma, Delta): largest gap and gap non-uniformity of ``front`` inside the extent of
    ``front_true``, maximised over the objectives; (inf, inf) for fewer than two points."""
    front, front_true = np.asarray(front), np.asarray(front_true)
    if len(front) <= 1:
        return np.inf, np.inf
    gamma = 0
    delta = 0
    for j in range(front_true.shape[1]):
        col = np.sort(front[:, j])
        gaps = np.diff(col)
        head = col[0] - np.min(front_true[:, j])
        tail = np.max(front_true[:, j]) - col[-1]
        gamma = max(np.max(gaps), head, tail, gamma)
        mean_gap = np.mean(gaps)
        num = head + tail + np.sum(np.abs(gaps - mean_gap))
        den = head + tail + (len(col) - 1) * mean_gap
        delta = max(delta, num / den)
    return gamma, delta


def hypervolume(front, ref_point):
    """Exact volume of the union of boxes [p, ref_point] over the points p of ``front``
    (minimisation; points not strictly better than ``ref_point`` in every objective add
    nothing).  m = 1, 2: sweep; m >= 3: slicing along the last objective."""
    P = np.asarray(front, dtype=np.float64).reshape(-1, len(ref_point))
    ref = np.asarray(ref_point, dtype=np.float64)
    P = P[np.all(P < ref, axis=1)]
    if P.shape[0] == 0:
        return 0.0
    m = P.shape[1]
    if m == 1:
        return float(ref[0] - P[:, 0].min())
    if m == 2:
        P = P[np.argsort(P[:, 0], kind="stable")]
        vol, best = 0.0, ref[1]
        for a, b in P:
            if b < best:
                vol += (ref[0] - a) * (best - b)
                best = b
        return float(vol)
    # slice along the last coordinate: between consecutive levels the (m-1)-dimensional
    # cross-section is the hypervolume of the points at or below the slab
    order = np.argsort(P[:, -1], kind="stable")
    P = P[order]
    levels = np.append(P[:, -1], ref[-1])
    vol = 0.0
    for k in range(P.shape[0]):
        depth = levels[k + 1] - levels[k]
        if depth > 0:
            vol += depth * hypervolume(P[:k + 1, :-1], ref[:-1])
    return float(vol)


def _common_points(front_true, front):
    both = {tuple(p) for p in front_true}.intersection({tuple(p) for p in front})
    return np.array(list(both))


def _ratio(best, value, larger_is_better):
    if larger_is_better:
        if value != 0:
            return best / value
        return np.inf if best != 0 else 1
    if best != 0:
        return value / best
    return np.inf if value != 0 else 1


def calculate_metrics(*named_results):
    """``(name, [OptimizeResult, ...])`` pairs -> (metrics, ratios): per metric a dict over the
    names; ``ratios`` relates every value to the best one (>= 1).  The reference front is the
    non-dominated set of all fronts together."""
    names, results = zip(*named_results)
    fronts = [extract_non_dominated_points(extract_function_values(r)) for r in results]
    front_true = extract_non_dominated_points(np.concatenate(fronts, axis=0))
    common = [_common_points(front_true, f) for f in fronts]
    nadir = np.max(front_true, axis=0)

    def mean_of(attr, res_list):
        vals = [getattr(r, attr, np.nan) for r in res_list if r.success]
        return np.mean(vals) if vals else np.nan

    spreads = [spread_metrics(c, front_true) for c in common]
    table = {
        "Hypervolume": [hypervolume(f, nadir) for f in fronts],
        "Gamma": [s[0] for s in spreads],
        "Delta": [s[1] for s in spreads],
        "Purity": [purity(c, front_true) for c in common],
        "Error rate": [np.mean([not r.success for r in res_list]) for res_list in results],
        "Avg computation time": [mean_of("time", res_list) for res_list in results],
        "Avg iterations": [mean_of("nit", res_list) for res_list in results],
        "Avg internal iterations": [mean_of("nit_internal", res_list) for res_list in results],
    }
    metrics = {key: dict(zip(names, vals)) for key, vals in table.items()}
    ratios = {}
    for key, vals in metrics.items():
        larger = key in ("Hypervolume", "Purity")
        best = max(vals.values()) if larger else min(vals.values())
        ratios[key] = {name: _ratio(best, v, larger) for name, v in vals.items()}
    return metrics, ratios
