#!/usr/bin/env python3
"""Multi-objective parity report (GPU box): for every G4 case the engine's deviation from the
reference's stored outputs next to the reference's OWN spread under a permuted feature order
(tests/golden/g10_reference_noise_floor.json).  Prints one JSON line per case; used to set the
tolerances of tests/test_gpu_multiobjective.py to max(1e-10, k x floor)."""
from __future__ import annotations

import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rel(a, b):
    den = np.linalg.norm(b)
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / (den if den > 0 else 1.0))


def main():
    from test_gpu_multiobjective import _cases
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.multiobjective import X_K, X_NEW, Y, device_dual, solve_dual

    dual_solver = sys.argv[1] if len(sys.argv) > 1 else "scipy"
    G = np.load(os.path.join(ROOT, "tests", "golden", "g4_multiobjective.npz"))
    floor = json.load(open(os.path.join(ROOT, "tests", "golden", "g10_reference_noise_floor.json")))["cases"]
    for tag, (make, _, kw) in _cases().items():
        p = make()
        m = p.n_objectives
        eng = p._engine()
        x0, y, lr = G[f"{tag}.x0"], G[f"{tag}.sub.y"], float(G[f"{tag}.sub.lr"])
        eng.set_x0(x0)
        eng.put(Y, y)
        f0, g0 = eng.eval_F(X_K)
        f_y = eng.prepare()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            w, dual_fun, nit = solve_dual(device_dual(eng, lr, f_y, f0 + g0, False), m, np.ones(m) / m, 1e-12,
                                          100000, solver=dual_solver)
        eng.recover(lr, w)
        x = eng.get(X_NEW)
        rec = dict(case=tag, m=m, solver=dual_solver,
                   sub_x_rel=rel(x, G[f"{tag}.sub.x"]), floor_sub_x_rel=floor[tag]["sub_x_rel"],
                   sub_w_abs=float(np.max(np.abs(w - G[f"{tag}.sub.weight"]))), floor_sub_w_abs=floor[tag]["sub_w_abs"],
                   sub_fun_rel=abs(-dual_fun - float(G[f"{tag}.sub.fun"])) / abs(float(G[f"{tag}.sub.fun"])),
                   floor_sub_fun_rel=floor[tag]["sub_fun_rel"], sub_nit=int(nit), ref_sub_nit=floor[tag]["sub_nit"])
        worst_x = worst_F = worst_e = 0.0
        nit_equal = True
        for v, nest in (("ista", False), ("fista", True)):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                res = minimize_proximal_gradient(*make().callbacks(), x0, nesterov=nest, tol=1e-5, max_iter=12,
                                                 return_all=True, dual_solver=dual_solver, **kw)
            nit_equal &= res.nit == int(G[f"{tag}.{v}.nit"])
            vecs = G[f"{tag}.{v}.vecs"]
            k = min(len(res.allvecs), len(vecs))
            for a, b in zip(res.allvecs[:k], vecs[:k]):
                worst_x = max(worst_x, rel(a, b))
            Fa, Fb = np.stack(res.allfuns)[:k], G[f"{tag}.{v}.allfuns"][:k]
            with np.errstate(invalid="ignore"):
                d = np.abs(Fa - Fb) / np.maximum(np.abs(Fb), 1e-300)
            worst_F = max(worst_F, float(np.nanmax(d)))
            ea, eb = np.asarray(res.allerrs)[:k - 1], G[f"{tag}.{v}.allerrs"][:k - 1]
            if len(ea):
                worst_e = max(worst_e, float(np.max(np.abs(ea - eb))))
        rec.update(trace_x_rel=worst_x, floor_trace_x_rel=floor[tag]["trace_x_rel"], trace_F_rel=worst_F,
                   floor_trace_F_rel=floor[tag]["trace_F_rel"], trace_err_abs=worst_e,
                   floor_trace_err_abs=floor[tag]["trace_err_abs"], nit_equal=bool(nit_equal))
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
