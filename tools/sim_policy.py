#!/usr/bin/env python3
"""Offline pass-planning simulator.  The accept / reject sequence of a solve does not depend on how its trials are
grouped into passes (every decision is taken on the sums of its own trial), so a planning rule can be judged on a
recorded sequence (tools/dump_trials.py) with a cost model of a pass fitted to per-shape kernel means
(tools/long_run.py `by_shape`):   tools/sim_policy.py trials.json [scale]"""
import json
import sys

S = 16


def cost(lag, nf, scale):
    """ms of a pass at n = 1e8 (scale = n / 1e8 for other sizes): measured means, round 4."""
    if lag == 0 and nf == 16:
        return 1.20 * scale
    if lag == 0:
        tab = {1: 0.80, 2: 0.80, 3: 0.81, 4: 0.82, 5: 0.82, 6: 0.83, 7: 0.83, 8: 0.83, 9: 0.90, 10: 0.90, 11: 0.91, 12: 0.95,
               13: 1.02, 14: 1.10, 15: 1.25}
        return tab[nf] * scale
    fresh = 0.87 if nf >= 8 else 0.80 + 0.00875 * nf
    if nf > 8:
        fresh = 0.87 + 0.0725 * (nf - 8)
    return (fresh + 0.033 * lag) * scale


def simulate(trials, plan, ckpt_lag=None, scale=1.0):
    """trials[k] = trials iteration k + 1 took.  plan(state) -> fresh trials of the next pass."""
    K = len(trials)
    nit, lag, t_ms, passes = 0, 0, 0.0, 0
    rej_left = trials[0] - 1          # rejections still to come before iteration nit + 1 is accepted
    hist = []                         # recent pass outcomes for adaptive rules
    replays = 0
    while nit < K:
        left = K - nit
        nf = plan(lag, left, hist)
        nf = max(1, min(nf, left, 2 * S - 1 - lag))
        ck = ckpt_lag is not None and lag >= ckpt_lag
        t_ms += cost(lag, nf, scale)
        replays += lag
        passes += 1
        acc = 0
        broke = False
        for _ in range(nf):
            if rej_left > 0:
                rej_left -= 1
                broke = True
                break
            acc += 1
            nit += 1
            if nit < K:
                rej_left = trials[nit] - 1
            else:
                break
        if ck:
            lag = acc                  # the replayed pair was stored: only this pass's acceptances lag
        elif acc == nf and not broke:
            lag = 0
        else:
            lag = lag + acc
        hist.append((nf, acc, broke))
        if nit >= K and lag > 0:       # one materialise-only pass
            t_ms += cost(lag, 0 + 1, scale)
            passes += 1
            lag = 0
    return dict(ms=t_ms, passes=passes, it_per_s=K / t_ms * 1e3, replays=replays)


def plan_current(lag, left, hist):
    if lag > 0:
        return 8
    if left >= 2 * S or left == S:
        return S
    return (left + 1) // 2 if left > S else left


def make_adaptive(after_break, recover):
    """fresh trials behind a broken chain = after_break; with nothing lagging: full chain unless one of the last
    `recover` passes broke, then 8."""
    def plan(lag, left, hist):
        if lag > 0:
            return after_break
        recent = hist[-recover:] if recover else []
        if any(b for _, _, b in recent):
            return min(8, left)
        return plan_current(0, left, hist)
    return plan


if __name__ == "__main__":
    d = json.load(open(sys.argv[1]))
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else d["n"] / 1e8
    tr = d["trials"]
    print("current            ", simulate(tr, plan_current, None, scale))
    for ck in (4, 6, 8, 10, 12):
        print(f"ckpt at lag >= {ck:2d}  ", simulate(tr, plan_current, ck, scale))
    for ab in (4, 6, 8, 10, 12):
        for rec in (0, 1, 2, 3):
            print(f"fresh {ab:2d} behind lag, 8 for {rec} passes after a break, ckpt 8", simulate(tr, make_adaptive(ab, rec), 8, scale))
