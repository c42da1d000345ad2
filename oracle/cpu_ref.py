"""CPU oracle: NumPy restatement of the zfista proximal-gradient path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``zfista_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / the timed CPU baseline.

Pinning: ``tests/golden/make_golden.py`` runs this restatement side by side
with the imported reference solver (``/root/reference/zfista/proximal_gradient.py``)
on identical callbacks and asserts identical iterates; the resulting vectors are
committed under ``tests/golden/`` and re-checked by ``tests/test_oracle_golden.py``.

What is restated (reference file:line, relative to /root/reference):

* single-objective trial            zfista/proximal_gradient.py:140-157
* dual value/gradient (m >= 2)      zfista/proximal_gradient.py:161-177
* m == 2 / m >= 3 dual solves       zfista/proximal_gradient.py:179-205
* primal recovery                   zfista/proximal_gradient.py:206-208
* backtracking line search          zfista/proximal_gradient.py:279-307
* outer loop, momentum, stopping    zfista/proximal_gradient.py:463-538
* result assembly                   zfista/proximal_gradient.py:451-457,495-509,539-554

Third-party arithmetic on the path that is NOT under /root/reference:
``scipy.optimize.minimize_scalar`` (bounded Brent) and
``scipy.optimize.minimize(method="trust-constr")`` (scipy unpinned in
pyproject.toml:18; 1.15.3 in this image).  The oracle calls the same SciPy
entry points with the same options, so the dual iterate sequence is whatever
the installed SciPy produces - exactly as for the reference.

Extra (oracle-only) result keys used by the parity tests: ``alllrs`` (learning
rate after each outer iteration) and ``alltrials`` (line-search trials used by
each outer iteration).  ``verbose=True`` prints a five-column row; the
reference's formatter raises IndexError there (five slots, four values,
proximal_gradient.py:511-520), a documented deviation.

NOT in the reference (an extension of the engine, restated here so that it has a checker too): ``f_diff``.  The
reference's acceptance test ``F(x+) - F(x_k) <= fun + tol`` (:303) subtracts O(|F|) numbers; with F(x_k) and g(x+)
cancelled it reads ``[f(x+) - f(y)] - <grad f(y), x+ - y> - |x+ - y|^2 / 2 / lr <= tol``.  When a callable
``f_diff(x_new, y)`` is given - f(x_new) - f(y) formed without the cancellation - the line search uses that form
(``acceptance="resolved"`` of zfista_amd.minimize_proximal_gradient); without it everything is the reference's.
"""
from __future__ import annotations

import time
import warnings

import numpy as np
from scipy.optimize import (
    BFGS,
    Bounds,
    LinearConstraint,
    OptimizeResult,
    minimize,
    minimize_scalar,
)

MSG_SUCCESS = "Optimization terminated successfully"  # proximal_gradient.py:527
MSG_MAXITER = "Maximum number of iterations reached"  # proximal_gradient.py:541
MSG_BACKTRACK = "Backtracking failed to find a suitable stepsize."  # :307
MSG_DEPRECATED = (
    "Using the deprecated option is not mathematically proven to converge. "
    "Please consider using the recommended condition instead."
)  # proximal_gradient.py:446-447


def _count_objectives(value) -> int:
    # proximal_gradient.py:143,467 - an ndarray return (even shape (1,)) sets m.
    return value.shape[0] if isinstance(value, np.ndarray) else 1


def momentum_sequence(n_terms: int, ratio=(0, 0.25)) -> np.ndarray:
    """beta_j, j = 1..n_terms: the factor applied after outer iteration j.

    proximal_gradient.py:531-535: t_j = sqrt(t_{j-1}^2 - a t_{j-1} + b) + 1/2,
    beta_j = (t_{j-1} - 1) / t_j, t_0 = 1 (so beta_1 == 0).  Depends on the
    iteration count only, never on data.
    """
    a, b = ratio
    out = np.empty(n_terms, dtype=np.float64)
    t_prev = 1
    for j in range(n_terms):
        t_next = np.sqrt(t_prev**2 - a * t_prev + b) + 0.5
        out[j] = (t_prev - 1) / t_next
        t_prev = t_next
    return out


def trial_single(f, g, jac_f, prox_wsum_g, lr, x_prev, y, deprecated=False):
    """One single-objective trial point and its model value (:140-157)."""
    f_y = f(y)
    F_prev = f(x_prev) + g(x_prev)
    grad = jac_f(y)
    x_new = prox_wsum_g(lr, y - lr * grad.flatten())
    step = x_new - y
    model = float(grad @ step + g(x_new) + np.linalg.norm(step) ** 2 / 2 / lr)
    if not deprecated:
        model += f_y - F_prev
    return OptimizeResult(x=x_new, fun=model, nit=1)


def dual_value_and_grad(w, g, prox_wsum_g, lr, y, J, f_y, F_prev, deprecated=False):
    """Negated dual of the scalarised subproblem and its gradient (:161-177)."""
    wJ = w @ J
    v = y - lr * wJ
    p = prox_wsum_g(lr * w, v)
    g_p = g(p)
    val = (
        -np.inner(w, g_p)
        - np.linalg.norm(p - v) ** 2 / 2 / lr
        + lr / 2 * np.linalg.norm(wJ) ** 2
    )
    grad = -g_p - J @ (p - y)
    if not deprecated:
        val += np.inner(w, F_prev - f_y)
        grad += F_prev - f_y
    return val, grad


def trial_multi(
    f, g, jac_f, prox_wsum_g, lr, x_prev, y, w0, tol=1e-12, max_iter=1000,
    deprecated=False,
):
    """Multi-objective trial: solve the simplex dual, recover the primal (:159-209)."""
    f_y = f(y)
    F_prev = f(x_prev) + g(x_prev)
    J = jac_f(y)
    m = _count_objectives(f_y)

    def dual(w):
        return dual_value_and_grad(w, g, prox_wsum_g, lr, y, J, f_y, F_prev, deprecated)

    if m == 2:
        sol = minimize_scalar(
            lambda s: dual(np.array([s, 1 - s]))[0],
            bounds=(0, 1),
            options={"maxiter": max_iter, "xatol": tol},
        )
        if not sol.success:
            warnings.warn(sol.message, stacklevel=2)
        weight = np.array([sol.x, 1 - sol.x])
    else:
        sol = minimize(
            fun=dual,
            x0=w0,
            method="trust-constr",
            jac=True,
            hess=BFGS(),
            bounds=Bounds(lb=0, ub=np.inf),
            constraints=LinearConstraint(np.ones(m), lb=1, ub=1),
            options={"gtol": tol, "xtol": tol, "barrier_tol": tol, "maxiter": max_iter},
        )
        if not sol.success:
            warnings.warn(sol.message, stacklevel=2)
        weight = sol.x
    x_new = prox_wsum_g(lr * weight, y - lr * weight @ J)
    return OptimizeResult(x=x_new, fun=-sol.fun, nit=sol.nit, weight=weight)


def trial(f, g, jac_f, prox_wsum_g, lr, x_prev, y, w0, tol, max_iter, deprecated, m):
    # :143-146 dispatches on type(f(y)); m is the same quantity taken from f(x0)
    # (:467), passed in so the callback count per trial equals the reference's.
    if m == 1:
        return trial_single(f, g, jac_f, prox_wsum_g, lr, x_prev, y, deprecated)
    return trial_multi(
        f, g, jac_f, prox_wsum_g, lr, x_prev, y, w0, tol, max_iter, deprecated
    )


def line_search(
    f, g, jac_f, prox_wsum_g, lr, x_prev, y, w0, tol, tol_internal,
    max_iter_internal, max_backtrack_iter, decay_rate, deprecated, warm_start, m, f_diff=None,
):
    """Shrink lr until the sufficient-decrease test holds (:279-308).

    Returns (x_new, lr, w0, subresult, n_trials)."""
    F_prev = f(x_prev) + g(x_prev)
    n_trials = 0
    accepted = False
    sub = None
    x_new = None
    while n_trials < max_backtrack_iter:
        n_trials += 1
        sub = trial(
            f, g, jac_f, prox_wsum_g, lr, x_prev, y, w0,
            tol_internal, max_iter_internal, deprecated, m,
        )
        x_new = sub.x
        F_new = f(x_new) + g(x_new)
        if w0 is not None and warm_start:
            w0 = sub.weight
        if decay_rate == 1:
            accepted = True
        elif f_diff is not None and m == 1:   # (extension, see the module docstring; not in the reference)
            df = f_diff(x_new, y)
            if deprecated:
                accepted = bool(df <= sub.fun + tol)
            else:
                step = x_new - y
                dot = float(jac_f(y).flatten() @ step)
                nrm = np.linalg.norm(step)
                accepted = bool((df - dot) - nrm * nrm / 2 / lr <= tol)
        elif deprecated:
            accepted = bool(np.all(f(x_new) - f(y) <= sub.fun + tol))
        else:
            accepted = bool(np.all(F_new - F_prev <= sub.fun + tol))
        if accepted:
            break
        lr *= decay_rate
    if not accepted:
        raise RuntimeError(MSG_BACKTRACK)
    return x_new, lr, w0, sub, n_trials


def minimize_proximal_gradient(
    f, g, jac_f, prox_wsum_g, x0,
    lr=1, tol=1e-5, tol_internal=1e-12, max_iter=1000000,
    max_iter_internal=100000, max_backtrack_iter=100, warm_start=False,
    decay_rate=0.5, nesterov=False, nesterov_ratio=(0, 0.25),
    return_all=False, verbose=False, deprecated=False, f_diff=None,
):
    """Oracle for zfista.minimize_proximal_gradient (proximal_gradient.py:311-555)."""
    if deprecated:
        warnings.warn(MSG_DEPRECATED, stacklevel=2)
    t_start = time.time()
    res = OptimizeResult(
        x0=x0, tol=tol, tol_internal=tol_internal,
        nesterov=nesterov, nesterov_ratio=nesterov_ratio,
    )
    if verbose:
        print("| niter | nit internal | max(abs(xk - yk)) | subprob func | learning rate |")
    x_prev = x_cur = y = x0
    f_x0 = f(x0)
    m = _count_objectives(f_x0)
    w0 = np.ones(m) / m if m > 1 else None
    t_prev = 1
    allvecs = [x0] if return_all else None
    allfuns = [f_x0 + g(x0)] if return_all else None
    allerrs = [] if return_all else None
    alllrs, alltrials = [], []
    finished = False
    nit = 0
    for nit in range(1, max_iter + 1):
        try:
            x_cur, lr, w0, sub, n_trials = line_search(
                f, g, jac_f, prox_wsum_g, lr, x_prev, y, w0,
                tol=tol_internal, tol_internal=tol_internal,
                max_iter_internal=max_iter_internal,
                max_backtrack_iter=max_backtrack_iter, decay_rate=decay_rate,
                deprecated=deprecated, warm_start=warm_start, m=m, f_diff=f_diff,
            )
        except Exception as exc:  # :493-509 - reported, not raised
            print(f"An error occurred: {exc}")
            failed = OptimizeResult()
            failed.update(
                success=False, message=f"Error: {str(exc)}", x=x_prev,
                fun=f(x_prev) + g(x_prev), nit=nit - 1,
                time=time.time() - t_start,
                allvecs=allvecs, allfuns=allfuns, allerrs=allerrs,
            )
            failed.alllrs, failed.alltrials = alllrs, alltrials
            return failed
        # :510 uses the builtin max over the array; the value is order-independent.
        err = np.max(np.abs(x_cur - y)) if x_cur.size else max(abs(x_cur - y))
        alllrs.append(lr)
        alltrials.append(n_trials)
        if verbose:
            print(f"|{nit:^7}|{sub.nit:^7}|{err:^+13.4e}|{sub.fun:^+13.4e}|{lr:^10.2e}|")
        if return_all:
            allvecs.append(x_cur)
            allfuns.append(f(x_cur) + g(x_cur))
            allerrs.append(err)
        if err < tol:  # strict, tested before the momentum update (:525)
            res.status, res.message, res.success = 1, MSG_SUCCESS, True
            finished = True
            break
        if nesterov:
            a, b = nesterov_ratio
            t_next = np.sqrt(t_prev**2 - a * t_prev + b) + 0.5
            beta = (t_prev - 1) / t_next
            y = x_cur + beta * (x_cur - x_prev)
            t_prev = t_next
        else:
            y = x_cur
        x_prev = x_cur
    if not finished:
        res.status, res.message, res.success = 0, MSG_MAXITER, False
        warnings.warn(res.message, stacklevel=2)
    res.update(
        x=x_cur, fun=f(x_cur) + g(x_cur), nit=nit,
        allvecs=allvecs, allfuns=allfuns, allerrs=allerrs,
        time=time.time() - t_start,
    )
    res.alllrs, res.alltrials = alllrs, alltrials
    return res
