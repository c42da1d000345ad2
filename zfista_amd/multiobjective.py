"""Multi-objective trial (m >= 2): zfista/proximal_gradient.py:159-209."""
from __future__ import annotations


def trial_generic(ops, f, g, jac_f, prox, lr, x_old, y, w0, tol, max_iter, deprecated):
    raise NotImplementedError("multi-objective path: under construction in this build")
