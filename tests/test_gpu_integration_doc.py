"""GPU: the reference-side binding printed in INTEGRATION.md (section 2) is executed as written
- a ctypes stub over the C ABI with no help from zfista_amd's host code - and must reproduce the
oracle's iterate bit for bit."""
import os
import re
import warnings

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_integration_md_stub_runs_as_written():
    import torch  # noqa: F401  (one HIP runtime in the process)

    from oracle import cpu_ref, problems_ref as P
    from zfista_amd import _lib
    from zfista_amd.engine import momentum_factors

    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.findall(r"```python\n(.*?)```", md, re.S)[0]
    assert 'C.CDLL("libzfista_hip.so")' in code
    ns = {}
    exec(code.replace('C.CDLL("libzfista_hip.so")', f'C.CDLL({_lib.LIB_PATH!r})'), ns)
    n = 10007
    d, c, lam = P.make_pdiag(n, seed=1)
    betas = np.concatenate([[0.0], momentum_factors(60, (0, 0.25))[0]])
    x, trace = ns["solve_diag_l1"](d, c, lam, np.zeros(n), betas, lr=0.45, tol=0.0, tol_internal=1e-12, decay_rate=0.5,
                                   max_iter=40, max_backtrack_iter=100, nesterov=1, deprecated=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = cpu_ref.minimize_proximal_gradient(*P.DiagQuadL1Ref(d, c, lam).callbacks(), np.zeros(n), lr=0.45,
                                                 nesterov=True, tol=0.0, max_iter=40, return_all=True)
    assert np.array_equal(x, exp.x)
    assert int((trace[:, 2] > 0).sum()) == 40
    np.testing.assert_allclose(trace[:40, 1], exp.allfuns[1:], rtol=1e-10)


def test_quickstart_example_runs():
    import subprocess
    import sys

    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "quickstart.py")], capture_output=True,
                         text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 5 and "Optimization terminated successfully" in lines[0]
    assert float(lines[0].split("|x - x*|_inf=")[1].split()[0]) < 1e-6
    assert float(lines[2].split("|dx|=")[1]) < 1e-8
