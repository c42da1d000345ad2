"""GPU: a decision vector of more than 2^31 elements on one MI355X (2.2e9 doubles = 17.6 GB per vector, ~125 GB in
all): the 64-bit indexing of the chained trial kernel and of its launch geometry.  tools/check_huge_n.py compares a
strided sample of x_16 - the last five elements and the neighbours of 2^31 and 2^32 included - with the oracle's
element recursion (P-diag is separable; lr = 0.45 accepts every trial), bit for bit."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_more_than_two_to_the_31_elements():
    import torch

    torch.cuda.empty_cache()   # (what earlier tests of this process left in torch's caching allocator)
    free, _ = torch.cuda.mem_get_info()
    if free < 150 * 2**30:
        pytest.skip("needs ~125 GB of free HBM")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_huge_n.py"), "2.2e9"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "equals the oracle recursion bit for bit" in out.stdout
