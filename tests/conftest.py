import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Order of the run under `-x`: kernel parity against the reference's golden vectors first, the files that start
# other processes (transport rehearsals: a rendezvous, a torchrun, a subprocess) last - a hiccup of the transport
# can then never keep the hot path's own parity tests from running.  Files not named keep their alphabetical place
# in between.
_FIRST = ("test_gpu_parity_diag", "test_gpu_parity_lasso", "test_gpu_temporal", "test_gpu_runahead", "test_gpu_noise_floor",
          "test_gpu_fullsize_golden", "test_gpu_fuzz_parity", "test_gpu_multiobjective", "test_gpu_mo_fullsize",
          "test_gpu_problem_library", "test_gpu_operator_lasso", "test_gpu_tensor_callbacks", "test_gpu_sharded",
          "test_gpu_checkpoint", "test_gpu_cfg5_fullsize", "test_gpu_huge_n")
_LAST = ("test_gpu_harness", "test_gpu_integration_doc", "test_gpu_libcomm", "test_gpu_bench_contract",
         "test_gpu_multiprocess", "test_dist_gloo")


def _rank_of(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if name in _FIRST:
        return (0, _FIRST.index(name))
    if name in _LAST:
        return (2, _LAST.index(name))
    return (1, 0)


def pytest_collection_modifyitems(session, config, items):
    items.sort(key=_rank_of)        # stable: the order inside a file is the file's own


class Golden:
    """Prefix view over one committed .npz fixture."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name))

    def __call__(self, key):
        return self.z[key]

    def has(self, key):
        return key in self.z.files


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return get


def rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    den = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (den if den > 0 else 1.0)
