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
