import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
P = 2147483647


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN, "hotpath_golden.json")) as f:
        return json.load(f)


def load_vectors(name):
    with open(os.path.join(GOLDEN, f"{name}-test-vectors.json")) as f:
        return json.load(f)["test_vectors"]


class SplitMix64:
    """Same generator as tests/golden/gen_golden.py and bench.py (SURVEY.md §8d inputs)."""

    def __init__(self, seed):
        self.s = seed & (2**64 - 1)

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return z ^ (z >> 31)

    def m31(self, nonzero=False):
        while True:
            v = self.next() >> 33
            if v < P and not (nonzero and v == 0):
                return v


def column(seed, n, nonzero=False):
    r = SplitMix64(seed)
    return np.array([r.m31(nonzero) for _ in range(n)], dtype=np.uint32)


def rand_column(seed, n, nonzero=False):
    """Fast numpy column for large sizes (uniform in [0,P) / [1,P))."""
    rng = np.random.default_rng(seed)
    lo = 1 if nonzero else 0
    return rng.integers(lo, P, size=n, dtype=np.uint32)


def golden_interp_values(e):
    """Input of a cfft_interpolate golden entry: inline list (log <= 5) or base64 of the LE32 words."""
    import base64
    if "values" in e:
        return np.array(e["values"], dtype=np.uint32)
    return np.frombuffer(base64.b64decode(e["values_b64"]), dtype="<u4").astype(np.uint32)
