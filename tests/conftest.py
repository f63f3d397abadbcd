"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path."""
import json
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
    """Loader for tests/golden (data minted from the reference's check.py; see make_golden.py)."""

    def __init__(self):
        with open(os.path.join(GOLDEN, "manifest.json")) as f:
            self.manifest = json.load(f)["fixtures"]

    def meta(self, name):
        return self.manifest[name]

    def load(self, name, key):
        entry = self.manifest[name]["files"][key]
        return np.load(os.path.join(GOLDEN, entry["file"]), allow_pickle=False)

    @staticmethod
    def to_bhsd(x, num_heads):
        """(B,S,H*d_k) -> [B,H,S,d_k] (check.py:14-16)."""
        B, S, dm = x.shape
        return np.ascontiguousarray(x.reshape(B, S, num_heads, dm // num_heads).transpose(0, 2, 1, 3))

    @staticmethod
    def to_bsd(x):
        """[B,H,S,d_k] -> (B,S,H*d_k) (check.py:24)."""
        B, H, S, d = x.shape
        return np.ascontiguousarray(x.transpose(0, 2, 1, 3).reshape(B, S, H * d))


@pytest.fixture(scope="session")
def golden():
    return Golden()
