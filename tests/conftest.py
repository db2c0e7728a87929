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


@pytest.fixture(scope="session")
def golden():
    return {n: np.load(os.path.join(GOLDEN, n + ".npz")) for n in ("keys", "voxels", "pair", "downsample", "pcl", "config1", "yard")}


def pose_error(T, T_ref):
    """(translation error in metres, rotation error in radians) between two 4x4 poses."""
    d = np.linalg.inv(T_ref) @ T
    c = min(1.0, max(-1.0, (np.trace(d[:3, :3]) - 1.0) / 2.0))
    return float(np.linalg.norm(d[:3, 3])), float(np.arccos(c))


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
