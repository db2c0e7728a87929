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


# The bars.  NORTH_STAR_* are what BASELINE.json's north_star demands of the path.  TIGHT_* are what the engine is held to
# against the oracle and against the NumPy golden vectors, both of which round the table payload to float32 exactly as the
# engine does: set in round 3 from tools/exp_parity_margins.py (H, b, chi2 and the PCL-mode score / gradient / Hessian
# agree to 7e-15 ... 2e-13, aligned poses to 1e-14 m), with room for ONE float32 rounding of ONE table entry landing on the
# other side (6e-8 of one voxel's term among thousands).  Against the pure-float64 (q0) vectors the float32 payload itself
# shows: 1e-9 ... 4e-8, bar Q0_H.
NORTH_STAR_H, NORTH_STAR_T, NORTH_STAR_R = 1e-5, 1e-4, 1e-5
TIGHT_H, TIGHT_T, TIGHT_R = 1e-10, 1e-9, 1e-9
Q0_H = 1e-6


def pose_error(T, T_ref):
    """(translation error in metres, rotation error in radians) between two 4x4 poses.  The angle comes from atan2 of the
    skew part's norm and the trace (arccos of the trace alone cannot resolve angles below 1e-8)."""
    d = np.linalg.inv(T_ref) @ T
    R = d[:3, :3]
    s = 0.5 * float(np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]))
    c = 0.5 * (float(np.trace(R)) - 1.0)
    return float(np.linalg.norm(d[:3, 3])), float(np.arctan2(s, c))


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
