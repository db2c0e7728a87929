"""Profiling driver for the per-frame pre-processing / merge path (used under rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, lio_slam_amd as la
w = bench.load_workload(2, 0, 1, None, False)
n = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
n.set_target(w["map"])
raw = np.ascontiguousarray(w["scan"], np.float32)
for _ in range(30):
    n.set_source_downsampled(raw, 0.5)
for _ in range(10):
    n.add_target_from_source(w["T_true"])
print("done", n.source_size(), n.num_voxels())
