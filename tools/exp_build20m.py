"""Map build at configs[3] size (20 M points): whole call and, under rocprofv3, per kernel."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, lio_slam_amd as la
w = bench.load_workload(2, 0, 1, None, False)
m, s = bench.load_dense_workload(w, 0, 1, None)
n = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
for _ in range(2):
    n.set_target(m); n.synchronize()
t0 = time.perf_counter(); n.set_target(m); n.synchronize(); print("set_target(20M) %.3f ms, %d voxels" % (1e3 * (time.perf_counter() - t0), n.num_voxels()))
