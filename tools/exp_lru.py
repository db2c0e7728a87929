"""Cost of ndt_config.capacity per keyframe merge: configs[2] map (21.6 k voxels), 118 k-point keyframes, with the map held
at `capacity` voxels against the unbounded map."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, lio_slam_amd as la
w = bench.load_workload(2, 0, 1, None, False)
d = bench.DeviceCloud(w["map"])
for cap in (0, 21000, 15000):
    n = la.NDT(max_iterations=10, eps=0.0, final_eval=0, capacity=cap)
    n.set_target(d.view()); n.synchronize()
    v0 = n.num_voxels()
    ts = []
    for j in range(16):
        n.set_source(w[f"scan{j % 8}"]); n.synchronize()
        t0 = time.perf_counter(); n.add_target_from_source(w[f"T{j % 8}"]); n.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print("capacity %6d: voxels %d -> %d, merge ms min %.4f median %.4f" % (cap, v0, n.num_voxels(), min(ts[3:]), float(np.median(ts[3:]))))
    n.close()
