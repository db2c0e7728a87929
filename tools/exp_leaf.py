"""Downsample timing vs leaf size and call order (is leaf 0.5 slower, or is whatever runs first slower?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, lio_slam_amd as la
w = bench.load_workload(2, 0, 1, None, False)
n = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
n.set_target(w["map"])
raw = np.ascontiguousarray(w["scan"], np.float32)
def t(f, reps=30):
    f(); n.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); n.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    ts = np.array(ts)
    return "median %.3f  min %.3f  max %.3f ms" % (np.median(ts), ts.min(), ts.max())
for leaf in (0.2, 0.5, 0.2, 0.5, 1.0, 0.5, 0.3, 0.5):
    print("leaf %.1f  %s -> %d pts" % (leaf, t(lambda: n.set_source_downsampled(raw, leaf)), n.source_size()), flush=True)
