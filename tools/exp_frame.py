"""Per-frame cost of the online loop on configs[2]-sized data: pre-process, align, keyframe merge."""
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
    t0 = time.perf_counter()
    for _ in range(reps): f()
    n.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps
print("set_source (117k, H2D + unpack)      %.3f ms" % t(lambda: n.set_source(raw)))
print("set_source_downsampled leaf 0.5      %.3f ms -> %d pts" % (t(lambda: n.set_source_downsampled(raw, 0.5)), n.source_size()))
print("set_source_downsampled leaf 0.2      %.3f ms -> %d pts" % (t(lambda: n.set_source_downsampled(raw, 0.2)), n.source_size()))
print("align (downsampled 0.2)              %.3f ms" % t(lambda: n.align(w["guess"])))
print("add_target_from_source (0.2)         %.3f ms" % t(lambda: n.add_target_from_source(w["T_true"]), 10))
n.set_source(raw)
print("align (117k)                         %.3f ms" % t(lambda: n.align(w["guess"])))
print("add_target (117k host pts)           %.3f ms" % t(lambda: n.add_target(raw, w["T_true"]), 10))
print("set_target (2M)                      %.3f ms" % t(lambda: n.set_target(w["map"]), 5))
