"""Where one frame of bench.py's online stream spends its time: set source / align / keyframe merge, each timed alone."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, lio_slam_amd as la
w = bench.load_workload(2, 0, 1, None, False)
ndt = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
ndt.set_target(w["map"])
dev = bench.DeviceCloud(w["scan0"])
L = la.lib(); h = ndt._h; res = la.Result()
g = np.ascontiguousarray(w["guess0"].T).reshape(16); Tk = np.ascontiguousarray(w["T0"].T).reshape(16)
def t(f, n=300):
    for _ in range(20): f()
    ndt.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    ndt.synchronize(); return 1e6 * (time.perf_counter() - t0) / n
print("set_source (device records) us", t(lambda: L.ndt_set_source(h, dev.ptr, dev.n, 12)))
print("align us", t(lambda: L.ndt_align(h, g.ctypes.data, ctypes.byref(res))))
print("merge from source us", t(lambda: L.ndt_add_target_from_source(h, Tk.ctypes.data), 50))
print("chain kernel us", ndt.time_chain(w["guess0"], 300), "residual kernel us", ndt.time_residual(w["guess0"], 300))
ndt.configure(max_iterations=1)
print("align, 1 iteration us", t(lambda: L.ndt_align(h, g.ctypes.data, ctypes.byref(res))))
ndt.configure(max_iterations=20)
print("align, 20 iterations us", t(lambda: L.ndt_align(h, g.ctypes.data, ctypes.byref(res))))
