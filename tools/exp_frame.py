"""Host-side split of one online frame (ctypes calls timed one by one, 2000 frames): stream-ordered hand-over, align, and
the loop as bench.py runs it."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, lio_slam_amd as la
w = bench.load_workload(2, 0, 1, None, False)
ndt = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
ndt.set_target(w["map"])
d = bench.DeviceCloud(w["scan0"])
L = la.lib(); h = ndt._h
res = la.Result(); rr = ctypes.byref(res)
g = np.ascontiguousarray(w["guess0"].T).reshape(16); gp = ctypes.c_void_p(g.ctypes.data)
args = (d.ptr, ctypes.c_size_t(d.n), ctypes.c_size_t(12))
for _ in range(50):
    L.ndt_set_source_async(h, *args); L.ndt_align(h, gp, rr)
N = 2000
ts = ta = 0.0
t0 = time.perf_counter()
for _ in range(N):
    a = time.perf_counter(); L.ndt_set_source_async(h, *args); b = time.perf_counter(); L.ndt_align(h, gp, rr); c = time.perf_counter()
    ts += b - a; ta += c - b
tot = time.perf_counter() - t0
print("set_source_async call us %.2f  align call us %.2f  loop us %.2f" % (1e6 * ts / N, 1e6 * ta / N, 1e6 * tot / N))
t0 = time.perf_counter()
for _ in range(N):
    L.ndt_set_source_async(h, *args); L.ndt_align(h, gp, rr)
print("untimed-inside loop us %.2f" % (1e6 * (time.perf_counter() - t0) / N))
