"""How much of a replay step is the Python binding (argument marshalling + result dicts)?"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, lio_slam_amd as la
from lio_slam_amd import synth
w = bench.load_workload(2, 0, 1, None, False)
n = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
n.set_target(w["map"])
scans = [w[f"scan{j}"] for j in range(8)]
B = 64
rng = np.random.default_rng(0)
pair_scan = np.asarray([p % 8 for p in range(B)], np.int32)
guesses = np.stack([w[f"T{p % 8}"] @ synth.perturbation(rng) for p in range(B)])
n.set_batch_sources(scans)
for _ in range(5): n.align_batch_resident(guesses, pair_scan)
t0 = time.perf_counter()
for _ in range(50): n.align_batch_resident(guesses, pair_scan)
n.synchronize(); a = 1e3 * (time.perf_counter() - t0) / 50
g = np.ascontiguousarray(guesses.transpose(0, 2, 1)).reshape(B, 16)
res = np.zeros(B, la.RESULT_DTYPE)
lib = la.lib()
t0 = time.perf_counter()
for _ in range(50):
    lib.ndt_align_batch_resident(n._h, pair_scan.ctypes.data, B, g.ctypes.data, C.cast(res.ctypes.data, C.POINTER(la.Result)))
n.synchronize(); b = 1e3 * (time.perf_counter() - t0) / 50
print("python method %.4f ms, bare C call %.4f ms per step of %d pairs" % (a, b, B))
n.set_source(w["scan"])
for _ in range(20): n.align(w["guess"])
t0 = time.perf_counter()
for _ in range(300): n.align(w["guess"])
a = 1e3 * (time.perf_counter() - t0) / 300
t = np.ascontiguousarray(np.asarray(w["guess"], np.float64).T).reshape(16)
r = la.Result()
t0 = time.perf_counter()
for _ in range(300): lib.ndt_align(n._h, t.ctypes.data, C.byref(r))
b = 1e3 * (time.perf_counter() - t0) / 300
print("single scan: python method %.4f ms, bare C call %.4f ms" % (a, b))
