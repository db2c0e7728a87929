"""Host-side cost of one frame of the online stream: the hand-over call, the align call, and an align with nothing to do."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, lio_slam_amd as la
w = bench.load_workload(2, 0, 1, None, False)
L = la.lib()
dev = [bench.DeviceCloud(w[f"scan{j}"]) for j in range(8)]
g = [np.ascontiguousarray(w[f"guess{j}"].T).reshape(16) for j in range(8)]
gp = [ctypes.c_void_p(x.ctypes.data) for x in g]
dp = [(d.ptr, ctypes.c_size_t(d.n), ctypes.c_size_t(12)) for d in dev]
res = la.Result(); rr = ctypes.byref(res)
pc = time.perf_counter
for iters in (10, 1):
    ndt = la.NDT(max_iterations=iters, eps=0.0, final_eval=0)
    ndt.set_target(w["map"]); h = ndt._h
    for f in range(60):
        j = f % 8; L.ndt_set_source_async(h, *dp[j]); L.ndt_align(h, gp[j], rr)
    ts, ta = [], []
    for f in range(400):
        j = f % 8
        t0 = pc(); L.ndt_set_source_async(h, *dp[j]); t1 = pc(); L.ndt_align(h, gp[j], rr); t2 = pc()
        ts.append(t1 - t0); ta.append(t2 - t1)
    print(f"max_iterations {iters}: hand-over call {1e6 * np.median(ts):.2f} us, align call {1e6 * np.median(ta):.2f} us (median of 400)")
    # align alone on an unchanged source (no hand-over in between)
    ta = []
    for f in range(400):
        t1 = pc(); L.ndt_align(h, gp[0 if False else (399 % 8)], rr); ta.append(pc() - t1)
    print(f"   align without a hand-over: {1e6 * np.median(ta):.2f} us")
    ndt.close()
t0 = pc()
for _ in range(100000): pass
print("python loop overhead per iteration ns", 1e9 * (pc() - t0) / 100000)
