"""Per-frame and per-merge times of the online stream from a fresh map (why the driver's 20-after-5 window reads lower than the steady state)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, lio_slam_amd as la
w = bench.load_workload(2, 0, 1, None, False)
ndt = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
dev = [bench.DeviceCloud(w[f"scan{j}"]) for j in range(8)]
L = la.lib(); h = ndt._h
res = la.Result(); rr = ctypes.byref(res)
g = [np.ascontiguousarray(w[f"guess{j}"].T).reshape(16) for j in range(8)]
T = [np.ascontiguousarray(w[f"T{j}"].T).reshape(16) for j in range(8)]
if os.environ.get("NDT_DEBUG"): pass
for rep in range(2):
    ndt.set_target(w["map"]); ndt.synchronize()
    print("rep", rep, "voxels", ndt.num_voxels())
    ft, mt = [], []
    for f in range(45):
        j = f % 8
        t0 = time.perf_counter()
        rc = L.ndt_set_source_async(h, dev[j].ptr, ctypes.c_size_t(dev[j].n), ctypes.c_size_t(12))
        rc = rc or L.ndt_align(h, ctypes.c_void_p(g[j].ctypes.data), rr)
        t1 = time.perf_counter()
        ft.append(1e6 * (t1 - t0))
        if f % 10 == 9:
            rc = rc or L.ndt_add_target_from_source(h, ctypes.c_void_p(T[j].ctypes.data))
            mt.append(1e6 * (time.perf_counter() - t1))
        assert rc == 0
    print("  align us per frame:", " ".join(f"{x:.0f}" for x in ft))
    print("  merge us:", " ".join(f"{x:.0f}" for x in mt), "voxels", ndt.num_voxels())
