"""Soak of the stream-ordered hand-over + polled result: N frames cycling over 8 scans against an unchanged map; every
result must equal, bit for bit, the first result of its scan (a stale mirror read or a missed unpack would show here)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, lio_slam_amd as la
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
w = bench.load_workload(2, 0, 1, None, False)
ndt = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
ndt.set_target(w["map"])
dev = [bench.DeviceCloud(w[f"scan{j}"]) for j in range(8)]
L = la.lib(); h = ndt._h
res = la.Result(); rr = ctypes.byref(res)
g = [np.ascontiguousarray(w[f"guess{j}"].T).reshape(16) for j in range(8)]
gp = [ctypes.c_void_p(x.ctypes.data) for x in g]
args = [(d.ptr, ctypes.c_size_t(d.n), ctypes.c_size_t(12)) for d in dev]
first = {}
bad = 0
t0 = time.time()
for f in range(N):
    j = (f * 5 + f // 7) % 8
    rc = L.ndt_set_source_async(h, *args[j]) or L.ndt_align(h, gp[j], rr)
    assert rc == 0, rc
    key = bytes(res)
    if j not in first:
        first[j] = key
    elif key != first[j]:
        bad += 1
        if bad < 5:
            print("MISMATCH frame", f, "scan", j, flush=True)
    if f % 20000 == 0:
        print("frame", f, "elapsed %.1f s" % (time.time() - t0), "mismatches", bad, flush=True)
print("done", N, "frames, mismatches", bad)
sys.exit(1 if bad else 0)
