"""The online stream (async hand-over, align, keyframe merge every 10th frame) for N frames; writes a digest of every
result.  Run once as is and once with NDT_NO_POLL=1 NDT_NO_MIRROR=1: the digests must be identical."""
import ctypes, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, lio_slam_amd as la
N = int(sys.argv[1]); out = sys.argv[2]
w = bench.load_workload(2, 0, 1, None, False)
ndt = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
ndt.set_target(w["map"])
dev = [bench.DeviceCloud(w[f"scan{j}"]) for j in range(8)]
L = la.lib(); h = ndt._h
res = la.Result(); rr = ctypes.byref(res)
g = [np.ascontiguousarray(w[f"guess{j}"].T).reshape(16) for j in range(8)]
T = [np.ascontiguousarray(w[f"T{j}"].T).reshape(16) for j in range(8)]
hsh = hashlib.sha256()
for f in range(N):
    j = f % 8
    if os.environ.get("SOAK_BLOCKING"):
        rc = L.ndt_set_source(h, dev[j].ptr, ctypes.c_size_t(dev[j].n), ctypes.c_size_t(12))
    else:
        rc = L.ndt_set_source_async(h, dev[j].ptr, ctypes.c_size_t(dev[j].n), ctypes.c_size_t(12))
    rc = rc or L.ndt_align(h, ctypes.c_void_p(g[j].ctypes.data), rr)
    assert rc == 0, (f, rc)
    hsh.update(bytes(res))
    if f % 10 == 9:
        rc = L.ndt_add_target_from_source(h, ctypes.c_void_p(T[j].ctypes.data))
        assert rc == 0, (f, rc)
        hsh.update(str(ndt.num_voxels()).encode())
open(out, "w").write(hsh.hexdigest() + "\n")
print("frames", N, "voxels", ndt.num_voxels(), hsh.hexdigest()[:16])
