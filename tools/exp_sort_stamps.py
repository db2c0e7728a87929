"""Phase timing of the one-sweep radix pass (diagnostic build libndt_mi355_stamps.so: make -C lio-slam_amd/csrc ../libndt_mi355_stamps.so).
Every tile of the last pass of a 2 M-point build leaves wall_clock64 stamps (100 MHz, one clock for the whole device): entry, ticket known, elements ranked (loads
included; its counts are published right after), look-back done (thread 0's digit), block released, scattered.  Prints the
distribution over the tiles, relative to the earliest entry."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lio_slam_amd as la
la.LIB_PATH = os.path.join(ROOT, "lio-slam_amd", "libndt_mi355_stamps.so")
import bench
import numpy as np
w = bench.load_workload(2, 0, 1, None, False)
ndt = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
d = bench.DeviceCloud(w["map"])
for _ in range(3):
    ndt.set_target(d.view()); ndt.synchronize()
nt = (len(w["map"]) + 4095) // 4096
if len(sys.argv) > 1 and sys.argv[1] == "merge":       # a keyframe merge instead: 1024-element tiles
    ndt.set_source(w["scan0"])
    for _ in range(3):
        ndt.add_target_from_source(w["T0"]); ndt.synchronize()
    nt = (len(w["scan0"]) + 1023) // 1024
buf = (ctypes.c_ulonglong * (8 * nt))()
assert la.lib().ndt_debug_os_stamps(buf, nt) == 0
st = np.array(list(buf), dtype=np.int64).reshape(nt, 8)[:, :6]
st = st - st[:, 0].min()
names = ["entry", "ticket", "ranked", "lookback", "released", "scattered"]
for k, nm in enumerate(names):
    c = st[:, k]
    print(f"{nm:10s} min {c.min():7d} p10 {int(np.percentile(c,10)):7d} median {int(np.median(c)):7d} p90 {int(np.percentile(c,90)):7d} max {c.max():7d}  (tile of max: {int(c.argmax())})")
print("10 ns ticks per tile phase (median):", [int(np.median(st[:, k + 1] - st[:, k])) for k in range(5)])
late = np.argsort(st[:, 2])[-8:]
print("latest to publish counts:", [(int(t), int(st[t, 0]), int(st[t, 1]), int(st[t, 2])) for t in late])
np.save("gpurun_out/os_stamps.npy", st)
