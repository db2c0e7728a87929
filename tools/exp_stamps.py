"""Phase timing of the chained single-scan launch (diagnostic build libndt_mi355_stamps.so: make -C lio-slam_amd/csrc ../libndt_mi355_stamps.so).
Prints, for blocks 0 and 100 of the last launch, the cycles between s_memtime stamps: start -> rows summed -> H,b -> step -> residual pass -> row stored."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lio_slam_amd as la
la.LIB_PATH = os.path.join(ROOT, "lio-slam_amd", "libndt_mi355_stamps.so")
import bench
import numpy as np
w = bench.load_workload(2, 0, 1, None, False)
ndt = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
ndt.set_target(w["map"]); ndt.set_source(w["scan0"])
for _ in range(5):
    r = ndt.align(w["guess0"])
print("chain us", ndt.time_chain(w["guess0"], 200), "residual us", ndt.time_residual(w["guess0"], 200))
r = ndt.align(w["guess0"])
buf = (ctypes.c_ulonglong * 16)()
assert la.lib().ndt_debug_stamps(buf) == 0
st = np.array(list(buf), dtype=np.int64).reshape(2, 8)
names = ["start->rows summed", "->H,b entries", "->GN step", "->residual pass", "->row stored"]
for b in range(2):
    d = np.diff(st[b, :6])
    print("  inside wave 0: tree", int(st[b, 6] - st[b, 1]), "H,b", int(st[b, 7] - st[b, 6]), "step", int(st[b, 2] - st[b, 7]))
    print("block", [0, 100][b], {n: int(v) for n, v in zip(names, d)}, "total cycles", int(st[b, 5] - st[b, 0]), "(100 MHz ticks x ? -> see s_memtime unit)")

wb = (ctypes.c_ulonglong * 32)()
if hasattr(la.lib(), "ndt_debug_wstamps") and la.lib().ndt_debug_wstamps(wb) == 0:
    ws = np.array(list(wb), dtype=np.int64).reshape(8, 4)
    t0 = ws[:, 0].min()
    print("block 100, per wave (cycles after the first wave entered the pass): pass start / pass end / row stored")
    for w_ in range(8):
        print("  wave", w_, int(ws[w_, 0] - t0), int(ws[w_, 1] - t0), int(ws[w_, 2] - t0))
