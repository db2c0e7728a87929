"""Soak: a few thousand mixed calls (downsample, align, merge, batch, PCL mode, save/load, re-target) on one handle and on
short-lived handles; free device memory must settle (grow-only buffers, no leak per call or per handle)."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lio_slam_amd as la
from lio_slam_amd import synth
hip = ctypes.CDLL("libamdhip64.so")
def free_mb():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value / 2**20
c = synth.make_config(1, small=True)
rng = np.random.default_rng(0)
n = la.NDT(max_iterations=8, eps=1e-3)
n.set_target(c["map"])
marks = []
t0 = time.time()
for it in range(1500):
    k = int(rng.integers(2000, len(c["scan"])))
    s = c["scan"][rng.permutation(len(c["scan"]))[:k]]
    g = c["T_true"] @ synth.perturbation(rng)
    op = it % 6
    if op == 0:
        n.set_source_downsampled(s, float(rng.choice([0.2, 0.5])))
        r = n.align(g, check=False)
    elif op == 1:
        n.set_source(s); r = n.align(g, check=False); n.add_target(s[::7], r["T"])
    elif op == 2:
        r = n.align_batch([s[: k // 2], s[k // 2:]], [g, c["T_true"], g], [0, 1, 1])[0]
    elif op == 3:
        n.set_source(s); r = n.pcl_align(g)
    elif op == 4:
        blob = n.save_map(); n.load_map(blob); n.set_source(s); r = n.align(g, check=False)
    else:
        m = la.NDT(); m.set_target(c["map"][::3]); m.set_source(s); r = m.align(g, check=False); m.close()
    assert np.all(np.isfinite(r["T"]))
    if it % 250 == 249:
        n.synchronize(); marks.append(free_mb()); print(it + 1, "calls, free MB", round(marks[-1], 1), "voxels", n.num_voxels(), flush=True)
    if it == 749:
        n.set_target(c["map"])                      # start the map over
print("elapsed %.1f s" % (time.time() - t0))
assert abs(marks[-1] - marks[2]) < 64, marks       # settled after the first few hundred calls
print("soak ok")
