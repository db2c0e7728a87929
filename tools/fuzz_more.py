"""Runs the randomised differential tests of tests/test_gpu_fuzz.py over many more seeds (not part of the suite)."""
import sys, os, time, traceback
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import test_gpu_fuzz as f
lo, hi = int(sys.argv[1]), int(sys.argv[2])
orig = np.random.default_rng
fails = 0
t0 = time.time()
for seed in range(lo, hi):
    for name, fn, base in (("scene", f.test_random_scene_and_config, 1000), ("down", f.test_random_downsample, 2000), ("pcl", f.test_random_pcl_mode, 3000)):
        try:
            fn(seed)
        except Exception:
            fails += 1
            print("FAIL", name, seed, flush=True)
            traceback.print_exc(limit=3)
    if seed % 20 == 0:
        print("seed", seed, "elapsed %.0f s" % (time.time() - t0), "fails", fails, flush=True)
print("done", lo, hi, "fails", fails)
