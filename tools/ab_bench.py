"""A/B helper: runs bench.py against another build of the library (same ABI), e.g. one compiled with different flags.
usage: python tools/ab_bench.py <path/to/lib.so> [bench.py arguments]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lio_slam_amd as la  # noqa: E402

la.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
