"""The oracle is compiled code, and a compiler can get it wrong: gcc 11.4's vectoriser at -O3 -march=x86-64-v3 dropped two of
three float32 roundings in finalize_voxel until round 3.  The library the tests use (the Makefile's -O3 build)
must compute, bit for bit, what an unoptimised -O0 build of the same source computes."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_optimised_build_equals_unoptimised_build(tmp_path):
    src = os.path.join(ROOT, "oracle", "ndt_oracle.c")
    lib3 = os.path.join(ROOT, "oracle", "libndt_oracle.so")
    assert os.path.exists(lib3), "run __graft_entry__.build() first"
    lib0 = str(tmp_path / "libndt_oracle_O0.so")
    subprocess.run(["gcc", "-O0", "-ffp-contract=off", "-fPIC", "-std=c11", "-D_GNU_SOURCE", f"-I{os.path.join(ROOT, 'oracle')}", "-shared", "-o", lib0,
                    src, "-lm", "-lpthread"], check=True, cwd=os.path.join(ROOT, "oracle"))
    digest = [subprocess.run([sys.executable, os.path.join(ROOT, "tests", "oracle_digest.py"), lib], check=True, capture_output=True, text=True,
                             timeout=300).stdout.strip() for lib in (lib3, lib0)]
    assert len(digest[0]) == 64 and digest[0] == digest[1]
