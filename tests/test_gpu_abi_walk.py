"""tests/cpp/abi_walk.cpp: the C ABI entry points the other C++ drivers do not reach (voxel capacity with eviction and a
table rebuild, stream-ordered source, ndt_align_static, downsampled source, save / load, one-rank peer exchange, batch,
group), return codes and invariants only — the same driver runs under the host-side AddressSanitizer build
(tools/build_asan.sh, NDT_TEST_EXE_DIR=asan)."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "lio-slam_amd", os.environ.get("NDT_TEST_EXE_DIR", ""), "abi_walk")


def test_abi_walk():
    assert os.path.exists(EXE), "run __graft_entry__.build() first"
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "abi walk ok" in out.stdout, out.stdout + out.stderr
