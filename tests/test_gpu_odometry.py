"""SURVEY.md §8(f) row N3 (LiDAR half): the per-frame routine of process.cc:60-153 through the C++ ScanToMapOdometry
shell, against the same loop written with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, pose_error, TIGHT_T, TIGHT_R

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "lio-slam_amd", os.environ.get("NDT_TEST_EXE_DIR", ""), "odometry_test")   # (NDT_TEST_EXE_DIR=asan: the host-ASan builds of tools/build_asan.sh)


def test_frame_loop_matches_oracle(tmp_path):
    from lio_slam_amd import synth
    from oracle.oracle import Oracle, voxel_downsample
    assert os.path.exists(EXE), "run __graft_entry__.build() first"
    rng = np.random.Generator(np.random.PCG64(31))
    scene = synth.room_scene(rng)
    T_il = synth.se3([0.0, 0.0, 0.02], [0.05, 0.0, 0.15])                      # LiDAR in the IMU/body frame
    poses = [synth.se3([0, 0, 0.015 * k], [0.22 * k, 0.03 * k, 0.5]) for k in range(10)]   # body poses, ~0.22 m apart
    raws = [synth.scan(scene, "vlp16", T @ T_il, rng) for T in poses]
    # the predicted pose of every frame: truth with an IMU-sized error (a few cm / tenths of a degree).  With the
    # reference's own 0.5 m scan leaf the 1 m NDT cells rarely reach 6 points and tracking is 10-20 cm off (its
    # author notes the poor matching at process.cc:95), so the test runs at a 0.2 m leaf.
    leaf = 0.2
    preds = [poses[0]] + [T @ synth.perturbation(rng, 0.01, 0.05, 0.1, 0.5) for T in poses[1:]]
    path = tmp_path / "seq.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<Qd", len(raws), leaf)); f.write(np.ascontiguousarray(T_il.T).tobytes())
        for r, T in zip(raws, preds):
            f.write(struct.pack("<Q", len(r))); f.write(np.ascontiguousarray(r, np.float32).tobytes())
            f.write(np.ascontiguousarray(T.T).tobytes())
    out = subprocess.run([EXE, str(path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rows = [l.split() for l in out.stdout.strip().split("\n")]
    assert len(rows) == len(raws)

    # the same loop on the CPU oracle
    o = Oracle()
    last_key = poses[0]
    o.add_target(voxel_downsample(raws[0], leaf, T_il), poses[0])
    n_key = 1
    for k in range(1, len(raws)):
        guess = preds[k]
        ds = voxel_downsample(raws[k], leaf, T_il)
        o.set_source(ds)
        r = o.align(guess)
        assert r["rc"] == 0
        pose = r["T"]
        dp = np.linalg.norm(pose[:3, 3] - last_key[:3, 3])
        dth = np.arccos(min(1.0, max(-1.0, 0.5 * (np.trace(last_key[:3, :3].T @ pose[:3, :3]) - 1.0))))
        key = dp > 0.3 or dth > 5.0 * 3.14 / 180.0
        if key:
            o.add_target(ds, pose); last_key = pose; n_key += 1
        aligned, gkey, iters = int(rows[k][0]), int(rows[k][1]), int(rows[k][2])
        Tg = np.array(rows[k][3:], float).reshape(4, 4).T
        assert aligned == 1 and gkey == int(key) and iters == r["iterations"], (k, rows[k][:3], key, r["iterations"])
        dt, dr = pose_error(Tg, pose)
        assert dt < TIGHT_T and dr < TIGHT_R, (k, dt, dr)             # measured 3e-16 m over the whole drive
        et, er = pose_error(Tg, poses[k])
        assert et < 0.02 and er < 0.005, (k, et, er)                              # and the truth is tracked
    assert int(rows[0][0]) == 0 and n_key >= 3                                   # several keyframes were merged
