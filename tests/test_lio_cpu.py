"""SURVEY.md §8(f) row N3, IMU / optimiser half: include/ndt_mi355_lio.hpp (C++, through tests/cpp/lio_test.cpp)
against oracle/lio_oracle.py, the NumPy restatement of integration_base.h / factor.h / imu_init.cc / utility.cc /
process.cc:155-258.  Host-only code: runs without a GPU.  Parity with the reference itself is unpinned (it needs
Eigen + Ceres + ROS, absent here, and holds no vectors); the departures both sides share are listed in the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from lio_slam_amd import synth
from oracle import lio_oracle as lo

EXE = os.path.join(ROOT, "lio-slam_amd", "lio_test")
D = np.array([0.02, -0.015, 0.01, 0.004, -0.003, 0.006, 0.03, -0.02, 0.01, 1e-4, -2e-4, 1e-4, 2e-3, -1e-3, 1e-3])
ND = np.array([0.002, -0.001, 0.003, 0.01, 0.02, -0.005])   # [rotvec, trans] of the stand-in NDT pose


def _streams(seed=3):
    rng = np.random.Generator(np.random.PCG64(seed))
    m = synth.Motion(t_move=1.5)
    imu = synth.imu_stream(m, 0.0, 2.5, 200, rng, bg=(0.002, -0.001, 0.0015), ba=(0.03, -0.02, 0.01))
    return imu[imu[:, 0] <= 1.3], imu[(imu[:, 0] > 1.3) & (imu[:, 0] <= 1.9)]


def _run(tmp_path, init, motion, flag):
    path = tmp_path / "lio.bin"
    with open(path, "wb") as f:
        for a in (init, motion):
            f.write(struct.pack("<d", len(a))); f.write(np.ascontiguousarray(a, np.float64).tobytes())
        f.write(struct.pack("<3d", 1.0, 1e-4, 1e-5)); f.write(D.tobytes()); f.write(ND.tobytes()); f.write(struct.pack("<d", flag))
    out = subprocess.run([EXE, str(path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    res = {}
    for line in out.stdout.strip().split("\n"):
        k, *v = line.split()
        res[k] = np.array(v, float)
    return res


def _state(v):
    return lo.State(v[0], v[1:5], v[5:8], v[8:11], v[11:14], v[14:17])


def _close(a, b, rtol, atol=0.0):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape
    assert np.max(np.abs(a - b)) <= atol + rtol * max(np.max(np.abs(b)), 1e-300), (np.max(np.abs(a - b)), np.max(np.abs(b)))


def _state_close(v, s, tol, tol_vb=None):
    tol_vb = tol if tol_vb is None else tol_vb
    _close(v[1:5] * np.sign(v[1]), s.q * np.sign(s.q[0]), 0, tol); _close(v[5:8], s.p, 0, tol); _close(v[8:11], s.v, 0, tol_vb)
    _close(v[11:14], s.bg, 0, tol_vb); _close(v[14:17], s.ba, 0, tol_vb)


@pytest.mark.parametrize("flag", [0, 1])
def test_cpp_matches_numpy_oracle(tmp_path, flag):
    assert os.path.exists(EXE), "run __graft_entry__.build() first"
    init, motion = _streams()
    r = _run(tmp_path, init, motion, flag)

    # static initialisation (imu_init.cc)
    si = lo.StaticImuInit(init_need_time=1.0)
    for row in init:
        si.add(row[0], row[1:4], row[4:7])
    assert si.success and int(r["init_success"][0]) == 1
    for k, v in (("gravity", si.gravity), ("init_bg", si.bg), ("init_ba", si.ba), ("cov_gyro_n", si.cov_gyro_n), ("cov_acce_n", si.cov_acce_n)):
        _close(r[k], v, 1e-12)
    assert abs(np.linalg.norm(si.gravity) - 9.81) < 1e-12 and si.gravity[2] < -9.8

    # pre-integration (integration_base.h)
    aw, gw = 1e-4 ** 2 * np.eye(3), 1e-5 ** 2 * np.eye(3)
    pre = lo.ImuPreint(init[-1, 4:7], init[-1, 1:4], si.ba, si.bg, si.cov_acce_n, si.cov_gyro_n, aw, gw)
    tl = init[-1, 0]
    for row in motion:
        pre.propagate(row[0] - tl, row[4:7], row[1:4]); tl = row[0]
    _close(r["sum_dt"], [pre.sum_dt], 1e-13); _close(r["delta_p"], pre.dp, 1e-11); _close(r["delta_v"], pre.dv, 1e-11)
    _close(r["delta_q"], pre.dq, 1e-12)
    _close(r["jacobian"].reshape(15, 15), pre.J, 1e-10); _close(r["covariance"].reshape(15, 15), pre.P, 1e-10)
    last = lo.State(init[-1, 0], bg=si.bg, ba=si.ba)
    pred = pre.predict(last, si.gravity)
    _state_close(r["predict"], pred, 1e-11)

    # factors (factor.h)
    cur = lo.boxplus(pred, D)
    _close(r["imu_residual"], lo.imu_residual(pre, si.gravity, last, cur, reference_rotation_residual=bool(flag)), 1e-6)
    npose = lo.boxplus(pred, np.concatenate([ND[3:], ND[:3], np.zeros(9)]))
    ndt_pose = np.eye(4); ndt_pose[:3, :3] = lo.q_to_R(npose.q); ndt_pose[:3, 3] = npose.p
    ga2, gp2 = (2.0 * 3.14 / 180) ** 2, 0.1 ** 2
    info = np.diag([1 / ga2] * 3 + [1 / gp2] * 3)
    _close(r["pc_residual"], lo.pc_residual(ndt_pose, info, cur), 1e-10)

    # optimisation without a prior, then with the prior it produced (process.cc:155-258)
    l1, c1, Jr1, Er1, it1 = lo.optimize(pre, si.gravity, last, cur, ndt_pose, info, None, bool(flag))
    _state_close(r["opt1_cur"], c1, 1e-6); _state_close(r["opt1_last"], l1, 1e-12)      # no prior: previous state untouched
    # the prior is a Schur complement of an information matrix whose IMU block is ~1e10 (covariance ~1e-10): the
    # subtraction loses ~7 digits, so the two implementations agree to ~1e-4 of the largest entry, not to 1e-9.
    # Jr itself is free in the sign / order of its rows (eigenvectors): compare Jr^T Jr, Jr^T Er and |Jr dx + Er|.
    _close(r["opt1_JtJ"].reshape(15, 15), Jr1.T @ Jr1, 1e-3); _close(r["opt1_JtE"], Jr1.T @ Er1, 1e-3, 1e-3)
    nm = np.linalg.norm(lo.marg_residual(c1, Jr1, Er1, lo.boxplus(c1, D)))
    assert abs(np.linalg.norm(r["marg_residual"]) - nm) < 1e-3 * nm + 1e-6
    pred2 = pre.predict(c1, si.gravity)
    cur2 = lo.boxplus(pred2, D)
    np2 = lo.boxplus(pred2, np.concatenate([ND[3:], ND[:3], np.zeros(9)]))
    ndt_pose2 = np.eye(4); ndt_pose2[:3, :3] = lo.q_to_R(np2.q); ndt_pose2[:3, 3] = np2.p
    l2, c2, Jr2, Er2, it2 = lo.optimize(pre, si.gravity, c1, cur2, ndt_pose2, info, (c1, Jr1, Er1), bool(flag))
    # with the prior the problem has a flat valley along velocity / biases: LM runs into the 10-iteration cap
    # (process.cc:218) on both sides, poses agree to 1e-5, velocity and biases to 1e-3
    assert it2 == 10 and int(r["opt2_iterations"][0]) == 10
    _state_close(r["opt2_cur"], c2, 1e-5, 1e-3); _state_close(r["opt2_last"], l2, 1e-5, 1e-3)
    _close(r["opt2_JtJ"].reshape(15, 15), Jr2.T @ Jr2, 1e-3)
    if not flag:
        # with the corrected rotation residual the optimum sits between the IMU prediction and the NDT pose
        assert np.linalg.norm(c1.p - ndt_pose[:3, 3]) < np.linalg.norm(cur.p - ndt_pose[:3, 3])
        assert float(r["opt1_cost"][0]) < 0.5 * 9 * np.log1p(sum(np.sum(b ** 2) for b in (lo.imu_residual(pre, si.gravity, last, cur), lo.pc_residual(ndt_pose, info, cur))) / 9) + 50


def test_preintegration_tracks_the_motion():
    """A noise-free stream integrates to the analytic motion (sanity of generator + pre-integration together)."""
    rng = np.random.Generator(np.random.PCG64(1))
    m = synth.Motion(t_move=0.5)
    imu = synth.imu_stream(m, 0.0, 3.0, 400, rng, gyro_sigma=0.0, acce_sigma=0.0)
    pre = lo.ImuPreint(imu[0, 4:7], imu[0, 1:4], np.zeros(3), np.zeros(3), np.ones(3) * 1e-4, np.ones(3) * 1e-6, np.eye(3) * 1e-8, np.eye(3) * 1e-10)
    for a, b in zip(imu[:-1], imu[1:]):
        pre.propagate(b[0] - a[0], b[4:7], b[1:4])
    s = pre.predict(lo.State(0.0), synth.GRAVITY)
    T = m.pose(3.0)
    assert np.linalg.norm(s.p - T[:3, 3]) < 5e-3
    assert np.linalg.norm(lo.log_R(lo.q_to_R(s.q).T @ T[:3, :3])) < 1e-3


def test_marginalize_against_schur_complement():
    rng = np.random.default_rng(0)
    A = rng.normal(size=(40, 30)); H = A.T @ A; b = rng.normal(size=30)
    Jr, Er = lo.marginalize(H, b, 15)
    H2 = H[15:, 15:] - H[15:, :15] @ np.linalg.solve(H[:15, :15], H[:15, 15:])
    b2 = b[15:] - H[15:, :15] @ np.linalg.solve(H[:15, :15], b[:15])
    assert np.allclose(Jr.T @ Jr, H2, rtol=1e-9, atol=1e-9) and np.allclose(Jr.T @ Er, -b2, rtol=1e-9, atol=1e-9)
