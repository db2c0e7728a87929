"""NumPy restatement of lio-slam's IMU / optimiser half — TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module; the product is include/ndt_mi355_lio.hpp (C++).  The reference's own code for
this part is in the tree but needs Eigen + Ceres + ROS, which are absent: it cannot be compiled here, so this file
follows it line by line instead, citing what it follows.  PARITY UNPINNED: the reference holds no tests or vectors.

  ImuPreint          src/slam/include/integration_base.h:11-30 (ctor), :39-114 (midPointIntegration),
                     :116-143 (propagate), :145-154 (predict), :173-199 (bias-corrected deltas)
  StaticImuInit      src/slam/src/imu_init.cc:5-62, src/slam/include/imu_init.h:12-21 (options)
  imu_residual       src/slam/include/factor.h:14 (sqrt information), :18-62
  pc_residual        src/slam/include/factor.h:222-257
  marg_residual      src/slam/include/factor.h:162-186
  marginalize        src/slam/src/utility.cc:95-124
  optimize           src/slam/src/process.cc:155-258

Where the restatement departs from the reference, and why (the same choices as the C++ header):
  * Ceres (DOGLEG trust region, autodiff) is absent: Levenberg-Marquardt on central-difference Jacobians, Cauchy loss
    (scale 3, process.cc:158) as per-block IRLS weights, <= 10 iterations (process.cc:218).
  * Rotation increments are right-multiplied rotation vectors, q <- q * Exp(dtheta): the convention of MargFactor's
    dx = 2 vec(q0^-1 q) (factor.h:175-180).  Ceres' EigenQuaternionParameterization left-multiplies a half-angle
    vector, which is inconsistent with that factor.
  * factor.h:43 multiplies by R(q_current)^T, which is not zero at the IMU-predicted state; `reference_rotation_residual`
    reproduces it, the default uses R(q_current).
  * With no prior (first optimised frame) the reference fixes the previous state, so Ceres' Jacobian has 15 columns and
    marginalize(H, b, ..., 15) at process.cc:255 reads out of range; here the Jacobian for the marginalisation always
    has all 30 columns.
"""
import numpy as np

EPS_MARG = 1e-8          # utility.h:138


# ------------------------------------------------------------------ quaternions (w, x, y, z) the way Eigen computes
def q_mul(a, b):
    aw, ax, ay, az = a; bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx])


def q_conj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def q_normalized(q):
    return q / np.sqrt(q @ q)


def q_to_R(q):
    """Eigen::QuaternionBase::toRotationMatrix (no normalisation: a non-unit q gives what Eigen gives)."""
    w, x, y, z = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1 - (txx + tzz), tyz - twx], [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def R_to_q(R):
    """Eigen::Quaternion(Matrix3) (Shepperd's branches as in Eigen)."""
    t = np.trace(R)
    if t > 0:
        t = np.sqrt(t + 1.0); w = 0.5 * t; t = 0.5 / t
        return np.array([w, (R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t])
    i = 0
    if R[1, 1] > R[0, 0]: i = 1
    if R[2, 2] > R[i, i]: i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    q = np.zeros(4); q[1 + i] = 0.5 * t; t = 0.5 / t
    q[0] = (R[k, j] - R[j, k]) * t; q[1 + j] = (R[j, i] + R[i, j]) * t; q[1 + k] = (R[k, i] + R[i, k]) * t
    return q


def log_R(R):
    """Eigen::AngleAxis(Matrix3): angle * axis, through the quaternion."""
    q = R_to_q(R)
    n = np.sqrt(q[1:] @ q[1:])
    if n == 0:
        return np.zeros(3)
    ang = 2 * np.arctan2(n, abs(q[0]))
    if q[0] < 0: n = -n
    return ang * q[1:] / n


def exp_q(th):
    """unit quaternion of the rotation vector th"""
    a = np.sqrt(th @ th)
    if a < 1e-12:
        return q_normalized(np.array([1.0, 0.5 * th[0], 0.5 * th[1], 0.5 * th[2]]))
    return np.concatenate([[np.cos(a / 2)], np.sin(a / 2) * th / a])


def hat(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


class State:   # utility.h:99-115
    def __init__(self, t=0.0, q=None, p=None, v=None, bg=None, ba=None):
        self.t = t
        self.q = np.array([1.0, 0, 0, 0]) if q is None else np.asarray(q, float)
        self.p = np.zeros(3) if p is None else np.asarray(p, float)
        self.v = np.zeros(3) if v is None else np.asarray(v, float)
        self.bg = np.zeros(3) if bg is None else np.asarray(bg, float)
        self.ba = np.zeros(3) if ba is None else np.asarray(ba, float)

    def copy(self):
        return State(self.t, self.q.copy(), self.p.copy(), self.v.copy(), self.bg.copy(), self.ba.copy())

    def transform(self):   # utility.cc:22-35 (Matrix4f: rounded to float32)
        T = np.eye(4); T[:3, :3] = q_to_R(self.q); T[:3, 3] = self.p
        return T.astype(np.float32).astype(np.float64)


# ------------------------------------------------------------------ static IMU initialisation
class StaticImuInit:
    def __init__(self, init_need_time=10.0, max_size=2000, min_size=100, max_gyro_var=0.5, max_acce_var=0.05, gravity_norm=9.81):
        self.o = dict(need=init_need_time, max=max_size, min=min_size, gv=max_gyro_var, av=max_acce_var, g=gravity_norm)
        self.buf = []; self.success = False; self.t0 = 0.0
        self.gravity = np.zeros(3); self.ba = np.zeros(3); self.bg = np.zeros(3)
        self.cov_gyro_n = np.zeros(3); self.cov_acce_n = np.zeros(3)

    def add(self, t, gyro, acce):   # imu_init.cc:5-31
        if self.success:
            return True
        if not self.buf:
            self.t0 = t
        self.buf.append((t, np.asarray(gyro, float), np.asarray(acce, float)))
        if t - self.t0 > self.o["need"]:
            self.try_init()
        while len(self.buf) > self.o["max"]:
            self.buf.pop(0)
        return True

    @staticmethod
    def _mean_cov(vals):   # utility.h:80-95: mean and per-axis variance over (n-1)
        vals = np.asarray(vals)
        m = vals.sum(0) / len(vals)
        return m, ((vals - m) ** 2).sum(0) / (len(vals) - 1)

    def try_init(self):   # imu_init.cc:34-61
        if len(self.buf) < self.o["min"]:
            return False
        mg, self.cov_gyro_n = self._mean_cov([b[1] for b in self.buf])
        ma, self.cov_acce_n = self._mean_cov([b[2] for b in self.buf])
        self.gravity = -ma / np.linalg.norm(ma) * self.o["g"]
        ma, self.cov_acce_n = self._mean_cov([b[2] + self.gravity for b in self.buf])
        if np.linalg.norm(self.cov_gyro_n) > self.o["gv"] or np.linalg.norm(self.cov_acce_n) > self.o["av"]:
            return False
        self.bg, self.ba = mg, ma
        self.success = True
        return True


# ------------------------------------------------------------------ pre-integration
class ImuPreint:
    def __init__(self, acc0, gyr0, ba, bg, cov_acce_n, cov_gyro_n, cov_acce_w, cov_gyro_w):   # integration_base.h:11-30
        self.acc0, self.gyr0 = np.asarray(acc0, float), np.asarray(gyr0, float)
        self.ba, self.bg = np.asarray(ba, float).copy(), np.asarray(bg, float).copy()
        self.J = np.eye(15); self.P = np.zeros((15, 15)); self.sum_dt = 0.0
        self.dp = np.zeros(3); self.dq = np.array([1.0, 0, 0, 0]); self.dv = np.zeros(3)
        n = np.zeros((18, 18))
        n[0:3, 0:3] = cov_acce_n[0] * np.eye(3); n[3:6, 3:6] = cov_gyro_n[0] * np.eye(3)
        n[6:9, 6:9] = cov_acce_n[0] * np.eye(3); n[9:12, 9:12] = cov_gyro_n[0] * np.eye(3)
        n[12:15, 12:15] = cov_acce_w; n[15:18, 15:18] = cov_gyro_w
        self.noise = n

    def propagate(self, dt, acc1, gyr1):   # :116-143 with :39-114
        acc1, gyr1 = np.asarray(acc1, float), np.asarray(gyr1, float)
        dq, dp, dv, ba, bg = self.dq, self.dp, self.dv, self.ba, self.bg
        un_acc0 = q_to_R(dq) @ (self.acc0 - ba)
        un_gyr = 0.5 * (self.gyr0 + gyr1) - bg
        rq = q_normalized(q_mul(dq, np.array([1.0, un_gyr[0] * dt / 2, un_gyr[1] * dt / 2, un_gyr[2] * dt / 2])))
        un_acc1 = q_to_R(rq) @ (acc1 - ba)
        un_acc = 0.5 * (un_acc0 + un_acc1)
        rp = dp + dv * dt + 0.5 * un_acc * dt * dt
        rv = dv + un_acc * dt
        w_x, a0_x, a1_x = hat(un_gyr), hat(self.acc0 - ba), hat(acc1 - ba)
        R0, R1, I3 = q_to_R(dq), q_to_R(rq), np.eye(3)
        F = np.zeros((15, 15))
        F[0:3, 0:3] = I3
        F[0:3, 3:6] = -0.25 * R0 @ a0_x * dt * dt + -0.25 * R1 @ a1_x @ (I3 - w_x * dt) * dt * dt
        F[0:3, 6:9] = I3 * dt
        F[0:3, 9:12] = -0.25 * (R0 + R1) * dt * dt
        F[0:3, 12:15] = -0.25 * R1 @ a1_x * dt * dt * -dt
        F[3:6, 3:6] = I3 - w_x * dt
        F[3:6, 12:15] = -1.0 * I3 * dt
        F[6:9, 3:6] = -0.5 * R0 @ a0_x * dt + -0.5 * R1 @ a1_x @ (I3 - w_x * dt) * dt
        F[6:9, 6:9] = I3
        F[6:9, 9:12] = -0.5 * (R0 + R1) * dt
        F[6:9, 12:15] = -0.5 * R1 @ a1_x * dt * -dt
        F[9:12, 9:12] = I3; F[12:15, 12:15] = I3
        V = np.zeros((15, 18))
        V[0:3, 0:3] = 0.25 * R0 * dt * dt
        V[0:3, 3:6] = 0.25 * -R1 @ a1_x * dt * dt * 0.5 * dt
        V[0:3, 6:9] = 0.25 * R1 * dt * dt
        V[0:3, 9:12] = V[0:3, 3:6]
        V[3:6, 3:6] = 0.5 * I3 * dt; V[3:6, 9:12] = 0.5 * I3 * dt
        V[6:9, 0:3] = 0.5 * R0 * dt
        V[6:9, 3:6] = 0.5 * -R1 @ a1_x * dt * 0.5 * dt
        V[6:9, 6:9] = 0.5 * R1 * dt
        V[6:9, 9:12] = V[6:9, 3:6]
        V[9:12, 12:15] = I3 * dt; V[12:15, 15:18] = I3 * dt
        self.J = F @ self.J
        self.P = F @ self.P @ F.T + V @ self.noise @ V.T
        self.dp, self.dq, self.dv = rp, q_normalized(rq), rv
        self.sum_dt += dt
        self.acc0, self.gyr0 = acc1, gyr1

    def predict(self, s, g):   # :145-154
        R = q_to_R(s.q); T = self.sum_dt
        return State(s.t + T, q_mul(s.q, self.dq), R @ self.dp + s.p + s.v * T + 0.5 * g * T * T, R @ self.dv + s.v + g * T,
                     self.bg.copy(), self.ba.copy())

    def delta_rotation(self, bg):   # :173-179 (theta2Q: (1, theta/2), not normalised)
        tem = self.J[3:6, 12:15] @ (bg - self.bg)
        return q_mul(self.dq, np.array([1.0, tem[0] / 2, tem[1] / 2, tem[2] / 2]))

    def delta_velocity(self, bg, ba):   # :183-189
        return self.dv + self.J[6:9, 12:15] @ (bg - self.bg) + self.J[6:9, 9:12] @ (ba - self.ba)

    def delta_position(self, bg, ba):   # :193-199
        return self.dp + self.J[0:3, 12:15] @ (bg - self.bg) + self.J[0:3, 9:12] @ (ba - self.ba)


# ------------------------------------------------------------------ factors
def imu_sqrt_info(pre):   # factor.h:14: LLT(covariance^-1).matrixL().transpose()
    return np.linalg.cholesky(np.linalg.inv(pre.P)).T


def imu_residual(pre, g, last, cur, sqrt_info=None, reference_rotation_residual=False):   # factor.h:18-62
    dt = pre.sum_dt
    dpos = pre.delta_position(cur.bg, cur.ba); dvel = pre.delta_velocity(cur.bg, cur.ba)
    dR = q_to_R(pre.delta_rotation(cur.bg))
    Rl = q_to_R(last.q).T
    Rc = q_to_R(cur.q)
    r = np.zeros(15)
    r[0:3] = Rl @ (cur.p - last.p - last.v * dt - g * dt * dt / 2) - dpos
    r[3:6] = log_R(dR.T @ Rl @ (Rc.T if reference_rotation_residual else Rc))
    r[6:9] = Rl @ (cur.v - last.v - g * dt) - dvel
    r[9:12] = cur.ba - last.ba
    r[12:15] = cur.bg - last.bg
    return (imu_sqrt_info(pre) if sqrt_info is None else sqrt_info) @ r


def pc_residual(ndt_pose, info, cur):   # factor.h:222-257
    sq = np.linalg.cholesky(info).T
    qn = R_to_q(ndt_pose[:3, :3])
    eR = q_to_R(q_conj(qn) / (qn @ qn)) @ q_to_R(cur.q)
    return sq @ np.concatenate([log_R(eR), cur.p - ndt_pose[:3, 3]])


def marg_dx(s0, s):   # factor.h:166-183
    tq = q_mul(q_conj(s0.q) / (s0.q @ s0.q), s.q)
    if tq[0] < 0: tq = -tq
    return np.concatenate([s.p - s0.p, 2.0 * tq[1:], s.v - s0.v, s.bg - s0.bg, s.ba - s0.ba])


def marg_residual(s0, Jr, Er, s):   # factor.h:185-186
    return Jr @ marg_dx(s0, s) + Er


def marginalize(H, b, m=15):   # utility.cc:95-124
    Amm = 0.5 * (H[:m, :m] + H[:m, :m].T)
    lam, V = np.linalg.eigh(Amm)
    inv = np.where(lam > EPS_MARG, 1.0 / np.where(lam > EPS_MARG, lam, 1.0), 0.0)
    Ainv = (V * inv) @ V.T
    H2 = H[m:, m:] - H[m:, :m] @ Ainv @ H[:m, m:]
    b2 = b[m:] - H[m:, :m] @ Ainv @ b[:m]
    lam2, V2 = np.linalg.eigh(H2)
    S = np.where(lam2 > EPS_MARG, lam2, 0.0)
    Sinv = np.where(lam2 > EPS_MARG, 1.0 / np.where(lam2 > EPS_MARG, lam2, 1.0), 0.0)
    return np.sqrt(S)[:, None] * V2.T, np.sqrt(Sinv) * (V2.T @ (-b2))


# ------------------------------------------------------------------ optimisation (process.cc:155-258)
def boxplus(s, d):   # local order of the parameter blocks: p, q, v, bg, ba (process.cc:172-182)
    o = s.copy()
    o.p = s.p + d[0:3]; o.q = q_normalized(q_mul(s.q, exp_q(d[3:6]))); o.v = s.v + d[6:9]; o.bg = s.bg + d[9:12]; o.ba = s.ba + d[12:15]
    return o


CAUCHY_A2 = 9.0   # ceres::CauchyLoss(3.0), process.cc:158


def _blocks(pre, g, ndt_pose, ndt_info, prior, last, cur, sqrt_info, ref_rot):
    out = []
    if prior is not None:
        out.append(marg_residual(prior[0], prior[1], prior[2], last))
    out.append(imu_residual(pre, g, last, cur, sqrt_info, ref_rot))
    out.append(pc_residual(ndt_pose, ndt_info, cur))
    return out


def _cost(blocks):
    return 0.5 * sum(CAUCHY_A2 * np.log1p((b @ b) / CAUCHY_A2) for b in blocks)


def _weighted(blocks):
    return np.concatenate([b * np.sqrt(1.0 / (1.0 + (b @ b) / CAUCHY_A2)) for b in blocks])


def _jacobian(fun, last, cur, cols, h=1e-6):
    r0 = fun(last, cur)
    J = np.zeros((len(r0), 30))
    for k in cols:
        d = np.zeros(30); d[k] = h
        rp = fun(boxplus(last, d[:15]), boxplus(cur, d[15:]))
        rm = fun(boxplus(last, -d[:15]), boxplus(cur, -d[15:]))
        J[:, k] = (rp - rm) / (2 * h)
    return r0, J


def optimize(pre, g, last, cur, ndt_pose, ndt_info, prior=None, reference_rotation_residual=False, max_iterations=10):
    """Returns (last', cur', Jr, Er, iterations).  prior = (state0, Jr, Er) or None (previous state held fixed)."""
    sq = imu_sqrt_info(pre)
    blocks = lambda l, c: _blocks(pre, g, ndt_pose, ndt_info, prior, l, c, sq, reference_rotation_residual)

    def wres(l, c):      # IRLS: weights frozen at the linearisation point
        return np.concatenate([b * w for b, w in zip(blocks(l, c), wres.w)])
    cols = list(range(30)) if prior is not None else list(range(15, 30))
    lam = 1e-4
    cost = _cost(blocks(last, cur))
    its = 0
    for its in range(1, max_iterations + 1):
        wres.w = [np.sqrt(1.0 / (1.0 + (b @ b) / CAUCHY_A2)) for b in blocks(last, cur)]
        r, J = _jacobian(wres, last, cur, cols)
        Jc = J[:, cols]
        H = Jc.T @ Jc; gr = Jc.T @ r
        stepped = False
        while lam <= 1e8:
            dx = np.linalg.solve(H + lam * np.diag(np.diag(H)) + 1e-12 * np.eye(len(cols)), -gr)
            d = np.zeros(30); d[cols] = dx
            l2, c2 = boxplus(last, d[:15]), boxplus(cur, d[15:])
            cn = _cost(blocks(l2, c2))
            if cn < cost:
                last, cur, cost = l2, c2, cn
                lam = max(lam / 10.0, 1e-12); stepped = True
                break
            lam *= 10.0
        if not stepped or np.linalg.norm(dx) < 1e-8:
            break
    # marginalisation of the previous state (process.cc:237-256); all 30 columns (see the header)
    wres.w = [np.sqrt(1.0 / (1.0 + (b @ b) / CAUCHY_A2)) for b in blocks(last, cur)]
    r, J = _jacobian(wres, last, cur, list(range(30)))
    Jr, Er = marginalize(J.T @ J, -J.T @ r, 15)
    return last, cur, Jr, Er, its
