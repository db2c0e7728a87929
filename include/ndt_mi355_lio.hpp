// ndt_mi355_lio.hpp — the IMU / optimiser half of lio-slam's per-frame routine, host-only
// C++17 without Eigen / Ceres / ROS, so that the whole loop  predictByIMU -> alignByPC -> optimize  can be replayed
// over recorded or synthetic IMU + LiDAR streams around the MI355X NDT engine (LioPipeline at the end of the file).
//
//   NavState            slam_czc::State                      src/slam/include/utility.h:99-115, utility.cc:22-35
//   StaticImuInit       slam_czc::StaticIMUInit              src/slam/src/imu_init.cc:5-62, include/imu_init.h:12-21
//   ImuPreintegration   IntegrationBase                      src/slam/include/integration_base.h:11-30, 39-154, 173-199
//   imuResidual         IMUFactor                            src/slam/include/factor.h:14, 18-62
//   pointCloudResidual  PointCloudFactor                     src/slam/include/factor.h:222-257
//   margResidual        MargFactor::Evaluate                 src/slam/include/factor.h:162-186
//   marginalize         slam_czc::marginalize                src/slam/src/utility.cc:95-124
//   optimize            Process::optimize                    src/slam/src/process.cc:155-258
//   LioPipeline         Process::dataAlign/dataProcess/initIMU/predictByIMU/alignByPC   process.cc:6-153, 318-349
//
// Departures from the reference (same list as oracle/lio_oracle.py, which is the checker for this file):
//   * Ceres is absent: Levenberg-Marquardt on central-difference Jacobians in place of DOGLEG + autodiff; the Cauchy
//     loss (scale 3, process.cc:158) as per-block IRLS weights; <= 10 iterations (process.cc:218).
//   * rotation increments are right-multiplied rotation vectors (the convention of MargFactor's dx, factor.h:175-180);
//   * factor.h:43 multiplies by R(q_current)^T, which is not zero at the IMU-predicted state:
//     OptimizeOptions::reference_rotation_residual reproduces it, the default uses R(q_current);
//   * with no prior the reference fixes the previous state, so its 15-column Jacobian is indexed out of range by
//     marginalize(H, b, ..., 15) at process.cc:255; here the marginalisation Jacobian always has 30 columns.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <deque>
#include <memory>
#include <stdexcept>
#include <vector>

namespace ndt_mi355 {
namespace lio {

using V3 = std::array<double, 3>;
using M3 = std::array<double, 9>;   // row-major
struct Quat { double w = 1, x = 0, y = 0, z = 0; };

inline V3 operator+(const V3& a, const V3& b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
inline V3 operator-(const V3& a, const V3& b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
inline V3 operator*(const V3& a, double s) { return {a[0] * s, a[1] * s, a[2] * s}; }
inline V3 operator*(double s, const V3& a) { return a * s; }
inline double dot(const V3& a, const V3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline double norm(const V3& a) { return std::sqrt(dot(a, a)); }
inline V3 mul(const M3& m, const V3& v) { return {m[0] * v[0] + m[1] * v[1] + m[2] * v[2], m[3] * v[0] + m[4] * v[1] + m[5] * v[2], m[6] * v[0] + m[7] * v[1] + m[8] * v[2]}; }
inline M3 mul(const M3& a, const M3& b) {
    M3 c{};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
    return c;
}
inline M3 transpose(const M3& a) { return {a[0], a[3], a[6], a[1], a[4], a[7], a[2], a[5], a[8]}; }
inline M3 hat(const V3& v) { return {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0}; }
inline M3 identity3() { return {1, 0, 0, 0, 1, 0, 0, 0, 1}; }

inline Quat qmul(const Quat& a, const Quat& b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
inline double qnorm2(const Quat& q) { return q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z; }
inline Quat qnormalized(const Quat& q) { const double n = std::sqrt(qnorm2(q)); return {q.w / n, q.x / n, q.y / n, q.z / n}; }
inline Quat qinverse(const Quat& q) { const double n = qnorm2(q); return {q.w / n, -q.x / n, -q.y / n, -q.z / n}; }
// Eigen::QuaternionBase::toRotationMatrix (no normalisation)
inline M3 toRotationMatrix(const Quat& q) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    return {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
}
// Eigen::Quaternion(Matrix3)
inline Quat fromRotationMatrix(const M3& R) {
    double t = R[0] + R[4] + R[8];
    Quat q;
    if (t > 0) {
        t = std::sqrt(t + 1.0); q.w = 0.5 * t; t = 0.5 / t;
        q.x = (R[7] - R[5]) * t; q.y = (R[2] - R[6]) * t; q.z = (R[3] - R[1]) * t;
        return q;
    }
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    double v[3];
    v[i] = 0.5 * t; t = 0.5 / t;
    q.w = (R[3 * k + j] - R[3 * j + k]) * t;
    v[j] = (R[3 * j + i] + R[3 * i + j]) * t;
    v[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
    return q;
}
// Eigen::AngleAxis(Matrix3): angle * axis
inline V3 logRotation(const M3& R) {
    const Quat q = fromRotationMatrix(R);
    double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
    if (n == 0) return {0, 0, 0};
    const double ang = 2 * std::atan2(n, std::fabs(q.w));
    if (q.w < 0) n = -n;
    return {ang * q.x / n, ang * q.y / n, ang * q.z / n};
}
inline Quat expQuat(const V3& th) {
    const double a = norm(th);
    if (a < 1e-12) return qnormalized({1.0, 0.5 * th[0], 0.5 * th[1], 0.5 * th[2]});
    const double s = std::sin(a / 2) / a;
    return {std::cos(a / 2), s * th[0], s * th[1], s * th[2]};
}

// ---- small dense matrices (row-major)
struct Mat {
    int r = 0, c = 0;
    std::vector<double> a;
    Mat() = default;
    Mat(int rows, int cols) : r(rows), c(cols), a((size_t)rows * cols, 0.0) {}
    double& operator()(int i, int j) { return a[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
    static Mat identity(int n) { Mat m(n, n); for (int i = 0; i < n; ++i) m(i, i) = 1.0; return m; }
    void setBlock(int i0, int j0, const M3& b, double s = 1.0) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) (*this)(i0 + i, j0 + j) = s * b[3 * i + j]; }
    M3 block3(int i0, int j0) const { M3 b{}; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) b[3 * i + j] = (*this)(i0 + i, j0 + j); return b; }
};
inline Mat matmul(const Mat& A, const Mat& B) {
    Mat C(A.r, B.c);
    for (int i = 0; i < A.r; ++i)
        for (int k = 0; k < A.c; ++k) { const double v = A(i, k); if (v == 0.0) continue; for (int j = 0; j < B.c; ++j) C(i, j) += v * B(k, j); }
    return C;
}
inline Mat transposed(const Mat& A) { Mat T(A.c, A.r); for (int i = 0; i < A.r; ++i) for (int j = 0; j < A.c; ++j) T(j, i) = A(i, j); return T; }
inline std::vector<double> matvec(const Mat& A, const std::vector<double>& x) {
    std::vector<double> y(A.r, 0.0);
    for (int i = 0; i < A.r; ++i) { double s = 0; for (int j = 0; j < A.c; ++j) s += A(i, j) * x[j]; y[i] = s; }
    return y;
}
// lower Cholesky factor; throws if not positive definite (Eigen::LLT would return garbage)
inline Mat cholesky(const Mat& A) {
    const int n = A.r;
    Mat L(n, n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = A(i, j);
            for (int k = 0; k < j; ++k) s -= L(i, k) * L(j, k);
            if (i == j) { if (!(s > 0.0)) throw std::runtime_error("cholesky: matrix is not positive definite"); L(i, i) = std::sqrt(s); }
            else L(i, j) = s / L(j, j);
        }
    return L;
}
// solves A x = b by Gaussian elimination with partial pivoting (A is copied)
inline std::vector<double> solve(Mat A, std::vector<double> b) {
    const int n = A.r;
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int i = k + 1; i < n; ++i) if (std::fabs(A(i, k)) > std::fabs(A(p, k))) p = i;
        if (A(p, k) == 0.0) throw std::runtime_error("solve: singular matrix");
        if (p != k) { for (int j = 0; j < n; ++j) std::swap(A(k, j), A(p, j)); std::swap(b[k], b[p]); }
        for (int i = k + 1; i < n; ++i) {
            const double f = A(i, k) / A(k, k);
            if (f == 0.0) continue;
            for (int j = k; j < n; ++j) A(i, j) -= f * A(k, j);
            b[i] -= f * b[k];
        }
    }
    std::vector<double> x(n);
    for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int j = i + 1; j < n; ++j) s -= A(i, j) * x[j]; x[i] = s / A(i, i); }
    return x;
}
inline Mat inverse(const Mat& A) {
    const int n = A.r;
    Mat X(n, n);
    for (int j = 0; j < n; ++j) {
        std::vector<double> e(n, 0.0); e[j] = 1.0;
        const std::vector<double> col = solve(A, e);
        for (int i = 0; i < n; ++i) X(i, j) = col[i];
    }
    return X;
}
// cyclic Jacobi eigen-decomposition of a symmetric matrix: A = V diag(lam) V^T (columns of V)
inline void symmetricEigen(const Mat& Ain, std::vector<double>& lam, Mat& V) {
    const int n = Ain.r;
    Mat A(n, n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A(i, j) = 0.5 * (Ain(i, j) + Ain(j, i));
    V = Mat::identity(n);
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0, dia = 0;
        for (int i = 0; i < n; ++i) { dia += A(i, i) * A(i, i); for (int j = i + 1; j < n; ++j) off += A(i, j) * A(i, j); }
        if (off == 0.0 || off <= 1e-34 * dia) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                if (A(p, q) == 0.0) continue;
                const double th = (A(q, q) - A(p, p)) / (2.0 * A(p, q));
                const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) { const double a = A(k, p), b = A(k, q); A(k, p) = c * a - s * b; A(k, q) = s * a + c * b; }
                for (int k = 0; k < n; ++k) { const double a = A(p, k), b = A(q, k); A(p, k) = c * a - s * b; A(q, k) = s * a + c * b; }
                for (int k = 0; k < n; ++k) { const double a = V(k, p), b = V(k, q); V(k, p) = c * a - s * b; V(k, q) = s * a + c * b; }
            }
    }
    lam.resize(n);
    for (int i = 0; i < n; ++i) lam[i] = A(i, i);
}

// ---- state (utility.h:99-115)
struct ImuSample { double t = 0; V3 gyro{0, 0, 0}; V3 acce{0, 0, 0}; };
struct NavState {
    double t = 0;
    Quat q;
    V3 p{0, 0, 0}, v{0, 0, 0}, bg{0, 0, 0}, ba{0, 0, 0};
    // State::getTransform (utility.cc:22-35): a Matrix4f — values rounded to float32; column-major double[16] here
    std::array<double, 16> transform() const {
        const M3 R = toRotationMatrix(q);
        std::array<double, 16> T{};
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T[c * 4 + r] = (double)(float)R[3 * r + c]; T[12 + r] = (double)(float)p[r]; }
        T[15] = 1.0;
        return T;
    }
};

// ---- static initialisation (imu_init.cc)
class StaticImuInit {
  public:
    struct Options {   // imu_init.h:12-21
        double init_need_time = 10; int init_imu_max_size = 2000; int init_imu_min_size = 100;
        double max_gyro_var = 0.5; double max_acce_var = 0.05; double gravity_norm = 9.81;
    };
    Options options;
    bool init_success = false;
    V3 gravity{0, 0, 0}, init_ba{0, 0, 0}, init_bg{0, 0, 0}, cov_gyro_n{0, 0, 0}, cov_acce_n{0, 0, 0};
    M3 cov_acce_w = identity3(), cov_gyro_w = identity3();

    bool addIMU(const ImuSample& imu) {   // imu_init.cc:5-31
        if (init_success) return true;
        if (buf_.empty()) begin_ = imu.t;
        buf_.push_back(imu);
        if (imu.t - begin_ > options.init_need_time) tryInit();
        while ((int)buf_.size() > options.init_imu_max_size) buf_.pop_front();
        return true;
    }
    bool tryInit() {   // imu_init.cc:34-61
        if ((int)buf_.size() < options.init_imu_min_size) return false;
        V3 mean_gyro, mean_acce;
        meanCov([](const ImuSample& s) { return s.gyro; }, mean_gyro, cov_gyro_n);
        meanCov([](const ImuSample& s) { return s.acce; }, mean_acce, cov_acce_n);
        gravity = mean_acce * (-1.0 / norm(mean_acce) * options.gravity_norm);
        const V3 g = gravity;
        meanCov([g](const ImuSample& s) { return s.acce + g; }, mean_acce, cov_acce_n);
        if (norm(cov_gyro_n) > options.max_gyro_var || norm(cov_acce_n) > options.max_acce_var) return false;
        init_bg = mean_gyro; init_ba = mean_acce;
        init_success = true;
        return true;
    }

  private:
    template <class G> void meanCov(G get, V3& mean, V3& cov) const {   // computeMeanAndCov, utility.h:80-95
        const double n = (double)buf_.size();
        V3 s{0, 0, 0};
        for (const auto& b : buf_) s = s + get(b);
        mean = s * (1.0 / n);
        V3 c{0, 0, 0};
        for (const auto& b : buf_) { const V3 d = get(b) - mean; c = c + V3{d[0] * d[0], d[1] * d[1], d[2] * d[2]}; }
        cov = c * (1.0 / (n - 1));
    }
    std::deque<ImuSample> buf_;
    double begin_ = 0;
};

// ---- pre-integration (integration_base.h).  State order of jacobian / covariance: p 0, theta 3, v 6, ba 9, bg 12.
class ImuPreintegration {
  public:
    ImuPreintegration(const V3& acc_0, const V3& gyr_0, const V3& linearized_ba, const V3& linearized_bg, const V3& cov_acce_n,
                      const V3& cov_gyro_n, const M3& cov_acce_w, const M3& cov_gyro_w)
        : acc_0(acc_0), gyr_0(gyr_0), linearized_ba(linearized_ba), linearized_bg(linearized_bg), jacobian(Mat::identity(15)), covariance(15, 15),
          noise(18, 18) {
        noise.setBlock(0, 0, identity3(), cov_acce_n[0]); noise.setBlock(3, 3, identity3(), cov_gyro_n[0]);
        noise.setBlock(6, 6, identity3(), cov_acce_n[0]); noise.setBlock(9, 9, identity3(), cov_gyro_n[0]);
        noise.setBlock(12, 12, cov_acce_w); noise.setBlock(15, 15, cov_gyro_w);
    }
    void propagate(double dt, const V3& acc_1, const V3& gyr_1) {   // :116-143 with midPointIntegration :39-114
        const V3 un_acc_0 = mul(toRotationMatrix(delta_q), acc_0 - linearized_ba);
        const V3 un_gyr = 0.5 * (gyr_0 + gyr_1) - linearized_bg;
        const Quat rq = qnormalized(qmul(delta_q, Quat{1, un_gyr[0] * dt / 2, un_gyr[1] * dt / 2, un_gyr[2] * dt / 2}));
        const V3 un_acc_1 = mul(toRotationMatrix(rq), acc_1 - linearized_ba);
        const V3 un_acc = 0.5 * (un_acc_0 + un_acc_1);
        const V3 rp = delta_p + delta_v * dt + 0.5 * un_acc * dt * dt;
        const V3 rv = delta_v + un_acc * dt;
        const M3 w_x = hat(un_gyr), a0_x = hat(acc_0 - linearized_ba), a1_x = hat(acc_1 - linearized_ba);
        const M3 R0 = toRotationMatrix(delta_q), R1 = toRotationMatrix(rq), I = identity3();
        M3 ImW{};
        for (int i = 0; i < 9; ++i) ImW[i] = I[i] - w_x[i] * dt;
        const M3 R0a0 = mul(R0, a0_x), R1a1 = mul(R1, a1_x), R1a1W = mul(R1a1, ImW);
        M3 R01{};
        for (int i = 0; i < 9; ++i) R01[i] = R0[i] + R1[i];
        Mat F(15, 15), V(15, 18);
        M3 b{};
        F.setBlock(0, 0, I);
        for (int i = 0; i < 9; ++i) b[i] = -0.25 * R0a0[i] * dt * dt + -0.25 * R1a1W[i] * dt * dt;
        F.setBlock(0, 3, b);
        F.setBlock(0, 6, I, dt);
        F.setBlock(0, 9, R01, -0.25 * dt * dt);
        F.setBlock(0, 12, R1a1, -0.25 * dt * dt * -dt);
        F.setBlock(3, 3, ImW);
        F.setBlock(3, 12, I, -1.0 * dt);
        for (int i = 0; i < 9; ++i) b[i] = -0.5 * R0a0[i] * dt + -0.5 * R1a1W[i] * dt;
        F.setBlock(6, 3, b);
        F.setBlock(6, 6, I);
        F.setBlock(6, 9, R01, -0.5 * dt);
        F.setBlock(6, 12, R1a1, -0.5 * dt * -dt);
        F.setBlock(9, 9, I); F.setBlock(12, 12, I);
        V.setBlock(0, 0, R0, 0.25 * dt * dt);
        V.setBlock(0, 3, R1a1, -0.25 * dt * dt * 0.5 * dt);
        V.setBlock(0, 6, R1, 0.25 * dt * dt);
        V.setBlock(0, 9, R1a1, -0.25 * dt * dt * 0.5 * dt);
        V.setBlock(3, 3, I, 0.5 * dt); V.setBlock(3, 9, I, 0.5 * dt);
        V.setBlock(6, 0, R0, 0.5 * dt);
        V.setBlock(6, 3, R1a1, -0.5 * dt * 0.5 * dt);
        V.setBlock(6, 6, R1, 0.5 * dt);
        V.setBlock(6, 9, R1a1, -0.5 * dt * 0.5 * dt);
        V.setBlock(9, 12, I, dt); V.setBlock(12, 15, I, dt);
        jacobian = matmul(F, jacobian);
        const Mat FP = matmul(matmul(F, covariance), transposed(F)), VN = matmul(matmul(V, noise), transposed(V));
        for (size_t i = 0; i < covariance.a.size(); ++i) covariance.a[i] = FP.a[i] + VN.a[i];
        delta_p = rp; delta_q = qnormalized(rq); delta_v = rv;
        sum_dt += dt;
        acc_0 = acc_1; gyr_0 = gyr_1;
    }
    NavState predict(const NavState& start, const V3& g) const {   // :145-154
        const M3 R = toRotationMatrix(start.q);
        NavState s;
        s.t = start.t + sum_dt;
        s.q = qmul(start.q, delta_q);
        s.v = mul(R, delta_v) + start.v + g * sum_dt;
        s.p = mul(R, delta_p) + start.p + start.v * sum_dt + 0.5 * g * sum_dt * sum_dt;
        s.bg = linearized_bg; s.ba = linearized_ba;
        return s;
    }
    Quat getDeltaRotation(const V3& bg) const {   // :173-179; theta2Q gives (1, theta/2), not normalised
        const V3 tem = mul(jacobian.block3(3, 12), bg - linearized_bg);
        return qmul(delta_q, Quat{1.0, tem[0] / 2, tem[1] / 2, tem[2] / 2});
    }
    V3 getDeltaVelocity(const V3& bg, const V3& ba) const {   // :183-189
        return delta_v + mul(jacobian.block3(6, 12), bg - linearized_bg) + mul(jacobian.block3(6, 9), ba - linearized_ba);
    }
    V3 getDeltaPosition(const V3& bg, const V3& ba) const {   // :193-199
        return delta_p + mul(jacobian.block3(0, 12), bg - linearized_bg) + mul(jacobian.block3(0, 9), ba - linearized_ba);
    }
    // IMUFactor's sqrt_info_ (factor.h:14): LLT(covariance^-1).matrixL().transpose()
    Mat sqrtInformation() const { return transposed(cholesky(inverse(covariance))); }

    V3 acc_0, gyr_0, linearized_ba, linearized_bg;
    Mat jacobian, covariance, noise;
    double sum_dt = 0.0;
    V3 delta_p{0, 0, 0};
    Quat delta_q;
    V3 delta_v{0, 0, 0};
};

// ---- factors
inline std::vector<double> imuResidual(const ImuPreintegration& pre, const V3& g, const NavState& last, const NavState& cur, const Mat& sqrt_info,
                                       bool reference_rotation_residual = false) {   // factor.h:18-62
    const double dt = pre.sum_dt;
    const V3 dpos = pre.getDeltaPosition(cur.bg, cur.ba), dvel = pre.getDeltaVelocity(cur.bg, cur.ba);
    const M3 dR = toRotationMatrix(pre.getDeltaRotation(cur.bg));
    const M3 Rl = transpose(toRotationMatrix(last.q));
    const M3 Rc = toRotationMatrix(cur.q);
    const V3 ep = mul(Rl, cur.p - last.p - last.v * dt - g * (dt * dt / 2)) - dpos;
    const V3 er = logRotation(mul(mul(transpose(dR), Rl), reference_rotation_residual ? transpose(Rc) : Rc));
    const V3 ev = mul(Rl, cur.v - last.v - g * dt) - dvel;
    const V3 eba = cur.ba - last.ba, ebg = cur.bg - last.bg;
    std::vector<double> r(15);
    for (int i = 0; i < 3; ++i) { r[i] = ep[i]; r[3 + i] = er[i]; r[6 + i] = ev[i]; r[9 + i] = eba[i]; r[12 + i] = ebg[i]; }
    return matvec(sqrt_info, r);
}
// ndt_pose: column-major 4x4; info: row-major 6x6, order [rotation, translation] (factor.h:243-251)
inline std::vector<double> pointCloudResidual(const std::array<double, 16>& ndt_pose, const Mat& sqrt_info, const NavState& cur) {   // factor.h:222-257
    M3 Rn{};
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rn[3 * r + c] = ndt_pose[c * 4 + r];
    const Quat qn = fromRotationMatrix(Rn);
    const V3 er = logRotation(mul(toRotationMatrix(qinverse(qn)), toRotationMatrix(cur.q)));
    std::vector<double> r = {er[0], er[1], er[2], cur.p[0] - ndt_pose[12], cur.p[1] - ndt_pose[13], cur.p[2] - ndt_pose[14]};
    return matvec(sqrt_info, r);
}
struct MargPrior { bool valid = false; NavState state; Mat Jr; std::vector<double> Er; };
inline std::vector<double> margDx(const NavState& s0, const NavState& s) {   // factor.h:166-183
    Quat tq = qmul(qinverse(s0.q), s.q);
    if (tq.w < 0) { tq.w = -tq.w; tq.x = -tq.x; tq.y = -tq.y; tq.z = -tq.z; }
    std::vector<double> dx(15);
    const double qv[3] = {tq.x, tq.y, tq.z};
    for (int i = 0; i < 3; ++i) { dx[i] = s.p[i] - s0.p[i]; dx[3 + i] = qv[i] * 2.0; dx[6 + i] = s.v[i] - s0.v[i]; dx[9 + i] = s.bg[i] - s0.bg[i]; dx[12 + i] = s.ba[i] - s0.ba[i]; }
    return dx;
}
inline std::vector<double> margResidual(const MargPrior& pr, const NavState& s) {   // factor.h:185-186
    std::vector<double> r = matvec(pr.Jr, margDx(pr.state, s));
    for (int i = 0; i < 15; ++i) r[i] += pr.Er[i];
    return r;
}
// utility.cc:95-124; eps = 1e-8 (utility.h:138)
inline void marginalize(const Mat& H, const std::vector<double>& b, Mat& Jr, std::vector<double>& Er, int m) {
    const double eps = 1e-8;
    const int n = H.r - m;
    Mat Amm(m, m), Amr(m, n), Arm(n, m), Arr(n, n);
    for (int i = 0; i < m; ++i) { for (int j = 0; j < m; ++j) Amm(i, j) = 0.5 * (H(i, j) + H(j, i)); for (int j = 0; j < n; ++j) Amr(i, j) = H(i, m + j); }
    for (int i = 0; i < n; ++i) { for (int j = 0; j < m; ++j) Arm(i, j) = H(m + i, j); for (int j = 0; j < n; ++j) Arr(i, j) = H(m + i, m + j); }
    std::vector<double> lam; Mat V;
    symmetricEigen(Amm, lam, V);
    Mat Ainv(m, m);
    for (int k = 0; k < m; ++k) {
        if (!(lam[k] > eps)) continue;
        const double il = 1.0 / lam[k];
        for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) Ainv(i, j) += V(i, k) * il * V(j, k);
    }
    const Mat ArmAinv = matmul(Arm, Ainv);
    const Mat corr = matmul(ArmAinv, Amr);
    Mat H2(n, n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) H2(i, j) = Arr(i, j) - corr(i, j);
    std::vector<double> bmm(b.begin(), b.begin() + m), b2(n);
    const std::vector<double> cb = matvec(ArmAinv, bmm);
    for (int i = 0; i < n; ++i) b2[i] = b[m + i] - cb[i];
    std::vector<double> lam2; Mat V2;
    symmetricEigen(H2, lam2, V2);
    Jr = Mat(n, n); Er.assign(n, 0.0);
    for (int k = 0; k < n; ++k) {
        const double S = lam2[k] > eps ? lam2[k] : 0.0, Sinv = lam2[k] > eps ? 1.0 / lam2[k] : 0.0;
        double vb = 0;
        for (int i = 0; i < n; ++i) { Jr(k, i) = std::sqrt(S) * V2(i, k); vb += V2(i, k) * (-b2[i]); }
        Er[k] = std::sqrt(Sinv) * vb;
    }
}

// ---- optimisation (process.cc:155-258)
inline NavState boxplus(const NavState& s, const double* d) {   // local order of the parameter blocks: p, q, v, bg, ba (process.cc:172-182)
    NavState o = s;
    for (int i = 0; i < 3; ++i) { o.p[i] = s.p[i] + d[i]; o.v[i] = s.v[i] + d[6 + i]; o.bg[i] = s.bg[i] + d[9 + i]; o.ba[i] = s.ba[i] + d[12 + i]; }
    o.q = qnormalized(qmul(s.q, expQuat({d[3], d[4], d[5]})));
    return o;
}
struct OptimizeOptions {
    int max_iterations = 10;                     // options.max_num_iterations, process.cc:218
    double cauchy_scale = 3.0;                   // ceres::CauchyLoss(3.0), process.cc:158
    bool reference_rotation_residual = false;    // factor.h:43 as written
};
struct OptimizeResult { NavState last, current; MargPrior prior; int iterations = 0; double cost = 0; };

inline OptimizeResult optimize(const ImuPreintegration& pre, const V3& g, const NavState& last_in, const NavState& cur_in,
                               const std::array<double, 16>& ndt_pose, const Mat& ndt_info, const MargPrior& prior, const OptimizeOptions& opt = {}) {
    const Mat sq_imu = pre.sqrtInformation();
    const Mat sq_pc = transposed(cholesky(ndt_info));   // factor.h:224
    const double a2 = opt.cauchy_scale * opt.cauchy_scale;
    using Blocks = std::vector<std::vector<double>>;
    auto blocks = [&](const NavState& l, const NavState& c) {
        Blocks out;
        if (prior.valid) out.push_back(margResidual(prior, l));
        out.push_back(imuResidual(pre, g, l, c, sq_imu, opt.reference_rotation_residual));
        out.push_back(pointCloudResidual(ndt_pose, sq_pc, c));
        return out;
    };
    auto sqn = [](const std::vector<double>& b) { double s = 0; for (double v : b) s += v * v; return s; };
    auto cost = [&](const Blocks& bl) { double c = 0; for (const auto& b : bl) c += a2 * std::log1p(sqn(b) / a2); return 0.5 * c; };
    auto weights = [&](const Blocks& bl) { std::vector<double> w; for (const auto& b : bl) w.push_back(std::sqrt(1.0 / (1.0 + sqn(b) / a2))); return w; };
    auto wres = [&](const NavState& l, const NavState& c, const std::vector<double>& w) {
        std::vector<double> r;
        const Blocks bl = blocks(l, c);
        for (size_t k = 0; k < bl.size(); ++k) for (double v : bl[k]) r.push_back(v * w[k]);
        return r;
    };
    auto jacobian = [&](const NavState& l, const NavState& c, const std::vector<double>& w, const std::vector<int>& cols, std::vector<double>& r0) {
        const double h = 1e-6;
        r0 = wres(l, c, w);
        Mat J((int)r0.size(), 30);
        for (int k : cols) {
            double d[30] = {0};
            d[k] = h;
            const std::vector<double> rp = wres(boxplus(l, d), boxplus(c, d + 15), w);
            d[k] = -h;
            const std::vector<double> rm = wres(boxplus(l, d), boxplus(c, d + 15), w);
            for (int i = 0; i < J.r; ++i) J(i, k) = (rp[i] - rm[i]) / (2 * h);
        }
        return J;
    };
    std::vector<int> cols;
    for (int k = prior.valid ? 0 : 15; k < 30; ++k) cols.push_back(k);   // no prior: the previous state is held constant (process.cc:190-198)
    const int nc = (int)cols.size();
    OptimizeResult res;
    NavState last = last_in, cur = cur_in;
    double lam = 1e-4, c0 = cost(blocks(last, cur));
    int its = 0;
    for (its = 1; its <= opt.max_iterations; ++its) {
        const std::vector<double> w = weights(blocks(last, cur));
        std::vector<double> r;
        const Mat J = jacobian(last, cur, w, cols, r);
        Mat H(nc, nc);
        std::vector<double> gr(nc, 0.0);
        for (int a = 0; a < nc; ++a) {
            for (int b = 0; b < nc; ++b) { double s = 0; for (int i = 0; i < J.r; ++i) s += J(i, cols[a]) * J(i, cols[b]); H(a, b) = s; }
            double s = 0; for (int i = 0; i < J.r; ++i) s += J(i, cols[a]) * r[i]; gr[a] = s;
        }
        bool stepped = false;
        double dxn = 0;
        while (lam <= 1e8) {
            Mat A = H;
            for (int a = 0; a < nc; ++a) A(a, a) += lam * H(a, a) + 1e-12;
            std::vector<double> rhs(nc);
            for (int a = 0; a < nc; ++a) rhs[a] = -gr[a];
            const std::vector<double> dx = solve(A, rhs);
            double d[30] = {0};
            dxn = 0;
            for (int a = 0; a < nc; ++a) { d[cols[a]] = dx[a]; dxn += dx[a] * dx[a]; }
            dxn = std::sqrt(dxn);
            const NavState l2 = boxplus(last, d), c2 = boxplus(cur, d + 15);
            const double cn = cost(blocks(l2, c2));
            if (cn < c0) { last = l2; cur = c2; c0 = cn; lam = std::max(lam / 10.0, 1e-12); stepped = true; break; }
            lam *= 10.0;
        }
        if (!stepped || dxn < 1e-8) break;
    }
    if (its > opt.max_iterations) its = opt.max_iterations;
    // marginalisation of the previous state (process.cc:237-256), all 30 columns
    {
        const std::vector<double> w = weights(blocks(last, cur));
        std::vector<int> all;
        for (int k = 0; k < 30; ++k) all.push_back(k);
        std::vector<double> r;
        const Mat J = jacobian(last, cur, w, all, r);
        Mat H(30, 30);
        std::vector<double> b(30, 0.0);
        for (int a = 0; a < 30; ++a) {
            for (int c = 0; c < 30; ++c) { double s = 0; for (int i = 0; i < J.r; ++i) s += J(i, a) * J(i, c); H(a, c) = s; }
            double s = 0; for (int i = 0; i < J.r; ++i) s += J(i, a) * r[i]; b[a] = -s;
        }
        marginalize(H, b, res.prior.Jr, res.prior.Er, 15);
        res.prior.valid = true;
        res.prior.state = cur;       // MargFactor(last_state_, ...) of the NEXT frame: last_state_ = current_state_ (process.cc:102,186-187)
    }
    res.last = last; res.current = cur; res.iterations = its; res.cost = c0;
    return res;
}

// ---- the per-frame loop (process.cc:6-153, 318-349) around any scan matcher with
//        bool beginFrame(cloud, pose16)   Matrix4d ndtPose()   bool endFrame(pose16)          (ScanToMapOdometry)
template <class Odometry>
class LioPipeline {
  public:
    struct Options {
        StaticImuInit::Options init;
        double acc_w = 1e-4, gyr_w = 1e-5;     // ACC_W / GYR_W of config/slam.yaml (process.cc:369-370)
        OptimizeOptions optimize;
        double ga2 = (2.0 * 3.14 / 180) * (2.0 * 3.14 / 180), gp2 = 0.1 * 0.1;   // ndt_info_ diagonal, process.cc:205-208
        bool use_ndt_hessian = false;          // odometry.ndt().getInformation() instead of the diagonal
    };
    explicit LioPipeline(Odometry& odo) : odo_(odo) {}
    Options& options() { return opt_; }

    void addIMU(const ImuSample& imu) { last_timestamp_imu_ = imu.t; imu_deque_.push_back(imu); }   // process.h:20-24

    // One LiDAR frame stamped t (process.cc:6-41 dataAlign + 43-58 dataProcess).  Returns false when the frame has to
    // wait for more IMU data (call again after addIMU) or was consumed by the initialisation.
    template <class Cloud> bool addScan(double t, const Cloud& cloud) {
        if (imu_deque_.empty()) return false;
        if (last_timestamp_imu_ <= t) return false;
        std::vector<ImuSample> imus;
        double imu_time = imu_deque_.front().t;
        while (!imu_deque_.empty() && imu_time < t) {   // also takes the first sample at / after t, as the reference does
            imu_time = imu_deque_.front().t;
            imus.push_back(imu_deque_.front());
            imu_deque_.pop_front();
        }
        if (!imu_init_flag_) {   // initIMU, process.cc:331-349
            if (!init_.init_success) {
                const double aw = opt_.acc_w * opt_.acc_w, gw = opt_.gyr_w * opt_.gyr_w;
                init_.options = opt_.init;
                init_.cov_acce_w = {aw, 0, 0, 0, aw, 0, 0, 0, aw}; init_.cov_gyro_w = {gw, 0, 0, 0, gw, 0, 0, 0, gw};
            }
            for (const auto& s : imus) init_.addIMU(s);
            if (init_.init_success && !imus.empty()) {
                last_imu_ = imus.back(); have_last_imu_ = true;
                newPreintegration(init_.init_ba, init_.init_bg);
                imu_init_flag_ = true;
                current_.t = imus.back().t;
                last_ = current_;
            }
            return false;   // data before a successful initialisation is dropped (process.cc:45-50)
        }
        for (const auto& s : imus) {   // predictByIMU, process.cc:318-329
            if (have_last_imu_) pre_->propagate(s.t - last_imu_.t, s.acce, s.gyro);
            last_imu_ = s; have_last_imu_ = true;
        }
        // alignByPC, process.cc:60-153
        if (first_scan_) {
            odo_.beginFrame(cloud, last_.transform());
            newPreintegration(init_.init_ba, init_.init_bg);
            first_scan_ = false;
            trajectory_.push_back(last_);
            return true;
        }
        current_ = pre_->predict(last_, init_.gravity);                       // process.cc:90
        odo_.beginFrame(cloud, current_.transform());                         // process.cc:93-94
        const auto ndt_pose = odo_.ndtPose();                                 // process.cc:95
        Mat info(6, 6);
        if (opt_.use_ndt_hessian) {
            const auto I = odo_.ndt().getInformation();
            for (int i = 0; i < 36; ++i) info.a[i] = I[i];
        } else {
            for (int i = 0; i < 3; ++i) { info(i, i) = 1.0 / opt_.ga2; info(3 + i, 3 + i) = 1.0 / opt_.gp2; }
        }
        last_result_ = optimize(*pre_, init_.gravity, last_, current_, ndt_pose, info, prior_, opt_.optimize);   // process.cc:97
        current_ = last_result_.current;
        prior_ = last_result_.prior;
        newPreintegration(current_.ba, current_.bg);                          // process.cc:101
        last_ = current_;
        trajectory_.push_back(last_);
        odo_.endFrame(current_.transform());                                  // process.cc:128-150
        return true;
    }
    const std::vector<NavState>& trajectory() const { return trajectory_; }
    const NavState& state() const { return last_; }
    const StaticImuInit& imuInit() const { return init_; }
    const OptimizeResult& lastOptimize() const { return last_result_; }
    bool initialised() const { return imu_init_flag_; }

  private:
    void newPreintegration(const V3& ba, const V3& bg) {
        pre_ = std::make_unique<ImuPreintegration>(last_imu_.acce, last_imu_.gyro, ba, bg, init_.cov_acce_n, init_.cov_gyro_n, init_.cov_acce_w, init_.cov_gyro_w);
    }
    Odometry& odo_;
    Options opt_;
    StaticImuInit init_;
    std::deque<ImuSample> imu_deque_;
    double last_timestamp_imu_ = -1;
    bool imu_init_flag_ = false, have_last_imu_ = false, first_scan_ = true;
    ImuSample last_imu_;
    std::unique_ptr<ImuPreintegration> pre_;
    NavState current_, last_;
    MargPrior prior_;
    OptimizeResult last_result_;
    std::vector<NavState> trajectory_;
};

}  // namespace lio
}  // namespace ndt_mi355
