// CPU-only driver of include/ndt_mi355_lio.hpp for tests/test_lio_cpu.py: static initialisation, pre-integration,
// factor residuals, optimisation with and without a prior.  Input (binary, doubles): n_init, n_init x (t, gyro3,
// acce3), n_motion, n_motion x (t, gyro3, acce3), init_need_time, acc_w, gyr_w, perturbation d[15] of the predicted state,
// ndt perturbation [rotvec3, trans3], reference_rotation_residual flag.  Output: labelled lines of %.17g values.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ndt_mi355_lio.hpp"

using namespace ndt_mi355::lio;

static void rd(FILE* f, double* v, size_t n) { if (fread(v, 8, n, f) != n) exit(3); }
static void pr(const char* tag, const double* v, size_t n) { printf("%s", tag); for (size_t i = 0; i < n; ++i) printf(" %.17g", v[i]); printf("\n"); }
static void prState(const char* tag, const NavState& s) {
    const double v[17] = {s.t, s.q.w, s.q.x, s.q.y, s.q.z, s.p[0], s.p[1], s.p[2], s.v[0], s.v[1], s.v[2], s.bg[0], s.bg[1], s.bg[2], s.ba[0], s.ba[1], s.ba[2]};
    pr(tag, v, 17);
}
static std::vector<ImuSample> rdImu(FILE* f) {
    double n;
    rd(f, &n, 1);
    std::vector<ImuSample> out((size_t)n);
    for (auto& s : out) { double v[7]; rd(f, v, 7); s.t = v[0]; s.gyro = {v[1], v[2], v[3]}; s.acce = {v[4], v[5], v[6]}; }
    return out;
}
static void prPrior(const char* tag, const MargPrior& p) {      // Jr^T Jr and Jr^T Er: free of the eigenvector sign / order
    std::vector<double> a(225), b(15);
    for (int i = 0; i < 15; ++i) {
        for (int j = 0; j < 15; ++j) { double s = 0; for (int k = 0; k < 15; ++k) s += p.Jr(k, i) * p.Jr(k, j); a[i * 15 + j] = s; }
        double s = 0; for (int k = 0; k < 15; ++k) s += p.Jr(k, i) * p.Er[k]; b[i] = s;
    }
    printf("%s_JtJ", tag); for (double v : a) printf(" %.17g", v); printf("\n");
    printf("%s_JtE", tag); for (double v : b) printf(" %.17g", v); printf("\n");
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    const auto init_imu = rdImu(f), motion = rdImu(f);
    double par[3], d[15], nd[6], flag;
    rd(f, par, 3); rd(f, d, 15); rd(f, nd, 6); rd(f, &flag, 1);
    fclose(f);

    StaticImuInit init;
    init.options.init_need_time = par[0];
    init.cov_acce_w = {par[1] * par[1], 0, 0, 0, par[1] * par[1], 0, 0, 0, par[1] * par[1]};
    init.cov_gyro_w = {par[2] * par[2], 0, 0, 0, par[2] * par[2], 0, 0, 0, par[2] * par[2]};
    for (const auto& s : init_imu) init.addIMU(s);
    printf("init_success %d\n", (int)init.init_success);
    pr("gravity", init.gravity.data(), 3); pr("init_bg", init.init_bg.data(), 3); pr("init_ba", init.init_ba.data(), 3);
    pr("cov_gyro_n", init.cov_gyro_n.data(), 3); pr("cov_acce_n", init.cov_acce_n.data(), 3);
    if (!init.init_success) return 0;

    ImuPreintegration pre(init_imu.back().acce, init_imu.back().gyro, init.init_ba, init.init_bg, init.cov_acce_n, init.cov_gyro_n, init.cov_acce_w, init.cov_gyro_w);
    double tl = init_imu.back().t;
    for (const auto& s : motion) { pre.propagate(s.t - tl, s.acce, s.gyro); tl = s.t; }
    pr("sum_dt", &pre.sum_dt, 1); pr("delta_p", pre.delta_p.data(), 3); pr("delta_v", pre.delta_v.data(), 3);
    const double dq[4] = {pre.delta_q.w, pre.delta_q.x, pre.delta_q.y, pre.delta_q.z};
    pr("delta_q", dq, 4);
    pr("jacobian", pre.jacobian.a.data(), 225); pr("covariance", pre.covariance.a.data(), 225);

    NavState last;
    last.t = init_imu.back().t; last.bg = init.init_bg; last.ba = init.init_ba;
    const NavState pred = pre.predict(last, init.gravity);
    prState("predict", pred);
    const NavState cur = boxplus(pred, d);
    const Mat sq = pre.sqrtInformation();
    const auto r_imu = imuResidual(pre, init.gravity, last, cur, sq, flag != 0);
    pr("imu_residual", r_imu.data(), 15);
    // an "NDT pose": the predicted pose moved by nd
    NavState np_ = pred;
    double dn[15] = {nd[3], nd[4], nd[5], nd[0], nd[1], nd[2], 0, 0, 0, 0, 0, 0, 0, 0, 0};
    np_ = boxplus(pred, dn);
    std::array<double, 16> ndt_pose{};
    { const M3 R = toRotationMatrix(np_.q); for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) ndt_pose[c * 4 + r] = R[3 * r + c]; ndt_pose[12 + r] = np_.p[r]; } ndt_pose[15] = 1; }
    Mat info(6, 6);
    const double ga2 = (2.0 * 3.14 / 180) * (2.0 * 3.14 / 180), gp2 = 0.1 * 0.1;   // process.cc:205-208
    for (int i = 0; i < 3; ++i) { info(i, i) = 1.0 / ga2; info(3 + i, 3 + i) = 1.0 / gp2; }
    const auto r_pc = pointCloudResidual(ndt_pose, transposed(cholesky(info)), cur);
    pr("pc_residual", r_pc.data(), 6);

    OptimizeOptions oo;
    oo.reference_rotation_residual = flag != 0;
    MargPrior none;
    const OptimizeResult r1 = optimize(pre, init.gravity, last, cur, ndt_pose, info, none, oo);
    prState("opt1_last", r1.last); prState("opt1_cur", r1.current);
    printf("opt1_iterations %d\nopt1_cost %.17g\n", r1.iterations, r1.cost);
    prPrior("opt1", r1.prior);
    const auto r_m = margResidual(r1.prior, boxplus(r1.current, d));
    pr("marg_residual", r_m.data(), 15);
    // second frame with the prior: same pre-integration again from the optimised state
    const NavState pred2 = pre.predict(r1.current, init.gravity);
    const NavState cur2 = boxplus(pred2, d);
    NavState np2 = boxplus(pred2, dn);
    std::array<double, 16> ndt_pose2{};
    { const M3 R = toRotationMatrix(np2.q); for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) ndt_pose2[c * 4 + r] = R[3 * r + c]; ndt_pose2[12 + r] = np2.p[r]; } ndt_pose2[15] = 1; }
    const OptimizeResult r2 = optimize(pre, init.gravity, r1.current, cur2, ndt_pose2, info, r1.prior, oo);
    prState("opt2_last", r2.last); prState("opt2_cur", r2.current);
    printf("opt2_iterations %d\nopt2_cost %.17g\n", r2.iterations, r2.cost);
    prPrior("opt2", r2.prior);
    return 0;
}
