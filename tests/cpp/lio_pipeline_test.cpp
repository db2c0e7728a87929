// Replays an IMU + LiDAR stream through LioPipeline<ScanToMapOdometry> (process.cc:6-153, 318-349 around the MI355X
// NDT engine).  Input (binary): header doubles {init_need_time, scan_leaf, acc_w, gyr_w, use_ndt_hessian},
// T_imu_lidar[16], n_imu, n_imu x (t, gyro3, acce3), n_scans, per scan {t, n} + n x xyz float32.
// Output per LiDAR frame that was processed: frame index, keyframes, NDT iterations, state (t q p v bg ba).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ndt_mi355_lio.hpp"
#include "ndt_mi355_odometry.hpp"

struct P3 { float x, y, z; };
using namespace ndt_mi355;

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    double hd[5];
    ScanToMapOdometry<P3> odo;
    if (fread(hd, 8, 5, f) != 5 || fread(odo.options().T_imu_lidar.data(), 8, 16, f) != 16) return 3;
    odo.options().scan_leaf = (float)hd[1];
    lio::LioPipeline<ScanToMapOdometry<P3>> pipe(odo);
    pipe.options().init.init_need_time = hd[0];
    pipe.options().acc_w = hd[2]; pipe.options().gyr_w = hd[3];
    pipe.options().use_ndt_hessian = hd[4] != 0;
    double n_imu = 0;
    if (fread(&n_imu, 8, 1, f) != 1) return 3;
    std::vector<lio::ImuSample> imu((size_t)n_imu);
    for (auto& s : imu) { double v[7]; if (fread(v, 8, 7, f) != 7) return 3; s.t = v[0]; s.gyro = {v[1], v[2], v[3]}; s.acce = {v[4], v[5], v[6]}; }
    double n_scans = 0;
    if (fread(&n_scans, 8, 1, f) != 1) return 3;
    size_t next_imu = 0;
    for (int k = 0; k < (int)n_scans; ++k) {
        double sh[2];
        if (fread(sh, 8, 2, f) != 2) return 3;
        std::vector<P3> scan((size_t)sh[1]);
        if (!scan.empty() && fread(scan.data(), 12, scan.size(), f) != scan.size()) return 3;
        while (next_imu < imu.size() && imu[next_imu].t <= sh[0] + 0.02) pipe.addIMU(imu[next_imu++]);   // the IMU runs ahead of the LiDAR
        const bool done = pipe.addScan(sh[0], scan);
        if (!done) { printf("%d skipped %d\n", k, (int)pipe.initialised()); continue; }
        const lio::NavState& s = pipe.state();
        printf("%d %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", k, odo.keyframes(),
               odo.ndt().getFinalNumIteration(), s.t, s.q.w, s.q.x, s.q.y, s.q.z, s.p[0], s.p[1], s.p[2], s.v[0], s.v[1], s.v[2], s.bg[0], s.bg[1], s.bg[2],
               s.ba[0], s.ba[1], s.ba[2]);
    }
    fclose(f);
    return 0;
}
