// Replays a sequence of raw scans through ScanToMapOdometry (the LiDAR half of process.cc:60-153).
// Input file: n_frames, scan leaf (double), T_imu_lidar[16 double], then per frame: n points, xyz float32, the
// predicted pose (what IMU pre-integration hands to process.cc:90-93).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ndt_mi355_odometry.hpp"

struct P3 { float x, y, z; };

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::uint64_t n_frames = 0;
    ndt_mi355::ScanToMapOdometry<P3> odo;
    double leaf = 0.5;
    if (fread(&n_frames, 8, 1, f) != 1 || fread(&leaf, 8, 1, f) != 1 || fread(odo.options().T_imu_lidar.data(), 8, 16, f) != 16) return 3;
    odo.options().scan_leaf = (float)leaf;
    for (std::uint64_t k = 0; k < n_frames; ++k) {
        std::uint64_t n = 0;
        if (fread(&n, 8, 1, f) != 1) return 3;
        std::vector<P3> scan(n);
        if (n && fread(scan.data(), 12, n, f) != n) return 3;
        ndt_mi355::ScanToMapOdometry<P3>::Matrix4d guess;
        if (fread(guess.data(), 8, 16, f) != 16) return 3;
        const bool aligned = odo.beginFrame(scan, guess);
        const auto pose = aligned ? odo.ndtPose() : guess;
        const bool key = aligned ? odo.endFrame(pose) : true;
        printf("%d %d %d", (int)aligned, (int)key, aligned ? odo.ndt().getFinalNumIteration() : 0);
        for (double v : pose) printf(" %.17g", v);
        printf("\n");
    }
    fclose(f);
    return 0;
}
