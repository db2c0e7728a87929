// Drives the C++ facade the way src/slam/src/process.cc drives its PCL member (lines 373-376, 78, 87, 94-95, 147),
// on clouds of the reference's own 32-byte point record.  Input: a binary file written by tests/test_gpu_facade.py;
// output: pose, Hessian, iteration count as text.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "ndt_mi355.hpp"

struct alignas(16) Point {   // slam_czc::Point (src/slam/include/utility.h:54-61)
    float x, y, z, w;
    float intensity, time;
    std::uint16_t ring;
};
static_assert(sizeof(Point) == 32, "layout of slam_czc::Point");
using PointCloud = std::vector<Point>;

// Stand-in for Eigen::Matrix4f (Eigen is not in the image): 16 column-major floats behind data(), default-constructible,
// copyable — what the facade needs of `Eigen::Matrix4f ndt_pose_` (src/slam/include/process.h:87).
struct Matrix4f {
    float m[16];
    float* data() { return m; }
    const float* data() const { return m; }
    float& operator()(int r, int c) { return m[c * 4 + r]; }
};
// Stand-in for State::getTransform() (src/slam/src/utility.cc:22-35): returns the Matrix4f by value
struct State {
    Matrix4f T;
    Matrix4f getTransform() const { return T; }
};

static std::shared_ptr<PointCloud> read_cloud(FILE* f) {
    std::uint64_t n = 0;
    if (fread(&n, 8, 1, f) != 1) exit(3);
    auto c = std::make_shared<PointCloud>(n);
    for (auto& p : *c) {
        float v[3];
        if (fread(v, 4, 3, f) != 3) exit(3);
        p = Point{v[0], v[1], v[2], 1.f, 0.f, 0.f, 0};
    }
    return c;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    auto pc_key = read_cloud(f), keyframe = read_cloud(f), scan = read_cloud(f);
    float guess[16];
    double Tk[16];
    if (fread(guess, 4, 16, f) != 16 || fread(Tk, 8, 16, f) != 16) return 3;
    fclose(f);

    ndt_mi355::NormalDistributionsTransform<Point, Point> ndt_;
    ndt_.setResolution(1.0);             // process.cc:373-376
    ndt_.setMaximumIterations(35);
    ndt_.setTransformationEpsilon(0.01);
    ndt_.setStepSize(0.1);
    ndt_.setInputTarget(pc_key);         // process.cc:78
    ndt_.addTargetCloud(*keyframe, Tk);  // process.cc:134-147, incrementally
    ndt_.setInputSource(scan);           // process.cc:87
    // process.cc:92-95 exactly as the reference has them (PointType::Ptr output; Eigen::Matrix4f ndt_pose_)
    State current_state_;
    for (int i = 0; i < 16; ++i) current_state_.T.m[i] = guess[i];
    Matrix4f ndt_pose_;
    std::shared_ptr<PointCloud> output(new PointCloud);
    ndt_pose_ = current_state_.getTransform();
    ndt_.align(*output, ndt_pose_);
    ndt_pose_ = ndt_.getFinalTransformation();
    std::array<double, 16> T = ndt_.getFinalTransformation();
    double m4f_diff = 0.0;                                // the Matrix4f is the float32 rounding of the double pose
    for (int i = 0; i < 16; ++i) m4f_diff = std::max(m4f_diff, std::fabs((double)ndt_pose_.m[i] - (double)(float)T[i]));
    auto H = ndt_.getHessian();
    printf("status %d converged %d iterations %d\n", ndt_.status(), (int)ndt_.hasConverged(), ndt_.getFinalNumIteration());
    for (double v : T) printf("%.17g ", v);
    printf("\n");
    for (double v : H) printf("%.17g ", v);
    printf("\n");
    // offline replay through the facade: the same pair twice and once from the result pose
    {
        std::array<double, 16> g0, g1 = T;
        for (int i = 0; i < 16; ++i) g0[i] = guess[i];
        std::vector<PointCloud> scans{*scan};
        auto rb = ndt_.alignBatch(scans, {0, 0, 0}, {g0, g1, g0});
        double d = 0.0;
        for (int i = 0; i < 16; ++i) d = std::max(d, std::fabs(rb[0].T[i] - T[i]));
        bool same = true;
        for (int i = 0; i < 16; ++i) same = same && rb[0].T[i] == rb[2].T[i];
        printf("batch %d %d %d %.3g %d\n", rb[0].status, rb[1].status, rb[0].iterations, d, (int)same);
    }
    // the same call sequence in PCL compatibility mode (what the reference's PCL build computes at process.cc:94)
    ndt_.setPclCompatibility(true);
    ndt_.align(*output, guess);          // a raw column-major float[16] works too
    std::array<double, 16> Tp = ndt_.getFinalTransformation();
    printf("status %d converged %d iterations %d probability %.17g\n", ndt_.status(), (int)ndt_.hasConverged(), ndt_.getFinalNumIteration(),
           ndt_.getTransformationProbability());
    for (double v : Tp) printf("%.17g ", v);
    printf("\n");
    printf("matrix4f %.3g\n", m4f_diff);
    // PCL-like error behaviour: a failing call does not throw; it is recorded and clears hasConverged()
    ndt_.setPclCompatibility(false);
    PointCloud empty;
    ndt_mi355::NormalDistributionsTransform<Point, Point> fresh;
    fresh.setResolution(-1.0f);          // invalid: recorded at first use
    fresh.setInputSource(scan);
    const int st_bad = fresh.status();
    ndt_mi355::NormalDistributionsTransform<Point, Point> no_target;
    no_target.setInputSource(scan);
    no_target.align(*output, ndt_pose_);
    printf("errors %d %d %d\n", st_bad, no_target.status(), (int)no_target.hasConverged());
    return 0;
}
