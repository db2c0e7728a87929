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
    PointCloud output;
    ndt_.align(output, guess);           // process.cc:94   (Matrix4f guess)
    auto T = ndt_.getFinalTransformation();   // process.cc:95
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
    ndt_.align(output, guess);
    auto Tp = ndt_.getFinalTransformation();
    printf("status %d converged %d iterations %d probability %.17g\n", ndt_.status(), (int)ndt_.hasConverged(), ndt_.getFinalNumIteration(),
           ndt_.getTransformationProbability());
    for (double v : Tp) printf("%.17g ", v);
    printf("\n");
    return 0;
}
