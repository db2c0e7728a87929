// ndt_mi355.hpp — header-only C++ facade over the C ABI (ndt_mi355.h) with the call surface lio-slam uses on
// its `pcl::NormalDistributionsTransform<slam_czc::Point, slam_czc::Point> ndt_` member
// (src/slam/include/process.h:86).  The complete surface the reference touches is eight methods:
//   setResolution / setMaximumIterations / setTransformationEpsilon / setStepSize   (src/slam/src/process.cc:373-376)
//   setInputTarget (process.cc:78,147)   setInputSource (process.cc:87)
//   align(output, guess) (process.cc:94)   getFinalTransformation() (process.cc:95)
// plus what the north star adds: hasConverged(), getFinalNumIteration(), getHessian() (the 6x6 information for
// PointCloudFactor, src/slam/include/factor.h:220-269) and addTargetCloud() (incremental keyframe merge in place
// of the rebuild at process.cc:134-147).
//
// No PCL / Eigen types are needed: clouds are anything with data()/size() of records whose first 12 bytes are
// float x,y,z (pcl::PointCloud<P>::points is a std::vector<P>; use cloud.points) and matrices are plain
// column-major arrays, the memory layout of Eigen::Matrix4f / Matrix4d.  With Eigen present:
//   Eigen::Matrix4f guess = state.getTransform();
//   ndt_.align(guess.data());  Eigen::Map<const Eigen::Matrix4f>(ndt_.getFinalTransformationF().data());
#pragma once
#include <array>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "ndt_mi355.h"

namespace ndt_mi355 {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

template <typename PointSource, typename PointTarget = PointSource>
class NormalDistributionsTransform {
    static_assert(sizeof(PointSource) >= 12 && sizeof(PointSource) % 4 == 0, "point record must start with float x,y,z");
    static_assert(sizeof(PointTarget) >= 12 && sizeof(PointTarget) % 4 == 0, "point record must start with float x,y,z");

  public:
    using Matrix4d = std::array<double, 16>;   // column-major
    using Matrix4f = std::array<float, 16>;
    using Matrix6d = std::array<double, 36>;

    explicit NormalDistributionsTransform(int device = 0) {
        ndt_default_config(&cfg_);
        ndt_pcl_default_params(&pcl_);
        cfg_.device = device;
    }
    ~NormalDistributionsTransform() { ndt_destroy(h_); }
    NormalDistributionsTransform(const NormalDistributionsTransform&) = delete;
    NormalDistributionsTransform& operator=(const NormalDistributionsTransform&) = delete;

    // --- the reference's setters (process.cc:373-376)
    void setResolution(float r) { cfg_.resolution = r; reconfigure(); }
    void setMaximumIterations(int n) { cfg_.max_iterations = n; reconfigure(); }
    void setTransformationEpsilon(double e) { cfg_.eps = e; reconfigure(); }
    // More-Thuente step cap of PCL's Newton NDT: used in PCL compatibility mode; Gauss-Newton has no line search
    void setStepSize(double s) { pcl_.step_size = s; }
    // PCL compatibility mode (ndt_pcl_align): the objective, Newton step and More-Thuente line search of
    // pcl::NormalDistributionsTransform, i.e. what the reference's build runs at process.cc:94.  Off by default:
    // the engine's Gauss-Newton NEARBY7 form is the north-star path.
    void setPclCompatibility(bool on) { pcl_mode_ = on; }
    void setOulierRatio(double r) { pcl_.outlier_ratio = r; }     // (sic) PCL's spelling
    void setOutlierRatio(double r) { pcl_.outlier_ratio = r; }
    double getTransformationProbability() const { return pres_.trans_probability; }
    const ndt_pcl_result& pclResult() const { return pres_; }
    // --- parameters PCL leaves at its defaults
    void setNearby(int k) { cfg_.nearby = k; reconfigure(); }
    void setMinPointsPerVoxel(int n) { cfg_.min_pts_in_voxel = n; reconfigure(); }
    void setOutlierThreshold(double t) { cfg_.res_outlier_th = t; reconfigure(); }
    const ndt_config& config() const { return cfg_; }

    // --- clouds.  The engine copies into HBM; the caller's buffer is free on return (PCL keeps the pointer).
    template <class Cloud> void setInputTarget(const Cloud& c) { check(ndt_set_target(handle(), c.data(), c.size(), sizeof(PointTarget))); }
    template <class Cloud> void setInputTarget(const std::shared_ptr<Cloud>& c) { setInputTarget(*c); }
    template <class Cloud> void addTargetCloud(const Cloud& c, const double* T_colmajor = nullptr) {
        check(ndt_add_target(handle(), c.data(), c.size(), sizeof(PointTarget), T_colmajor));
    }
    template <class Cloud> void setInputSource(const Cloud& c) { check(ndt_set_source(handle(), c.data(), c.size(), sizeof(PointSource))); }
    template <class Cloud> void setInputSource(const std::shared_ptr<Cloud>& c) { setInputSource(*c); }

    // --- align.  PCL's first argument (`output`, a scratch cloud the reference never reads, process.cc:92) is
    // accepted and left untouched.  Errors do not throw here (PCL prints and clears converged_; the reference
    // ignores both, process.cc:94-95): status() tells.
    void align(const double* guess_colmajor) {
        if (pcl_mode_) {
            status_ = ndt_pcl_align(handle(), &pcl_, guess_colmajor, &pres_);
            res_ = ndt_result{};
            for (int i = 0; i < 16; ++i) res_.T[i] = pres_.T[i];
            for (int i = 0; i < 36; ++i) res_.H[i] = -pres_.hessian[i];   // the score is maximised: -Hessian is the curvature
            for (int i = 0; i < 6; ++i) res_.b[i] = pres_.gradient[i];    // (order [t, euler angles], not [rot, trans])
            res_.iterations = pres_.iterations;
            res_.converged = status_ == NDT_OK ? pres_.converged : 0;
            res_.status = status_;
            return;
        }
        status_ = ndt_align(handle(), guess_colmajor, &res_);
        if (status_ != NDT_OK) res_.converged = 0;
    }
    void align(const float* guess_colmajor) {
        double g[16];
        for (int i = 0; i < 16; ++i) g[i] = guess_colmajor[i];
        align(g);
    }
    template <class Cloud, class Scalar> void align(Cloud& /*output*/, const Scalar* guess_colmajor) { align(guess_colmajor); }

    Matrix4d getFinalTransformation() const { Matrix4d m; for (int i = 0; i < 16; ++i) m[i] = res_.T[i]; return m; }
    Matrix4f getFinalTransformationF() const { Matrix4f m; for (int i = 0; i < 16; ++i) m[i] = (float)res_.T[i]; return m; }
    bool hasConverged() const { return res_.converged != 0; }
    int getFinalNumIteration() const { return res_.iterations; }
    Matrix6d getHessian() const { Matrix6d m; for (int i = 0; i < 36; ++i) m[i] = res_.H[i]; return m; }
    // H as a positive-definite information matrix for PointCloudFactor::Create (src/slam/src/process.cc:205-211)
    Matrix6d getInformation(double scale = 1.0, double rel_floor = 1e-6) const {
        Matrix6d m{};
        if (ndt_information_from_hessian(res_.H, scale, rel_floor, m.data()) != NDT_OK) m.fill(0.0);
        return m;
    }
    // Scan pre-processing and keyframe step on the device (process.cc:62-73,87 and 134-147): no host round trip
    template <class Cloud> void setInputSourceRaw(const Cloud& c, float leaf, const double* T_extrinsic_colmajor = nullptr) {
        check(ndt_set_source_downsampled(handle(), c.data(), c.size(), sizeof(PointSource), leaf, T_extrinsic_colmajor));
    }
    void addSourceToTarget(const double* T_colmajor) { check(ndt_add_target_from_source(handle(), T_colmajor)); }
    // Voxel-map serialisation (ndt_map_save / ndt_map_load): e.g. to resume a session without the keyframe clouds
    std::vector<unsigned char> saveMap() {
        size_t need = 0;
        check(ndt_map_save(handle(), nullptr, 0, &need));
        std::vector<unsigned char> blob(need);
        check(ndt_map_save(handle(), blob.data(), blob.size(), &need));
        return blob;
    }
    void loadMap(const std::vector<unsigned char>& blob) { check(ndt_map_load(handle(), blob.data(), blob.size())); }
    // Offline replay (ndt_align_batch; no counterpart in the reference, which aligns one frame at a time): many
    // (scan, initial pose) pairs against the resident target in one batched Gauss-Newton loop.  pair_scan[p] is the
    // scan of pair p (empty: pair p uses scan p); guesses are column-major 4x4.  results[p].status tells how pair p ended.
    template <class Cloud>
    std::vector<ndt_result> alignBatch(const std::vector<Cloud>& scans, const std::vector<int32_t>& pair_scan,
                                       const std::vector<Matrix4d>& guesses) {
        std::vector<PointSource> pts;
        std::vector<int64_t> off(scans.size() + 1, 0);
        for (size_t i = 0; i < scans.size(); ++i) {
            off[i + 1] = off[i] + (int64_t)scans[i].size();
            pts.insert(pts.end(), scans[i].data(), scans[i].data() + scans[i].size());
        }
        std::vector<double> g(16 * guesses.size());
        for (size_t p = 0; p < guesses.size(); ++p) for (int i = 0; i < 16; ++i) g[16 * p + i] = guesses[p][i];
        std::vector<ndt_result> out(guesses.size());
        check(ndt_align_batch(handle(), pts.data(), sizeof(PointSource), off.data(), scans.size(), pair_scan.empty() ? nullptr : pair_scan.data(),
                              guesses.size(), g.data(), out.data()));
        return out;
    }
    const ndt_result& result() const { return res_; }
    int status() const { return status_; }
    std::string lastError() const { return h_ ? ndt_last_error(h_) : "no handle"; }

  private:
    ndt_handle* handle() {
        if (!h_) {
            h_ = ndt_create(&cfg_);
            if (!h_) throw Error(NDT_ERR_HIP, "ndt_create failed: no MI355X visible or bad configuration");
        }
        return h_;
    }
    void reconfigure() { if (h_) check(ndt_configure(h_, &cfg_)); }
    void check(int rc) { if (rc != NDT_OK) throw Error(rc, ndt_last_error(h_)); }

    ndt_config cfg_{};
    ndt_pcl_params pcl_{};
    ndt_pcl_result pres_{};
    bool pcl_mode_ = false;
    ndt_handle* h_ = nullptr;
    ndt_result res_{};
    int status_ = NDT_OK;
};

}  // namespace ndt_mi355
