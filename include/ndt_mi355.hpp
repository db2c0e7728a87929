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
// No PCL / Eigen headers are needed, and the reference's call sites compile unchanged against it:
//   clouds   : anything with data()/size() of records whose first 12 bytes are float x,y,z — pcl::PointCloud<P> has both
//              (PCL >= 1.11), so has std::vector<P>; shared pointers to either are accepted (process.cc:78,87,147);
//   matrices : align(out, guess) takes any fixed 4x4 type with data() over 16 column-major floats or doubles, by value or
//              reference (Eigen::Matrix4f at process.cc:94), and getFinalTransformation() returns a value that converts to
//              any such type (`ndt_pose_ = ndt_.getFinalTransformation();`, process.cc:95) and is itself a
//              std::array<double,16>.  Raw `const float*` / `const double*` column-major pointers work too.
// Error behaviour follows PCL's: the eight reference methods and the other void methods never throw — a failure is
// printed on stderr (PCL_ERROR's role), recorded in status() / lastError(), and clears hasConverged(); the reference
// checks none of it (process.cc:94-95).  setThrowOnError(true) turns every recorded failure into an ndt_mi355::Error.
// Methods that return data (saveMap, alignBatch) always throw on failure.
#pragma once
#include <array>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "ndt_mi355.h"

namespace ndt_mi355 {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

namespace detail {
// a 4x4 matrix type: has data() over float or double (Eigen::Matrix4f / Matrix4d, std::array<float,16>, ...)
template <class M, class = void> struct matrix_scalar { using type = void; };
template <class M>
struct matrix_scalar<M, std::void_t<decltype(std::declval<const M&>().data())>> {
    using type = std::remove_cv_t<std::remove_pointer_t<decltype(std::declval<const M&>().data())>>;
};
template <class M>
inline constexpr bool is_matrix_v = std::is_same_v<typename matrix_scalar<M>::type, float> || std::is_same_v<typename matrix_scalar<M>::type, double>;
}  // namespace detail

// What getFinalTransformation() returns: a column-major double[16] (std::array) that converts to any 4x4 matrix type
// with a default constructor and a mutable data() over 16 floats or doubles — `Eigen::Matrix4f m = ...;` and
// `ndt_pose_ = ndt_.getFinalTransformation();` (process.cc:95) compile as written against PCL.
struct FinalTransformation : std::array<double, 16> {
    template <class M, class = std::enable_if_t<detail::is_matrix_v<M> && !std::is_base_of_v<std::array<double, 16>, M>>>
    operator M() const {
        M m;
        using S = typename detail::matrix_scalar<M>::type;
        for (int i = 0; i < 16; ++i) m.data()[i] = (S)(*this)[i];
        return m;
    }
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

    // failures are recorded (status(), lastError()) and printed; true: they are thrown as ndt_mi355::Error as well
    void setThrowOnError(bool on) { throw_ = on; }

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
    // IncNdt3d's options: freeze voxels beyond n points (0 = never); (Sigma + 1e-3 I)^-1 instead of the eigenvalue floor
    void setMaxPointsPerVoxel(int n) { cfg_.max_pts_in_voxel = n; reconfigure(); }
    void setInformationRule(int rule) { cfg_.info_rule = rule; reconfigure(); }
    void setVoxelCapacity(int capacity) { cfg_.capacity = capacity; reconfigure(); }   // IncNdt3d capacity_: capacity - 1 voxels kept, least recently touched go
    void setOutlierThreshold(double t) { cfg_.res_outlier_th = t; reconfigure(); }
    const ndt_config& config() const { return cfg_; }

    // --- clouds.  The engine copies into HBM; the caller's buffer is free on return (PCL keeps the pointer).
    template <class Cloud> void setInputTarget(const Cloud& c) {
        if (ndt_handle* h = handle("setInputTarget")) note("setInputTarget", ndt_set_target(h, c.data(), c.size(), sizeof(PointTarget)));
    }
    template <class Cloud> void setInputTarget(const std::shared_ptr<Cloud>& c) { setInputTarget(*c); }
    template <class Cloud> void addTargetCloud(const Cloud& c, const double* T_colmajor = nullptr) {
        if (ndt_handle* h = handle("addTargetCloud")) note("addTargetCloud", ndt_add_target(h, c.data(), c.size(), sizeof(PointTarget), T_colmajor));
    }
    template <class Cloud> void setInputSource(const Cloud& c) {
        if (ndt_handle* h = handle("setInputSource")) note("setInputSource", ndt_set_source(h, c.data(), c.size(), sizeof(PointSource)));
    }
    template <class Cloud> void setInputSource(const std::shared_ptr<Cloud>& c) { setInputSource(*c); }

    // --- align.  PCL's first argument (`output`, a scratch cloud the reference never reads, process.cc:92) is
    // accepted and left untouched.  Errors do not throw here (PCL prints and clears converged_; the reference
    // ignores both, process.cc:94-95): status() tells.
    void align(const double* guess_colmajor) {
        ndt_handle* h = handle("align");
        if (!h) { res_ = ndt_result{}; res_.status = status_; return; }
        if (pcl_mode_) {
            status_ = ndt_pcl_align(h, &pcl_, guess_colmajor, &pres_);
            res_ = ndt_result{};
            for (int i = 0; i < 16; ++i) res_.T[i] = pres_.T[i];
            for (int i = 0; i < 36; ++i) res_.H[i] = -pres_.hessian[i];   // the score is maximised: -Hessian is the curvature
            for (int i = 0; i < 6; ++i) res_.b[i] = pres_.gradient[i];    // (order [t, euler angles], not [rot, trans])
            res_.iterations = pres_.iterations;
            res_.converged = status_ == NDT_OK ? pres_.converged : 0;
            res_.status = status_;
            note("align", status_);
            return;
        }
        status_ = ndt_align(h, guess_colmajor, &res_);
        if (status_ != NDT_OK) res_.converged = 0;
        note("align", status_);
    }
    void align(const float* guess_colmajor) {
        double g[16];
        for (int i = 0; i < 16; ++i) g[i] = guess_colmajor[i];
        align(g);
    }
    template <class Cloud, class Scalar> void align(Cloud& /*output*/, const Scalar* guess_colmajor) { align(guess_colmajor); }
    // `ndt_.align(*output, ndt_pose_)` with an Eigen::Matrix4f (process.cc:94): any matrix type with data() over 16
    // column-major floats or doubles, taken by reference or by value
    template <class Cloud, class M, class = std::enable_if_t<detail::is_matrix_v<M>>>
    void align(Cloud& /*output*/, const M& guess) { align(guess.data()); }
    template <class M, class = std::enable_if_t<detail::is_matrix_v<M>>>
    void align(const M& guess) { align(guess.data()); }

    FinalTransformation getFinalTransformation() const { FinalTransformation m; for (int i = 0; i < 16; ++i) m[i] = res_.T[i]; return m; }
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
        if (ndt_handle* h = handle("setInputSourceRaw"))
            note("setInputSourceRaw", ndt_set_source_downsampled(h, c.data(), c.size(), sizeof(PointSource), leaf, T_extrinsic_colmajor));
    }
    void addSourceToTarget(const double* T_colmajor) {
        if (ndt_handle* h = handle("addSourceToTarget")) note("addSourceToTarget", ndt_add_target_from_source(h, T_colmajor));
    }
    // Voxel-map serialisation (ndt_map_save / ndt_map_load): e.g. to resume a session without the keyframe clouds
    std::vector<unsigned char> saveMap() {
        size_t need = 0;
        check(ndt_map_save(handle_or_throw(), nullptr, 0, &need));
        std::vector<unsigned char> blob(need);
        check(ndt_map_save(handle_or_throw(), blob.data(), blob.size(), &need));
        return blob;
    }
    void loadMap(const std::vector<unsigned char>& blob) {
        if (ndt_handle* h = handle("loadMap")) note("loadMap", ndt_map_load(h, blob.data(), blob.size()));
    }
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
        check(ndt_align_batch(handle_or_throw(), pts.data(), sizeof(PointSource), off.data(), scans.size(), pair_scan.empty() ? nullptr : pair_scan.data(),
                              guesses.size(), g.data(), out.data()));
        return out;
    }
    const ndt_result& result() const { return res_; }
    int status() const { return status_; }
    std::string lastError() const { return err_; }

  private:
    // the engine is created on first use (the reference constructs its member before any parameter is known)
    ndt_handle* handle(const char* where) {
        if (!h_) {
            h_ = ndt_create(&cfg_);
            if (!h_) { fail(where, NDT_ERR_HIP, "ndt_create failed: no MI355X visible or bad configuration"); return nullptr; }
        }
        return h_;
    }
    ndt_handle* handle_or_throw() {
        if (!handle("ndt_mi355")) throw Error(status_, err_);
        return h_;
    }
    void reconfigure() { if (h_) note("set*", ndt_configure(h_, &cfg_)); }
    void note(const char* where, int rc) {
        status_ = rc;
        if (rc != NDT_OK) fail(where, rc, ndt_last_error(h_));
    }
    void fail(const char* where, int rc, const std::string& msg) {
        status_ = rc;
        err_ = std::string(where) + ": " + msg;
        res_.converged = 0;
        std::fprintf(stderr, "[ndt_mi355] %s (status %d)\n", err_.c_str(), rc);   // PCL_ERROR's role
        if (throw_) throw Error(rc, err_);
    }
    void check(int rc) { if (rc != NDT_OK) { status_ = rc; err_ = ndt_last_error(h_); throw Error(rc, err_); } }

    ndt_config cfg_{};
    ndt_pcl_params pcl_{};
    ndt_pcl_result pres_{};
    bool pcl_mode_ = false;
    ndt_handle* h_ = nullptr;
    ndt_result res_{};
    int status_ = NDT_OK;
    std::string err_;
    bool throw_ = false;
};

}  // namespace ndt_mi355
