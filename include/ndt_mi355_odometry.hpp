// ndt_mi355_odometry.hpp — the LiDAR half of lio-slam's per-frame routine, Process::alignByPC
// (src/slam/src/process.cc:60-153), on the MI355X engine, without ROS / PCL / Ceres:
//   first frame  : filtered scan -> keyframe map                                   (process.cc:75-85)
//   every frame  : extrinsic transform + VoxelGrid 0.5 m -> source                 (process.cc:62-73, 87)
//                  align from the predicted pose                                   (process.cc:90-95)
//                  keyframe gate |dp| > 0.3 m or |dtheta| > 5 deg, merge the scan   (process.cc:128-150)
// The predicted pose is the caller's (the reference takes it from IMU pre-integration, process.cc:90, which stays
// on the host unchanged); predictConstantVelocity() is a stand-in for streams without an IMU.  The graph
// optimisation between align and the keyframe gate (process.cc:97) also stays with the caller: pass the optimised
// pose to endFrame() (or the NDT pose itself).  Clouds never return to the host.
#pragma once
#include <cmath>

#include "ndt_mi355.hpp"

namespace ndt_mi355 {

template <typename Point>
class ScanToMapOdometry {
  public:
    using Matrix4d = std::array<double, 16>;   // column-major
    struct Options {
        float scan_leaf = 0.5f;        // voxel.setLeafSize(0.5, 0.5, 0.5)            process.cc:69
        double p_thresh = 0.3;         // keyframe translation gate, metres            process.h:110
        double q_thresh_deg = 5.0;     // keyframe rotation gate, degrees              process.h:111
        Matrix4d T_imu_lidar = identity();   // TIL_ / QIL_                            process.cc:63-65
    };

    explicit ScanToMapOdometry(int device = 0) : ndt_(device) {
        ndt_.setResolution(1.0f);              // process.cc:373-376
        ndt_.setMaximumIterations(35);
        ndt_.setTransformationEpsilon(0.01);
        ndt_.setStepSize(0.1);
        last_ = last_key_ = prev_ = identity();
    }
    Options& options() { return opt_; }
    NormalDistributionsTransform<Point>& ndt() { return ndt_; }

    // Returns true when the frame was aligned (false for the first frame, which only starts the map).
    template <class Cloud> bool beginFrame(const Cloud& raw_scan, const Matrix4d& predicted_pose) {
        ndt_.setInputSourceRaw(raw_scan, opt_.scan_leaf, opt_.T_imu_lidar.data());
        if (first_) {                                      // process.cc:75-85
            ndt_.addSourceToTarget(predicted_pose.data());
            prev_ = last_ = last_key_ = predicted_pose;
            first_ = false;
            ++n_key_;
            return false;
        }
        ndt_.align(predicted_pose.data());                 // process.cc:94
        return true;
    }
    Matrix4d ndtPose() const { return ndt_.getFinalTransformation(); }     // process.cc:95

    // Keyframe gate on the (optimised) pose, process.cc:128-150.  Returns true when the scan joined the map.
    bool endFrame(const Matrix4d& pose) {
        prev_ = last_;
        last_ = pose;
        const double dp = std::sqrt(sq(pose[12] - last_key_[12]) + sq(pose[13] - last_key_[13]) + sq(pose[14] - last_key_[14]));
        // rotation angle of R_key^T R  (|Q2theta(q_key^-1 q)|, process.cc:130)
        double tr = 0.0;
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 3; ++r) tr += last_key_[c * 4 + r] * pose[c * 4 + r];
        const double cs = std::fmin(1.0, std::fmax(-1.0, 0.5 * (tr - 1.0)));
        const double dth = std::acos(cs);
        if (dp > opt_.p_thresh || dth > opt_.q_thresh_deg * 3.14 / 180.0) {     // 3.14 as in process.cc:132
            ndt_.addSourceToTarget(pose.data());
            last_key_ = pose;
            ++n_key_;
            return true;
        }
        return false;
    }

    // pose_k+1 = pose_k * (pose_k-1^-1 * pose_k): stands in for IntegrationBase::predict where there is no IMU
    Matrix4d predictConstantVelocity() const { return mul(last_, mul(inv(prev_), last_)); }
    int keyframes() const { return n_key_; }

    static Matrix4d identity() { Matrix4d m{}; m[0] = m[5] = m[10] = m[15] = 1.0; return m; }
    static Matrix4d mul(const Matrix4d& a, const Matrix4d& b) {
        Matrix4d m{};
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) { double s = 0; for (int k = 0; k < 4; ++k) s += a[k * 4 + r] * b[c * 4 + k]; m[c * 4 + r] = s; }
        return m;
    }
    static Matrix4d inv(const Matrix4d& a) {   // rigid transform
        Matrix4d m = identity();
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 3; ++r) m[c * 4 + r] = a[r * 4 + c];
        for (int r = 0; r < 3; ++r) m[12 + r] = -(m[r] * a[12] + m[4 + r] * a[13] + m[8 + r] * a[14]);
        return m;
    }

  private:
    static double sq(double v) { return v * v; }
    NormalDistributionsTransform<Point> ndt_;
    Options opt_;
    Matrix4d last_, last_key_, prev_;
    bool first_ = true;
    int n_key_ = 0;
};

}  // namespace ndt_mi355
