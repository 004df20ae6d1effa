// obvi_types.h -- host-side mirror of the reference's basic types and factor structs for the
// optimisation path.  Same names, fields and meaning as the reference (cited per item) so code
// written against ObVi-SLAM's optimiser reads the same; Eigen is replaced by std::array because
// the values only travel into the flat arrays of the C ABI (include/obvi_ba.h).
#ifndef OBVI_HOST_TYPES_H_
#define OBVI_HOST_TYPES_H_

#include <array>
#include <cmath>
#include <cstdint>
#include <memory>
#include <vector>

namespace vslam_types_refactor {

// include/refactoring/types/vslam_basic_types_refactor.h
typedef uint64_t FrameId;
typedef uint64_t FeatureId;
typedef uint64_t CameraId;
typedef uint64_t ObjectId;   // vslam_obj_opt_types_refactor.h:23
typedef uint64_t FeatureFactorId;
typedef uint8_t FactorType;

// low_level_feature_pose_graph.h:18-23, object_pose_graph.h:18-20
static const FactorType kReprojectionErrorFactorTypeId = 0;
static const FactorType kPairwiseErrorFactorTypeId = 1;
static const FactorType kObjectObservationFactorTypeId = 2;
static const FactorType kShapeDimPriorFactorTypeId = 3;
static const FactorType kLongTermMapFactorTypeId = 4;
static const FactorType kPairwiseRobotPoseFactorTypeId = 5;

static const int kEllipsoidPoseParameterizationSize = 4;   // CONSTRAIN_ELLIPSOID_ORIENTATION (CMakeLists.txt:8-15)
static const int kEllipsoidParamterizationSize = 7;        // (sic) vslam_obj_opt_types_refactor.h:20

typedef std::array<double, 6> RawPose3d;      // [t(3), axis-angle(3)]  vslam_types_conversion.h:13-21
typedef std::array<double, 3> Position3d;
typedef std::array<double, 7> RawEllipsoid;   // [x y z yaw dx dy dz]
typedef std::array<double, 2> PixelCoord;
typedef std::array<double, 4> BbCorners;      // (min_x, max_x, min_y, max_y)  vslam_obj_opt_types_refactor.h:184-191
typedef std::array<double, 3> ObjectDim;
template <int N> using Covariance = std::array<double, N * N>;   // row-major
typedef std::shared_ptr<RawPose3d> RawPose3dPtr;
typedef std::shared_ptr<Position3d> Position3dPtr;
typedef std::shared_ptr<RawEllipsoid> RawEllipsoidPtr;

struct CameraIntrinsicsMat { double fx, fy, cx, cy; };
// Pose3D: translation + orientation; the orientation is kept as the axis-angle vector angle*axis
struct Pose3D {
  Position3d transl_{{0, 0, 0}};
  std::array<double, 3> orientation_{{0, 0, 0}};
};
typedef Pose3D CameraExtrinsics;   // camera pose in the robot frame (reprojection_cost_functor.h:155-158)

// ---- small SE(3) helpers (vslam_types_math_util.h:15-93) ----------------------------------
typedef std::array<double, 9> Mat3;
inline Mat3 rotationFromAxisAngle(const std::array<double, 3>& a) {
  const double th = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  if (!(th > 0.0)) return Mat3{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
  const double x = a[0] / th, y = a[1] / th, z = a[2] / th, s = std::sin(th), c = std::cos(th), oc = 1.0 - c;
  return Mat3{{oc * x * x + c, oc * x * y - s * z, oc * x * z + s * y, oc * x * y + s * z, oc * y * y + c, oc * y * z - s * x,
               oc * x * z - s * y, oc * y * z + s * x, oc * z * z + c}};
}
inline std::array<double, 3> axisAngleFromRotation(const Mat3& R) {
  // Eigen::AngleAxis(Quaternion(R)): quaternion branches + angle = 2 atan2(|v|, |w|)
  double q[4];
  double t = R[0] + R[4] + R[8];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    q[i] = 0.5 * t; t = 0.5 / t;
    q[3] = (R[3 * k + j] - R[3 * j + k]) * t; q[j] = (R[3 * j + i] + R[3 * i + j]) * t; q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
  }
  double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  if (n == 0.0) return {{0, 0, 0}};
  const double angle = 2.0 * std::atan2(n, std::fabs(q[3]));
  if (q[3] < 0.0) n = -n;
  return {{angle * q[0] / n, angle * q[1] / n, angle * q[2] / n}};
}
inline Pose3D convertToPose3D(const RawPose3d& r) { Pose3D p; p.transl_ = {{r[0], r[1], r[2]}}; p.orientation_ = {{r[3], r[4], r[5]}}; return p; }
inline RawPose3d convertPoseToArray(const Pose3D& p) { return RawPose3d{{p.transl_[0], p.transl_[1], p.transl_[2], p.orientation_[0], p.orientation_[1], p.orientation_[2]}}; }
// pose_2 expressed in the frame of pose_1  (vslam_types_math_util.h:29-37)
inline Pose3D getPose2RelativeToPose1(const Pose3D& p1, const Pose3D& p2) {
  const Mat3 R1 = rotationFromAxisAngle(p1.orientation_), R2 = rotationFromAxisAngle(p2.orientation_);
  Pose3D out;
  Mat3 R;
  for (int i = 0; i < 3; ++i) {
    out.transl_[i] = R1[i] * (p2.transl_[0] - p1.transl_[0]) + R1[3 + i] * (p2.transl_[1] - p1.transl_[1]) + R1[6 + i] * (p2.transl_[2] - p1.transl_[2]);
    for (int j = 0; j < 3; ++j) R[3 * i + j] = R1[i] * R2[j] + R1[3 + i] * R2[3 + j] + R1[6 + i] * R2[6 + j];
  }
  out.orientation_ = axisAngleFromRotation(R);
  return out;
}
inline Pose3D combinePoses(const Pose3D& p1, const Pose3D& p2_rel_to_1) {   // vslam_types_math_util.h:54-62
  const Mat3 R1 = rotationFromAxisAngle(p1.orientation_), R2 = rotationFromAxisAngle(p2_rel_to_1.orientation_);
  Pose3D out;
  Mat3 R;
  for (int i = 0; i < 3; ++i) {
    out.transl_[i] = p1.transl_[i] + R1[3 * i] * p2_rel_to_1.transl_[0] + R1[3 * i + 1] * p2_rel_to_1.transl_[1] + R1[3 * i + 2] * p2_rel_to_1.transl_[2];
    for (int j = 0; j < 3; ++j) R[3 * i + j] = R1[3 * i] * R2[j] + R1[3 * i + 1] * R2[3 + j] + R1[3 * i + 2] * R2[6 + j];
  }
  out.orientation_ = axisAngleFromRotation(R);
  return out;
}
inline Position3d getPositionRelativeToPose(const Pose3D& pose, const Position3d& p) {   // :39-44
  const Mat3 R = rotationFromAxisAngle(pose.orientation_);
  Position3d o;
  for (int i = 0; i < 3; ++i) o[i] = R[i] * (p[0] - pose.transl_[0]) + R[3 + i] * (p[1] - pose.transl_[1]) + R[6 + i] * (p[2] - pose.transl_[2]);
  return o;
}
inline Position3d combinePoseAndPosition(const Pose3D& pose, const Position3d& p) {      // :46-52
  const Mat3 R = rotationFromAxisAngle(pose.orientation_);
  Position3d o;
  for (int i = 0; i < 3; ++i) o[i] = pose.transl_[i] + R[3 * i] * p[0] + R[3 * i + 1] * p[1] + R[3 * i + 2] * p[2];
  return o;
}

// ---- factor structs -----------------------------------------------------------------------
struct ReprojectionErrorFactor {   // low_level_feature_pose_graph.h:90-124
  FrameId frame_id_;
  FeatureId feature_id_;
  CameraId camera_id_;
  PixelCoord feature_pos_;
  double reprojection_error_std_dev_;
  FactorType getFactorType() const { return kReprojectionErrorFactorTypeId; }
};
struct RelPoseFactor {             // low_level_feature_pose_graph.h:126-160
  FrameId frame_id_1_;
  FrameId frame_id_2_;
  Pose3D measured_pose_deviation_;
  Covariance<6> pose_deviation_cov_;
  FactorType getFactorType() const { return kPairwiseRobotPoseFactorTypeId; }
};
struct ObjectObservationFactor {   // object_pose_graph.h:88-126
  FrameId frame_id_;
  CameraId camera_id_;
  ObjectId object_id_;
  BbCorners bounding_box_corners_;
  Covariance<4> bounding_box_corners_covariance_;
  double detection_confidence_ = 1.0;
};
struct ShapeDimPriorFactor {       // object_pose_graph.h:128-148
  ObjectId object_id_;
  ObjectDim mean_shape_dim_;
  Covariance<3> shape_dim_cov_;
};
// LTM prior data consumed by IndependentObjectMapFactor (long_term_map_factor_creator.h:265-322)
struct LongTermMapObjectPrior {
  ObjectId object_id_;
  RawEllipsoid ellipsoid_mean_;
  Covariance<7> covariance_;
};

// generateOdomCov: relative_pose_factor_utils.h:17-36 (diagonal, sigma floor 1e-3)
inline Covariance<6> generateOdomCov(const Pose3D& rel, double transl_error_mult_for_transl_error, double transl_error_mult_for_rot_error,
                                     double rot_error_mult_for_transl_error, double rot_error_mult_for_rot_error) {
  const double ang = std::sqrt(rel.orientation_[0] * rel.orientation_[0] + rel.orientation_[1] * rel.orientation_[1] + rel.orientation_[2] * rel.orientation_[2]);
  const double tn = std::sqrt(rel.transl_[0] * rel.transl_[0] + rel.transl_[1] * rel.transl_[1] + rel.transl_[2] * rel.transl_[2]);
  Covariance<6> cov{};
  for (int i = 0; i < 3; ++i) {
    const double sd_t = std::fabs(rel.transl_[i]) * transl_error_mult_for_transl_error + std::fabs(ang) * rot_error_mult_for_transl_error;
    const double sd_r = std::fabs(rel.orientation_[i]) * rot_error_mult_for_rot_error + tn * transl_error_mult_for_rot_error;
    cov[7 * i] = std::max(1e-6, sd_t * sd_t);
    cov[7 * (3 + i)] = std::max(1e-6, sd_r * sd_r);
  }
  return cov;
}

}  // namespace vslam_types_refactor
#endif  // OBVI_HOST_TYPES_H_
