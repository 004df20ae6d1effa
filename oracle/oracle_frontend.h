// oracle_frontend.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// CPU restatement of the visual-feature front-end gating arithmetic of the reference
// (/root/reference/include/refactoring/visual_feature_frontend/visual_feature_front_end.h):
//   getNormalizedEpipolarErrorVec        :52-132
//   isReprojectionErrorFactorInlier      :511-602
//   checkMinParallaxRequirements_        :726-800
// written with 4x4 homogeneous transforms the way the reference composes Eigen::Affine3d objects.  Nothing under
// obvi-slam_amd/ may include this file.
#ifndef OBVI_ORACLE_FRONTEND_H_
#define OBVI_ORACLE_FRONTEND_H_

#include <cmath>
#include <cstdint>

namespace oracle {

struct Affine { double m[4][4]; };
inline Affine affine_identity() { Affine a{}; for (int i = 0; i < 4; ++i) a.m[i][i] = 1.0; return a; }
inline Affine affine_mul(const Affine& a, const Affine& b) {
  Affine c{};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double s = 0.0; for (int k = 0; k < 4; ++k) s += a.m[i][k] * b.m[k][j]; c.m[i][j] = s; }
  return c;
}
inline Affine affine_inverse(const Affine& a) {   // rigid: [R t]^-1 = [R^T  -R^T t]
  Affine c = affine_identity();
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c.m[i][j] = a.m[j][i];
  for (int i = 0; i < 3; ++i) c.m[i][3] = -(c.m[i][0] * a.m[0][3] + c.m[i][1] * a.m[1][3] + c.m[i][2] * a.m[2][3]);
  return c;
}
// Eigen::Translation3d(t) * Eigen::AngleAxisd(angle, axis)   (convertToAffine, vslam_types_math_util.h:15-19)
inline Affine affine_from_translation_angle_axis(const double* t, double angle, const double* axis) {
  Affine a = affine_identity();
  const double s = std::sin(angle), c = std::cos(angle), oc = 1.0 - c, x = axis[0], y = axis[1], z = axis[2];
  a.m[0][0] = oc * x * x + c;     a.m[0][1] = oc * x * y - s * z; a.m[0][2] = oc * x * z + s * y;
  a.m[1][0] = oc * x * y + s * z; a.m[1][1] = oc * y * y + c;     a.m[1][2] = oc * y * z - s * x;
  a.m[2][0] = oc * x * z - s * y; a.m[2][1] = oc * y * z + s * x; a.m[2][2] = oc * z * z + c;
  a.m[0][3] = t[0]; a.m[1][3] = t[1]; a.m[2][3] = t[2];
  return a;
}
inline Affine affine_from_pose6(const double* p) {   // (t, axis-angle vector)
  const double th = std::sqrt(p[3] * p[3] + p[4] * p[4] + p[5] * p[5]);
  const double ex[3] = {1.0, 0.0, 0.0};
  if (!(th > 0.0)) return affine_from_translation_angle_axis(p, 0.0, ex);
  const double ax[3] = {p[3] / th, p[4] / th, p[5] / th};
  return affine_from_translation_angle_axis(p, th, ax);
}
inline Affine affine_from_quat_translation(const double* q_xyzw, const double* t) {   // camera extrinsics as the C ABI takes them
  const double n = std::sqrt(q_xyzw[0] * q_xyzw[0] + q_xyzw[1] * q_xyzw[1] + q_xyzw[2] * q_xyzw[2] + q_xyzw[3] * q_xyzw[3]);
  const double x = q_xyzw[0] / n, y = q_xyzw[1] / n, z = q_xyzw[2] / n, w = q_xyzw[3] / n;
  Affine a = affine_identity();
  a.m[0][0] = 1 - 2 * (y * y + z * z); a.m[0][1] = 2 * (x * y - z * w);     a.m[0][2] = 2 * (x * z + y * w);
  a.m[1][0] = 2 * (x * y + z * w);     a.m[1][1] = 1 - 2 * (x * x + z * z); a.m[1][2] = 2 * (y * z - x * w);
  a.m[2][0] = 2 * (x * z - y * w);     a.m[2][1] = 2 * (y * z + x * w);     a.m[2][2] = 1 - 2 * (x * x + y * y);
  a.m[0][3] = t[0]; a.m[1][3] = t[1]; a.m[2][3] = t[2];
  return a;
}

// :110-131
inline void epipolar_error_vec(const double* K1 /*fx fy cx cy*/, const double* K2, const Affine& cam_to_robot_1, const Affine& cam_to_robot_2, const double* pixel1,
                               const double* pixel2, const Affine& robot_to_world_1, const Affine& robot_to_world_2, double* out) {
  const Affine cam1_to_cam2 = affine_mul(affine_mul(affine_inverse(affine_mul(robot_to_world_2, cam_to_robot_2)), robot_to_world_1), cam_to_robot_1);
  // intrinsics2 * cam1_to_cam2 * Zero
  const double c[3] = {cam1_to_cam2.m[0][3], cam1_to_cam2.m[1][3], cam1_to_cam2.m[2][3]};
  const double he[3] = {K2[0] * c[0] + K2[2] * c[2], K2[1] * c[1] + K2[3] * c[2], c[2]};
  const double epipole[2] = {he[0] / he[2], he[1] / he[2]};
  // intrinsics2 * cam1_to_cam2 * intrinsics1^-1 * (pixel1, 1)
  const double n1[3] = {(pixel1[0] - K1[2]) / K1[0], (pixel1[1] - K1[3]) / K1[1], 1.0};
  double p[3];
  for (int i = 0; i < 3; ++i) p[i] = cam1_to_cam2.m[i][0] * n1[0] + cam1_to_cam2.m[i][1] * n1[1] + cam1_to_cam2.m[i][2] * n1[2] + cam1_to_cam2.m[i][3];
  const double hx[3] = {K2[0] * p[0] + K2[2] * p[2], K2[1] * p[1] + K2[3] * p[2], p[2]};
  const double x1_in2[2] = {hx[0] / hx[2], hx[1] / hx[2]};
  double u[2] = {x1_in2[0] - epipole[0], x1_in2[1] - epipole[1]};
  const double nn = u[0] * u[0] + u[1] * u[1];
  if (nn > 0.0) { const double n = std::sqrt(nn); u[0] /= n; u[1] /= n; }   // Eigen normalized()
  const double d = (pixel2[0] - epipole[0]) * u[0] + (pixel2[1] - epipole[1]) * u[1];
  out[0] = epipole[0] + d * u[0] - pixel2[0];
  out[1] = epipole[1] + d * u[1] - pixel2[1];
}

// relative_pose.transl_.norm() and relative_pose.orientation_.angle() of getPose2RelativeToPose1 (:757-765)
inline void relative_motion(const double* pose1, const double* pose2, double* transl_norm, double* angle) {
  const Affine rel = affine_mul(affine_inverse(affine_from_pose6(pose1)), affine_from_pose6(pose2));
  *transl_norm = std::sqrt(rel.m[0][3] * rel.m[0][3] + rel.m[1][3] * rel.m[1][3] + rel.m[2][3] * rel.m[2][3]);
  // Eigen::AngleAxisd(rotation matrix): through the quaternion; angle = 2 atan2(|vec|, |w|)
  const double R[3][3] = {{rel.m[0][0], rel.m[0][1], rel.m[0][2]}, {rel.m[1][0], rel.m[1][1], rel.m[1][2]}, {rel.m[2][0], rel.m[2][1], rel.m[2][2]}};
  double q[4];   // x y z w
  const double tr = R[0][0] + R[1][1] + R[2][2];
  if (tr > 0.0) {
    double t = std::sqrt(tr + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
    q[0] = (R[2][1] - R[1][2]) * t; q[1] = (R[0][2] - R[2][0]) * t; q[2] = (R[1][0] - R[0][1]) * t;
  } else {
    int i = 0;
    if (R[1][1] > R[0][0]) i = 1;
    if (R[2][2] > R[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
    q[i] = 0.5 * t; t = 0.5 / t;
    q[3] = (R[k][j] - R[j][k]) * t; q[j] = (R[j][i] + R[i][j]) * t; q[k] = (R[k][i] + R[i][k]) * t;
  }
  *angle = 2.0 * std::atan2(std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]), std::fabs(q[3]));
}

}  // namespace oracle
#endif  // OBVI_ORACLE_FRONTEND_H_
