// frontend_math.h -- arithmetic of the visual-feature front-end gating (include/obvi_frontend.h), host + device.
// Follows visual_feature_front_end.h:52-132 (epipolar error vector) and :726-800 (parallax test) of the reference.
#ifndef OBVI_FRONTEND_MATH_H_
#define OBVI_FRONTEND_MATH_H_

#include <math.h>

#include "ba_math.h"

namespace obvi {

struct Rigid { double R[9]; double t[3]; };   // x -> R x + t, R row-major

// convertToAffine(Pose3D): Translation * AngleAxis (Rodrigues, as Eigen::AngleAxis::toRotationMatrix)
OBVI_HD void rigid_from_pose(const double* p, Rigid* T) {
  const double th = sqrt(p[3] * p[3] + p[4] * p[4] + p[5] * p[5]);
  if (th > 0.0) {
    const double x = p[3] / th, y = p[4] / th, z = p[5] / th, s = sin(th), c = cos(th), oc = 1.0 - c;
    T->R[0] = oc * x * x + c;     T->R[1] = oc * x * y - s * z; T->R[2] = oc * x * z + s * y;
    T->R[3] = oc * x * y + s * z; T->R[4] = oc * y * y + c;     T->R[5] = oc * y * z - s * x;
    T->R[6] = oc * x * z - s * y; T->R[7] = oc * y * z + s * x; T->R[8] = oc * z * z + c;
  } else {
    for (int k = 0; k < 9; ++k) T->R[k] = (k % 4 == 0) ? 1.0 : 0.0;
  }
  T->t[0] = p[0]; T->t[1] = p[1]; T->t[2] = p[2];
}
// camera -> robot from the DevCam (which stores the inverse: robot -> camera)
OBVI_HD void rigid_cam_to_robot(const DevCam& c, Rigid* T) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T->R[3 * i + j] = c.Rinv[3 * j + i];
  for (int i = 0; i < 3; ++i) T->t[i] = -(T->R[3 * i] * c.tinv[0] + T->R[3 * i + 1] * c.tinv[1] + T->R[3 * i + 2] * c.tinv[2]);
}
OBVI_HD void rigid_mul(const Rigid& A, const Rigid& B, Rigid* C) {   // C = A o B
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) C->R[3 * i + j] = A.R[3 * i] * B.R[j] + A.R[3 * i + 1] * B.R[3 + j] + A.R[3 * i + 2] * B.R[6 + j];
    C->t[i] = A.R[3 * i] * B.t[0] + A.R[3 * i + 1] * B.t[1] + A.R[3 * i + 2] * B.t[2] + A.t[i];
  }
}
OBVI_HD void rigid_inv(const Rigid& A, Rigid* B) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B->R[3 * i + j] = A.R[3 * j + i];
  for (int i = 0; i < 3; ++i) B->t[i] = -(B->R[3 * i] * A.t[0] + B->R[3 * i + 1] * A.t[1] + B->R[3 * i + 2] * A.t[2]);
}

// getNormalizedEpipolarErrorVec (:52-132): view 1 = the reference observation, view 2 = the candidate.
//   cam1_to_cam2 = (world<-robot2 * robot<-cam2)^-1 * world<-robot1 * robot<-cam1
//   epipole = K2 * cam1_to_cam2 * 0;  x1_in2 = K2 * cam1_to_cam2 * K1^-1 * (pixel1, 1);  both de-homogenised
//   u = normalised(x1_in2 - epipole);  result = epipole + ((pixel2 - epipole) . u) u - pixel2
OBVI_HD void epipolar_error_vec(const DevCam& cam1, const DevCam& cam2, const double* pose1, const double* pose2, const double* pixel1,
                                       const double* pixel2, double* err) {
  Rigid w_r1, w_r2, r_c1, r_c2, w_c1, w_c2, c2_w, c1_c2;
  rigid_from_pose(pose1, &w_r1); rigid_from_pose(pose2, &w_r2);
  rigid_cam_to_robot(cam1, &r_c1); rigid_cam_to_robot(cam2, &r_c2);
  rigid_mul(w_r1, r_c1, &w_c1); rigid_mul(w_r2, r_c2, &w_c2);
  rigid_inv(w_c2, &c2_w);
  rigid_mul(c2_w, w_c1, &c1_c2);
  // K2 * t
  const double ex = cam2.fx * c1_c2.t[0] + cam2.cx * c1_c2.t[2], ey = cam2.fy * c1_c2.t[1] + cam2.cy * c1_c2.t[2], ez = c1_c2.t[2];
  const double epx = ex / ez, epy = ey / ez;
  const double n[3] = {(pixel1[0] - cam1.cx) / cam1.fx, (pixel1[1] - cam1.cy) / cam1.fy, 1.0};
  double q[3];
  for (int i = 0; i < 3; ++i) q[i] = c1_c2.R[3 * i] * n[0] + c1_c2.R[3 * i + 1] * n[1] + c1_c2.R[3 * i + 2] * n[2] + c1_c2.t[i];
  const double hx = cam2.fx * q[0] + cam2.cx * q[2], hy = cam2.fy * q[1] + cam2.cy * q[2];
  const double x1x = hx / q[2], x1y = hy / q[2];
  double ux = x1x - epx, uy = x1y - epy;
  const double nn = ux * ux + uy * uy;
  if (nn > 0.0) { const double inv = 1.0 / sqrt(nn); ux *= inv; uy *= inv; }   // Eigen's normalized(): unchanged when the norm is zero
  const double d = (pixel2[0] - epx) * ux + (pixel2[1] - epy) * uy;
  err[0] = epx + d * ux - pixel2[0];
  err[1] = epy + d * uy - pixel2[1];
}

// relative pose of pose 2 in the frame of pose 1 (getPose2RelativeToPose1, vslam_types_math_util.h:29-36): norm of the translation
// and angle of the rotation as Eigen::AngleAxis(Matrix3) reports it (always in [0, pi]: via the quaternion, angle = 2 atan2(|vec|, |w|))
OBVI_HD void relative_motion(const double* pose1, const double* pose2, double* transl_norm, double* angle) {
  Rigid a, b, ai, rel;
  rigid_from_pose(pose1, &a); rigid_from_pose(pose2, &b);
  rigid_inv(a, &ai);
  rigid_mul(ai, b, &rel);
  *transl_norm = sqrt(rel.t[0] * rel.t[0] + rel.t[1] * rel.t[1] + rel.t[2] * rel.t[2]);
  // rotation matrix -> quaternion (Eigen's branches) -> angle
  const double* R = rel.R;
  double w, x, y, z;
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0.0) {
    double t = sqrt(tr + 1.0); w = 0.5 * t; t = 0.5 / t;
    x = (R[7] - R[5]) * t; y = (R[2] - R[6]) * t; z = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    double v[3];
    v[i] = 0.5 * t; t = 0.5 / t;
    w = (R[3 * k + j] - R[3 * j + k]) * t;
    v[j] = (R[3 * j + i] + R[3 * i + j]) * t;
    v[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    x = v[0]; y = v[1]; z = v[2];
  }
  const double vn = sqrt(x * x + y * y + z * z);
  *angle = 2.0 * atan2(vn, fabs(w));
}

}  // namespace obvi
#endif  // OBVI_FRONTEND_MATH_H_
