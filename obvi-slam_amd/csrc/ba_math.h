// ba_math.h -- per-factor arithmetic of the bundle-adjustment kernels (gfx950 device code; the
// same inline functions also compile for the host so the CPU test-suite can check them without
// a GPU).  Every function cites the reference lines (under /root/reference) it reproduces.
//
//  * reprojection (the hot factor, N_r ~ 3e6): closed-form residual + Jacobian, with the
//    pose-only part (R^T, -R^T t, right Jacobian of SO(3)) hoisted into a per-pose cache so the
//    per-observation work is ~150 FMAs and no transcendental.
//  * bounding box / shape prior / LTM prior / relative pose (N <= ~3e4): forward-mode duals
//    through the same operation sequence as the reference functors, which is how the reference
//    itself differentiates them (ceres::AutoDiffCostFunction).
#ifndef OBVI_BA_MATH_H_
#define OBVI_BA_MATH_H_

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define OBVI_HD __host__ __device__ __forceinline__
#else
#define OBVI_HD inline
#endif

namespace obvi {

// vslam_math_util.h:17
#define OBVI_SMALL_ANGLE 1e-8
// ellipsoid_utils.h:22 declares the constant as `float`; T(kDim...) widens that float.
#define OBVI_DIM_REG ((double)1e-3f)

struct DevCam {        // inverse extrinsics + intrinsics, one per camera
  double Rinv[9];      // R_e^T           (reprojection_cost_functor.cpp:10-13)
  double tinv[3];      // -R_e^T t_e
  double fx, fy, cx, cy;
  double depth_min;    // -inf: no depth clamp (the production functor a3); 1e-15: z <- max(z, depth_min) with the gated derivative of the
                       // analytic-Jacobian functor a2 (reprojection_cost_functor_analytic_jacobian.h:160, 289-292, 591)
};
// reprojection_cost_functor_analytic_jacobian.h:591
#define OBVI_ANALYTIC_EPSILON 1e-15

struct PoseCache {     // per robot pose, recomputed whenever poses change
  double Rinv[9];      // R(aa)^T
  double tinv[3];      // -R^T t
  double Jr[9];        // right Jacobian of SO(3) at aa; all-zero in the small-angle branch
  double pad[3];
};

// ---------------------------------------------------------------------------------------
// Pose cache.  vslam_math_util.h:357-375: angle = |aa|; angle > 1e-8 -> AngleAxis(-angle,
// aa/angle), else AngleAxis(0, e_x) -- a constant, so autodiff sees a zero derivative w.r.t.
// aa there; Jr = 0 reproduces that.
// d(R^T v)/d(aa) = [R^T v]x Jr(aa),  Jr = I - (1-cos t)/t^2 [aa]x + (t - sin t)/t^3 [aa]x^2.
//
// analytic = true (the analytic-Jacobian functor a2, reprojection_cost_functor_analytic_jacobian.h:63-70): the rotation is a smooth
// function of aa through aa = 0 -- symforce builds it from the quaternion (aa sin(th/2)/th, cos(th/2)) with th = sqrt(|aa|^2 + 1e-15),
// i.e. R = I + (sin th/th) [aa]x + ((1-cos th)/th^2) [aa]x^2, which differs from the exponential map by less than 1e-15 absolute in R
// and in dR/daa at every aa -- so below the threshold the series of the exponential map and of its right Jacobian take the place
// of the production functor's constant branch.
// ---------------------------------------------------------------------------------------
OBVI_HD void make_pose_cache(const double* pose, PoseCache* pc, bool analytic = false) {
  const double ax = pose[3], ay = pose[4], az = pose[5];
  const double t2 = ax * ax + ay * ay + az * az;
  const double t = sqrt(t2);
  if (analytic && !(t > OBVI_SMALL_ANGLE)) {
    // R^T = I - [aa]x + (1/2) [aa]x^2 ;  Jr = I - (1/2) [aa]x + (1/6) [aa]x^2   (next terms are O(t^3) < 1e-24)
    pc->Rinv[0] = 1.0 + 0.5 * (ax * ax - t2); pc->Rinv[1] = az + 0.5 * ax * ay;         pc->Rinv[2] = -ay + 0.5 * ax * az;
    pc->Rinv[3] = -az + 0.5 * ax * ay;        pc->Rinv[4] = 1.0 + 0.5 * (ay * ay - t2); pc->Rinv[5] = ax + 0.5 * ay * az;
    pc->Rinv[6] = ay + 0.5 * ax * az;         pc->Rinv[7] = -ax + 0.5 * ay * az;        pc->Rinv[8] = 1.0 + 0.5 * (az * az - t2);
    const double a = 0.5, b = 1.0 / 6.0;
    pc->Jr[0] = 1.0 + b * (ax * ax - t2); pc->Jr[1] = a * az + b * ax * ay;     pc->Jr[2] = -a * ay + b * ax * az;
    pc->Jr[3] = -a * az + b * ax * ay;    pc->Jr[4] = 1.0 + b * (ay * ay - t2); pc->Jr[5] = a * ax + b * ay * az;
    pc->Jr[6] = a * ay + b * ax * az;     pc->Jr[7] = -a * ax + b * ay * az;    pc->Jr[8] = 1.0 + b * (az * az - t2);
  } else if (t > OBVI_SMALL_ANGLE) {
    const double s = sin(t), c = cos(t);
    const double ux = ax / t, uy = ay / t, uz = az / t;
    const double oc = 1.0 - c;
    // R^T = c I + (1-c) u u^T - s [u]x
    pc->Rinv[0] = oc * ux * ux + c;      pc->Rinv[1] = oc * ux * uy + s * uz; pc->Rinv[2] = oc * ux * uz - s * uy;
    pc->Rinv[3] = oc * ux * uy - s * uz; pc->Rinv[4] = oc * uy * uy + c;      pc->Rinv[5] = oc * uy * uz + s * ux;
    pc->Rinv[6] = oc * ux * uz + s * uy; pc->Rinv[7] = oc * uy * uz - s * ux; pc->Rinv[8] = oc * uz * uz + c;
    // stable coefficients: (1-cos t)/t^2 = 0.5 (sin(t/2)/(t/2))^2 ; (t - sin t)/t^3 by series for small t
    const double sh = sin(0.5 * t) / (0.5 * t);
    const double a = 0.5 * sh * sh;
    double b;
    if (t < 0.25) {
      b = (1.0 / 6.0) - t2 * ((1.0 / 120.0) - t2 * ((1.0 / 5040.0) - t2 * ((1.0 / 362880.0) - t2 * (1.0 / 39916800.0))));
    } else {
      b = (t - s) / (t2 * t);
    }
    // [aa]x^2 = aa aa^T - t^2 I
    pc->Jr[0] = 1.0 + b * (ax * ax - t2); pc->Jr[1] = a * az + b * ax * ay;     pc->Jr[2] = -a * ay + b * ax * az;
    pc->Jr[3] = -a * az + b * ax * ay;    pc->Jr[4] = 1.0 + b * (ay * ay - t2); pc->Jr[5] = a * ax + b * ay * az;
    pc->Jr[6] = a * ay + b * ax * az;     pc->Jr[7] = -a * ax + b * ay * az;    pc->Jr[8] = 1.0 + b * (az * az - t2);
  } else {
    for (int i = 0; i < 9; ++i) { pc->Rinv[i] = (i % 4 == 0) ? 1.0 : 0.0; pc->Jr[i] = 0.0; }
  }
  for (int i = 0; i < 3; ++i)
    pc->tinv[i] = -(pc->Rinv[3 * i] * pose[0] + pc->Rinv[3 * i + 1] * pose[1] + pc->Rinv[3 * i + 2] * pose[2]);
  pc->pad[0] = pc->pad[1] = pc->pad[2] = 0.0;
}

// ---------------------------------------------------------------------------------------
// Reprojection residual (+ Jacobians): ReprojectionCostFunctor::runOperator
// (reprojection_cost_functor.h:56-93), getProjectedPixelLocationRectified
// (vslam_math_util.h:347-394), constants of the ctor (reprojection_cost_functor.cpp:14-17):
//   r = (f/sigma) * (p_cam.xy / p_cam.z - (pixel - c)/f),  p_cam = R_e^T (R^T (X - t) - t_e)
// Jp is 2x6 row-major [d/dt, d/daa], Jl 2x3 row-major.
// ---------------------------------------------------------------------------------------
template <bool JAC>
OBVI_HD void reproj_eval(const PoseCache& pc, const DevCam& cam, const double* X, double px, double py,
                         double sigma, double* r, double* Jp, double* Jl) {
  const double prx = pc.Rinv[0] * X[0] + pc.Rinv[1] * X[1] + pc.Rinv[2] * X[2] + pc.tinv[0];
  const double pry = pc.Rinv[3] * X[0] + pc.Rinv[4] * X[1] + pc.Rinv[5] * X[2] + pc.tinv[1];
  const double prz = pc.Rinv[6] * X[0] + pc.Rinv[7] * X[1] + pc.Rinv[8] * X[2] + pc.tinv[2];
  const double x = cam.Rinv[0] * prx + cam.Rinv[1] * pry + cam.Rinv[2] * prz + cam.tinv[0];
  const double y = cam.Rinv[3] * prx + cam.Rinv[4] * pry + cam.Rinv[5] * prz + cam.tinv[1];
  const double zraw = cam.Rinv[6] * prx + cam.Rinv[7] * pry + cam.Rinv[8] * prz + cam.tinv[2];
  // depth clamp of the analytic-Jacobian functor (depth_min = 1e-15; reprojection_cost_functor_analytic_jacobian.h:160); depth_min = -inf
  // leaves z alone (the production functor has no clamp, vslam_math_util.h:376-394)
  const double z = zraw < cam.depth_min ? cam.depth_min : zraw;
  const double mx = cam.fx / sigma, my = cam.fy / sigma;
  const double u = x / z, v = y / z;
  r[0] = mx * (u - (px - cam.cx) / cam.fx);
  r[1] = my * (v - (py - cam.cy) / cam.fy);
  if (JAC) {
    const double iz = 1.0 / z;
    // derivative of the clamp (:289-292): ((z - eps > 0) - (z - eps < 0) + 1) / 2 = 1 above eps, 1/2 at it, 0 below; 1 without a clamp
    const double dz = zraw - cam.depth_min;
    const double gate = 0.5 * ((double)((dz > 0.0) - (dz < 0.0)) + 1.0);
    // rows of d r / d p_cam
    const double p00 = mx * iz, p02 = -mx * u * iz * gate;
    const double p11 = my * iz, p12 = -my * v * iz * gate;
    // A = Pj * R_e^T  (2x3)
    const double a00 = p00 * cam.Rinv[0] + p02 * cam.Rinv[6], a01 = p00 * cam.Rinv[1] + p02 * cam.Rinv[7], a02 = p00 * cam.Rinv[2] + p02 * cam.Rinv[8];
    const double a10 = p11 * cam.Rinv[3] + p12 * cam.Rinv[6], a11 = p11 * cam.Rinv[4] + p12 * cam.Rinv[7], a12 = p11 * cam.Rinv[5] + p12 * cam.Rinv[8];
    // d/dX = A R^T ; d/dt = -d/dX
    for (int k = 0; k < 3; ++k) {
      const double j0 = a00 * pc.Rinv[k] + a01 * pc.Rinv[3 + k] + a02 * pc.Rinv[6 + k];
      const double j1 = a10 * pc.Rinv[k] + a11 * pc.Rinv[3 + k] + a12 * pc.Rinv[6 + k];
      Jl[k] = j0; Jl[3 + k] = j1; Jp[k] = -j0; Jp[6 + k] = -j1;
    }
    // d/daa = A [p_r]x Jr ;  a^T [p]x = (a x p)^T
    const double b00 = a01 * prz - a02 * pry, b01 = a02 * prx - a00 * prz, b02 = a00 * pry - a01 * prx;
    const double b10 = a11 * prz - a12 * pry, b11 = a12 * prx - a10 * prz, b12 = a10 * pry - a11 * prx;
    for (int k = 0; k < 3; ++k) {
      Jp[3 + k] = b00 * pc.Jr[k] + b01 * pc.Jr[3 + k] + b02 * pc.Jr[6 + k];
      Jp[9 + k] = b10 * pc.Jr[k] + b11 * pc.Jr[3 + k] + b12 * pc.Jr[6 + k];
    }
  }
}

// ceres::HuberLoss [Ceres-doc]: rho(s) and rho'(s) for s = |r|^2.  rho'' <= 0 for Huber, so the
// Ceres corrector reduces to scaling residual and Jacobian by sqrt(rho').
OBVI_HD void huber_eval(double s, double a, double* rho0, double* rho1) {
  const double b = a * a;
  if (s > b) {
    const double rt = sqrt(s);
    *rho0 = 2.0 * a * rt - b;
    const double w = a / rt;
    *rho1 = w > 2.2250738585072014e-308 ? w : 2.2250738585072014e-308;
  } else {
    *rho0 = s; *rho1 = 1.0;
  }
}

// ---------------------------------------------------------------------------------------
// forward-mode dual (the role ceres::Jet<double,N> plays in the reference)
// ---------------------------------------------------------------------------------------
template <int N>
struct Dual {
  double v;
  double d[N];
  OBVI_HD Dual() {}
  OBVI_HD Dual(double c) : v(c) { for (int i = 0; i < N; ++i) d[i] = 0.0; }  // NOLINT
};
template <int N> OBVI_HD Dual<N> dvar(double c, int k) { Dual<N> r(c); r.d[k] = 1.0; return r; }
template <int N> OBVI_HD Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> OBVI_HD Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> OBVI_HD Dual<N> operator-(const Dual<N>& a) { Dual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> OBVI_HD Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.v * b.d[i] + a.d[i] * b.v; return r; }
template <int N> OBVI_HD Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; const double inv = 1.0 / b.v; r.v = a.v * inv; for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv; return r; }
template <int N> OBVI_HD Dual<N> operator+(const Dual<N>& a, double b) { Dual<N> r = a; r.v += b; return r; }
template <int N> OBVI_HD Dual<N> operator-(const Dual<N>& a, double b) { Dual<N> r = a; r.v -= b; return r; }
template <int N> OBVI_HD Dual<N> operator*(const Dual<N>& a, double b) { Dual<N> r; r.v = a.v * b; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b; return r; }
template <int N> OBVI_HD Dual<N> operator*(double a, const Dual<N>& b) { return b * a; }
template <int N> OBVI_HD Dual<N> operator/(const Dual<N>& a, double b) { return a * (1.0 / b); }
template <int N> OBVI_HD Dual<N> dsqrt(const Dual<N>& a) { Dual<N> r; r.v = sqrt(a.v); const double s = 0.5 / r.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s; return r; }
template <int N> OBVI_HD Dual<N> dsin(const Dual<N>& a) { Dual<N> r; r.v = sin(a.v); const double c = cos(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * c; return r; }
template <int N> OBVI_HD Dual<N> dcos(const Dual<N>& a) { Dual<N> r; r.v = cos(a.v); const double s = -sin(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s; return r; }
template <int N> OBVI_HD Dual<N> datan2(const Dual<N>& y, const Dual<N>& x) { Dual<N> r; r.v = atan2(y.v, x.v); const double inv = 1.0 / (x.v * x.v + y.v * y.v); for (int i = 0; i < N; ++i) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) * inv; return r; }
template <int N> OBVI_HD Dual<N> dabs(const Dual<N>& a) { return a.v < 0.0 ? -a : a; }

// Rodrigues rotation of (angle, unit axis), the form Eigen::AngleAxis::toRotationMatrix evaluates
template <int N> OBVI_HD void dual_rotation(const Dual<N>& angle, const Dual<N>* axis, Dual<N>* R) {
  const Dual<N> s = dsin(angle), c = dcos(angle), oc = Dual<N>(1.0) - c;
  const Dual<N> cx = oc * axis[0], cy = oc * axis[1], cz = oc * axis[2];
  const Dual<N> sx = s * axis[0], sy = s * axis[1], sz = s * axis[2];
  Dual<N> t = cx * axis[1]; R[1] = t - sz; R[3] = t + sz;
  t = cx * axis[2]; R[2] = t + sy; R[6] = t - sy;
  t = cy * axis[2]; R[5] = t - sx; R[7] = t + sx;
  R[0] = cx * axis[0] + c; R[4] = cy * axis[1] + c; R[8] = cz * axis[2] + c;
}
template <int N> OBVI_HD void dual_identity(Dual<N>* R) { for (int i = 0; i < 9; ++i) R[i] = Dual<N>((i % 4 == 0) ? 1.0 : 0.0); }

// inverse robot pose (R^T, -R^T t) with the small-angle branch of ellipsoid_utils.h:172-192
template <int N> OBVI_HD void dual_inverse_pose(const Dual<N>* pose, Dual<N>* Rinv, Dual<N>* tinv) {
  const Dual<N> angle = dsqrt(pose[3] * pose[3] + pose[4] * pose[4] + pose[5] * pose[5]);
  if (angle.v > OBVI_SMALL_ANGLE) {
    const Dual<N> axis[3] = {pose[3] / angle, pose[4] / angle, pose[5] / angle};
    dual_rotation(-angle, axis, Rinv);
  } else {
    dual_identity(Rinv);
  }
  for (int i = 0; i < 3; ++i) tinv[i] = -(Rinv[3 * i] * pose[0] + Rinv[3 * i + 1] * pose[1] + Rinv[3 * i + 2] * pose[2]);
}

// rotation of a (t, aa) block, forward: PoseArrayToAffine (vslam_math_util.h:121-141) / VectorToAxisAngle (:31-42): identity at or below 1e-8
template <int N>
OBVI_HD void dual_forward_rotation(const Dual<N>* pose, Dual<N>* R) {
  typedef Dual<N> D12;
  const D12 angle = dsqrt(pose[3] * pose[3] + pose[4] * pose[4] + pose[5] * pose[5]);
  if (!(angle.v > OBVI_SMALL_ANGLE)) { dual_identity(R); return; }
  const D12 axis[3] = {pose[3] / angle, pose[4] / angle, pose[5] / angle};
  dual_rotation(angle, axis, R);
}

// BoundingBoxFactor::operator() (bounding_box_factor.h:68-136) over
// getCornerLocationsVectorRectified (ellipsoid_utils.h:160-273).  13 directions: ellipsoid 0..6,
// pose 7..12.  Returns false (constant residual, zero Jacobian) in the invalid-ellipse case.
// N = 13: value + Jacobian; N = 1: value only at about twice the cost of plain doubles (trial-point cost)
// dir >= 0 (N = 1): the single slot carries the derivative along parameter `dir` -- the lane-parallel linearisation kernels of small
// problems give every parameter its own lane, which costs a lane two plain evaluations instead of N + 1
template <int N> OBVI_HD Dual<N> dvar_n(double c, int k, int dir = -1) {
  Dual<N> r(c);
  if (dir >= 0) r.d[0] = (k == dir) ? 1.0 : 0.0;
  else if (k < N) r.d[k] = 1.0;
  return r;
}
// OD = 7: the reference's build (CONSTRAIN_ELLIPSOID_ORIENTATION, CMakeLists.txt:8-15): block (x y z yaw dx dy dz), rotation about z.
// OD = 9 (round 6): the unconstrained block of vslam_obj_opt_types_refactor.h:15-21 -- (x y z ax ay az dx dy dz), rotation VectorToAxisAngle(ax ay az)
// (ellipsoid_utils.h:217-229 `#else`; vslam_math_util.h:31-42: the constant AngleAxis(0, e_x) at or below 1e-8) -- OD + 6 directions, ellipsoid first.
template <int N, int OD = 7>
OBVI_HD bool bbox_eval_n(const double* ell_v, const double* pose_v, const DevCam& cam, const double* rect_corners,
                         const double* sqrt_inf, double invalid_err, Dual<N>* res, int dir = -1) {
  typedef Dual<N> D13;
  static_assert(OD == 7 || OD == 9, "ellipsoid block: 7 (yaw only) or 9 (axis-angle)");
  D13 ell[OD], pose[6];
  for (int k = 0; k < OD; ++k) ell[k] = dvar_n<N>(ell_v[k], k, dir);
  for (int k = 0; k < 6; ++k) pose[k] = dvar_n<N>(pose_v[k], OD + k, dir);
  D13 Rinv[9], tinv[3];
  dual_inverse_pose(pose, Rinv, tinv);
  D13 Rcw[9], tcw[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      Rcw[3 * i + j] = Rinv[j] * cam.Rinv[3 * i] + Rinv[3 + j] * cam.Rinv[3 * i + 1] + Rinv[6 + j] * cam.Rinv[3 * i + 2];
    tcw[i] = tinv[0] * cam.Rinv[3 * i] + tinv[1] * cam.Rinv[3 * i + 1] + tinv[2] * cam.Rinv[3 * i + 2] + cam.tinv[i];
  }
  D13 M[12];
  if (OD == 7) {
    const D13 cy = dcos(ell[3]), sy = dsin(ell[3]);
    for (int i = 0; i < 3; ++i) {
      M[4 * i + 0] = Rcw[3 * i] * cy + Rcw[3 * i + 1] * sy;
      M[4 * i + 1] = Rcw[3 * i + 1] * cy - Rcw[3 * i] * sy;
      M[4 * i + 2] = Rcw[3 * i + 2];
      M[4 * i + 3] = Rcw[3 * i] * ell[0] + Rcw[3 * i + 1] * ell[1] + Rcw[3 * i + 2] * ell[2] + tcw[i];
    }
  } else {
    D13 Re[9];
    dual_forward_rotation(ell, Re);   // reads ell[3..5]: the same (angle, axis) -> Rodrigues form and small-angle branch as a pose's rotation
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) M[4 * i + j] = Rcw[3 * i] * Re[j] + Rcw[3 * i + 1] * Re[3 + j] + Rcw[3 * i + 2] * Re[6 + j];
      M[4 * i + 3] = Rcw[3 * i] * ell[0] + Rcw[3 * i + 1] * ell[1] + Rcw[3 * i + 2] * ell[2] + tcw[i];
    }
  }
  D13 dm[3];
  for (int k = 0; k < 3; ++k) { const D13 h = ell[OD - 3 + k] * 0.5; dm[k] = h * h + OBVI_DIM_REG; }
#define OBVI_Q(a, b) (M[4 * a] * dm[0] * M[4 * b] + M[4 * a + 1] * dm[1] * M[4 * b + 1] + M[4 * a + 2] * dm[2] * M[4 * b + 2] - M[4 * a + 3] * M[4 * b + 3])
  const D13 q11 = OBVI_Q(0, 0), q13 = OBVI_Q(0, 2), q22 = OBVI_Q(1, 1), q23 = OBVI_Q(1, 2), q33 = OBVI_Q(2, 2);
#undef OBVI_Q
  const D13 xin = q13 * q13 - q11 * q33, yin = q23 * q23 - q22 * q33;
  if (xin.v <= 0.0 || yin.v <= 0.0) {
    for (int i = 0; i < 4; ++i) res[i] = D13(invalid_err);
    return false;
  }
  const D13 xs = dsqrt(xin), ys = dsqrt(yin);
  D13 dev[4];
  dev[0] = (q13 + xs) / q33 - rect_corners[0]; dev[1] = (q13 - xs) / q33 - rect_corners[1];
  dev[2] = (q23 + ys) / q33 - rect_corners[2]; dev[3] = (q23 - ys) / q33 - rect_corners[3];
  for (int i = 0; i < 4; ++i)
    res[i] = dev[0] * sqrt_inf[4 * i] + dev[1] * sqrt_inf[4 * i + 1] + dev[2] * sqrt_inf[4 * i + 2] + dev[3] * sqrt_inf[4 * i + 3];
  return true;
}
typedef Dual<13> D13;
OBVI_HD bool bbox_eval(const double* ell_v, const double* pose_v, const DevCam& cam, const double* rect_corners,
                       const double* sqrt_inf, double invalid_err, D13* res) {
  return bbox_eval_n<13>(ell_v, pose_v, cam, rect_corners, sqrt_inf, invalid_err, res);
}
// value only / one direction, ellipsoid block size at run time
OBVI_HD bool bbox_eval_1(int od, const double* ell_v, const double* pose_v, const DevCam& cam, const double* rect_corners,
                         const double* sqrt_inf, double invalid_err, Dual<1>* res, int dir = -1) {
  return od == 9 ? bbox_eval_n<1, 9>(ell_v, pose_v, cam, rect_corners, sqrt_inf, invalid_err, res, dir)
                 : bbox_eval_n<1, 7>(ell_v, pose_v, cam, rect_corners, sqrt_inf, invalid_err, res, dir);
}

// RelativePoseFactor::operator() (relative_pose_factor.h:32-61).  12 directions: pose_before
// 0..5, pose_after 6..11.  Pose rotation: PoseArrayToAffine (vslam_math_util.h:121-141);
// rotation log: Eigen::AngleAxis(Matrix3) via its quaternion (see oracle/README.md).
template <int N>
OBVI_HD void relpose_eval_n(const double* pa_v, const double* pb_v, const double* t_meas, const double* R_meas,
                            const double* sqrt_inf, Dual<N>* res, int dir = -1) {
  typedef Dual<N> D12;
  D12 pa[6], pb[6];
  for (int k = 0; k < 6; ++k) { pa[k] = dvar_n<N>(pa_v[k], k, dir); pb[k] = dvar_n<N>(pb_v[k], 6 + k, dir); }
  D12 Rb[9], Ra[9];
  dual_forward_rotation(pa, Rb);   // "before"
  dual_forward_rotation(pb, Ra);   // "after"
  const D12 dt[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
  D12 u[6];
  for (int i = 0; i < 3; ++i) u[i] = Rb[i] * dt[0] + Rb[3 + i] * dt[1] + Rb[6 + i] * dt[2] - t_meas[i];
  D12 Rrel[9], E[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
    Rrel[3 * i + j] = Rb[i] * Ra[j] + Rb[3 + i] * Ra[3 + j] + Rb[6 + i] * Ra[6 + j];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
    E[3 * i + j] = Rrel[3 * i] * R_meas[3 * j] + Rrel[3 * i + 1] * R_meas[3 * j + 1] + Rrel[3 * i + 2] * R_meas[3 * j + 2];
  // matrix -> quaternion (x y z w)
  D12 q[4];
  D12 t = E[0] + E[4] + E[8];
  if (t.v > 0.0) {
    t = dsqrt(t + 1.0);
    q[3] = t * 0.5;
    t = D12(0.5) / t;
    q[0] = (E[7] - E[5]) * t; q[1] = (E[2] - E[6]) * t; q[2] = (E[3] - E[1]) * t;
  } else {
    int i = 0;
    if (E[4].v > E[0].v) i = 1;
    if (E[8].v > E[4 * i].v) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = dsqrt(E[4 * i] - E[4 * j] - E[4 * k] + 1.0);
    q[i] = t * 0.5;
    t = D12(0.5) / t;
    q[3] = (E[3 * k + j] - E[3 * j + k]) * t;
    q[j] = (E[3 * j + i] + E[3 * i + j]) * t;
    q[k] = (E[3 * k + i] + E[3 * i + k]) * t;
  }
  D12 n = dsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  if (n.v != 0.0) {
    const D12 angle = datan2(n, dabs(q[3])) * 2.0;
    if (q[3].v < 0.0) n = -n;
    for (int a = 0; a < 3; ++a) u[3 + a] = angle * (q[a] / n);
  } else {
    for (int a = 0; a < 3; ++a) u[3 + a] = D12(0.0);
  }
  for (int i = 0; i < 6; ++i) {
    D12 acc = u[0] * sqrt_inf[6 * i];
    for (int j = 1; j < 6; ++j) acc = acc + u[j] * sqrt_inf[6 * i + j];
    res[i] = acc;
  }
}
typedef Dual<12> D12;
OBVI_HD void relpose_eval(const double* pa_v, const double* pb_v, const double* t_meas, const double* R_meas,
                          const double* sqrt_inf, D12* res) {
  relpose_eval_n<12>(pa_v, pb_v, t_meas, R_meas, sqrt_inf, res);
}

// ShapePriorFactor (shape_prior_factor.h:46-61) and IndependentObjectMapFactor
// (independent_object_map_factor.h:21-33) are linear in the ellipsoid block:
// r = A (e_sub - mean),  J = A placed in the matching columns.
OBVI_HD void shape_prior_eval(const double* ell, const double* mean3, const double* sqrt_inf, double* r, double* J /*3 x od*/, int od = 7) {
  const double d0 = ell[od - 3] - mean3[0], d1 = ell[od - 2] - mean3[1], d2 = ell[od - 1] - mean3[2];   // shape_prior_factor.h:49-51: the last three entries
  for (int i = 0; i < 3; ++i) {
    r[i] = d0 * sqrt_inf[3 * i] + d1 * sqrt_inf[3 * i + 1] + d2 * sqrt_inf[3 * i + 2];
    if (J) {
      for (int k = 0; k < od - 3; ++k) J[od * i + k] = 0.0;
      for (int k = 0; k < 3; ++k) J[od * i + od - 3 + k] = sqrt_inf[3 * i + k];
    }
  }
}
OBVI_HD void ltm_prior_eval(const double* ell, const double* mean, const double* sqrt_inf, double* r, double* J /*od x od*/, int od = 7) {
  for (int i = 0; i < od; ++i) {
    double acc = 0.0;
    for (int k = 0; k < od; ++k) acc += (ell[k] - mean[k]) * sqrt_inf[od * i + k];
    r[i] = acc;
    if (J) for (int k = 0; k < od; ++k) J[od * i + k] = sqrt_inf[od * i + k];
  }
}

}  // namespace obvi
#endif  // OBVI_BA_MATH_H_
