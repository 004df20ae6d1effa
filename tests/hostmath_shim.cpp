// Host build of the product's per-factor arithmetic (obvi-slam_amd/csrc/ba_math.h) so the CPU
// test-suite can compare it with the oracle without a GPU.  Test infrastructure only.
#include <cmath>
#include "../obvi-slam_amd/csrc/ba_math.h"
using namespace obvi;
static void make_cam(const double* K4, const double* e, DevCam* c) {
  double qx = e[0], qy = e[1], qz = e[2], qw = e[3];
  const double n = std::sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
  qx /= n; qy /= n; qz /= n; qw /= n;
  const double R[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw),
                       2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                       2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)};
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c->Rinv[3 * i + j] = R[3 * j + i];
  for (int i = 0; i < 3; ++i) c->tinv[i] = -(c->Rinv[3 * i] * e[4] + c->Rinv[3 * i + 1] * e[5] + c->Rinv[3 * i + 2] * e[6]);
  c->fx = K4[0]; c->fy = K4[1]; c->cx = K4[2]; c->cy = K4[3];
  c->depth_min = -INFINITY;
}
extern "C" {
void hostmath_reproj(const double* pose, const double* X, const double* K4, const double* ext7, const double* pix,
                     double sigma, double* r, double* Jp, double* Jl) {
  DevCam cam; make_cam(K4, ext7, &cam);
  PoseCache pc; make_pose_cache(pose, &pc);
  reproj_eval<true>(pc, cam, X, pix[0], pix[1], sigma, r, Jp, Jl);
}
// the analytic-Jacobian variant (obvi_ba_options.reprojection_variant = OBVI_REPROJECTION_ANALYTIC)
void hostmath_reproj_analytic(const double* pose, const double* X, const double* K4, const double* ext7, const double* pix,
                              double sigma, double* r, double* Jp, double* Jl) {
  DevCam cam; make_cam(K4, ext7, &cam);
  cam.depth_min = OBVI_ANALYTIC_EPSILON;
  PoseCache pc; make_pose_cache(pose, &pc, true);
  reproj_eval<true>(pc, cam, X, pix[0], pix[1], sigma, r, Jp, Jl);
}
int hostmath_bbox(const double* ell, const double* pose, const double* K4, const double* ext7, const double* rect,
                  const double* sqrt_inf, double invalid, double* r, double* Je, double* Jp) {
  DevCam cam; make_cam(K4, ext7, &cam);
  D13 res[4];
  const bool ok = bbox_eval(ell, pose, cam, rect, sqrt_inf, invalid, res);
  for (int a = 0; a < 4; ++a) { r[a] = res[a].v; for (int k = 0; k < 7; ++k) Je[7 * a + k] = res[a].d[k]; for (int k = 0; k < 6; ++k) Jp[6 * a + k] = res[a].d[7 + k]; }
  return ok ? 1 : 0;
}
// the 9-parameter ellipsoid block (obvi_ba_options.object_block_size = 9): 15 directions, ellipsoid first
int hostmath_bbox9(const double* ell, const double* pose, const double* K4, const double* ext7, const double* rect,
                   const double* sqrt_inf, double invalid, double* r, double* Je, double* Jp) {
  DevCam cam; make_cam(K4, ext7, &cam);
  Dual<15> res[4];
  const bool ok = bbox_eval_n<15, 9>(ell, pose, cam, rect, sqrt_inf, invalid, res);
  for (int a = 0; a < 4; ++a) { r[a] = res[a].v; for (int k = 0; k < 9; ++k) Je[9 * a + k] = res[a].d[k]; for (int k = 0; k < 6; ++k) Jp[6 * a + k] = res[a].d[9 + k]; }
  // the one-direction form the lane-parallel kernels use must give the same columns
  for (int dir = 0; dir < 15; ++dir) {
    Dual<1> one[4];
    bbox_eval_n<1, 9>(ell, pose, cam, rect, sqrt_inf, invalid, one, dir);
    for (int a = 0; a < 4; ++a) if (one[a].d[0] != res[a].d[dir] || one[a].v != res[a].v) return -1;
  }
  return ok ? 1 : 0;
}
void hostmath_relpose(const double* pa, const double* pb, const double* t, const double* R, const double* si, double* r, double* Ja, double* Jb) {
  D12 res[6]; relpose_eval(pa, pb, t, R, si, res);
  for (int a = 0; a < 6; ++a) { r[a] = res[a].v; for (int k = 0; k < 6; ++k) { Ja[6 * a + k] = res[a].d[k]; Jb[6 * a + k] = res[a].d[6 + k]; } }
}
}
