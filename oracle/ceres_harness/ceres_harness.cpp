// ceres_harness.cpp -- TEST INFRASTRUCTURE ONLY, optional (needs Ceres; see CMakeLists.txt beside this file).
//
// Solves a flat problem file (obvi-slam_amd/python/synth.py: dump_flat) with ceres::Solve exactly the way the reference does:
// AutoDiffCostFunction over the residual functors, HuberLoss per residual block, constant parameter blocks, and the option block of
// ObjectPoseGraphOptimizer::solveOptimization (include/refactoring/optimization/object_pose_graph_optimizer.h:651-676 of the
// reference: max_num_iterations, num_threads = 20, SPARSE_SCHUR, use_nonmonotonic_steps, the three tolerances, initial / max
// trust-region radius; everything else the Ceres default).  The residual functors are the oracle's templates
// (oracle/oracle_factors.h, each citing the reference functor it restates) instantiated with ceres::Jet, i.e. the arithmetic the
// reference's Ceres path differentiates.  Output: one JSON object on stdout -- iterations (cost, step_is_successful, ...), timings of
// Solver::Summary, final parameter blocks -- which tests/ and bench.py compare with the oracle and the HIP path ("reference Ceres
// path").
//
//   ceres_harness problem.flat [threads=20]
#include <ceres/ceres.h>
#include <ceres/jet.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace oracle {
template <class T, int N> inline double val(const ceres::Jet<T, N>& j) { return j.a; }   // the oracle's branch tests look at the value
}  // namespace oracle
#include "oracle_factors.h"

namespace {
using oracle::CameraConst;

struct Flat {
  std::vector<CameraConst> cams;
  std::vector<double> poses, points, objects;
  std::vector<uint8_t> pose_const, point_const, object_const;
  std::vector<uint32_t> rp_pose, rp_point, rp_cam; std::vector<double> rp_pixel; double rp_sigma = 1, rp_huber = 1;
  std::vector<uint32_t> bb_obj, bb_pose, bb_cam; std::vector<double> bb_corners, bb_cov; double bb_huber = 1, bb_invalid = 1e3;
  std::vector<uint32_t> sp_obj; std::vector<double> sp_mean, sp_cov; double sp_huber = 1;
  std::vector<uint32_t> lt_obj; std::vector<double> lt_mean, lt_cov; double lt_huber = 1;
  std::vector<uint32_t> rl_a, rl_b; std::vector<double> rl_t, rl_aa, rl_cov; double rl_huber = 1;
  int32_t max_it = 3, nonmono = 1; double ftol = 0, gtol = 0, ptol = 0, radius = 100, max_radius = 1e4;
};

template <class T> bool rd(FILE* f, std::vector<T>* v) {
  uint64_t n = 0;
  if (std::fread(&n, 8, 1, f) != 1) return false;
  v->resize(n);
  return n == 0 || std::fread(v->data(), sizeof(T), n, f) == n;
}
bool rdd(FILE* f, double* d) { return std::fread(d, 8, 1, f) == 1; }

// file layout: see synth.dump_flat (little endian; arrays as u64 count + payload, scalars as f64)
bool load(const char* path, Flat* p) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  char magic[8];
  if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "OBVIFLT1", 8) != 0) { std::fclose(f); return false; }
  std::vector<double> K, ext;
  bool ok = rd(f, &K) && rd(f, &ext) && rd(f, &p->poses) && rd(f, &p->pose_const) && rd(f, &p->points) && rd(f, &p->point_const) && rd(f, &p->objects) &&
            rd(f, &p->object_const) && rd(f, &p->rp_pose) && rd(f, &p->rp_point) && rd(f, &p->rp_cam) && rd(f, &p->rp_pixel) && rdd(f, &p->rp_sigma) && rdd(f, &p->rp_huber) &&
            rd(f, &p->bb_obj) && rd(f, &p->bb_pose) && rd(f, &p->bb_cam) && rd(f, &p->bb_corners) && rd(f, &p->bb_cov) && rdd(f, &p->bb_huber) && rdd(f, &p->bb_invalid) &&
            rd(f, &p->sp_obj) && rd(f, &p->sp_mean) && rd(f, &p->sp_cov) && rdd(f, &p->sp_huber) && rd(f, &p->lt_obj) && rd(f, &p->lt_mean) && rd(f, &p->lt_cov) && rdd(f, &p->lt_huber) &&
            rd(f, &p->rl_a) && rd(f, &p->rl_b) && rd(f, &p->rl_t) && rd(f, &p->rl_aa) && rd(f, &p->rl_cov) && rdd(f, &p->rl_huber);
  double prm[7];
  ok = ok && std::fread(prm, 8, 7, f) == 7;
  std::fclose(f);
  if (!ok) return false;
  p->max_it = (int32_t)prm[0]; p->nonmono = (int32_t)prm[1]; p->ftol = prm[2]; p->gtol = prm[3]; p->ptol = prm[4]; p->radius = prm[5]; p->max_radius = prm[6];
  p->cams.resize(K.size() / 4);
  for (size_t i = 0; i < p->cams.size(); ++i) oracle::make_camera_const(&K[4 * i], &ext[7 * i], &p->cams[i]);
  return true;
}

struct Reproj {
  const CameraConst* cam; double pixel[2], sigma;
  template <class T> bool operator()(const T* pose, const T* point, T* r) const { oracle::reprojection_residual(pose, point, *cam, pixel, sigma, r); return true; }
};
struct Bbox {
  const CameraConst* cam; double rect[4], sqrt_inf[16], invalid;
  template <class T> bool operator()(const T* ell, const T* pose, T* r) const { oracle::bbox_residual(ell, pose, *cam, rect, sqrt_inf, invalid, r); return true; }
};
struct Shape {
  double mean[3], sqrt_inf[9];
  template <class T> bool operator()(const T* ell, T* r) const { oracle::shape_prior_residual(ell, mean, sqrt_inf, r); return true; }
};
struct Ltm {
  double mean[7], sqrt_inf[49];
  template <class T> bool operator()(const T* ell, T* r) const { oracle::ltm_prior_residual(ell, mean, sqrt_inf, r); return true; }
};
struct RelPose {
  double t[3], R[9], sqrt_inf[36];
  template <class T> bool operator()(const T* a, const T* b, T* r) const { oracle::relpose_residual(a, b, t, R, sqrt_inf, r); return true; }
};

void aa_to_matrix(const double* a, double* R) {   // measured_pose_deviation.orientation_.toRotationMatrix() (relative_pose_factor.cpp:11-12)
  const double th = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  if (!(th > 0.0)) { for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0; return; }
  const double axis[3] = {a[0] / th, a[1] / th, a[2] / th};
  oracle::angle_axis_to_matrix(th, axis, R);
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: ceres_harness problem.flat [threads]\n"); return 2; }
  Flat p;
  if (!load(argv[1], &p)) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  const int threads = argc > 2 ? std::atoi(argv[2]) : 20;
  ceres::Problem problem;
  const size_t P = p.poses.size() / 6, L = p.points.size() / 3, O = p.objects.size() / 7;
  for (size_t i = 0; i < P; ++i) problem.AddParameterBlock(&p.poses[6 * i], 6);
  for (size_t i = 0; i < L; ++i) problem.AddParameterBlock(&p.points[3 * i], 3);
  for (size_t i = 0; i < O; ++i) problem.AddParameterBlock(&p.objects[7 * i], 7);
  for (size_t i = 0; i < p.rp_pose.size(); ++i) {
    Reproj* f = new Reproj{&p.cams[p.rp_cam[i]], {p.rp_pixel[2 * i], p.rp_pixel[2 * i + 1]}, p.rp_sigma};
    problem.AddResidualBlock(new ceres::AutoDiffCostFunction<Reproj, 2, 6, 3>(f), new ceres::HuberLoss(p.rp_huber), &p.poses[6 * p.rp_pose[i]], &p.points[3 * p.rp_point[i]]);
  }
  for (size_t i = 0; i < p.bb_obj.size(); ++i) {
    Bbox* f = new Bbox();
    const CameraConst& c = p.cams[p.bb_cam[i]];
    f->cam = &c; f->invalid = p.bb_invalid;
    double m4[16];
    if (!oracle::spd_inverse_sqrt(&p.bb_cov[16 * i], 4, m4)) { std::fprintf(stderr, "bbox covariance %zu not SPD\n", i); return 2; }
    const double sc[4] = {c.fx, c.fx, c.fy, c.fy};   // bounding_box_factor.cpp:26-39
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) f->sqrt_inf[4 * a + b] = m4[4 * a + b] * sc[b];
    f->rect[0] = (p.bb_corners[4 * i] - c.cx) / c.fx; f->rect[1] = (p.bb_corners[4 * i + 1] - c.cx) / c.fx;
    f->rect[2] = (p.bb_corners[4 * i + 2] - c.cy) / c.fy; f->rect[3] = (p.bb_corners[4 * i + 3] - c.cy) / c.fy;
    problem.AddResidualBlock(new ceres::AutoDiffCostFunction<Bbox, 4, 7, 6>(f), new ceres::HuberLoss(p.bb_huber), &p.objects[7 * p.bb_obj[i]], &p.poses[6 * p.bb_pose[i]]);
  }
  for (size_t i = 0; i < p.sp_obj.size(); ++i) {
    Shape* f = new Shape();
    std::memcpy(f->mean, &p.sp_mean[3 * i], sizeof(f->mean));
    if (!oracle::spd_inverse_sqrt(&p.sp_cov[9 * i], 3, f->sqrt_inf)) return 2;
    problem.AddResidualBlock(new ceres::AutoDiffCostFunction<Shape, 3, 7>(f), new ceres::HuberLoss(p.sp_huber), &p.objects[7 * p.sp_obj[i]]);
  }
  for (size_t i = 0; i < p.lt_obj.size(); ++i) {
    Ltm* f = new Ltm();
    std::memcpy(f->mean, &p.lt_mean[7 * i], sizeof(f->mean));
    if (!oracle::spd_inverse_sqrt(&p.lt_cov[49 * i], 7, f->sqrt_inf)) return 2;
    problem.AddResidualBlock(new ceres::AutoDiffCostFunction<Ltm, 7, 7>(f), new ceres::HuberLoss(p.lt_huber), &p.objects[7 * p.lt_obj[i]]);
  }
  for (size_t i = 0; i < p.rl_a.size(); ++i) {
    RelPose* f = new RelPose();
    std::memcpy(f->t, &p.rl_t[3 * i], sizeof(f->t));
    aa_to_matrix(&p.rl_aa[3 * i], f->R);
    if (!oracle::spd_inverse_sqrt(&p.rl_cov[36 * i], 6, f->sqrt_inf)) return 2;
    problem.AddResidualBlock(new ceres::AutoDiffCostFunction<RelPose, 6, 6, 6>(f), new ceres::HuberLoss(p.rl_huber), &p.poses[6 * p.rl_a[i]], &p.poses[6 * p.rl_b[i]]);
  }
  for (size_t i = 0; i < P; ++i) if (p.pose_const[i]) problem.SetParameterBlockConstant(&p.poses[6 * i]);
  for (size_t i = 0; i < L; ++i) if (p.point_const[i]) problem.SetParameterBlockConstant(&p.points[3 * i]);
  for (size_t i = 0; i < O; ++i) if (p.object_const[i]) problem.SetParameterBlockConstant(&p.objects[7 * i]);

  // object_pose_graph_optimizer.h:651-676
  ceres::Solver::Options options;
  options.max_num_iterations = p.max_it;
  options.num_threads = threads;
  options.linear_solver_type = ceres::SPARSE_SCHUR;
  options.use_nonmonotonic_steps = p.nonmono != 0;
  options.function_tolerance = p.ftol;
  options.gradient_tolerance = p.gtol;
  options.parameter_tolerance = p.ptol;
  options.initial_trust_region_radius = p.radius;
  options.max_trust_region_radius = p.max_radius;
  ceres::Solver::Summary summary;
  ceres::Solve(options, &problem, &summary);

  std::printf("{\"ceres_version\": \"%s\", \"threads\": %d, \"termination_type\": %d, \"num_iterations\": %zu, \"initial_cost\": %.17g, \"final_cost\": %.17g,\n",
              CERES_VERSION_STRING, threads, (int)summary.termination_type, summary.iterations.size(), summary.initial_cost, summary.final_cost);
  std::printf(" \"total_time_in_seconds\": %.6f, \"linear_solver_time_in_seconds\": %.6f, \"jacobian_evaluation_time_in_seconds\": %.6f, \"residual_evaluation_time_in_seconds\": %.6f,\n",
              summary.total_time_in_seconds, summary.linear_solver_time_in_seconds, summary.jacobian_evaluation_time_in_seconds, summary.residual_evaluation_time_in_seconds);
  std::printf(" \"iterations\": [");
  for (size_t i = 0; i < summary.iterations.size(); ++i) {
    const ceres::IterationSummary& it = summary.iterations[i];
    std::printf("%s{\"iteration\": %d, \"cost\": %.17g, \"cost_change\": %.17g, \"step_is_valid\": %d, \"step_is_successful\": %d, \"step_norm\": %.17g, \"relative_decrease\": %.17g, \"trust_region_radius\": %.17g, \"gradient_max_norm\": %.17g, \"iteration_time_in_seconds\": %.6f}",
                i ? ", " : "", it.iteration, it.cost, it.cost_change, (int)it.step_is_valid, (int)it.step_is_successful, it.step_norm, it.relative_decrease, it.trust_region_radius, it.gradient_max_norm, it.iteration_time_in_seconds);
  }
  std::printf("],\n \"poses\": [");
  for (size_t i = 0; i < p.poses.size(); ++i) std::printf("%s%.17g", i ? "," : "", p.poses[i]);
  std::printf("],\n \"objects\": [");
  for (size_t i = 0; i < p.objects.size(); ++i) std::printf("%s%.17g", i ? "," : "", p.objects[i]);
  std::printf("]}\n");
  return summary.IsSolutionUsable() ? 0 : 1;
}
