// Test infrastructure (see lockstep_shim.h): the C ABI forwarded to the HIP library and to the CPU oracle in lock step.
// Compiled WITHOUT the renaming header, so obvi_ba_* below are the real entry points of libobvi_ba.so.
// One JSON line per compared call goes to $OBVI_LOCKSTEP_LOG.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/obvi_ba.h"

struct oracle_handle;
extern "C" {
int oracle_ba_create(const obvi_ba_options*, oracle_handle**);
void oracle_ba_destroy(oracle_handle*);
int oracle_ba_reset(oracle_handle*);
int oracle_ba_set_cameras(oracle_handle*, int32_t, const double*, const double*);
int oracle_ba_set_poses(oracle_handle*, int64_t, const double*, const uint8_t*);
int oracle_ba_set_points(oracle_handle*, int64_t, const double*, const uint8_t*);
int oracle_ba_set_objects(oracle_handle*, int64_t, const double*, const uint8_t*);
int oracle_ba_set_const_flags(oracle_handle*, const uint8_t*, const uint8_t*, const uint8_t*);
int oracle_ba_set_reproj(oracle_handle*, int64_t, const uint32_t*, const uint32_t*, const uint16_t*, const double*, const double*, double, double);
int oracle_ba_set_bbox(oracle_handle*, int64_t, const uint32_t*, const uint32_t*, const uint16_t*, const double*, const double*, double, double);
int oracle_ba_set_shape_priors(oracle_handle*, int64_t, const uint32_t*, const double*, const double*, double);
int oracle_ba_set_ltm_priors(oracle_handle*, int64_t, const uint32_t*, const double*, const double*, double);
int oracle_ba_set_relpose(oracle_handle*, int64_t, const uint32_t*, const uint32_t*, const double*, const double*, const double*, double);
int oracle_ba_set_active_mask(oracle_handle*, int32_t, const uint8_t*);
int oracle_ba_set_parameter_priors(oracle_handle*, int64_t, const uint8_t*, const uint32_t*, const uint8_t*, const double*, const double*);
int oracle_ba_column_sqnorms(oracle_handle*, double*, double*, double*);
int oracle_ba_evaluate(oracle_handle*, int32_t, double*, double*, double*);
int oracle_ba_solve(oracle_handle*, const obvi_solver_params*, obvi_summary*);
int oracle_ba_get_iterations(const oracle_handle*, obvi_iteration_summary*, int32_t);
int oracle_ba_select_outliers(oracle_handle*, int32_t, double, uint8_t*, int64_t*);
int oracle_ba_object_covariances(oracle_handle*, int64_t, const uint32_t*, const uint32_t*, double*);
int oracle_ba_snapshot(oracle_handle*);
int oracle_ba_restore(oracle_handle*);
int oracle_ba_get_poses(oracle_handle*, double*);
int oracle_ba_get_points(oracle_handle*, double*);
int oracle_ba_get_objects(oracle_handle*, double*);
int oracle_ba_update_points(oracle_handle*, int64_t, const double*);
int oracle_ba_update_poses(oracle_handle*, int64_t, const double*);
int oracle_ba_update_objects(oracle_handle*, int64_t, const double*);
int64_t oracle_ba_num_residuals(const oracle_handle*);
int64_t oracle_ba_num_factors(const oracle_handle*, int32_t);
}

namespace {
// Optional third backend, the ARBITER: $OBVI_LOCKSTEP_ARBITER = path of oracle/libobvi_oracle_ld.so (the oracle's source with every
// solver-level sum in extended precision).  Same `oracle_` symbols as the checker this file links, hence dlopen(RTLD_LOCAL) + dlsym.
// It receives every upload, starts every solve from the same values, and its solve is logged beside the other two: when HIP and the
// oracle part ways, |HIP - arbiter| against |oracle - arbiter| says whether one of them is the outlier.
struct Arbiter {
  void* lib = nullptr;
  template <class F> void sym(F& f, const char* name) { f = lib ? reinterpret_cast<F>(dlsym(lib, name)) : nullptr; if (lib && !f) { std::fprintf(stderr, "lockstep: arbiter lacks %s\n", name); std::abort(); } }
  decltype(&oracle_ba_create) create; decltype(&oracle_ba_destroy) destroy; decltype(&oracle_ba_reset) reset; decltype(&oracle_ba_set_cameras) set_cameras; decltype(&oracle_ba_set_poses) set_poses;
  decltype(&oracle_ba_set_points) set_points; decltype(&oracle_ba_set_objects) set_objects; decltype(&oracle_ba_set_const_flags) set_const_flags;
  decltype(&oracle_ba_set_reproj) set_reproj; decltype(&oracle_ba_set_bbox) set_bbox; decltype(&oracle_ba_set_shape_priors) set_shape_priors;
  decltype(&oracle_ba_set_ltm_priors) set_ltm_priors; decltype(&oracle_ba_set_relpose) set_relpose; decltype(&oracle_ba_set_active_mask) set_active_mask;
  decltype(&oracle_ba_set_parameter_priors) set_parameter_priors; decltype(&oracle_ba_solve) solve; decltype(&oracle_ba_get_iterations) get_iterations;
  decltype(&oracle_ba_snapshot) snapshot; decltype(&oracle_ba_restore) restore; decltype(&oracle_ba_get_poses) get_poses; decltype(&oracle_ba_get_points) get_points;
  decltype(&oracle_ba_get_objects) get_objects; decltype(&oracle_ba_update_points) update_points; decltype(&oracle_ba_update_poses) update_poses; decltype(&oracle_ba_update_objects) update_objects;
  void (*set_threads)(int32_t);
  Arbiter() {
    const char* path = std::getenv("OBVI_LOCKSTEP_ARBITER");
    if (!path || !*path) return;
    lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) { std::fprintf(stderr, "lockstep: cannot load the arbiter %s: %s\n", path, dlerror()); std::abort(); }
    sym(create, "oracle_ba_create"); sym(destroy, "oracle_ba_destroy"); sym(reset, "oracle_ba_reset"); sym(set_cameras, "oracle_ba_set_cameras"); sym(set_poses, "oracle_ba_set_poses"); sym(set_points, "oracle_ba_set_points");
    sym(set_objects, "oracle_ba_set_objects"); sym(set_const_flags, "oracle_ba_set_const_flags"); sym(set_reproj, "oracle_ba_set_reproj"); sym(set_bbox, "oracle_ba_set_bbox");
    sym(set_shape_priors, "oracle_ba_set_shape_priors"); sym(set_ltm_priors, "oracle_ba_set_ltm_priors"); sym(set_relpose, "oracle_ba_set_relpose"); sym(set_active_mask, "oracle_ba_set_active_mask");
    sym(set_parameter_priors, "oracle_ba_set_parameter_priors"); sym(solve, "oracle_ba_solve"); sym(get_iterations, "oracle_ba_get_iterations"); sym(snapshot, "oracle_ba_snapshot");
    sym(restore, "oracle_ba_restore"); sym(get_poses, "oracle_ba_get_poses"); sym(get_points, "oracle_ba_get_points"); sym(get_objects, "oracle_ba_get_objects");
    sym(update_points, "oracle_ba_update_points"); sym(update_poses, "oracle_ba_update_poses"); sym(update_objects, "oracle_ba_update_objects"); sym(set_threads, "oracle_set_threads");
    if (const char* t = std::getenv("OBVI_LOCKSTEP_ARBITER_THREADS")) set_threads(std::atoi(t));
  }
};
Arbiter& arb() { static Arbiter a; return a; }
#define ARB(call) do { if (arb().lib && L_(h)->arb) (void)arb().call; } while (0)
struct Lock {
  obvi_ba_handle* hip = nullptr;
  oracle_handle* ora = nullptr;
  oracle_handle* arb = nullptr;
  int64_t P = 0, L = 0, O = 0;
};
Lock* L_(obvi_ba_handle* h) { return reinterpret_cast<Lock*>(h); }
const Lock* L_(const obvi_ba_handle* h) { return reinterpret_cast<const Lock*>(h); }
FILE* log_file() {
  static FILE* f = [] { const char* p = std::getenv("OBVI_LOCKSTEP_LOG"); return p ? std::fopen(p, "w") : nullptr; }();
  return f;
}
double max_abs_diff(const std::vector<double>& a, const std::vector<double>& b) {
  double m = 0.0;
  for (size_t i = 0; i < a.size() && i < b.size(); ++i) { const double d = std::fabs(a[i] - b[i]); if (!(d <= m)) m = d; }
  return m;
}
// without the yaw (index 3 of an ellipsoid block): an ellipsoid with equal horizontal axes does not constrain it, and it drifts freely in any solver
double object_diff(const std::vector<double>& a, const std::vector<double>& b) {
  double m = 0.0;
  for (size_t i = 0; i < a.size() && i < b.size(); ++i) if (i % 7 != 3) { const double d = std::fabs(a[i] - b[i]); if (!(d <= m)) m = d; }
  return m;
}
// The same difference in units of the ORACLE's own uncertainty: sqrt(d^T Sigma^-1 d) per object, d = HIP block - oracle block (all seven
// entries, the yaw included: a direction the data does not constrain has a large variance and weighs accordingly), Sigma = the oracle's 7x7 covariance block of the
// object at ITS end state (oracle_ba_object_covariances: the block of (J^T J)^-1 the long-term map stores).  Returns the largest over the objects; -1 where the
// oracle's normal equations are rank deficient there (no covariance exists) or an object's block is not positive definite.
double whitened_object_diff(oracle_handle* ora, int64_t O, const std::vector<double>& hip, const std::vector<double>& orc) {
  if (O <= 0) return 0.0;
  std::vector<uint32_t> idx((size_t)O);
  for (int64_t o = 0; o < O; ++o) idx[(size_t)o] = (uint32_t)o;
  std::vector<double> cov((size_t)49 * (size_t)O);
  if (oracle_ba_object_covariances(ora, O, idx.data(), idx.data(), cov.data()) != 0) return -1.0;
  double worst = 0.0;
  for (int64_t o = 0; o < O; ++o) {
    const double* S = &cov[(size_t)49 * (size_t)o];
    bool zero = true;
    for (int k = 0; k < 49; ++k) if (S[k] != 0.0) zero = false;
    if (zero) continue;                                   // a constant / unused object: no block
    double Lc[49] = {0}, y[7];
    bool ok = true;
    for (int i = 0; i < 7 && ok; ++i)
      for (int j = 0; j <= i; ++j) {
        double v = S[7 * i + j];
        for (int k = 0; k < j; ++k) v -= Lc[7 * i + k] * Lc[7 * j + k];
        if (i == j) { if (!(v > 0.0)) { ok = false; break; } Lc[7 * i + i] = std::sqrt(v); } else Lc[7 * i + j] = v / Lc[7 * j + j];
      }
    if (!ok) return -1.0;
    double q = 0.0;
    for (int i = 0; i < 7; ++i) {                         // y = L^-1 d ;  d^T Sigma^-1 d = |y|^2
      double v = hip[(size_t)(7 * o + i)] - orc[(size_t)(7 * o + i)];
      if (i == 3) v = std::remainder(v, 3.14159265358979323846);   // an ellipsoid's yaw is defined modulo pi; unobservable yaws drift by thousands of radians in any two runs
      for (int k = 0; k < i; ++k) v -= Lc[7 * i + k] * y[k];
      y[i] = v / Lc[7 * i + i]; q += y[i] * y[i];
    }
    worst = std::max(worst, std::sqrt(q));
  }
  return worst;
}
double rel(double a, double b) { return std::fabs(a - b) / std::max(std::fabs(b), 1e-300); }
int both(int rh, int ro, const char* what) {
  if ((rh == 0) != (ro == 0)) { std::fprintf(stderr, "lockstep: %s: HIP status %d, oracle status %d\n", what, rh, ro); if (FILE* f = log_file()) { std::fprintf(f, "{\"call\": \"status\", \"what\": \"%s\", \"hip\": %d, \"oracle\": %d}\n", what, rh, ro); std::fflush(f); } }
  return rh;
}
}  // namespace

extern "C" {
int lock_ba_create(const obvi_ba_options* opt, obvi_ba_handle** out) {
  Lock* l = new Lock();
  const int rh = obvi_ba_create(opt, &l->hip), ro = oracle_ba_create(opt, &l->ora);
  if (rh != 0 || ro != 0) { if (l->hip) obvi_ba_destroy(l->hip); if (l->ora) oracle_ba_destroy(l->ora); delete l; *out = nullptr; return rh ? rh : ro; }
  if (arb().lib && arb().create(opt, &l->arb) != 0) l->arb = nullptr;
  *out = reinterpret_cast<obvi_ba_handle*>(l);
  return 0;
}
void lock_ba_destroy(obvi_ba_handle* h) { if (!h) return; Lock* l = L_(h); obvi_ba_destroy(l->hip); oracle_ba_destroy(l->ora); if (l->arb) arb().destroy(l->arb); delete l; }
const char* lock_ba_last_error(const obvi_ba_handle* h) { return h ? obvi_ba_last_error(L_(h)->hip) : "null handle"; }
int lock_ba_set_cameras(obvi_ba_handle* h, int32_t n, const double* K, const double* e) { ARB(set_cameras(L_(h)->arb, n, K, e)); return both(obvi_ba_set_cameras(L_(h)->hip, n, K, e), oracle_ba_set_cameras(L_(h)->ora, n, K, e), "set_cameras"); }
int lock_ba_set_poses(obvi_ba_handle* h, int64_t n, const double* v, const uint8_t* c) { L_(h)->P = n; ARB(set_poses(L_(h)->arb, n, v, c)); return both(obvi_ba_set_poses(L_(h)->hip, n, v, c), oracle_ba_set_poses(L_(h)->ora, n, v, c), "set_poses"); }
int lock_ba_set_points(obvi_ba_handle* h, int64_t n, const double* v, const uint8_t* c) { L_(h)->L = n; ARB(set_points(L_(h)->arb, n, v, c)); return both(obvi_ba_set_points(L_(h)->hip, n, v, c), oracle_ba_set_points(L_(h)->ora, n, v, c), "set_points"); }
int lock_ba_set_objects(obvi_ba_handle* h, int64_t n, const double* v, const uint8_t* c) { L_(h)->O = n; ARB(set_objects(L_(h)->arb, n, v, c)); return both(obvi_ba_set_objects(L_(h)->hip, n, v, c), oracle_ba_set_objects(L_(h)->ora, n, v, c), "set_objects"); }
int lock_ba_set_const_flags(obvi_ba_handle* h, const uint8_t* a, const uint8_t* b, const uint8_t* c) { ARB(set_const_flags(L_(h)->arb, a, b, c)); return both(obvi_ba_set_const_flags(L_(h)->hip, a, b, c), oracle_ba_set_const_flags(L_(h)->ora, a, b, c), "set_const_flags"); }
int lock_ba_set_reproj(obvi_ba_handle* h, int64_t n, const uint32_t* a, const uint32_t* b, const uint16_t* c, const double* px, const double* sg, double ss, double hu) {
  ARB(set_reproj(L_(h)->arb, n, a, b, c, px, sg, ss, hu)); return both(obvi_ba_set_reproj(L_(h)->hip, n, a, b, c, px, sg, ss, hu), oracle_ba_set_reproj(L_(h)->ora, n, a, b, c, px, sg, ss, hu), "set_reproj");
}
int lock_ba_set_bbox(obvi_ba_handle* h, int64_t n, const uint32_t* a, const uint32_t* b, const uint16_t* c, const double* co, const double* cv, double hu, double inv) {
  ARB(set_bbox(L_(h)->arb, n, a, b, c, co, cv, hu, inv)); return both(obvi_ba_set_bbox(L_(h)->hip, n, a, b, c, co, cv, hu, inv), oracle_ba_set_bbox(L_(h)->ora, n, a, b, c, co, cv, hu, inv), "set_bbox");
}
int lock_ba_set_shape_priors(obvi_ba_handle* h, int64_t n, const uint32_t* a, const double* m, const double* c, double hu) { ARB(set_shape_priors(L_(h)->arb, n, a, m, c, hu)); return both(obvi_ba_set_shape_priors(L_(h)->hip, n, a, m, c, hu), oracle_ba_set_shape_priors(L_(h)->ora, n, a, m, c, hu), "set_shape_priors"); }
int lock_ba_set_ltm_priors(obvi_ba_handle* h, int64_t n, const uint32_t* a, const double* m, const double* c, double hu) { ARB(set_ltm_priors(L_(h)->arb, n, a, m, c, hu)); return both(obvi_ba_set_ltm_priors(L_(h)->hip, n, a, m, c, hu), oracle_ba_set_ltm_priors(L_(h)->ora, n, a, m, c, hu), "set_ltm_priors"); }
int lock_ba_set_relpose(obvi_ba_handle* h, int64_t n, const uint32_t* a, const uint32_t* b, const double* t, const double* aa, const double* c, double hu) {
  ARB(set_relpose(L_(h)->arb, n, a, b, t, aa, c, hu)); return both(obvi_ba_set_relpose(L_(h)->hip, n, a, b, t, aa, c, hu), oracle_ba_set_relpose(L_(h)->ora, n, a, b, t, aa, c, hu), "set_relpose");
}
int lock_ba_set_active_mask(obvi_ba_handle* h, int32_t t, const uint8_t* m) { ARB(set_active_mask(L_(h)->arb, t, m)); return both(obvi_ba_set_active_mask(L_(h)->hip, t, m), oracle_ba_set_active_mask(L_(h)->ora, t, m), "set_active_mask"); }
int lock_ba_set_parameter_priors(obvi_ba_handle* h, int64_t n, const uint8_t* k, const uint32_t* b, const uint8_t* p, const double* m, const double* s) {
  ARB(set_parameter_priors(L_(h)->arb, n, k, b, p, m, s)); return both(obvi_ba_set_parameter_priors(L_(h)->hip, n, k, b, p, m, s), oracle_ba_set_parameter_priors(L_(h)->ora, n, k, b, p, m, s), "set_parameter_priors");
}
int64_t lock_ba_num_residuals(const obvi_ba_handle* h) { return obvi_ba_num_residuals(L_(h)->hip); }
int64_t lock_ba_num_factors(const obvi_ba_handle* h, int32_t t) { return obvi_ba_num_factors(L_(h)->hip, t); }
int lock_ba_reset(obvi_ba_handle* h) { ARB(reset(L_(h)->arb)); return both(obvi_ba_reset(L_(h)->hip), oracle_ba_reset(L_(h)->ora), "reset"); }
int lock_ba_snapshot(obvi_ba_handle* h) { ARB(snapshot(L_(h)->arb)); return both(obvi_ba_snapshot(L_(h)->hip), oracle_ba_snapshot(L_(h)->ora), "snapshot"); }
int lock_ba_restore(obvi_ba_handle* h) { ARB(restore(L_(h)->arb)); return both(obvi_ba_restore(L_(h)->hip), oracle_ba_restore(L_(h)->ora), "restore"); }
int lock_ba_get_poses(obvi_ba_handle* h, double* out) { return obvi_ba_get_poses(L_(h)->hip, out); }
int lock_ba_get_points(obvi_ba_handle* h, double* out) { return obvi_ba_get_points(L_(h)->hip, out); }
int lock_ba_get_objects(obvi_ba_handle* h, double* out) { return obvi_ba_get_objects(L_(h)->hip, out); }
int lock_ba_get_state(obvi_ba_handle* h, double* poses, double* points, double* objects) { return obvi_ba_get_state(L_(h)->hip, poses, points, objects); }
int lock_ba_update_points(obvi_ba_handle* h, int64_t n, const double* x) { ARB(update_points(L_(h)->arb, n, x)); return both(obvi_ba_update_points(L_(h)->hip, n, x), oracle_ba_update_points(L_(h)->ora, n, x), "update_points"); }
int lock_ba_update_state(obvi_ba_handle* h, const double* po, const double* pt, const double* ob) {
  Lock* l = L_(h);
  if (l->arb) { if (po) arb().update_poses(l->arb, l->P, po); if (pt) arb().update_points(l->arb, l->L, pt); if (ob) arb().update_objects(l->arb, l->O, ob); }
  if (po) oracle_ba_update_poses(l->ora, l->P, po);
  if (pt) oracle_ba_update_points(l->ora, l->L, pt);
  if (ob) oracle_ba_update_objects(l->ora, l->O, ob);
  return obvi_ba_update_state(l->hip, po, pt, ob);
}
int lock_ba_prepare(obvi_ba_handle* h) { return obvi_ba_prepare(L_(h)->hip); }   // the oracles have no symbolic phase
int lock_ba_get_iterations(const obvi_ba_handle* h, obvi_iteration_summary* out, int32_t cap) { return obvi_ba_get_iterations(L_(h)->hip, out, cap); }

int lock_ba_evaluate(obvi_ba_handle* h, int32_t loss, double* cost, double* res, double* sq) {
  Lock* l = L_(h);
  const int64_t nres = obvi_ba_num_residuals(l->hip);
  int64_t nfac = 0;
  for (int t : {0, 2, 3, 4, 5}) nfac += obvi_ba_num_factors(l->hip, t);
  std::vector<double> rh((size_t)nres), ro((size_t)nres), qh((size_t)nfac), qo((size_t)nfac);
  double ch = 0.0, co = 0.0;
  const int a = obvi_ba_evaluate(l->hip, loss, &ch, rh.data(), qh.data()), b = oracle_ba_evaluate(l->ora, loss, &co, ro.data(), qo.data());
  if (FILE* f = log_file()) {
    double rmax = 0.0; for (double v : ro) rmax = std::max(rmax, std::fabs(v));
    double qmax = 0.0; for (double v : qo) qmax = std::max(qmax, std::fabs(v));
    std::fprintf(f, "{\"call\": \"evaluate\", \"loss\": %d, \"cost_rel\": %.3e, \"residual_rel\": %.3e, \"sqnorm_rel\": %.3e}\n", loss, rel(ch, co), max_abs_diff(rh, ro) / std::max(rmax, 1e-300), max_abs_diff(qh, qo) / std::max(qmax, 1e-300));
    std::fflush(f);
  }
  if (cost) *cost = ch;
  if (res) std::copy(rh.begin(), rh.end(), res);
  if (sq) std::copy(qh.begin(), qh.end(), sq);
  return both(a, b, "evaluate");
}

int lock_ba_solve(obvi_ba_handle* h, const obvi_solver_params* prm, obvi_summary* sum) {
  Lock* l = L_(h);
  obvi_summary so; std::memset(&so, 0, sizeof(so));
  const int a = obvi_ba_solve(l->hip, prm, sum), b = oracle_ba_solve(l->ora, prm, &so);
  std::vector<double> ph((size_t)6 * l->P), po(ph.size()), xh((size_t)3 * l->L), xo(xh.size()), oh((size_t)7 * l->O), oo(oh.size());
  obvi_ba_get_poses(l->hip, ph.data()); obvi_ba_get_points(l->hip, xh.data()); obvi_ba_get_objects(l->hip, oh.data());
  oracle_ba_get_poses(l->ora, po.data()); oracle_ba_get_points(l->ora, xo.data()); oracle_ba_get_objects(l->ora, oo.data());
  if (FILE* f = log_file()) {
    std::vector<obvi_iteration_summary> ih((size_t)std::max(1, sum->num_iterations)), io((size_t)std::max(1, so.num_iterations));
    const int nh = obvi_ba_get_iterations(l->hip, ih.data(), (int32_t)ih.size()), no = oracle_ba_get_iterations(l->ora, io.data(), (int32_t)io.size());
    int same_flags = nh == no ? 1 : 0;
    double it_cost_rel = 0.0;
    for (int i = 0; i < std::min(nh, no); ++i) { if (ih[i].step_is_successful != io[i].step_is_successful) same_flags = 0; it_cost_rel = std::max(it_cost_rel, rel(ih[i].cost, io[i].cost)); }
    std::fprintf(f, "{\"call\": \"solve\", \"poses\": %lld, \"points\": %lld, \"objects\": %lld, \"iterations_hip\": %d, \"iterations_oracle\": %d, \"termination_hip\": %d, \"termination_oracle\": %d, "
                    "\"same_accept_sequence\": %d, \"initial_cost\": %.17g, \"initial_cost_rel\": %.3e, \"final_cost_rel\": %.3e, \"max_iteration_cost_rel\": %.3e, \"pose_diff\": %.3e, \"point_diff\": %.3e, \"object_diff\": %.3e, \"object_diff_whitened\": %.3e, "
                    "\"params_reduced_equal\": %d, \"function_tolerance\": %.3e, \"max_num_iterations\": %d, \"message_hip\": \"%.40s\", \"message_oracle\": \"%.40s\"}\n",
                 (long long)l->P, (long long)l->L, (long long)l->O, sum->num_iterations, so.num_iterations, sum->termination_type, so.termination_type, same_flags, so.initial_cost,
                 rel(sum->initial_cost, so.initial_cost), rel(sum->final_cost, so.final_cost), it_cost_rel, max_abs_diff(ph, po), max_abs_diff(xh, xo), object_diff(oh, oo), whitened_object_diff(l->ora, l->O, oh, oo),
                 (sum->num_parameters_reduced == so.num_parameters_reduced && sum->num_residuals_reduced == so.num_residuals_reduced) ? 1 : 0,
                 prm->function_tolerance, (int)prm->max_num_iterations, sum->message, so.message);
    std::fflush(f);
  }
  if (arb().lib && l->arb) {
    // the same solve once more, with every solver-level sum in extended precision; then the arbiter, too, continues from the HIP result
    obvi_summary sa; std::memset(&sa, 0, sizeof(sa));
    arb().solve(l->arb, prm, &sa);
    std::vector<double> pa(ph.size()), xa(xh.size()), oa(oh.size());
    arb().get_poses(l->arb, pa.data()); arb().get_points(l->arb, xa.data()); arb().get_objects(l->arb, oa.data());
    if (FILE* f = log_file()) {
      std::vector<obvi_iteration_summary> ia((size_t)std::max(1, sa.num_iterations)), ih((size_t)std::max(1, sum->num_iterations)), io((size_t)std::max(1, so.num_iterations));
      const int na = arb().get_iterations(l->arb, ia.data(), (int32_t)ia.size());
      const int nh = obvi_ba_get_iterations(l->hip, ih.data(), (int32_t)ih.size()), no = oracle_ba_get_iterations(l->ora, io.data(), (int32_t)io.size());
      // how long each fp64 run follows the arbiter: first iteration whose accept flag differs, or whose cost is further than 1e-6 away
      auto follows = [&](const std::vector<obvi_iteration_summary>& x, int nx) {
        int k = 0;
        while (k < std::min(nx, na) && x[k].step_is_successful == ia[k].step_is_successful && rel(x[k].cost, ia[k].cost) <= 1e-6) ++k;
        return k;
      };
      std::fprintf(f, "{\"call\": \"arbiter\", \"iterations_arbiter\": %d, \"iterations_hip\": %d, \"iterations_oracle\": %d, \"termination_arbiter\": %d, \"hip_follows\": %d, \"oracle_follows\": %d, "
                      "\"final_cost_arbiter\": %.17g, \"hip_final_cost_rel\": %.3e, \"oracle_final_cost_rel\": %.3e, \"hip_pose_diff\": %.3e, \"oracle_pose_diff\": %.3e, \"hip_object_diff\": %.3e, \"oracle_object_diff\": %.3e}\n",
                   na, nh, no, sa.termination_type, follows(ih, nh), follows(io, no), sa.final_cost, rel(sum->final_cost, sa.final_cost), rel(so.final_cost, sa.final_cost),
                   max_abs_diff(ph, pa), max_abs_diff(po, pa), object_diff(oh, oa), object_diff(oo, oa));
      std::fflush(f);
    }
    arb().update_poses(l->arb, l->P, ph.data()); arb().update_points(l->arb, l->L, xh.data()); arb().update_objects(l->arb, l->O, oh.data());
  }
  // lock step: the oracle continues from the HIP path's result
  oracle_ba_update_poses(l->ora, l->P, ph.data()); oracle_ba_update_points(l->ora, l->L, xh.data()); oracle_ba_update_objects(l->ora, l->O, oh.data());
  return both(a, b, "solve");
}

int lock_ba_select_outliers(obvi_ba_handle* h, int32_t type, double fraction, uint8_t* mask, int64_t* nex) {
  Lock* l = L_(h);
  const int64_t n = obvi_ba_num_factors(l->hip, type);
  std::vector<uint8_t> mo((size_t)std::max<int64_t>(1, n));
  int64_t eh = 0, eo = 0;
  const int a = obvi_ba_select_outliers(l->hip, type, fraction, mask, &eh), b = oracle_ba_select_outliers(l->ora, type, fraction, mo.data(), &eo);
  int64_t differ = 0;
  for (int64_t i = 0; i < n; ++i) differ += mask[i] != mo[i];
  if (FILE* f = log_file()) { std::fprintf(f, "{\"call\": \"select_outliers\", \"type\": %d, \"factors\": %lld, \"excluded_hip\": %lld, \"excluded_oracle\": %lld, \"masks_differ\": %lld}\n", type, (long long)n, (long long)eh, (long long)eo, (long long)differ); std::fflush(f); }
  if (nex) *nex = eh;
  return both(a, b, "select_outliers");
}

int lock_ba_object_covariances(obvi_ba_handle* h, int64_t n, const uint32_t* a, const uint32_t* b, double* cov) {
  Lock* l = L_(h);
  std::vector<double> co((size_t)49 * std::max<int64_t>(1, n));
  const int rh = obvi_ba_object_covariances(l->hip, n, a, b, cov), ro = oracle_ba_object_covariances(l->ora, n, a, b, co.data());
  if (FILE* f = log_file()) {
    double worst = 0.0;
    for (int64_t i = 0; i < n && rh == 0 && ro == 0; ++i) {
      double scale = 0.0, d = 0.0;
      for (int k = 0; k < 49; ++k) { scale = std::max(scale, std::fabs(co[49 * i + k])); d = std::max(d, std::fabs(cov[49 * i + k] - co[49 * i + k])); }
      worst = std::max(worst, d / std::max(scale, 1e-300));
    }
    std::fprintf(f, "{\"call\": \"object_covariances\", \"pairs\": %lld, \"status_hip\": %d, \"status_oracle\": %d, \"block_rel\": %.3e}\n", (long long)n, rh, ro, worst);
    std::fflush(f);
  }
  return both(rh, ro, "object_covariances");
}

int lock_ba_column_sqnorms(obvi_ba_handle* h, double* p6, double* x3, double* o7) {
  Lock* l = L_(h);
  std::vector<double> po((size_t)6 * l->P + 1), xo((size_t)3 * l->L + 1), oo((size_t)7 * l->O + 1), ph(po.size()), xh(xo.size()), oh(oo.size());
  const int a = obvi_ba_column_sqnorms(l->hip, ph.data(), xh.data(), oh.data()), b = oracle_ba_column_sqnorms(l->ora, po.data(), xo.data(), oo.data());
  if (FILE* f = log_file()) {
    double worst = 0.0;
    auto cmp = [&](const std::vector<double>& x, const std::vector<double>& y) { for (size_t i = 0; i + 1 < x.size(); ++i) worst = std::max(worst, std::fabs(x[i] - y[i]) / std::max(1.0, std::fabs(y[i]))); };
    cmp(ph, po); cmp(xh, xo); cmp(oh, oo);
    std::fprintf(f, "{\"call\": \"column_sqnorms\", \"rel\": %.3e}\n", worst);
    std::fflush(f);
  }
  if (p6) std::copy(ph.begin(), ph.end() - 1, p6);
  if (x3) std::copy(xh.begin(), xh.end() - 1, x3);
  if (o7) std::copy(oh.begin(), oh.end() - 1, o7);
  return both(a, b, "column_sqnorms");
}
}  // extern "C"
