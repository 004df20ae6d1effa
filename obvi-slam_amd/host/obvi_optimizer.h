// obvi_optimizer.h -- host-side mirror of the reference's optimisation layer on top of the C ABI:
//   obvi::Problem                    takes the place of ceres::Problem in the signatures (thin: records
//                                    the flat problem of the last build and owns the device handle)
//   ObjectPoseGraphOptimizer         buildPoseGraphOptimization / solveOptimization / clearPastOptimizationData
//                                    (include/refactoring/optimization/object_pose_graph_optimizer.h:93-797)
//   runPgoPlusEllipsoids             (include/refactoring/optimization/pose_graph_plus_objects_optimizer.h:23-353)
//   OptimizationLogger               ceres_opt_summary.csv with the reference's columns
//                                    (include/debugging/optimization_logger.h:166-304)
// No evaluation happens on the host: everything numeric goes through include/obvi_ba.h.
#ifndef OBVI_HOST_OPTIMIZER_H_
#define OBVI_HOST_OPTIMIZER_H_

#include <obvi_ba.h>

#include <chrono>
#include <atomic>
#include <condition_variable>
#include <thread>
#include <cstdio>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <mutex>
#include <set>
#include <sstream>

#include "obvi_params.h"
#include "obvi_pose_graph.h"

namespace obvi {

typedef uint64_t ResidualBlockId;   // stands in for ceres::ResidualBlockId

struct IterationSummary { int iteration; double cost, cost_change, step_norm, gradient_max_norm; bool step_is_successful; };
// the fields of ceres::Solver::Summary the reference consumes (object_pose_graph_optimizer.h:676-706,
// optimization_logger.h:192-203)
struct SolverSummary {
  int termination_type = OBVI_FAILURE;
  bool usable = false;
  double initial_cost = 0, final_cost = 0, fixed_cost = 0;
  double total_time_in_seconds = 0, linear_solver_time_in_seconds = 0, jacobian_evaluation_time_in_seconds = 0, residual_evaluation_time_in_seconds = 0;
  int num_parameters_reduced = 0, num_residuals_reduced = 0;
  std::vector<IterationSummary> iterations;
  std::string message;
  bool IsSolutionUsable() const { return usable; }
  std::string FullReport() const {
    std::ostringstream o;
    o << "obvi_ba solve: " << message << "  iterations " << iterations.size() << "  cost " << initial_cost << " -> " << final_cost << "  time "
      << total_time_in_seconds << " s (linear solver " << linear_solver_time_in_seconds << ", jacobian " << jacobian_evaluation_time_in_seconds
      << ", residual " << residual_evaluation_time_in_seconds << ")";
    return o.str();
  }
};

// Flat problem in the layout of the C ABI plus the ids needed to scatter results back.
struct FlatProblem {
  std::vector<vslam_types_refactor::CameraId> cameras;                 // camera index -> CameraId
  std::vector<double> cam_K, cam_ext;
  std::vector<vslam_types_refactor::FrameId> frames;                   // pose index -> FrameId (ascending)
  std::vector<vslam_types_refactor::FeatureId> features;               // point index -> FeatureId (ascending)
  std::vector<vslam_types_refactor::ObjectId> objects;                 // object index -> ObjectId (ascending)
  std::vector<double*> pose_ptrs, point_ptrs, object_ptrs;             // the pose graph's parameter blocks
  std::vector<uint8_t> pose_const, point_const, object_const;
  std::vector<uint32_t> rp_pose, rp_point; std::vector<uint16_t> rp_cam; std::vector<double> rp_pixel, rp_sigma;
  std::vector<uint32_t> bb_obj, bb_pose; std::vector<uint16_t> bb_cam; std::vector<double> bb_corners, bb_cov;
  std::vector<uint32_t> sp_obj; std::vector<double> sp_mean, sp_cov;
  std::vector<uint32_t> lt_obj; std::vector<double> lt_mean, lt_cov;
  std::vector<uint32_t> rl_a, rl_b; std::vector<double> rl_t, rl_aa, rl_cov;
  double rl_huber = 1.0;
  // residual blocks in Problem::GetResidualBlocks order (= evaluate order of the ABI: types 0,2,3,4,5)
  std::vector<vslam_types_refactor::FactorInfo> blocks;
  // empty, capacity kept: a sliding window rebuilds a problem of the same size for every frame
  void reset() {
    cameras.clear(); cam_K.clear(); cam_ext.clear(); frames.clear(); features.clear(); objects.clear(); pose_ptrs.clear(); point_ptrs.clear(); object_ptrs.clear();
    pose_const.clear(); point_const.clear(); object_const.clear(); rp_pose.clear(); rp_point.clear(); rp_cam.clear(); rp_pixel.clear(); rp_sigma.clear();
    bb_obj.clear(); bb_pose.clear(); bb_cam.clear(); bb_corners.clear(); bb_cov.clear(); sp_obj.clear(); sp_mean.clear(); sp_cov.clear();
    lt_obj.clear(); lt_mean.clear(); lt_cov.clear(); rl_a.clear(); rl_b.clear(); rl_t.clear(); rl_aa.clear(); rl_cov.clear(); rl_huber = 1.0; blocks.clear();
  }
};

// Process-wide choices for every device handle the mirror creates (obvi_ba_options): which of the reference's two reprojection
// functors runs (0: ReprojectionCostFunctor, what residual_creator.h:251-264 instantiates; 1: ReprojectionCostFunctorAnalyticJacobian)
// and whether the device sums run in a fixed order (bit-identical reruns, as Ceres at a fixed num_threads).
struct BackendOptions { int reprojection_variant = OBVI_REPROJECTION_AUTODIFF; bool deterministic = false; };
inline BackendOptions& backendOptions() { static BackendOptions o; return o; }
inline obvi_ba_options makeHandleOptions(int device_id) {
  obvi_ba_options opt{};
  opt.device_id = device_id; opt.object_block_size = 7;
  opt.reprojection_variant = backendOptions().reprojection_variant; opt.deterministic = backendOptions().deterministic ? 1 : 0;
  return opt;
}

// Device handles outlive the Problem that used them: creating one allocates its streams, events, pinned pages and worker pool (about
// 25 ms, 10 ms to destroy) and a session makes a new Problem at every global-BA frame (runPgoPlusEllipsoids).  A handle a Problem gives
// back is reset (obvi_ba_reset) and handed to the next Problem with the same device and options.  drain() destroys what is parked (the drivers call it before they exit).
class HandlePool {
 public:
  static HandlePool& instance() { static HandlePool* p = new HandlePool; return *p; }   // never destroyed: no HIP calls from static destructors
  obvi_ba_handle* acquire(const obvi_ba_options& opt) {
    {
      std::unique_lock<std::mutex> lock(mu_);
      for (;;) {
        for (size_t i = 0; i < parked_.size(); ++i)
          if (same(parked_[i].first, opt)) { obvi_ba_handle* h = parked_[i].second; parked_.erase(parked_.begin() + (long)i); return h; }
        if (warming_ == 0) break;
        warm_cv_.wait(lock);   // a handle that is being created in the background (warm) is worth waiting for
      }
    }
    obvi_ba_handle* h = nullptr;
    const int rc = obvi_ba_create(&opt, &h);
    if (rc != OBVI_OK) { std::cerr << "obvi_ba_create failed: status " << rc << " (no HIP device? there is no CPU path)" << std::endl; return nullptr; }
    return h;
  }
  void release(const obvi_ba_options& opt, obvi_ba_handle* h) {
    if (h == nullptr) return;
    // as obvi_ba_create left it: the next user may set fewer factor families than this one did, share nothing, install no hook
    const int rc = obvi_ba_reset(h);
    if (rc != OBVI_OK) { obvi_ba_destroy(h); return; }
    std::lock_guard<std::mutex> lock(mu_);
    if (parked_.size() >= kMaxParked) { obvi_ba_destroy(h); return; }
    parked_.push_back({opt, h});
  }
  // Creates a handle on a thread of its own and parks it: a host that knows it will optimise calls this first thing, so that the start of
  // the HIP runtime and the handle's allocations (90-130 ms in a fresh process) run beside its own start-up work -- reading the scene,
  // filling the pose graph -- instead of in front of the first optimisation.  acquire() waits for a warm-up in flight.
  void warm(const obvi_ba_options& opt) {
    { std::lock_guard<std::mutex> lock(mu_); ++warming_; }
    std::thread([this, opt] {
      obvi_ba_handle* h = nullptr;
      const int rc = obvi_ba_create(&opt, &h);
      std::lock_guard<std::mutex> lock(mu_);
      if (rc == OBVI_OK && h != nullptr) parked_.push_back({opt, h});
      --warming_;
      warm_cv_.notify_all();
    }).detach();
  }
  void drain() {
    { std::unique_lock<std::mutex> lock(mu_); warm_cv_.wait(lock, [&] { return warming_ == 0; }); }
    std::vector<std::pair<obvi_ba_options, obvi_ba_handle*>> all;
    { std::lock_guard<std::mutex> lock(mu_); all.swap(parked_); }
    for (auto& e : all) obvi_ba_destroy(e.second);
  }
 private:
  static constexpr size_t kMaxParked = 8;
  static bool same(const obvi_ba_options& a, const obvi_ba_options& b) {
    return a.device_id == b.device_id && a.object_block_size == b.object_block_size && a.reprojection_variant == b.reprojection_variant && a.deterministic == b.deterministic;
  }
  std::mutex mu_;
  std::condition_variable warm_cv_;
  int warming_ = 0;   // handles being created by warm(), guarded by mu_
  std::vector<std::pair<obvi_ba_options, obvi_ba_handle*>> parked_;
};

class Problem {
 public:
  explicit Problem(int device_id = 0, bool dry_run = false) : device_id_(device_id), dry_run_(dry_run) {}
  Problem(const Problem&) = delete;
  Problem& operator=(const Problem&) = delete;
  ~Problem() { if (h_) HandlePool::instance().release(opt_, h_); }
  // residual blocks added outside the factor store: runPgoPlusEllipsoids adds RelativePoseFactor blocks directly
  // to its ceres::Problem (pose_graph_plus_objects_optimizer.h:129-159); they stay for every later build on it.
  void AddRelativePoseResidualBlock(const vslam_types_refactor::RelPoseFactor& f, double huber) { extra_relpose_.push_back(f); extra_relpose_huber_ = huber; }
  const std::vector<vslam_types_refactor::RelPoseFactor>& extraRelativePoseBlocks() const { return extra_relpose_; }
  double extraRelativePoseHuber() const { return extra_relpose_huber_; }
  FlatProblem flat;
  bool dryRun() const { return dry_run_; }
  obvi_ba_handle* handle() {
    if (!h_ && !dry_run_) { opt_ = makeHandleOptions(device_id_); h_ = HandlePool::instance().acquire(opt_); }
    return h_;
  }
  // Hands the device handle to the pool until the next handle() call: a stage that runs on a Problem of its own in between
  // (runPgoPlusEllipsoids at a global-BA frame) then works on this handle -- its streams, pinned pages and grown allocations -- instead
  // of creating a second one.  Whatever was on the device is gone; the next build uploads everything anyway.
  void parkHandle() { if (h_) { HandlePool::instance().release(opt_, h_); h_ = nullptr; uploaded_ahead_ = false; } }
  // set by ObjectPoseGraphOptimizer::uploadAndPlanAhead: `flat` is on the device with its symbolic plan; the next solveOptimization hands over values only
  bool uploadedAhead() const { return uploaded_ahead_; }
  void setUploadedAhead(bool v) { uploaded_ahead_ = v; }
 private:
  bool uploaded_ahead_ = false;
  int device_id_; bool dry_run_;
  obvi_ba_handle* h_ = nullptr;
  obvi_ba_options opt_{};
  std::vector<vslam_types_refactor::RelPoseFactor> extra_relpose_;
  double extra_relpose_huber_ = 1.0;
};

// ceres::Covariance as the long-term-map extraction uses it (src/refactoring/long_term_map/long_term_object_map_extraction.cpp:419-433,
// include/refactoring/long_term_map/long_term_object_map_extraction.h:318-340, 499-513): Compute() for a list of pairs of
// ellipsoid blocks of the problem that is on the device, then GetCovarianceBlock() per pair.  The parameter values are
// taken from the pose graph's blocks (as Ceres reads them in place), the factors are those of the last build.
class Covariance {
 public:
  bool Compute(const std::vector<std::pair<vslam_types_refactor::ObjectId, vslam_types_refactor::ObjectId>>& covariance_blocks, Problem* problem) {
    blocks_.clear();
    if (problem == nullptr) return false;
    obvi_ba_handle* h = problem->handle();
    if (h == nullptr) return false;
    const FlatProblem& fp = problem->flat;
    auto index_of = [&](vslam_types_refactor::ObjectId id, uint32_t* out) {
      const auto it = std::lower_bound(fp.objects.begin(), fp.objects.end(), id);
      if (it == fp.objects.end() || *it != id) return false;
      *out = (uint32_t)(it - fp.objects.begin());
      return true;
    };
    std::vector<uint32_t> a(covariance_blocks.size()), b(covariance_blocks.size());
    for (size_t i = 0; i < covariance_blocks.size(); ++i)
      if (!index_of(covariance_blocks[i].first, &a[i]) || !index_of(covariance_blocks[i].second, &b[i])) { std::cerr << "Covariance::Compute: object is not a parameter block of the problem" << std::endl; return false; }
    auto gather = [](const std::vector<double*>& ptrs, int dim) { std::vector<double> v(ptrs.size() * dim); for (size_t i = 0; i < ptrs.size(); ++i) std::copy_n(ptrs[i], dim, &v[dim * i]); return v; };
    const std::vector<double> poses = gather(fp.pose_ptrs, 6), points = gather(fp.point_ptrs, 3), objects = gather(fp.object_ptrs, 7);
    int rc = obvi_ba_set_poses(h, (int64_t)fp.frames.size(), poses.data(), fp.pose_const.data());
    if (!rc) rc = obvi_ba_set_points(h, (int64_t)fp.features.size(), points.data(), fp.point_const.data());
    if (!rc) rc = obvi_ba_set_objects(h, (int64_t)fp.objects.size(), objects.data(), fp.object_const.data());
    std::vector<double> cov(49 * covariance_blocks.size());
    if (!rc) rc = obvi_ba_object_covariances(h, (int64_t)covariance_blocks.size(), a.data(), b.data(), cov.data());
    if (rc) { std::cerr << "Covariance computation failed: " << obvi_ba_last_error(h) << std::endl; return false; }      // LOG(WARNING) at :435-437
    for (size_t i = 0; i < covariance_blocks.size(); ++i) blocks_[covariance_blocks[i]] = std::vector<double>(cov.begin() + 49 * i, cov.begin() + 49 * (i + 1));
    return true;
  }
  // 7x7, row-major (Ceres' convention)
  bool GetCovarianceBlock(vslam_types_refactor::ObjectId a, vslam_types_refactor::ObjectId b, double* covariance_block) const {
    const auto it = blocks_.find({a, b});
    if (it == blocks_.end()) return false;
    std::copy(it->second.begin(), it->second.end(), covariance_block);
    return true;
  }
 private:
  std::map<std::pair<vslam_types_refactor::ObjectId, vslam_types_refactor::ObjectId>, std::vector<double>> blocks_;
};


// ---- covariance extraction with rank-deficiency handling (long_term_object_map_extraction.cpp:507-760, 764-927, 929-1062) --------
// When Covariance::Compute fails (rank-deficient Jacobian), the reference finds the Jacobian columns with the smallest squared
// norms -- (rank deficiency + kRankDeficiencyColsBuffer) of them -- and gives each of those parameters a ParameterPrior
// (parameter_prior.h:17-50) centred on its current estimate with std_dev = 1 / sqrt(n* - n_col), n* the smallest norm NOT selected:
// every selected column is lifted to n*.  Then it retries, up to kMaxJacobianExtractionRetries times.
struct CovarianceRankRepair { int block_kind; uint64_t block_id; int param_idx; double col_sqnorm, prior_std_dev; int retry; };   // kind 0 frame, 1 feature, 2 object
constexpr int kMaxJacobianExtractionRetries = 5, kRankDeficiencyColsBuffer = 50;   // long_term_object_map_extraction.h:20-21
// findRankDeficiencies (:610-660) + the prior parameters of addPriorToProblemParams (:764-927) on the column norms as the device
// reports them (obvi_ba_column_sqnorms; -1 = not a parameter of the problem).  The reference takes the rank deficiency from a sparse
// QR of the Jacobian; here it is the number of columns below min_col_norm (at least one): with the buffer of 50 the selection is
// dominated by the buffer either way.  Pure function: CPU-testable.
struct ColumnNormView { const std::vector<double>* sqnorms; int block_dim; int block_kind; const std::vector<uint64_t>* ids; };
inline std::vector<CovarianceRankRepair> selectParameterPriors(const std::vector<ColumnNormView>& views, double min_col_norm, int retry, double* min_non_prob_col_norm_out = nullptr) {
  struct Col { double n; int view; size_t block; int param; };
  std::vector<Col> cols;
  for (size_t v = 0; v < views.size(); ++v)
    for (size_t i = 0; i < views[v].sqnorms->size(); ++i) {
      const double n = (*views[v].sqnorms)[i];
      if (n >= 0.0) cols.push_back({n, (int)v, i / views[v].block_dim, (int)(i % views[v].block_dim)});
    }
  std::vector<CovarianceRankRepair> out;
  if (cols.empty()) return out;
  size_t rank_deficiency = 0;
  for (const Col& c : cols) if (c.n < min_col_norm) ++rank_deficiency;
  rank_deficiency = std::max<size_t>(rank_deficiency, 1);
  const size_t count = std::min(rank_deficiency + (size_t)kRankDeficiencyColsBuffer + 1, cols.size());   // :623-624
  std::partial_sort(cols.begin(), cols.begin() + count, cols.end(), [](const Col& a, const Col& b) { return a.n < b.n; });
  const double min_non_prob = cols[count - 1].n;                                                           // smallest_n.back() :656
  if (min_non_prob_col_norm_out) *min_non_prob_col_norm_out = min_non_prob;
  for (size_t k = 0; k + 1 < count; ++k) {
    const Col& c = cols[k];
    const double gap = min_non_prob - c.n;
    if (!(gap > 0.0)) continue;   // equal to the first non-problem column: 1 / sqrt(0) in the reference; nothing to lift
    out.push_back({views[c.view].block_kind, (*views[c.view].ids)[c.block], c.param, c.n, 1.0 / std::sqrt(gap), retry});
  }
  return out;
}

// extractCovarianceWithRankDeficiencyHandling (:929-1062) for the object blocks of the problem on the device
inline bool extractCovarianceWithRankDeficiencyHandling(const std::vector<std::pair<vslam_types_refactor::ObjectId, vslam_types_refactor::ObjectId>>& covariance_blocks, Problem* problem,
                                                        double min_col_norm, Covariance* covariance, std::vector<CovarianceRankRepair>* repairs) {
  obvi_ba_handle* h = problem ? problem->handle() : nullptr;
  if (h == nullptr) return false;
  const FlatProblem& fp = problem->flat;
  std::vector<uint8_t> kind, param; std::vector<uint32_t> block; std::vector<double> mean, stddev;
  obvi_ba_set_parameter_priors(h, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
  bool ok = covariance->Compute(covariance_blocks, problem);
  for (int retry_count = 1; !ok && retry_count <= kMaxJacobianExtractionRetries; ++retry_count) {
    std::cerr << "Retrying rank deficient jacobian, retry num " << retry_count << std::endl;
    std::vector<double> np(6 * fp.frames.size()), nl(3 * fp.features.size()), no(7 * fp.objects.size());
    if (obvi_ba_column_sqnorms(h, np.data(), nl.data(), no.data())) { std::cerr << "column norms failed: " << obvi_ba_last_error(h) << std::endl; return false; }
    std::vector<uint64_t> frames(fp.frames.begin(), fp.frames.end()), feats(fp.features.begin(), fp.features.end()), objs(fp.objects.begin(), fp.objects.end());
    const std::vector<CovarianceRankRepair> sel = selectParameterPriors({{&nl, 3, 1, &feats}, {&np, 6, 0, &frames}, {&no, 7, 2, &objs}}, min_col_norm, retry_count);
    if (sel.empty()) { std::cerr << "No rank deficient columns identified" << std::endl; break; }
    for (const CovarianceRankRepair& r : sel) {
      // the flat problem's lists are in ascending id order
      const std::vector<uint64_t>& ids = r.block_kind == 0 ? frames : r.block_kind == 1 ? feats : objs;
      const uint32_t idx = (uint32_t)(std::lower_bound(ids.begin(), ids.end(), r.block_id) - ids.begin());
      const double* values = r.block_kind == 0 ? fp.pose_ptrs[idx] : r.block_kind == 1 ? fp.point_ptrs[idx] : fp.object_ptrs[idx];
      kind.push_back((uint8_t)(r.block_kind == 0 ? 0 : r.block_kind == 1 ? 1 : 2)); block.push_back(idx); param.push_back((uint8_t)r.param_idx);
      mean.push_back(values[r.param_idx]); stddev.push_back(r.prior_std_dev);
      if (repairs) repairs->push_back(r);
    }
    if (obvi_ba_set_parameter_priors(h, (int64_t)kind.size(), kind.data(), block.data(), param.data(), mean.data(), stddev.data())) { std::cerr << "parameter priors rejected: " << obvi_ba_last_error(h) << std::endl; return false; }
    ok = covariance->Compute(covariance_blocks, problem);
    if (!ok) std::cerr << "Covariance extraction failed on retry " << retry_count << " with additional priors; consider revising buffer " << kRankDeficiencyColsBuffer << std::endl;
  }
  obvi_ba_set_parameter_priors(h, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
  return ok;
}

}  // namespace obvi

namespace vslam_types_refactor {

// include/debugging/optimization_logger.h:151-304
// IterationLogger / IterationLoggerFactory (include/debugging/optimization_logger.h:29-147): one CSV per optimisation type,
// "ceres_iterations_<type>.csv", a row per LM iteration -- the reference's own definition of what an iteration is.
class IterationLogger {
 public:
  IterationLogger(const std::string& logging_directory, const std::string& logging_type)
      : output_file_path_((logging_directory.empty() || logging_directory.back() == '/' ? logging_directory : logging_directory + "/") + "ceres_iterations_" + logging_type + ".csv") {
    std::ofstream csv_file(output_file_path_, std::ios::trunc);
    csv_file << "optimization_id,iteration_num,cost,cost_change,step_norm,step_norm_per_param,is_successful\n";
  }
  void logIterations(const std::string& optimization_identifier, const obvi::SolverSummary& solver_summary) {
    iteration_info_.push_back({optimization_identifier, solver_summary.num_parameters_reduced, solver_summary.iterations});
  }
  void writeLatestData() {
    if (iteration_info_.empty()) return;
    std::ofstream csv_file(output_file_path_, std::ios::app);
    for (const Entry& e : iteration_info_)
      for (const obvi::IterationSummary& it : e.iterations)
        csv_file << e.id << "," << std::to_string(it.iteration) << "," << std::to_string(it.cost) << "," << std::to_string(it.cost_change) << "," << std::to_string(it.step_norm) << ","
                 << std::to_string(it.step_norm / e.num_parameters_reduced) << "," << (it.step_is_successful ? 1 : 0) << "\n";
    iteration_info_.clear();
  }
  const std::string& path() const { return output_file_path_; }

 private:
  struct Entry { std::string id; int num_parameters_reduced; std::vector<obvi::IterationSummary> iterations; };
  std::string output_file_path_;
  std::vector<Entry> iteration_info_;
};
class IterationLoggerFactory {
 public:
  inline const static std::string kPendingEstimatorOptimizationType = "pending_obj_est";
  inline const static std::string kVfAdjustOptimizationType = "vf_adjust";
  inline const static std::string kPrePgoTrackOptimizationType = "pre_pgo_track";
  inline const static std::string kPGOOptimizationType = "pgo";
  inline const static std::string kLBAPhase1OptimizationType = "lba_phase_1";
  inline const static std::string kLBAPhase2OptimizationType = "lba_phase_2";
  inline const static std::string kGBAPhase1OptimizationType = "gba_phase_1";
  inline const static std::string kGBAPhase2OptimizationType = "gba_phase_2";
  static IterationLoggerFactory& getInstance() { static IterationLoggerFactory factory_instance; return factory_instance; }
  static void setLoggingDirectory(const std::string& logging_directory) {
    IterationLoggerFactory& f = getInstance();
    f.logging_directory_ = logging_directory; f.initialized_ = true; f.iteration_loggers_by_type_.clear();
  }
  std::shared_ptr<IterationLogger> getOrCreateLoggerOfType(const std::string& logger_type) {
    if (!initialized_) return nullptr;   // the reference logs "Not initialized with target directory" and returns null: callers skip
    auto it = iteration_loggers_by_type_.find(logger_type);
    if (it == iteration_loggers_by_type_.end()) it = iteration_loggers_by_type_.emplace(logger_type, std::make_shared<IterationLogger>(logging_directory_, logger_type)).first;
    return it->second;
  }
  void writeAllIterationLoggerStates() { for (const auto& l : iteration_loggers_by_type_) if (l.second) l.second->writeLatestData(); }

 private:
  IterationLoggerFactory() = default;
  bool initialized_ = false;
  std::string logging_directory_;
  std::unordered_map<std::string, std::shared_ptr<IterationLogger>> iteration_loggers_by_type_;
};

class OptimizationLogger {
 public:
  explicit OptimizationLogger(const std::string& output_file_path) : output_file_path_(output_file_path) {}
  void setOptimizationTypeParams(const FrameId& max_frame_id, const bool& global_ba, const bool& global_pgo, const bool& outliers_excluded, const size_t& attempt_num = 0) {
    info_.max_frame_id_ = max_frame_id; info_.global_ba_ = global_ba; info_.global_pgo_ = global_pgo; info_.local_ba_ = !global_ba;
    info_.outliers_excluded_opt_ = outliers_excluded; info_.attempt_num_ = attempt_num;
  }
  void setOptimizationParams(size_t num_objects, size_t num_features, size_t num_frames) { info_.num_objects_ = num_objects; info_.num_visual_features_ = num_features; info_.num_poses_ = num_frames; }
  void extractOptimizationTimingResults(const obvi::SolverSummary& s) {
    info_.total_ceres_time_ = s.total_time_in_seconds; info_.linear_solver_time_ = s.linear_solver_time_in_seconds;
    info_.jacobian_time_ = s.jacobian_evaluation_time_in_seconds; info_.residual_time_ = s.residual_evaluation_time_in_seconds;
    info_.num_ceres_iterations_ = s.iterations.size();
    // :205-236 the per-iteration rows go to the logger of the optimisation's type, identified by "<max frame>_<attempt>"
    std::string opt_type;
    if (info_.global_pgo_) opt_type = IterationLoggerFactory::kPGOOptimizationType;
    else if (info_.global_ba_) opt_type = info_.outliers_excluded_opt_ ? IterationLoggerFactory::kGBAPhase2OptimizationType : IterationLoggerFactory::kGBAPhase1OptimizationType;
    else if (info_.local_ba_) opt_type = info_.outliers_excluded_opt_ ? IterationLoggerFactory::kLBAPhase2OptimizationType : IterationLoggerFactory::kLBAPhase1OptimizationType;
    else return;
    const std::shared_ptr<IterationLogger> iteration_logger = IterationLoggerFactory::getInstance().getOrCreateLoggerOfType(opt_type);
    if (iteration_logger != nullptr) iteration_logger->logIterations(std::to_string(info_.max_frame_id_) + "_" + std::to_string(info_.attempt_num_), s);
  }
  void writeOptInfoHeader() {
    if (output_file_path_.empty()) return;
    std::ofstream f(output_file_path_, std::ios::trunc);
    f << "max_frame_id,outliers_excluded?,local_ba?,global_ba?,global_pgo?,num_poses,num_objects,num_visual_features,total_ceres_time,"
         "linear_solver_time,jacobian_time,residual_time,num_ceres_iterations\n";
  }
  void writeCurrentOptInfo() {
    if (!output_file_path_.empty()) {
      std::ofstream f(output_file_path_, std::ios::app);
      f << info_.max_frame_id_ << "," << (info_.outliers_excluded_opt_ ? 1 : 0) << "," << (info_.local_ba_ ? 1 : 0) << "," << (info_.global_ba_ ? 1 : 0) << ","
        << (info_.global_pgo_ ? 1 : 0) << "," << info_.num_poses_ << "," << info_.num_objects_ << "," << info_.num_visual_features_ << ","
        << std::to_string(info_.total_ceres_time_) << "," << std::to_string(info_.linear_solver_time_) << "," << std::to_string(info_.jacobian_time_) << ","
        << std::to_string(info_.residual_time_) << "," << info_.num_ceres_iterations_ << "\n";
    }
    info_ = Info();
  }
 private:
  struct Info {
    FrameId max_frame_id_ = 0; bool outliers_excluded_opt_ = false, local_ba_ = false, global_ba_ = false, global_pgo_ = false;
    size_t num_poses_ = 0, num_objects_ = 0, num_visual_features_ = 0, num_ceres_iterations_ = 0, attempt_num_ = 0;
    double total_ceres_time_ = 0, linear_solver_time_ = 0, jacobian_time_ = 0, residual_time_ = 0;
  };
  std::string output_file_path_;
  Info info_;
};

}  // namespace vslam_types_refactor

namespace pose_graph_optimizer {
using namespace vslam_types_refactor;   // NOLINT (the reference's optimiser header does the same through its includes)
typedef ObjectAndReprojectionFeaturePoseGraph PoseGraphType;

// One persistent host thread for work that runs BESIDE a solve (solveOptimization's `beside` hooks): post() hands it a job, wait() returns when the job has
// ended.  Between jobs the thread spins for a while before it sleeps (OBVI_HOST_BESIDE_SPIN_US, default 4000): in a session a job arrives every few
// milliseconds, and a thread that slept wakes up on a core that has gone idle -- measured on the 300-frame session, a thread created per job ran the same work
// (frame data, window build, upload, symbolic phase) in 3.3 ms instead of 1.9.
class BesideThread {
 public:
  BesideThread() = default;
  BesideThread(const BesideThread&) = delete;
  BesideThread& operator=(const BesideThread&) = delete;
  ~BesideThread() {
    if (!thread_.joinable()) return;
    { std::lock_guard<std::mutex> lock(m_); stop_ = true; state_.store(kPosted, std::memory_order_release); }
    cv_.notify_all();
    thread_.join();
  }
  void post(const std::function<void()>& job) {
    if (!thread_.joinable()) thread_ = std::thread([this] { loop(); });
    { std::lock_guard<std::mutex> lock(m_); job_ = job; state_.store(kPosted, std::memory_order_release); }
    cv_.notify_all();
  }
  // returns when the job has ended; an exception the job threw is rethrown HERE, on the caller's thread (the same exception would have propagated
  // on the caller's thread had the work run serially), never std::terminate on the second thread.  Spins briefly, then sleeps: K sessions in one
  // process on few CPUs must not burn a core each.
  void wait() {
    const auto t0 = std::chrono::steady_clock::now();
    while (state_.load(std::memory_order_acquire) != kDone) {
      if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() < 200) { std::this_thread::yield(); continue; }
      std::unique_lock<std::mutex> lock(m_);
      done_cv_.wait_for(lock, std::chrono::milliseconds(2), [&] { return state_.load(std::memory_order_acquire) == kDone; });
    }
    state_.store(kIdle, std::memory_order_release);
    if (error_) { std::exception_ptr e = error_; error_ = nullptr; std::rethrow_exception(e); }
  }
 private:
  enum { kIdle = 0, kPosted = 1, kDone = 2 };
  void loop() {
    static const long spin_us = std::getenv("OBVI_HOST_BESIDE_SPIN_US") ? std::atol(std::getenv("OBVI_HOST_BESIDE_SPIN_US")) : 4000;
    for (;;) {
      const auto t0 = std::chrono::steady_clock::now();
      while (state_.load(std::memory_order_acquire) != kPosted && std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() < spin_us) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
      }
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> lock(m_);
        cv_.wait(lock, [&] { return state_.load(std::memory_order_acquire) == kPosted; });
        if (stop_) return;
        job.swap(job_);
      }
      try { job(); } catch (...) { error_ = std::current_exception(); }   // kDone is ALWAYS reached: wait() can never spin on a job that died
      { std::lock_guard<std::mutex> lock(m_); state_.store(kDone, std::memory_order_release); }
      done_cv_.notify_all();
    }
  }
  std::thread thread_;
  std::mutex m_;
  std::condition_variable cv_, done_cv_;
  std::exception_ptr error_;
  std::atomic<int> state_{kIdle};
  std::function<void()> job_;
  bool stop_ = false;
};

// solveOptimization's hooks around its wait for the device (see there)
struct BesideSolve { std::function<void()> start, join; };

// What buildPoseGraphOptimization returns: the reference hands back std::unordered_map<ceres::ResidualBlockId, FactorInfo>
// (object_pose_graph_optimizer.h:126, :631).  Residual block ids here are the positions in the flat problem's block list, so the map is
// that list; building a hash map of tens of thousands of entries per window cost as much as selecting the factors.
class ResidualBlockInfoMap {
 public:
  ResidualBlockInfoMap() = default;
  explicit ResidualBlockInfoMap(const std::vector<FactorInfo>& blocks) : blocks_(blocks) {}
  size_t size() const { return blocks_.size(); }
  bool empty() const { return blocks_.empty(); }
  size_t count(const obvi::ResidualBlockId& id) const { return (size_t)id < blocks_.size() ? 1 : 0; }
  const FactorInfo& at(const obvi::ResidualBlockId& id) const { if ((size_t)id >= blocks_.size()) throw std::out_of_range("ResidualBlockInfoMap::at"); return blocks_[(size_t)id]; }
  const std::vector<FactorInfo>& blocks() const { return blocks_; }   // position = residual block id
 private:
  std::vector<FactorInfo> blocks_;
};

// The reference's per-factor seam (object_pose_graph_optimizer.h:98-113, used at :1016-1052), type-erased: what the reference-shaped runner (obvi_runner.h) forwards a
// caller's refresh_residual_checker / residual_creator through.  For every residual block a build selects, in Problem::GetResidualBlocks order:
//   keep(factor, pose_graph)                 set and true  -> the caller holds a residual for it from an earlier build and does not want it refreshed: it stays
//                                                            (the reference keeps the existing ceres block and does not call the creator, :1018-1032)
//   create(factor, params, pose_graph, problem, id)  false -> "Could not make residual": the factor is LEFT OUT of the problem, as the reference leaves it out
//                                                            (:1042-1051; the block count and ids the caller sees are those of the remaining blocks)
// The numeric residual itself is not the creator's to define here -- the factors are the five of the path, evaluated on the device --, and the loss is the family's
// (one Huber parameter per factor type, as ObjectVisualPoseGraphResidualParams has it): what a creator decides is WHICH factors enter.
struct FactorHooks {
  typedef std::pair<vslam_types_refactor::FactorType, vslam_types_refactor::FeatureFactorId> Key;
  std::function<bool(const Key&, const std::shared_ptr<PoseGraphType>&)> keep;
  std::function<bool(const Key&, const pose_graph_optimization::ObjectVisualPoseGraphResidualParams&, const std::shared_ptr<PoseGraphType>&, obvi::Problem*, obvi::ResidualBlockId&)> create;
  explicit operator bool() const { return (bool)create; }
};

class ObjectPoseGraphOptimizer {
 public:
  ObjectPoseGraphOptimizer() = default;
  void setFactorHooks(const FactorHooks& hooks) { factor_hooks_ = hooks; }
  const FactorHooks& factorHooks() const { return factor_hooks_; }
  size_t factorsLeftOutByTheCreator() const { return n_vetoed_total_; }

  // object_pose_graph_optimizer.h:126-632.  Same selection rules; instead of adding Ceres residual and
  // parameter blocks the selected factors are flattened into problem->flat.
  ResidualBlockInfoMap buildPoseGraphOptimization(
      const OptimizationScopeParams& optimization_scope, const pose_graph_optimization::ObjectVisualPoseGraphResidualParams& residual_params,
      std::shared_ptr<PoseGraphType>& pose_graph, obvi::Problem* problem, std::optional<OptimizationLogger>& opt_logger,
      const FactorInfoSet& excluded_feature_factor_types_and_ids = {}) {
    residual_params_ = residual_params;
    const bool timing_ = std::getenv("OBVI_HOST_TIMING") != nullptr && std::getenv("OBVI_HOST_TIMING")[0] == '2';
    auto t_last_ = std::chrono::steady_clock::now();
    auto lap_ = [&](const char* what) {
      if (!timing_) return;
      const auto now = std::chrono::steady_clock::now();
      std::cerr << "    build: " << what << " " << std::chrono::duration<double, std::milli>(now - t_last_).count() << " ms" << std::endl;
      t_last_ = now;
    };
    std::set<FrameId> optimized_frames;
    std::unordered_set<ObjectId> ltm_object_ids;
    std::map<ObjectId, FactorInfoSet> objects_to_include;
    std::map<FactorType, std::set<FeatureFactorId>> required_feature_factors;

    const bool use_object_only_factors = optimization_scope.include_object_factors_ && !optimization_scope.fix_objects_;      // :165-170
    const bool use_feature_pose_factors = optimization_scope.include_visual_factors_;
    const bool use_relative_pose_factors = optimization_scope.min_low_level_feature_observations_per_frame_ > 0 && use_feature_pose_factors;
    const bool use_object_pose_factors = optimization_scope.include_object_factors_;
    const bool use_object_param_blocks = optimization_scope.include_object_factors_;
    const bool fix_object_param_blocks = optimization_scope.fix_objects_;
    const bool fix_ltm_param_blocks = optimization_scope.fix_objects_ || optimization_scope.fix_ltm_objects_;
    const bool fix_visual_feature_param_blocks = optimization_scope.fix_visual_features_;
    const bool fix_pose_param_blocks = optimization_scope.fix_poses_;

    for (const FrameId& f : pose_graph->getFrameIds())                                                                       // :196-203
      if (f >= optimization_scope.min_frame_id_ && f <= optimization_scope.max_frame_id_) optimized_frames.insert(f);

    auto excluded = [&](const FactorInfo& fi) {                                                                              // :886-905
      if (optimization_scope.factor_types_to_exclude.count(fi.first)) return true;
      if (excluded_feature_factor_types_and_ids.count(fi) && fi.first != kLongTermMapFactorTypeId && fi.first != kShapeDimPriorFactorTypeId) return true;
      return false;
    };
    // Visual factors (:205-238): the reference collects the window's factor ids in a set, groups them per feature in a map of sets,
    // drops the features with too few sightings (:826-861) and inserts the rest into the set of required factors.  The same selection
    // on flat arrays (a window holds tens of thousands of sightings and is rebuilt for every frame): the pose graph's per-frame factor
    // records, a sighting count per feature slot; the required factors are then flattened straight from the records (below), and the
    // frames that need a relative-pose factor (:240-299) are known once that pass has counted the sightings each frame keeps.
    typedef PoseGraphType::VisualFactorRecord VisualRecord;
    std::vector<std::pair<const VisualRecord*, size_t>> spans;   // the window's records, frame by frame
    std::vector<uint32_t> included_slots;            // feature slots of the included features, ascending feature id
    uint32_t min_obs = 0;
    const bool any_excluded = !excluded_feature_factor_types_and_ids.empty();
    auto visual_excluded = [&](const VisualRecord& r) { return any_excluded && excluded_feature_factor_types_and_ids.count({kReprojectionErrorFactorTypeId, r.id}) != 0; };
    if (use_feature_pose_factors) {
      min_obs = (uint32_t)optimization_scope.min_low_level_feature_observations_;
      const uint32_t first_count = std::max<uint32_t>(min_obs, 1);
      sightings_.assign(pose_graph->numFeatureSlots(), 0);
      if (optimization_scope.factor_types_to_exclude.count(kReprojectionErrorFactorTypeId) == 0)
        pose_graph->forEachVisualRecordSpanBetweenFrameIdsInclusive(optimization_scope.min_frame_id_, optimization_scope.max_frame_id_, [&](const VisualRecord* r, size_t n) {
          spans.emplace_back(r, n);
          for (const VisualRecord* e = r + n; r != e; ++r) {
            if (visual_excluded(*r)) continue;
            if (++sightings_[r->feature_slot] == first_count) included_slots.push_back(r->feature_slot);
          }
        });
      // ascending feature id (the keys beside the slots: a sort through featureIdOfSlot is twice as slow)
      std::vector<std::pair<FeatureId, uint32_t>>& keyed = scratch_keyed_;
      keyed.clear();
      for (const uint32_t slot : included_slots) keyed.emplace_back(pose_graph->featureIdOfSlot(slot), slot);
      if (!std::is_sorted(keyed.begin(), keyed.end())) std::sort(keyed.begin(), keyed.end());
      for (size_t i = 0; i < keyed.size(); ++i) included_slots[i] = keyed[i].second;
    }
    lap_("sightings per feature, included features");
    if (use_object_param_blocks) pose_graph->getLongTermMapObjects(ltm_object_ids);                                         // :301-306
    if (use_object_pose_factors) {                                                                                           // :308-340
      FactorInfoSet matching;
      pose_graph->getObservationFactorsBetweenFrameIdsInclusive(optimization_scope.min_frame_id_, optimization_scope.max_frame_id_, matching);
      for (const FactorInfo& fi : matching) {
        if (excluded(fi)) continue;
        ObjectId obj;
        if (pose_graph->getObjectIdForObjObservationFactor(fi, obj)) objects_to_include[obj].insert(fi);
      }
      applyMinObs(optimization_scope.min_object_observations_, objects_to_include, required_feature_factors, ltm_object_ids);
    }
    if (use_object_only_factors) {                                                                                           // :342-405
      std::unordered_set<ObjectId> with_object_only;
      for (const auto& o : objects_to_include) if (!fix_ltm_param_blocks || !ltm_object_ids.count(o.first)) with_object_only.insert(o.first);
      if (!fix_ltm_param_blocks && optimization_scope.force_include_ltm_objs_) with_object_only.insert(ltm_object_ids.begin(), ltm_object_ids.end());
      std::unordered_map<ObjectId, FactorInfoSet> by_obj;
      pose_graph->getOnlyObjectFactorsForObjects(with_object_only, optimization_scope.use_pom_, true, by_obj);
      for (const auto& o : by_obj) {
        objects_to_include[o.first].insert(o.second.begin(), o.second.end());
        for (const FactorInfo& fi : o.second) required_feature_factors[fi.first].insert(fi.second);
      }
    }

    lap_("object factors");
    // ---- flatten (the part that replaces addOrRefreshResidualBlocksForRequiredFactors, :415, :991-1055) ----
    obvi::FlatProblem& fp = problem->flat;
    fp.reset();
    std::map<CameraId, uint16_t> cam_index;
    for (const auto& c : pose_graph->intrinsics()) {
      CameraExtrinsics e;
      if (!pose_graph->getExtrinsicsForCamera(c.first, e)) continue;
      cam_index[c.first] = (uint16_t)fp.cameras.size();
      fp.cameras.push_back(c.first);
      fp.cam_K.insert(fp.cam_K.end(), {c.second.fx, c.second.fy, c.second.cx, c.second.cy});
      // Pose3D orientation (axis-angle) -> quaternion xyzw
      const double th = std::sqrt(e.orientation_[0] * e.orientation_[0] + e.orientation_[1] * e.orientation_[1] + e.orientation_[2] * e.orientation_[2]);
      const double s = th > 0 ? std::sin(th / 2) / th : 0.5;
      fp.cam_ext.insert(fp.cam_ext.end(), {e.orientation_[0] * s, e.orientation_[1] * s, e.orientation_[2] * s, std::cos(th / 2), e.transl_[0], e.transl_[1], e.transl_[2]});
    }
    std::map<FrameId, uint32_t> pose_index;
    for (const FrameId& f : optimized_frames) {
      double* p = nullptr;
      if (!pose_graph->getPosePointers(f, &p)) continue;
      pose_index[f] = (uint32_t)fp.frames.size(); fp.frames.push_back(f); fp.pose_ptrs.push_back(p);
    }
    // constness of poses (:424-472)
    fp.pose_const.assign(fp.frames.size(), 0);
    if (fix_pose_param_blocks) {
      std::fill(fp.pose_const.begin(), fp.pose_const.end(), 1);
    } else if (optimization_scope.min_frame_id_ == 0) {
      auto it = pose_index.find(0); if (it != pose_index.end()) fp.pose_const[it->second] = 1;
    } else {
      const uint32_t n_const = std::max<uint32_t>(1, optimization_scope.poses_prior_to_window_to_keep_constant_);
      for (uint32_t k = 0; k < n_const; ++k) {
        const FrameId f = optimization_scope.min_frame_id_ + k;
        if (f > optimization_scope.max_frame_id_) break;
        auto it = pose_index.find(f); if (it != pose_index.end()) fp.pose_const[it->second] = 1;
      }
    }
    std::vector<int32_t>& point_of_slot = point_of_slot_;   // feature slot -> point of the flat problem (valid for the included slots)
    if (use_feature_pose_factors) {
      point_of_slot.resize(pose_graph->numFeatureSlots());
      fp.features.reserve(included_slots.size()); fp.point_ptrs.reserve(included_slots.size());
      for (const uint32_t slot : included_slots) {
        double* p = pose_graph->featurePointerOfSlot(slot);
        point_of_slot[slot] = p ? (int32_t)fp.features.size() : -1;
        if (p) { fp.features.push_back(pose_graph->featureIdOfSlot(slot)); fp.point_ptrs.push_back(p); }
      }
    }
    fp.point_const.assign(fp.features.size(), fix_visual_feature_param_blocks ? 1 : 0);                                     // :488-520
    std::map<ObjectId, uint32_t> object_index;
    if (use_object_param_blocks) {                                                                                          // :532-603
      for (const auto& o : objects_to_include) {
        double* p = nullptr;
        if (!pose_graph->getObjectParamPointers(o.first, &p)) continue;
        object_index[o.first] = (uint32_t)fp.objects.size(); fp.objects.push_back(o.first); fp.object_ptrs.push_back(p);
        const bool is_ltm = ltm_object_ids.count(o.first) != 0;
        fp.object_const.push_back((fix_object_param_blocks || (fix_ltm_param_blocks && is_ltm)) ? 1 : 0);
      }
    }
    lap_("parameter blocks");
    const auto& rp = residual_params;
    // residual blocks, type by type in the evaluate order of the ABI; inside a type by factor id
    std::unordered_map<FrameId, size_t> obs_per_frame;
    {                                                                                                                        // residual_creator.h:168-264
      // Residual blocks go by factor id, and the records come frame by frame in id order unless frames were filled out of order: the
      // first pass writes them as they come and notices a descent; only then the required records are sorted and written again.
      std::vector<const VisualRecord*>& sorted = scratch_records_;
      sorted.clear();
      size_t n = 0;
      auto put = [&](const VisualRecord& v, int64_t pose, int32_t cam) {
        const int32_t point = point_of_slot[v.feature_slot];
        if (pose < 0 || point < 0 || cam < 0) return;
        fp.rp_pose[n] = (uint32_t)pose; fp.rp_point[n] = (uint32_t)point; fp.rp_cam[n] = (uint16_t)cam;
        fp.rp_pixel[2 * n] = v.px; fp.rp_pixel[2 * n + 1] = v.py; fp.rp_sigma[n] = v.sigma;
        fp.blocks[n] = {kReprojectionErrorFactorTypeId, v.id};
        ++n;
      };
      auto pose_of = [&](const FrameId& f) { const auto pi = pose_index.find(f); return pi == pose_index.end() ? (int64_t)-1 : (int64_t)pi->second; };
      auto cam_of = [&](const CameraId& c) { const auto ci = cam_index.find(c); return ci == cam_index.end() ? (int32_t)-1 : (int32_t)ci->second; };
      size_t total = 0;
      for (const auto& sp : spans) total += sp.second;
      fp.rp_pose.resize(total); fp.rp_point.resize(total); fp.rp_cam.resize(total); fp.rp_pixel.resize(2 * total); fp.rp_sigma.resize(total); fp.blocks.resize(total);
      bool ascending = true, first = true; FeatureFactorId last_id = 0;
      // A global-BA frame flattens millions of records: the frames' spans are then written by ranges of spans on host threads (each span's
      // share of the arrays is known after a counting pass: same arrays, same order); a window's few ten thousand go the plain way.
      const unsigned hw = std::getenv("OBVI_HOST_BUILD_THREADS") ? (unsigned)std::atoi(std::getenv("OBVI_HOST_BUILD_THREADS")) : std::thread::hardware_concurrency();   // (knob: 1 = the plain loop)
      const size_t n_threads = total >= ((size_t)1 << 18) && hw > 1 ? std::min<size_t>({(size_t)8, (size_t)hw, spans.size()}) : 1;
      if (n_threads > 1) {
        std::vector<size_t> kept_of(spans.size(), 0), at(spans.size() + 1, 0);
        struct RangeIds { bool any = false, ascending = true; FeatureFactorId first_id = 0, last_id = 0; };
        std::vector<RangeIds> ids(n_threads);
        auto ranges = [&](auto&& body) {
          std::vector<std::thread> th;
          for (size_t t = 0; t < n_threads; ++t) th.emplace_back([&, t] { body(t, spans.size() * t / n_threads, spans.size() * (t + 1) / n_threads); });
          for (auto& x : th) x.join();
        };
        ranges([&](size_t t, size_t s0, size_t s1) {
          RangeIds& r = ids[t];
          for (size_t si = s0; si < s1; ++si) {
            size_t kept = 0;
            for (const VisualRecord* v = spans[si].first, *e = v + spans[si].second; v != e; ++v) {
              if (sightings_[v->feature_slot] < min_obs || visual_excluded(*v)) continue;
              ++kept;
              if (r.any && v->id < r.last_id) r.ascending = false;
              if (!r.any) { r.first_id = v->id; r.any = true; }
              r.last_id = v->id;
            }
            kept_of[si] = kept;
          }
        });
        for (const RangeIds& r : ids) {
          if (!r.any) continue;
          if (!r.ascending || (!first && r.first_id < last_id)) ascending = false;
          last_id = r.last_id; first = false;
        }
        for (size_t si = 0; si < spans.size(); ++si) at[si + 1] = at[si] + kept_of[si];
        if (ascending) {
          // (a record whose pose / point / camera is not in the problem is dropped by put(): then the arrays are compacted below)
          std::vector<size_t> written(spans.size(), 0);
          ranges([&](size_t, size_t s0, size_t s1) {
            for (size_t si = s0; si < s1; ++si) {
              if (spans[si].second == 0) continue;
              const int64_t pose = pose_of(spans[si].first->frame_id);
              CameraId last_cam = spans[si].first->camera_id; int32_t cam = cam_of(last_cam);
              size_t w = at[si];
              for (const VisualRecord* v = spans[si].first, *e = v + spans[si].second; v != e; ++v) {
                if (sightings_[v->feature_slot] < min_obs || visual_excluded(*v)) continue;
                if (v->camera_id != last_cam) { last_cam = v->camera_id; cam = cam_of(last_cam); }
                const int32_t point = point_of_slot[v->feature_slot];
                if (pose < 0 || point < 0 || cam < 0) continue;
                fp.rp_pose[w] = (uint32_t)pose; fp.rp_point[w] = (uint32_t)point; fp.rp_cam[w] = (uint16_t)cam;
                fp.rp_pixel[2 * w] = v->px; fp.rp_pixel[2 * w + 1] = v->py; fp.rp_sigma[w] = v->sigma;
                fp.blocks[w] = {kReprojectionErrorFactorTypeId, v->id};
                ++w;
              }
              written[si] = w - at[si];
            }
          });
          bool holes = false;
          for (size_t si = 0; si < spans.size(); ++si) holes = holes || written[si] != kept_of[si];
          n = at[spans.size()];
          if (holes) {   // rare: compact in order
            n = 0;
            for (size_t si = 0; si < spans.size(); ++si)
              for (size_t k = 0; k < written[si]; ++k, ++n) {
                const size_t from = at[si] + k;
                fp.rp_pose[n] = fp.rp_pose[from]; fp.rp_point[n] = fp.rp_point[from]; fp.rp_cam[n] = fp.rp_cam[from];
                fp.rp_pixel[2 * n] = fp.rp_pixel[2 * from]; fp.rp_pixel[2 * n + 1] = fp.rp_pixel[2 * from + 1]; fp.rp_sigma[n] = fp.rp_sigma[from]; fp.blocks[n] = fp.blocks[from];
              }
          }
        }
        for (size_t si = 0; si < spans.size(); ++si) if (kept_of[si]) obs_per_frame[spans[si].first->frame_id] += kept_of[si];
      } else
      for (const auto& sp : spans) {
        if (sp.second == 0) continue;
        const int64_t pose = pose_of(sp.first->frame_id);
        CameraId last_cam = sp.first->camera_id; int32_t cam = cam_of(last_cam);
        size_t kept = 0;
        for (const VisualRecord* v = sp.first, *e = sp.first + sp.second; v != e; ++v) {
          if (sightings_[v->feature_slot] < min_obs || visual_excluded(*v)) continue;
          ++kept;
          if (!first && v->id < last_id) ascending = false;
          last_id = v->id; first = false;
          if (v->camera_id != last_cam) { last_cam = v->camera_id; cam = cam_of(last_cam); }
          put(*v, pose, cam);
        }
        if (kept) obs_per_frame[sp.first->frame_id] += kept;
      }
      if (!ascending) {
        for (const auto& sp : spans)
          for (const VisualRecord* v = sp.first, *e = sp.first + sp.second; v != e; ++v)
            if (sightings_[v->feature_slot] >= min_obs && !visual_excluded(*v)) sorted.push_back(v);
        std::sort(sorted.begin(), sorted.end(), [](const VisualRecord* a, const VisualRecord* b) { return a->id < b->id; });
        n = 0;
        for (const VisualRecord* v : sorted) put(*v, pose_of(v->frame_id), cam_of(v->camera_id));
      }
      fp.rp_pose.resize(n); fp.rp_point.resize(n); fp.rp_cam.resize(n); fp.rp_pixel.resize(2 * n); fp.rp_sigma.resize(n); fp.blocks.resize(n);
    }
    lap_("reprojection factors");
    if (use_relative_pose_factors) {                                                                                         // :240-299
      for (const FrameId& f : optimized_frames) {
        auto it = obs_per_frame.find(f);
        if (it != obs_per_frame.end() && it->second >= optimization_scope.min_low_level_feature_observations_per_frame_) continue;
        FactorInfoSet rel;
        pose_graph->getPoseFactorInfoByFrameId(f, optimization_scope.min_frame_id_, optimization_scope.max_frame_id_, rel);
        for (const FactorInfo& fi : rel) required_feature_factors[fi.first].insert(fi.second);
      }
    }
    for (const FactorType& t : optimization_scope.factor_types_to_exclude) required_feature_factors.erase(t);               // :407-410
    for (FeatureFactorId id : required_feature_factors[kObjectObservationFactorTypeId]) {                                    // residual_creator.h:20-117
      ObjectObservationFactor f;
      if (!pose_graph->getObjectObservationFactor(id, f)) continue;
      auto oi = object_index.find(f.object_id_); auto pi = pose_index.find(f.frame_id_); auto ci = cam_index.find(f.camera_id_);
      if (oi == object_index.end() || pi == pose_index.end() || ci == cam_index.end()) continue;
      fp.bb_obj.push_back(oi->second); fp.bb_pose.push_back(pi->second); fp.bb_cam.push_back(ci->second);
      fp.bb_corners.insert(fp.bb_corners.end(), f.bounding_box_corners_.begin(), f.bounding_box_corners_.end());
      fp.bb_cov.insert(fp.bb_cov.end(), f.bounding_box_corners_covariance_.begin(), f.bounding_box_corners_covariance_.end());
      fp.blocks.push_back({kObjectObservationFactorTypeId, id});
    }
    for (FeatureFactorId id : required_feature_factors[kShapeDimPriorFactorTypeId]) {                                        // residual_creator.h:119-166
      ShapeDimPriorFactor f;
      if (!pose_graph->getShapeDimPriorFactor(id, f)) continue;
      auto oi = object_index.find(f.object_id_); if (oi == object_index.end()) continue;
      fp.sp_obj.push_back(oi->second); fp.sp_mean.insert(fp.sp_mean.end(), f.mean_shape_dim_.begin(), f.mean_shape_dim_.end());
      fp.sp_cov.insert(fp.sp_cov.end(), f.shape_dim_cov_.begin(), f.shape_dim_cov_.end());
      fp.blocks.push_back({kShapeDimPriorFactorTypeId, id});
    }
    for (FeatureFactorId id : required_feature_factors[kLongTermMapFactorTypeId]) {                                          // long_term_map_factor_creator.h:265-322
      LongTermMapObjectPrior f;
      if (!pose_graph->getLongTermMapFactor(id, f)) continue;
      auto oi = object_index.find(f.object_id_); if (oi == object_index.end()) continue;
      fp.lt_obj.push_back(oi->second); fp.lt_mean.insert(fp.lt_mean.end(), f.ellipsoid_mean_.begin(), f.ellipsoid_mean_.end());
      fp.lt_cov.insert(fp.lt_cov.end(), f.covariance_.begin(), f.covariance_.end());
      fp.blocks.push_back({kLongTermMapFactorTypeId, id});
    }
    auto add_relpose = [&](const RelPoseFactor& f, const FactorInfo& info) {                                                 // residual_creator.h:266-345
      auto a = pose_index.find(f.frame_id_1_), b = pose_index.find(f.frame_id_2_);
      if (a == pose_index.end() || b == pose_index.end()) return;
      fp.rl_a.push_back(a->second); fp.rl_b.push_back(b->second);
      fp.rl_t.insert(fp.rl_t.end(), f.measured_pose_deviation_.transl_.begin(), f.measured_pose_deviation_.transl_.end());
      fp.rl_aa.insert(fp.rl_aa.end(), f.measured_pose_deviation_.orientation_.begin(), f.measured_pose_deviation_.orientation_.end());
      fp.rl_cov.insert(fp.rl_cov.end(), f.pose_deviation_cov_.begin(), f.pose_deviation_cov_.end());
      fp.blocks.push_back(info);
    };
    fp.rl_huber = rp.relative_pose_factor_huber_loss_;
    for (FeatureFactorId id : required_feature_factors[kPairwiseRobotPoseFactorTypeId]) { RelPoseFactor f; if (pose_graph->getPoseFactor(id, f)) add_relpose(f, {kPairwiseRobotPoseFactorTypeId, id}); }
    if (!problem->extraRelativePoseBlocks().empty()) {
      fp.rl_huber = problem->extraRelativePoseHuber();
      uint64_t k = 0;
      for (const RelPoseFactor& f : problem->extraRelativePoseBlocks()) add_relpose(f, {kPairwiseRobotPoseFactorTypeId, (FeatureFactorId)(~0ull - k++)});
    }
    last_optimized_nodes_ = optimized_frames.size(); last_optimized_features_ = fp.features.size(); last_optimized_objects_ = fp.objects.size();
    if (opt_logger.has_value()) opt_logger->setOptimizationParams(last_optimized_objects_, last_optimized_features_, last_optimized_nodes_);   // :625-629
    lap_("small factor families");
    if (factor_hooks_) applyFactorHooks(residual_params, pose_graph, problem);
    return ResidualBlockInfoMap(fp.blocks);
  }

  // the caller's per-factor hooks over the blocks the build selected (FactorHooks above): vetoed blocks are struck from the flat problem
  void applyFactorHooks(const pose_graph_optimization::ObjectVisualPoseGraphResidualParams& residual_params, std::shared_ptr<PoseGraphType>& pose_graph, obvi::Problem* problem) {
    obvi::FlatProblem& fp = problem->flat;
    const size_t n = fp.blocks.size();
    std::vector<uint8_t> keep(n, 1);
    size_t dropped = 0, next_id = 0;
    for (size_t i = 0; i < n; ++i) {
      const FactorHooks::Key key = fp.blocks[i];
      if (factor_hooks_.keep && factor_hooks_.keep(key, pose_graph)) { ++next_id; continue; }
      obvi::ResidualBlockId id = (obvi::ResidualBlockId)next_id;
      if (!factor_hooks_.create(key, residual_params, pose_graph, problem, id)) {
        keep[i] = 0; ++dropped;
        std::cerr << "Could not make residual for factor type " << (int)key.first << " and factor id " << key.second << std::endl;   // object_pose_graph_optimizer.h:1049-1050
      } else {
        ++next_id;
      }
    }
    n_vetoed_total_ += dropped;
    if (dropped == 0) return;
    size_t at = 0;
    auto rows = [&](auto& vec, size_t width, size_t count) {   // keeps the rows (of `width` entries) of a family whose blocks are [at, at + count)
      size_t w = 0;
      for (size_t i = 0; i < count; ++i) {
        if (!keep[at + i]) continue;
        if (w != i) for (size_t k = 0; k < width; ++k) vec[w * width + k] = vec[i * width + k];
        ++w;
      }
      vec.resize(w * width);
    };
    size_t c = fp.rp_pose.size();
    rows(fp.rp_pose, 1, c); rows(fp.rp_point, 1, c); rows(fp.rp_cam, 1, c); rows(fp.rp_pixel, 2, c); rows(fp.rp_sigma, 1, c); at += c;
    c = fp.bb_obj.size();
    rows(fp.bb_obj, 1, c); rows(fp.bb_pose, 1, c); rows(fp.bb_cam, 1, c); rows(fp.bb_corners, 4, c); rows(fp.bb_cov, 16, c); at += c;
    c = fp.sp_obj.size();
    rows(fp.sp_obj, 1, c); rows(fp.sp_mean, 3, c); rows(fp.sp_cov, 9, c); at += c;
    c = fp.lt_obj.size();
    rows(fp.lt_obj, 1, c); rows(fp.lt_mean, 7, c); rows(fp.lt_cov, 49, c); at += c;
    c = fp.rl_a.size();
    rows(fp.rl_a, 1, c); rows(fp.rl_b, 1, c); rows(fp.rl_t, 3, c); rows(fp.rl_aa, 3, c); rows(fp.rl_cov, 36, c); at += c;
    size_t w = 0;
    for (size_t i = 0; i < n; ++i) if (keep[i]) fp.blocks[w++] = fp.blocks[i];
    fp.blocks.resize(w);
  }

  // Phase II of a two-phase optimisation (offline_problem_runner.h:803-892) re-runs buildPoseGraphOptimization with the excluded
  // factors.  What that rebuild selects can be stated on the flat problem phase I already built: an excluded visual / bounding-box
  // factor is masked; a feature left with fewer than min_low_level_feature_observations sightings, or an object left with fewer
  // than min_object_observations boxes (long-term-map objects excepted, :826-861), loses all its factors and so drops out; the
  // object-only factors follow the objects (:342-405).  Returns false when the rebuild would ADD something the phase-I problem does
  // not hold (relative-pose factors for a frame that fell below min_low_level_feature_observations_per_frame_, :240-299): the caller
  // then rebuilds as the reference does.
  struct PhaseTwoMasks { std::vector<uint8_t> rp, bb, sp, lt; size_t n_features = 0, n_objects = 0; };
  template <class PoseGraphPtr>
  bool excludeFromBuiltProblem(const OptimizationScopeParams& scope, const PoseGraphPtr& pose_graph, const FactorInfoSet& excluded, const obvi::Problem& problem,
                               PhaseTwoMasks* out, const std::vector<uint8_t>* rp_keep = nullptr, const std::vector<uint8_t>* bb_keep = nullptr) const {
    // rp_keep / bb_keep: the exclusion already as one byte per factor of the flat problem (obvi_ba_select_outliers), instead of the set
    const obvi::FlatProblem& fp = problem.flat;
    const size_t n_rp = fp.rp_pose.size(), n_bb = fp.bb_obj.size(), n_sp = fp.sp_obj.size(), n_lt = fp.lt_obj.size();
    if (fp.blocks.size() < n_rp + n_bb + n_sp + n_lt) return false;
    auto is_excluded = [&](size_t block) {
      if (block < n_rp && rp_keep != nullptr) return (*rp_keep)[block] == 0;
      if (block >= n_rp && block < n_rp + n_bb && bb_keep != nullptr) return (*bb_keep)[block - n_rp] == 0;
      return excluded.count(fp.blocks[block]) != 0;
    };
    out->rp.assign(n_rp, 1); out->bb.assign(n_bb, 1); out->sp.assign(n_sp, 1); out->lt.assign(n_lt, 1);
    // visual factors
    std::vector<uint32_t> sightings(fp.features.size(), 0), per_frame_before(fp.frames.size(), 0), per_frame_after(fp.frames.size(), 0);
    for (size_t i = 0; i < n_rp; ++i) {
      ++per_frame_before[fp.rp_pose[i]];
      if (is_excluded(i)) out->rp[i] = 0; else ++sightings[fp.rp_point[i]];
    }
    out->n_features = 0;
    for (uint32_t c : sightings) if (c >= scope.min_low_level_feature_observations_) ++out->n_features;
    for (size_t i = 0; i < n_rp; ++i) {
      if (out->rp[i] && sightings[fp.rp_point[i]] < scope.min_low_level_feature_observations_) out->rp[i] = 0;
      if (out->rp[i]) ++per_frame_after[fp.rp_pose[i]];
    }
    if (scope.include_visual_factors_ && scope.min_low_level_feature_observations_per_frame_ > 0) {   // (use_relative_pose_factors of the build) a frame that newly falls below the per-frame minimum would get odometry factors
      for (size_t f = 0; f < fp.frames.size(); ++f) {
        const bool below_before = per_frame_before[f] < scope.min_low_level_feature_observations_per_frame_, below_after = per_frame_after[f] < scope.min_low_level_feature_observations_per_frame_;
        if (below_after != below_before) return false;
      }
    }
    // objects
    std::unordered_set<ObjectId> ltm_object_ids;
    pose_graph->getLongTermMapObjects(ltm_object_ids);
    const bool fix_ltm = scope.fix_objects_ || scope.fix_ltm_objects_;
    std::vector<uint32_t> boxes(fp.objects.size(), 0);
    for (size_t i = 0; i < n_bb; ++i) { if (is_excluded(n_rp + i)) out->bb[i] = 0; else ++boxes[fp.bb_obj[i]]; }
    std::vector<uint8_t> included(fp.objects.size(), 0), object_only(fp.objects.size(), 0);
    out->n_objects = 0;
    for (size_t o = 0; o < fp.objects.size(); ++o) {
      const bool is_ltm = ltm_object_ids.count(fp.objects[o]) != 0;
      included[o] = boxes[o] >= scope.min_object_observations_ || (boxes[o] > 0 && is_ltm);
      object_only[o] = (included[o] && (!fix_ltm || !is_ltm)) || (!fix_ltm && scope.force_include_ltm_objs_ && is_ltm);
      if (included[o] || object_only[o]) ++out->n_objects;
    }
    for (size_t i = 0; i < n_bb; ++i) if (!included[fp.bb_obj[i]]) out->bb[i] = 0;
    for (size_t i = 0; i < n_sp; ++i) if (!object_only[fp.sp_obj[i]]) out->sp[i] = 0;
    for (size_t i = 0; i < n_lt; ++i) if (!object_only[fp.lt_obj[i]]) out->lt[i] = 0;
    return true;
  }
  void setPhaseTwoLogCounts(const PhaseTwoMasks& m, std::optional<OptimizationLogger>& opt_logger) {
    last_optimized_features_ = m.n_features; last_optimized_objects_ = m.n_objects;
    if (opt_logger.has_value()) opt_logger->setOptimizationParams(last_optimized_objects_, last_optimized_features_, last_optimized_nodes_);
  }

  // object_pose_graph_optimizer.h:634-707
  bool solveOptimization(obvi::Problem* problem, const pose_graph_optimization::OptimizationSolverParams& solver_params,
                         std::optional<OptimizationLogger>& opt_logger, std::vector<obvi::ResidualBlockId>* residual_block_id_ptrs = nullptr,
                         std::vector<double>* residual_ptrs = nullptr, std::shared_ptr<obvi::SolverSummary> solver_summary = nullptr,
                         const PhaseTwoMasks* phase_two_masks = nullptr, bool keep_for_phase_two = false, const BesideSolve* beside = nullptr) {
    // beside (optional): the caller's hooks around the wait for the device inside obvi_ba_solve -- start() right before it, join() right after it and
    // before the result is written into the pose graph's blocks (the runner plans the next window on a second thread there: obvi_runner.h)
    if (problem == nullptr) return false;
    obvi_ba_handle* h = problem->handle();
    if (h == nullptr) { std::cerr << "solveOptimization: no device handle" << std::endl; return false; }
    const obvi::FlatProblem& fp = problem->flat;
    const auto& rp = residual_params_;
    auto gather = [](const std::vector<double*>& ptrs, int dim) { std::vector<double> v(ptrs.size() * dim); for (size_t i = 0; i < ptrs.size(); ++i) std::copy_n(ptrs[i], dim, &v[dim * i]); return v; };
    std::vector<double> poses = gather(fp.pose_ptrs, 6), points = gather(fp.point_ptrs, 3), objects = gather(fp.object_ptrs, 7);
    const auto t_upload = std::chrono::steady_clock::now();
    int rc = 0;
    if (phase_two_masks != nullptr) {
      // phase II on the problem phase I left on the device: the values go back to the snapshot taken behind phase I's upload (the
      // host side was restored by the caller, :811), the excluded factors are masked; no flattening, no upload, and the library
      // keeps its symbolic plan (the masks select a subset of what it was built for)
      rc = obvi_ba_restore(h);
      if (!rc) rc = obvi_ba_set_active_mask(h, OBVI_FACTOR_REPROJECTION, phase_two_masks->rp.data());
      if (!rc) rc = obvi_ba_set_active_mask(h, OBVI_FACTOR_BBOX, phase_two_masks->bb.data());
      if (!rc) rc = obvi_ba_set_active_mask(h, OBVI_FACTOR_SHAPE_PRIOR, phase_two_masks->sp.data());
      if (!rc) rc = obvi_ba_set_active_mask(h, OBVI_FACTOR_LTM_PRIOR, phase_two_masks->lt.data());
      if (rc) { std::cerr << "obvi_ba phase-II masks failed: " << obvi_ba_last_error(h) << std::endl; return false; }
    }
    if (phase_two_masks == nullptr) {
    if (problem->uploadedAhead()) {   // structure and plan went up while the previous window was being solved: the values exist only now
      rc = obvi_ba_update_state(h, poses.data(), points.data(), objects.data());
      problem->setUploadedAhead(false);
    } else {
      rc = uploadFlatProblem(h, fp, rp, poses, points, objects);
    }
    if (rc) { std::cerr << "obvi_ba upload failed: " << obvi_ba_last_error(h) << std::endl; return false; }
    if ((residual_ptrs != nullptr || keep_for_phase_two) && obvi_ba_snapshot(h)) return false;   // phase I of a two-phase optimisation: phase II starts from these values
    }
    obvi_solver_params p{solver_params.max_num_iterations_, solver_params.allow_non_monotonic_steps_ ? 1 : 0, solver_params.function_tolerance_,
                         solver_params.gradient_tolerance_, solver_params.parameter_tolerance_, solver_params.initial_trust_region_radius_,
                         solver_params.max_trust_region_radius_};
    obvi_summary s;
    const auto t_solve = std::chrono::steady_clock::now();
    if (beside != nullptr && beside->start) beside->start();
    rc = obvi_ba_solve(h, &p, &s);
    const auto t_solved = std::chrono::steady_clock::now();
    if (beside != nullptr && beside->join) { beside->join(); time_beside_wait_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_solved).count(); }
    if (rc) { std::cerr << "obvi_ba_solve failed: " << obvi_ba_last_error(h) << std::endl; return false; }
    obvi::SolverSummary summary;
    summary.termination_type = s.termination_type; summary.usable = s.is_solution_usable != 0;
    summary.initial_cost = s.initial_cost; summary.final_cost = s.final_cost; summary.fixed_cost = s.fixed_cost;
    summary.total_time_in_seconds = s.total_time_in_seconds; summary.linear_solver_time_in_seconds = s.linear_solver_time_in_seconds;
    summary.jacobian_evaluation_time_in_seconds = s.jacobian_evaluation_time_in_seconds; summary.residual_evaluation_time_in_seconds = s.residual_evaluation_time_in_seconds;
    summary.num_parameters_reduced = s.num_parameters_reduced; summary.num_residuals_reduced = s.num_residuals_reduced; summary.message = s.message;
    std::vector<obvi_iteration_summary> its((size_t)std::max(s.num_iterations, 1));
    const int nit = obvi_ba_get_iterations(h, its.data(), (int32_t)its.size());
    for (int i = 0; i < nit; ++i) summary.iterations.push_back({its[i].iteration, its[i].cost, its[i].cost_change, its[i].step_norm, its[i].gradient_max_norm, its[i].step_is_successful != 0});

    if (residual_block_id_ptrs != nullptr) { residual_block_id_ptrs->resize(fp.blocks.size()); for (size_t i = 0; i < fp.blocks.size(); ++i) (*residual_block_id_ptrs)[i] = i; }   // :679-681
    if (residual_ptrs != nullptr) {                                                                                          // :682-693
      residual_ptrs->assign((size_t)obvi_ba_num_residuals(h), 0.0);
      if (obvi_ba_evaluate(h, /*apply_loss=*/0, nullptr, residual_ptrs->data(), nullptr)) return false;
    }
    if (solver_summary != nullptr) *solver_summary = summary;
    if (summary.termination_type == OBVI_FAILURE) std::cerr << "obvi_ba optimization failed: " << summary.message << std::endl;
    if (opt_logger.has_value()) opt_logger->extractOptimizationTimingResults(summary);
    // Ceres mutates the parameter blocks in place; copy the device state back into the pose graph's blocks -- unless the solve
    // failed: Ceres leaves the user's blocks as they were when the solution is not usable (the library hands the entry state back)
    if (summary.IsSolutionUsable()) {
      if (obvi_ba_get_state(h, poses.data(), points.data(), objects.data())) return false;
      for (size_t i = 0; i < fp.pose_ptrs.size(); ++i) std::copy_n(&poses[6 * i], 6, fp.pose_ptrs[i]);
      for (size_t i = 0; i < fp.point_ptrs.size(); ++i) std::copy_n(&points[3 * i], 3, fp.point_ptrs[i]);
      for (size_t i = 0; i < fp.object_ptrs.size(); ++i) std::copy_n(&objects[7 * i], 7, fp.object_ptrs[i]);
    }
    last_summary_ = summary;
    const auto t_end = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    time_upload_ms_ += ms(t_upload, t_solve); time_solve_ms_ += ms(t_solve, t_solved); time_readback_ms_ += ms(t_solved, t_end); time_lm_ms_ += 1e3 * s.total_time_in_seconds; ++n_solves_;
    return summary.IsSolutionUsable();
  }

  // What buildPoseGraphOptimization left in `problem->flat`, onto the device WITH its symbolic plan, ahead of the solve (obvi_ba_prepare): for a caller that
  // builds window f+1 while window f is solved.  The values uploaded are placeholders (zeros: the blocks' values may be written by another thread at this
  // moment, and the symbolic phase reads none); solveOptimization puts the real ones there (obvi_ba_update_state).
  bool uploadAndPlanAhead(obvi::Problem* problem) {
    if (problem == nullptr || problem->dryRun()) return false;
    obvi_ba_handle* h = problem->handle();
    if (h == nullptr) return false;
    const obvi::FlatProblem& fp = problem->flat;
    const std::vector<double> poses(fp.pose_ptrs.size() * 6, 0.0), points(fp.point_ptrs.size() * 3, 0.0), objects(fp.object_ptrs.size() * 7, 0.0);
    int rc = uploadFlatProblem(h, fp, residual_params_, poses, points, objects);
    if (!rc) rc = obvi_ba_prepare(h);
    if (rc) { std::cerr << "obvi_ba upload ahead failed: " << obvi_ba_last_error(h) << std::endl; return false; }
    problem->setUploadedAhead(true);
    return true;
  }
  // the build another optimiser instance made (the one that planned ahead) becomes this one's: what solveOptimization and the logger read of a build
  void adoptBuild(const ObjectPoseGraphOptimizer& other, std::optional<OptimizationLogger>& opt_logger) {
    residual_params_ = other.residual_params_;
    last_optimized_objects_ = other.last_optimized_objects_; last_optimized_features_ = other.last_optimized_features_; last_optimized_nodes_ = other.last_optimized_nodes_;
    if (opt_logger.has_value()) opt_logger->setOptimizationParams(last_optimized_objects_, last_optimized_features_, last_optimized_nodes_);
  }
  double besideWaitMs() const { return time_beside_wait_ms_; }

  void clearPastOptimizationData() { last_optimized_objects_ = last_optimized_features_ = last_optimized_nodes_ = 0; }     // :792-797
  const obvi::SolverSummary& lastSummary() const { return last_summary_; }
  // where a solveOptimization call spends its wall time: upload (set_*), obvi_ba_solve (symbolic phase + LM loop), read-back; the LM loop alone
  void printTiming(std::ostream& os) const {
    if (n_solves_ == 0) return;
    os << "solveOptimization x" << n_solves_ << ": upload " << time_upload_ms_ / n_solves_ << " ms, obvi_ba_solve " << time_solve_ms_ / n_solves_ << " ms (LM loop "
       << time_lm_ms_ / n_solves_ << " ms), evaluate + read-back " << time_readback_ms_ / n_solves_ << " ms per call" << std::endl;
  }

 private:
  static int uploadFlatProblem(obvi_ba_handle* h, const obvi::FlatProblem& fp, const pose_graph_optimization::ObjectVisualPoseGraphResidualParams& rp,
                               const std::vector<double>& poses, const std::vector<double>& points, const std::vector<double>& objects) {
    int rc = obvi_ba_set_cameras(h, (int32_t)fp.cameras.size(), fp.cam_K.data(), fp.cam_ext.data());
    if (!rc) rc = obvi_ba_set_poses(h, (int64_t)fp.frames.size(), poses.data(), fp.pose_const.data());
    if (!rc) rc = obvi_ba_set_points(h, (int64_t)fp.features.size(), points.data(), fp.point_const.data());
    if (!rc) rc = obvi_ba_set_objects(h, (int64_t)fp.objects.size(), objects.data(), fp.object_const.data());
    if (!rc) rc = obvi_ba_set_reproj(h, (int64_t)fp.rp_pose.size(), fp.rp_pose.data(), fp.rp_point.data(), fp.rp_cam.data(), fp.rp_pixel.data(), fp.rp_sigma.data(), 0.0,
                                     rp.visual_residual_params_.reprojection_error_huber_loss_param_);
    if (!rc) rc = obvi_ba_set_bbox(h, (int64_t)fp.bb_obj.size(), fp.bb_obj.data(), fp.bb_pose.data(), fp.bb_cam.data(), fp.bb_corners.data(), fp.bb_cov.data(),
                                   rp.object_residual_params_.object_observation_huber_loss_param_, rp.object_residual_params_.invalid_ellipsoid_error_val_);
    if (!rc) rc = obvi_ba_set_shape_priors(h, (int64_t)fp.sp_obj.size(), fp.sp_obj.data(), fp.sp_mean.data(), fp.sp_cov.data(), rp.object_residual_params_.shape_dim_prior_factor_huber_loss_param_);
    if (!rc) rc = obvi_ba_set_ltm_priors(h, (int64_t)fp.lt_obj.size(), fp.lt_obj.data(), fp.lt_mean.data(), fp.lt_cov.data(), rp.long_term_map_params_.pair_huber_loss_param_);
    if (!rc) rc = obvi_ba_set_relpose(h, (int64_t)fp.rl_a.size(), fp.rl_a.data(), fp.rl_b.data(), fp.rl_t.data(), fp.rl_aa.data(), fp.rl_cov.data(), fp.rl_huber);
    return rc;
  }
  template <class Id>
  static void applyMinObs(size_t min_obs, std::map<Id, FactorInfoSet>& by_id, std::map<FactorType, std::set<FeatureFactorId>>& required, const std::unordered_set<Id>& ignore) {   // :826-861
    for (auto it = by_id.begin(); it != by_id.end();) {
      if (it->second.size() >= min_obs || ignore.count(it->first)) { for (const FactorInfo& fi : it->second) required[fi.first].insert(fi.second); ++it; }
      else it = by_id.erase(it);
    }
  }
  pose_graph_optimization::ObjectVisualPoseGraphResidualParams residual_params_;
  size_t last_optimized_objects_ = 0, last_optimized_features_ = 0, last_optimized_nodes_ = 0;
  FactorHooks factor_hooks_;
  size_t n_vetoed_total_ = 0;
  // scratch of buildPoseGraphOptimization, kept between calls (a window is built for every frame)
  std::vector<const PoseGraphType::VisualFactorRecord*> scratch_records_;
  std::vector<uint32_t> sightings_;
  std::vector<int32_t> point_of_slot_;
  std::vector<std::pair<FeatureId, uint32_t>> scratch_keyed_;
  obvi::SolverSummary last_summary_;
  double time_upload_ms_ = 0, time_solve_ms_ = 0, time_readback_ms_ = 0, time_lm_ms_ = 0, time_beside_wait_ms_ = 0; size_t n_solves_ = 0;
};

// pose_graph_plus_objects_optimizer.h:23-353
inline bool runPgoPlusEllipsoids(const FrameId& max_frame_id, const OptimizationScopeParams& optimization_scope_params,
                                 const pose_graph_optimization::ObjectVisualPoseGraphResidualParams& residual_params,
                                 const pose_graph_optimization::PoseGraphPlusObjectsOptimizationParams& pgo_solver_params, const bool& final_run,
                                 std::optional<OptimizationLogger>& opt_logger, std::shared_ptr<PoseGraphType>& pose_graph, int device_id = 0,
                                 const int& attempt_num = 0) {
  const bool timing = std::getenv("OBVI_HOST_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::cerr << "  runPgoPlusEllipsoids: " << what << " " << std::chrono::duration<double, std::milli>(now - t_last).count() << " ms" << std::endl;
    t_last = now;
  };
  std::unordered_map<FrameId, RawPose3d> raw;
  pose_graph->getRobotPoseEstimates(raw);
  ObjectPoseGraphOptimizer optimizer;
  obvi::Problem problem(device_id);
  lap("estimates + problem (device handle)");
  for (FrameId f = 1; f <= max_frame_id; ++f) {                                                                              // :94-127
    if (!raw.count(f) || !raw.count(f - 1)) { std::cerr << "Could not find current estimate for frame num " << f << std::endl; return false; }
    RelPoseFactor rel;
    rel.frame_id_1_ = f - 1; rel.frame_id_2_ = f;
    rel.measured_pose_deviation_ = getPose2RelativeToPose1(convertToPose3D(raw.at(f - 1)), convertToPose3D(raw.at(f)));
    const auto& c = pgo_solver_params.relative_pose_cov_params_;
    rel.pose_deviation_cov_ = generateOdomCov(rel.measured_pose_deviation_, c.transl_error_mult_for_transl_error_, c.transl_error_mult_for_rot_error_,
                                              c.rot_error_mult_for_transl_error_, c.rot_error_mult_for_rot_error_);
    problem.AddRelativePoseResidualBlock(rel, pgo_solver_params.relative_pose_factor_huber_loss_);                           // :129-159
  }
  OptimizationScopeParams scope_pgo = optimization_scope_params;                                                              // :161-165
  scope_pgo.include_visual_factors_ = false;
  scope_pgo.poses_prior_to_window_to_keep_constant_ = 1;
  // (the reference keys this by feature id in a map filled from a copy of all feature estimates and looks every feature's first frame and that
  // frame's pose up in two more maps; the graph's dense feature slots hold the same entries -- 300 000 at a global-BA frame -- and the poses
  // are an array by frame here: same values, no hashing)
  struct RelativeToFirst { uint32_t slot; FrameId first; Position3d relative; };
  std::vector<RelativeToFirst> relative_positions_from_first;
  // the rotation of a frame's pose once per frame (getPositionRelativeToPose / combinePoseAndPosition form it per call: the same matrix
  // for every feature first seen from that frame)
  struct FramePose { bool have = false; Mat3 R; Position3d t; };
  std::vector<FramePose> pose_of_frame((size_t)max_frame_id + 1);
  auto cache_poses = [&]() {
    for (FramePose& fp : pose_of_frame) fp.have = false;
    for (const auto& r : raw) if (r.first <= max_frame_id) { FramePose& fp = pose_of_frame[r.first]; fp.have = true; fp.R = rotationFromAxisAngle({{r.second[3], r.second[4], r.second[5]}}); fp.t = {{r.second[0], r.second[1], r.second[2]}}; }
  };
  cache_poses();
  if (pgo_solver_params.enable_visual_non_opt_feature_adjustment_post_pgo_) {                                                // :167-199
    const size_t n_slots = pose_graph->numFeatureSlots();
    relative_positions_from_first.reserve(n_slots);
    for (uint32_t slot = 0; slot < n_slots; ++slot) {
      const double* p = pose_graph->featurePointerOfSlot(slot);
      const FrameId first = pose_graph->firstObservedFrameOfSlot(slot);
      if (p == nullptr || first == PoseGraphType::kNoFrame || first > max_frame_id || !pose_of_frame[first].have) continue;
      const Mat3& R = pose_of_frame[first].R; const Position3d& t = pose_of_frame[first].t;
      Position3d o;   // getPositionRelativeToPose
      for (int i = 0; i < 3; ++i) o[i] = R[i] * (p[0] - t[0]) + R[3 + i] * (p[1] - t[1]) + R[6 + i] * (p[2] - t[2]);
      relative_positions_from_first.push_back({slot, first, o});
    }
  }
  lap("relative-pose factors + feature positions relative to their first frame");
  if (opt_logger.has_value()) opt_logger->setOptimizationTypeParams(max_frame_id, false, true, true, attempt_num);          // :201-204
  optimizer.buildPoseGraphOptimization(scope_pgo, residual_params, pose_graph, &problem, opt_logger);
  lap("build (objects + relative poses)");
  // The features-only adjustment behind the pose-graph solve has every feature sighting of the scope to flatten, upload and plan (config #3: 60-90 ms), and its
  // STRUCTURE depends on nothing the pose-graph solve computes: a second thread builds it on a problem (and device handle) of its own while that solve runs; the
  // values -- poses from the solve, features moved along with their first frame -- are handed over before it is solved (solveOptimization: obvi_ba_update_state).
  // OBVI_HOST_PLAN_AHEAD=0: one after the other on the stage's one problem.
  static const bool plan_ahead = !std::getenv("OBVI_HOST_PLAN_AHEAD") || std::atoi(std::getenv("OBVI_HOST_PLAN_AHEAD")) != 0;
  const bool vf_beside = plan_ahead && pgo_solver_params.enable_visual_feats_only_opt_post_pgo_ && !problem.dryRun();
  OptimizationScopeParams scope_vf = optimization_scope_params;
  scope_vf.fix_poses_ = true; scope_vf.fix_objects_ = true; scope_vf.include_object_factors_ = false;
  obvi::Problem vf_problem(device_id, problem.dryRun());
  ObjectPoseGraphOptimizer vf_optimizer;
  BesideThread vf_thread;
  bool vf_planned = false;
  BesideSolve beside_pgo;
  if (vf_beside) {
    beside_pgo.start = [&]() {
      vf_thread.post([&]() {
        std::optional<OptimizationLogger> no_logger;
        vf_optimizer.buildPoseGraphOptimization(scope_vf, residual_params, pose_graph, &vf_problem, no_logger);
        vf_planned = vf_optimizer.uploadAndPlanAhead(&vf_problem);
      });
    };
    beside_pgo.join = [&]() { vf_thread.wait(); };
  }
  if (!optimizer.solveOptimization(&problem, final_run ? pgo_solver_params.final_pgo_optimization_solver_params_ : pgo_solver_params.pgo_optimization_solver_params_,
                                   opt_logger, nullptr, nullptr, nullptr, nullptr, false, vf_beside ? &beside_pgo : nullptr)) {   // :221-232
    std::cerr << "Pose-graph + object optimization failed at max frame id " << max_frame_id << std::endl;
    return false;
  }
  if (opt_logger.has_value()) opt_logger->writeCurrentOptInfo();
  lap("solve (pose graph + objects)");
  if (pgo_solver_params.enable_visual_non_opt_feature_adjustment_post_pgo_) {                                                // :238-283
    pose_graph->getRobotPoseEstimates(raw);
    cache_poses();
    for (const RelativeToFirst& f : relative_positions_from_first) {
      if (!pose_of_frame[f.first].have) continue;
      const Mat3& R = pose_of_frame[f.first].R; const Position3d& t = pose_of_frame[f.first].t; const Position3d& p = f.relative;
      double* pos = pose_graph->featurePointerOfSlot(f.slot);   // combinePoseAndPosition
      for (int i = 0; i < 3; ++i) pos[i] = t[i] + R[3 * i] * p[0] + R[3 * i + 1] * p[1] + R[3 * i + 2] * p[2];
    }
  }
  lap("features follow their first frame");
  if (pgo_solver_params.enable_visual_feats_only_opt_post_pgo_) {                                                            // :284-350
    std::optional<OptimizationLogger> null_logger;
    obvi::Problem* vf = &problem;
    if (vf_beside && vf_planned) { optimizer.adoptBuild(vf_optimizer, null_logger); vf = &vf_problem; problem.parkHandle(); }
    else optimizer.buildPoseGraphOptimization(scope_vf, residual_params, pose_graph, &problem, null_logger);
    lap(vf_beside && vf_planned ? "build (features only): done beside the pose-graph solve" : "build (features only)");
    if (!optimizer.solveOptimization(vf, final_run ? pgo_solver_params.final_post_pgo_vf_adjustment_solver_params_ : pgo_solver_params.post_pgo_vf_adjustment_solver_params_,
                                     null_logger)) {
      std::cerr << "Visual feature adjustment after pose-graph optimization failed at max frame id " << max_frame_id << std::endl;
      return false;
    }
    // :338-345
    const std::shared_ptr<IterationLogger> vf_adjust_logger = IterationLoggerFactory::getInstance().getOrCreateLoggerOfType(IterationLoggerFactory::kVfAdjustOptimizationType);
    if (vf_adjust_logger != nullptr) vf_adjust_logger->logIterations(std::to_string(max_frame_id) + "_" + std::to_string(attempt_num), optimizer.lastSummary());
    lap("solve (features only)");
  }
  return true;
}

}  // namespace pose_graph_optimizer
#endif  // OBVI_HOST_OPTIMIZER_H_
