// obvi_runner.h -- host-side mirror of the session loop and the two-phase controller:
//   OfflineProblemRunner::runOptimization / runOptimizationIteration
//       include/refactoring/offline/offline_problem_runner.h:100-274, 376-916
// The runner is parameterised like the reference's (window provider, GBA checker, iteration-parameter
// provider, frame data adder); numeric work goes through ObjectPoseGraphOptimizer -> include/obvi_ba.h.
#ifndef OBVI_HOST_RUNNER_H_
#define OBVI_HOST_RUNNER_H_

#include <chrono>
#include <functional>
#include <memory>

#include "obvi_optimizer.h"

namespace vslam_types_refactor {

// Input container (the role of UnassociatedBoundingBoxOfflineProblemData, offline_problem_data.h:110-399, reduced
// to what reaches the optimisation path: associations are given, the front ends are out of scope).
struct OfflineProblemData {
  std::unordered_map<CameraId, CameraIntrinsicsMat> camera_intrinsics_by_camera_;
  std::unordered_map<CameraId, CameraExtrinsics> camera_extrinsics_by_camera_;
  std::vector<Pose3D> robot_poses_;                         // initial trajectory estimate (odometry), index = FrameId
  std::unordered_map<FeatureId, Position3d> initial_feature_positions_;
  struct VisualObs { FeatureId feature_id; CameraId camera_id; PixelCoord pixel; };
  std::vector<std::vector<VisualObs>> visual_obs_by_frame_;
  struct BoxObs { ObjectId object_id; CameraId camera_id; BbCorners corners; Covariance<4> cov; };
  std::vector<std::vector<BoxObs>> box_obs_by_frame_;
  std::unordered_map<ObjectId, RawEllipsoid> initial_ellipsoids_;
  std::unordered_map<ObjectId, std::string> object_class_;
  std::unordered_map<std::string, std::pair<ObjectDim, Covariance<3>>> shape_priors_by_class_;
  std::vector<LongTermMapObjectPrior> long_term_map_;       // objects carried over from a previous session
  double reprojection_error_std_dev_ = 1.5;
  FrameId getMaxFrameId() const { return robot_poses_.empty() ? 0 : robot_poses_.size() - 1; }
};

typedef ObjectAndReprojectionFeaturePoseGraph MainPg;
typedef std::shared_ptr<MainPg> MainPgPtr;

// The frame data adder of the reference (pose_graph_frame_data_adder.h:8-266) minus the front ends: the new
// frame's pose is the previous *optimised* pose composed with the odometry increment of the initial trajectory;
// features / objects enter the graph the first time they are observed; a consecutive-frame odometry factor is
// stored for every frame (used by buildPoseGraphOptimization only for frames with few observations).
// `visual_feature_adder` (optional): the visual-feature front end (obvi_visual_feature_front_end.h) decides which of the frame's
// observations and features enter the graph (pose_graph_frame_data_adder.h:75-82); without it every observation is added.
typedef std::function<bool(const OfflineProblemData&, const MainPgPtr&, const FrameId& /*min_frame_id*/, const FrameId& /*max_frame_id*/)> VisualFeatureAdder;
inline void addFrameDataToPoseGraph(const OfflineProblemData& d, const MainPgPtr& pg, const FrameId& frame,
                                    const pose_graph_optimization::RelativePoseCovarianceOdomModelParams& odom,
                                    const VisualFeatureAdder& visual_feature_adder = nullptr, const FrameId& min_frame_id = 0, bool provisional_pose = false) {
  // provisional_pose: the previous frame is still being optimised (the runner adds this frame's data beside that solve and reads no value of the graph);
  // the caller sets the pose once the previous one exists
  if (frame == 0) {
    pg->addFrame(0, d.robot_poses_[0]);
  } else {
    const Pose3D rel = getPose2RelativeToPose1(d.robot_poses_[frame - 1], d.robot_poses_[frame]);
    if (provisional_pose) pg->addFrame(frame, d.robot_poses_[frame]);
    else pg->addFrame(frame, combinePoses(convertToPose3D(pg->getRobotPose(frame - 1).value()), rel));
    RelPoseFactor f;
    f.frame_id_1_ = frame - 1; f.frame_id_2_ = frame; f.measured_pose_deviation_ = rel;
    f.pose_deviation_cov_ = generateOdomCov(rel, odom.transl_error_mult_for_transl_error_, odom.transl_error_mult_for_rot_error_,
                                            odom.rot_error_mult_for_transl_error_, odom.rot_error_mult_for_rot_error_);
    pg->addPoseFactor(f);
  }
  if (visual_feature_adder) {
    if (!visual_feature_adder(d, pg, min_frame_id, frame)) std::cerr << "visual feature front end failed at frame " << frame << std::endl;
  } else if (frame < d.visual_obs_by_frame_.size())
    for (const auto& o : d.visual_obs_by_frame_[frame]) {
      if (!pg->hasFeature(o.feature_id)) pg->addFeature(o.feature_id, d.initial_feature_positions_.at(o.feature_id));
      pg->addVisualFactor(ReprojectionErrorFactor{frame, o.feature_id, o.camera_id, o.pixel, d.reprojection_error_std_dev_});
    }
  if (frame < d.box_obs_by_frame_.size())
    for (const auto& o : d.box_obs_by_frame_[frame]) {
      double* p = nullptr;
      if (!pg->getObjectParamPointers(o.object_id, &p)) {
        // objects are created with consecutive ids in first-observation order: the scene must be numbered that way
        const ObjectId id = pg->addNewEllipsoid(d.initial_ellipsoids_.at(o.object_id), d.object_class_.at(o.object_id));
        if (id != o.object_id) std::cerr << "object id mismatch " << id << " vs " << o.object_id << std::endl;
        const auto& pr = d.shape_priors_by_class_.at(d.object_class_.at(o.object_id));
        pg->addShapeDimPrior(ShapeDimPriorFactor{o.object_id, pr.first, pr.second});
      }
      pg->addObjectObservation(ObjectObservationFactor{frame, o.camera_id, o.object_id, o.corners, o.cov, 1.0});
    }
}

// identifyMergeObjectsBasedOnCenterProximity (bounding_box_front_end_helpers.h:266-356): objects of one semantic class whose centres
// are at most max_distance_for_merge apart are merged pairwise, closest pairs first, every object in at most one merge; a
// long-term-map object is the one merged into (two long-term-map objects are never merged); otherwise the second of the pair.
inline void identifyMergeObjectsBasedOnCenterProximity(const MainPgPtr& pose_graph, const double& max_distance_for_merge, const bool& x_y_only_merge,
                                                       std::unordered_map<ObjectId, std::unordered_set<ObjectId>>& merge_results) {
  if (max_distance_for_merge < 0) return;
  std::unordered_set<ObjectId> long_term_map_objects, involved_in_merge;
  pose_graph->getLongTermMapObjects(long_term_map_objects);
  std::unordered_map<ObjectId, std::pair<std::string, RawEllipsoid>> object_estimates_raw;
  pose_graph->getObjectEstimates(object_estimates_raw);
  std::map<std::string, std::vector<std::pair<ObjectId, Position3d>>> object_centers_by_semantic_class;
  std::vector<ObjectId> ids;
  for (const auto& e : object_estimates_raw) ids.push_back(e.first);
  std::sort(ids.begin(), ids.end());   // the reference walks an unordered_map; ascending ids make the pairing reproducible
  for (ObjectId id : ids) { const auto& e = object_estimates_raw.at(id); object_centers_by_semantic_class[e.first].push_back({id, Position3d{{e.second[0], e.second[1], e.second[2]}}}); }
  std::vector<std::pair<double, std::pair<ObjectId, ObjectId>>> possible_merge_objs_with_dist;
  for (const auto& cls : object_centers_by_semantic_class)
    for (size_t i = 0; i < cls.second.size(); ++i)
      for (size_t j = i + 1; j < cls.second.size(); ++j) {
        const Position3d &a = cls.second[i].second, &b = cls.second[j].second;
        const double dz = x_y_only_merge ? 0.0 : a[2] - b[2];
        const double dist = std::sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + dz * dz);
        if (dist <= max_distance_for_merge && (!long_term_map_objects.count(cls.second[i].first) || !long_term_map_objects.count(cls.second[j].first)))
          possible_merge_objs_with_dist.push_back({dist, {cls.second[i].first, cls.second[j].first}});
      }
  std::stable_sort(possible_merge_objs_with_dist.begin(), possible_merge_objs_with_dist.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
  for (const auto& c : possible_merge_objs_with_dist) {
    if (involved_in_merge.count(c.second.first) || involved_in_merge.count(c.second.second)) continue;
    involved_in_merge.insert(c.second.first); involved_in_merge.insert(c.second.second);
    if (long_term_map_objects.count(c.second.first)) merge_results[c.second.first] = {c.second.second};
    else merge_results[c.second.second] = {c.second.first};
  }
}

struct OptimizationRecord {   // one row per solve, for tests / logging
  FrameId min_frame, max_frame; std::string kind; int iterations; double initial_cost, final_cost; size_t n_poses, n_features, n_objects, n_excluded;
};

// an object of the long-term map as the extraction hands it over: estimate + its 7x7 marginal covariance
struct LongTermMapEntry { ObjectId object_id_; std::array<double, 7> ellipsoid_mean_; std::array<double, 49> covariance_; };

// offline_problem_runner.h:17-25
enum VisualizationTypeEnum { BEFORE_ANY_OPTIMIZATION, BEFORE_EACH_OPTIMIZATION, AFTER_EACH_OPTIMIZATION, AFTER_PGO_PLUS_OBJ_OPTIMIZATION, AFTER_ALL_OPTIMIZATION, AFTER_ALL_POSTPROCESSING };
// limit_trajectory_evaluation_params.h:15-30
struct LimitTrajectoryEvaluationParams {
  bool should_limit_trajectory_evaluation_ = false;
  FrameId max_frame_id_ = 0;   // ignored if should_limit_trajectory_evaluation_ is false
  bool operator==(const LimitTrajectoryEvaluationParams& rhs) const { return should_limit_trajectory_evaluation_ == rhs.should_limit_trajectory_evaluation_ && max_frame_id_ == rhs.max_frame_id_; }
  bool operator!=(const LimitTrajectoryEvaluationParams& rhs) const { return !operator==(rhs); }
};

// OfflineProblemRunner comes in two shapes (one name: a variadic primary with two specialisations).
//   OfflineProblemRunner<OutputProblemData>: the input, factor, cache and pose-graph types this path has fixed; the output type and its extractor stay the
//     caller's, as in the reference (offline_problem_runner.h:63-67, :100-107, :269-273).  What the driver and runFullOptimization use.
//   OfflineProblemRunner<InputProblemData, VisualFeatureFactorType, OutputProblemData, CachedFactorInfo, PoseGraphType>: the reference's own template parameter
//     list and its fifteen-argument constructor (:27-98), below.
template <typename... Ts> class OfflineProblemRunner;

template <typename OutputProblemData>
class OfflineProblemRunner<OutputProblemData> {
 public:
  using LongTermMapEntry = vslam_types_refactor::LongTermMapEntry;
  using OutputDataExtractor = std::function<void(const OfflineProblemData&, const MainPgPtr&, const pose_graph_optimizer::OptimizationFactorsEnabledParams&, OutputProblemData&)>;
  OfflineProblemRunner(const pose_graph_optimization::ObjectVisualPoseGraphResidualParams& residual_params,
                       const pose_graph_optimization::PoseGraphPlusObjectsOptimizationParams& pgo_solver_params,
                       const std::function<FrameId(const FrameId&)>& window_provider_func,
                       const OutputDataExtractor& output_data_extractor,
                       const std::function<bool(const FrameId&)>& gba_checker,
                       const std::function<pose_graph_optimization::OptimizationIterationParams(const FrameId&)>& iteration_params_provider_func,
                       int device_id = 0)
      : residual_params_(residual_params), pgo_solver_params_(pgo_solver_params), window_provider_func_(window_provider_func), output_data_extractor_(output_data_extractor),
        gba_checker_(gba_checker), iteration_params_provider_func_(iteration_params_provider_func), device_id_(device_id) {}

  // offline_problem_runner.h:100-274
  bool runOptimization(const OfflineProblemData& problem_data, const pose_graph_optimizer::OptimizationFactorsEnabledParams& enabled,
                       std::optional<OptimizationLogger>& opt_logger, OutputProblemData& output_problem_data, const FrameId& start_at_frame = 0,
                       const bool& add_data_for_starting_frame = true) {
    pose_graph_.reset();
    if (opt_logger.has_value()) opt_logger->writeOptInfoHeader();
    // two problem objects (two device handles): the session's current one and the one the NEXT window is planned on beside this window's last solve
    obvi::Problem problem_objects[2] = {obvi::Problem(device_id_), obvi::Problem(device_id_)};
    int current = 0;
    ahead_ = Ahead();
    pose_graph_optimizer::OptimizationScopeParams scope;                                                                     // :115-142
    scope.min_low_level_feature_observations_per_frame_ = enabled.min_low_level_feature_observations_per_frame_;
    scope.fix_poses_ = enabled.fix_poses_; scope.fix_objects_ = enabled.fix_objects_; scope.fix_visual_features_ = enabled.fix_visual_features_;
    scope.fix_ltm_objects_ = enabled.fix_ltm_objects_; scope.include_visual_factors_ = enabled.include_visual_factors_;
    scope.include_object_factors_ = enabled.include_object_factors_; scope.use_pom_ = enabled.use_pom_;
    scope.poses_prior_to_window_to_keep_constant_ = enabled.poses_prior_to_window_to_keep_constant_;
    scope.min_low_level_feature_observations_ = enabled.min_low_level_feature_observations_;
    scope.min_object_observations_ = enabled.min_object_observations_;
    FrameId max_frame_id = problem_data.getMaxFrameId();
    if (limit_trajectory_eval_params_.should_limit_trajectory_evaluation_) max_frame_id = std::min(limit_trajectory_eval_params_.max_frame_id_, max_frame_id);   // :143-147
    current_problem_data_ = &problem_data;
    MainPgPtr pose_graph;
    if (pose_graph_creator_) pose_graph_creator_(problem_data, pose_graph);                                                  // :149-150 (a checkpoint's graph: run_opt_from_pg_state.cpp:182-185)
    else pose_graph = std::make_shared<MainPg>(problem_data.camera_extrinsics_by_camera_, problem_data.camera_intrinsics_by_camera_);
    if (!pose_graph) return false;
    for (const auto& ltm : problem_data.long_term_map_)
      pose_graph->addLongTermMapObject(ltm.object_id_, ltm.ellipsoid_mean_, problem_data.object_class_.count(ltm.object_id_) ? problem_data.object_class_.at(ltm.object_id_) : "", ltm);
    if (start_at_frame == 0 && add_data_for_starting_frame) {                                                                // :160-162
      if (frame_data_adder_) frame_data_adder_(problem_data, pose_graph, 0, 0);
      else addFrameDataToPoseGraph(problem_data, pose_graph, 0, residual_params_.relative_pose_cov_params_, visual_feature_adder_, 0);
    }
    visualize(pose_graph, 0, max_frame_id, BEFORE_ANY_OPTIMIZATION, 0);                                                       // :164-169
    const FrameId first_frame = std::max<FrameId>(1, start_at_frame);
    for (FrameId next_frame_id = first_frame; next_frame_id <= max_frame_id; ++next_frame_id) {                               // :174-226
      if (have_continue_opt_checker_ && !continue_opt_checker_) { std::cerr << "Halted optimization due to continue checker reporting false" << std::endl; return false; }   // :183-187
      const FrameId start_opt_with_frame = window_provider_func_(next_frame_id);
      scope.min_frame_id_ = start_opt_with_frame; scope.max_frame_id_ = next_frame_id;
      const auto t_add0 = std::chrono::steady_clock::now();
      const bool planned_ahead = ahead_.valid && ahead_.frame == next_frame_id && ahead_.start == start_opt_with_frame;
      if (planned_ahead) {
        // the frame's data entered the graph beside the previous window's last solve, with a provisional pose: now that the previous pose is optimised, the
        // pose the adder would have given it (pose_graph_frame_data_adder.h:30-45; same arithmetic as addFrameDataToPoseGraph)
        const Pose3D rel = getPose2RelativeToPose1(problem_data.robot_poses_[next_frame_id - 1], problem_data.robot_poses_[next_frame_id]);
        const RawPose3d init = convertPoseToArray(combinePoses(convertToPose3D(pose_graph->getRobotPose(next_frame_id - 1).value()), rel));
        double* pose_ptr = nullptr;
        if (!pose_graph->getPosePointers(next_frame_id, &pose_ptr)) return false;
        std::copy_n(init.data(), 6, pose_ptr);
        current = 1 - current;   // the problem that was planned ahead is the session's problem now
      } else if (next_frame_id != start_at_frame || add_data_for_starting_frame) {                                           // :196-199
        if (ahead_.valid) { std::cerr << "a window planned ahead was not used (frame " << ahead_.frame << ")" << std::endl; return false; }   // its frame data is in the graph already
        if (frame_data_adder_) frame_data_adder_(problem_data, pose_graph, start_opt_with_frame, next_frame_id);
        else addFrameDataToPoseGraph(problem_data, pose_graph, next_frame_id, residual_params_.relative_pose_cov_params_, visual_feature_adder_, start_opt_with_frame);
      }
      time_add_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_add0).count();
      obvi::Problem& problem = problem_objects[current];
      if (gba_checker_(next_frame_id)) problem_objects[1 - current].parkHandle();   // a global-BA iteration runs its pose-graph stage on a problem of its own: this handle is free for it
      // Plan the next window beside this one's solves?  Its STRUCTURE (frame data, window, factor selection) depends on nothing this window computes; its
      // start values do, and are handed over when they exist.  Not when a hook could see or change the graph in between (a custom adder, the visual front end,
      // a visualization callback), not across a global-BA frame (that iteration starts with other problems), not with the phase-II cross-check.
      ahead_job_ = nullptr;
      const FrameId ahead_frame = next_frame_id + 1;
      static const bool phase_two_check = std::getenv("OBVI_HOST_PHASE2_CHECK") && std::atoi(std::getenv("OBVI_HOST_PHASE2_CHECK")) != 0;
      if (planAheadEnabled() && !phase_two_check && ahead_frame <= max_frame_id && !frame_data_adder_ && !visual_feature_adder_ && !visualization_callback_ && !factor_hooks_set_ && !gba_checker_(ahead_frame)) {
        obvi::Problem* ahead_problem = &problem_objects[1 - current];
        ahead_job_ = [this, &problem_data, &pose_graph, ahead_frame, ahead_problem, scope]() {
          const auto t0 = std::chrono::steady_clock::now();
          Ahead a;
          a.frame = ahead_frame; a.start = window_provider_func_(ahead_frame);
          addFrameDataToPoseGraph(problem_data, pose_graph, ahead_frame, residual_params_.relative_pose_cov_params_, nullptr, a.start, /*provisional_pose=*/true);
          const auto t1 = std::chrono::steady_clock::now();
          pose_graph_optimizer::OptimizationScopeParams ahead_scope = scope;
          ahead_scope.min_frame_id_ = a.start; ahead_scope.max_frame_id_ = ahead_frame;
          std::optional<OptimizationLogger> null_logger;
          a.block_info = ahead_optimizer_.buildPoseGraphOptimization(ahead_scope, residual_params_, pose_graph, ahead_problem, null_logger);
          const auto t2 = std::chrono::steady_clock::now();
          ahead_graph_done_.store(true, std::memory_order_release);   // from here on the job works on the flat problem and the library only
          a.uploaded = ahead_optimizer_.uploadAndPlanAhead(ahead_problem);
          a.valid = true;
          ahead_ = std::move(a);
          const auto t3 = std::chrono::steady_clock::now();
          const auto ms = [](auto x, auto y) { return std::chrono::duration<double, std::milli>(y - x).count(); };
          time_ahead_ms_ += ms(t0, t3); time_ahead_add_ms_ += ms(t0, t1); time_ahead_build_ms_ += ms(t1, t2); time_ahead_upload_ms_ += ms(t2, t3); ++n_ahead_;
        };
      }
      if (!runOptimizationIteration(start_opt_with_frame, next_frame_id, enabled, scope, max_frame_id, opt_logger, pose_graph, problem, 0, planned_ahead)) return false;
      IterationLoggerFactory::getInstance().writeAllIterationLoggerStates();                                                 // :219
    }
    if (ahead_.valid) { std::cerr << "a window planned ahead was not used (frame " << ahead_.frame << ")" << std::endl; return false; }
    obvi::Problem& problem = problem_objects[current];
    problem_objects[1 - current].parkHandle();
    if (!runOptimizationIteration(0, max_frame_id, enabled, scope, max_frame_id, opt_logger, pose_graph, problem, 1)) return false;   // :232-243
    visualize(pose_graph, 0, max_frame_id, AFTER_ALL_OPTIMIZATION, 1);                                                        // :245-250
    if (!mergeObjectsAtSessionEnd(max_frame_id, enabled, scope, opt_logger, pose_graph, problem)) return false;              // :254-262
    visualize(pose_graph, 0, max_frame_id, AFTER_ALL_POSTPROCESSING, 1);                                                      // :263-268
    IterationLoggerFactory::getInstance().writeAllIterationLoggerStates();
    // The output extractor's covariance
    // step (IndependentEllipsoidsLongTermObjectMapExtractor::extractLongTermObjectMap, long_term_object_map_extraction.h:
    // 381-527: marginal covariance of every ellipsoid from the final problem) runs on the device.
    long_term_map_.clear();
    if (extract_long_term_map_ && !problem.flat.objects.empty()) {
      std::vector<std::pair<ObjectId, ObjectId>> blocks;
      for (const ObjectId& o : problem.flat.objects) blocks.push_back({o, o});
      obvi::Covariance covariance;
      covariance_rank_repairs_.clear();
      if (!obvi::extractCovarianceWithRankDeficiencyHandling(blocks, &problem, ltm_tunable_params_.min_col_norm_, &covariance, &covariance_rank_repairs_)) return false;
      for (size_t i = 0; i < problem.flat.objects.size(); ++i) {
        LongTermMapEntry e;
        e.object_id_ = problem.flat.objects[i];
        std::copy_n(problem.flat.object_ptrs[i], 7, e.ellipsoid_mean_.begin());
        if (!covariance.GetCovarianceBlock(e.object_id_, e.object_id_, e.covariance_.data())) return false;
        long_term_map_.push_back(e);
      }
    }
    pose_graph_ = pose_graph;
    if (output_data_extractor_) output_data_extractor_(problem_data, pose_graph, enabled, output_problem_data);   // :269-272
    return true;
  }
  // the session's pose graph after runOptimization (what the extractor was given): checkpoint writers
  const MainPgPtr& poseGraph() const { return pose_graph_; }
  // :918-958 until there are no more objects to merge, merge and re-run the final optimisation (attempt numbers 2, 3, ...)
  bool mergeObjectsAtSessionEnd(const FrameId& max_frame_id, const pose_graph_optimizer::OptimizationFactorsEnabledParams& enabled,
                                const pose_graph_optimizer::OptimizationScopeParams& scope, std::optional<OptimizationLogger>& opt_logger, MainPgPtr& pose_graph,
                                obvi::Problem& merged_problem /* the reference builds a fresh ceres::Problem per round (:935); here the session's problem object is rebuilt
                                                                 in place, so that the long-term-map extraction afterwards sees the merged problem */) {
    if (!object_merger_) return true;
    int post_process_round = 2;
    while (object_merger_(pose_graph)) {
      optimizer_.clearPastOptimizationData();
      pose_graph_optimizer::OptimizationScopeParams merged_scope = scope;
      merged_scope.min_frame_id_ = 0; merged_scope.max_frame_id_ = max_frame_id;
      if (!runOptimizationIteration(0, max_frame_id, enabled, merged_scope, max_frame_id, opt_logger, pose_graph, merged_problem, post_process_round)) return false;
      ++n_merge_rounds_;
      post_process_round++;
    }
    return true;
  }
  void setPoseGraphCreator(const std::function<void(const OfflineProblemData&, MainPgPtr&)>& creator) { pose_graph_creator_ = creator; }
  // the reference's remaining constructor hooks (offline_problem_runner.h:27-98), optional here:
  //   frame data adder (:47-51): replaces the built-in addFrameDataToPoseGraph; called with (data, graph, first frame of the window, new frame), (.., 0, 0) for frame 0
  //   visualization callback (:62-67): called where the reference calls it (:164, :392, :512, :908, :245, :263)
  //   limit on the evaluated trajectory (:143-147)
  //   continue-optimisation checker (:183-187): the reference tests the std::function OBJECT (`if (!continue_opt_checker_)`), never calls it; so does this
  using FrameDataAdder = std::function<void(const OfflineProblemData&, const MainPgPtr&, const FrameId&, const FrameId&)>;
  using VisualizationCallback = std::function<void(const OfflineProblemData&, const MainPgPtr&, const FrameId&, const FrameId&, const VisualizationTypeEnum&, const int&)>;
  void setFrameDataAdder(const FrameDataAdder& adder) { frame_data_adder_ = adder; }
  void setVisualizationCallback(const VisualizationCallback& cb) { visualization_callback_ = cb; }
  void setLimitTrajectoryEvaluationParams(const LimitTrajectoryEvaluationParams& p) { limit_trajectory_eval_params_ = p; }
  void setContinueOptChecker(const std::function<bool()>& checker) { continue_opt_checker_ = checker; have_continue_opt_checker_ = true; }
  void setObjectMerger(const std::function<bool(const MainPgPtr&)>& merger) { object_merger_ = merger; }
  // the per-factor seam (obvi_optimizer.h FactorHooks): the hooks run on the thread that builds, so the next window is then not planned on a second thread
  void setFactorHooks(const pose_graph_optimizer::FactorHooks& hooks) { optimizer_.setFactorHooks(hooks); ahead_optimizer_.setFactorHooks(hooks); factor_hooks_set_ = (bool)hooks; }
  size_t factorsLeftOutByTheCreator() const { return optimizer_.factorsLeftOutByTheCreator() + ahead_optimizer_.factorsLeftOutByTheCreator(); }
  size_t mergeRounds() const { return n_merge_rounds_; }
  const std::vector<OptimizationRecord>& records() const { return records_; }
  void printTiming(std::ostream& os) const {
    optimizer_.printTiming(os);
    if (n_iterations_) os << "runOptimizationIteration x" << n_iterations_ << ": phase-I build " << time_build_ms_ / n_iterations_ << " ms, pose-graph copy " << time_copy_ms_ / n_iterations_ << " ms per call; phase II on the phase-I problem (masks) x"
                          << n_phase_two_masked_ << ", rebuilt x" << n_phase_two_rebuilt_ << std::endl;
    if (n_ahead_) os << "windows planned ahead (frame data, build, upload, symbolic phase beside the previous window's last solve) x" << n_ahead_ << ": " << time_ahead_ms_ / n_ahead_
                     << " ms each on the second thread (frame data " << time_ahead_add_ms_ / n_ahead_ << ", build " << time_ahead_build_ms_ / n_ahead_ << ", upload + symbolic phase "
                     << time_ahead_upload_ms_ / n_ahead_ << "), " << optimizer_.besideWaitMs() / n_ahead_ << " ms of it after the solve had ended" << std::endl;
    if (n_stage_beside_) os << "global BAs built, uploaded and planned beside their pose-graph + object stage x" << n_stage_beside_ << ": " << time_stage_beside_ms_ / n_stage_beside_
                            << " ms each on the second thread, waited for " << time_stage_beside_wait_ms_ / n_stage_beside_ << " ms after the stage" << std::endl;
    if (n_pgo_) os << "pose-graph + object stages at the global-BA frames x" << n_pgo_ << ": " << time_pgo_ms_ / n_pgo_ << " ms per call (own problem: handle, build, solves)" << std::endl;
    if (n_iterations_) os << "frame data adder " << time_add_ms_ / n_iterations_ << " ms, outlier selection on the host " << time_select_ms_ / n_iterations_ << " ms per frame" << std::endl;
    if (check_.windows) os << "phase2_check windows " << check_.windows << " failures " << check_.failures << " iteration_mismatches " << check_.iteration_mismatches << " size_mismatches "
                           << check_.size_mismatches << " max_initial_cost_rel " << check_.max_initial_cost_rel << " max_final_cost_rel " << check_.max_final_cost_rel << " max_value_diff "
                           << check_.max_value_diff << " points " << check_.points << " points_apart " << check_.points_apart << " objects " << check_.objects << " objects_apart " << check_.objects_apart << std::endl;
  }
  void setVisualFeatureAdder(const VisualFeatureAdder& adder) { visual_feature_adder_ = adder; }
  void setExtractLongTermMap(bool on) { extract_long_term_map_ = on; }
  void setLongTermMapTunableParams(const LongTermMapExtractionTunableParams& p) { ltm_tunable_params_ = p; }
  const std::vector<obvi::CovarianceRankRepair>& covarianceRankRepairs() const { return covariance_rank_repairs_; }
  const std::vector<LongTermMapEntry>& longTermMap() const { return long_term_map_; }

 private:
  // offline_problem_runner.h:337-374
  static bool isConsecutivePosesStable_(const MainPgPtr& pg, const FrameId& min_f, const FrameId& max_f, const double& transl_tol, const double& orient_tol) {
    for (FrameId f = min_f + 1; f <= max_f; ++f) {
      const auto a = pg->getRobotPose(f - 1), b = pg->getRobotPose(f);
      if (!a.has_value() || !b.has_value()) continue;
      const Pose3D rel = getPose2RelativeToPose1(convertToPose3D(a.value()), convertToPose3D(b.value()));
      const double tn = std::sqrt(rel.transl_[0] * rel.transl_[0] + rel.transl_[1] * rel.transl_[1] + rel.transl_[2] * rel.transl_[2]);
      const double an = std::sqrt(rel.orientation_[0] * rel.orientation_[0] + rel.orientation_[1] * rel.orientation_[1] + rel.orientation_[2] * rel.orientation_[2]);
      if (tn > transl_tol || std::fabs(an) > orient_tol) return false;
    }
    return true;
  }

  // values of the parameter blocks of a flattened problem (makeCopyDeepCopyValues / setValuesFromAnotherPoseGraph restricted to them)
  struct ValueSnapshot { std::vector<double*> ptrs[3]; std::vector<double> values[3]; };
  static ValueSnapshot snapshotValues(const obvi::FlatProblem& fp) {
    ValueSnapshot s;
    const std::vector<double*>* src[3] = {&fp.pose_ptrs, &fp.point_ptrs, &fp.object_ptrs};
    const int dim[3] = {6, 3, 7};
    for (int k = 0; k < 3; ++k) {
      s.ptrs[k] = *src[k];
      s.values[k].resize(src[k]->size() * dim[k]);
      for (size_t i = 0; i < src[k]->size(); ++i) std::copy_n((*src[k])[i], dim[k], &s.values[k][dim[k] * i]);
    }
    return s;
  }
  static void restoreValues(const ValueSnapshot& s) {
    const int dim[3] = {6, 3, 7};
    for (int k = 0; k < 3; ++k) for (size_t i = 0; i < s.ptrs[k].size(); ++i) std::copy_n(&s.values[k][dim[k] * i], dim[k], s.ptrs[k][i]);
  }

  // offline_problem_runner.h:376-916
  bool runOptimizationIteration(const FrameId& start_opt_with_frame, const FrameId& next_frame_id, const pose_graph_optimizer::OptimizationFactorsEnabledParams& enabled,
                                const pose_graph_optimizer::OptimizationScopeParams& scope, const FrameId& max_frame_id, std::optional<OptimizationLogger>& opt_logger,
                                MainPgPtr& pose_graph, obvi::Problem& problem, const int& attempt_num = 0, bool planned_ahead = false) {   // (planned_ahead: by value, set below at a global-BA frame)
    // planned_ahead: `problem` holds this window's build already, uploaded with its symbolic plan (runOptimization)
    const pose_graph_optimization::OptimizationIterationParams iteration_params = iteration_params_provider_func_(next_frame_id);
    visualize(pose_graph, start_opt_with_frame, next_frame_id, BEFORE_EACH_OPTIMIZATION, attempt_num);                        // :392-397
    if (opt_logger.has_value()) opt_logger->setOptimizationTypeParams(next_frame_id, start_opt_with_frame == 0, false, false, attempt_num);
    const bool global_ba = gba_checker_(next_frame_id);                                                                      // :407
    bool run_visual_feature_opt = true;
    if (global_ba) {                                                                                                         // :410-520
      bool run_pgo;
      if (next_frame_id == max_frame_id && attempt_num > 0) {
        run_pgo = enabled.use_pose_graph_on_final_global_ba_;
        if (run_pgo) run_visual_feature_opt = enabled.use_visual_features_on_final_global_ba_;
      } else {
        run_pgo = enabled.use_pose_graph_on_global_ba_;
        if (run_pgo) run_visual_feature_opt = enabled.use_visual_features_on_global_ba_;
      }
      if (run_pgo) {
        pose_graph_optimizer::OptimizationScopeParams tracking = scope;                                                      // :440-496 "tracking before PGO"
        tracking.min_frame_id_ = next_frame_id - std::min<FrameId>(next_frame_id, tracking.poses_prior_to_window_to_keep_constant_);
        std::optional<OptimizationLogger> null_logger;
        optimizer_.buildPoseGraphOptimization(tracking, residual_params_, pose_graph, &problem, null_logger);
        if (!optimizer_.solveOptimization(&problem, pgo_solver_params_.pre_pgo_tracking_solver_params_, null_logger)) std::cerr << "Tracking failed" << std::endl;
        {   // :486-496
          const std::shared_ptr<IterationLogger> tracking_logger = IterationLoggerFactory::getInstance().getOrCreateLoggerOfType(IterationLoggerFactory::kPrePgoTrackOptimizationType);
          if (tracking_logger != nullptr) tracking_logger->logIterations(std::to_string(next_frame_id), optimizer_.lastSummary());
        }
        record("pre_pgo_track", tracking.min_frame_id_, next_frame_id, problem, 0);
        const auto t_pgo0 = std::chrono::steady_clock::now();
        // The global BA that follows the stage has the whole trajectory's structure to flatten, upload and plan (config #3: 89 + 11 + 38 ms), none of which
        // depends on what the stage computes: a second thread does it on this problem's handle while the stage runs on another one (the session's second
        // problem object parked its handle for that: runOptimization); the values are handed over afterwards.  The stage reads the graph's structure and
        // writes block values; the build reads structure only.
        const bool plan_beside_stage = planAheadEnabled() && run_visual_feature_opt && !planned_ahead && !problem.dryRun() && !factor_hooks_set_;
        if (plan_beside_stage) {
          stage_beside_thread_.post([this, &scope, &pose_graph, &problem]() {
            const auto t0 = std::chrono::steady_clock::now();
            std::optional<OptimizationLogger> null_logger;
            Ahead a;
            a.block_info = ahead_optimizer_.buildPoseGraphOptimization(scope, residual_params_, pose_graph, &problem, null_logger);
            a.uploaded = ahead_optimizer_.uploadAndPlanAhead(&problem);
            a.valid = true;
            ahead_ = std::move(a);
            time_stage_beside_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); ++n_stage_beside_;
          });
        } else {
          problem.parkHandle();   // the stage builds a Problem of its own (:pose_graph_plus_objects_optimizer.h:86): let it have this one's device handle
        }
        if (!pose_graph_optimizer::runPgoPlusEllipsoids(next_frame_id, scope, residual_params_, pgo_solver_params_, next_frame_id == max_frame_id, opt_logger, pose_graph,
                                                        device_id_, attempt_num))
          std::cerr << "PGO+objs failed at frame " << next_frame_id << std::endl;
        if (plan_beside_stage) {
          const auto t_w0 = std::chrono::steady_clock::now();
          stage_beside_thread_.wait();
          time_stage_beside_wait_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_w0).count();
          planned_ahead = ahead_.valid;
        }
        time_pgo_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_pgo0).count(); ++n_pgo_;
        records_.push_back({0, next_frame_id, "pgo", 0, 0, 0, (size_t)next_frame_id + 1, 0, 0, 0});
        visualize(pose_graph, start_opt_with_frame, next_frame_id, AFTER_PGO_PLUS_OBJ_OPTIMIZATION, attempt_num);             // :512-518
      }
    }
    if (!run_visual_feature_opt) return true;                                                                                // :522
    bool two_phase = iteration_params.feature_outlier_percentage_ > 0;
    const std::string kind = global_ba ? "gba" : "lba";
    // PHASE I  (:541-660)
    const auto t_b0 = std::chrono::steady_clock::now();
    pose_graph_optimizer::ResidualBlockInfoMap block_info;
    if (planned_ahead && ahead_.valid) {
      block_info = std::move(ahead_.block_info);
      optimizer_.adoptBuild(ahead_optimizer_, opt_logger);
      ahead_ = Ahead();
    } else {
      block_info = optimizer_.buildPoseGraphOptimization(scope, residual_params_, pose_graph, &problem, opt_logger);
    }
    // The next window is planned beside this iteration's solves: the job starts with the phase-I solve; its first part (frame data into the graph, build:
    // the graph's structure is written, then read; no value is read) has ended before this thread goes on behind that solve, the rest (upload, symbolic
    // phase: the flat problem and the library only) may run until the last solve has ended.
    std::function<void()> ahead_job; ahead_job.swap(ahead_job_);
    bool ahead_posted = false, ahead_joined = false;
    const std::function<void()> ahead_wait_all = [&]() { if (ahead_posted && !ahead_joined) { ahead_thread_.wait(); ahead_joined = true; } };
    struct JoinAtExit { const std::function<void()>& f; ~JoinAtExit() { f(); } } join_at_exit{ahead_wait_all};
    pose_graph_optimizer::BesideSolve beside_first, beside_last;
    if (ahead_job) {
      beside_first.start = [&]() { ahead_graph_done_.store(false, std::memory_order_release); ahead_thread_.post(ahead_job); ahead_posted = true; };
      beside_first.join = [&]() { while (!ahead_graph_done_.load(std::memory_order_acquire)) std::this_thread::yield(); };
      beside_last.join = ahead_wait_all;
    }
    const auto t_b1 = std::chrono::steady_clock::now();
    // :594 deep-copies the whole pose graph; all that is ever read back (:811, :903) are the values of the parameter blocks this
    // optimisation can move, i.e. the blocks of the phase-I problem: those are saved (the device-side equivalent is obvi_ba_snapshot)
    const ValueSnapshot pose_graph_copy = snapshotValues(problem.flat);
    const auto t_b2 = std::chrono::steady_clock::now();
    time_build_ms_ += std::chrono::duration<double, std::milli>(t_b1 - t_b0).count(); time_copy_ms_ += std::chrono::duration<double, std::milli>(t_b2 - t_b1).count(); ++n_iterations_;
    std::vector<obvi::ResidualBlockId> residual_block_ids;
    std::vector<double> residuals;
    // :689-800 read every residual back, square-sum them per block and pick the top fraction per factor type through a std::map keyed
    // by the value.  obvi_ba_select_outliers makes the same selection on the device (K8; same de-duplication of equal values,
    // tests/test_gpu_parity.py::test_two_phase_outlier_rejection) and hands back one byte per factor instead of every residual.
    // OBVI_HOST_SELECT_ON_HOST=1: the literal host route.
    static const bool select_on_host = std::getenv("OBVI_HOST_SELECT_ON_HOST") && std::atoi(std::getenv("OBVI_HOST_SELECT_ON_HOST")) != 0;
    const bool device_selection = two_phase && !select_on_host;
    const bool ok1 = (two_phase && !device_selection) ? optimizer_.solveOptimization(&problem, iteration_params.phase_one_opt_params_, opt_logger, &residual_block_ids, &residuals)
                                                      : optimizer_.solveOptimization(&problem, iteration_params.phase_one_opt_params_, opt_logger, nullptr, nullptr, nullptr, nullptr, /*keep_for_phase_two=*/two_phase,
                                                                                     &beside_first);
    if (!two_phase) ahead_wait_all();
    if (!ok1) { std::cerr << "Phase I Optimization failed at max frame id " << next_frame_id << std::endl; return false; }
    if (opt_logger.has_value()) opt_logger->writeCurrentOptInfo();
    record(kind + "_phase_1", scope.min_frame_id_, next_frame_id, problem, 0);
    const auto t_sel0 = std::chrono::steady_clock::now();
    FactorInfoSet excluded;                       // the literal route's set; with the device selection it is only materialised when the rebuild needs it
    std::vector<uint8_t> keep_rp, keep_bb;        // device selection: one byte per factor of the flat problem
    size_t n_excluded_device = 0;
    auto materialise_excluded = [&]() {
      const obvi::FlatProblem& fp = problem.flat;
      for (size_t i = 0; i < keep_rp.size(); ++i) if (!keep_rp[i]) excluded.insert(fp.blocks[i]);
      for (size_t i = 0; i < keep_bb.size(); ++i) if (!keep_bb[i]) excluded.insert(fp.blocks[keep_rp.size() + i]);
    };
    if (device_selection) {
      const obvi::FlatProblem& fp = problem.flat;
      keep_rp.assign(fp.rp_pose.size(), 1); keep_bb.assign(fp.bb_obj.size(), 1);
      const struct { int32_t type; std::vector<uint8_t>* keep; } fams[2] = {{OBVI_FACTOR_REPROJECTION, &keep_rp}, {OBVI_FACTOR_BBOX, &keep_bb}};
      for (const auto& fam : fams) {
        if (fam.keep->empty()) continue;
        int64_t n_out = 0;
        if (obvi_ba_select_outliers(problem.handle(), fam.type, iteration_params.feature_outlier_percentage_, fam.keep->data(), &n_out)) { std::cerr << "outlier selection failed: " << obvi_ba_last_error(problem.handle()) << std::endl; return false; }
        n_excluded_device += (size_t)n_out;
      }
    }
    // per-block squared residuals with the hard-coded block sizes (:689-749)
    std::map<FactorType, std::vector<std::pair<double, obvi::ResidualBlockId>>> by_type;
    if (two_phase && !device_selection) {
      size_t idx = 0;
      for (obvi::ResidualBlockId id : residual_block_ids) {
        const FactorType t = id < problem.flat.blocks.size() ? problem.flat.blocks[id].first : block_info.at(id).first;   // block ids index the flat problem's block list
        size_t n;
        if (t == kReprojectionErrorFactorTypeId) n = 2; else if (t == kObjectObservationFactorTypeId) n = 4; else if (t == kShapeDimPriorFactorTypeId) n = 3;
        else if (t == kLongTermMapFactorTypeId) n = kEllipsoidParamterizationSize; else if (t == kPairwiseRobotPoseFactorTypeId) n = 6; else if (t == kPairwiseErrorFactorTypeId) n = 1;
        else { two_phase = false; break; }
        double total = 0;
        for (size_t i = 0; i < n && idx < residuals.size(); ++i, ++idx) total += residuals[idx] * residuals[idx];
        if (t == kReprojectionErrorFactorTypeId || t == kObjectObservationFactorTypeId) by_type[t].push_back({total, id});
      }
      if (idx != residuals.size()) two_phase = false;
    }
    if (two_phase && !device_selection) {                                                                                    // :769-800
      for (auto& tv : by_type) {
        // the reference fills a std::map<double, id, greater> (:769-800): descending by value, equal values collapse into one entry
        // that keeps the id inserted last.  Same list from a stable sort.
        std::vector<std::pair<double, obvi::ResidualBlockId>>& v = tv.second;
        std::stable_sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
        std::vector<obvi::ResidualBlockId> ordered;
        for (size_t i = 0; i < v.size(); ++i) if (i + 1 == v.size() || v[i + 1].first != v[i].first) ordered.push_back(v[i].second);
        const size_t n_outliers = (size_t)(ordered.size() * iteration_params.feature_outlier_percentage_);
        excluded.reserve(excluded.size() + n_outliers);
        for (size_t i = 0; i < n_outliers; ++i) excluded.insert(ordered[i] < problem.flat.blocks.size() ? problem.flat.blocks[ordered[i]] : block_info.at(ordered[i]));
      }
    }
    time_select_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_sel0).count();
    if (two_phase) {                                                                                                         // PHASE II :803-892
      if (opt_logger.has_value()) opt_logger->setOptimizationTypeParams(next_frame_id, start_opt_with_frame == 0, false, true, attempt_num);
      restoreValues(pose_graph_copy);                                                            // :811
      // :812-:830 rebuild the problem with the excluded factors.  The same selection is made on the problem phase I left on the device
      // whenever the rebuild would only remove things (OBVI_HOST_PHASE2_REBUILD=1: always rebuild, as the reference does)
      static const bool always_rebuild = std::getenv("OBVI_HOST_PHASE2_REBUILD") && std::atoi(std::getenv("OBVI_HOST_PHASE2_REBUILD")) != 0;
      pose_graph_optimizer::ObjectPoseGraphOptimizer::PhaseTwoMasks masks;
      bool ok2;
      const bool have_masks = !always_rebuild && optimizer_.excludeFromBuiltProblem(scope, pose_graph, excluded, problem, &masks, device_selection ? &keep_rp : nullptr, device_selection ? &keep_bb : nullptr);
      static const bool check_rebuild = std::getenv("OBVI_HOST_PHASE2_CHECK") && std::atoi(std::getenv("OBVI_HOST_PHASE2_CHECK")) != 0;
      if (device_selection && (!have_masks || check_rebuild)) materialise_excluded();
      if (have_masks) {
        optimizer_.setPhaseTwoLogCounts(masks, opt_logger);
        ok2 = optimizer_.solveOptimization(&problem, iteration_params.phase_two_opt_params_, opt_logger, nullptr, nullptr, nullptr, &masks, false, &beside_last);
        ++n_phase_two_masked_;
        // OBVI_HOST_PHASE2_CHECK=1 (tests): the same phase II the reference's way -- rebuild with the excluded set, upload, solve -- on a
        // scratch handle from the same start values, window by window; the session goes on with the masked result
        static const bool check = std::getenv("OBVI_HOST_PHASE2_CHECK") && std::atoi(std::getenv("OBVI_HOST_PHASE2_CHECK")) != 0;
        if (check && ok2) {
          const obvi::SolverSummary fast = optimizer_.lastSummary();
          const ValueSnapshot fast_values = snapshotValues(problem.flat);
          restoreValues(pose_graph_copy);
          if (!check_problem_) check_problem_ = std::make_unique<obvi::Problem>(device_id_);
          pose_graph_optimizer::ObjectPoseGraphOptimizer ref_optimizer;
          std::optional<OptimizationLogger> null_logger;
          ref_optimizer.buildPoseGraphOptimization(scope, residual_params_, pose_graph, check_problem_.get(), null_logger, excluded);
          const bool ok_ref = ref_optimizer.solveOptimization(check_problem_.get(), iteration_params.phase_two_opt_params_, null_logger);
          const obvi::SolverSummary& ref = ref_optimizer.lastSummary();
          check_.windows++;
          if (!ok_ref) check_.failures++;
          check_.max_initial_cost_rel = std::max(check_.max_initial_cost_rel, std::fabs(fast.initial_cost - ref.initial_cost) / std::max(1e-300, std::fabs(ref.initial_cost)));
          check_.max_final_cost_rel = std::max(check_.max_final_cost_rel, std::fabs(fast.final_cost - ref.final_cost) / std::max(1e-300, std::fabs(ref.final_cost)));
          if (fast.iterations.size() != ref.iterations.size()) check_.iteration_mismatches++;
          if (fast.num_parameters_reduced != ref.num_parameters_reduced || fast.num_residuals_reduced != ref.num_residuals_reduced) check_.size_mismatches++;
          // rebuilt-problem values vs the masked run's values of the same blocks
          const ValueSnapshot ref_values = snapshotValues(check_problem_->flat);
          std::unordered_map<const double*, const double*> fast_of;
          const int dim[3] = {6, 3, 7};
          for (int k = 0; k < 3; ++k) for (size_t i = 0; i < fast_values.ptrs[k].size(); ++i) fast_of[fast_values.ptrs[k][i]] = &fast_values.values[k][dim[k] * i];
          for (int k = 0; k < 3; ++k)
            for (size_t i = 0; i < ref_values.ptrs[k].size(); ++i) {
              const auto it = fast_of.find(ref_values.ptrs[k][i]);
              if (it == fast_of.end()) { check_.size_mismatches++; continue; }
              double d = 0.0;
              for (int c = 0; c < dim[k]; ++c) {
                if (k == 2 && c == 3) continue;   // the yaw of an ellipsoid with dx == dy is a flat direction of the objective
                d = std::max(d, std::fabs(it->second[c] - ref_values.values[k][dim[k] * i + c]));
              }
              if (k == 0) check_.max_value_diff = std::max(check_.max_value_diff, d);   // poses
              else if (k == 1) { check_.points++; if (d > 1e-2) check_.points_apart++; }   // features: a few are barely constrained in depth and amplify any flipped iteration
              else if (std::fabs(it->second[4] - it->second[5]) < 0.1 * std::max(it->second[4], it->second[5])) { /* dx ~ dy: yaw, and with it the centre, float */ }
              else { check_.objects++; if (d > 1e-2) { check_.objects_apart++; if (std::getenv("OBVI_HOST_PHASE2_CHECK_VERBOSE")) { std::cerr << "object apart at frame " << next_frame_id << ":"; for (int c = 0; c < 7; ++c) std::cerr << " " << it->second[c] << "/" << ref_values.values[k][7 * i + c]; std::cerr << std::endl; } } }             // objects: one whose boxes are (nearly) all in the constant invalid-ellipse branch floats
            }
          restoreValues(pose_graph_copy);   // blocks the rebuilt problem does not hold keep their start values in both routes
          restoreValues(fast_values);
        }
      } else {
        optimizer_.buildPoseGraphOptimization(scope, residual_params_, pose_graph, &problem, opt_logger, excluded);
        ok2 = optimizer_.solveOptimization(&problem, iteration_params.phase_two_opt_params_, opt_logger, nullptr, nullptr, nullptr, nullptr, false, &beside_last);
        ++n_phase_two_rebuilt_;
      }
      if (!ok2) {
        std::cerr << "Phase II Optimization failed at max frame id " << next_frame_id << std::endl;
        return false;
      }
      if (opt_logger.has_value()) opt_logger->writeCurrentOptInfo();
      record(kind + "_phase_2", scope.min_frame_id_, next_frame_id, problem, device_selection ? n_excluded_device : excluded.size());
    }
    if (iteration_params.allow_reversion_after_detecting_jumps_ &&                                                           // :895-905
        !isConsecutivePosesStable_(pose_graph, scope.min_frame_id_, scope.max_frame_id_, iteration_params.consecutive_pose_transl_tol_, iteration_params.consecutive_pose_orient_tol_)) {
      std::cerr << "Detecting jumps after optimization. Reverting..." << std::endl;
      restoreValues(pose_graph_copy);
      records_.push_back({scope.min_frame_id_, next_frame_id, "reverted", 0, 0, 0, 0, 0, 0, 0});
    }
    visualize(pose_graph, start_opt_with_frame, next_frame_id, AFTER_EACH_OPTIMIZATION, attempt_num);                         // :908-913
    return true;
  }
  void visualize(const MainPgPtr& pose_graph, const FrameId& min_f, const FrameId& max_f, const VisualizationTypeEnum& when, const int& attempt_num) {
    if (visualization_callback_ && current_problem_data_ != nullptr) visualization_callback_(*current_problem_data_, pose_graph, min_f, max_f, when, attempt_num);
  }
  void record(const std::string& kind, FrameId min_f, FrameId max_f, const obvi::Problem& problem, size_t n_excl) {
    const obvi::SolverSummary& s = optimizer_.lastSummary();
    records_.push_back({min_f, max_f, kind, (int)s.iterations.size(), s.initial_cost, s.final_cost, problem.flat.frames.size(), problem.flat.features.size(), problem.flat.objects.size(), n_excl});
  }

  pose_graph_optimization::ObjectVisualPoseGraphResidualParams residual_params_;
  pose_graph_optimization::PoseGraphPlusObjectsOptimizationParams pgo_solver_params_;
  std::function<FrameId(const FrameId&)> window_provider_func_;
  OutputDataExtractor output_data_extractor_;
  MainPgPtr pose_graph_;
  std::function<bool(const FrameId&)> gba_checker_;
  std::function<pose_graph_optimization::OptimizationIterationParams(const FrameId&)> iteration_params_provider_func_;
  int device_id_;
  std::function<void(const OfflineProblemData&, MainPgPtr&)> pose_graph_creator_;
  FrameDataAdder frame_data_adder_;
  VisualizationCallback visualization_callback_;
  LimitTrajectoryEvaluationParams limit_trajectory_eval_params_;
  std::function<bool()> continue_opt_checker_;
  bool have_continue_opt_checker_ = false;
  const OfflineProblemData* current_problem_data_ = nullptr;
  std::function<bool(const MainPgPtr&)> object_merger_;
  size_t n_merge_rounds_ = 0;
  pose_graph_optimizer::ObjectPoseGraphOptimizer optimizer_;
  // planning the next window beside this window's last solve (runOptimization; OBVI_HOST_PLAN_AHEAD=0: off)
  struct Ahead { bool valid = false, uploaded = false; FrameId frame = 0, start = 0; pose_graph_optimizer::ResidualBlockInfoMap block_info; };
  Ahead ahead_;
  std::function<void()> ahead_job_;
  pose_graph_optimizer::ObjectPoseGraphOptimizer ahead_optimizer_;   // the build beside a solve has scratch of its own
  bool factor_hooks_set_ = false;
  pose_graph_optimizer::BesideThread stage_beside_thread_, ahead_thread_;
  std::atomic<bool> ahead_graph_done_{true};
  double time_stage_beside_ms_ = 0, time_stage_beside_wait_ms_ = 0; size_t n_stage_beside_ = 0;
  double time_ahead_ms_ = 0, time_ahead_add_ms_ = 0, time_ahead_build_ms_ = 0, time_ahead_upload_ms_ = 0; size_t n_ahead_ = 0;
  static bool planAheadEnabled() { static const bool on = !std::getenv("OBVI_HOST_PLAN_AHEAD") || std::atoi(std::getenv("OBVI_HOST_PLAN_AHEAD")) != 0; return on; }
  std::vector<OptimizationRecord> records_;
  double time_build_ms_ = 0, time_copy_ms_ = 0, time_add_ms_ = 0, time_select_ms_ = 0, time_pgo_ms_ = 0; size_t n_pgo_ = 0; size_t n_iterations_ = 0, n_phase_two_masked_ = 0, n_phase_two_rebuilt_ = 0;
  struct PhaseTwoCheck { size_t windows = 0, failures = 0, iteration_mismatches = 0, size_mismatches = 0, points = 0, points_apart = 0, objects = 0, objects_apart = 0; double max_initial_cost_rel = 0, max_final_cost_rel = 0, max_value_diff = 0; } check_;
  std::unique_ptr<obvi::Problem> check_problem_;
  bool extract_long_term_map_ = false;
  VisualFeatureAdder visual_feature_adder_;
  LongTermMapExtractionTunableParams ltm_tunable_params_;
  std::vector<obvi::CovarianceRankRepair> covariance_rank_repairs_;
  std::vector<LongTermMapEntry> long_term_map_;
};

// stands in for ceres::IterationCallback in the reference-shaped constructor below: the LM loop runs on the device, nothing calls it per iteration
namespace obvi_placeholder { struct IterationCallback { virtual ~IterationCallback() = default; }; }

// The reference's own shape (offline_problem_runner.h:27-98): five template parameters, fifteen constructor arguments in the reference's order, so that a
// construction site written for the reference (offline_problem_runner construction in ellipsoid_estimator / run_opt_from_pg_state) compiles against this header with
// ceres::Problem* -> obvi::Problem*, ceres::ResidualBlockId -> obvi::ResidualBlockId, ceres::IterationCallback -> obvi_placeholder::IterationCallback.
//   used     residual_params, limit_trajectory_eval_params, pgo_solver_params, continue_opt_checker (tested as the reference tests it), window_provider_func,
//            pose_graph_creator, frame_data_adder, output_data_extractor, visualization_callback, iteration_params_provider_func, object_merger, gba_checker
//   used     (round 6) refresh_residual_checker, residual_creator: run on the host at every build over the factors the build selected (obvi_optimizer.h FactorHooks):
//            a creator that returns false leaves its factor out, cached info is kept per factor and offered to the refresh checker as the reference does.  What a
//            creator cannot do here is define the residual's arithmetic (the five factors of the path run on the device) or a loss of its own.
//   ignored  ceres_callback_creator: see above.  ignoredHooks() names the ignored ones that were non-empty, so that a caller relying on one finds out.
// InputProblemData, VisualFeatureFactorType and PoseGraphType must be the types this path has fixed; CachedFactorInfo is free (it only types the ignored hooks).
template <typename InputProblemData, typename VisualFeatureFactorType, typename OutputProblemData, typename CachedFactorInfo, typename PoseGraphType>
class OfflineProblemRunner<InputProblemData, VisualFeatureFactorType, OutputProblemData, CachedFactorInfo, PoseGraphType> : public OfflineProblemRunner<OutputProblemData> {
  static_assert(std::is_same<InputProblemData, OfflineProblemData>::value, "this backend runs the offline path's input type (OfflineProblemData; the reference's UnassociatedBoundingBoxOfflineProblemData after its front ends)");
  static_assert(std::is_same<VisualFeatureFactorType, ReprojectionErrorFactor>::value, "the visual factor of this path is the reprojection error factor");
  static_assert(std::is_same<PoseGraphType, MainPg>::value, "the pose graph of this path is ObjectAndReprojectionFeaturePoseGraph");
  using Base = OfflineProblemRunner<OutputProblemData>;
 public:
  using RefreshResidualChecker = std::function<bool(const std::pair<FactorType, FeatureFactorId>&, const std::shared_ptr<PoseGraphType>&, const CachedFactorInfo&)>;
  using ResidualCreator = std::function<bool(const std::pair<FactorType, FeatureFactorId>&, const pose_graph_optimization::ObjectVisualPoseGraphResidualParams&,
                                             const std::shared_ptr<PoseGraphType>&, obvi::Problem*, obvi::ResidualBlockId&, CachedFactorInfo&)>;
  using CallbackCreator = std::function<std::vector<std::shared_ptr<obvi_placeholder::IterationCallback>>(const InputProblemData&, const std::shared_ptr<PoseGraphType>&, const FrameId&, const FrameId&)>;
  OfflineProblemRunner(const pose_graph_optimization::ObjectVisualPoseGraphResidualParams& residual_params,
                       const LimitTrajectoryEvaluationParams& limit_trajectory_eval_params,
                       const pose_graph_optimization::PoseGraphPlusObjectsOptimizationParams& pgo_solver_params,
                       const std::function<bool()>& continue_opt_checker,
                       const std::function<FrameId(const FrameId&)>& window_provider_func,
                       const RefreshResidualChecker& refresh_residual_checker,
                       const ResidualCreator& residual_creator,
                       const std::function<void(const InputProblemData&, std::shared_ptr<PoseGraphType>&)>& pose_graph_creator,
                       const std::function<void(const InputProblemData&, const std::shared_ptr<PoseGraphType>&, const FrameId&, const FrameId&)>& frame_data_adder,
                       const std::function<void(const InputProblemData&, const std::shared_ptr<PoseGraphType>&, const pose_graph_optimizer::OptimizationFactorsEnabledParams&, OutputProblemData&)>& output_data_extractor,
                       const CallbackCreator& ceres_callback_creator,
                       const std::function<void(const InputProblemData&, const std::shared_ptr<PoseGraphType>&, const FrameId&, const FrameId&, const VisualizationTypeEnum&, const int&)>& visualization_callback,
                       const std::function<pose_graph_optimization::OptimizationIterationParams(const FrameId&)>& iteration_params_provider_func,
                       const std::function<bool(const std::shared_ptr<PoseGraphType>&)> object_merger,
                       const std::function<bool(const FrameId&)>& gba_checker,
                       int device_id = 0)
      : Base(residual_params, pgo_solver_params, window_provider_func, output_data_extractor, gba_checker, iteration_params_provider_func, device_id) {
    Base::setLimitTrajectoryEvaluationParams(limit_trajectory_eval_params);
    Base::setContinueOptChecker(continue_opt_checker);
    if (pose_graph_creator) Base::setPoseGraphCreator(pose_graph_creator);
    if (frame_data_adder) Base::setFrameDataAdder(frame_data_adder);
    if (visualization_callback) Base::setVisualizationCallback(visualization_callback);
    if (object_merger) Base::setObjectMerger(object_merger);
    // The per-factor seam (object_pose_graph_optimizer.h:98-113): the caller's creator decides, factor by factor, whether a residual is made -- a `false` leaves the
    // factor out, as in the reference (:1042-1051) --, and its CachedFactorInfo lives here, keyed by factor, exactly as residual_blocks_and_cached_info_by_factor_id_
    // holds it there: a factor the caller has a cached residual for is offered to refresh_residual_checker first and re-created only if that says so (:1018-1032).
    if (residual_creator) {
      auto cache = std::make_shared<std::map<std::pair<FactorType, FeatureFactorId>, CachedFactorInfo>>();
      pose_graph_optimizer::FactorHooks hooks;
      if (refresh_residual_checker) {
        hooks.keep = [cache, refresh_residual_checker](const std::pair<FactorType, FeatureFactorId>& key, const std::shared_ptr<PoseGraphType>& pg) {
          const auto it = cache->find(key);
          if (it == cache->end()) return false;                       // no residual yet: create
          if (!refresh_residual_checker(key, pg, it->second)) return true;   // keep what is there
          cache->erase(it);                                            // refresh: the creator runs again
          return false;
        };
      } else {
        hooks.keep = [cache](const std::pair<FactorType, FeatureFactorId>& key, const std::shared_ptr<PoseGraphType>&) { return cache->count(key) != 0; };
      }
      hooks.create = [cache, residual_creator](const std::pair<FactorType, FeatureFactorId>& key, const pose_graph_optimization::ObjectVisualPoseGraphResidualParams& params,
                                              const std::shared_ptr<PoseGraphType>& pg, obvi::Problem* problem, obvi::ResidualBlockId& id) {
        CachedFactorInfo info{};
        if (!residual_creator(key, params, pg, problem, id, info)) return false;
        (*cache)[key] = info;
        return true;
      };
      Base::setFactorHooks(hooks);
    }
    if (ceres_callback_creator) ignored_hooks_.push_back("ceres_callback_creator");   // the LM loop runs on the device: nothing to call per iteration
  }
  const std::vector<std::string>& ignoredHooks() const { return ignored_hooks_; }
 private:
  std::vector<std::string> ignored_hooks_;
};

}  // namespace vslam_types_refactor
#endif  // OBVI_HOST_RUNNER_H_
