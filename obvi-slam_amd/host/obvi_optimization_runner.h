// obvi_optimization_runner.h -- runFullOptimization: the entry the reference's executables call
// (include/run_optimization_utils/optimization_runner.h:22-651; offline_object_visual_slam_main.cpp and
// run_opt_from_pg_state.cpp:290-312 both end in it).  It wires the window provider (run_opt_utils.h:101-116), the global-BA
// checker (:195-203), the per-frame solver-parameter provider (:204-216), the post-session object merger (:545-640) and the
// output extraction around OfflineProblemRunner::runOptimization, for inputs whose front ends have already run (associations are
// given: the front ends are out of scope).
#ifndef OBVI_HOST_OPTIMIZATION_RUNNER_H_
#define OBVI_HOST_OPTIMIZATION_RUNNER_H_

#include <functional>
#include <optional>
#include <string>
#include <unordered_map>
#include <vector>

#include "obvi_checkpoint_io.h"
#include "obvi_runner.h"

namespace vslam_types_refactor {

// post_session_object_merge_params_ of the bounding-box front-end parameters (full_ov_slam_config.h)
struct PostSessionObjectMergeParams { double max_merge_distance_ = -1.0; bool x_y_only_merge_ = true; };

// The members of FullOVSLAMConfig (include/refactoring/configuration/full_ov_slam_config.h) that reach the optimisation path, under
// the reference's names; defaults are config/base7a_2_fallback.json.
struct FullOVSLAMConfig {
  pose_graph_optimization::OptimizationIterationParams local_ba_iteration_params_, global_ba_iteration_params_, final_ba_iteration_params_;
  pose_graph_optimization::PoseGraphPlusObjectsOptimizationParams pgo_solver_params_;
  pose_graph_optimization::ObjectVisualPoseGraphResidualParams object_visual_pose_graph_residual_params_;
  pose_graph_optimizer::OptimizationFactorsEnabledParams optimization_factors_enabled_params_;
  SlidingWindowParams sliding_window_params_;
  PostSessionObjectMergeParams post_session_object_merge_params_;   // bounding_box_front_end_params_.post_session_object_merge_params_
  LongTermMapExtractionTunableParams ltm_tunable_params_;
  // what a parameter file adds (obvi_config_io.h): its schema version and id, the pixel noise of the visual factors (visual_feature_params_), the shape priors by
  // semantic class (shape_dimension_priors_), the limit on the evaluated trajectory
  int config_schema_version_ = 12;                   // base7a_2_fallback.json's
  std::string config_version_id_ = "base7a_2";
  struct VisualFeatureParams { double reprojection_error_std_dev_ = 1.5; } visual_feature_params_;
  std::unordered_map<std::string, std::pair<ObjectDim, Covariance<3>>> shape_dimension_priors_;
  LimitTrajectoryEvaluationParams limit_traj_eval_params_;

  static pose_graph_optimization::OptimizationSolverParams solverParams(int iterations, double function_tolerance) {
    pose_graph_optimization::OptimizationSolverParams p;
    p.max_num_iterations_ = iterations; p.allow_non_monotonic_steps_ = true; p.function_tolerance_ = function_tolerance; p.gradient_tolerance_ = 1e-10; p.parameter_tolerance_ = 1e-8;
    p.initial_trust_region_radius_ = 100; p.max_trust_region_radius_ = 1e4;
    return p;
  }
  static FullOVSLAMConfig base7a2Fallback() {
    FullOVSLAMConfig c;
    auto& rp = c.object_visual_pose_graph_residual_params_;
    rp.object_residual_params_.object_observation_huber_loss_param_ = 0.5; rp.object_residual_params_.shape_dim_prior_factor_huber_loss_param_ = 10;
    rp.object_residual_params_.invalid_ellipsoid_error_val_ = 1000; rp.visual_residual_params_.reprojection_error_huber_loss_param_ = 1.0;
    rp.long_term_map_params_.pair_huber_loss_param_ = 1.0; rp.relative_pose_factor_huber_loss_ = 1.0;
    auto& en = c.optimization_factors_enabled_params_;
    en.include_object_factors_ = true; en.include_visual_factors_ = true; en.fix_poses_ = en.fix_objects_ = en.fix_visual_features_ = en.fix_ltm_objects_ = false;
    en.poses_prior_to_window_to_keep_constant_ = 5; en.min_object_observations_ = 10; en.min_low_level_feature_observations_ = 5; en.min_low_level_feature_observations_per_frame_ = 50;
    en.use_pose_graph_on_global_ba_ = true; en.use_visual_features_on_global_ba_ = false; en.use_pose_graph_on_final_global_ba_ = true; en.use_visual_features_on_final_global_ba_ = true;
    auto& pgo = c.pgo_solver_params_;
    pgo.relative_pose_factor_huber_loss_ = 5.0; pgo.enable_visual_feats_only_opt_post_pgo_ = true; pgo.enable_visual_non_opt_feature_adjustment_post_pgo_ = true;
    pgo.relative_pose_cov_params_ = {0.1, 0.1, 0.1, 0.1};
    pgo.pgo_optimization_solver_params_ = solverParams(250, 1e-6); pgo.final_pgo_optimization_solver_params_ = solverParams(300, 1e-6);
    pgo.post_pgo_vf_adjustment_solver_params_ = solverParams(250, 1e-6); pgo.final_post_pgo_vf_adjustment_solver_params_ = solverParams(300, 1e-6);
    pgo.pre_pgo_tracking_solver_params_ = solverParams(50, 1e-3);
    c.local_ba_iteration_params_.phase_one_opt_params_ = solverParams(50, 1e-3); c.local_ba_iteration_params_.phase_two_opt_params_ = solverParams(100, 1e-4);
    c.global_ba_iteration_params_.phase_one_opt_params_ = solverParams(250, 1e-6); c.global_ba_iteration_params_.phase_two_opt_params_ = solverParams(250, 1e-6);
    c.final_ba_iteration_params_.phase_one_opt_params_ = solverParams(300, 1e-6); c.final_ba_iteration_params_.phase_two_opt_params_ = solverParams(300, 1e-6);
    c.ltm_tunable_params_.far_feature_threshold_ = 75.0; c.ltm_tunable_params_.min_col_norm_ = 5e-4; c.ltm_tunable_params_.fallback_to_prev_for_failed_extraction_ = true;
    c.post_session_object_merge_params_.max_merge_distance_ = 2.0; c.post_session_object_merge_params_.x_y_only_merge_ = true;
    return c;
  }
};

// What runFullOptimization hands back (LongTermObjectMapAndResults, output_problem_data.h:11-40, reduced to the path's outputs)
struct LongTermObjectMapAndResults {
  std::vector<LongTermMapEntry> long_term_map_;          // ellipsoid means + 7x7 marginal covariances
  std::unordered_map<FrameId, RawPose3d> robot_pose_results_;
  std::unordered_map<ObjectId, RawEllipsoid> ellipsoid_results_;
  std::unordered_map<FeatureId, Position3d> visual_feature_results_;
  std::vector<OptimizationRecord> records_;
  size_t post_session_merge_rounds_ = 0;
  std::vector<obvi::CovarianceRankRepair> covariance_rank_repairs_;                  // parameters that needed a prior for the covariance to exist
};

// The reference's remaining runner hooks as runFullOptimization hands them on (optimization_runner.h:262-272 continue_opt_checker = ros::ok, :433-497 the
// visualization callback, offline_object_visual_slam_main.cpp:1046-1049 the limit on the evaluated trajectory), and the switch that routes the session through the
// runner's reference-shaped specialisation (five template parameters, fifteen constructor arguments; obvi_runner.h) instead of the short constructor + setters.
struct RunnerHooks {
  bool reference_shaped_runner_ = false;
  LimitTrajectoryEvaluationParams limit_trajectory_eval_params_;
  std::function<bool()> continue_opt_checker_ = []() { return true; };
  std::function<void(const OfflineProblemData&, const MainPgPtr&, const FrameId&, const FrameId&, const VisualizationTypeEnum&, const int&)> visualization_callback_;
  std::vector<std::string> ignored_hooks_;   // out: the constructor hooks the reference-shaped runner accepted and does not use
  // The residual creator the reference-shaped construction site passes: by default one that makes every residual (the reference's creator fails only on a factor
  // it cannot build).  creator_rejects_every_ = N > 0 (the driver's --creator-rejects-every N; tests): a creator that cannot make the residual of every visual /
  // bounding-box factor whose id is N - 1 modulo N -- the seam in action: those factors must be left out of every problem of the session.
  int creator_rejects_every_ = 0;
  size_t creator_calls_ = 0, creator_rejections_ = 0, refresh_calls_ = 0, factors_left_out_ = 0;   // out
};

// optimization_runner.h:22-651.  `problem_data` carries what the reference passes as bounding_boxes / visual_features / robot_poses /
// long_term_map after its front ends; `pose_graph_creator` may be empty (a fresh graph) or hand out a checkpoint's graph.
inline bool runFullOptimization(std::optional<OptimizationLogger>& opt_logger, const FullOVSLAMConfig& config, const OfflineProblemData& problem_data,
                                const std::function<void(const OfflineProblemData&, MainPgPtr&)>& pose_graph_creator, const std::string& output_checkpoints_dir,
                                LongTermObjectMapAndResults& output_results, const FrameId& start_at_frame = 0, const bool& add_data_for_starting_frame = true,
                                int device_id = 0, bool extract_long_term_map = true, MainPgPtr* pose_graph_out = nullptr,
                                const VisualFeatureAdder& visual_feature_adder = nullptr, RunnerHooks* hooks = nullptr) {
  const FrameId max_frame_id = problem_data.getMaxFrameId();                                                                 // :186-189
  std::function<FrameId(const FrameId&)> window_provider_func = [&](const FrameId& f) { return provideOptimizationWindow(f, max_frame_id, config.sliding_window_params_); };
  std::function<bool(const FrameId&)> gba_checker = [&](const FrameId& max_frame_to_opt) {                                  // :195-203
    return max_frame_to_opt - window_provider_func(max_frame_to_opt) > config.sliding_window_params_.local_ba_window_size_;
  };
  std::function<pose_graph_optimization::OptimizationIterationParams(const FrameId&)> solver_params_provider_func = [&](const FrameId& max_frame_to_opt) {   // :204-216
    if (max_frame_to_opt == max_frame_id) return config.final_ba_iteration_params_;
    if (max_frame_to_opt % config.sliding_window_params_.global_ba_frequency_ == 0) return config.global_ba_iteration_params_;
    return config.local_ba_iteration_params_;
  };
  // the output extractor (optimization_runner.h:217-260): estimates of the final pose graph; the long-term map the runner extracted from the
  // final problem rides along (the reference's extractor calls its long-term-map extractor here)
  using Runner = OfflineProblemRunner<LongTermObjectMapAndResults>;
  Runner* runner_ptr = nullptr;
  Runner::OutputDataExtractor output_data_extractor = [&runner_ptr](const OfflineProblemData&, const MainPgPtr& pose_graph, const pose_graph_optimizer::OptimizationFactorsEnabledParams&,
                                                                     LongTermObjectMapAndResults& output_problem_data) {
    pose_graph->getRobotPoseEstimates(output_problem_data.robot_pose_results_);
    pose_graph->getObjectEstimates(output_problem_data.ellipsoid_results_);
    pose_graph->getVisualFeatureEstimates(output_problem_data.visual_feature_results_);
    if (runner_ptr) { output_problem_data.long_term_map_ = runner_ptr->longTermMap(); output_problem_data.covariance_rank_repairs_ = runner_ptr->covarianceRankRepairs(); }
  };
  // :545-640 the post-session merger: decide by centre proximity, merge in the pose graph (the front end's own bookkeeping of the
  // merged objects is association state and not part of this path)
  const std::function<bool(const MainPgPtr&)> object_merger = [&](const MainPgPtr& pose_graph) {
    std::unordered_map<ObjectId, std::unordered_set<ObjectId>> merge_results;
    identifyMergeObjectsBasedOnCenterProximity(pose_graph, config.post_session_object_merge_params_.max_merge_distance_, config.post_session_object_merge_params_.x_y_only_merge_, merge_results);
    if (merge_results.empty()) return false;
    return pose_graph->mergeObjects(merge_results);
  };
  std::unique_ptr<Runner> runner_holder;
  if (hooks != nullptr && hooks->reference_shaped_runner_) {
    // the construction site as the reference writes it (optimization_runner.h:509-543): every hook through the constructor, in the reference's order
    struct CachedInfo {};   // the reference: util::EmptyStruct
    using ReferenceShaped = OfflineProblemRunner<OfflineProblemData, ReprojectionErrorFactor, LongTermObjectMapAndResults, CachedInfo, MainPg>;
    const ReferenceShaped::RefreshResidualChecker refresh_residual_checker = [hooks](const std::pair<FactorType, FeatureFactorId>&, const MainPgPtr&, const CachedInfo&) { ++hooks->refresh_calls_; return true; };   // :273-279 (always refresh)
    const int every = hooks->creator_rejects_every_;
    const ReferenceShaped::ResidualCreator residual_creator = [hooks, every](const std::pair<FactorType, FeatureFactorId>& factor, const pose_graph_optimization::ObjectVisualPoseGraphResidualParams&, const MainPgPtr&,
                                                                              obvi::Problem*, obvi::ResidualBlockId&, CachedInfo&) {                                             // :280-298
      ++hooks->creator_calls_;
      const bool observation = factor.first == kReprojectionErrorFactorTypeId || factor.first == kObjectObservationFactorTypeId;
      if (every > 0 && observation && factor.second % (FeatureFactorId)every == (FeatureFactorId)(every - 1)) { ++hooks->creator_rejections_; return false; }
      return true;
    };
    const std::function<void(const OfflineProblemData&, const MainPgPtr&, const FrameId&, const FrameId&)> frame_data_adder =                                                         // :364-418
        [&](const OfflineProblemData& data, const MainPgPtr& pose_graph, const FrameId& min_frame_id, const FrameId& frame_to_add) {
          addFrameDataToPoseGraph(data, pose_graph, frame_to_add, config.object_visual_pose_graph_residual_params_.relative_pose_cov_params_, visual_feature_adder, min_frame_id);
        };
    const ReferenceShaped::CallbackCreator ceres_callback_creator = [](const OfflineProblemData&, const MainPgPtr&, const FrameId&, const FrameId&) {                                  // :420-431
      return std::vector<std::shared_ptr<obvi_placeholder::IterationCallback>>{};
    };
    auto shaped = std::make_unique<ReferenceShaped>(config.object_visual_pose_graph_residual_params_, hooks->limit_trajectory_eval_params_, config.pgo_solver_params_, hooks->continue_opt_checker_,
                                                    window_provider_func, refresh_residual_checker, residual_creator, pose_graph_creator, frame_data_adder, output_data_extractor,
                                                    ceres_callback_creator, hooks->visualization_callback_, solver_params_provider_func, object_merger, gba_checker, device_id);
    hooks->ignored_hooks_ = shaped->ignoredHooks();
    runner_holder = std::move(shaped);
  } else {
    runner_holder = std::make_unique<Runner>(config.object_visual_pose_graph_residual_params_, config.pgo_solver_params_, window_provider_func, output_data_extractor, gba_checker, solver_params_provider_func, device_id);
    if (pose_graph_creator) runner_holder->setPoseGraphCreator(pose_graph_creator);
    runner_holder->setObjectMerger(object_merger);
    if (visual_feature_adder) runner_holder->setVisualFeatureAdder(visual_feature_adder);   // the visual front end decides what enters the graph (obvi_visual_feature_front_end.h)
    if (hooks != nullptr) {
      runner_holder->setLimitTrajectoryEvaluationParams(hooks->limit_trajectory_eval_params_);
      if (hooks->visualization_callback_) runner_holder->setVisualizationCallback(hooks->visualization_callback_);
      runner_holder->setContinueOptChecker(hooks->continue_opt_checker_);
    }
  }
  Runner& runner = *runner_holder;
  runner_ptr = &runner;
  runner.setExtractLongTermMap(extract_long_term_map);
  runner.setLongTermMapTunableParams(config.ltm_tunable_params_);
  const bool ok = runner.runOptimization(problem_data, config.optimization_factors_enabled_params_, opt_logger, output_results, start_at_frame, add_data_for_starting_frame);
  if (std::getenv("OBVI_HOST_TIMING")) runner.printTiming(std::cerr);
  if (hooks != nullptr) hooks->factors_left_out_ = runner.factorsLeftOutByTheCreator();
  output_results.records_ = runner.records();
  output_results.post_session_merge_rounds_ = runner.mergeRounds();
  const MainPgPtr pose_graph = runner.poseGraph();
  if (!ok || !pose_graph) return false;
  if (!output_checkpoints_dir.empty()) {   // :499-507 the final state as a checkpoint (kLtmCheckpointOutputFileBaseName + .json): what run_opt_from_pg_state / ltm_extraction_only replay
    const std::string dir = output_checkpoints_dir.back() == '/' ? output_checkpoints_dir : output_checkpoints_dir + "/";
    if (!outputPoseGraphToFile(pose_graph, dir + "long_term_map_checkpoint.json")) std::cerr << "could not write the final checkpoint to " << dir << std::endl;
  }
  if (pose_graph_out) *pose_graph_out = pose_graph;
  return true;
}

}  // namespace vslam_types_refactor
#endif  // OBVI_HOST_OPTIMIZATION_RUNNER_H_
