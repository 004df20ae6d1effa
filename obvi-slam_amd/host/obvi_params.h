// obvi_params.h -- parameter structs of the optimisation path, field-for-field the reference's
// (include/refactoring/optimization/optimization_solver_params.h:10-207,
//  include/refactoring/optimization/optimization_factors_enabled_params.h:12-110,
//  include/refactoring/configuration/full_ov_slam_config.h sliding-window block).
#ifndef OBVI_HOST_PARAMS_H_
#define OBVI_HOST_PARAMS_H_

#include <cmath>
#include <cstdint>
#include <unordered_set>

#include "obvi_types.h"

namespace pose_graph_optimization {

struct OptimizationSolverParams {
  int max_num_iterations_ = 100;
  bool allow_non_monotonic_steps_ = false;
  double function_tolerance_ = 1e-6;
  double gradient_tolerance_ = 1e-10;
  double parameter_tolerance_ = 1e-8;
  double initial_trust_region_radius_ = 1e4;
  double max_trust_region_radius_ = 1e16;
};

struct OptimizationIterationParams {
  bool allow_reversion_after_detecting_jumps_ = true;
  double consecutive_pose_transl_tol_ = 1.0;
  double consecutive_pose_orient_tol_ = M_PI;
  double feature_outlier_percentage_ = .1;
  OptimizationSolverParams phase_one_opt_params_;
  OptimizationSolverParams phase_two_opt_params_;
};

struct ObjectResidualParams {
  double object_observation_huber_loss_param_ = 1;
  double shape_dim_prior_factor_huber_loss_param_ = 1;
  double invalid_ellipsoid_error_val_ = 1e6;
};
struct PairwiseLongTermMapResidualParams { double pair_huber_loss_param_ = 1; };
struct VisualFeaturePoseGraphResidualParams { double reprojection_error_huber_loss_param_ = 1; };
struct RelativePoseCovarianceOdomModelParams {
  double transl_error_mult_for_transl_error_ = 0.025;
  double transl_error_mult_for_rot_error_ = 0.025;
  double rot_error_mult_for_transl_error_ = 0.025;
  double rot_error_mult_for_rot_error_ = 0.025;
};
struct ObjectVisualPoseGraphResidualParams {
  ObjectResidualParams object_residual_params_;
  VisualFeaturePoseGraphResidualParams visual_residual_params_;
  PairwiseLongTermMapResidualParams long_term_map_params_;
  double relative_pose_factor_huber_loss_ = 1.0;
  RelativePoseCovarianceOdomModelParams relative_pose_cov_params_;
};
struct PoseGraphPlusObjectsOptimizationParams {
  double relative_pose_factor_huber_loss_ = 1.0;
  bool enable_visual_feats_only_opt_post_pgo_ = false;
  bool enable_visual_non_opt_feature_adjustment_post_pgo_ = false;
  RelativePoseCovarianceOdomModelParams relative_pose_cov_params_;
  OptimizationSolverParams pgo_optimization_solver_params_;
  OptimizationSolverParams final_pgo_optimization_solver_params_;
  OptimizationSolverParams post_pgo_vf_adjustment_solver_params_;
  OptimizationSolverParams final_post_pgo_vf_adjustment_solver_params_;
  OptimizationSolverParams pre_pgo_tracking_solver_params_;
};

}  // namespace pose_graph_optimization

namespace pose_graph_optimizer {

struct OptimizationFactorsEnabledParams {
  uint32_t min_low_level_feature_observations_per_frame_ = 50;
  bool include_object_factors_ = true;
  bool include_visual_factors_ = true;
  bool fix_poses_ = true;
  bool fix_objects_ = true;
  bool fix_visual_features_ = true;
  bool fix_ltm_objects_ = false;
  bool use_pom_ = false;
  uint32_t poses_prior_to_window_to_keep_constant_ = 1;
  uint32_t min_object_observations_ = 1;
  uint32_t min_low_level_feature_observations_ = 3;
  bool use_pose_graph_on_global_ba_ = false;
  bool use_visual_features_on_global_ba_ = false;
  bool use_pose_graph_on_final_global_ba_ = false;
  bool use_visual_features_on_final_global_ba_ = false;
};

struct OptimizationScopeParams {
  uint32_t min_low_level_feature_observations_per_frame_ = 50;
  bool include_object_factors_ = true;
  bool include_visual_factors_ = true;
  bool fix_poses_ = false;
  bool fix_objects_ = false;
  bool fix_visual_features_ = false;
  bool use_pom_ = false;
  bool fix_ltm_objects_ = false;
  uint32_t poses_prior_to_window_to_keep_constant_ = 1;
  uint32_t min_object_observations_ = 1;
  uint32_t min_low_level_feature_observations_ = 3;
  std::unordered_set<vslam_types_refactor::FactorType> factor_types_to_exclude;
  vslam_types_refactor::FrameId min_frame_id_ = 0;
  vslam_types_refactor::FrameId max_frame_id_ = 0;
  bool force_include_ltm_objs_ = false;
};

}  // namespace pose_graph_optimizer

namespace vslam_types_refactor {
// long_term_map_extraction_tunable_params.h:11-17
struct LongTermMapExtractionTunableParams { double far_feature_threshold_ = 75; double min_col_norm_ = 5e-9; bool fallback_to_prev_for_failed_extraction_ = true; };
struct SlidingWindowParams {   // full_ov_slam_config.h; values of config/base7a_2_fallback.json
  FrameId global_ba_frequency_ = 30;
  FrameId local_ba_window_size_ = 50;
};
// run_opt_utils.h:101-116
inline FrameId provideOptimizationWindow(const FrameId& max_frame_to_opt, const FrameId& max_frame_id, const SlidingWindowParams& p) {
  if (max_frame_to_opt == max_frame_id) return 0;
  if ((max_frame_to_opt % p.global_ba_frequency_) == 0) return 0;
  if (max_frame_to_opt < p.local_ba_window_size_) return 0;
  return max_frame_to_opt - p.local_ba_window_size_;
}
}  // namespace vslam_types_refactor
#endif  // OBVI_HOST_PARAMS_H_
