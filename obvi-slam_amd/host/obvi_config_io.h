// obvi_config_io.h -- the reference's parameter files (config/*.json, written by cv::FileStorage) read into the members of
// FullOVSLAMConfig that reach the optimisation path.
//
// The reference: readConfiguration (include/file_io/cv_file_storage/config_file_storage_io.h:1884-1898) reads the top-level entry
// "config" into FullOVSLAMConfig and THROWS std::invalid_argument unless config_schema_version equals kCurrentConfigSchemaVersion
// (full_ov_slam_config.h:24: 14).  Same here; `accept_older_schema` (not in the reference) lets the schema-11 / 12 files that most of
// config/ consists of (base7a_2_fallback.json is 12) through: the entries they lack -- schema 14 added the pre-PGO tracking and the two
// post-PGO feature-adjustment solver blocks -- keep the values FullOVSLAMConfig::base7a2Fallback() gives them.
//
// Layout of the file as the reference's Serializable* classes write it (config_file_storage_io.h:23-1870): nested maps under the members'
// names without the trailing underscore; bool as 0 / 1; FrameId (sliding_window_params, limit_traj_eval_params.max_frame_id) as
// SerializableUint64 = a decimal string; Eigen matrices as {"Rows", "Cols", "Data"} row-major; shape_dimension_priors as a list of
// {semantic_class, obj_dim_mean, dim_covariance}.  Entries outside the path (front ends, camera topics, bounding-box covariance
// generator, sparsifier, the long-term-map extractor's own solver blocks) are not read.
#ifndef OBVI_HOST_CONFIG_IO_H_
#define OBVI_HOST_CONFIG_IO_H_

#include <algorithm>
#include <fstream>
#include <ostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "obvi_optimization_runner.h"

namespace vslam_types_refactor {

constexpr int kCurrentConfigSchemaVersion = 14;   // full_ov_slam_config.h:24

namespace config_detail {
using obvi::json::Value;
using checkpoint_detail::member;
using checkpoint_detail::read_id;
using checkpoint_detail::read_num;

inline bool read_flag(const Value& v) { return v.kind == Value::Bool ? v.boolean : read_num(v) != 0.0; }   // cv::FileStorage writes bool as int
inline void read_solver_params(const Value& v, pose_graph_optimization::OptimizationSolverParams& p) {   // optimization_solver_params.h:10-60
  p.max_num_iterations_ = (int)read_num(member(v, "max_num_iterations"));
  p.allow_non_monotonic_steps_ = read_flag(member(v, "allow_non_monotonic_steps"));
  p.function_tolerance_ = read_num(member(v, "function_tolerance"));
  p.gradient_tolerance_ = read_num(member(v, "gradient_tolerance"));
  p.parameter_tolerance_ = read_num(member(v, "parameter_tolerance"));
  p.initial_trust_region_radius_ = read_num(member(v, "initial_trust_region_radius"));
  p.max_trust_region_radius_ = read_num(member(v, "max_trust_region_radius"));
}
inline void read_iteration_params(const Value& v, pose_graph_optimization::OptimizationIterationParams& p) {   // :62-84
  p.allow_reversion_after_detecting_jumps_ = read_flag(member(v, "allow_reversion_after_detecting_jumps"));
  p.consecutive_pose_transl_tol_ = read_num(member(v, "consecutive_pose_transl_tol"));
  p.consecutive_pose_orient_tol_ = read_num(member(v, "consecutive_pose_orient_tol"));
  p.feature_outlier_percentage_ = read_num(member(v, "feature_outlier_percentage"));
  read_solver_params(member(v, "phase_one_opt_params"), p.phase_one_opt_params_);
  read_solver_params(member(v, "phase_two_opt_params"), p.phase_two_opt_params_);
}
inline void read_rel_pose_cov(const Value& v, pose_graph_optimization::RelativePoseCovarianceOdomModelParams& p) {
  p.transl_error_mult_for_transl_error_ = read_num(member(v, "transl_error_mult_for_transl_error"));
  p.transl_error_mult_for_rot_error_ = read_num(member(v, "transl_error_mult_for_rot_error"));
  p.rot_error_mult_for_transl_error_ = read_num(member(v, "rot_error_mult_for_transl_error"));
  p.rot_error_mult_for_rot_error_ = read_num(member(v, "rot_error_mult_for_rot_error"));
}
inline void read_matrix(const Value& v, int rows, int cols, double* out) {   // vslam_basic_types_file_storage_io.h:19-69
  if ((int)read_num(member(v, "Rows")) != rows || (int)read_num(member(v, "Cols")) != cols) throw checkpoint_detail::ReadError("matrix of unexpected size");
  const Value& d = member(v, "Data");
  if (d.kind != Value::Array || (int)d.array.size() != rows * cols) throw checkpoint_detail::ReadError("matrix data of unexpected length");
  for (int i = 0; i < rows * cols; ++i) out[i] = read_num(d.array[(size_t)i]);
}
}  // namespace config_detail

// Returns false (reason in *error) when the text is not a configuration of this layout; the schema gate is the caller's (readConfiguration).
inline bool readConfigurationFromString(const std::string& text, FullOVSLAMConfig& configuration, std::string* error = nullptr) {
  using namespace config_detail;   // NOLINT
  Value root;
  std::string err;
  if (!obvi::json::Parser(text).parse(&root, &err)) { if (error) *error = "not JSON: " + err; return false; }
  try {
    const Value& c = member(root, "config");
    FullOVSLAMConfig out = FullOVSLAMConfig::base7a2Fallback();
    out.config_schema_version_ = (int)read_num(member(c, "config_schema_version"));
    const Value& id = member(c, "config_version_id");
    out.config_version_id_ = id.kind == Value::String ? id.string : std::string();
    out.visual_feature_params_.reprojection_error_std_dev_ = read_num(member(member(c, "visual_feature_params"), "reprojection_error_std_dev"));
    read_iteration_params(member(c, "local_ba_iteration_params"), out.local_ba_iteration_params_);
    read_iteration_params(member(c, "global_ba_iteration_params"), out.global_ba_iteration_params_);
    read_iteration_params(member(c, "final_ba_iteration_params"), out.final_ba_iteration_params_);
    {   // optimization_solver_params.h:142-207
      const Value& p = member(c, "pgo_solver_params");
      auto& pgo = out.pgo_solver_params_;
      pgo.relative_pose_factor_huber_loss_ = read_num(member(p, "relative_pose_factor_huber_loss"));
      pgo.enable_visual_feats_only_opt_post_pgo_ = read_flag(member(p, "enable_visual_feats_only_opt_post_pgo"));
      pgo.enable_visual_non_opt_feature_adjustment_post_pgo_ = read_flag(member(p, "enable_visual_non_opt_feature_adjustment_post_pgo"));
      read_rel_pose_cov(member(p, "relative_pose_cov_params"), pgo.relative_pose_cov_params_);
      read_solver_params(member(p, "pgo_optimization_solver_params"), pgo.pgo_optimization_solver_params_);
      read_solver_params(member(p, "final_pgo_optimization_solver_params"), pgo.final_pgo_optimization_solver_params_);
      // schema 14: three more solver blocks
      const struct { const char* key; pose_graph_optimization::OptimizationSolverParams* dst; } newer[3] = {
          {"post_pgo_vf_adjustment_solver_params", &pgo.post_pgo_vf_adjustment_solver_params_},
          {"final_post_pgo_vf_adjustment_solver_params", &pgo.final_post_pgo_vf_adjustment_solver_params_},
          {"pre_pgo_tracking_solver_params", &pgo.pre_pgo_tracking_solver_params_}};
      for (const auto& e : newer) {
        const Value* v = p.find(e.key);
        if (v != nullptr) read_solver_params(*v, *e.dst);
        else if (out.config_schema_version_ >= kCurrentConfigSchemaVersion) throw checkpoint_detail::ReadError(std::string("missing member ") + e.key);
      }
    }
    {
      const Value& p = member(c, "ltm_tunable_params");
      out.ltm_tunable_params_.far_feature_threshold_ = read_num(member(p, "far_feature_threshold"));
      out.ltm_tunable_params_.min_col_norm_ = read_num(member(p, "min_col_norm"));
      out.ltm_tunable_params_.fallback_to_prev_for_failed_extraction_ = read_flag(member(p, "fallback_to_prev_for_failed_extraction"));
    }
    {
      const Value& list = member(member(c, "shape_dimension_priors"), "dimension_prior_label");
      if (list.kind != Value::Array) throw checkpoint_detail::ReadError("dimension_prior_label: list expected");
      out.shape_dimension_priors_.clear();
      for (const Value& e : list.array) {
        const Value& cls = member(e, "semantic_class");
        if (cls.kind != Value::String) throw checkpoint_detail::ReadError("semantic_class: string expected");
        ObjectDim mean; Covariance<3> cov;
        read_matrix(member(e, "obj_dim_mean"), 3, 1, mean.data());
        read_matrix(member(e, "dim_covariance"), 3, 3, cov.data());
        out.shape_dimension_priors_[cls.string] = {mean, cov};
      }
    }
    {
      const Value& p = member(member(c, "bounding_box_front_end_params"), "post_session_object_merge_params");
      out.post_session_object_merge_params_.max_merge_distance_ = read_num(member(p, "max_merge_distance"));
      out.post_session_object_merge_params_.x_y_only_merge_ = read_flag(member(p, "x_y_only_merge"));
    }
    {
      const Value& p = member(c, "sliding_window_params");
      out.sliding_window_params_.global_ba_frequency_ = read_id(member(p, "global_ba_frequency"));
      out.sliding_window_params_.local_ba_window_size_ = read_id(member(p, "local_ba_window_size"));
    }
    {   // optimization_factors_enabled_params.h:12-110
      const Value& p = member(c, "optimization_factors_enabled_params");
      auto& en = out.optimization_factors_enabled_params_;
      en.min_low_level_feature_observations_per_frame_ = (uint32_t)read_num(member(p, "min_low_level_feature_observations_per_frame"));
      en.include_object_factors_ = read_flag(member(p, "include_object_factors"));
      en.include_visual_factors_ = read_flag(member(p, "include_visual_factors"));
      en.fix_poses_ = read_flag(member(p, "fix_poses"));
      en.fix_objects_ = read_flag(member(p, "fix_objects"));
      en.fix_visual_features_ = read_flag(member(p, "fix_visual_features"));
      en.fix_ltm_objects_ = read_flag(member(p, "fix_ltm_objects"));
      en.use_pom_ = read_flag(member(p, "use_pom"));
      en.poses_prior_to_window_to_keep_constant_ = (uint32_t)read_num(member(p, "poses_prior_to_window_to_keep_constant"));
      en.min_object_observations_ = (uint32_t)read_num(member(p, "min_object_observations"));
      en.min_low_level_feature_observations_ = (uint32_t)read_num(member(p, "min_low_level_feature_observations"));
      en.use_pose_graph_on_global_ba_ = read_flag(member(p, "use_pose_graph_on_global_ba"));
      en.use_visual_features_on_global_ba_ = read_flag(member(p, "use_visual_features_on_global_ba"));
      en.use_pose_graph_on_final_global_ba_ = read_flag(member(p, "use_pose_graph_on_final_global_ba"));
      en.use_visual_features_on_final_global_ba_ = read_flag(member(p, "use_visual_features_on_final_global_ba"));
    }
    {   // optimization_solver_params.h:86-140
      const Value& p = member(c, "object_visual_pose_graph_residual_params");
      auto& rp = out.object_visual_pose_graph_residual_params_;
      const Value& o = member(p, "object_residual_params");
      rp.object_residual_params_.object_observation_huber_loss_param_ = read_num(member(o, "object_observation_huber_loss_param"));
      rp.object_residual_params_.shape_dim_prior_factor_huber_loss_param_ = read_num(member(o, "shape_dim_prior_factor_huber_loss_param"));
      rp.object_residual_params_.invalid_ellipsoid_error_val_ = read_num(member(o, "invalid_ellipsoid_error_val"));
      rp.visual_residual_params_.reprojection_error_huber_loss_param_ = read_num(member(member(p, "visual_residual_params"), "reprojection_error_huber_loss_param"));
      rp.long_term_map_params_.pair_huber_loss_param_ = read_num(member(member(p, "long_term_map_params"), "pair_huber_loss_param"));
      rp.relative_pose_factor_huber_loss_ = read_num(member(p, "relative_pose_factor_huber_loss"));
      read_rel_pose_cov(member(p, "relative_pose_cov_params"), rp.relative_pose_cov_params_);
    }
    {   // limit_trajectory_evaluation_params.h:15-30
      const Value& p = member(c, "limit_traj_eval_params");
      out.limit_traj_eval_params_.should_limit_trajectory_evaluation_ = read_flag(member(p, "should_limit_trajectory_evaluation"));
      out.limit_traj_eval_params_.max_frame_id_ = read_id(member(p, "max_frame_id"));
    }
    configuration = out;
    return true;
  } catch (const std::runtime_error& e) {
    if (error) *error = e.what();
    return false;
  }
}

// config_file_storage_io.h:1884-1898.  Throws std::invalid_argument for a file of another schema version (as the reference does), and
// std::runtime_error for a file that cannot be read as a configuration at all (the reference would go on with default-constructed
// members: cv::FileStorage yields empty nodes).
inline void readConfiguration(const std::string& config_file_name, FullOVSLAMConfig& configuration, bool accept_older_schema = false) {
  std::ifstream in(config_file_name, std::ios::binary);
  if (!in) throw std::runtime_error("could not open configuration " + config_file_name);
  std::ostringstream text;
  text << in.rdbuf();
  std::string error;
  FullOVSLAMConfig read;
  if (!readConfigurationFromString(text.str(), read, &error)) throw std::runtime_error("configuration " + config_file_name + ": " + error);
  if (read.config_schema_version_ != kCurrentConfigSchemaVersion && !(accept_older_schema && read.config_schema_version_ < kCurrentConfigSchemaVersion))
    throw std::invalid_argument("configuration " + config_file_name + " has schema version " + std::to_string(read.config_schema_version_) + ", this reader takes version " +
                                std::to_string(kCurrentConfigSchemaVersion) + (accept_older_schema ? " or older" : ""));
  configuration = read;
}

// The entries read above, written back in the same layout (the reference: writeConfiguration, config_file_storage_io.h:1875-1882, for the whole struct):
// what --print-config shows, and the other half of the round trip the reference's test makes (test/file_io/cv_file_storage/config_file_storage_io_tests.cc:28).
inline void writeConfigurationToStream(std::ostream& os, const FullOVSLAMConfig& c) {
  const auto old_precision = os.precision(17);
  auto solver = [&](const char* key, const pose_graph_optimization::OptimizationSolverParams& p, const char* indent, bool last) {
    os << indent << "\"" << key << "\": {\"max_num_iterations\": " << p.max_num_iterations_ << ", \"allow_non_monotonic_steps\": " << (p.allow_non_monotonic_steps_ ? 1 : 0)
       << ", \"function_tolerance\": " << p.function_tolerance_ << ", \"gradient_tolerance\": " << p.gradient_tolerance_ << ", \"parameter_tolerance\": " << p.parameter_tolerance_
       << ", \"initial_trust_region_radius\": " << p.initial_trust_region_radius_ << ", \"max_trust_region_radius\": " << p.max_trust_region_radius_ << "}" << (last ? "\n" : ",\n");
  };
  auto iteration = [&](const char* key, const pose_graph_optimization::OptimizationIterationParams& p) {
    os << "  \"" << key << "\": {\n    \"allow_reversion_after_detecting_jumps\": " << (p.allow_reversion_after_detecting_jumps_ ? 1 : 0) << ", \"consecutive_pose_transl_tol\": " << p.consecutive_pose_transl_tol_
       << ", \"consecutive_pose_orient_tol\": " << p.consecutive_pose_orient_tol_ << ", \"feature_outlier_percentage\": " << p.feature_outlier_percentage_ << ",\n";
    solver("phase_one_opt_params", p.phase_one_opt_params_, "    ", false);
    solver("phase_two_opt_params", p.phase_two_opt_params_, "    ", true);
    os << "  },\n";
  };
  auto rel_cov = [&](const pose_graph_optimization::RelativePoseCovarianceOdomModelParams& p) {
    os << "\"relative_pose_cov_params\": {\"transl_error_mult_for_transl_error\": " << p.transl_error_mult_for_transl_error_ << ", \"transl_error_mult_for_rot_error\": " << p.transl_error_mult_for_rot_error_
       << ", \"rot_error_mult_for_transl_error\": " << p.rot_error_mult_for_transl_error_ << ", \"rot_error_mult_for_rot_error\": " << p.rot_error_mult_for_rot_error_ << "}";
  };
  auto matrix = [&](int rows, int cols, const double* v) {
    os << "{\"Rows\": " << rows << ", \"Cols\": " << cols << ", \"Data\": [";
    for (int i = 0; i < rows * cols; ++i) os << (i ? ", " : "") << v[i];
    os << "]}";
  };
  os << "{\"config\": {\n  \"config_schema_version\": " << c.config_schema_version_ << ",\n  \"config_version_id\": \"" << c.config_version_id_ << "\",\n";
  os << "  \"visual_feature_params\": {\"reprojection_error_std_dev\": " << c.visual_feature_params_.reprojection_error_std_dev_ << "},\n";
  iteration("local_ba_iteration_params", c.local_ba_iteration_params_);
  iteration("global_ba_iteration_params", c.global_ba_iteration_params_);
  iteration("final_ba_iteration_params", c.final_ba_iteration_params_);
  const auto& pgo = c.pgo_solver_params_;
  os << "  \"pgo_solver_params\": {\n    \"relative_pose_factor_huber_loss\": " << pgo.relative_pose_factor_huber_loss_ << ", \"enable_visual_feats_only_opt_post_pgo\": " << (pgo.enable_visual_feats_only_opt_post_pgo_ ? 1 : 0)
     << ", \"enable_visual_non_opt_feature_adjustment_post_pgo\": " << (pgo.enable_visual_non_opt_feature_adjustment_post_pgo_ ? 1 : 0) << ",\n    ";
  rel_cov(pgo.relative_pose_cov_params_);
  os << ",\n";
  solver("pgo_optimization_solver_params", pgo.pgo_optimization_solver_params_, "    ", false);
  solver("final_pgo_optimization_solver_params", pgo.final_pgo_optimization_solver_params_, "    ", false);
  solver("post_pgo_vf_adjustment_solver_params", pgo.post_pgo_vf_adjustment_solver_params_, "    ", false);
  solver("final_post_pgo_vf_adjustment_solver_params", pgo.final_post_pgo_vf_adjustment_solver_params_, "    ", false);
  solver("pre_pgo_tracking_solver_params", pgo.pre_pgo_tracking_solver_params_, "    ", true);
  os << "  },\n  \"ltm_tunable_params\": {\"far_feature_threshold\": " << c.ltm_tunable_params_.far_feature_threshold_ << ", \"min_col_norm\": " << c.ltm_tunable_params_.min_col_norm_
     << ", \"fallback_to_prev_for_failed_extraction\": " << (c.ltm_tunable_params_.fallback_to_prev_for_failed_extraction_ ? 1 : 0) << "},\n";
  os << "  \"shape_dimension_priors\": {\"dimension_prior_label\": [";
  {
    std::vector<std::string> classes;
    for (const auto& e : c.shape_dimension_priors_) classes.push_back(e.first);
    std::sort(classes.begin(), classes.end());
    for (size_t i = 0; i < classes.size(); ++i) {
      const auto& e = c.shape_dimension_priors_.at(classes[i]);
      os << (i ? ",\n    " : "\n    ") << "{\"semantic_class\": \"" << classes[i] << "\", \"obj_dim_mean\": ";
      matrix(3, 1, e.first.data());
      os << ", \"dim_covariance\": ";
      matrix(3, 3, e.second.data());
      os << "}";
    }
  }
  os << "]},\n  \"bounding_box_front_end_params\": {\"post_session_object_merge_params\": {\"max_merge_distance\": " << c.post_session_object_merge_params_.max_merge_distance_
     << ", \"x_y_only_merge\": " << (c.post_session_object_merge_params_.x_y_only_merge_ ? 1 : 0) << "}},\n";
  os << "  \"sliding_window_params\": {\"global_ba_frequency\": \"" << c.sliding_window_params_.global_ba_frequency_ << "\", \"local_ba_window_size\": \"" << c.sliding_window_params_.local_ba_window_size_ << "\"},\n";
  const auto& en = c.optimization_factors_enabled_params_;
  os << "  \"optimization_factors_enabled_params\": {\"min_low_level_feature_observations_per_frame\": " << en.min_low_level_feature_observations_per_frame_ << ", \"include_object_factors\": " << (en.include_object_factors_ ? 1 : 0)
     << ", \"include_visual_factors\": " << (en.include_visual_factors_ ? 1 : 0) << ", \"fix_poses\": " << (en.fix_poses_ ? 1 : 0) << ", \"fix_objects\": " << (en.fix_objects_ ? 1 : 0)
     << ", \"fix_visual_features\": " << (en.fix_visual_features_ ? 1 : 0) << ", \"fix_ltm_objects\": " << (en.fix_ltm_objects_ ? 1 : 0) << ", \"use_pom\": " << (en.use_pom_ ? 1 : 0)
     << ",\n    \"poses_prior_to_window_to_keep_constant\": " << en.poses_prior_to_window_to_keep_constant_ << ", \"min_object_observations\": " << en.min_object_observations_
     << ", \"min_low_level_feature_observations\": " << en.min_low_level_feature_observations_ << ", \"use_pose_graph_on_global_ba\": " << (en.use_pose_graph_on_global_ba_ ? 1 : 0)
     << ", \"use_visual_features_on_global_ba\": " << (en.use_visual_features_on_global_ba_ ? 1 : 0) << ", \"use_pose_graph_on_final_global_ba\": " << (en.use_pose_graph_on_final_global_ba_ ? 1 : 0)
     << ", \"use_visual_features_on_final_global_ba\": " << (en.use_visual_features_on_final_global_ba_ ? 1 : 0) << "},\n";
  const auto& rp = c.object_visual_pose_graph_residual_params_;
  os << "  \"object_visual_pose_graph_residual_params\": {\n    \"object_residual_params\": {\"object_observation_huber_loss_param\": " << rp.object_residual_params_.object_observation_huber_loss_param_
     << ", \"shape_dim_prior_factor_huber_loss_param\": " << rp.object_residual_params_.shape_dim_prior_factor_huber_loss_param_ << ", \"invalid_ellipsoid_error_val\": " << rp.object_residual_params_.invalid_ellipsoid_error_val_
     << "},\n    \"visual_residual_params\": {\"reprojection_error_huber_loss_param\": " << rp.visual_residual_params_.reprojection_error_huber_loss_param_ << "}, \"long_term_map_params\": {\"pair_huber_loss_param\": "
     << rp.long_term_map_params_.pair_huber_loss_param_ << "}, \"relative_pose_factor_huber_loss\": " << rp.relative_pose_factor_huber_loss_ << ",\n    ";
  rel_cov(rp.relative_pose_cov_params_);
  os << "\n  },\n  \"limit_traj_eval_params\": {\"should_limit_trajectory_evaluation\": " << (c.limit_traj_eval_params_.should_limit_trajectory_evaluation_ ? 1 : 0) << ", \"max_frame_id\": \""
     << c.limit_traj_eval_params_.max_frame_id_ << "\"}\n}}\n";
  os.precision(old_precision);
}

}  // namespace vslam_types_refactor
#endif  // OBVI_HOST_CONFIG_IO_H_
