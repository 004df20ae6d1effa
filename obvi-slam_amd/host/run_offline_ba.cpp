// run_offline_ba.cpp -- driver of the host-side mirror: the shape of the reference's
// offline_object_visual_slam_main / run_opt_from_pg_state for a scene whose associations are given.
//   run_offline_ba <scene.txt> <out.json> [--window W] [--gba-frequency F] [--device D] [--csv ceres_opt_summary.csv] [--ltm]
//   run_offline_ba <scene.txt> <out.json> --dump-build MIN MAX [--excluded-every K] [--phase-two-masks]   (no GPU: flattening only)
//   run_offline_ba <scene.txt> <out.json> --pending-objects [--device D]   refineInitialEstimateForPendingObjects over every object of the scene
// Parameter values: config/base7a_2_fallback.json of the reference.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>

#include "obvi_pending_object_estimator.h"
#include "obvi_runner.h"

using namespace vslam_types_refactor;   // NOLINT

static bool loadScene(const std::string& path, OfflineProblemData* d) {
  std::ifstream in(path);
  if (!in) return false;
  std::string tag; int version; size_t n;
  in >> tag >> version;
  if (tag != "obvi_scene" || version != 1) return false;
  in >> tag >> n;   // cameras
  for (size_t i = 0; i < n; ++i) {
    CameraId id; CameraIntrinsicsMat k; CameraExtrinsics e;
    in >> id >> k.fx >> k.fy >> k.cx >> k.cy >> e.transl_[0] >> e.transl_[1] >> e.transl_[2] >> e.orientation_[0] >> e.orientation_[1] >> e.orientation_[2];
    d->camera_intrinsics_by_camera_[id] = k; d->camera_extrinsics_by_camera_[id] = e;
  }
  in >> tag >> n;   // frames
  d->robot_poses_.resize(n);
  for (size_t i = 0; i < n; ++i) { Pose3D& p = d->robot_poses_[i]; in >> p.transl_[0] >> p.transl_[1] >> p.transl_[2] >> p.orientation_[0] >> p.orientation_[1] >> p.orientation_[2]; }
  d->visual_obs_by_frame_.resize(n); d->box_obs_by_frame_.resize(n);
  in >> tag >> n;   // features
  for (size_t i = 0; i < n; ++i) { FeatureId id; Position3d p; in >> id >> p[0] >> p[1] >> p[2]; d->initial_feature_positions_[id] = p; }
  in >> tag >> n;   // visual_obs
  for (size_t i = 0; i < n; ++i) { FrameId f; OfflineProblemData::VisualObs o; in >> f >> o.feature_id >> o.camera_id >> o.pixel[0] >> o.pixel[1]; d->visual_obs_by_frame_.at(f).push_back(o); }
  in >> tag >> n;   // objects
  for (size_t i = 0; i < n; ++i) { ObjectId id; std::string cls; RawEllipsoid e; in >> id >> cls; for (double& v : e) in >> v; d->initial_ellipsoids_[id] = e; d->object_class_[id] = cls; }
  in >> tag >> n;   // box_obs
  for (size_t i = 0; i < n; ++i) {
    FrameId f; OfflineProblemData::BoxObs o; double var;
    in >> f >> o.object_id >> o.camera_id >> o.corners[0] >> o.corners[1] >> o.corners[2] >> o.corners[3] >> var;
    o.cov.fill(0.0); for (int k = 0; k < 4; ++k) o.cov[5 * k] = var;
    d->box_obs_by_frame_.at(f).push_back(o);
  }
  in >> tag >> n;   // classes
  for (size_t i = 0; i < n; ++i) {
    std::string name; ObjectDim m; double s[3];
    in >> name >> m[0] >> m[1] >> m[2] >> s[0] >> s[1] >> s[2];
    Covariance<3> c{}; for (int k = 0; k < 3; ++k) c[4 * k] = s[k] * s[k];
    d->shape_priors_by_class_[name] = {m, c};
  }
  return (bool)in;
}

static pose_graph_optimization::OptimizationSolverParams sp(int it, double ftol) {
  pose_graph_optimization::OptimizationSolverParams p;
  p.max_num_iterations_ = it; p.allow_non_monotonic_steps_ = true; p.function_tolerance_ = ftol; p.gradient_tolerance_ = 1e-10; p.parameter_tolerance_ = 1e-8;
  p.initial_trust_region_radius_ = 100; p.max_trust_region_radius_ = 1e4;
  return p;
}

int main(int argc, char** argv) {
  if (argc < 3) { std::cerr << "usage: run_offline_ba scene.txt out.json [options]" << std::endl; return 2; }
  SlidingWindowParams sw;
  int device = 0; std::string csv; bool dump = false, ltm = false, pending = false, masks_of_unexcluded_build = false; FrameId dump_min = 0, dump_max = 0; int excluded_every = 0;
  for (int i = 3; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--window") && i + 1 < argc) sw.local_ba_window_size_ = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "--gba-frequency") && i + 1 < argc) sw.global_ba_frequency_ = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "--device") && i + 1 < argc) device = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--csv") && i + 1 < argc) csv = argv[++i];
    else if (!std::strcmp(argv[i], "--ltm")) ltm = true;
    else if (!std::strcmp(argv[i], "--pending-objects")) pending = true;
    else if (!std::strcmp(argv[i], "--dump-build") && i + 2 < argc) { dump = true; dump_min = std::strtoull(argv[++i], nullptr, 10); dump_max = std::strtoull(argv[++i], nullptr, 10); }
    else if (!std::strcmp(argv[i], "--excluded-every") && i + 1 < argc) excluded_every = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--phase-two-masks")) masks_of_unexcluded_build = true;
  }
  const auto t_main0 = std::chrono::steady_clock::now();
  OfflineProblemData data;
  if (!loadScene(argv[1], &data)) { std::cerr << "could not read scene " << argv[1] << std::endl; return 2; }
  const FrameId max_frame_id = data.getMaxFrameId();

  // config/base7a_2_fallback.json
  pose_graph_optimization::ObjectVisualPoseGraphResidualParams rp;
  rp.object_residual_params_.object_observation_huber_loss_param_ = 0.5; rp.object_residual_params_.shape_dim_prior_factor_huber_loss_param_ = 10;
  rp.object_residual_params_.invalid_ellipsoid_error_val_ = 1000; rp.visual_residual_params_.reprojection_error_huber_loss_param_ = 1.0;
  rp.long_term_map_params_.pair_huber_loss_param_ = 1.0; rp.relative_pose_factor_huber_loss_ = 1.0;
  pose_graph_optimizer::OptimizationFactorsEnabledParams en;
  en.include_object_factors_ = true; en.include_visual_factors_ = true; en.fix_poses_ = en.fix_objects_ = en.fix_visual_features_ = en.fix_ltm_objects_ = false;
  en.poses_prior_to_window_to_keep_constant_ = 5; en.min_object_observations_ = 10; en.min_low_level_feature_observations_ = 5; en.min_low_level_feature_observations_per_frame_ = 50;
  en.use_pose_graph_on_global_ba_ = true; en.use_visual_features_on_global_ba_ = false; en.use_pose_graph_on_final_global_ba_ = true; en.use_visual_features_on_final_global_ba_ = true;
  pose_graph_optimization::PoseGraphPlusObjectsOptimizationParams pgo;
  pgo.relative_pose_factor_huber_loss_ = 5.0; pgo.enable_visual_feats_only_opt_post_pgo_ = true; pgo.enable_visual_non_opt_feature_adjustment_post_pgo_ = true;
  pgo.relative_pose_cov_params_ = {0.1, 0.1, 0.1, 0.1};
  pgo.pgo_optimization_solver_params_ = sp(250, 1e-6); pgo.final_pgo_optimization_solver_params_ = sp(300, 1e-6);
  pgo.post_pgo_vf_adjustment_solver_params_ = sp(250, 1e-6); pgo.final_post_pgo_vf_adjustment_solver_params_ = sp(300, 1e-6); pgo.pre_pgo_tracking_solver_params_ = sp(50, 1e-3);
  pose_graph_optimization::OptimizationIterationParams local_ba, global_ba, final_ba;
  local_ba.phase_one_opt_params_ = sp(50, 1e-3); local_ba.phase_two_opt_params_ = sp(100, 1e-4);
  global_ba.phase_one_opt_params_ = sp(250, 1e-6); global_ba.phase_two_opt_params_ = sp(250, 1e-6);
  final_ba.phase_one_opt_params_ = sp(300, 1e-6); final_ba.phase_two_opt_params_ = sp(300, 1e-6);

  std::ofstream out(argv[2]);
  out << std::setprecision(17);
  if (dump) {
    // flattening only: every frame's data goes into the pose graph, then one build for [min, max]
    MainPgPtr pg = std::make_shared<MainPg>(data.camera_extrinsics_by_camera_, data.camera_intrinsics_by_camera_);
    for (FrameId f = 0; f <= max_frame_id; ++f) addFrameDataToPoseGraph(data, pg, f, rp.relative_pose_cov_params_);
    pose_graph_optimizer::OptimizationScopeParams scope;
    scope.min_low_level_feature_observations_per_frame_ = en.min_low_level_feature_observations_per_frame_;
    scope.poses_prior_to_window_to_keep_constant_ = en.poses_prior_to_window_to_keep_constant_;
    scope.min_object_observations_ = en.min_object_observations_; scope.min_low_level_feature_observations_ = en.min_low_level_feature_observations_;
    scope.min_frame_id_ = dump_min; scope.max_frame_id_ = dump_max;
    FactorInfoSet excluded;
    if (excluded_every > 0) {
      FactorInfoSet all;
      pg->getVisualFeatureFactorIdsBetweenFrameIdsInclusive(dump_min, dump_max, all);
      for (const auto& fi : all) if (fi.second % excluded_every == 0) excluded.insert(fi);
      FactorInfoSet boxes;
      pg->getObservationFactorsBetweenFrameIdsInclusive(dump_min, dump_max, boxes);
      for (const auto& fi : boxes) if (fi.second % excluded_every == 0) excluded.insert(fi);
    }
    const FactorInfoSet excluded_for_masks = excluded;
    if (masks_of_unexcluded_build) excluded.clear();
    obvi::Problem problem(0, /*dry_run=*/true);
    pose_graph_optimizer::ObjectPoseGraphOptimizer optimizer;
    std::optional<OptimizationLogger> no_logger;
    auto info = optimizer.buildPoseGraphOptimization(scope, rp, pg, &problem, no_logger, excluded);
    const obvi::FlatProblem& fp = problem.flat;
    auto arr = [&](const char* name, const auto& v, bool last = false) { out << "\"" << name << "\": ["; for (size_t i = 0; i < v.size(); ++i) out << (i ? "," : "") << (double)v[i]; out << "]" << (last ? "" : ",\n"); };
    out << "{";
    arr("frames", fp.frames); arr("features", fp.features); arr("objects", fp.objects); arr("pose_const", fp.pose_const); arr("point_const", fp.point_const);
    arr("object_const", fp.object_const); arr("rp_pose", fp.rp_pose); arr("rp_point", fp.rp_point); arr("rp_pixel", fp.rp_pixel); arr("bb_obj", fp.bb_obj); arr("bb_pose", fp.bb_pose);
    arr("sp_obj", fp.sp_obj); arr("rl_a", fp.rl_a); arr("rl_b", fp.rl_b);
    {   // factor ids of the residual blocks, family by family (blocks are ordered reprojection, bounding box, shape prior, LTM prior, relative pose)
      std::vector<double> ids[5];
      for (const auto& b : fp.blocks) {
        const int fam = b.first == kReprojectionErrorFactorTypeId ? 0 : b.first == kObjectObservationFactorTypeId ? 1 : b.first == kShapeDimPriorFactorTypeId ? 2 : b.first == kLongTermMapFactorTypeId ? 3 : 4;
        ids[fam].push_back((double)b.second);
      }
      arr("rp_id", ids[0]); arr("bb_id", ids[1]); arr("sp_id", ids[2]); arr("rl_id", ids[4]);
    }
    if (masks_of_unexcluded_build) {
      // --phase-two-masks: the problem above was built WITHOUT the excluded set; these are the masks excludeFromBuiltProblem derives
      // for it (what phase II applies on the device instead of rebuilding)
      pose_graph_optimizer::ObjectPoseGraphOptimizer::PhaseTwoMasks m;
      const bool ok = optimizer.excludeFromBuiltProblem(scope, pg, excluded_for_masks, problem, &m);
      out << "\"masks_ok\": " << (ok ? 1 : 0) << ",\n";
      arr("mask_rp", m.rp); arr("mask_bb", m.bb); arr("mask_sp", m.sp);
      out << "\"mask_n_features\": " << m.n_features << ", \"mask_n_objects\": " << m.n_objects << ",\n";
    }
    out << "\"num_blocks\": " << info.size() << ", \"num_excluded\": " << excluded.size() << "}\n";
    return 0;
  }

  if (pending) {
    // every object of the scene as a pending object: rough estimate = the scene's initial ellipsoid, observations = its boxes,
    // robot poses = the scene's trajectory (constant), pending_object_estimator_params of config/base7a_2_fallback.json
    MainPgPtr pg = std::make_shared<MainPg>(data.camera_extrinsics_by_camera_, data.camera_intrinsics_by_camera_);
    for (FrameId f = 0; f <= max_frame_id; ++f) pg->addFrame(f, data.robot_poses_[f]);
    std::unordered_map<ObjectId, UninitializedEllispoidInfo> info;
    for (FrameId f = 0; f <= max_frame_id; ++f)
      for (const auto& b : data.box_obs_by_frame_[f]) {
        ObjectObservationFactor o; o.frame_id_ = f; o.camera_id_ = b.camera_id; o.object_id_ = b.object_id; o.bounding_box_corners_ = b.corners; o.bounding_box_corners_covariance_ = b.cov;
        info[b.object_id].observation_factors_.push_back(o);
      }
    std::unordered_map<ObjectId, RawEllipsoid> rough;
    for (auto& i : info) { i.second.semantic_class_ = data.object_class_.at(i.first); rough[i.first] = data.initial_ellipsoids_.at(i.first); }
    PendingObjectEstimatorParams pe;
    pe.object_residual_params_ = rp.object_residual_params_;
    pe.solver_params_ = sp(100, 1e-6);
    std::unordered_map<ObjectId, RawEllipsoid> refined;
    obvi_summary summary{};
    const bool ok = refineInitialEstimateForPendingObjects(rough, info, pg, data.shape_priors_by_class_, pe, device, &refined, &summary);
    out << "{\"ok\": " << (ok ? "true" : "false") << ", \"iterations\": " << summary.num_iterations << ", \"initial_cost\": " << summary.initial_cost
        << ", \"final_cost\": " << summary.final_cost << ",\n\"objects\": {";
    bool f0 = true;
    for (const auto& o : refined) { out << (f0 ? "" : ",") << "\"" << o.first << "\": ["; for (int k = 0; k < 7; ++k) out << (k ? "," : "") << o.second[k]; out << "]"; f0 = false; }
    out << "}}\n";
    return ok ? 0 : 1;
  }

  std::function<FrameId(const FrameId&)> window_provider = [&](const FrameId& f) { return provideOptimizationWindow(f, max_frame_id, sw); };
  std::function<bool(const FrameId&)> gba_checker = [&](const FrameId& f) { return f - window_provider(f) > sw.local_ba_window_size_; };   // optimization_runner.h:195-203
  std::function<pose_graph_optimization::OptimizationIterationParams(const FrameId&)> params_provider = [&](const FrameId& f) {              // :204-216
    if (f == max_frame_id) return final_ba;
    if (f % sw.global_ba_frequency_ == 0) return global_ba;
    return local_ba;
  };
  OfflineProblemRunner runner(rp, pgo, window_provider, gba_checker, params_provider, device);
  std::optional<OptimizationLogger> logger;
  if (!csv.empty()) logger.emplace(csv);
  MainPgPtr pg;
  runner.setExtractLongTermMap(ltm);
  const auto t_run0 = std::chrono::steady_clock::now();
  const bool ok = runner.runOptimization(data, en, logger, pg);
  const auto t_run1 = std::chrono::steady_clock::now();
  if (std::getenv("OBVI_HOST_TIMING")) {
    runner.printTiming(std::cerr);
    std::cerr << "driver: scene load + setup " << std::chrono::duration<double, std::milli>(t_run0 - t_main0).count() << " ms, runOptimization "
              << std::chrono::duration<double, std::milli>(t_run1 - t_run0).count() << " ms" << std::endl;
  }
  out << "{\"ok\": " << (ok ? "true" : "false") << ", \"records\": [";
  bool first = true;
  for (const auto& r : runner.records()) {
    out << (first ? "" : ",") << "\n {\"min_frame\": " << r.min_frame << ", \"max_frame\": " << r.max_frame << ", \"kind\": \"" << r.kind << "\", \"iterations\": " << r.iterations
        << ", \"initial_cost\": " << r.initial_cost << ", \"final_cost\": " << r.final_cost << ", \"n_poses\": " << r.n_poses << ", \"n_features\": " << r.n_features
        << ", \"n_objects\": " << r.n_objects << ", \"n_excluded\": " << r.n_excluded << "}";
    first = false;
  }
  out << "],\n\"poses\": [";
  if (pg) for (FrameId f = 0; f <= max_frame_id; ++f) { const RawPose3d p = pg->getRobotPose(f).value(); out << (f ? "," : "") << "[" << p[0] << "," << p[1] << "," << p[2] << "," << p[3] << "," << p[4] << "," << p[5] << "]"; }
  out << "],\n\"objects\": {";
  if (pg) { std::unordered_map<ObjectId, RawEllipsoid> objs; pg->getObjectEstimates(objs); bool f0 = true; for (const auto& o : objs) { out << (f0 ? "" : ",") << "\"" << o.first << "\": ["; for (int k = 0; k < 7; ++k) out << (k ? "," : "") << o.second[k]; out << "]"; f0 = false; } }
  out << "}";
  if (ltm) {   // long-term map: ellipsoid mean + 7x7 marginal covariance per object (the input of the next session's IndependentObjectMapFactor)
    out << ",\n\"long_term_map\": {";
    bool f0 = true;
    for (const auto& e : runner.longTermMap()) {
      out << (f0 ? "" : ",") << "\n \"" << e.object_id_ << "\": {\"mean\": [";
      for (int k = 0; k < 7; ++k) out << (k ? "," : "") << e.ellipsoid_mean_[k];
      out << "], \"covariance\": [";
      for (int k = 0; k < 49; ++k) out << (k ? "," : "") << e.covariance_[k];
      out << "]}";
      f0 = false;
    }
    out << "}";
  }
  out << "}\n";
  return ok ? 0 : 1;
}
