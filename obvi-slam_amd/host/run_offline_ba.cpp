// run_offline_ba.cpp -- driver of the host-side mirror: the shape of the reference's
// offline_object_visual_slam_main / run_opt_from_pg_state for a scene whose associations are given.
//   run_offline_ba <scene.txt> <out.json> [--window W] [--gba-frequency F] [--device D] [--csv ceres_opt_summary.csv] [--ltm]
//   run_offline_ba <scene.txt> <out.json> --dump-build MIN MAX [--excluded-every K] [--phase-two-masks] [--frames-reversed]   (no GPU: flattening only)
//   run_offline_ba <scene.txt> <out.json> ... [--save-checkpoint DIR] [--iteration-log-dir DIR] [--merge-distance M]
//   run_offline_ba --from-checkpoint <pose_graph_state.json> <out.json> [--ltm] [--device D]   the shape of run_opt_from_pg_state: final global BA (+ long-term map) from a checkpoint
//   run_offline_ba --checkpoint-roundtrip <in.json> <out.json>                                  (no GPU) read a pose-graph state and write it back
//   run_offline_ba <scene.txt> <out.json> --pending-objects [--device D]   refineInitialEstimateForPendingObjects over every object of the scene
//   any mode: [--deterministic] fixed-order device sums (bit-identical reruns)   [--analytic-reprojection] the reference's analytic-Jacobian reprojection functor
//   the reference's own files: [--params-config-file config/X.json [--accept-older-config-schema]] [--print-config]
//     [--long-term-map-input F] [--long-term-map-output F]   [--robot-poses-results-file F] [--ellipsoids-results-file F] [--visual-feature-results-file F]
//     run_offline_ba --reference-inputs <out.json> --intrinsics-file A --extrinsics-file B --poses-by-node-id-file C --low-level-feats-dir D   (instead of a scene)
//     run_offline_ba --long-term-map-roundtrip <in.json> <out.json>                                (no GPU)
//   [--sessions-in-process K] K sessions over the scene at once, a host thread each   [--reference-shaped-runner] OfflineProblemRunner<5 types>(15 arguments)   [--creator-rejects-every N] ... with a residual_creator that fails on every N-th observation factor   [--max-frame N]
// Parameter values without a parameter file: config/base7a_2_fallback.json of the reference.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#if defined(__GLIBC__)
#include <malloc.h>
#endif
#if defined(__linux__)
#include <sched.h>
#endif
#include <fstream>
#include <iomanip>
#include <iostream>

#include "obvi_optimization_runner.h"
#include "obvi_config_io.h"
#include "obvi_ltm_io.h"
#include "obvi_results_io.h"
#include "obvi_reference_inputs_io.h"
#include "obvi_visual_feature_front_end.h"
#include "obvi_pending_object_estimator.h"

using namespace vslam_types_refactor;   // NOLINT

// The same scene as raw arrays (scene_io.write_scene_binary): "OBVISCN1", then per section a uint64 count and the section's arrays as
// little-endian fp64 / int64 -- a 3 M-sighting scene (BASELINE config 3) loads in a fraction of a second instead of parsing 200 MB of text.
static bool loadSceneBinary(std::ifstream& in, OfflineProblemData* d) {
  auto rd = [&](void* p, size_t bytes) { in.read(static_cast<char*>(p), (std::streamsize)bytes); return (bool)in; };
  auto count = [&](uint64_t* n) { return rd(n, 8); };
  uint64_t n = 0;
  if (!count(&n)) return false;   // cameras: id, K4, t3, aa3
  for (uint64_t i = 0; i < n; ++i) {
    double v[11]; if (!rd(v, sizeof(v))) return false;
    CameraIntrinsicsMat k; CameraExtrinsics e;
    k.fx = v[1]; k.fy = v[2]; k.cx = v[3]; k.cy = v[4];
    for (int q = 0; q < 3; ++q) { e.transl_[q] = v[5 + q]; e.orientation_[q] = v[8 + q]; }
    d->camera_intrinsics_by_camera_[(CameraId)v[0]] = k; d->camera_extrinsics_by_camera_[(CameraId)v[0]] = e;
  }
  if (!count(&n)) return false;   // frames: pose6
  d->robot_poses_.resize(n);
  { std::vector<double> v(6 * n); if (!rd(v.data(), 8 * v.size())) return false;
    for (uint64_t i = 0; i < n; ++i) { Pose3D& p = d->robot_poses_[i]; for (int q = 0; q < 3; ++q) { p.transl_[q] = v[6 * i + q]; p.orientation_[q] = v[6 * i + 3 + q]; } } }
  d->visual_obs_by_frame_.resize(n); d->box_obs_by_frame_.resize(n);
  if (!count(&n)) return false;   // features: xyz (id = index)
  { std::vector<double> v(3 * n); if (!rd(v.data(), 8 * v.size())) return false;
    d->initial_feature_positions_.reserve(n);
    for (uint64_t i = 0; i < n; ++i) d->initial_feature_positions_[(FeatureId)i] = Position3d{{v[3 * i], v[3 * i + 1], v[3 * i + 2]}}; }
  if (!count(&n)) return false;   // visual_obs: int64 frame, feature, camera; then pixels
  { std::vector<int64_t> ix(3 * n); std::vector<double> px(2 * n);
    if (!rd(ix.data(), 8 * ix.size()) || !rd(px.data(), 8 * px.size())) return false;
    std::vector<uint32_t> per(d->visual_obs_by_frame_.size(), 0);
    for (uint64_t i = 0; i < n; ++i) { if ((uint64_t)ix[3 * i] >= per.size()) return false; ++per[ix[3 * i]]; }
    for (size_t f = 0; f < per.size(); ++f) d->visual_obs_by_frame_[f].reserve(per[f]);
    for (uint64_t i = 0; i < n; ++i) {
      OfflineProblemData::VisualObs o; o.feature_id = (FeatureId)ix[3 * i + 1]; o.camera_id = (CameraId)ix[3 * i + 2]; o.pixel[0] = px[2 * i]; o.pixel[1] = px[2 * i + 1];
      d->visual_obs_by_frame_[ix[3 * i]].push_back(o);
    } }
  if (!count(&n)) return false;   // objects: id, class index, ellipsoid7; classes follow below
  std::vector<std::array<double, 9>> objs(n);
  for (auto& o : objs) if (!rd(o.data(), sizeof(double) * 9)) return false;
  if (!count(&n)) return false;   // box_obs: frame, object, camera, corners4, variance
  for (uint64_t i = 0; i < n; ++i) {
    double v[8]; if (!rd(v, sizeof(v))) return false;
    OfflineProblemData::BoxObs o; o.object_id = (ObjectId)v[1]; o.camera_id = (CameraId)v[2];
    for (int q = 0; q < 4; ++q) o.corners[q] = v[3 + q];
    o.cov.fill(0.0); for (int q = 0; q < 4; ++q) o.cov[5 * q] = v[7];
    if ((uint64_t)v[0] >= d->box_obs_by_frame_.size()) return false;
    d->box_obs_by_frame_[(size_t)v[0]].push_back(o);
  }
  if (!count(&n)) return false;   // classes: name (32 bytes, zero padded), mean3, sd3
  std::vector<std::string> names;
  for (uint64_t i = 0; i < n; ++i) {
    char name[32]; double v[6];
    if (!rd(name, 32) || !rd(v, sizeof(v))) return false;
    name[31] = 0; names.emplace_back(name);
    ObjectDim m; Covariance<3> c{};
    for (int q = 0; q < 3; ++q) { m[q] = v[q]; c[4 * q] = v[3 + q] * v[3 + q]; }
    d->shape_priors_by_class_[names.back()] = {m, c};
  }
  for (const auto& o : objs) {
    if ((size_t)o[1] >= names.size()) return false;
    RawEllipsoid e; for (int q = 0; q < 7; ++q) e[q] = o[2 + q];
    d->initial_ellipsoids_[(ObjectId)o[0]] = e; d->object_class_[(ObjectId)o[0]] = names[(size_t)o[1]];
  }
  return true;
}

static bool loadScene(const std::string& path, OfflineProblemData* d) {
  {
    std::ifstream bin(path, std::ios::binary);
    char magic[8] = {};
    if (bin && bin.read(magic, 8) && !std::memcmp(magic, "OBVISCN1", 8)) return loadSceneBinary(bin, d);
  }
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

static void writeResults(std::ostream& out, bool ok, const LongTermObjectMapAndResults& res, FrameId max_frame_id, bool ltm) {
  out << "{\"ok\": " << (ok ? "true" : "false") << ", \"records\": [";
  bool first = true;
  for (const auto& r : res.records_) {
    out << (first ? "" : ",") << "\n {\"min_frame\": " << r.min_frame << ", \"max_frame\": " << r.max_frame << ", \"kind\": \"" << r.kind << "\", \"iterations\": " << r.iterations
        << ", \"initial_cost\": " << r.initial_cost << ", \"final_cost\": " << r.final_cost << ", \"n_poses\": " << r.n_poses << ", \"n_features\": " << r.n_features
        << ", \"n_objects\": " << r.n_objects << ", \"n_excluded\": " << r.n_excluded << "}";
    first = false;
  }
  out << "],\n\"merge_rounds\": " << res.post_session_merge_rounds_ << ",\n\"poses\": [";
  for (FrameId f = 0; f <= max_frame_id && !res.robot_pose_results_.empty(); ++f) {
    const auto it = res.robot_pose_results_.find(f);
    if (it == res.robot_pose_results_.end()) { out << (f ? "," : "") << "null"; continue; }
    const RawPose3d& p = it->second;
    out << (f ? "," : "") << "[" << p[0] << "," << p[1] << "," << p[2] << "," << p[3] << "," << p[4] << "," << p[5] << "]";
  }
  out << "],\n\"objects\": {";
  { bool f0 = true; for (const auto& o : res.ellipsoid_results_) { out << (f0 ? "" : ",") << "\"" << o.first << "\": ["; for (int k = 0; k < 7; ++k) out << (k ? "," : "") << o.second[k]; out << "]"; f0 = false; } }
  out << "},\n\"covariance_rank_repairs\": [";
  for (size_t i = 0; i < res.covariance_rank_repairs_.size(); ++i) {
    const auto& r = res.covariance_rank_repairs_[i];
    out << (i ? "," : "") << "\n {\"block_kind\": " << r.block_kind << ", \"block_id\": " << r.block_id << ", \"param_idx\": " << r.param_idx << ", \"col_sqnorm\": " << r.col_sqnorm
        << ", \"prior_std_dev\": " << r.prior_std_dev << ", \"retry\": " << r.retry << "}";
  }
  out << "]";
  if (ltm) {   // long-term map: ellipsoid mean + 7x7 marginal covariance per object (the input of the next session's IndependentObjectMapFactor)
    out << ",\n\"long_term_map\": {";
    bool f0 = true;
    for (const auto& e : res.long_term_map_) {
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
}

int main(int argc, char** argv) {
  // this process is the library's only host: its worker threads may stay on the calling thread's block of logical CPUs (a library does not
  // decide that for an application: host_util.h, OBVI_HOST_AFFINITY)
  setenv("OBVI_HOST_AFFINITY", "1", 0);
  if (argc < 3) { std::cerr << "usage: run_offline_ba scene.txt out.json [options] | --from-checkpoint state.json out.json | --checkpoint-roundtrip in.json out.json" << std::endl; return 2; }
  if (!std::strcmp(argv[1], "--long-term-map-roundtrip")) {   // no GPU: the reader and the writer of obvi_ltm_io.h
    if (argc < 4) return 2;
    LongTermObjectMapFile map;
    if (!readLongTermObjectMapFromFile(argv[2], map)) return 1;
    return writeLongTermObjectMapToFile(argv[3], map) ? 0 : 1;
  }
  if (!std::strcmp(argv[1], "--checkpoint-roundtrip")) {   // no GPU: the reader and the writer of obvi_checkpoint_io.h
    if (argc < 4) return 2;
    ObjectAndReprojectionFeaturePoseGraphState st;
    if (!readPoseGraphStateFromFile(argv[2], st)) return 1;
    return outputPoseGraphStateToFile(st, argv[3]) ? 0 : 1;
  }
  const bool from_checkpoint = !std::strcmp(argv[1], "--from-checkpoint");
  if (from_checkpoint && argc < 4) return 2;
  const int first_opt = from_checkpoint ? 4 : 3;
  const char* scene_path = from_checkpoint ? nullptr : argv[1];
  const char* out_path = from_checkpoint ? argv[3] : argv[2];
  FullOVSLAMConfig config = FullOVSLAMConfig::base7a2Fallback();   // config/base7a_2_fallback.json
  // --params-config-file F (the reference's --params_config_file, offline_object_visual_slam_main.cpp:731): one of the reference's config/*.json; read before the
  // other options, which then override single values.  --accept-older-config-schema: files older than schema 14 (the reference's reader refuses those).
  bool config_from_file = false, print_config = false;
  {
    const char* config_file = nullptr; bool older = false;
    for (int i = first_opt; i < argc; ++i) {
      if (!std::strcmp(argv[i], "--params-config-file") && i + 1 < argc) config_file = argv[i + 1];
      else if (!std::strcmp(argv[i], "--accept-older-config-schema")) older = true;
      else if (!std::strcmp(argv[i], "--print-config")) print_config = true;
    }
    if (config_file != nullptr) {
      try { readConfiguration(config_file, config, older); config_from_file = true; }
      catch (const std::exception& e) { std::cerr << "run_offline_ba: " << e.what() << std::endl; return 3; }
    }
  }
  SlidingWindowParams& sw = config.sliding_window_params_;
  bool global_ba_only = false;
  int device = 0; std::string csv, checkpoint_dir, iteration_log_dir; bool dump = false, ltm = false, pending = false, masks_of_unexcluded_build = false, visual_front_end = false, front_end_only = false;
  RunnerHooks hooks; bool count_visualization_calls = false; int sessions_in_process = 1; ReferenceInputFiles reference_inputs; std::string ltm_in_path, ltm_out_path, robot_poses_results_file, ellipsoids_results_file, visual_feature_results_file;
  VisualFeatureFrontendParams front_end_params;   // visual_feature_params of config/base7a_2_fallback.json: pixel parallax 5 px enforced, pose parallax not
  front_end_params.enforce_min_robot_pose_parallax_requirement_ = false; FrameId dump_min = 0, dump_max = 0; int excluded_every = 0; bool frames_reversed = false;
  for (int i = first_opt; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--window") && i + 1 < argc) sw.local_ba_window_size_ = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "--gba-frequency") && i + 1 < argc) sw.global_ba_frequency_ = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "--device") && i + 1 < argc) device = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--csv") && i + 1 < argc) csv = argv[++i];
    else if (!std::strcmp(argv[i], "--ltm")) ltm = true;
    else if (!std::strcmp(argv[i], "--global-ba")) global_ba_only = true;   // the whole scene goes into the pose graph, then ONE optimisation: the final global BA
    else if (!std::strcmp(argv[i], "--save-checkpoint") && i + 1 < argc) checkpoint_dir = argv[++i];
    else if (!std::strcmp(argv[i], "--iteration-log-dir") && i + 1 < argc) iteration_log_dir = argv[++i];
    else if (!std::strcmp(argv[i], "--merge-distance") && i + 1 < argc) config.post_session_object_merge_params_.max_merge_distance_ = std::atof(argv[++i]);
    else if (!std::strcmp(argv[i], "--pending-objects")) pending = true;
    else if (!std::strcmp(argv[i], "--analytic-reprojection")) obvi::backendOptions().reprojection_variant = OBVI_REPROJECTION_ANALYTIC;   // ReprojectionCostFunctorAnalyticJacobian instead of ReprojectionCostFunctor
    else if (!std::strcmp(argv[i], "--deterministic")) obvi::backendOptions().deterministic = true;   // fixed-order device sums: reruns are bit-identical
    else if (!std::strcmp(argv[i], "--visual-front-end")) visual_front_end = true;
    else if (!std::strcmp(argv[i], "--front-end-only")) { visual_front_end = true; front_end_only = true; }
    else if (!std::strcmp(argv[i], "--no-epipolar")) front_end_params.enforce_epipolar_error_requirement_ = false;
    else if (!std::strcmp(argv[i], "--pose-parallax")) front_end_params.enforce_min_robot_pose_parallax_requirement_ = true;
    else if (!std::strcmp(argv[i], "--dump-build") && i + 2 < argc) { dump = true; dump_min = std::strtoull(argv[++i], nullptr, 10); dump_max = std::strtoull(argv[++i], nullptr, 10); }
    else if (!std::strcmp(argv[i], "--excluded-every") && i + 1 < argc) excluded_every = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--frames-reversed")) frames_reversed = true;   // --dump-build: the frames' sightings enter the pose graph last frame first (factor ids descend with the frame)
    else if (!std::strcmp(argv[i], "--phase-two-masks")) masks_of_unexcluded_build = true;
    else if (!std::strcmp(argv[i], "--reference-shaped-runner")) hooks.reference_shaped_runner_ = true;   // OfflineProblemRunner<5 types>(15 arguments), as the reference constructs it
    else if (!std::strcmp(argv[i], "--creator-rejects-every") && i + 1 < argc) { hooks.reference_shaped_runner_ = true; hooks.creator_rejects_every_ = std::atoi(argv[++i]); }   // a residual_creator that fails on every N-th observation factor (the per-factor seam)
    else if (!std::strcmp(argv[i], "--max-frame") && i + 1 < argc) { hooks.limit_trajectory_eval_params_.should_limit_trajectory_evaluation_ = true; hooks.limit_trajectory_eval_params_.max_frame_id_ = std::strtoull(argv[++i], nullptr, 10); }
    else if (!std::strcmp(argv[i], "--count-visualization-calls")) count_visualization_calls = true;
    else if (!std::strcmp(argv[i], "--params-config-file") && i + 1 < argc) ++i;   // (read above)
    else if (!std::strcmp(argv[i], "--intrinsics-file") && i + 1 < argc) reference_inputs.intrinsics_file = argv[++i];                  // with `--reference-inputs` in the scene's place:
    else if (!std::strcmp(argv[i], "--extrinsics-file") && i + 1 < argc) reference_inputs.extrinsics_file = argv[++i];                  // the reference's --intrinsics_file, --extrinsics_file,
    else if (!std::strcmp(argv[i], "--poses-by-node-id-file") && i + 1 < argc) reference_inputs.poses_by_node_id_file = argv[++i];      // --poses_by_node_id_file, --low_level_feats_dir
    else if (!std::strcmp(argv[i], "--low-level-feats-dir") && i + 1 < argc) reference_inputs.low_level_feats_dir = argv[++i];
    else if (!std::strcmp(argv[i], "--robot-poses-results-file") && i + 1 < argc) robot_poses_results_file = argv[++i];          // the reference's --robot_poses_results_file
    else if (!std::strcmp(argv[i], "--ellipsoids-results-file") && i + 1 < argc) ellipsoids_results_file = argv[++i];            // ... --ellipsoids_results_file
    else if (!std::strcmp(argv[i], "--visual-feature-results-file") && i + 1 < argc) visual_feature_results_file = argv[++i];    // ... --visual_feature_results_file
    else if (!std::strcmp(argv[i], "--long-term-map-input") && i + 1 < argc) ltm_in_path = argv[++i];     // the reference's --long_term_map_input: the previous session's map file
    else if (!std::strcmp(argv[i], "--long-term-map-output") && i + 1 < argc) { ltm_out_path = argv[++i]; ltm = true; }   // ... --long_term_map_output (implies --ltm)
    else if (!std::strcmp(argv[i], "--accept-older-config-schema") || !std::strcmp(argv[i], "--print-config")) {}
    else if (!std::strcmp(argv[i], "--sessions-in-process") && i + 1 < argc) sessions_in_process = std::max(1, std::atoi(argv[++i]));   // K sessions over the scene at once, a host thread each (results: out, out.1, ...)
  }
  if (print_config) { writeConfigurationToStream(std::cout, config); return 0; }
  {   // offline_object_visual_slam_main.cpp:1008-1023: a global BA needs visual features or the pose graph (or both); the reference exits
    const auto& e = config.optimization_factors_enabled_params_;
    if (!e.use_visual_features_on_global_ba_ && !e.use_pose_graph_on_global_ba_) { std::cerr << "Must have either visual features or pose graph (or both) for global ba; review/fix your config" << std::endl; return 1; }
    if (!e.use_visual_features_on_final_global_ba_ && !e.use_pose_graph_on_final_global_ba_) { std::cerr << "Must have either visual features or pose graph (or both) for final global ba; review/fix your config" << std::endl; return 1; }
  }   // the configuration in force (file + options), in the parameter file's layout
  if (!iteration_log_dir.empty()) IterationLoggerFactory::setLoggingDirectory(iteration_log_dir);   // offline_object_visual_slam_main.cpp:676
  const auto t_main0 = std::chrono::steady_clock::now();
#if defined(__GLIBC__)
  // A window's flat arrays (hundreds of kB) are allocated and freed once per frame, half of them on the runner's second thread: with glibc's defaults they
  // are mmap'ed / the arena is trimmed every time and every frame pays the page faults again (300-frame session 1.9 -> 1.7 s with these three).  A host
  // program's choice, not the library's.
  mallopt(M_MMAP_THRESHOLD, 1 << 30); mallopt(M_TRIM_THRESHOLD, 1 << 30); mallopt(M_TOP_PAD, 64 << 20);
#endif
  // the device handle of the session is created beside the scene load and the pose-graph fill (HIP runtime start + allocations: ~0.1 s)
  // (two: the runner plans the next window / the global BA on a second handle beside the solve that is running, unless OBVI_HOST_PLAN_AHEAD=0)
  if (sessions_in_process > 1 && !std::getenv("OBVI_HOST_PLAN_AHEAD")) {
    // planning ahead keeps a second thread busy per session: it pays while the host has CPUs to spare (16 CPUs: 435 against 408 frames/s at K = 4, 465 against 578 at K = 8)
    unsigned cpus = std::max(1u, std::thread::hardware_concurrency());
#if defined(__linux__)
    { cpu_set_t set; if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = (unsigned)std::max(1, CPU_COUNT(&set)); }
    { std::ifstream quota("/sys/fs/cgroup/cpu.max"); std::string q; double period = 0; if (quota >> q >> period && q != "max" && period > 0) cpus = std::min<unsigned>(cpus, (unsigned)std::max(1.0, std::floor(std::atof(q.c_str()) / period))); }
#endif
    if (4u * (unsigned)sessions_in_process > cpus) setenv("OBVI_HOST_PLAN_AHEAD", "0", 1);
  }
  if (!dump && !front_end_only) {
    const bool plan_ahead = !std::getenv("OBVI_HOST_PLAN_AHEAD") || std::atoi(std::getenv("OBVI_HOST_PLAN_AHEAD")) != 0;
    for (int k = 0; k < sessions_in_process * (plan_ahead ? 3 : 1) && k < 8; ++k)   // (planned ahead: the session's two problems + the pose-graph stage's)
      obvi::HandlePool::instance().warm(obvi::makeHandleOptions(device));
  }
  OfflineProblemData data;
  MainPgPtr checkpoint_graph;
  if (from_checkpoint) {
    // run_opt_from_pg_state.cpp:160-312: the checkpoint's pose graph is handed out by the pose_graph_creator, no frame data is added,
    // the optimisation starts at (and consists of) the last frame: final global BA, then the long-term map
    ObjectAndReprojectionFeaturePoseGraphState st;
    if (!readPoseGraphStateFromFile(argv[2], st)) return 2;
    checkpoint_graph = MainPg::createObjectAndReprojectionFeaturePoseGraphFromState(st);
    const auto& L = st.reprojection_low_level_feature_pose_graph_state_.low_level_pg_state_;
    data.camera_intrinsics_by_camera_ = L.camera_intrinsics_by_camera_; data.camera_extrinsics_by_camera_ = L.camera_extrinsics_by_camera_;
    data.robot_poses_.resize(L.max_frame_id_ + 1);
    for (const auto& p : L.robot_poses_) if (p.first < data.robot_poses_.size()) data.robot_poses_[p.first] = convertToPose3D(p.second);
    data.visual_obs_by_frame_.resize(data.robot_poses_.size()); data.box_obs_by_frame_.resize(data.robot_poses_.size());
    data.shape_priors_by_class_ = st.obj_only_pose_graph_state_.mean_and_cov_by_semantic_class_;
    for (const auto& e : st.obj_only_pose_graph_state_.semantic_class_for_object_) data.object_class_[e.first] = e.second;   // (the map file written at the end names every object's class)
  } else if (!std::strcmp(scene_path, "--reference-inputs")) {   // the reference executable's own input files instead of a scene (obvi_reference_inputs_io.h): visual-feature sessions
    std::string error;
    LimitTrajectoryEvaluationParams limit = hooks.limit_trajectory_eval_params_;
    if (!limit.should_limit_trajectory_evaluation_ && config_from_file) limit = config.limit_traj_eval_params_;
    if (!loadReferenceInputs(reference_inputs, limit, &data, &error)) { std::cerr << "run_offline_ba --reference-inputs: " << error << std::endl; return 2; }
  } else if (!loadScene(scene_path, &data)) { std::cerr << "could not read scene " << scene_path << std::endl; return 2; }
  if (!ltm_in_path.empty()) {
    // offline_object_visual_slam_main.cpp:789-805: the map of the previous session.  Every mapped ellipsoid enters the pose graph under its id with its estimate and
    // the prior (estimate, covariance) before the first frame (runOptimization); the scene's sightings of those ids then are sightings of map objects, and objects
    // the scene sees for the first time take the ids behind the map's.
    LongTermObjectMapFile map;
    if (!readLongTermObjectMapFromFile(ltm_in_path, map)) return 2;
    data.long_term_map_.clear();
    for (const auto& e : map.ellipsoid_results_) {
      data.long_term_map_.push_back(LongTermMapObjectPrior{e.first, e.second.second, map.ellipsoid_covariances_.at(e.first)});
      data.object_class_[e.first] = e.second.first;
    }
  }
  if (config_from_file) {   // what the reference takes from its parameter file and this driver otherwise from the scene: pixel noise, shape priors by class, trajectory limit
    data.reprojection_error_std_dev_ = config.visual_feature_params_.reprojection_error_std_dev_;
    for (const auto& e : config.shape_dimension_priors_) data.shape_priors_by_class_[e.first] = e.second;
    if (!hooks.limit_trajectory_eval_params_.should_limit_trajectory_evaluation_) hooks.limit_trajectory_eval_params_ = config.limit_traj_eval_params_;
  }
  const auto t_loaded = std::chrono::steady_clock::now();
  const FrameId max_frame_id = data.getMaxFrameId();
  const pose_graph_optimization::ObjectVisualPoseGraphResidualParams& rp = config.object_visual_pose_graph_residual_params_;
  const pose_graph_optimizer::OptimizationFactorsEnabledParams& en = config.optimization_factors_enabled_params_;

  std::ofstream out(out_path);
  out << std::setprecision(17);
  if (dump) {
    // flattening only: every frame's data goes into the pose graph, then one build for [min, max]
    MainPgPtr pg = checkpoint_graph ? checkpoint_graph : std::make_shared<MainPg>(data.camera_extrinsics_by_camera_, data.camera_intrinsics_by_camera_);
    if (!checkpoint_graph) {
      const VisualFeatureAdder none = [](const OfflineProblemData&, const MainPgPtr&, const FrameId&, const FrameId&) { return true; };
      for (FrameId f = 0; f <= max_frame_id; ++f) addFrameDataToPoseGraph(data, pg, f, rp.relative_pose_cov_params_, frames_reversed ? none : VisualFeatureAdder(nullptr));
      if (frames_reversed)   // the sightings enter last frame first, as when a front end initialises a feature late and adds its earlier sightings then
        for (FrameId k = 0; k <= max_frame_id; ++k) {
          const FrameId f = max_frame_id - k;
          if (f >= data.visual_obs_by_frame_.size()) continue;
          for (const auto& o : data.visual_obs_by_frame_[f]) {
            if (!pg->hasFeature(o.feature_id)) pg->addFeature(o.feature_id, data.initial_feature_positions_.at(o.feature_id));
            pg->addVisualFactor(ReprojectionErrorFactor{f, o.feature_id, o.camera_id, o.pixel, data.reprojection_error_std_dev_});
          }
        }
    }
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
    {   // the whole flat problem: values of the parameter blocks and the measurement data (what solveOptimization uploads)
      std::vector<double> poses, points, objects;
      for (const double* q : fp.pose_ptrs) poses.insert(poses.end(), q, q + 6);
      for (const double* q : fp.point_ptrs) points.insert(points.end(), q, q + 3);
      for (const double* q : fp.object_ptrs) objects.insert(objects.end(), q, q + 7);
      arr("poses", poses); arr("points", points); arr("object_values", objects); arr("cam_K", fp.cam_K); arr("cam_ext", fp.cam_ext);
      arr("rp_cam", fp.rp_cam); arr("rp_sigma", fp.rp_sigma); arr("bb_cam", fp.bb_cam); arr("bb_corners", fp.bb_corners); arr("bb_cov", fp.bb_cov);
      arr("sp_mean", fp.sp_mean); arr("sp_cov", fp.sp_cov); arr("rl_t", fp.rl_t); arr("rl_aa", fp.rl_aa); arr("rl_cov", fp.rl_cov);
    }
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
    { std::vector<double> ids; for (size_t i = 0; i < fp.rp_pose.size(); ++i) ids.push_back((double)info.at((obvi::ResidualBlockId)i).second); arr("rp_factor_ids", ids); }
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
    pe.solver_params_ = FullOVSLAMConfig::solverParams(100, 1e-6);
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

  // --visual-front-end: the frame's visual observations go through VisualFeatureFrontend (epipolar votes, parallax) instead of straight
  // into the pose graph; --front-end-only: no optimisation, the frames are added one after the other and the decisions are reported
  obvi_ba_handle* front_end_handle = nullptr;
  std::unique_ptr<VisualFeatureFrontend> front_end;
  VisualFeatureAdder visual_adder;
  if (visual_front_end) {
    const obvi_ba_options opt = obvi::makeHandleOptions(device);
    if (obvi_ba_create(&opt, &front_end_handle) != 0) { std::cerr << "no device handle for the visual front end" << std::endl; return 1; }
    const SlidingWindowParams sw2 = config.sliding_window_params_;
    front_end = std::make_unique<VisualFeatureFrontend>(front_end_handle, [sw2, max_frame_id](const FrameId& f) { return f - provideOptimizationWindow(f, max_frame_id, sw2) > sw2.local_ba_window_size_; },
                                                        front_end_params);
    visual_adder = [&](const OfflineProblemData& d, const MainPgPtr& pg, const FrameId& mn, const FrameId& mx) { return front_end->addVisualFeatureObservations(d, pg, mn, mx); };
  }
  auto front_end_report = [&](std::ostream& o) {
    o << "{\"added\": " << front_end->numAddedFeatures() << ", \"pending\": " << front_end->numPendingFeatures() << ", \"pending_initialized\": " << front_end->numPendingInitializedFeatures()
      << ", \"rejected_factors\": " << front_end->numRejectedFactors() << ", \"vote_questions\": " << front_end->numVoteQuestions() << ", \"vote_calls\": " << front_end->numVoteCalls() << "}";
  };
  if (front_end_only) {
    MainPgPtr pg = std::make_shared<MainPg>(data.camera_extrinsics_by_camera_, data.camera_intrinsics_by_camera_);
    std::vector<size_t> feats_after, factors_after;
    for (FrameId f = 0; f <= max_frame_id; ++f) {
      addFrameDataToPoseGraph(data, pg, f, rp.relative_pose_cov_params_, visual_adder, f == 0 ? 0 : provideOptimizationWindow(f, max_frame_id, config.sliding_window_params_));
      FactorInfoSet all;
      pg->getVisualFeatureFactorIdsBetweenFrameIdsInclusive(0, f, all);
      feats_after.push_back(pg->featurePositions().size()); factors_after.push_back(all.size());
    }
    out << "{\"front_end\": "; front_end_report(out);
    out << ",\n\"features_after_frame\": ["; for (size_t i = 0; i < feats_after.size(); ++i) out << (i ? "," : "") << feats_after[i];
    out << "],\n\"factors_after_frame\": ["; for (size_t i = 0; i < factors_after.size(); ++i) out << (i ? "," : "") << factors_after[i];
    // every reprojection factor that entered the graph: frame, feature, camera
    out << "],\n\"factors\": [";
    std::vector<std::array<uint64_t, 3>> fs;
    pg->forEachVisualFactorBetweenFrameIdsInclusive(0, max_frame_id, [&](FeatureFactorId, const ReprojectionErrorFactor& f) { fs.push_back({{f.frame_id_, f.feature_id_, f.camera_id_}}); });
    std::sort(fs.begin(), fs.end());
    for (size_t i = 0; i < fs.size(); ++i) out << (i ? "," : "") << "[" << fs[i][0] << "," << fs[i][1] << "," << fs[i][2] << "]";
    out << "]}\n";
    front_end.reset(); obvi_ba_destroy(front_end_handle);
    return 0;
  }

  if (sessions_in_process > 1) {
    // K sessions in ONE process, a host thread each (SURVEY 8e "2 sessions per GPU"): the streams of one process share the device kernel by kernel, where the
    // queues of K processes take turns (scripts/concurrent_sessions.py measures both).  Every session has its own runner, pose graph, device handles and
    // summary CSV; the scene is read-only and shared, and so are the library's host threads.  The plain session only: no checkpoint, no front end, no iteration logs.
    if (from_checkpoint || global_ba_only || visual_front_end || pending || !iteration_log_dir.empty()) { std::cerr << "--sessions-in-process: plain sessions only" << std::endl; return 2; }
    out.close();
    std::vector<int> oks((size_t)sessions_in_process, 0);
    std::vector<double> seconds((size_t)sessions_in_process, 0.0);
    std::vector<std::thread> threads;
    const auto t_all0 = std::chrono::steady_clock::now();
    for (int k = 0; k < sessions_in_process; ++k)
      threads.emplace_back([&, k]() {
        const std::string suffix = k == 0 ? std::string() : "." + std::to_string(k);
        std::optional<OptimizationLogger> session_logger;
        if (!csv.empty()) session_logger.emplace(csv + suffix);
        LongTermObjectMapAndResults session_results;
        RunnerHooks session_hooks = hooks;
        const auto t0 = std::chrono::steady_clock::now();
        const bool ok = runFullOptimization(session_logger, config, data, nullptr, std::string(), session_results, 0, true, device, ltm, nullptr, nullptr, &session_hooks);
        seconds[(size_t)k] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::ofstream session_out(out_path + suffix);
        session_out << std::setprecision(17);
        writeResults(session_out, ok, session_results, max_frame_id, ltm);
        oks[(size_t)k] = ok ? 1 : 0;
      });
    for (auto& t : threads) t.join();
    const double all_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_all0).count();
    std::cout << "{\"sessions_in_process\": " << sessions_in_process << ", \"all_sessions_s\": " << all_s << ", \"session_s\": [";
    for (int k = 0; k < sessions_in_process; ++k) std::cout << (k ? ", " : "") << seconds[(size_t)k];
    std::cout << "], \"frames\": " << (max_frame_id + 1) << "}" << std::endl;
    obvi::HandlePool::instance().drain();
    return std::all_of(oks.begin(), oks.end(), [](int v) { return v == 1; }) ? 0 : 1;
  }

  std::optional<OptimizationLogger> logger;
  // offline_object_visual_slam_main.cpp:672-681: the summary CSV lives in the logging directory, next to the per-iteration CSVs
  if (csv.empty() && !iteration_log_dir.empty()) csv = (iteration_log_dir.back() == '/' ? iteration_log_dir : iteration_log_dir + "/") + "ceres_opt_summary.csv";
  if (!csv.empty()) logger.emplace(csv);
  std::function<void(const OfflineProblemData&, MainPgPtr&)> creator;
  if (from_checkpoint) creator = [&](const OfflineProblemData&, MainPgPtr& pg) { pg = checkpoint_graph; };
  if (global_ba_only && !from_checkpoint) {
    // BASELINE config 3 "run as specified" through this layer: every frame's data enters the pose graph (the data adder of
    // offline_problem_runner.h:376-417, no optimisation in between), then the runner is started AT the last frame -- window provider ->
    // global BA -> runPgoPlusEllipsoids + the two-phase optimisation with global_ba_iteration_params, i.e. what run_opt_from_pg_state does
    // with a checkpoint (run_opt_from_pg_state.cpp:160-312) without the JSON in between
    checkpoint_graph = std::make_shared<MainPg>(data.camera_extrinsics_by_camera_, data.camera_intrinsics_by_camera_);
    for (FrameId f = 0; f <= max_frame_id; ++f) addFrameDataToPoseGraph(data, checkpoint_graph, f, rp.relative_pose_cov_params_, visual_adder);
    creator = [&](const OfflineProblemData&, MainPgPtr& pg) { pg = checkpoint_graph; };
  }
  const bool start_at_end = from_checkpoint || global_ba_only;
  LongTermObjectMapAndResults results;
  const auto t_run0 = std::chrono::steady_clock::now();
  std::array<size_t, 6> visualization_calls{};   // by VisualizationTypeEnum
  if (count_visualization_calls)
    hooks.visualization_callback_ = [&](const OfflineProblemData&, const MainPgPtr&, const FrameId&, const FrameId&, const VisualizationTypeEnum& when, const int&) { visualization_calls[(size_t)when]++; };
  const bool ok = runFullOptimization(logger, config, data, creator, checkpoint_dir, results, start_at_end ? max_frame_id : 0, !start_at_end, device, ltm, nullptr, visual_adder, &hooks);
  if (count_visualization_calls || hooks.reference_shaped_runner_) {
    std::cerr << "runner_hooks {\"visualization_calls\": [";
    for (size_t k = 0; k < visualization_calls.size(); ++k) std::cerr << (k ? ", " : "") << visualization_calls[k];
    std::cerr << "], \"ignored_hooks\": [";
    for (size_t k = 0; k < hooks.ignored_hooks_.size(); ++k) std::cerr << (k ? ", " : "") << "\"" << hooks.ignored_hooks_[k] << "\"";
    std::cerr << "], \"creator_calls\": " << hooks.creator_calls_ << ", \"creator_rejections\": " << hooks.creator_rejections_ << ", \"refresh_calls\": " << hooks.refresh_calls_
              << ", \"factors_left_out\": " << hooks.factors_left_out_ << "}" << std::endl;
  }
  if (front_end) { std::cerr << "front_end "; front_end_report(std::cerr); std::cerr << std::endl; front_end.reset(); obvi_ba_destroy(front_end_handle); }
  const auto t_run1 = std::chrono::steady_clock::now();
  IterationLoggerFactory::getInstance().writeAllIterationLoggerStates();                                                     // offline_object_visual_slam_main.cpp:1108
  if (std::getenv("OBVI_HOST_TIMING"))
    std::cerr << "driver: scene load + setup " << std::chrono::duration<double, std::milli>(t_run0 - t_main0).count() << " ms, runFullOptimization "
              << std::chrono::duration<double, std::milli>(t_run1 - t_run0).count() << " ms" << std::endl;
  obvi::HandlePool::instance().drain();
  writeResults(out, ok, results, max_frame_id, ltm);
  auto class_of = [&](ObjectId id) { const auto it = data.object_class_.find(id); return it == data.object_class_.end() ? std::string() : it->second; };
  if (ok) {   // offline_object_visual_slam_main.cpp:1046-1054, 1094-1102
    bool written = true;
    if (!robot_poses_results_file.empty()) written = writeTextFile(robot_poses_results_file, writeRobotPoseResultsToString(results.robot_pose_results_)) && written;
    if (!ellipsoids_results_file.empty()) {
      std::map<ObjectId, std::pair<std::string, RawEllipsoid>> ellipsoids;
      for (const auto& e : results.ellipsoid_results_) ellipsoids[e.first] = {class_of(e.first), e.second};
      written = writeTextFile(ellipsoids_results_file, writeEllipsoidResultsToString(ellipsoids)) && written;
    }
    if (!visual_feature_results_file.empty()) written = writeTextFile(visual_feature_results_file, writeVisualFeatureResultsToString(results.visual_feature_results_)) && written;
    if (!written) { std::cerr << "could not write a results file" << std::endl; return 1; }
  }
  if (!ltm_out_path.empty()) {   // offline_object_visual_slam_main.cpp:1056-1076: written whether or not the optimisation succeeded (a failed session of a chain hands the previous map on)
    LongTermObjectMapFile map;
    for (const auto& e : results.long_term_map_) {
      map.ellipsoid_results_[e.object_id_] = {class_of(e.object_id_), e.ellipsoid_mean_};
      map.ellipsoid_covariances_[e.object_id_] = e.covariance_;
    }
    for (const auto& e : results.ellipsoid_results_) map.prev_traj_est_ellipsoid_results_[e.first] = {class_of(e.first), e.second};
    if (map.ellipsoid_results_.empty() && config.ltm_tunable_params_.fallback_to_prev_for_failed_extraction_ && !ltm_in_path.empty()) {   // :1057-1068: nothing extracted: the map the session started from
      std::cerr << "Long term map extraction failed; falling back to previous long-term map" << std::endl;
      if (!readLongTermObjectMapFromFile(ltm_in_path, map)) return 1;
    }
    if (!writeLongTermObjectMapToFile(ltm_out_path, map)) { std::cerr << "could not write the long-term map to " << ltm_out_path << std::endl; return 1; }
  }
  if (global_ba_only) {   // one JSON line for bench.py's end_to_end_cpp: wall clock of this process, stage by stage
    const auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::cout << "{\"scene_load_ms\": " << ms(t_main0, t_loaded) << ", \"pose_graph_ms\": " << ms(t_loaded, t_run0) << ", \"run_full_optimization_ms\": " << ms(t_run0, t_run1) << ", \"ok\": " << (ok ? "true" : "false")
              << ", \"records\": [";
    bool first = true;
    for (const auto& r : results.records_) {
      std::cout << (first ? "" : ", ") << "{\"kind\": \"" << r.kind << "\", \"iterations\": " << r.iterations << ", \"initial_cost\": " << r.initial_cost << ", \"final_cost\": " << r.final_cost
                << ", \"n_poses\": " << r.n_poses << ", \"n_features\": " << r.n_features << ", \"n_objects\": " << r.n_objects << ", \"n_excluded\": " << r.n_excluded << "}";
      first = false;
    }
    std::cout << "]}" << std::endl;
  }
  return ok ? 0 : 1;
}
