// obvi_pending_object_estimator.h -- host-side mirror of
//   refineInitialEstimateForPendingObjects   src/refactoring/bounding_box_frontend/pending_object_estimator.cpp:11-151
//   PendingObjectEstimatorParams             include/refactoring/bounding_box_frontend/pending_object_estimator.h
// The reference puts every pending ellipsoid, its bounding-box factors (Huber) and its shape prior into one ceres::Problem,
// sets every robot pose it touched constant and solves once.  Same here through the C ABI: objects variable, poses constant,
// no features -- a reduced system of independent 7x7 blocks.  Same error behaviour: factors whose camera is unknown are
// skipped with a message (:58-71); a failed solve is fatal in the reference (exit(1), :132-134) and `false` here.
#ifndef OBVI_HOST_PENDING_OBJECT_ESTIMATOR_H_
#define OBVI_HOST_PENDING_OBJECT_ESTIMATOR_H_

#include <obvi_ba.h>

#include <iostream>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "obvi_optimizer.h"
#include "obvi_params.h"
#include "obvi_pose_graph.h"

namespace vslam_types_refactor {

struct PendingObjectEstimatorParams {
  pose_graph_optimization::ObjectResidualParams object_residual_params_;
  pose_graph_optimization::OptimizationSolverParams solver_params_;
};
// the part of UninitializedEllispoidInfo the estimator reads (feature_based_bounding_box_front_end.h)
struct UninitializedEllispoidInfo {
  std::string semantic_class_;
  std::vector<ObjectObservationFactor> observation_factors_;
};

template <class PoseGraphPtr>
inline bool refineInitialEstimateForPendingObjects(const std::unordered_map<ObjectId, RawEllipsoid>& rough_initial_estimates,
                                                   const std::unordered_map<ObjectId, UninitializedEllispoidInfo>& uninitialized_obj_info,
                                                   const PoseGraphPtr& pose_graph,
                                                   const std::unordered_map<std::string, std::pair<ObjectDim, Covariance<3>>>& mean_and_cov_by_semantic_class,
                                                   const PendingObjectEstimatorParams& estimator_params, int device_id,
                                                   std::unordered_map<ObjectId, RawEllipsoid>* updated_estimates, obvi_summary* summary_out = nullptr,
                                                   const std::string& optimization_identifier = "" /* "<frame id>_<camera id>" of the triggering observation (.cpp:134-135) */) {
  std::unordered_map<FrameId, RawPose3d> robot_pose_estimates;
  pose_graph->getRobotPoseEstimates(robot_pose_estimates);
  // flat problem: objects in id order, the frames that carry an observation, the cameras of the pose graph
  std::map<ObjectId, uint32_t> obj_index;
  std::vector<double> objects;
  for (const auto& est : rough_initial_estimates) obj_index[est.first] = 0;
  for (auto& o : obj_index) { o.second = (uint32_t)(objects.size() / 7); const RawEllipsoid& e = rough_initial_estimates.at(o.first); objects.insert(objects.end(), e.begin(), e.end()); }
  std::map<CameraId, uint16_t> cam_index;
  std::vector<double> cam_K, cam_ext;
  std::map<FrameId, uint32_t> pose_index;
  std::vector<double> poses;
  std::vector<uint32_t> bb_obj, bb_pose, sp_obj; std::vector<uint16_t> bb_cam; std::vector<double> bb_corners, bb_cov, sp_mean, sp_cov;
  for (const auto& obj_info : uninitialized_obj_info) {
    const auto oi = obj_index.find(obj_info.first);
    if (oi == obj_index.end()) { std::cerr << "pending object " << obj_info.first << " has no rough estimate" << std::endl; return false; }   // raw_ests.at() throws in the reference
    for (const ObjectObservationFactor& obs : obj_info.second.observation_factors_) {
      CameraExtrinsics e; CameraIntrinsicsMat k;
      if (!pose_graph->getExtrinsicsForCamera(obs.camera_id_, e)) { std::cerr << "In using factor for pending obj " << obj_info.first << " could not find extrinsics for camera " << obs.camera_id_ << "; not adding to pose graph" << std::endl; continue; }
      if (!pose_graph->getIntrinsicsForCamera(obs.camera_id_, k)) { std::cerr << "In using factor for pending obj " << obj_info.first << " could not find intrinsics for camera " << obs.camera_id_ << "; not adding to pose graph" << std::endl; continue; }
      const auto rp = robot_pose_estimates.find(obs.frame_id_);
      if (rp == robot_pose_estimates.end()) { std::cerr << "pending obj " << obj_info.first << ": no pose for frame " << obs.frame_id_ << std::endl; continue; }
      if (!cam_index.count(obs.camera_id_)) {
        cam_index[obs.camera_id_] = (uint16_t)(cam_K.size() / 4);
        cam_K.insert(cam_K.end(), {k.fx, k.fy, k.cx, k.cy});
        const double th = std::sqrt(e.orientation_[0] * e.orientation_[0] + e.orientation_[1] * e.orientation_[1] + e.orientation_[2] * e.orientation_[2]);
        const double s = th > 0 ? std::sin(th / 2) / th : 0.5;
        cam_ext.insert(cam_ext.end(), {e.orientation_[0] * s, e.orientation_[1] * s, e.orientation_[2] * s, std::cos(th / 2), e.transl_[0], e.transl_[1], e.transl_[2]});
      }
      if (!pose_index.count(obs.frame_id_)) { pose_index[obs.frame_id_] = (uint32_t)(poses.size() / 6); poses.insert(poses.end(), rp->second.begin(), rp->second.end()); }
      bb_obj.push_back(oi->second); bb_pose.push_back(pose_index[obs.frame_id_]); bb_cam.push_back(cam_index[obs.camera_id_]);
      bb_corners.insert(bb_corners.end(), obs.bounding_box_corners_.begin(), obs.bounding_box_corners_.end());
      bb_cov.insert(bb_cov.end(), obs.bounding_box_corners_covariance_.begin(), obs.bounding_box_corners_covariance_.end());
    }
    const auto cls = mean_and_cov_by_semantic_class.find(obj_info.second.semantic_class_);
    if (cls == mean_and_cov_by_semantic_class.end()) { std::cerr << "no shape prior for semantic class " << obj_info.second.semantic_class_ << std::endl; return false; }   // .at() throws in the reference
    sp_obj.push_back(oi->second);
    sp_mean.insert(sp_mean.end(), cls->second.first.begin(), cls->second.first.end());
    sp_cov.insert(sp_cov.end(), cls->second.second.begin(), cls->second.second.end());
  }
  if (updated_estimates) *updated_estimates = rough_initial_estimates;
  if (objects.empty() || cam_K.empty()) return true;                       // nothing to refine

  const obvi_ba_options opt = obvi::makeHandleOptions(device_id);
  obvi_ba_handle* h = obvi::HandlePool::instance().acquire(opt);
  if (h == nullptr) return false;
  const std::vector<uint8_t> pose_const(poses.size() / 6, 1), obj_const(objects.size() / 7, 0);       // problem.SetParameterBlockConstant(robot_pose_block), :99-103
  const auto& rp = estimator_params.object_residual_params_;
  int rc = obvi_ba_set_cameras(h, (int32_t)(cam_K.size() / 4), cam_K.data(), cam_ext.data());
  if (!rc) rc = obvi_ba_set_poses(h, (int64_t)pose_const.size(), poses.data(), pose_const.data());
  if (!rc) rc = obvi_ba_set_points(h, 0, nullptr, nullptr);
  if (!rc) rc = obvi_ba_set_objects(h, (int64_t)obj_const.size(), objects.data(), obj_const.data());
  if (!rc) rc = obvi_ba_set_bbox(h, (int64_t)bb_obj.size(), bb_obj.data(), bb_pose.data(), bb_cam.data(), bb_corners.data(), bb_cov.data(),
                                 rp.object_observation_huber_loss_param_, rp.invalid_ellipsoid_error_val_);
  if (!rc) rc = obvi_ba_set_shape_priors(h, (int64_t)sp_obj.size(), sp_obj.data(), sp_mean.data(), sp_cov.data(), rp.shape_dim_prior_factor_huber_loss_param_);
  const auto& sp = estimator_params.solver_params_;
  // the reference sets the five options below and leaves the trust-region radii at Ceres' defaults (:105-118)
  obvi_solver_params p{sp.max_num_iterations_, sp.allow_non_monotonic_steps_ ? 1 : 0, sp.function_tolerance_, sp.gradient_tolerance_, sp.parameter_tolerance_, 1e4, 1e16};
  obvi_summary s{};
  if (!rc) rc = obvi_ba_solve(h, &p, &s);
  if (!rc) rc = obvi_ba_get_objects(h, objects.data());
  if (rc) std::cerr << "pending object estimation: " << obvi_ba_last_error(h) << std::endl;
  if (!rc) {   // pending_object_estimator.cpp:134-143: the iteration rows of this solve
    const std::shared_ptr<IterationLogger> logger = IterationLoggerFactory::getInstance().getOrCreateLoggerOfType(IterationLoggerFactory::kPendingEstimatorOptimizationType);
    if (logger != nullptr) {
      obvi::SolverSummary summary;
      summary.num_parameters_reduced = s.num_parameters_reduced;
      std::vector<obvi_iteration_summary> its((size_t)std::max(s.num_iterations, 1));
      const int nit = obvi_ba_get_iterations(h, its.data(), (int32_t)its.size());
      for (int i = 0; i < nit; ++i) summary.iterations.push_back({its[i].iteration, its[i].cost, its[i].cost_change, its[i].step_norm, its[i].gradient_max_norm, its[i].step_is_successful != 0});
      logger->logIterations(optimization_identifier, summary);
    }
  }
  obvi::HandlePool::instance().release(opt, h);
  if (rc) return false;
  if (summary_out) *summary_out = s;
  if (s.termination_type == OBVI_FAILURE) { std::cerr << "Ceres optimization failed for pending  object estimation" << std::endl; return false; }
  if (updated_estimates) for (const auto& o : obj_index) std::copy_n(&objects[7 * o.second], 7, (*updated_estimates)[o.first].begin());
  return true;
}

}  // namespace vslam_types_refactor
#endif  // OBVI_HOST_PENDING_OBJECT_ESTIMATOR_H_
