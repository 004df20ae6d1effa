// obvi_pose_graph.h -- factor-graph store with the reference's accessors.
// Mirrors ObjectAndReprojectionFeaturePoseGraph (include/refactoring/optimization/object_pose_graph.h:276-1230)
// and its base ReprojectionLowLevelFeaturePoseGraph (low_level_feature_pose_graph.h:245-762): parameter
// blocks are individually heap-allocated and handed out as raw double* (getPosePointers :315-321,
// getFeaturePointers :692-700, getObjectParamPointers object_pose_graph.h:495-503); the optimiser
// mutates them in place.
#ifndef OBVI_HOST_POSE_GRAPH_H_
#define OBVI_HOST_POSE_GRAPH_H_

#include <algorithm>
#include <map>
#include <optional>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <utility>

#include "obvi_types.h"

namespace vslam_types_refactor {

typedef std::pair<FactorType, FeatureFactorId> FactorInfo;
struct FactorInfoHash { size_t operator()(const FactorInfo& f) const { return std::hash<uint64_t>()((uint64_t)f.first << 56 ^ f.second); } };
typedef std::unordered_set<FactorInfo, FactorInfoHash> FactorInfoSet;   // util::BoostHashSet<pair<FactorType, FeatureFactorId>>

class ObjectAndReprojectionFeaturePoseGraph {
 public:
  ObjectAndReprojectionFeaturePoseGraph(const std::unordered_map<CameraId, CameraExtrinsics>& extrinsics,
                                        const std::unordered_map<CameraId, CameraIntrinsicsMat>& intrinsics)
      : camera_extrinsics_by_camera_(extrinsics), camera_intrinsics_by_camera_(intrinsics) {}

  // ---- frames -----------------------------------------------------------------------------
  void addFrame(const FrameId& frame_id, const Pose3D& initial_pose_estimate) {          // low_level...h:267-277
    robot_poses_[frame_id] = std::make_shared<RawPose3d>(convertPoseToArray(initial_pose_estimate));
    max_frame_id_ = robot_poses_.size() == 1 ? frame_id : std::max(max_frame_id_, frame_id);
    min_frame_id_ = robot_poses_.size() == 1 ? frame_id : std::min(min_frame_id_, frame_id);
  }
  std::unordered_set<FrameId> getFrameIds() const { std::unordered_set<FrameId> s; for (const auto& p : robot_poses_) s.insert(p.first); return s; }
  bool getPosePointers(const FrameId& frame_id, double** pose_ptr) {
    auto it = robot_poses_.find(frame_id);
    if (it == robot_poses_.end()) return false;
    *pose_ptr = it->second->data();
    return true;
  }
  std::optional<RawPose3d> getRobotPose(const FrameId& frame_id) const {
    auto it = robot_poses_.find(frame_id);
    if (it == robot_poses_.end()) return std::nullopt;
    return *it->second;
  }
  void getRobotPoseEstimates(std::unordered_map<FrameId, RawPose3d>& out) const { out.clear(); for (const auto& p : robot_poses_) out[p.first] = *p.second; }
  FrameId getMaxFrameId() const { return max_frame_id_; }

  // ---- cameras ----------------------------------------------------------------------------
  bool getExtrinsicsForCamera(const CameraId& c, CameraExtrinsics& e) const { auto it = camera_extrinsics_by_camera_.find(c); if (it == camera_extrinsics_by_camera_.end()) return false; e = it->second; return true; }
  bool getIntrinsicsForCamera(const CameraId& c, CameraIntrinsicsMat& k) const { auto it = camera_intrinsics_by_camera_.find(c); if (it == camera_intrinsics_by_camera_.end()) return false; k = it->second; return true; }
  const std::unordered_map<CameraId, CameraExtrinsics>& extrinsics() const { return camera_extrinsics_by_camera_; }
  const std::unordered_map<CameraId, CameraIntrinsicsMat>& intrinsics() const { return camera_intrinsics_by_camera_; }

  // ---- visual features ---------------------------------------------------------------------
  void addFeature(const FeatureId& feature_id, const Position3d& position) { feature_positions_[feature_id] = std::make_shared<Position3d>(position); }
  bool hasFeature(const FeatureId& id) const { return feature_positions_.count(id) != 0; }
  FeatureFactorId addVisualFactor(const ReprojectionErrorFactor& factor) {               // low_level...h:341-369
    const FeatureFactorId id = next_visual_factor_id_++;
    factors_[id] = factor;
    visual_factors_by_frame_[factor.frame_id_].push_back(id);
    visual_factors_by_feature_[factor.feature_id_].push_back(id);
    auto it = first_observed_frame_by_feature_.find(factor.feature_id_);
    if (it == first_observed_frame_by_feature_.end() || factor.frame_id_ < it->second) first_observed_frame_by_feature_[factor.feature_id_] = factor.frame_id_;
    return id;
  }
  bool getVisualFactor(const FeatureFactorId& id, ReprojectionErrorFactor& f) const { auto it = factors_.find(id); if (it == factors_.end()) return false; f = it->second; return true; }
  bool getFeaturePointers(const FeatureId& id, double** ptr) { auto it = feature_positions_.find(id); if (it == feature_positions_.end()) return false; *ptr = it->second->data(); return true; }
  void getVisualFeatureFactorIdsBetweenFrameIdsInclusive(const FrameId& min_f, const FrameId& max_f, FactorInfoSet& out) const {   // :390-415
    for (const auto& fr : visual_factors_by_frame_)
      if (fr.first >= min_f && fr.first <= max_f) for (FeatureFactorId id : fr.second) out.insert({kReprojectionErrorFactorTypeId, id});
  }
  // the factors of getVisualFeatureFactorIdsBetweenFrameIdsInclusive handed to `fn(id, factor)` one by one (no id set is built)
  template <class F>
  void forEachVisualFactorBetweenFrameIdsInclusive(const FrameId& min_f, const FrameId& max_f, F&& fn) const {
    if (max_f - min_f < visual_factors_by_frame_.size()) {       // frames in ascending order (ids are then usually ascending as well)
      for (FrameId f = min_f; f <= max_f; ++f) {
        const auto fr = visual_factors_by_frame_.find(f);
        if (fr == visual_factors_by_frame_.end()) continue;
        for (FeatureFactorId id : fr->second) { const auto it = factors_.find(id); if (it != factors_.end()) fn(id, it->second); }
      }
      return;
    }
    for (const auto& fr : visual_factors_by_frame_)
      if (fr.first >= min_f && fr.first <= max_f)
        for (FeatureFactorId id : fr.second) { const auto it = factors_.find(id); if (it != factors_.end()) fn(id, it->second); }
  }
  bool getFeatureIdForObservationFactor(const FactorInfo& info, FeatureId& feature_id) const {
    if (info.first != kReprojectionErrorFactorTypeId) return false;
    auto it = factors_.find(info.second); if (it == factors_.end()) return false; feature_id = it->second.feature_id_; return true;
  }
  void getVisualFeatureEstimates(std::unordered_map<FeatureId, Position3d>& out) const { out.clear(); for (const auto& p : feature_positions_) out[p.first] = *p.second; }
  bool getFirstObservedFrameForFeature(const FeatureId& id, FrameId& frame) const { auto it = first_observed_frame_by_feature_.find(id); if (it == first_observed_frame_by_feature_.end()) return false; frame = it->second; return true; }
  void updateVisualPositionParams(const FeatureId& id, const Position3d& p) { *feature_positions_.at(id) = p; }
  const std::unordered_map<FeatureId, Position3dPtr>& featurePositions() const { return feature_positions_; }

  // ---- relative pose (odometry) factors -----------------------------------------------------
  FeatureFactorId addPoseFactor(const RelPoseFactor& f) {                                 // low_level...h:371-381
    const FeatureFactorId id = next_pose_factor_id_++;
    pose_factors_[id] = f;
    pose_factors_by_frame_[f.frame_id_1_].push_back(id);
    pose_factors_by_frame_[f.frame_id_2_].push_back(id);
    return id;
  }
  bool getPoseFactor(const FeatureFactorId& id, RelPoseFactor& f) const { auto it = pose_factors_.find(id); if (it == pose_factors_.end()) return false; f = it->second; return true; }
  // factors touching `frame_id` whose two frames both lie in [min, max]  (low_level...h:436-466)
  void getPoseFactorInfoByFrameId(const FrameId& frame_id, const FrameId& min_f, const FrameId& max_f, FactorInfoSet& out) const {
    auto it = pose_factors_by_frame_.find(frame_id);
    if (it == pose_factors_by_frame_.end()) return;
    for (FeatureFactorId id : it->second) {
      const RelPoseFactor& f = pose_factors_.at(id);
      if (f.frame_id_1_ >= min_f && f.frame_id_1_ <= max_f && f.frame_id_2_ >= min_f && f.frame_id_2_ <= max_f) out.insert({kPairwiseRobotPoseFactorTypeId, id});
    }
  }

  // ---- objects ----------------------------------------------------------------------------
  ObjectId addNewEllipsoid(const RawEllipsoid& estimate, const std::string& semantic_class) {   // object_pose_graph.h:397-415
    const ObjectId id = next_object_id_++;
    ellipsoid_estimates_[id] = std::make_shared<RawEllipsoid>(estimate);
    semantic_class_for_object_[id] = semantic_class;
    return id;
  }
  void addLongTermMapObject(const ObjectId& id, const RawEllipsoid& estimate, const std::string& semantic_class, const LongTermMapObjectPrior& prior) {
    ellipsoid_estimates_[id] = std::make_shared<RawEllipsoid>(estimate);
    semantic_class_for_object_[id] = semantic_class;
    next_object_id_ = std::max(next_object_id_, id + 1);
    long_term_map_object_ids_.insert(id);
    ltm_factors_[next_ltm_factor_id_] = prior;
    ltm_factor_for_object_[id] = next_ltm_factor_id_++;
  }
  bool getObjectParamPointers(const ObjectId& id, double** ptr) { auto it = ellipsoid_estimates_.find(id); if (it == ellipsoid_estimates_.end()) return false; *ptr = it->second->data(); return true; }
  void getObjectEstimates(std::unordered_map<ObjectId, RawEllipsoid>& out) const { out.clear(); for (const auto& p : ellipsoid_estimates_) out[p.first] = *p.second; }
  void getLongTermMapObjects(std::unordered_set<ObjectId>& out) const { out = long_term_map_object_ids_; }
  FeatureFactorId addObjectObservation(const ObjectObservationFactor& f) {               // object_pose_graph.h:431-447
    const FeatureFactorId id = next_obj_factor_id_++;
    object_observation_factors_[id] = f;
    observation_factors_by_frame_[f.frame_id_].push_back(id);
    observation_factors_by_object_[f.object_id_].push_back(id);
    return id;
  }
  FeatureFactorId addShapeDimPrior(const ShapeDimPriorFactor& f) {                        // object_pose_graph.h:449-461
    const FeatureFactorId id = next_obj_factor_id_++;
    shape_dim_prior_factors_[id] = f;
    shape_dim_factor_for_object_[f.object_id_] = id;
    return id;
  }
  bool getObjectObservationFactor(const FeatureFactorId& id, ObjectObservationFactor& f) const { auto it = object_observation_factors_.find(id); if (it == object_observation_factors_.end()) return false; f = it->second; return true; }
  bool getShapeDimPriorFactor(const FeatureFactorId& id, ShapeDimPriorFactor& f) const { auto it = shape_dim_prior_factors_.find(id); if (it == shape_dim_prior_factors_.end()) return false; f = it->second; return true; }
  bool getLongTermMapFactor(const FeatureFactorId& id, LongTermMapObjectPrior& f) const { auto it = ltm_factors_.find(id); if (it == ltm_factors_.end()) return false; f = it->second; return true; }
  void getObservationFactorsBetweenFrameIdsInclusive(const FrameId& min_f, const FrameId& max_f, FactorInfoSet& out) const {        // :553-577
    for (const auto& fr : observation_factors_by_frame_)
      if (fr.first >= min_f && fr.first <= max_f) for (FeatureFactorId id : fr.second) out.insert({kObjectObservationFactorTypeId, id});
  }
  bool getObjectIdForObjObservationFactor(const FactorInfo& info, ObjectId& object_id) const {
    if (info.first != kObjectObservationFactorTypeId) return false;
    auto it = object_observation_factors_.find(info.second); if (it == object_observation_factors_.end()) return false; object_id = it->second.object_id_; return true;
  }
  // shape priors (and, unless the LTM objects are fixed, LTM priors) of the given objects  (object_pose_graph.h:627-690)
  void getOnlyObjectFactorsForObjects(const std::unordered_set<ObjectId>& objects, const bool& use_pom, const bool& include_ltm_factors,
                                      std::unordered_map<ObjectId, FactorInfoSet>& out) const {
    (void)use_pom;   // pairwise object map factors are an empty stub in the reference (pairwise_object_map_factor.h:18-24)
    for (const ObjectId& o : objects) {
      auto s = shape_dim_factor_for_object_.find(o);
      if (s != shape_dim_factor_for_object_.end()) out[o].insert({kShapeDimPriorFactorTypeId, s->second});
      if (include_ltm_factors) { auto l = ltm_factor_for_object_.find(o); if (l != ltm_factor_for_object_.end()) out[o].insert({kLongTermMapFactorTypeId, l->second}); }
    }
  }

  // ---- value copy / restore (object_pose_graph.h:1025-1121) ---------------------------------
  std::shared_ptr<ObjectAndReprojectionFeaturePoseGraph> makeCopyDeepCopyValues() const {
    auto c = std::make_shared<ObjectAndReprojectionFeaturePoseGraph>(*this);
    for (auto& p : c->robot_poses_) p.second = std::make_shared<RawPose3d>(*p.second);
    for (auto& p : c->feature_positions_) p.second = std::make_shared<Position3d>(*p.second);
    for (auto& p : c->ellipsoid_estimates_) p.second = std::make_shared<RawEllipsoid>(*p.second);
    return c;
  }
  void setValuesFromAnotherPoseGraph(const std::shared_ptr<ObjectAndReprojectionFeaturePoseGraph>& other) {
    for (auto& p : robot_poses_) { auto it = other->robot_poses_.find(p.first); if (it != other->robot_poses_.end()) *p.second = *it->second; }
    for (auto& p : feature_positions_) { auto it = other->feature_positions_.find(p.first); if (it != other->feature_positions_.end()) *p.second = *it->second; }
    for (auto& p : ellipsoid_estimates_) { auto it = other->ellipsoid_estimates_.find(p.first); if (it != other->ellipsoid_estimates_.end()) *p.second = *it->second; }
  }

 private:
  std::unordered_map<CameraId, CameraExtrinsics> camera_extrinsics_by_camera_;
  std::unordered_map<CameraId, CameraIntrinsicsMat> camera_intrinsics_by_camera_;
  std::unordered_map<FrameId, RawPose3dPtr> robot_poses_;
  FrameId min_frame_id_ = 0, max_frame_id_ = 0;
  std::unordered_map<FeatureId, Position3dPtr> feature_positions_;
  std::unordered_map<FeatureFactorId, ReprojectionErrorFactor> factors_;
  std::unordered_map<FrameId, std::vector<FeatureFactorId>> visual_factors_by_frame_;
  std::unordered_map<FeatureId, std::vector<FeatureFactorId>> visual_factors_by_feature_;
  std::unordered_map<FeatureId, FrameId> first_observed_frame_by_feature_;
  std::unordered_map<FeatureFactorId, RelPoseFactor> pose_factors_;
  std::unordered_map<FrameId, std::vector<FeatureFactorId>> pose_factors_by_frame_;
  std::unordered_map<ObjectId, RawEllipsoidPtr> ellipsoid_estimates_;
  std::unordered_map<ObjectId, std::string> semantic_class_for_object_;
  std::unordered_set<ObjectId> long_term_map_object_ids_;
  std::unordered_map<FeatureFactorId, ObjectObservationFactor> object_observation_factors_;
  std::unordered_map<FeatureFactorId, ShapeDimPriorFactor> shape_dim_prior_factors_;
  std::unordered_map<FeatureFactorId, LongTermMapObjectPrior> ltm_factors_;
  std::unordered_map<ObjectId, FeatureFactorId> shape_dim_factor_for_object_, ltm_factor_for_object_;
  std::unordered_map<FrameId, std::vector<FeatureFactorId>> observation_factors_by_frame_;
  std::unordered_map<ObjectId, std::vector<FeatureFactorId>> observation_factors_by_object_;
  FeatureFactorId next_visual_factor_id_ = 0, next_pose_factor_id_ = 0, next_obj_factor_id_ = 0, next_ltm_factor_id_ = 0;
  ObjectId next_object_id_ = 0;
};

}  // namespace vslam_types_refactor
#endif  // OBVI_HOST_POSE_GRAPH_H_
