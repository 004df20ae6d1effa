// obvi_pose_graph.h -- factor-graph store with the reference's accessors.
// Mirrors ObjectAndReprojectionFeaturePoseGraph (include/refactoring/optimization/object_pose_graph.h:276-1230)
// and its base ReprojectionLowLevelFeaturePoseGraph (low_level_feature_pose_graph.h:245-762): parameter
// blocks are individually heap-allocated and handed out as raw double* (getPosePointers :315-321,
// getFeaturePointers :692-700, getObjectParamPointers object_pose_graph.h:495-503); the optimiser
// mutates them in place.
#ifndef OBVI_HOST_POSE_GRAPH_H_
#define OBVI_HOST_POSE_GRAPH_H_

#include <limits>
#include <algorithm>
#include <iostream>
#include <map>
#include <memory>
#include <optional>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "obvi_types.h"

namespace vslam_types_refactor {

typedef std::pair<FactorType, FeatureFactorId> FactorInfo;
struct FactorInfoHash { size_t operator()(const FactorInfo& f) const { return std::hash<uint64_t>()((uint64_t)f.first << 56 ^ f.second); } };
typedef std::unordered_set<FactorInfo, FactorInfoHash> FactorInfoSet;   // util::BoostHashSet<pair<FactorType, FeatureFactorId>>

// ---- state structs: the reference's (low_level_feature_pose_graph.h:25-65, object_pose_graph.h:22-86), same member names ----------
struct LowLevelFeaturePoseGraphState {
  std::unordered_map<CameraId, CameraExtrinsics> camera_extrinsics_by_camera_;
  std::unordered_map<CameraId, CameraIntrinsicsMat> camera_intrinsics_by_camera_;
  FactorType visual_factor_type_ = kReprojectionErrorFactorTypeId;
  FrameId min_frame_id_ = 0, max_frame_id_ = 0;
  FeatureFactorId max_feature_factor_id_ = 0, max_pose_factor_id_ = 0;
  std::unordered_map<FrameId, RawPose3d> robot_poses_;
  std::unordered_map<FrameId, FactorInfoSet> pose_factors_by_frame_;
  std::unordered_map<FrameId, std::vector<FactorInfo>> visual_feature_factors_by_frame_;
  std::unordered_map<FeatureId, FactorInfoSet> visual_factors_by_feature_;
  std::unordered_map<FeatureFactorId, RelPoseFactor> pose_factors_;
  std::unordered_map<FeatureFactorId, ReprojectionErrorFactor> factors_;
  std::unordered_map<FeatureId, FrameId> last_observed_frame_by_feature_, first_observed_frame_by_feature_;
};
struct ReprojectionLowLevelFeaturePoseGraphState {
  LowLevelFeaturePoseGraphState low_level_pg_state_;
  FeatureId min_feature_id_ = 0, max_feature_id_ = 0;
  std::unordered_map<FeatureId, Position3d> feature_positions_;
};
struct ObjOnlyPoseGraphState {
  std::unordered_map<std::string, std::pair<ObjectDim, Covariance<3>>> mean_and_cov_by_semantic_class_;
  ObjectId min_object_id_ = 0, max_object_id_ = 0;
  std::unordered_map<ObjectId, RawEllipsoid> ellipsoid_estimates_;
  std::unordered_map<ObjectId, std::string> semantic_class_for_object_;
  std::unordered_map<ObjectId, FrameId> last_observed_frame_by_object_, first_observed_frame_by_object_;
  FeatureFactorId min_object_observation_factor_ = 0, max_object_observation_factor_ = 0, min_obj_specific_factor_ = 0, max_obj_specific_factor_ = 0;
  std::unordered_set<ObjectId> long_term_map_object_ids_;
  std::unordered_map<FeatureFactorId, ObjectObservationFactor> object_observation_factors_;
  std::unordered_map<FeatureFactorId, ShapeDimPriorFactor> shape_dim_prior_factors_;
  std::unordered_map<FrameId, FactorInfoSet> observation_factors_by_frame_;
  std::unordered_map<ObjectId, FactorInfoSet> observation_factors_by_object_, object_only_factors_by_object_;
};
struct ObjectAndReprojectionFeaturePoseGraphState {
  ReprojectionLowLevelFeaturePoseGraphState reprojection_low_level_feature_pose_graph_state_;
  ObjOnlyPoseGraphState obj_only_pose_graph_state_;
};

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
  void addFeature(const FeatureId& feature_id, const Position3d& position) {
    Position3dPtr& p = feature_positions_[feature_id];
    p = std::make_shared<Position3d>(position);
    slot_position_[featureSlot_(feature_id)] = p->data();
  }
  bool hasFeature(const FeatureId& id) const { return feature_positions_.count(id) != 0; }
  FeatureFactorId addVisualFactor(const ReprojectionErrorFactor& factor) {               // low_level...h:341-369
    const FeatureFactorId id = next_visual_factor_id_++;
    factors_[id] = factor;
    const uint32_t slot = featureSlot_(factor.feature_id_);
    if (factor.frame_id_ < slot_first_frame_[slot]) slot_first_frame_[slot] = factor.frame_id_;
    visual_records_by_frame_[factor.frame_id_].push_back({id, factor.feature_id_, factor.frame_id_, factor.camera_id_, slot,
                                                          factor.feature_pos_[0], factor.feature_pos_[1], factor.reprojection_error_std_dev_});
    visual_factors_by_frame_[factor.frame_id_].push_back(id);
    visual_factors_by_feature_[factor.feature_id_].push_back(id);
    auto it = first_observed_frame_by_feature_.find(factor.feature_id_);
    if (it == first_observed_frame_by_feature_.end() || factor.frame_id_ < it->second) first_observed_frame_by_feature_[factor.feature_id_] = factor.frame_id_;
    return id;
  }
  // the reprojection factors of a feature (getFactorsForFeature, low_level_feature_pose_graph.h, restricted to the visual ones)
  const std::vector<FeatureFactorId>& visualFactorIdsOfFeature(const FeatureId& id) const { static const std::vector<FeatureFactorId> none; auto it = visual_factors_by_feature_.find(id); return it == visual_factors_by_feature_.end() ? none : it->second; }
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
  // The same walk over a flat copy of the factors (one record per factor, per frame in the order of visual_factors_by_frame_; kept by
  // addVisualFactor): a sliding window looks at every visual factor of its frames twice per frame of the trajectory, and a hash lookup
  // per factor is most of what building the window's problem costs.  A feature's slot is a dense index (featureIdOfSlot,
  // featurePointerOfSlot, numFeatureSlots) for per-feature scratch arrays in place of maps keyed by the feature id.
  struct VisualFactorRecord { FeatureFactorId id; FeatureId feature_id; FrameId frame_id; CameraId camera_id; uint32_t feature_slot; double px, py, sigma; };
  template <class F>
  void forEachVisualRecordBetweenFrameIdsInclusive(const FrameId& min_f, const FrameId& max_f, F&& fn) const {
    if (max_f - min_f < visual_records_by_frame_.size()) {
      for (FrameId f = min_f; f <= max_f; ++f) {
        const auto fr = visual_records_by_frame_.find(f);
        if (fr != visual_records_by_frame_.end()) for (const VisualFactorRecord& r : fr->second) fn(r);
      }
      return;
    }
    for (const auto& fr : visual_records_by_frame_)
      if (fr.first >= min_f && fr.first <= max_f) for (const VisualFactorRecord& r : fr.second) fn(r);
  }
  // ... a frame's records at once: fn(first record, count)
  template <class F>
  void forEachVisualRecordSpanBetweenFrameIdsInclusive(const FrameId& min_f, const FrameId& max_f, F&& fn) const {
    if (max_f - min_f < visual_records_by_frame_.size()) {
      for (FrameId f = min_f; f <= max_f; ++f) {
        const auto fr = visual_records_by_frame_.find(f);
        if (fr != visual_records_by_frame_.end() && !fr->second.empty()) fn(fr->second.data(), fr->second.size());
      }
      return;
    }
    for (const auto& fr : visual_records_by_frame_)
      if (fr.first >= min_f && fr.first <= max_f && !fr.second.empty()) fn(fr.second.data(), fr.second.size());
  }
  size_t numFeatureSlots() const { return slot_feature_.size(); }
  FeatureId featureIdOfSlot(uint32_t slot) const { return slot_feature_[slot]; }
  double* featurePointerOfSlot(uint32_t slot) const { return slot_position_[slot]; }   // nullptr: factors seen, feature not added (yet)
  // getFirstObservedFrameForFeature by slot, without the hash lookup (kNoFrame: no factor seen)
  static constexpr FrameId kNoFrame = std::numeric_limits<FrameId>::max();
  FrameId firstObservedFrameOfSlot(uint32_t slot) const { return slot_first_frame_[slot]; }
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

  void getObjectEstimates(std::unordered_map<ObjectId, std::pair<std::string, RawEllipsoid>>& out) const {                // object_pose_graph.h:505-520
    out.clear();
    for (const auto& p : ellipsoid_estimates_) { auto c = semantic_class_for_object_.find(p.first); out[p.first] = {c == semantic_class_for_object_.end() ? std::string() : c->second, *p.second}; }
  }
  // mergeObjects (object_pose_graph.h:739-830): the observation factors of the merged objects are re-pointed at the object they are
  // merged into; the merged objects, their estimates, classes and object-only factors (shape priors) are removed.
  bool mergeObjects(const std::unordered_map<ObjectId, std::unordered_set<ObjectId>>& objects_to_merge) {
    std::unordered_set<ObjectId> objects_to_remove;
    for (const auto& merge_group : objects_to_merge) {
      for (const ObjectId& merge_obj : merge_group.second) {
        if (objects_to_remove.count(merge_obj)) std::cerr << "Object " << merge_obj << " expected to be merged into multiple objects (second was " << merge_group.first << "). This will cause unexpected behavior." << std::endl;
        if (objects_to_merge.count(merge_obj)) std::cerr << "Object " << merge_obj << " is supposed to be merged into " << merge_group.first << " and have other objects merged into it. This will cause unexpected behavior." << std::endl;
        const auto src = observation_factors_by_object_.find(merge_obj);
        if (src == observation_factors_by_object_.end()) continue;
        const std::vector<FeatureFactorId> moved = src->second;
        for (FeatureFactorId id : moved) {
          auto f = object_observation_factors_.find(id);
          if (f == object_observation_factors_.end()) { std::cerr << "Could not find object observation factor with id " << id << " skipping" << std::endl; continue; }
          f->second.object_id_ = merge_group.first;
          observation_factors_by_object_[merge_group.first].push_back(id);
        }
      }
      objects_to_remove.insert(merge_group.second.begin(), merge_group.second.end());
      auto& v = observation_factors_by_object_[merge_group.first];
      std::sort(v.begin(), v.end());
    }
    for (const ObjectId& o : objects_to_remove) {
      ellipsoid_estimates_.erase(o); semantic_class_for_object_.erase(o); long_term_map_object_ids_.erase(o);
      auto sp = shape_dim_factor_for_object_.find(o);
      if (sp != shape_dim_factor_for_object_.end()) { shape_dim_prior_factors_.erase(sp->second); shape_dim_factor_for_object_.erase(sp); }
      auto lt = ltm_factor_for_object_.find(o);
      if (lt != ltm_factor_for_object_.end()) { ltm_factors_.erase(lt->second); ltm_factor_for_object_.erase(lt); }
      observation_factors_by_object_.erase(o);
    }
    return true;
  }

  // ---- state (checkpoints): getState / createObjectAndReprojectionFeaturePoseGraphFromState (object_pose_graph.h:1185-1212) -------
  void setShapePriorsBySemanticClass(const std::unordered_map<std::string, std::pair<ObjectDim, Covariance<3>>>& m) { mean_and_cov_by_semantic_class_ = m; }
  const std::unordered_map<std::string, std::pair<ObjectDim, Covariance<3>>>& shapePriorsBySemanticClass() const { return mean_and_cov_by_semantic_class_; }
  void getState(ObjectAndReprojectionFeaturePoseGraphState& pose_graph_state) const {
    pose_graph_state = ObjectAndReprojectionFeaturePoseGraphState();
    ReprojectionLowLevelFeaturePoseGraphState& R = pose_graph_state.reprojection_low_level_feature_pose_graph_state_;
    LowLevelFeaturePoseGraphState& L = R.low_level_pg_state_;
    ObjOnlyPoseGraphState& O = pose_graph_state.obj_only_pose_graph_state_;
    L.camera_extrinsics_by_camera_ = camera_extrinsics_by_camera_; L.camera_intrinsics_by_camera_ = camera_intrinsics_by_camera_;
    L.visual_factor_type_ = kReprojectionErrorFactorTypeId;
    L.min_frame_id_ = min_frame_id_; L.max_frame_id_ = max_frame_id_;
    // The reference stores the LAST id it issued and hands out max + 1 next (low_level_feature_pose_graph.h:452-453, 485-486, 604-605);
    // the counters here hold the NEXT id, so the state gets next - 1 (0 while nothing was issued, the reference's initial value :258-259)
    L.max_feature_factor_id_ = next_visual_factor_id_ ? next_visual_factor_id_ - 1 : 0; L.max_pose_factor_id_ = next_pose_factor_id_ ? next_pose_factor_id_ - 1 : 0;
    for (const auto& p : robot_poses_) L.robot_poses_[p.first] = *p.second;
    for (const auto& f : pose_factors_by_frame_) for (FeatureFactorId id : f.second) L.pose_factors_by_frame_[f.first].insert({kPairwiseRobotPoseFactorTypeId, id});
    for (const auto& f : visual_factors_by_frame_) for (FeatureFactorId id : f.second) L.visual_feature_factors_by_frame_[f.first].push_back({kReprojectionErrorFactorTypeId, id});
    for (const auto& f : visual_factors_by_feature_) for (FeatureFactorId id : f.second) L.visual_factors_by_feature_[f.first].insert({kReprojectionErrorFactorTypeId, id});
    L.pose_factors_ = pose_factors_; L.factors_ = factors_;
    L.first_observed_frame_by_feature_ = first_observed_frame_by_feature_;
    for (const auto& f : factors_) { auto it = L.last_observed_frame_by_feature_.find(f.second.feature_id_); if (it == L.last_observed_frame_by_feature_.end() || it->second < f.second.frame_id_) L.last_observed_frame_by_feature_[f.second.feature_id_] = f.second.frame_id_; }
    bool first = true;
    for (const auto& p : feature_positions_) {
      R.feature_positions_[p.first] = *p.second;
      R.min_feature_id_ = first ? p.first : std::min(R.min_feature_id_, p.first); R.max_feature_id_ = first ? p.first : std::max(R.max_feature_id_, p.first); first = false;
    }
    O.mean_and_cov_by_semantic_class_ = mean_and_cov_by_semantic_class_;
    // object ids: smallest / largest id in the graph (object_pose_graph.h:831-838), never below the last id issued (:356-357)
    O.min_object_id_ = 0; O.max_object_id_ = next_object_id_ ? next_object_id_ - 1 : 0;
    { bool any = false; for (const auto& p : ellipsoid_estimates_) { O.min_object_id_ = any ? std::min(O.min_object_id_, p.first) : p.first; O.max_object_id_ = std::max(O.max_object_id_, p.first); any = true; } }
    for (const auto& p : ellipsoid_estimates_) O.ellipsoid_estimates_[p.first] = *p.second;
    O.semantic_class_for_object_ = semantic_class_for_object_;
    O.long_term_map_object_ids_ = long_term_map_object_ids_;
    O.object_observation_factors_ = object_observation_factors_; O.shape_dim_prior_factors_ = shape_dim_prior_factors_;
    // the reference numbers observation factors and object-specific (shape prior) factors in two sequences (object_pose_graph.h:391-392, 419-420);
    // here they share one counter, so each kind reports the ids it actually holds: resuming either way cannot reuse an id
    O.min_object_observation_factor_ = O.max_object_observation_factor_ = O.min_obj_specific_factor_ = O.max_obj_specific_factor_ = 0;
    { bool any = false; for (const auto& f : object_observation_factors_) { O.min_object_observation_factor_ = any ? std::min(O.min_object_observation_factor_, f.first) : f.first; O.max_object_observation_factor_ = std::max(O.max_object_observation_factor_, f.first); any = true; } }
    { bool any = false; for (const auto& f : shape_dim_prior_factors_) { O.min_obj_specific_factor_ = any ? std::min(O.min_obj_specific_factor_, f.first) : f.first; O.max_obj_specific_factor_ = std::max(O.max_obj_specific_factor_, f.first); any = true; } }
    for (const auto& f : object_observation_factors_) {
      auto lo = O.first_observed_frame_by_object_.find(f.second.object_id_);
      if (lo == O.first_observed_frame_by_object_.end() || f.second.frame_id_ < lo->second) O.first_observed_frame_by_object_[f.second.object_id_] = f.second.frame_id_;
      auto hi = O.last_observed_frame_by_object_.find(f.second.object_id_);
      if (hi == O.last_observed_frame_by_object_.end() || f.second.frame_id_ > hi->second) O.last_observed_frame_by_object_[f.second.object_id_] = f.second.frame_id_;
    }
    for (const auto& f : observation_factors_by_frame_) for (FeatureFactorId id : f.second) O.observation_factors_by_frame_[f.first].insert({kObjectObservationFactorTypeId, id});
    for (const auto& f : observation_factors_by_object_) for (FeatureFactorId id : f.second) O.observation_factors_by_object_[f.first].insert({kObjectObservationFactorTypeId, id});
    for (const auto& f : shape_dim_factor_for_object_) O.object_only_factors_by_object_[f.first].insert({kShapeDimPriorFactorTypeId, f.second});
  }
  // The long-term-map priors are not part of a checkpoint (the reference re-creates them from the long-term map file through its
  // long_term_map_factor_provider): attach them afterwards with addLongTermMapObject.
  static std::shared_ptr<ObjectAndReprojectionFeaturePoseGraph> createObjectAndReprojectionFeaturePoseGraphFromState(const ObjectAndReprojectionFeaturePoseGraphState& st) {
    const ReprojectionLowLevelFeaturePoseGraphState& R = st.reprojection_low_level_feature_pose_graph_state_;
    const LowLevelFeaturePoseGraphState& L = R.low_level_pg_state_;
    const ObjOnlyPoseGraphState& O = st.obj_only_pose_graph_state_;
    auto pg = std::make_shared<ObjectAndReprojectionFeaturePoseGraph>(L.camera_extrinsics_by_camera_, L.camera_intrinsics_by_camera_);
    for (const auto& p : L.robot_poses_) pg->robot_poses_[p.first] = std::make_shared<RawPose3d>(p.second);
    pg->min_frame_id_ = L.min_frame_id_; pg->max_frame_id_ = L.max_frame_id_;
    for (const auto& p : R.feature_positions_) pg->feature_positions_[p.first] = std::make_shared<Position3d>(p.second);
    pg->factors_ = L.factors_;
    // per-frame lists keep the file's order; per-feature sets are stored in ascending id order
    for (const auto& f : L.visual_feature_factors_by_frame_) for (const FactorInfo& fi : f.second) pg->visual_factors_by_frame_[f.first].push_back(fi.second);
    for (const auto& f : L.visual_factors_by_feature_) { auto& v = pg->visual_factors_by_feature_[f.first]; for (const FactorInfo& fi : f.second) v.push_back(fi.second); std::sort(v.begin(), v.end()); }
    pg->first_observed_frame_by_feature_ = L.first_observed_frame_by_feature_;
    pg->pose_factors_ = L.pose_factors_;
    for (const auto& f : L.pose_factors_by_frame_) { auto& v = pg->pose_factors_by_frame_[f.first]; for (const FactorInfo& fi : f.second) v.push_back(fi.second); std::sort(v.begin(), v.end()); }
    // max_* is the last id issued (the reference continues with max + 1); an empty graph with max 0 has issued nothing yet
    FeatureFactorId next_v = (L.factors_.empty() && L.max_feature_factor_id_ == 0) ? 0 : L.max_feature_factor_id_ + 1;
    FeatureFactorId next_p = (L.pose_factors_.empty() && L.max_pose_factor_id_ == 0) ? 0 : L.max_pose_factor_id_ + 1;
    for (const auto& f : L.factors_) next_v = std::max(next_v, f.first + 1);
    for (const auto& f : L.pose_factors_) next_p = std::max(next_p, f.first + 1);
    pg->next_visual_factor_id_ = next_v; pg->next_pose_factor_id_ = next_p;
    pg->mean_and_cov_by_semantic_class_ = O.mean_and_cov_by_semantic_class_;
    for (const auto& p : O.ellipsoid_estimates_) pg->ellipsoid_estimates_[p.first] = std::make_shared<RawEllipsoid>(p.second);
    pg->semantic_class_for_object_ = O.semantic_class_for_object_;
    pg->long_term_map_object_ids_ = O.long_term_map_object_ids_;
    pg->object_observation_factors_ = O.object_observation_factors_; pg->shape_dim_prior_factors_ = O.shape_dim_prior_factors_;
    for (const auto& f : O.shape_dim_prior_factors_) pg->shape_dim_factor_for_object_[f.second.object_id_] = f.first;
    for (const auto& f : O.observation_factors_by_frame_) { auto& v = pg->observation_factors_by_frame_[f.first]; for (const FactorInfo& fi : f.second) v.push_back(fi.second); std::sort(v.begin(), v.end()); }
    for (const auto& f : O.observation_factors_by_object_) { auto& v = pg->observation_factors_by_object_[f.first]; for (const FactorInfo& fi : f.second) v.push_back(fi.second); std::sort(v.begin(), v.end()); }
    ObjectId next_o = (O.ellipsoid_estimates_.empty() && O.max_object_id_ == 0) ? 0 : O.max_object_id_ + 1;
    for (const auto& p : O.ellipsoid_estimates_) next_o = std::max(next_o, p.first + 1);
    pg->next_object_id_ = next_o;
    FeatureFactorId next_f = (O.object_observation_factors_.empty() && O.shape_dim_prior_factors_.empty() && O.max_object_observation_factor_ == 0 && O.max_obj_specific_factor_ == 0)
                                 ? 0 : std::max(O.max_object_observation_factor_, O.max_obj_specific_factor_) + 1;
    for (const auto& f : O.object_observation_factors_) next_f = std::max(next_f, f.first + 1);
    for (const auto& f : O.shape_dim_prior_factors_) next_f = std::max(next_f, f.first + 1);
    pg->next_obj_factor_id_ = next_f;
    pg->rebuildVisualIndex_();
    return pg;
  }

  // ---- value copy / restore (object_pose_graph.h:1025-1121) ---------------------------------
  std::shared_ptr<ObjectAndReprojectionFeaturePoseGraph> makeCopyDeepCopyValues() const {
    auto c = std::make_shared<ObjectAndReprojectionFeaturePoseGraph>(*this);
    for (auto& p : c->robot_poses_) p.second = std::make_shared<RawPose3d>(*p.second);
    for (auto& p : c->feature_positions_) p.second = std::make_shared<Position3d>(*p.second);
    for (auto& p : c->ellipsoid_estimates_) p.second = std::make_shared<RawEllipsoid>(*p.second);
    for (const auto& p : c->feature_positions_) c->slot_position_[c->feature_slot_.at(p.first)] = p.second->data();   // the copy's own values
    return c;
  }
  void setValuesFromAnotherPoseGraph(const std::shared_ptr<ObjectAndReprojectionFeaturePoseGraph>& other) {
    for (auto& p : robot_poses_) { auto it = other->robot_poses_.find(p.first); if (it != other->robot_poses_.end()) *p.second = *it->second; }
    for (auto& p : feature_positions_) { auto it = other->feature_positions_.find(p.first); if (it != other->feature_positions_.end()) *p.second = *it->second; }
    for (auto& p : ellipsoid_estimates_) { auto it = other->ellipsoid_estimates_.find(p.first); if (it != other->ellipsoid_estimates_.end()) *p.second = *it->second; }
  }

 private:
  uint32_t featureSlot_(const FeatureId& id) {
    const auto it = feature_slot_.find(id);
    if (it != feature_slot_.end()) return it->second;
    const uint32_t slot = (uint32_t)slot_feature_.size();
    feature_slot_.emplace(id, slot); slot_feature_.push_back(id); slot_position_.push_back(nullptr); slot_first_frame_.push_back(kNoFrame);
    return slot;
  }
  void rebuildVisualIndex_() {   // from factors_ / visual_factors_by_frame_ / feature_positions_ (a graph made from a state)
    feature_slot_.clear(); slot_feature_.clear(); slot_position_.clear(); slot_first_frame_.clear(); visual_records_by_frame_.clear();
    std::vector<FeatureId> ids;
    for (const auto& p : feature_positions_) ids.push_back(p.first);
    for (const auto& f : factors_) ids.push_back(f.second.feature_id_);
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    for (const FeatureId& id : ids) featureSlot_(id);
    for (const auto& p : feature_positions_) slot_position_[feature_slot_.at(p.first)] = p.second->data();
    for (const auto& p : first_observed_frame_by_feature_) { const auto it = feature_slot_.find(p.first); if (it != feature_slot_.end()) slot_first_frame_[it->second] = p.second; }
    for (const auto& fr : visual_factors_by_frame_) {
      auto& v = visual_records_by_frame_[fr.first];
      for (FeatureFactorId id : fr.second) {
        const auto it = factors_.find(id);
        if (it == factors_.end()) continue;
        const ReprojectionErrorFactor& f = it->second;
        v.push_back({id, f.feature_id_, f.frame_id_, f.camera_id_, feature_slot_.at(f.feature_id_), f.feature_pos_[0], f.feature_pos_[1], f.reprojection_error_std_dev_});
      }
    }
  }
  std::unordered_map<FrameId, std::vector<VisualFactorRecord>> visual_records_by_frame_;
  std::unordered_map<FeatureId, uint32_t> feature_slot_;
  std::vector<FeatureId> slot_feature_;
  std::vector<double*> slot_position_;
  std::vector<FrameId> slot_first_frame_;
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
  std::unordered_map<std::string, std::pair<ObjectDim, Covariance<3>>> mean_and_cov_by_semantic_class_;
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
