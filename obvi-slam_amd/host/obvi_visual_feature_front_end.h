// obvi_visual_feature_front_end.h -- host mirror of the stateful half of the reference's visual-feature front end
// (include/refactoring/visual_feature_frontend/visual_feature_front_end.h:134-800): which reprojection factors and which features
// of a new frame enter the pose graph.  The arithmetic of its two tests -- the epipolar-consistency votes of
// isReprojectionErrorFactorInlier (:511-602) and checkMinParallaxRequirements_ (:726-800) -- runs on the device behind
// include/obvi_frontend.h; this class keeps the reference's bookkeeping (the caches of pending features, what is added when:
// addVisualFeatureObservations :262-450, addFactorsAndRobotPoseToCache_ :640-697, getInitialFeaturePosition_ :699-724) and turns a
// frame's questions into a few batched calls.
//
// Batching without changing the outcome.  The reference walks the frame's features one by one and asks its questions in
// sequence; the questions of one feature depend on each other's answers (a stereo feature's second observation votes against the
// first one if that was just added; a cache is cleaned before it is tested for parallax), the questions of different features do
// not: every feature touches only its own cache and its own factors in the pose graph.  So every feature is a small task that runs
// the reference's statements in the reference's order until it needs an answer; all tasks' open questions go to the device in one
// call, the answers are handed back, and the tasks run on -- at most (observations per feature) + 2 rounds per frame.
#ifndef OBVI_HOST_VISUAL_FEATURE_FRONT_END_H_
#define OBVI_HOST_VISUAL_FEATURE_FRONT_END_H_

#include <functional>
#include <iostream>
#include <map>
#include <optional>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "obvi_frontend.h"
#include "obvi_runner.h"

namespace vslam_types_refactor {

// visual_feature_front_end.h:165-214
struct VisualFeatureCachedInfo {
  bool is_cache_cleaned_ = false;
  std::map<FrameId, std::vector<ReprojectionErrorFactor>> frame_ids_and_reprojection_err_factors_;
  std::map<FrameId, std::optional<Pose3D>> frame_ids_and_poses_;
  void addFactorsAndRobotPose(const FrameId& frame_id, const std::vector<ReprojectionErrorFactor>& factors, const std::optional<Pose3D>& robot_pose) {
    frame_ids_and_reprojection_err_factors_[frame_id] = factors;
    frame_ids_and_poses_[frame_id] = robot_pose;
  }
  FrameId getMinFrameId() const { return frame_ids_and_reprojection_err_factors_.begin()->first; }
};

struct VisualFeatureFrontendParams {   // the constructor arguments (:216-256), defaults :470-483
  double min_visual_feature_parallax_pixel_requirement_ = 5.0;
  double min_visual_feature_parallax_robot_transl_requirement_ = 0.1;
  double min_visual_feature_parallax_robot_orient_requirement_ = 0.05;
  bool enforce_min_pixel_parallax_requirement_ = true;
  bool enforce_min_robot_pose_parallax_requirement_ = true;
  double inlier_epipolar_err_thresh_ = 8.0;
  size_t check_pase_n_frames_for_epipolar_err_ = 5;   // (sic)
  bool enforce_epipolar_error_requirement_ = true;
  bool early_votes_return_ = true;
  double inlier_majority_percentage_ = 0.5;
};

class VisualFeatureFrontend {
 public:
  // `handle`: any handle of the library (it supplies the device and the stream; the bundle-adjustment state is not touched).
  VisualFeatureFrontend(obvi_ba_handle* handle, const std::function<bool(const FrameId&)>& gba_checker, const VisualFeatureFrontendParams& params)
      : handle_(handle), gba_checker_(gba_checker), p_(params) {}

  // visual_feature_front_end.h:262-450; max_frame_id is the frame being added
  bool addVisualFeatureObservations(const OfflineProblemData& data, const MainPgPtr& pose_graph, const FrameId& min_frame_id, const FrameId& max_frame_id) {
    if (!prepareCameras_(pose_graph) || !preparePoses_(data)) return false;
    std::vector<Task> tasks;
    if (max_frame_id < data.visual_obs_by_frame_.size()) {
      // one task per feature of the frame, its observations in file order (the reference iterates pixel_by_camera_id)
      std::unordered_map<FeatureId, size_t> task_of;
      for (const auto& o : data.visual_obs_by_frame_[max_frame_id]) {
        auto it = task_of.find(o.feature_id);
        if (it == task_of.end()) { it = task_of.emplace(o.feature_id, tasks.size()).first; tasks.emplace_back(); tasks.back().feature_id = o.feature_id; }
        tasks[it->second].factors.push_back(ReprojectionErrorFactor{max_frame_id, o.feature_id, o.camera_id, o.pixel, data.reprojection_error_std_dev_});
      }
    }
    std::optional<Pose3D> init_robot_pose;
    if (max_frame_id < data.robot_poses_.size()) init_robot_pose = data.robot_poses_[max_frame_id];
    for (Task& t : tasks) {
      t.init_robot_pose = init_robot_pose;
      const bool added = added_feature_ids_.count(t.feature_id) != 0;
      const bool in_init_cache = pending_feature_factors_for_initialized_features_.count(t.feature_id) != 0;
      if (in_init_cache && !added) { std::cerr << "Feature " << t.feature_id << " should already be added to the pose graph!" << std::endl; return false; }
      t.kind = in_init_cache ? Task::kInitializedInCache : (added ? Task::kAdded : Task::kPending);
      if (t.kind == Task::kPending && !pending_feature_factors_.count(t.feature_id)) pending_feature_factors_[t.feature_id] = VisualFeatureCachedInfo();
    }
    // run the tasks in lock step
    for (;;) {
      bool any_open = false;
      for (Task& t : tasks) { if (!t.done) advance_(t, data, pose_graph, min_frame_id, max_frame_id); any_open |= !t.done; }
      if (!any_open) break;
      if (!answerVotes_(tasks, data) || !answerParallax_(tasks, min_frame_id)) return false;
    }
    if (gba_checker_(max_frame_id)) {                                                                                        // :415-447
      std::vector<FeatureId> ids;
      for (const auto& c : pending_feature_factors_) ids.push_back(c.first);
      std::vector<uint8_t> ok;
      if (!parallaxOf_(ids, pending_feature_factors_, min_frame_id, &ok)) return false;
      for (size_t i = 0; i < ids.size(); ++i)
        if (ok[i]) {
          const auto pos = data.initial_feature_positions_.find(ids[i]);
          if (pos == data.initial_feature_positions_.end()) continue;
          initializeFeature_(data, pose_graph, ids[i], pos->second);
        }
    }
    return true;
  }

  size_t numPendingFeatures() const { return pending_feature_factors_.size(); }
  size_t numAddedFeatures() const { return added_feature_ids_.size(); }
  size_t numPendingInitializedFeatures() const { return pending_feature_factors_for_initialized_features_.size(); }
  uint64_t numVoteQuestions() const { return n_vote_questions_; }
  uint64_t numVoteCalls() const { return n_vote_calls_; }
  uint64_t numRejectedFactors() const { return n_rejected_; }

 private:
  typedef std::map<FrameId, std::vector<ReprojectionErrorFactor>> FactorsByFrame;
  struct Question { ReprojectionErrorFactor candidate; FactorsByFrame refs; bool inlier = false; };
  struct Task {
    enum Kind { kInitializedInCache, kAdded, kPending } kind = kPending;
    FeatureId feature_id = 0;
    std::vector<ReprojectionErrorFactor> factors;
    std::optional<Pose3D> init_robot_pose;
    bool done = false;
    int pc = 0;                  // where the task's program stands
    size_t r = 0;                // kAdded: index of the factor being decided
    std::vector<Question> votes; // open vote questions (answered between two advance_ calls)
    bool wants_parallax = false, parallax_ok = false;
    // a cache add in flight (addFactorsAndRobotPoseToCache_ with the epipolar test): which cache, and which branch asked the questions
    VisualFeatureCachedInfo* cache = nullptr;
    bool cache_was_cleaned = false;
  };

  // ---- addFactorsAndRobotPoseToCache_ (:640-697), split at its questions -----------------------------------------------------
  // returns true if the add is complete, false if questions were posted (finishCacheAdd_ completes it)
  bool beginCacheAdd_(Task& t, VisualFeatureCachedInfo& cache, const FrameId& frame_id) {
    if (!p_.enforce_epipolar_error_requirement_) { cache.addFactorsAndRobotPose(frame_id, t.factors, t.init_robot_pose); return true; }
    t.cache = &cache; t.cache_was_cleaned = cache.is_cache_cleaned_;
    t.votes.clear();
    if (cache.is_cache_cleaned_) {
      for (const auto& f : t.factors) t.votes.push_back(Question{f, cache.frame_ids_and_reprojection_err_factors_, false});
    } else {
      cache.addFactorsAndRobotPose(frame_id, t.factors, t.init_robot_pose);
      for (const auto& fr : cache.frame_ids_and_reprojection_err_factors_)
        for (const auto& f : fr.second) t.votes.push_back(Question{f, cache.frame_ids_and_reprojection_err_factors_, false});
    }
    return t.votes.empty();
  }
  void finishCacheAdd_(Task& t, const FrameId& frame_id) {
    VisualFeatureCachedInfo& cache = *t.cache;
    if (t.cache_was_cleaned) {
      std::vector<ReprojectionErrorFactor> to_add;
      for (const Question& q : t.votes) { if (q.inlier) to_add.push_back(q.candidate); else ++n_rejected_; }
      if (!to_add.empty()) cache.addFactorsAndRobotPose(frame_id, to_add, t.init_robot_pose);
    } else {
      FactorsByFrame cleaned;
      for (const Question& q : t.votes) if (q.inlier) cleaned[q.candidate.frame_id_].push_back(q.candidate);
      if (!cleaned.empty()) { cache.frame_ids_and_reprojection_err_factors_ = cleaned; cache.is_cache_cleaned_ = true; }
    }
    t.votes.clear(); t.cache = nullptr;
  }

  // getFactorsByFeatureIdFromPoseGraph_ (:486-509)
  void factorsFromPoseGraph_(const MainPgPtr& pg, const FeatureId& feature_id, const FrameId& candidate_frame, FactorsByFrame* out) const {
    const FrameId min_frame = candidate_frame - p_.check_pase_n_frames_for_epipolar_err_;   // unsigned wrap as in the reference
    for (FeatureFactorId id : pg->visualFactorIdsOfFeature(feature_id)) {
      ReprojectionErrorFactor f;
      if (pg->getVisualFactor(id, f) && f.frame_id_ > min_frame) (*out)[f.frame_id_].push_back(f);
    }
  }

  void initializeFeature_(const OfflineProblemData& data, const MainPgPtr& pg, const FeatureId& feature_id, const Position3d& unadjusted) {   // :393-411, :423-445
    const VisualFeatureCachedInfo& cache = pending_feature_factors_.at(feature_id);
    // getInitialFeaturePosition_ (:699-724): the position relative to the first observing frame's initial pose, re-attached to its optimised pose
    Position3d initial_position = unadjusted;
    const FrameId first = cache.getMinFrameId();
    const std::optional<RawPose3d> optim_first = pg->getRobotPose(first);
    if (first < data.robot_poses_.size() && optim_first.has_value())
      initial_position = combinePoseAndPosition(convertToPose3D(optim_first.value()), getPositionRelativeToPose(data.robot_poses_[first], unadjusted));
    pg->addFeature(feature_id, initial_position);
    for (const auto& fr : cache.frame_ids_and_reprojection_err_factors_) for (const auto& f : fr.second) pg->addVisualFactor(f);
    pending_feature_factors_.erase(feature_id);
    added_feature_ids_.insert(feature_id);
  }

  // ---- one task: the reference's statements for one feature of the frame, run until an answer is needed ------------------------
  void advance_(Task& t, const OfflineProblemData& data, const MainPgPtr& pg, const FrameId& min_frame_id, const FrameId& max_frame_id) {
    for (;;) {
      switch (t.kind) {
        case Task::kInitializedInCache: {                                                                                    // :322-345
          VisualFeatureCachedInfo& cache = pending_feature_factors_for_initialized_features_[t.feature_id];
          if (t.pc == 0) { t.pc = 1; if (!beginCacheAdd_(t, cache, max_frame_id)) return; }
          if (t.pc == 1) {
            if (t.cache != nullptr) finishCacheAdd_(t, max_frame_id);
            if (cache.is_cache_cleaned_) for (const auto& fr : cache.frame_ids_and_reprojection_err_factors_) for (const auto& f : fr.second) pg->addVisualFactor(f);
            pending_feature_factors_for_initialized_features_.erase(t.feature_id);
            t.done = true;
          }
          return;
        }
        case Task::kAdded: {                                                                                                 // :346-380
          if (t.pc == 0) {   // the next observation of the feature: votes of its factors in the pose graph
            if (t.r >= t.factors.size()) { t.done = true; return; }
            t.votes.assign(1, Question{t.factors[t.r], {}, false});
            factorsFromPoseGraph_(pg, t.feature_id, t.factors[t.r].frame_id_, &t.votes[0].refs);
            t.pc = 1;
            if (!t.votes[0].refs.empty()) return;   // isReprojectionErrorFactorInlierInPoseGraph_ asks only with references (:614-622)
          }
          if (t.pc == 1) {
            const bool had_refs = !t.votes[0].refs.empty(), inlier = had_refs && t.votes[0].inlier;
            t.votes.clear();
            if (inlier) { pg->addVisualFactor(t.factors[t.r]); ++t.r; t.pc = 0; continue; }
            if (had_refs) { ++n_rejected_; ++t.r; t.pc = 0; continue; }
            // not seen in the last frames: all of the frame's factors go to the cache of initialised features (:357-378)
            VisualFeatureCachedInfo& cache = pending_feature_factors_for_initialized_features_[t.feature_id];
            t.pc = 2;
            if (!beginCacheAdd_(t, cache, max_frame_id)) return;
          }
          if (t.pc == 2) {
            if (t.cache != nullptr) finishCacheAdd_(t, max_frame_id);
            ++t.r; t.pc = 0;
            continue;
          }
          return;
        }
        case Task::kPending: {                                                                                               // :381-412
          VisualFeatureCachedInfo& cache = pending_feature_factors_[t.feature_id];
          if (t.pc == 0) { t.pc = 1; if (!beginCacheAdd_(t, cache, max_frame_id)) return; }
          if (t.pc == 1) {
            if (t.cache != nullptr) finishCacheAdd_(t, max_frame_id);
            t.pc = 2; t.wants_parallax = true;
            return;
          }
          if (t.pc == 2) {
            t.wants_parallax = false;
            if (t.parallax_ok) {
              const auto pos = data.initial_feature_positions_.find(t.feature_id);
              if (pos != data.initial_feature_positions_.end()) initializeFeature_(data, pg, t.feature_id, pos->second);
            }
            t.done = true;
          }
          (void)min_frame_id;
          return;
        }
      }
    }
  }

  // ---- the device calls -------------------------------------------------------------------------------------------------------
  bool prepareCameras_(const MainPgPtr& pg) {
    if (!cam_index_.empty()) return true;
    for (const auto& k : pg->intrinsics()) {
      CameraExtrinsics e;
      if (!pg->getExtrinsicsForCamera(k.first, e)) continue;
      cam_index_[k.first] = (uint16_t)(cam_K_.size() / 4);
      cam_K_.insert(cam_K_.end(), {k.second.fx, k.second.fy, k.second.cx, k.second.cy});
      const double th = std::sqrt(e.orientation_[0] * e.orientation_[0] + e.orientation_[1] * e.orientation_[1] + e.orientation_[2] * e.orientation_[2]);
      const double s = th > 0.0 ? std::sin(th / 2) / th : 0.5;
      cam_ext_.insert(cam_ext_.end(), {e.orientation_[0] * s, e.orientation_[1] * s, e.orientation_[2] * s, std::cos(th / 2), e.transl_[0], e.transl_[1], e.transl_[2]});
    }
    return !cam_index_.empty();
  }
  bool preparePoses_(const OfflineProblemData& data) {   // the INITIAL estimates of the frames (getRobotPoseEstimateForFrameAffine, :536, :575)
    if (pose6_.size() == 6 * data.robot_poses_.size()) return true;
    pose6_.clear();
    for (const Pose3D& p : data.robot_poses_) pose6_.insert(pose6_.end(), {p.transl_[0], p.transl_[1], p.transl_[2], p.orientation_[0], p.orientation_[1], p.orientation_[2]});
    return true;
  }
  bool answerVotes_(std::vector<Task>& tasks, const OfflineProblemData& data) {
    std::vector<Question*> qs;
    for (Task& t : tasks) if (!t.done) for (Question& q : t.votes) qs.push_back(&q);
    if (qs.empty()) return true;
    std::vector<uint32_t> cand_pose, ref_pose, ref_frame;
    std::vector<uint16_t> cand_cam, ref_cam;
    std::vector<double> cand_pixel, ref_pixel;
    std::vector<uint64_t> ref_ptr(1, 0);
    std::vector<uint8_t> ref_skip, usable(qs.size(), 1);
    const FrameId n_poses = data.robot_poses_.size();
    for (size_t i = 0; i < qs.size(); ++i) {
      const ReprojectionErrorFactor& c = qs[i]->candidate;
      const auto cc = cam_index_.find(c.camera_id_);
      if (cc == cam_index_.end() || c.frame_id_ >= n_poses) usable[i] = 0;   // the reference returns false (no intrinsics / extrinsics / initial pose)
      cand_pose.push_back(usable[i] ? (uint32_t)c.frame_id_ : 0u); cand_cam.push_back(usable[i] ? cc->second : (uint16_t)0);
      cand_pixel.push_back(c.feature_pos_[0]); cand_pixel.push_back(c.feature_pos_[1]);
      for (const auto& fr : qs[i]->refs)
        for (const ReprojectionErrorFactor& f : fr.second) {
          const auto rc = cam_index_.find(f.camera_id_);
          const bool same = f.frame_id_ == c.frame_id_ && f.feature_id_ == c.feature_id_ && f.camera_id_ == c.camera_id_;   // shouldBeTheSame
          if (!same && (rc == cam_index_.end() || f.frame_id_ >= n_poses)) usable[i] = 0;
          ref_pose.push_back(f.frame_id_ < n_poses ? (uint32_t)f.frame_id_ : 0u); ref_cam.push_back(rc == cam_index_.end() ? (uint16_t)0 : rc->second);
          ref_pixel.push_back(f.feature_pos_[0]); ref_pixel.push_back(f.feature_pos_[1]);
          ref_frame.push_back((uint32_t)fr.first); ref_skip.push_back(same ? 1 : 0);
        }
      ref_ptr.push_back(ref_pose.size());
    }
    const obvi_epipolar_params ep{p_.inlier_epipolar_err_thresh_, p_.inlier_majority_percentage_, p_.early_votes_return_ ? 1 : 0, 0};
    std::vector<uint8_t> inlier(qs.size(), 0);
    if (ref_pose.empty()) { ref_pose.push_back(0); ref_cam.push_back(0); ref_pixel.assign(2, 0.0); ref_frame.push_back(0); ref_skip.push_back(0); }
    const int rc = obvi_frontend_epipolar_votes(handle_, (int32_t)(cam_K_.size() / 4), cam_K_.data(), cam_ext_.data(), (int64_t)n_poses, pose6_.data(), (int64_t)qs.size(), cand_pose.data(),
                                                cand_cam.data(), cand_pixel.data(), ref_ptr.data(), ref_pose.data(), ref_cam.data(), ref_pixel.data(), ref_frame.data(), ref_skip.data(), &ep,
                                                nullptr, nullptr, inlier.data());
    if (rc != 0) { std::cerr << "obvi_frontend_epipolar_votes failed (" << rc << ")" << std::endl; return false; }
    for (size_t i = 0; i < qs.size(); ++i) qs[i]->inlier = usable[i] && inlier[i] != 0;
    n_vote_questions_ += qs.size(); ++n_vote_calls_;
    return true;
  }
  // checkMinParallaxRequirements_ (:726-800) of the caches of `ids`
  bool parallaxOf_(const std::vector<FeatureId>& ids, const std::unordered_map<FeatureId, VisualFeatureCachedInfo>& caches, const FrameId& min_frame_id, std::vector<uint8_t>* ok) {
    ok->assign(ids.size(), 0);
    if (ids.empty()) return true;
    std::vector<uint64_t> frame_ptr(1, 0), obs_ptr(1, 0);
    std::vector<uint8_t> has_pose;
    std::vector<double> pose6, pixel;
    for (const FeatureId& id : ids) {
      const VisualFeatureCachedInfo& c = caches.at(id);
      for (auto it = c.frame_ids_and_reprojection_err_factors_.lower_bound(min_frame_id); it != c.frame_ids_and_reprojection_err_factors_.end(); ++it) {   // getOrderedFrameIdsGreaterThan
        const auto pz = c.frame_ids_and_poses_.find(it->first);
        const bool hp = pz != c.frame_ids_and_poses_.end() && pz->second.has_value();
        has_pose.push_back(hp ? 1 : 0);
        for (int k = 0; k < 3; ++k) pose6.push_back(hp ? pz->second->transl_[k] : 0.0);
        for (int k = 0; k < 3; ++k) pose6.push_back(hp ? pz->second->orientation_[k] : 0.0);
        std::map<CameraId, PixelCoord> by_cam;   // getCamIdsAndPixelsByFrame (:205-213): one pixel per camera, the last one wins
        for (const auto& f : it->second) by_cam[f.camera_id_] = f.feature_pos_;
        for (const auto& e : by_cam) { pixel.push_back(e.second[0]); pixel.push_back(e.second[1]); }
        obs_ptr.push_back(pixel.size() / 2);
      }
      frame_ptr.push_back(has_pose.size());
    }
    if (has_pose.empty()) return true;
    if (pixel.empty()) pixel.assign(2, 0.0);
    const obvi_parallax_params pp{p_.min_visual_feature_parallax_pixel_requirement_, p_.min_visual_feature_parallax_robot_transl_requirement_,
                                  p_.min_visual_feature_parallax_robot_orient_requirement_, p_.enforce_min_pixel_parallax_requirement_ ? 1 : 0,
                                  p_.enforce_min_robot_pose_parallax_requirement_ ? 1 : 0};
    const int rc = obvi_frontend_parallax(handle_, (int64_t)ids.size(), frame_ptr.data(), has_pose.data(), pose6.data(), obs_ptr.data(), pixel.data(), &pp, ok->data());
    if (rc != 0) { std::cerr << "obvi_frontend_parallax failed (" << rc << ")" << std::endl; return false; }
    return true;
  }
  bool answerParallax_(std::vector<Task>& tasks, const FrameId& min_frame_id) {
    std::vector<FeatureId> ids;
    std::vector<Task*> who;
    for (Task& t : tasks) if (!t.done && t.wants_parallax && t.votes.empty()) { ids.push_back(t.feature_id); who.push_back(&t); }
    std::vector<uint8_t> ok;
    if (!parallaxOf_(ids, pending_feature_factors_, min_frame_id, &ok)) return false;
    for (size_t i = 0; i < who.size(); ++i) who[i]->parallax_ok = ok[i] != 0;
    return true;
  }

  obvi_ba_handle* handle_;
  std::function<bool(const FrameId&)> gba_checker_;
  VisualFeatureFrontendParams p_;
  std::unordered_set<FeatureId> added_feature_ids_;
  std::unordered_map<FeatureId, VisualFeatureCachedInfo> pending_feature_factors_;
  // initialised features that reappeared after a gap are cached apart from the not yet initialised ones (:459-463)
  std::unordered_map<FeatureId, VisualFeatureCachedInfo> pending_feature_factors_for_initialized_features_;
  std::unordered_map<CameraId, uint16_t> cam_index_;
  std::vector<double> cam_K_, cam_ext_, pose6_;
  uint64_t n_vote_questions_ = 0, n_vote_calls_ = 0, n_rejected_ = 0;
};

}  // namespace vslam_types_refactor
#endif  // OBVI_HOST_VISUAL_FEATURE_FRONT_END_H_
