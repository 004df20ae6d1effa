/*
 * obvi_frontend.h -- C ABI of the visual-feature front-end gating that sits immediately before the bundle-adjustment path
 * (SURVEY.md 8f #3): the per-observation epipolar-consistency votes and the per-feature parallax test that decide which
 * reprojection factors and which features enter the pose graph.
 *
 * Reference (all under /root/reference/include/refactoring/visual_feature_frontend/visual_feature_front_end.h):
 *   getNormalizedEpipolarErrorVec            :52-132   vector from a pixel to its projection on the epipolar line of another view
 *   isReprojectionErrorFactorInlier          :511-602  votes of the reference observations of the same feature, majority rule,
 *                                                      early return after the earliest reference frame
 *   checkMinParallaxRequirements_            :726-800  any pair of cached frames with enough robot motion / pixel displacement
 * The stateful part (caches of pending features, what is added when: addVisualFeatureObservations :262-450) is host logic and is
 * mirrored in obvi-slam_amd/host/obvi_visual_feature_front_end.h, which runs a frame's features in lock step and batches their
 * questions into these two calls.
 *
 * Conventions as in obvi_ba.h: host pointers owned by the caller, fp64, 0 / negative obvi_status, nothing throws.  The handle
 * supplies the device and the stream; the calls do not touch the bundle-adjustment state.
 */
#ifndef OBVI_FRONTEND_H_
#define OBVI_FRONTEND_H_

#include <stdint.h>

#include "obvi_ba.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  double inlier_epipolar_err_thresh;   /* 8.0 px  (config visual_feature_params.inlier_epipolar_err_thresh) */
  double inlier_majority_percentage;   /* 0.5     (visual_feature_front_end.h:483) */
  int32_t early_votes_return;          /* 1       (:481) decide on the votes of the EARLIEST reference frame only (:596-599) */
  int32_t reserved;
} obvi_epipolar_params;

/* isReprojectionErrorFactorInlier for n_cand candidate observations.
 *   cameras   [n_cams] intrinsics fx fy cx cy and extrinsics T_robot<-camera as qx qy qz qw tx ty tz (as obvi_ba_set_cameras)
 *   pose6     [n_poses][6] robot poses T_world<-robot as (t, axis-angle): the INITIAL estimates of the frames
 *             (input_problem_data.getRobotPoseEstimateForFrameAffine, :536, :575), not the optimised ones
 *   candidate i: pose cand_pose[i], camera cand_cam[i], pixel cand_pixel[2 i..]
 *   its reference observations: entries [ref_ptr[i], ref_ptr[i+1]) of ref_*, in ascending frame order (the std::map order of
 *             frame_ids_and_factors); ref_frame groups them by frame, ref_skip[k] != 0 marks a reference that
 *             shouldBeTheSame as the candidate (low_level_feature_pose_graph.h:122-125) and does not vote (:551-553)
 * Outputs (any may be NULL): votes[i], voters[i] as counted up to the point the reference returns, inlier[i] =
 * votes / voters > inlier_majority_percentage in IEEE arithmetic (no voters: 0/0 -> not an inlier). */
int obvi_frontend_epipolar_votes(obvi_ba_handle* h, int32_t n_cams, const double* fx_fy_cx_cy, const double* ext_qxyzw_t,
                                 int64_t n_poses, const double* pose6, int64_t n_cand, const uint32_t* cand_pose,
                                 const uint16_t* cand_cam, const double* cand_pixel, const uint64_t* ref_ptr,
                                 const uint32_t* ref_pose, const uint16_t* ref_cam, const double* ref_pixel,
                                 const uint32_t* ref_frame, const uint8_t* ref_skip, const obvi_epipolar_params* params,
                                 uint32_t* votes, uint32_t* voters, uint8_t* inlier);
/* the error vectors themselves (test hook): err[k] for every (candidate, reference) pair in ref order, [n_refs][2] */
int obvi_frontend_epipolar_errors(obvi_ba_handle* h, int32_t n_cams, const double* fx_fy_cx_cy, const double* ext_qxyzw_t,
                                  int64_t n_poses, const double* pose6, int64_t n_cand, const uint32_t* cand_pose,
                                  const uint16_t* cand_cam, const double* cand_pixel, const uint64_t* ref_ptr,
                                  const uint32_t* ref_pose, const uint16_t* ref_cam, const double* ref_pixel, double* err);

typedef struct {
  double min_visual_feature_parallax_pixel_requirement;         /* 5.0 px  */
  double min_visual_feature_parallax_robot_transl_requirement;  /* 0.1 m   */
  double min_visual_feature_parallax_robot_orient_requirement;  /* 0.05 rad */
  int32_t enforce_min_pixel_parallax_requirement;                /* 1 */
  int32_t enforce_min_robot_pose_parallax_requirement;           /* 0 in config/base7a_2_fallback.json */
} obvi_parallax_params;

/* checkMinParallaxRequirements_ for n_feat pending features.  Feature f caches the frames [frame_ptr[f], frame_ptr[f+1]) -- ascending,
 * already restricted to frames >= min_frame_id (getOrderedFrameIdsGreaterThan, :733-734) --; cached frame k carries an optional robot
 * pose (has_pose[k], pose6[6 k..]) and the pixels [obs_ptr[k], obs_ptr[k+1]) of pixel[], one per camera
 * (getCamIdsAndPixelsByFrame, :205-213).  satisfied[f] = some pair of cached frames meets the enabled requirements. */
int obvi_frontend_parallax(obvi_ba_handle* h, int64_t n_feat, const uint64_t* frame_ptr, const uint8_t* has_pose,
                           const double* pose6, const uint64_t* obs_ptr, const double* pixel, const obvi_parallax_params* params,
                           uint8_t* satisfied);

#ifdef __cplusplus
}
#endif
#endif /* OBVI_FRONTEND_H_ */
