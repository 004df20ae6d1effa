// frontend_kernels.hip -- visual-feature front-end gating on the device (include/obvi_frontend.h).
// Two embarrassingly parallel questions the reference asks per new observation / per pending feature, batched per frame by the
// host mirror: the epipolar-consistency votes (visual_feature_front_end.h:511-602 on top of :52-132) and the minimum-parallax
// test (:726-800).  A frame of the reference's data sets brings ~1 500 observations with ~5 reference frames each: the work is
// small and latency-bound, so the kernels are one thread per candidate / per feature, and the call is one upload, one launch,
// one read-back on the handle's stream.
#include <algorithm>
#include <vector>

#include "../../include/obvi_frontend.h"
#include "ba_device.h"
#include "frontend_math.h"
#include "host_util.h"

namespace obvi {
namespace {

constexpr int kBlock = 128;

struct EpipolarBatch {
  const DevCam* cams; const double* poses;
  int64_t n_cand; const uint32_t* cand_pose; const uint16_t* cand_cam; const double* cand_pixel;
  const uint64_t* ref_ptr; const uint32_t* ref_pose; const uint16_t* ref_cam; const double* ref_pixel; const uint32_t* ref_frame; const uint8_t* ref_skip;
  double thresh, majority; int early_return;
};

__global__ void __launch_bounds__(kBlock) k_epipolar_votes(EpipolarBatch b, uint32_t* votes_out, uint32_t* voters_out, uint8_t* inlier_out, double* err_out) {
  const int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x;
  if (i >= b.n_cand) return;
  const DevCam cam2 = b.cams[b.cand_cam[i]];
  const double* pose2 = b.poses + 6 * (int64_t)b.cand_pose[i];
  const double px2[2] = {b.cand_pixel[2 * i], b.cand_pixel[2 * i + 1]};
  uint32_t votes = 0, voters = 0;
  const uint64_t k0 = b.ref_ptr[i], k1 = b.ref_ptr[i + 1];
  for (uint64_t k = k0; k < k1; ++k) {
    if (err_out == nullptr) {
      // :596-599 with early_votes_return_ the decision is taken after the first (earliest) reference frame, whatever it held
      if (b.early_return && k > k0 && b.ref_frame[k] != b.ref_frame[k0]) break;
      if (b.ref_skip && b.ref_skip[k]) continue;                                         // shouldBeTheSame(candidate) :551-553
    }
    double e[2];
    epipolar_error_vec(b.cams[b.ref_cam[k]], cam2, b.poses + 6 * (int64_t)b.ref_pose[k], pose2, b.ref_pixel + 2 * k, px2, e);
    if (err_out) { err_out[2 * k] = e[0]; err_out[2 * k + 1] = e[1]; continue; }
    if (sqrt(e[0] * e[0] + e[1] * e[1]) < b.thresh) ++votes;                              // :590-593
    ++voters;
  }
  if (err_out) return;
  if (votes_out) votes_out[i] = votes;
  if (voters_out) voters_out[i] = voters;
  if (inlier_out) inlier_out[i] = ((double)votes / (double)voters) > b.majority ? 1 : 0;   // 0 / 0 = NaN: not an inlier (:598, :601)
}

struct ParallaxBatch {
  int64_t n_feat; const uint64_t* frame_ptr; const uint8_t* has_pose; const double* pose6; const uint64_t* obs_ptr; const double* pixel;
  double min_pixel, min_transl, min_orient; int enforce_pixel, enforce_pose;
};
__global__ void __launch_bounds__(kBlock) k_parallax(ParallaxBatch b, uint8_t* satisfied) {
  const int64_t f = blockIdx.x * (int64_t)kBlock + threadIdx.x;
  if (f >= b.n_feat) return;
  const uint64_t k0 = b.frame_ptr[f], k1 = b.frame_ptr[f + 1];
  uint8_t ok = 0;
  if (k1 - k0 > 1) {                                                                        // :735-737
    for (uint64_t i = k0; i + 1 < k1 && !ok; ++i)
      for (uint64_t j = i + 1; j < k1 && !ok; ++j) {
        bool pixel_req = false, pose_req = false;
        if (b.enforce_pose && b.has_pose[i] && b.has_pose[j]) {                             // :755-768
          double tn, ang;
          relative_motion(b.pose6 + 6 * i, b.pose6 + 6 * j, &tn, &ang);
          pose_req = tn >= b.min_transl || ang >= b.min_orient;
        }
        if (b.enforce_pixel)                                                                 // :769-781
          for (uint64_t a = b.obs_ptr[i]; a < b.obs_ptr[i + 1]; ++a)
            for (uint64_t c = b.obs_ptr[j]; c < b.obs_ptr[j + 1]; ++c) {
              const double dx = b.pixel[2 * a] - b.pixel[2 * c], dy = b.pixel[2 * a + 1] - b.pixel[2 * c + 1];
              if (sqrt(dx * dx + dy * dy) >= b.min_pixel) pixel_req = true;
            }
        bool req;                                                                            // :782-795
        if (b.enforce_pose && !b.enforce_pixel) req = pose_req;
        else if (!b.enforce_pose && b.enforce_pixel) req = pixel_req;
        else if (b.enforce_pose && b.enforce_pixel) req = pose_req && pixel_req;
        else req = true;
        if (req) ok = 1;
      }
  }
  satisfied[f] = ok;
}

int epipolar_call(obvi_ba_handle* h, int32_t n_cams, const double* K4, const double* ext7, int64_t n_poses, const double* pose6, int64_t n_cand,
                  const uint32_t* cand_pose, const uint16_t* cand_cam, const double* cand_pixel, const uint64_t* ref_ptr, const uint32_t* ref_pose,
                  const uint16_t* ref_cam, const double* ref_pixel, const uint32_t* ref_frame, const uint8_t* ref_skip, const obvi_epipolar_params* prm,
                  uint32_t* votes, uint32_t* voters, uint8_t* inlier, double* err) {
  if (!h) return OBVI_ERR_INVALID_ARGUMENT;
  if (n_cams < 0 || n_poses < 0 || n_cand < 0 || (n_cand > 0 && (!K4 || !ext7 || !pose6 || !cand_pose || !cand_pixel || !ref_ptr))) return handle_fail(h, OBVI_ERR_INVALID_ARGUMENT, "epipolar votes: bad arguments");
  if (n_cand == 0) return OBVI_OK;
  const uint64_t n_ref = ref_ptr[n_cand];
  if (n_ref > 0 && (!ref_pose || !ref_pixel || (!err && !ref_frame))) return handle_fail(h, OBVI_ERR_INVALID_ARGUMENT, "epipolar votes: bad arguments");
  if (!err && !prm) return handle_fail(h, OBVI_ERR_INVALID_ARGUMENT, "epipolar votes: no parameters");
  for (int64_t i = 0; i < n_cand; ++i) {
    if (cand_pose[i] >= (uint64_t)n_poses || (cand_cam ? cand_cam[i] : 0) >= n_cams || ref_ptr[i + 1] < ref_ptr[i]) return handle_fail(h, OBVI_ERR_OUT_OF_RANGE, "epipolar votes: candidate index out of range");
  }
  for (uint64_t k = 0; k < n_ref; ++k)
    if (ref_pose[k] >= (uint64_t)n_poses || (ref_cam ? ref_cam[k] : 0) >= n_cams) return handle_fail(h, OBVI_ERR_OUT_OF_RANGE, "epipolar votes: reference index out of range");
  try {
    OBVI_HIP(hipSetDevice(handle_device(h)));
    hipStream_t s = handle_stream(h);
    std::vector<DevCam> cams((size_t)n_cams);
    for (int c = 0; c < n_cams; ++c) make_dev_cam(K4 + 4 * c, ext7 + 7 * c, &cams[c]);
    std::vector<uint16_t> ccam((size_t)n_cand, 0), rcam((size_t)n_ref, 0);
    if (cand_cam) std::copy(cand_cam, cand_cam + n_cand, ccam.begin());
    if (ref_cam) std::copy(ref_cam, ref_cam + n_ref, rcam.begin());
    DevBuf<DevCam> d_cams; DevBuf<double> d_poses, d_cpix, d_rpix, d_err; DevBuf<uint32_t> d_cpose, d_rpose, d_rframe, d_votes, d_voters;
    DevBuf<uint16_t> d_ccam, d_rcam; DevBuf<uint64_t> d_ptr; DevBuf<uint8_t> d_skip, d_inl;
    d_cams.upload(cams, s); d_poses.upload(pose6, (size_t)(6 * n_poses), s); d_cpix.upload(cand_pixel, (size_t)(2 * n_cand), s);
    d_cpose.upload(cand_pose, (size_t)n_cand, s); d_ccam.upload(ccam, s); d_ptr.upload(ref_ptr, (size_t)n_cand + 1, s);
    d_rpose.upload(ref_pose, (size_t)n_ref, s); d_rcam.upload(rcam, s); d_rpix.upload(ref_pixel, (size_t)(2 * n_ref), s);
    if (ref_frame) d_rframe.upload(ref_frame, (size_t)n_ref, s);
    if (ref_skip) d_skip.upload(ref_skip, (size_t)n_ref, s);
    d_votes.resize((size_t)n_cand); d_voters.resize((size_t)n_cand); d_inl.resize((size_t)n_cand);
    if (err) d_err.resize((size_t)(2 * n_ref) + 1);
    EpipolarBatch b{d_cams.get(), d_poses.get(), n_cand, d_cpose.get(), d_ccam.get(), d_cpix.get(), d_ptr.get(), d_rpose.get(), d_rcam.get(), d_rpix.get(),
                    ref_frame ? d_rframe.get() : nullptr, ref_skip ? d_skip.get() : nullptr, prm ? prm->inlier_epipolar_err_thresh : 0.0, prm ? prm->inlier_majority_percentage : 0.0,
                    prm ? prm->early_votes_return : 0};
    hipLaunchKernelGGL(k_epipolar_votes, dim3((unsigned)((n_cand + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, b, d_votes.get(), d_voters.get(), d_inl.get(), err ? d_err.get() : nullptr);
    OBVI_HIP(hipGetLastError());
    if (err) d_err.download(err, (size_t)(2 * n_ref), s);
    else {
      if (votes) d_votes.download(votes, (size_t)n_cand, s);
      if (voters) d_voters.download(voters, (size_t)n_cand, s);
      if (inlier) d_inl.download(inlier, (size_t)n_cand, s);
    }
    OBVI_HIP(hipStreamSynchronize(s));
    return OBVI_OK;
  } catch (const HipError& e) {
    return handle_fail(h, OBVI_ERR_HIP, e.what);
  } catch (const std::exception& e) {
    return handle_fail(h, OBVI_ERR_HIP, e.what());
  } catch (...) {
    return handle_fail(h, OBVI_ERR_HIP, "unknown host exception");
  }
}

}  // namespace
}  // namespace obvi

extern "C" {

int obvi_frontend_epipolar_votes(obvi_ba_handle* h, int32_t n_cams, const double* K4, const double* ext7, int64_t n_poses, const double* pose6, int64_t n_cand,
                                 const uint32_t* cand_pose, const uint16_t* cand_cam, const double* cand_pixel, const uint64_t* ref_ptr, const uint32_t* ref_pose,
                                 const uint16_t* ref_cam, const double* ref_pixel, const uint32_t* ref_frame, const uint8_t* ref_skip, const obvi_epipolar_params* params,
                                 uint32_t* votes, uint32_t* voters, uint8_t* inlier) {
  return obvi::epipolar_call(h, n_cams, K4, ext7, n_poses, pose6, n_cand, cand_pose, cand_cam, cand_pixel, ref_ptr, ref_pose, ref_cam, ref_pixel, ref_frame, ref_skip, params, votes, voters, inlier, nullptr);
}
int obvi_frontend_epipolar_errors(obvi_ba_handle* h, int32_t n_cams, const double* K4, const double* ext7, int64_t n_poses, const double* pose6, int64_t n_cand,
                                  const uint32_t* cand_pose, const uint16_t* cand_cam, const double* cand_pixel, const uint64_t* ref_ptr, const uint32_t* ref_pose,
                                  const uint16_t* ref_cam, const double* ref_pixel, double* err) {
  if (!err) return OBVI_ERR_INVALID_ARGUMENT;
  return obvi::epipolar_call(h, n_cams, K4, ext7, n_poses, pose6, n_cand, cand_pose, cand_cam, cand_pixel, ref_ptr, ref_pose, ref_cam, ref_pixel, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, err);
}

int obvi_frontend_parallax(obvi_ba_handle* h, int64_t n_feat, const uint64_t* frame_ptr, const uint8_t* has_pose, const double* pose6, const uint64_t* obs_ptr,
                           const double* pixel, const obvi_parallax_params* prm, uint8_t* satisfied) {
  using namespace obvi;   // NOLINT
  if (!h) return OBVI_ERR_INVALID_ARGUMENT;
  if (n_feat < 0 || !prm || (n_feat > 0 && (!frame_ptr || !satisfied))) return handle_fail(h, OBVI_ERR_INVALID_ARGUMENT, "parallax: bad arguments");
  if (n_feat == 0) return OBVI_OK;
  const uint64_t n_frames = frame_ptr[n_feat];
  if (n_frames > 0 && (!obs_ptr || !has_pose || !pose6)) return handle_fail(h, OBVI_ERR_INVALID_ARGUMENT, "parallax: bad arguments");
  for (int64_t f = 0; f < n_feat; ++f) if (frame_ptr[f + 1] < frame_ptr[f]) return handle_fail(h, OBVI_ERR_OUT_OF_RANGE, "parallax: frame offsets must ascend");
  const uint64_t n_obs = n_frames ? obs_ptr[n_frames] : 0;
  if (n_obs > 0 && !pixel) return handle_fail(h, OBVI_ERR_INVALID_ARGUMENT, "parallax: bad arguments");
  try {
    OBVI_HIP(hipSetDevice(handle_device(h)));
    hipStream_t s = handle_stream(h);
    DevBuf<uint64_t> d_fptr, d_optr; DevBuf<uint8_t> d_has, d_out; DevBuf<double> d_pose, d_pix;
    d_fptr.upload(frame_ptr, (size_t)n_feat + 1, s);
    static const uint64_t zero_ptr[1] = {0};
    d_optr.upload(n_frames ? obs_ptr : zero_ptr, (size_t)n_frames + 1, s);
    d_has.upload(has_pose, (size_t)n_frames, s); d_pose.upload(pose6, (size_t)(6 * n_frames), s); d_pix.upload(pixel, (size_t)(2 * n_obs), s);
    d_out.resize((size_t)n_feat);
    ParallaxBatch b{n_feat, d_fptr.get(), d_has.get(), d_pose.get(), d_optr.get(), d_pix.get(), prm->min_visual_feature_parallax_pixel_requirement,
                    prm->min_visual_feature_parallax_robot_transl_requirement, prm->min_visual_feature_parallax_robot_orient_requirement,
                    prm->enforce_min_pixel_parallax_requirement, prm->enforce_min_robot_pose_parallax_requirement};
    hipLaunchKernelGGL(k_parallax, dim3((unsigned)((n_feat + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, b, d_out.get());
    OBVI_HIP(hipGetLastError());
    d_out.download(satisfied, (size_t)n_feat, s);
    OBVI_HIP(hipStreamSynchronize(s));
    return OBVI_OK;
  } catch (const HipError& e) {
    return handle_fail(h, OBVI_ERR_HIP, e.what);
  } catch (const std::exception& e) {
    return handle_fail(h, OBVI_ERR_HIP, e.what());
  } catch (...) {
    return handle_fail(h, OBVI_ERR_HIP, "unknown host exception");
  }
}

}  // extern "C"
