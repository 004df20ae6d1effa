// ba_handle.h -- what the translation units of libobvi_ba.so share: the handle (struct obvi_ba_handle: host mirrors of the problem, device
// buffers, the symbolic plan, the state of the last solve) and the small helpers every entry point uses (error plumbing, the per-call staging
// scope and API timer, the views of the handle that the kernel launchers take).  Split out of obvi_ba.cpp in round 5:
//   abi.cpp     lifetime, evaluate, debug / profiling hooks, covariances, outlier selection, state, multi-GPU switches
//   upload.cpp  obvi_ba_set_* (parameter blocks, factors, masks)
//   plan.cpp    the symbolic phase: reduced program, elimination order, Schur work lists, tile plan (prepare_plan), mask-only re-plan
//   lm.cpp      one LM step on the device (submit_step) and the trust-region loop (obvi_ba_solve)
#ifndef OBVI_BA_HANDLE_H_
#define OBVI_BA_HANDLE_H_
#include "../../include/obvi_ba.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <limits>
#include <mutex>
#include <numeric>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "ba_device.h"
#include "host_util.h"

using namespace obvi;  // NOLINT

namespace obvi_lib {


inline double wall_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

enum Phase { PH_POSE_CACHE = 0, PH_POINT_PASS, PH_POSE_PASS, PH_SMALL, PH_DIAG, PH_SCHUR, PH_SCHUR_BLOCKS, PH_CHOL, PH_BACKSUB, PH_APPLY, PH_COST, PH_COUNT };
inline const char* const kPhaseNames[PH_COUNT] = {"pose_cache", "point_pass", "pose_pass", "small_factors", "reduced_diag", "schur_window", "schur_blocks",
                                     "cholesky_solve", "point_backsub", "apply_step", "cost"};


}  // namespace obvi_lib
using namespace obvi_lib;  // NOLINT

struct obvi_ba_handle {
  int device = 0;
  int reproj_variant = OBVI_REPROJECTION_AUTODIFF;   // obvi_ba_options.reprojection_variant
  int od = 7;                                        // obvi_ba_options.object_block_size: parameters of an ellipsoid block, 7 (x y z yaw dx dy dz) or 9 (x y z ax ay az dx dy dz)
  bool deterministic = false;                        // obvi_ba_options.deterministic
  int32_t det_stride = 0;                            // ... workgroups each partial-sum slot behind d_scal has room for (ensure_det_slots)
  bool fused_potrf = true;                           // k_update_potrf (updates of level l + potrf of level l + 1 in one grid); switched off for the rest of the handle's life
                                                     // after a potrf workgroup timed out waiting for its jobs (HIP does not promise dispatch order): two launches per level then
  int potrf_wait_timeouts = 0;
  hipStream_t stream = nullptr;
  obvi::StagingArena staging;   // pinned; the uploads of an API call are copied through it (host_util.h)
  std::string err;

  // ---- host mirrors ----
  std::vector<DevCam> h_cams;
  int64_t P = 0, L = 0, O = 0;
  std::vector<uint8_t> h_pose_const, h_point_const, h_object_const;   // (h_point_const: in the INTERNAL feature order, below)
  // Internal feature numbering (round 6).  Everything the kernels index by feature -- d_point and its copies, the per-feature arrays, the observation lists, the plan --
  // uses ids in the order of the features' FIRST observing pose (obvi_ba_set_reproj decides it, for problems of OBVI_POINT_RENUMBER_MIN observations or more): a
  // wavefront's 64 observations then belong to neighbouring poses and the pose-cache gathers of the point pass, the back-substitution and the trial cost share cache
  // lines (1.6 % of an LM step of config #3 with the generator's random ids, profiles/r06_point_order.txt).  The caller's numbering stops at the ABI: set / get / update
  // of features, constness flags, parameter priors and column norms go through these maps (empty = identity).
  std::vector<uint32_t> h_pt_new_of_old, h_pt_old_of_new, scr_point_internal;
  bool pt_map_applied = false;   // d_point / h_point_const are in the internal order (false: no numbering, or the caller has just set a feature count the observations do not fit -- an error state until it is repaired)
  DevBuf<uint32_t> d_pt_new_of_old, d_pt_old_of_new, d_pt_map_tmp;
  DevBuf<double> d_pt_tmp;
  std::vector<double> h_obj_xy;   // (x, y) of every object AS UPLOADED (obvi_ba_set_objects): the spatial key that orders the shared tail (plan.cpp) -- the same on every rank
  // reprojection: sorted by (point, pose); perm[sorted] = caller index
  int64_t n_rp = 0;
  std::vector<uint32_t> h_rp_pose, h_rp_point, h_rp_perm, h_rp_inv, h_point_ptr;
  std::vector<uint8_t> h_rp_active;  // sorted order
  std::vector<int32_t> h_rp_yrow;
  std::vector<uint32_t> h_rq_src;    // CSR-by-pose copy: position -> index into the CSC-by-point arrays
  std::vector<uint32_t> scr_cursor, scr_wave_obs, scr_long_points, scr_pose_ptr;   // scratch of set_reproj, kept between calls
  std::vector<uint8_t> scr_pose_used, scr_obj_used, scr_point_used, scr_point_var, scr_is_pad; std::vector<int32_t> scr_pose_vid, scr_obj_vid;   // ... of prepare_masks
  double rp_huber = 1.0;
  int64_t n_bb = 0, n_sp = 0, n_lt = 0, n_rl = 0;
  std::vector<uint32_t> h_bb_obj, h_bb_pose, h_sp_obj, h_lt_obj, h_rl_a, h_rl_b;
  std::vector<uint8_t> h_bb_active, h_sp_active, h_lt_active, h_rl_active;
  double bb_huber = 1.0, bb_invalid = 1e6, sp_huber = 1.0, lt_huber = 1.0, rl_huber = 1.0;
  // largest block / camera index each factor family refers to (-1: none): re-checked against the current block counts before every
  // evaluate / solve, because blocks and cameras may be re-uploaded (with other counts) after the factors
  int64_t max_rp_pose = -1, max_rp_point = -1, max_rp_cam = -1, max_bb_obj = -1, max_bb_pose = -1, max_bb_cam = -1, max_sp_obj = -1, max_lt_obj = -1, max_rl_pose = -1;
  // bounding boxes as uploaded (pixels, (cov^-1)^1/2): the rectified corners and sqrt_inf on the device depend on the cameras and
  // are re-derived when the cameras change
  std::vector<uint16_t> h_bb_cam;
  std::vector<double> h_bb_corners, h_bb_m4;

  // ---- device: parameters ----
  DevBuf<DevCam> d_cams;
  DevBuf<double> d_pose, d_point, d_obj;           // current
  DevBuf<double> d_pose_c, d_point_c, d_obj_c;     // candidate
  DevBuf<double> d_pose_b, d_point_b, d_obj_b;     // best (minimum cost) iterate
  DevBuf<double> d_pose_s, d_point_s, d_obj_s;     // snapshot
  DevBuf<double> d_pose_e, d_point_e, d_obj_e;     // state at solve entry (handed back after a FAILURE)
  bool have_snapshot = false;
  DevBuf<PoseCache> d_pc, d_pc_c;
  bool pc_valid = false;                 // d_pc belongs to the poses in d_pose (an accepted step hands the candidate's cache over)
  bool tiles_cleared = false;            // the tiles and step accumulators were already cleared behind the previous LM step
  DevBuf<int32_t> d_pose_vid, d_obj_vid;
  DevBuf<uint8_t> d_point_var;
  // ---- device: factors ----
  DevBuf<uint32_t> d_rp_pose, d_rp_point, d_rp_perm, d_point_ptr, d_wave_obs, d_long_points;
  int64_t n_point_waves = 0, n_long_points = 0;
  DevBuf<uint16_t> d_rp_cam;
  DevBuf<double2> d_rp_pixel;
  DevBuf<double> d_rp_sigma;
  DevBuf<uint8_t> d_rp_active;
  DevBuf<int32_t> d_rp_yrow;             // per observation: row of its pose in the reduced system (prepare())
  DevBuf<uint32_t> d_rq_point, d_rq_pose_ptr;
  DevBuf<uint16_t> d_raw_cam; DevBuf<double2> d_raw_pixel; DevBuf<double> d_raw_sigma; DevBuf<uint32_t> d_rq_src;   // obvi_ba_set_reproj: the caller's arrays as they came + the by-pose order, sources of the gather on the device
  DevBuf<uint16_t> d_rq_cam;
  DevBuf<double2> d_rq_pixel;
  DevBuf<double> d_rq_sigma;
  DevBuf<uint8_t> d_rq_active;
  DevBuf<uint32_t> d_bb_obj, d_bb_pose, d_sp_obj, d_lt_obj, d_rl_a, d_rl_b;
  DevBuf<uint16_t> d_bb_cam;
  DevBuf<double> d_bb_rect, d_bb_sqrt_inf, d_sp_mean, d_sp_sqrt_inf, d_lt_mean, d_lt_sqrt_inf, d_rl_t, d_rl_R, d_rl_sqrt_inf;
  DevBuf<uint8_t> d_bb_active, d_sp_active, d_lt_active, d_rl_active;
  DevBuf<double> d_bb_blk;                                    // per-factor blocks of the bounding-box factors (k_bbox_gather)
  DevBuf<double> d_sm_blk; DevBuf<uint32_t> d_smt_ptr, d_smt_idx;   // deterministic mode: the same for the priors and relative-pose factors (k_small_gather)
  int32_t bb_pairs_unique = 1;
  DevBuf<uint32_t> d_bbo_ptr, d_bbo_idx, d_bbp_ptr, d_bbp_idx;   // ... and the factor lists by object / by pose (prepare())
  // ---- device: reduced system ----
  DevBuf<double> d_Hdiag, d_g, d_scale, d_lam, d_S, d_rhs, d_y, d_Linv;
  DevBuf<double> d_Ci, d_u, d_scale_l, d_Z, d_gl, d_lam_l;
  DevBuf<uint32_t> d_blk_row, d_blk_col, d_blk_ptr, d_pair_a, d_pair_b, d_chunk_ptr, d_chunk_points;
  DevBuf<int32_t> d_row_of_nat, d_chunk_f0, d_chunk_group;
  DevBuf<uint32_t> d_batch_first, d_batch_slot, d_slot_src;
  DevBuf<PlanVisit> d_plan_visits; DevBuf<uint32_t> d_plan_wg_ptr, d_plan_wg_slot0; DevBuf<int32_t> d_plan_frame;   // inputs of the device-side slot fill (plan.cpp, plan_kernels.hip)
  int32_t schur_twins = 0;
  int64_t nchunks = 0, npairs_window = 0;
  DevBuf<int32_t> d_tiles, d_lvl_k, d_trsm_ik, d_upd_ij, d_upd_kptr, d_upd_k, d_rh_i, d_rh_kptr, d_rh_k, d_col_ptr, d_col_i, d_bw_kj, d_bw_chains;
  DevBuf<int32_t> d_row_ptr, d_row_j, d_cov_slab, d_cov_cols, d_cov_first;   // row structure of L; covariance extraction scratch
  DevBuf<double> d_cov_Y, d_cov_out;
  std::vector<int32_t> h_obj_vid;          // object -> reduced object index (elimination order) or -1
  std::vector<int32_t> h_row_split;        // per level: workgroups per tile row in the multi-right-hand-side forward substitution
  DevBuf<uint8_t> d_upd_flag, d_is_pad;
  DevBuf<int32_t> d_job_signal, d_k_need, d_diag_done, d_pre_ptr, d_pre_j;
  DevBuf<int32_t> d_pose_row, d_obj_row;
  DevBuf<double> d_scal;
  DevBuf<double> d_eval_res, d_eval_sq;
  DevBuf<uint8_t> d_sel_mask;
  DevBuf<uint32_t> d_rp_inv;
  SelectScratch sel_scratch;
  uint64_t api_calls = 0;     // entry points run on this handle so far (OBVI_API_BEGIN)
  uint64_t eval_sq_call = 0;  // ... when d_eval_sq was last filled with the un-robustified block norms of the CURRENT state: a selection that is the very next call reuses them
  bool rp_inv_on_device = false;   // d_rp_inv holds h_rp_inv of the current reprojection factors
  double* h_scal = nullptr;  // pinned; the device writes the scalar block of an LM step straight into it and, behind a system-scope fence, the sequence number [SC_COUNT]
                             // (k_zero_tiles): the host polls the number instead of sleeping in hipStreamSynchronize (whose wake-up costs tens of microseconds)
  double scal_seq = 0.0;

  // ---- reduced-program bookkeeping (prepare()) ----
  bool dirty = true;                     // the symbolic plan must be rebuilt (blocks / factors / constness changed)
  bool mask_dirty = false;               // only factor masks changed since the plan was built: prepare_masks() may keep the plan
  // what the plan was built for: variable blocks and active factors.  A later state whose variable blocks and active factors are
  // subsets of these runs on the same plan (rows of dropped blocks become padding, masked observations contribute zeros)
  std::vector<int32_t> plan_pose_vid, plan_obj_vid;
  std::vector<uint8_t> plan_point_var, plan_is_pad, plan_rp_active, plan_bb_active, plan_sp_active, plan_lt_active, plan_rl_active;
  int64_t live_rows = 0;                 // 6 (variable poses) + od (variable objects) of the current state (== m_canon right after a full plan)
  int64_t nPv = 0, nOv = 0, nLv = 0, m = 0, m_canon = 0, num_params = 0, num_residuals = 0;
  int32_t nt = 0;
  int64_t nblk = 0, npairs = 0;
  int32_t nlevels = 0, nbw = 0;
  std::vector<int32_t> h_lvl_k_ptr, h_trsm_ptr, h_upd_ptr, h_rh_ptr, h_crit_upd, h_crit_rh, h_slices, h_bw_ptr;
  std::vector<int32_t> h_pose_row, h_obj_row, h_row_of_nat;   // reduced pose / object index -> first row of its diagonal block in the tile grid
  std::vector<uint8_t> h_is_pad;                // rows of the tile grid that belong to no block (identity)
  std::vector<int64_t> h_canon_row;   // canonical reduced index (poses by index, then objects) -> row of the tile grid
  int32_t ntiles = 0;
  int64_t n_trsm_jobs = 0, n_upd_products = 0;
  double chol_flops = 0.0;

  // ---- parameter priors (covariance extraction only) ----
  std::vector<uint8_t> h_pp_kind, h_pp_param; std::vector<uint32_t> h_pp_block; std::vector<double> h_pp_mean, h_pp_std;
  DevBuf<double> d_extra_c, d_extra_l;
  bool use_extra = false;                // the next submit_step adds d_extra_c / d_extra_l to the diagonal (obvi_ba_object_covariances)
  // ---- last solve ----
  std::vector<obvi_iteration_summary> iterations;
  obvi_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  std::vector<uint8_t> h_is_shared;      // per object index (caller order)
  int32_t rank = 0, world = 1;
  double tail_order_hash = 0.0;          // 40-bit hash (as double) of the shared objects' indices in tail order (plan.cpp)
  std::vector<int32_t> h_shared_ov;      // reduced object indices of the shared objects in the order of the tail (the same on every rank: plan.cpp)
  DevBuf<int32_t> d_shared_ov;
  DevBuf<uint8_t> d_obj_shared;
  DevBuf<double> d_xbuf, d_xbuf2;        // exchange buffers: main stream (tail, scalars) / side stream (shared blocks)
  int32_t tail_t0 = -1, tail_level0 = -1;   // first tile / first level of the shared tail (-1: none)

  // ---- phase timing ----
  hipEvent_t ev[PH_COUNT + 1] = {};       // start of each phase (+ end of the step) on the main stream
  hipEvent_t ev_end[PH_COUNT] = {};       // end of a phase that ran on the side stream
  bool phase_on_side[PH_COUNT] = {};
  hipStream_t stream2 = nullptr;          // side stream: kernels that do not depend on the point pass / Schur complement run beside them
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int ck_used = 0;
  int profiling = 0;                       // 2: per-kernel events inside the tile Cholesky
  std::vector<hipEvent_t> ck_pool; std::vector<int> ck_tags;
  double ck_ms[CK_COUNT] = {}; int64_t ck_launches[CK_COUNT] = {};
  double phase_ms[PH_COUNT] = {};
  int64_t phase_launches[PH_COUNT] = {};
};

namespace obvi_lib {


inline int fail(obvi_ba_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}
inline int hip_fail(obvi_ba_handle* h, const HipError& e) {
  char buf[512];
  std::snprintf(buf, sizeof(buf), "%s: %s (%s:%d)", e.what, hipGetErrorString(e.code), e.file, e.line);
  return fail(h, OBVI_ERR_HIP, buf);
}

constexpr size_t kStagingBytes = (size_t)16 << 20;   // a sliding window's upload is ~5 MB, its plan ~2 MB; what does not fit is copied the plain way
// every API call runs with its handle's staging arena as the destination of DevBuf::upload / h2d_async (host_util.h)
struct StagingScope {
  StagingArena* prev;
  explicit StagingScope(const obvi_ba_handle* h) : prev(tl_staging) { tl_staging = h ? const_cast<StagingArena*>(&h->staging) : nullptr; if (h) ++const_cast<obvi_ba_handle*>(h)->api_calls; }
  ~StagingScope() { tl_staging = prev; }
};
// OBVI_API_TIMING=1: wall time per entry point (and of the symbolic phase inside obvi_ba_solve), summed over the process, on stderr at exit
struct ApiTimes {
  struct Row { const char* name; double ms = 0.0; int64_t calls = 0; };
  std::mutex mu;
  std::vector<Row> rows;
  void add(const char* name, double ms) {
    std::lock_guard<std::mutex> lock(mu);
    for (Row& r : rows) if (r.name == name || std::strcmp(r.name, name) == 0) { r.ms += ms; ++r.calls; return; }
    rows.push_back({name, ms, 1});
  }
  ~ApiTimes() {
    for (const Row& r : rows) std::fprintf(stderr, "api timing: %-30s %9.2f ms in %7lld calls (%8.4f ms each)\n", r.name, r.ms, (long long)r.calls, r.ms / (double)r.calls);
  }
};
inline ApiTimes* api_times() {
  static ApiTimes* t = std::getenv("OBVI_API_TIMING") ? new ApiTimes : nullptr;
  static const bool registered = t && (std::atexit([] { delete api_times(); }), true);
  (void)registered;
  return t;
}
struct ApiTimer {
  const char* name; double t0;
  explicit ApiTimer(const char* n) : name(n), t0(api_times() ? wall_s() : 0.0) {}
  ~ApiTimer() { if (ApiTimes* t = api_times()) t->add(name, 1e3 * (wall_s() - t0)); }
};
#define OBVI_API_BEGIN try { StagingScope staging_scope_(h); ApiTimer api_timer_(__func__);
#define OBVI_API_END(h)                                           \
  }                                                               \
  catch (const HipError& e) { return hip_fail(h, e); }            \
  catch (const std::bad_alloc&) { return fail(h, OBVI_ERR_HIP, "host allocation failed"); } \
  catch (const std::exception& e) { return fail(h, OBVI_ERR_HIP, std::string("host exception: ") + e.what()); } \
  catch (...) { return fail(h, OBVI_ERR_HIP, "unknown host exception"); }

inline void make_cam(const double* K4, const double* e, DevCam* c) {
  // inverse of the extrinsics T_robot<-camera: cam_to_robot_tf_inv_ (reprojection_cost_functor.cpp:10-13)
  double q[4] = {e[0], e[1], e[2], e[3]};
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (double& v : q) v /= n;
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
                          {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
                          {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) c->Rinv[3 * i + j] = R[j][i];
    c->tinv[i] = -(R[0][i] * e[4] + R[1][i] * e[5] + R[2][i] * e[6]);
  }
  c->fx = K4[0]; c->fy = K4[1]; c->cx = K4[2]; c->cy = K4[3];
  c->depth_min = -std::numeric_limits<double>::infinity();   // the production functor: no clamp (set_cameras sets it for the analytic variant)
}

inline void sync(obvi_ba_handle* h) { OBVI_HIP(hipStreamSynchronize(h->stream)); h->staging.rewind(); }
// End of a function that uploaded from its caller's buffers or from local vectors: everything that went through the arena is safe
// without waiting; a copy that went straight from pageable memory is not.
inline void finish_upload(obvi_ba_handle* h) { if (h->staging.spilled || tl_staging != &h->staging) sync(h); }

// Waits for the scalar block of the step just submitted: polls the sequence number the device writes behind the block, and asks the
// stream now and then so that a failed launch surfaces as an error instead of a hang.
inline void wait_scalars(obvi_ba_handle* h) {
  volatile const double* seq = h->h_scal + SC_COUNT;
  for (;;) {
    for (int spin = 0; spin < 4096; ++spin) {
      if (*seq == h->scal_seq) { std::atomic_thread_fence(std::memory_order_acquire); return; }
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#else
      std::this_thread::yield();
#endif
    }
    const hipError_t q = hipStreamQuery(h->stream);
    if (q == hipSuccess) { if (*seq == h->scal_seq) { std::atomic_thread_fence(std::memory_order_acquire); return; } sync(h); if (*seq != h->scal_seq) throw HipError{hipErrorUnknown, "the step's scalar block never arrived", __FILE__, __LINE__}; return; }
    if (q != hipErrorNotReady) throw HipError{q, "hipStreamQuery", __FILE__, __LINE__};
  }
}

inline BlocksDev blocks_dev(const obvi_ba_handle* h) {
  BlocksDev b;
  b.P = h->P; b.L = h->L; b.O = h->O; b.nPv = h->nPv; b.nOv = h->nOv; b.m = h->m; b.pose_row = h->d_pose_row.get(); b.obj_row = h->d_obj_row.get();
  b.obj_shared = (h->allreduce && !h->h_shared_ov.empty()) ? h->d_obj_shared.get() : nullptr; b.shared_owner = h->rank == 0 ? 1 : 0;
  b.pose_vid = h->d_pose_vid.get(); b.obj_vid = h->d_obj_vid.get(); b.point_var = h->d_point_var.get();
  b.analytic_rotation = h->reproj_variant == OBVI_REPROJECTION_ANALYTIC ? 1 : 0;
  b.od = h->od;
  b.deterministic = h->deterministic ? h->det_stride : 0;
  return b;
}
inline ReprojDev reproj_dev(const obvi_ba_handle* h) {
  ReprojDev r;
  r.n = h->n_rp; r.pose = h->d_rp_pose.get(); r.point = h->d_rp_point.get(); r.cam = h->d_rp_cam.get();
  r.pixel = h->d_rp_pixel.get(); r.sigma = h->d_rp_sigma.get(); r.active = h->d_rp_active.get();
  r.point_ptr = h->d_point_ptr.get(); r.huber = h->rp_huber; r.yrow = h->d_rp_yrow.get();
  return r;
}
inline ReprojPoseDev reproj_pose_dev(const obvi_ba_handle* h) {
  ReprojPoseDev r;
  r.n = h->n_rp; r.point = h->d_rq_point.get(); r.cam = h->d_rq_cam.get(); r.pixel = h->d_rq_pixel.get(); r.sigma = h->d_rq_sigma.get();
  r.active = h->d_rq_active.get(); r.pose_ptr = h->d_rq_pose_ptr.get(); r.huber = h->rp_huber;
  return r;
}
inline SmallFactorsDev small_dev(const obvi_ba_handle* h) {
  SmallFactorsDev s;
  s.od = h->od;
  s.n_bb = h->n_bb; s.bb_obj = h->d_bb_obj.get(); s.bb_pose = h->d_bb_pose.get(); s.bb_cam = h->d_bb_cam.get();
  s.bb_rect = h->d_bb_rect.get(); s.bb_sqrt_inf = h->d_bb_sqrt_inf.get(); s.bb_active = h->d_bb_active.get();
  s.bb_huber = h->bb_huber; s.bb_invalid = h->bb_invalid;
  s.sm_blk = h->d_sm_blk.get(); s.smt_ptr = h->d_smt_ptr.get(); s.smt_idx = h->d_smt_idx.get();
  s.bb_pairs_unique = h->bb_pairs_unique; s.bb_blk = h->d_bb_blk.get(); s.bbo_ptr = h->d_bbo_ptr.get(); s.bbo_idx = h->d_bbo_idx.get(); s.bbp_ptr = h->d_bbp_ptr.get(); s.bbp_idx = h->d_bbp_idx.get();
  s.n_sp = h->n_sp; s.sp_obj = h->d_sp_obj.get(); s.sp_mean = h->d_sp_mean.get(); s.sp_sqrt_inf = h->d_sp_sqrt_inf.get();
  s.sp_active = h->d_sp_active.get(); s.sp_huber = h->sp_huber;
  s.n_lt = h->n_lt; s.lt_obj = h->d_lt_obj.get(); s.lt_mean = h->d_lt_mean.get(); s.lt_sqrt_inf = h->d_lt_sqrt_inf.get();
  s.lt_active = h->d_lt_active.get(); s.lt_huber = h->lt_huber;
  s.n_rl = h->n_rl; s.rl_a = h->d_rl_a.get(); s.rl_b = h->d_rl_b.get(); s.rl_t = h->d_rl_t.get(); s.rl_R = h->d_rl_R.get();
  s.rl_sqrt_inf = h->d_rl_sqrt_inf.get(); s.rl_active = h->d_rl_active.get(); s.rl_huber = h->rl_huber;
  return s;
}
inline ReducedDev reduced_dev(const obvi_ba_handle* h) {
  ReducedDev r;
  r.Hdiag = h->d_Hdiag.get(); r.g = h->d_g.get(); r.scale = h->d_scale.get(); r.lam = h->d_lam.get();
  r.S = h->d_S.get(); r.rhs = h->d_rhs.get(); r.y = h->d_y.get(); r.nt = h->nt;
  r.extra = h->use_extra ? h->d_extra_c.get() : nullptr;
  return r;
}
inline PointDev point_dev(const obvi_ba_handle* h) {
  PointDev p;
  p.Ci = h->d_Ci.get(); p.u = h->d_u.get(); p.scale = h->d_scale_l.get(); p.Z = h->d_Z.get(); p.gl = h->d_gl.get(); p.lam = h->d_lam_l.get();
  p.extra = h->use_extra ? h->d_extra_l.get() : nullptr;
  return p;
}
inline CholPlan chol_plan(const obvi_ba_handle* h) {
  CholPlan c;
  c.nt = h->nt; c.nlevels = h->nlevels; c.deterministic = h->deterministic ? 1 : 0; c.fused_potrf = h->fused_potrf ? 1 : 0;
  c.lvl_k_ptr = h->h_lvl_k_ptr.data(); c.lvl_k = h->d_lvl_k.get();
  c.trsm_ptr = h->h_trsm_ptr.data(); c.trsm_ik = h->d_trsm_ik.get();
  c.upd_ptr = h->h_upd_ptr.data(); c.upd_ij = h->d_upd_ij.get(); c.upd_kptr = h->d_upd_kptr.get(); c.upd_k = h->d_upd_k.get();
  c.rh_ptr = h->h_rh_ptr.data(); c.rh_i = h->d_rh_i.get(); c.rh_kptr = h->d_rh_kptr.get(); c.rh_k = h->d_rh_k.get();
  c.col_ptr = h->d_col_ptr.get(); c.col_i = h->d_col_i.get(); c.nbw = h->nbw; c.bw_ptr = h->h_bw_ptr.data(); c.bw_kj = h->d_bw_kj.get(); c.bw_chains = h->d_bw_chains.get();
  c.row_ptr = h->d_row_ptr.get(); c.row_j = h->d_row_j.get();
  c.upd_flag = h->d_upd_flag.get(); c.job_signal = h->d_job_signal.get(); c.k_need = h->d_k_need.get(); c.pre_ptr = h->d_pre_ptr.get(); c.pre_j = h->d_pre_j.get(); c.diag_done = h->d_diag_done.get(); c.crit_upd = h->h_crit_upd.data(); c.crit_rh = h->h_crit_rh.data(); c.slices = h->h_slices.data();
  return c;
}

// Host threads of the symbolic phase (OBVI_HOST_THREADS, default: the machine's, at most 16).  fn(part, begin, end) gets
// contiguous ranges in order, so results concatenated by part are those of the sequential loop.
// The CPUs this process may actually run on: the smaller of its affinity mask and its cgroup's CPU quota (a container limited to a few CPUs still
// reports the machine's hardware_concurrency(); sixteen spinning workers on four CPUs' worth of quota are slower than one thread).
inline int usable_cpus_uncached() {
  int n = (int)std::max(1u, std::thread::hardware_concurrency());
#if defined(__linux__)
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
  auto quota = [](const char* path_quota, const char* path_period) -> double {
    // cgroup v2: "cpu.max" holds "<quota|max> <period>"; v1: two files
    FILE* f = std::fopen(path_quota, "r");
    if (!f) return 0.0;
    char a[64] = {0}; long long period = 0, q = 0;
    double out = 0.0;
    if (path_period == nullptr) {
      if (std::fscanf(f, "%63s %lld", a, &period) == 2 && std::strcmp(a, "max") != 0 && period > 0) out = std::atof(a) / (double)period;
    } else if (std::fscanf(f, "%lld", &q) == 1 && q > 0) {
      if (FILE* g = std::fopen(path_period, "r")) { if (std::fscanf(g, "%lld", &period) == 1 && period > 0) out = (double)q / (double)period; std::fclose(g); }
    }
    std::fclose(f);
    return out;
  };
  double q = quota("/sys/fs/cgroup/cpu.max", nullptr);
  if (q <= 0.0) q = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
  if (q > 0.0) n = std::min(n, std::max(1, (int)(q + 0.5)));
#endif
  return n;
}
inline int usable_cpus() {
  static const int n = usable_cpus_uncached();   // once per process: it reads two files, and host_threads() is asked several times per plan
  return n;
}
inline int host_threads() {
  const char* v = std::getenv("OBVI_HOST_THREADS");
  const int n = v ? std::atoi(v) : std::min(16, usable_cpus());
  return std::max(1, n);
}
inline HostPool& host_pool() {
  static HostPool pool(std::max(0, host_threads() - 1));   // process-wide; the calling thread is the last worker
  return pool;
}
template <class F>
inline void parallel_ranges(int64_t n, int parts, F&& fn) {
  parts = (int)std::max<int64_t>(1, std::min<int64_t>(parts, n));
  if (parts == 1) { fn(0, (int64_t)0, n); return; }
  // nothing may escape a worker (std::terminate across the C ABI): the first exception is kept and rethrown on the caller's thread
  // after every range has run
  std::exception_ptr first_error;
  std::mutex error_mutex;
  const std::function<void(int)> guarded = [&](int t) {
    try { fn(t, n * t / parts, n * (t + 1) / parts); }
    catch (...) { std::lock_guard<std::mutex> lock(error_mutex); if (!first_error) first_error = std::current_exception(); }
  };
  host_pool().run(parts, guarded);
  if (first_error) std::rethrow_exception(first_error);
}

// Every index a factor family holds must refer to a block / camera of the CURRENT upload (set_poses / set_points / set_objects /
// set_cameras may have been called again, with smaller counts, after the factors).  0, or OBVI_ERR_OUT_OF_RANGE with the message set.
inline int validate_indices(obvi_ba_handle* h) {
  const int64_t ncam = (int64_t)h->h_cams.size();
  auto bad = [&](const char* what) { return fail(h, OBVI_ERR_OUT_OF_RANGE, std::string(what) + " refer to a block that is not in the current upload (blocks / cameras were re-uploaded after the factors)"); };
  if (h->n_rp > 0 && (h->max_rp_pose >= h->P || h->max_rp_point >= h->L || h->max_rp_cam >= ncam)) return bad("reprojection factors");
  if (h->n_rp > 0 && (int64_t)h->h_point_ptr.size() != h->L + 1) return bad("reprojection factors (point count changed)");
  if (h->n_bb > 0 && (h->max_bb_obj >= h->O || h->max_bb_pose >= h->P || h->max_bb_cam >= ncam)) return bad("bounding-box factors");
  if (h->n_sp > 0 && h->max_sp_obj >= h->O) return bad("shape priors");
  if (h->n_lt > 0 && h->max_lt_obj >= h->O) return bad("long-term-map priors");
  if (h->n_rl > 0 && h->max_rl_pose >= h->P) return bad("relative-pose factors");
  for (size_t i = 0; i < h->h_pp_kind.size(); ++i) {
    const int64_t cnt = h->h_pp_kind[i] == 0 ? h->P : h->h_pp_kind[i] == 1 ? h->L : h->O;
    if ((int64_t)h->h_pp_block[i] >= cnt) return bad("parameter priors");
  }
  if (!h->h_is_shared.empty() && (int64_t)h->h_is_shared.size() != h->O) return fail(h, OBVI_ERR_OUT_OF_RANGE, "set_shared_objects: flags were given for another object count");
  if ((int64_t)h->h_pose_const.size() != h->P || (int64_t)h->h_point_const.size() != h->L || (int64_t)h->h_object_const.size() != h->O) return fail(h, OBVI_ERR_OUT_OF_RANGE, "constness flags do not match the block counts");
  return OBVI_OK;
}
template <class T>
int64_t max_index(const T* v, int64_t n) { int64_t m = -1; for (int64_t i = 0; i < n; ++i) m = std::max<int64_t>(m, (int64_t)v[i]); return m; }

// rectified corners and sqrt_inf of the bounding-box factors from the caller's corners / (cov^-1)^1/2 and the CURRENT cameras
// (bounding_box_factor.cpp:26-39: sqrt_inf = (cov^-1)^(1/2) diag(fx,fx,fy,fy); corners rectified)
inline void bake_bbox(obvi_ba_handle* h) {
  const int64_t n = h->n_bb;
  if (n == 0 || h->max_bb_cam >= (int64_t)h->h_cams.size()) return;   // validate_indices reports the latter
  std::vector<double> rect(4 * n), si(16 * n);
  for (int64_t i = 0; i < n; ++i) {
    const DevCam& c = h->h_cams[h->h_bb_cam[i]];
    const double sc[4] = {c.fx, c.fx, c.fy, c.fy};
    const double* m4 = &h->h_bb_m4[16 * i];
    const double* corners = &h->h_bb_corners[4 * i];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) si[16 * i + 4 * a + b] = m4[4 * a + b] * sc[b];
    rect[4 * i] = (corners[0] - c.cx) / c.fx; rect[4 * i + 1] = (corners[1] - c.cx) / c.fx;
    rect[4 * i + 2] = (corners[2] - c.cy) / c.fy; rect[4 * i + 3] = (corners[3] - c.cy) / c.fy;
  }
  h->d_bb_rect.upload(rect, h->stream); h->d_bb_sqrt_inf.upload(si, h->stream);
  OBVI_HIP(hipStreamSynchronize(h->stream));
}

// ---------------------------------------------------------------------------------------
// Reduced program [Ceres-doc Program::RemoveFixedBlocks], Schur pair lists, tile plan.
// ---------------------------------------------------------------------------------------
inline bool prepare_masks(obvi_ba_handle* h);
inline void prepare_plan(obvi_ba_handle* h);
// Deterministic mode: room behind the scalar block for one partial sum per workgroup of the largest grid that leaves any (ba_device.h;
// the grids are those of the launchers at the end of ba_kernels.hip, launch_det_reduce refuses a larger one).  The block is reallocated
// when the problem outgrows it -- only between API calls: every call clears the scalars before its first launch.
inline void ensure_det_slots(obvi_ba_handle* h) {
  if (!h->deterministic) return;
  const int64_t small = (h->n_bb + 3) / 4 + (h->n_sp + h->n_lt + 63) / 64 + (h->n_rl + 3) / 4, ns = h->n_bb + h->n_sp + h->n_lt + h->n_rl;
  int64_t need = std::max<int64_t>({(h->n_point_waves + 3) / 4, (h->n_long_points + 255) / 256, small, (8 * (h->P + h->O) + 255) / 256, 2048 + (h->P + h->O + 255) / 256,
                                    h->P + (ns + 255) / 256, (h->n_rp + 255) / 256, (ns + 63) / 64});
  if (need > kDetMaxStride) throw HipError{hipErrorInvalidValue, "deterministic mode: the problem needs more partial-sum slots than kDetMaxStride", __FILE__, __LINE__};
  if (need <= h->det_stride) return;
  const char* min_env = std::getenv("OBVI_DET_MIN_STRIDE");
  int64_t stride = std::max(1, min_env ? std::atoi(min_env) : 4096);   // (the tests start small to see the block grow)
  while (stride < need) stride *= 2;
  sync(h);
  h->d_scal.resize(SC_COUNT + (size_t)kDetSlots * (size_t)stride);
  h->det_stride = (int32_t)stride;
}
inline StepClear step_clear(obvi_ba_handle* h, double fixed_cost) {
  StepClear c;
  c.hdiag = h->d_Hdiag.get(); c.n_hdiag = (int64_t)h->d_Hdiag.size();
  c.g = h->d_g.get(); c.n_g = (int64_t)h->d_g.size();
  c.rhs = h->d_rhs.get(); c.n_rhs = (int64_t)h->d_rhs.size();
  c.diag_done = h->d_diag_done.get(); c.n_done = (int64_t)h->d_diag_done.size();
  c.scal = h->d_scal.get(); c.n_scal = SC_COUNT; c.fixed_slot = SC_COST_FIXED; c.fixed_cost = fixed_cost;
  c.n_max = std::max({c.n_hdiag, c.n_g, c.n_rhs, c.n_done, c.n_scal});
  c.pub_host = nullptr; c.pub_seq = 0.0;
  return c;
}
inline void record(obvi_ba_handle* h, int idx, hipStream_t on = nullptr) {
  if (h->profiling < 1) return;
  OBVI_HIP(hipEventRecord(h->ev[idx], on ? on : h->stream));
  if (idx < PH_COUNT) h->phase_on_side[idx] = on != nullptr && on != h->stream;
}
inline void record_end(obvi_ba_handle* h, int idx, hipStream_t on) { if (h->profiling >= 1) OBVI_HIP(hipEventRecord(h->ev_end[idx], on)); }

// ---- plan.cpp
void prepare(obvi_ba_handle* h);
void prepare_plan(obvi_ba_handle* h);
bool prepare_masks(obvi_ba_handle* h);
// ---- lm.cpp
void submit_step(obvi_ba_handle* h, double radius, bool first_iter, bool solve, bool keep_factor = false);

inline double scal_gmax(const obvi_ba_handle* h) { double v; std::memcpy(&v, &h->h_scal[SC_GMAX_BITS], sizeof(v)); return v; }

inline void copy_current(obvi_ba_handle* h, DevBuf<double>& dp, DevBuf<double>& dl, DevBuf<double>& dobj) {
  hipStream_t s = h->stream;
  dp.resize((size_t)6 * h->P + 1); dl.resize((size_t)3 * h->L + 1); dobj.resize((size_t)h->od * h->O + 1);
  launch_copy3(s, dp.get(), h->d_pose.get(), 6 * h->P, dl.get(), h->d_point.get(), 3 * h->L, dobj.get(), h->d_obj.get(), h->od * h->O);
}
inline void restore_from(obvi_ba_handle* h, const DevBuf<double>& dp, const DevBuf<double>& dl, const DevBuf<double>& dobj) {
  hipStream_t s = h->stream;
  launch_copy3(s, h->d_pose.get(), dp.get(), 6 * h->P, h->d_point.get(), dl.get(), 3 * h->L, h->d_obj.get(), dobj.get(), h->od * h->O);
  h->pc_valid = false;
}

inline bool check_ready(obvi_ba_handle* h) {
  if (h->h_cams.empty() && (h->n_rp > 0 || h->n_bb > 0)) return false;
  return true;
}

// 1 / std_dev^2 of every parameter prior at its parameter's place: compact reduced index for poses / objects, [L][3] for points
inline bool pt_mapped(const obvi_ba_handle* h) { return h->pt_map_applied && !h->h_pt_new_of_old.empty() && (int64_t)h->h_pt_new_of_old.size() == h->L; }
inline int64_t pt_internal(const obvi_ba_handle* h, int64_t caller_index) { return pt_mapped(h) ? (int64_t)h->h_pt_new_of_old[(size_t)caller_index] : caller_index; }
// d_point (caller order, just uploaded) -> internal order; and back into d_pt_tmp for a download
inline void points_to_internal(obvi_ba_handle* h, DevBuf<double>& buf) {
  if (!pt_mapped(h) || h->L == 0) return;
  h->d_pt_tmp.resize((size_t)3 * h->L + 1);
  obvi::launch_permute_rows3(h->stream, h->d_pt_tmp.get(), buf.get(), h->d_pt_old_of_new.get(), h->L);
  buf.swap(h->d_pt_tmp);
}
inline const double* points_in_caller_order(obvi_ba_handle* h, const DevBuf<double>& buf) {
  if (!pt_mapped(h) || h->L == 0) return buf.get();
  h->d_pt_tmp.resize((size_t)3 * h->L + 1);
  obvi::launch_permute_rows3(h->stream, h->d_pt_tmp.get(), buf.get(), h->d_pt_new_of_old.get(), h->L);
  return h->d_pt_tmp.get();
}
inline void upload_parameter_prior_diagonals(obvi_ba_handle* h) {
  if (h->h_pp_kind.empty()) return;
  std::vector<double> ec((size_t)h->m_canon + 1, 0.0), el((size_t)3 * h->L + 1, 0.0);
  std::vector<int32_t> pose_vid((size_t)h->P + 1), obj_vid((size_t)h->O + 1);
  if (h->P) h->d_pose_vid.download(pose_vid.data(), (size_t)h->P, h->stream);
  if (h->O) h->d_obj_vid.download(obj_vid.data(), (size_t)h->O, h->stream);
  sync(h);
  for (size_t i = 0; i < h->h_pp_kind.size(); ++i) {
    const double w = 1.0 / (h->h_pp_std[i] * h->h_pp_std[i]);
    const int64_t b = h->h_pp_block[i];
    if (h->h_pp_kind[i] == 0) { if (pose_vid[b] >= 0) ec[6 * (int64_t)pose_vid[b] + h->h_pp_param[i]] += w; }
    else if (h->h_pp_kind[i] == 1) el[3 * pt_internal(h, b) + h->h_pp_param[i]] += w;
    else if (obj_vid[b] >= 0) ec[6 * h->nPv + h->od * (int64_t)obj_vid[b] + h->h_pp_param[i]] += w;
  }
  h->d_extra_c.upload(ec, h->stream); h->d_extra_l.upload(el, h->stream);
  sync(h);
}

template <class T>
inline void set_mask(std::vector<uint8_t>& host, DevBuf<uint8_t>& dev, const uint8_t* mask, int64_t n, hipStream_t s, const T* perm_sorted_to_orig) {
  host.resize(n);
  for (int64_t i = 0; i < n; ++i) host[i] = mask ? (mask[perm_sorted_to_orig ? perm_sorted_to_orig[i] : i] != 0) : 1;
  dev.upload(host, s);
}


}  // namespace obvi_lib
#endif  // OBVI_BA_HANDLE_H_
