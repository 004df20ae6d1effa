// obvi_ba.cpp -- C ABI of libobvi_ba (include/obvi_ba.h): problem upload, reduced-program
// bookkeeping, symbolic tile plan of the Schur complement and the Levenberg-Marquardt loop that
// drives the gfx950 kernels.  One handle == one GPU == one HIP stream.  There is no CPU
// compute path in this library: without a HIP device obvi_ba_create fails with OBVI_ERR_NO_DEVICE.
//
// Trust-region logic: [Ceres-doc] TrustRegionMinimizer / LevenbergMarquardtStrategy /
// TrustRegionStepEvaluator with the options the reference sets at
// include/refactoring/optimization/object_pose_graph_optimizer.h:651-672 and Ceres defaults
// otherwise (Jacobi scaling, min/max LM diagonal 1e-6/1e32, min_relative_decrease 1e-3,
// max 5 consecutive non-monotonic / invalid steps, min trust-region radius 1e-32).
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

namespace {

double wall_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

enum Phase { PH_POSE_CACHE = 0, PH_POINT_PASS, PH_POSE_PASS, PH_SMALL, PH_DIAG, PH_SCHUR, PH_SCHUR_BLOCKS, PH_CHOL, PH_BACKSUB, PH_APPLY, PH_COST, PH_COUNT };
const char* kPhaseNames[PH_COUNT] = {"pose_cache", "point_pass", "pose_pass", "small_factors", "reduced_diag", "schur_window", "schur_blocks",
                                     "cholesky_solve", "point_backsub", "apply_step", "cost"};

}  // namespace

struct obvi_ba_handle {
  int device = 0;
  int reproj_variant = OBVI_REPROJECTION_AUTODIFF;   // obvi_ba_options.reprojection_variant
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
  std::vector<uint8_t> h_pose_const, h_point_const, h_object_const;
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
  int64_t live_rows = 0;                 // 6 (variable poses) + 7 (variable objects) of the current state (== m_canon right after a full plan)
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
  std::vector<int32_t> h_shared_ov;      // reduced object indices of the shared objects, in object-index order
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

namespace {

int fail(obvi_ba_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}
int hip_fail(obvi_ba_handle* h, const HipError& e) {
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

void make_cam(const double* K4, const double* e, DevCam* c) {
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

void sync(obvi_ba_handle* h) { OBVI_HIP(hipStreamSynchronize(h->stream)); h->staging.rewind(); }
// End of a function that uploaded from its caller's buffers or from local vectors: everything that went through the arena is safe
// without waiting; a copy that went straight from pageable memory is not.
void finish_upload(obvi_ba_handle* h) { if (h->staging.spilled || tl_staging != &h->staging) sync(h); }

// Waits for the scalar block of the step just submitted: polls the sequence number the device writes behind the block, and asks the
// stream now and then so that a failed launch surfaces as an error instead of a hang.
void wait_scalars(obvi_ba_handle* h) {
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

BlocksDev blocks_dev(const obvi_ba_handle* h) {
  BlocksDev b;
  b.P = h->P; b.L = h->L; b.O = h->O; b.nPv = h->nPv; b.nOv = h->nOv; b.m = h->m; b.pose_row = h->d_pose_row.get(); b.obj_row = h->d_obj_row.get();
  b.obj_shared = (h->allreduce && !h->h_shared_ov.empty()) ? h->d_obj_shared.get() : nullptr; b.shared_owner = h->rank == 0 ? 1 : 0;
  b.pose_vid = h->d_pose_vid.get(); b.obj_vid = h->d_obj_vid.get(); b.point_var = h->d_point_var.get();
  b.analytic_rotation = h->reproj_variant == OBVI_REPROJECTION_ANALYTIC ? 1 : 0;
  b.deterministic = h->deterministic ? h->det_stride : 0;
  return b;
}
ReprojDev reproj_dev(const obvi_ba_handle* h) {
  ReprojDev r;
  r.n = h->n_rp; r.pose = h->d_rp_pose.get(); r.point = h->d_rp_point.get(); r.cam = h->d_rp_cam.get();
  r.pixel = h->d_rp_pixel.get(); r.sigma = h->d_rp_sigma.get(); r.active = h->d_rp_active.get();
  r.point_ptr = h->d_point_ptr.get(); r.huber = h->rp_huber; r.yrow = h->d_rp_yrow.get();
  return r;
}
ReprojPoseDev reproj_pose_dev(const obvi_ba_handle* h) {
  ReprojPoseDev r;
  r.n = h->n_rp; r.point = h->d_rq_point.get(); r.cam = h->d_rq_cam.get(); r.pixel = h->d_rq_pixel.get(); r.sigma = h->d_rq_sigma.get();
  r.active = h->d_rq_active.get(); r.pose_ptr = h->d_rq_pose_ptr.get(); r.huber = h->rp_huber;
  return r;
}
SmallFactorsDev small_dev(const obvi_ba_handle* h) {
  SmallFactorsDev s;
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
ReducedDev reduced_dev(const obvi_ba_handle* h) {
  ReducedDev r;
  r.Hdiag = h->d_Hdiag.get(); r.g = h->d_g.get(); r.scale = h->d_scale.get(); r.lam = h->d_lam.get();
  r.S = h->d_S.get(); r.rhs = h->d_rhs.get(); r.y = h->d_y.get(); r.nt = h->nt;
  r.extra = h->use_extra ? h->d_extra_c.get() : nullptr;
  return r;
}
PointDev point_dev(const obvi_ba_handle* h) {
  PointDev p;
  p.Ci = h->d_Ci.get(); p.u = h->d_u.get(); p.scale = h->d_scale_l.get(); p.Z = h->d_Z.get(); p.gl = h->d_gl.get(); p.lam = h->d_lam_l.get();
  p.extra = h->use_extra ? h->d_extra_l.get() : nullptr;
  return p;
}
CholPlan chol_plan(const obvi_ba_handle* h) {
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
int host_threads() {
  const char* v = std::getenv("OBVI_HOST_THREADS");
  const int n = v ? std::atoi(v) : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  return std::max(1, n);
}
HostPool& host_pool() {
  static HostPool pool(std::max(0, host_threads() - 1));   // process-wide; the calling thread is the last worker
  return pool;
}
template <class F>
void parallel_ranges(int64_t n, int parts, F&& fn) {
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
int validate_indices(obvi_ba_handle* h) {
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
void bake_bbox(obvi_ba_handle* h) {
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
bool prepare_masks(obvi_ba_handle* h);
void prepare_plan(obvi_ba_handle* h);
// Deterministic mode: room behind the scalar block for one partial sum per workgroup of the largest grid that leaves any (ba_device.h;
// the grids are those of the launchers at the end of ba_kernels.hip, launch_det_reduce refuses a larger one).  The block is reallocated
// when the problem outgrows it -- only between API calls: every call clears the scalars before its first launch.
void ensure_det_slots(obvi_ba_handle* h) {
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
void prepare(obvi_ba_handle* h) {
  prepare_plan(h);
  ensure_det_slots(h);
}
void prepare_plan(obvi_ba_handle* h) {
  if (!h->dirty && !h->mask_dirty) return;
  ApiTimer api_timer_(h->dirty ? "  prepare (symbolic phase)" : "  prepare (masks only)");
  if (!h->dirty && h->mask_dirty && prepare_masks(h)) return;
  // OBVI_DEBUG_PREPARE: stage times of the symbolic phase on stderr
  const bool stage_times = std::getenv("OBVI_DEBUG_PREPARE") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto stage = [&](const char* name) {
    if (!stage_times) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "prepare: %-28s %8.2f ms\n", name, std::chrono::duration<double, std::milli>(t - t_prev).count());
    t_prev = t;
  };
  const int64_t P = h->P, L = h->L, O = h->O;
  std::vector<uint8_t> pose_used(P, 0), obj_used(O, 0), point_used(L, 0);
  int64_t nres = 0;
  for (int64_t a = 0; a < h->n_rp; ++a) {
    if (!h->h_rp_active[a]) continue;
    const uint32_t p = h->h_rp_pose[a], l = h->h_rp_point[a];
    const bool cp = h->h_pose_const[p], cl = h->h_point_const[l];
    if (cp && cl) continue;
    nres += 2;
    if (!cp) pose_used[p] = 1;
    if (!cl) point_used[l] = 1;
  }
  for (int64_t i = 0; i < h->n_bb; ++i) {
    if (!h->h_bb_active[i]) continue;
    const uint32_t o = h->h_bb_obj[i], p = h->h_bb_pose[i];
    const bool co = h->h_object_const[o], cp = h->h_pose_const[p];
    if (co && cp) continue;
    nres += 4;
    if (!co) obj_used[o] = 1;
    if (!cp) pose_used[p] = 1;
  }
  for (int64_t i = 0; i < h->n_sp; ++i) if (h->h_sp_active[i] && !h->h_object_const[h->h_sp_obj[i]]) { nres += 3; obj_used[h->h_sp_obj[i]] = 1; }
  for (int64_t i = 0; i < h->n_lt; ++i) if (h->h_lt_active[i] && !h->h_object_const[h->h_lt_obj[i]]) { nres += 7; obj_used[h->h_lt_obj[i]] = 1; }
  for (int64_t i = 0; i < h->n_rl; ++i) {
    if (!h->h_rl_active[i]) continue;
    const uint32_t a = h->h_rl_a[i], b = h->h_rl_b[i];
    const bool ca = h->h_pose_const[a], cb = h->h_pose_const[b];
    if (ca && cb) continue;
    nres += 6;
    if (!ca) pose_used[a] = 1;
    if (!cb) pose_used[b] = 1;
  }
  if (!h->h_is_shared.empty()) for (int64_t o = 0; o < O; ++o) if (h->h_is_shared[o]) obj_used[o] = 1;   // shared objects exist on every rank
  std::vector<int32_t> pose_vid(P, -1), obj_vid(O, -1);
  std::vector<uint8_t> point_var(L, 0);
  h->nPv = h->nOv = h->nLv = 0;
  std::vector<int32_t> nat(P, -1);     // rank among the variable poses in pose-index (frame) order
  for (int64_t p = 0; p < P; ++p) if (!h->h_pose_const[p] && pose_used[p]) nat[p] = (int32_t)h->nPv++;
  for (int64_t o = 0; o < O; ++o) if (!h->h_object_const[o] && obj_used[o]) obj_vid[o] = (int32_t)h->nOv++;   // provisional: index order
  for (int64_t l = 0; l < L; ++l) if (!h->h_point_const[l] && point_used[l]) { point_var[l] = 1; h->nLv++; }
  const int64_t nPv = h->nPv;

  stage("reduced program");
  // ---- elimination order: nested dissection of the frame chain, objects inside the tree.  reach[f] = largest
  //      frame rank f couples to through a shared point or an odometry factor; a separator
  //      [s0, s1) with s1 > reach of everything left of s0 decouples the two sides.  Cut positions
  //      are multiples of 32 poses (= 3 tiles) so tree nodes never share a tile.
  {
    std::vector<int32_t> reach(nPv);
    for (int64_t f = 0; f < nPv; ++f) reach[f] = (int32_t)f;
    for (int64_t l = 0; l < L; ++l) {
      if (!point_var[l]) continue;
      int32_t lo = INT32_MAX, hi = -1;
      for (uint32_t a = h->h_point_ptr[l]; a < h->h_point_ptr[l + 1]; ++a) {
        if (!h->h_rp_active[a]) continue;
        const int32_t f = nat[h->h_rp_pose[a]];
        if (f >= 0) { lo = std::min(lo, f); hi = std::max(hi, f); }
      }
      if (hi < 0) continue;
      for (uint32_t a = h->h_point_ptr[l]; a < h->h_point_ptr[l + 1]; ++a) {   // every frame of the track couples to its last one
        if (!h->h_rp_active[a]) continue;
        const int32_t f = nat[h->h_rp_pose[a]];
        if (f >= 0) reach[f] = std::max(reach[f], hi);
      }
      (void)lo;
    }
    for (int64_t i = 0; i < h->n_rl; ++i) {
      if (!h->h_rl_active[i]) continue;
      const int32_t fa = nat[h->h_rl_a[i]], fb = nat[h->h_rl_b[i]];
      if (fa >= 0 && fb >= 0) reach[std::min(fa, fb)] = std::max(reach[std::min(fa, fb)], std::max(fa, fb));
    }
    // ---- the tree: nodes in elimination (post-) order; a node owns the frames [p0,p1) (a leaf, or a separator) and
    //      covers the frame range [lo,hi) of its subtree
    struct Node { int32_t lo, hi, p0, p1, left, right; };
    std::vector<Node> nodes;
    const int32_t G = std::getenv("OBVI_ND_G") ? std::atoi(std::getenv("OBVI_ND_G")) : 4;   // cut granularity in poses (tuning knob)
    const int32_t kLeaf = std::getenv("OBVI_ND_LEAF") ? std::atoi(std::getenv("OBVI_ND_LEAF")) : 64;   // tuning knob (poses per leaf)
    const bool balance = !std::getenv("OBVI_ND_BALANCE") || std::atoi(std::getenv("OBVI_ND_BALANCE")) != 0;   // tuning knob
    const double sep_frac = std::getenv("OBVI_ND_SEPFRAC") ? std::atof(std::getenv("OBVI_ND_SEPFRAC")) : 0.5;   // tuning knob: a range is cut only if the separator is at most this part of it
    std::function<int32_t(int32_t, int32_t)> build = [&](int32_t lo, int32_t hi) -> int32_t {
      auto leaf = [&]() { nodes.push_back({lo, hi, lo, hi, -1, -1}); return (int32_t)nodes.size() - 1; };
      if (hi - lo <= kLeaf) return leaf();
      // the separator [s0, s1) is placed so that the two sides are equally long (the longer side sets the depth of the
      // elimination tree): first cut in the middle to learn the separator's width, then shift the cut left by half of it
      int32_t s0 = 0, s1 = 0;
      for (int pass = 0; pass < 2; ++pass) {
        const int32_t width = pass == 0 ? 0 : s1 - s0;
        s0 = ((lo + hi - (balance ? width : 0)) / 2 / G) * G;
        if (s0 <= lo) s0 = lo + G;
        int32_t far = s0 - 1;
        for (int32_t f = lo; f < s0; ++f) far = std::max(far, reach[f]);
        s1 = std::min<int32_t>(hi, ((far + 1 + G - 1) / G) * G);
        if (s1 <= s0) s1 = std::min<int32_t>(hi, s0 + G);
      }
      if ((double)(s1 - s0) > sep_frac * (double)(hi - lo) || s1 >= hi) return leaf();
      const int32_t l = build(lo, s0), r = build(s1, hi);
      nodes.push_back({lo, hi, s0, s1, l, r});
      return (int32_t)nodes.size() - 1;
    };
    const int32_t root = nPv > 0 ? build(0, (int32_t)nPv) : -1;
    // ---- objects: each goes to the deepest node whose subtree covers every frame that observes it (it is then
    //      eliminated together with that node); inside a node by first observing frame
    std::vector<int32_t> fa(O, INT32_MAX), fb(O, -1);
    for (int64_t i = 0; i < h->n_bb; ++i) {
      if (!h->h_bb_active[i]) continue;
      const int32_t f = nat[h->h_bb_pose[i]];
      const uint32_t o = h->h_bb_obj[i];
      if (f >= 0) { fa[o] = std::min(fa[o], f); fb[o] = std::max(fb[o], f); }
    }
    std::vector<std::vector<int64_t>> node_objs(nodes.size() + 1);   // last slot: no tree (no variable pose)
    std::vector<int64_t> tail_objs;                                   // shared across ranks: eliminated last, in object-index order
    for (int64_t o = 0; o < O; ++o) {
      if (obj_vid[o] < 0) continue;
      if (!h->h_is_shared.empty() && h->h_is_shared[o]) { tail_objs.push_back(o); continue; }
      int32_t n = root;
      if (n >= 0 && fb[o] >= 0) {
        for (;;) {
          const Node& nd = nodes[n];
          if (nd.left < 0) break;
          if (fb[o] < nd.p0) n = nd.left; else if (fa[o] >= nd.p1) n = nd.right; else break;
        }
      }
      node_objs[n >= 0 ? n : (int32_t)nodes.size()].push_back(o);
    }
    for (auto& v : node_objs) std::stable_sort(v.begin(), v.end(), [&](int64_t x, int64_t y) { return fa[x] < fa[y]; });
    // ---- rows of the tile grid: node after node, every node starts on a tile boundary
    std::vector<int32_t> pos(nPv);
    h->h_pose_row.assign(nPv, 0); h->h_obj_row.assign(h->nOv, 0);
    int64_t row = 0;
    int32_t next_pose = 0, next_obj = 0;
    std::vector<std::pair<int64_t, int64_t>> used;   // row ranges in use (the rest is padding)
    auto place_node = [&](int32_t p0, int32_t p1, const std::vector<int64_t>& objs) {
      row = ((row + kTile - 1) / kTile) * kTile;
      const int64_t start = row;
      for (int32_t f = p0; f < p1; ++f) { pos[f] = next_pose; h->h_pose_row[next_pose++] = (int32_t)row; row += 6; }
      for (int64_t o : objs) { obj_vid[o] = next_obj; h->h_obj_row[next_obj++] = (int32_t)row; row += 7; }
      if (row > start) used.push_back({start, row});
    };
    for (size_t n = 0; n < nodes.size(); ++n) {
      const int64_t r0 = row;
      place_node(nodes[n].p0, nodes[n].p1, node_objs[n]);
      if (std::getenv("OBVI_DEBUG_PLAN")) std::fprintf(stderr, "node %zu: frames [%d,%d) of subtree [%d,%d) %s objects %zu rows %lld tiles %lld\n", n, nodes[n].p0, nodes[n].p1, nodes[n].lo, nodes[n].hi,
                                                       nodes[n].left < 0 ? "leaf" : "separator", node_objs[n].size(), (long long)(row - ((r0 + kTile - 1) / kTile) * kTile), (long long)((row + kTile - 1) / kTile - (r0 + kTile - 1) / kTile));
    }
    place_node(0, 0, node_objs[nodes.size()]);
    h->tail_t0 = -1;
    h->h_shared_ov.clear();
    if (!tail_objs.empty()) {
      row = ((row + kTile - 1) / kTile) * kTile;
      h->tail_t0 = (int32_t)(row / kTile);
      place_node(0, 0, tail_objs);
      for (int64_t o : tail_objs) h->h_shared_ov.push_back(obj_vid[o]);
    }
    for (int64_t p = 0; p < P; ++p) if (nat[p] >= 0) pose_vid[p] = pos[nat[p]];
    h->h_row_of_nat.resize(nPv);
    for (int64_t f = 0; f < nPv; ++f) h->h_row_of_nat[f] = h->h_pose_row[pos[f]];
    h->m = row;
    h->nt = (int32_t)std::max<int64_t>(1, (h->m + kTile - 1) / kTile);
    h->h_is_pad.assign((size_t)h->nt * kTile, 1);
    for (const auto& u : used) for (int64_t r = u.first; r < u.second; ++r) h->h_is_pad[r] = 0;
  }
  {   // the back-substitution reads the pose step of an observation through one index instead of pose -> variable id -> row
    std::vector<int32_t>& yrow = h->h_rp_yrow;   // member: stays alive until the copy has been issued and synchronised
    yrow.resize((size_t)h->n_rp);
    for (int64_t a = 0; a < h->n_rp; ++a) {
      const int32_t v = h->h_rp_active[a] ? pose_vid[h->h_rp_pose[a]] : -1;
      yrow[a] = v >= 0 ? h->h_pose_row[v] : -1;
    }
    h->d_rp_yrow.upload(yrow, h->stream);
  }
  h->m_canon = 6 * nPv + 7 * h->nOv;
  h->h_canon_row.resize(h->m_canon);
  for (int64_t p = 0; p < P; ++p) if (nat[p] >= 0) for (int k = 0; k < 6; ++k) h->h_canon_row[6 * (int64_t)nat[p] + k] = (int64_t)h->h_pose_row[pose_vid[p]] + k;
  {
    int64_t rank = 0;   // canonical order = object index order
    for (int64_t o = 0; o < O; ++o) if (obj_vid[o] >= 0) { for (int k = 0; k < 7; ++k) h->h_canon_row[6 * nPv + 7 * rank + k] = (int64_t)h->h_obj_row[obj_vid[o]] + k; ++rank; }
  }
  h->num_params = h->m_canon + 3 * h->nLv;
  h->num_residuals = nres;
  const int32_t nt = h->nt;
  const int64_t m_pad = (int64_t)nt * kTile;

  stage("ordering");
  // ---- Schur complement work lists.  k_schur_window takes every ordered observation pair (i >= j) of a point whose
  //      frame distance is below the window's offset count; a point is visited once per row chunk that holds one of
  //      its observations.  The remaining pairs (a, b) with row(a) >= row(b) go to k_schur_blocks grouped by 6x6
  //      block.  The tile mask gets every block.
  const int32_t nt_ = h->nt;
  std::vector<uint8_t> mask((size_t)nt_ * nt_, 0);
  auto mark = [&](int64_t row, int dr, int64_t col, int dc) {
    const int t0 = (int)(row / kTile), t1 = (int)((row + dr - 1) / kTile), c0 = (int)(col / kTile), c1 = (int)((col + dc - 1) / kTile);
    for (int ti = t0; ti <= t1; ++ti) for (int tj = c0; tj <= c1; ++tj) if (ti >= tj) mask[(size_t)ti * nt_ + tj] = 1;
  };
  auto env_int = [](const char* name, int dflt) { const char* v = std::getenv(name); return v ? std::atoi(v) : dflt; };
  constexpr int32_t SR = kSchurRows, SBACK = kSchurWindowFrames - kSchurRows;
  const int64_t max_visits = std::max(8, env_int("OBVI_SCHUR_VISITS", 1 << 20));   // visits per workgroup (tuning knob)
  struct Pair { uint64_t key; uint32_t a, b; };
  std::vector<Pair> pairs;
  struct Visit { int32_t chunk; uint32_t l, beg, k; bool twin; uint64_t tiles; };
  std::vector<Visit> visit_list;
  int64_t n_window_pairs = 0;
  bool any_twin = false;
  // pose pairs that share a point: collected in a bitmap (one store per pair of sightings) and turned into tile marks once per
  // pose pair afterwards -- a point contributes k (k + 1) / 2 pairs and most of them repeat
  const bool pair_bitmap = h->nPv <= env_int("OBVI_PAIR_BITMAP_MAX", 8192);   // 64 MB at most; beyond it the tile marks are made pair by pair (tuning knob)
  std::vector<uint8_t> pose_pair(pair_bitmap ? (size_t)h->nPv * (size_t)h->nPv : 0, 0);
  {
    // points are independent: ranges of points on host threads (the bitmap is shared: every writer stores the same 1), lists joined in
    // point order.  Without the bitmap the tile marks go straight into the mask: one thread.
    const int parts = pair_bitmap ? (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), L / 256)) : 1;   // the workers exist (host_pool): a range of a few hundred points is worth handing out
    std::vector<std::vector<Pair>> pairs_t(parts);
    std::vector<std::vector<Visit>> visits_t(parts);
    // small windows: every range marks its pose pairs in a bitmap of its own (a few KB), merged afterwards -- sixteen threads storing
    // into the same forty cache lines were slower than one
    const bool private_bitmaps = pair_bitmap && parts > 1 && (size_t)h->nPv * (size_t)h->nPv <= ((size_t)1 << 18);
    std::vector<std::vector<uint8_t>> pose_pair_t(private_bitmaps ? parts : 0);
    std::vector<int64_t> window_pairs_t(parts, 0);
    std::vector<uint8_t> twin_t(parts, 0);
    parallel_ranges(L, parts, [&](int part, int64_t l0, int64_t l1) {
      struct Ob { uint32_t a; int32_t vid, f; };
      std::vector<Ob> obs;
      std::vector<int32_t> chunks;
      std::vector<Pair>& pairs = pairs_t[part];
      std::vector<Visit>& visit_list = visits_t[part];
      if (private_bitmaps) pose_pair_t[part].assign((size_t)h->nPv * (size_t)h->nPv, 0);
      uint8_t* const pose_pair_w = private_bitmaps ? pose_pair_t[part].data() : pose_pair.data();
      int64_t n_window_pairs = 0;
      bool any_twin = false;
      for (int64_t l = l0; l < l1; ++l) {
        if (!point_var[l]) continue;
        const uint32_t beg = h->h_point_ptr[l], end = h->h_point_ptr[l + 1];
        obs.clear();
        for (uint32_t a = beg; a < end; ++a) {
          if (!h->h_rp_active[a]) continue;
          const int32_t v = pose_vid[h->h_rp_pose[a]];
          if (v >= 0) obs.push_back({a, v, nat[h->h_rp_pose[a]]});
        }
        // the strip kernel takes a point unless one of its frames holds more than two observations
        // (the observations of a point are sorted by pose, hence by frame: equal frames are neighbours)
        bool windowed = true, twin = false;
        for (size_t i = 0; i < obs.size() && windowed;) {
          size_t e = i + 1;
          while (e < obs.size() && obs[e].f == obs[i].f) ++e;
          if (e - i > 2) windowed = false;
          if (e - i == 2) twin = true;
          i = e;
        }
        if (windowed) {
          chunks.clear();
          for (const Ob& x : obs) { if (chunks.empty() || chunks.back() != x.f / SR) chunks.push_back(x.f / SR); }
          for (int32_t c : chunks) {
            // frames of the strip [fbase, fbase + 48) the point covers, then the 16x16 tiles (r, c) of the 3 x 18 strip it touches
            const int32_t fbase = c * SR - SBACK;
            uint64_t m = 0, tiles = 0;
            for (const Ob& x : obs) if (x.f >= fbase && x.f < (c + 1) * SR) m |= 1ull << (x.f - fbase);
            auto frames_of_tile = [](int t0) { return ((2ull << ((16 * t0 + 15) / 6)) - 1) & ~((1ull << ((16 * t0) / 6)) - 1); };
            constexpr int kColTiles = kSchurWindowFrames * 6 / 16, kRowTiles = SR * 6 / 16, kRowTile0 = SBACK * 6 / 16;
            static_assert(kRowTiles == 3, "three row tiles per chunk (bit 3 tc + tr of a visit's tile word)");
            uint64_t rows = 0;                                                        // row tiles the point touches
            for (int tr = 0; tr < kRowTiles; ++tr) if (m & frames_of_tile(tr + kRowTile0)) rows |= 1ull << tr;
            for (int tc = 0; tc < kColTiles; ++tc) {
              if (!(m & frames_of_tile(tc))) continue;
              uint64_t allowed = 0;                                                   // lower triangle: tc <= tr + kRowTile0
              for (int tr = 0; tr < kRowTiles; ++tr) if (tc <= tr + kRowTile0) allowed |= 1ull << tr;
              tiles |= (rows & allowed) << (3 * tc);
            }
            visit_list.push_back({c, (uint32_t)l, beg, (uint32_t)(end - beg), twin, tiles});
          }
          any_twin = any_twin || twin;
        }
        // every pair lies inside the strip of its later frame's chunk iff the point's first frame lies inside the strip of its last frame
        const bool all_in_window = windowed && !obs.empty() && obs.front().f >= (obs.back().f / SR) * SR - SBACK;
        if (all_in_window) n_window_pairs += (int64_t)(obs.size() * (obs.size() + 1) / 2);
        if (all_in_window && pair_bitmap) {
          for (size_t i = 0; i < obs.size(); ++i)
            for (size_t j = 0; j <= i; ++j)
              __atomic_store_n(&pose_pair_w[(size_t)std::max(obs[i].vid, obs[j].vid) * (size_t)h->nPv + (size_t)std::min(obs[i].vid, obs[j].vid)], (uint8_t)1, __ATOMIC_RELAXED);
          continue;
        }
        if (all_in_window) n_window_pairs -= (int64_t)(obs.size() * (obs.size() + 1) / 2);   // counted pair by pair below
        for (size_t i = 0; i < obs.size(); ++i)
          for (size_t j = 0; j <= i; ++j) {
            const Ob& x = obs[i]; const Ob& y = obs[j];
            if (pair_bitmap) __atomic_store_n(&pose_pair_w[(size_t)std::max(x.vid, y.vid) * (size_t)h->nPv + (size_t)std::min(x.vid, y.vid)], (uint8_t)1, __ATOMIC_RELAXED);
            else mark(h->h_pose_row[std::max(x.vid, y.vid)], 6, h->h_pose_row[std::min(x.vid, y.vid)], 6);
            // inside the strip of the later frame's chunk?  (same test as the kernel's inverse map)
            const int32_t fp = std::max(x.f, y.f), fq = std::min(x.f, y.f);
            if (windowed && fq >= (fp / SR) * SR - SBACK) { ++n_window_pairs; continue; }
            const Ob& hi = x.vid >= y.vid ? x : y; const Ob& lo = x.vid >= y.vid ? y : x;
            pairs.push_back({(uint64_t)hi.vid * (uint64_t)(h->nPv + 1) + (uint64_t)lo.vid, hi.a, lo.a});
            if (i != j && x.vid == y.vid) pairs.push_back({(uint64_t)hi.vid * (uint64_t)(h->nPv + 1) + (uint64_t)lo.vid, lo.a, hi.a});
          }
      }
      window_pairs_t[part] = n_window_pairs; twin_t[part] = any_twin ? 1 : 0;
    });
    for (const auto& bm : pose_pair_t) for (size_t i = 0; i < bm.size(); ++i) pose_pair[i] |= bm[i];
    for (int t = 0; t < parts; ++t) {
      pairs.insert(pairs.end(), pairs_t[t].begin(), pairs_t[t].end());
      visit_list.insert(visit_list.end(), visits_t[t].begin(), visits_t[t].end());
      n_window_pairs += window_pairs_t[t]; any_twin = any_twin || twin_t[t];
      std::vector<Pair>().swap(pairs_t[t]); std::vector<Visit>().swap(visits_t[t]);
    }
  }
  if (pair_bitmap)
    for (int64_t hi = 0; hi < h->nPv; ++hi) {
      const uint8_t* row = &pose_pair[(size_t)hi * (size_t)h->nPv];
      for (int64_t lo = 0; lo <= hi; ++lo) if (row[lo]) mark(h->h_pose_row[hi], 6, h->h_pose_row[lo], 6);
    }
  stage("schur pairs / visits");
  // one work list per (chunk, column group): the visits with a tile in that group
  constexpr int kGroups = (kSchurWindowFrames * 6 / 16) / kSchurGroupCols, kGroupBits = 3 * kSchurGroupCols;
  struct GVisit { int32_t chunk, group; uint32_t l, beg, k; bool twin; uint32_t bits; };
  std::vector<GVisit> gv;
  {
    // ordered by chunk, then by group descending, visits of a list in point order: a counting sort over the (chunk, group) buckets
    int32_t max_chunk = -1;
    for (const Visit& v : visit_list) max_chunk = std::max(max_chunk, v.chunk);
    std::vector<size_t> start((size_t)(max_chunk + 1) * kGroups + 1, 0);
    auto bucket = [&](int32_t chunk, int g) { return (size_t)chunk * kGroups + (size_t)(kGroups - 1 - g); };
    auto group_bits = [&](const Visit& v, int g) { return (uint32_t)(v.tiles >> (kGroupBits * g)) & ((1u << kGroupBits) - 1u); };
    for (const Visit& v : visit_list)
      for (int g = 0; g < kGroups; ++g) if (group_bits(v, g)) ++start[bucket(v.chunk, g) + 1];
    for (size_t b2 = 1; b2 < start.size(); ++b2) start[b2] += start[b2 - 1];
    gv.resize(start.back());
    for (const Visit& v : visit_list)
      for (int g = 0; g < kGroups; ++g) {
        const uint32_t bits = group_bits(v, g);
        if (bits) gv[start[bucket(v.chunk, g)]++] = {v.chunk, g, v.l, v.beg, v.k, v.twin, bits};
      }
  }
  // slices of a work list: enough workgroups to fill the device on small problems, at most max_visits visits each
  // (deterministic mode: a work list is never cut -- one workgroup, hence one writer, per strip)
  const int64_t slice = h->deterministic ? ((int64_t)1 << 40) : std::min<int64_t>(max_visits, std::max<int64_t>(64, (int64_t)gv.size() / env_int("OBVI_SCHUR_WGS", 1536)));
  // per workgroup: batches of visits that fit the kernel's LDS buffer.  A visit is laid out as consecutive 144-byte
  // slots: one per strip frame over the range of its row frames and of its column frames in the group (source: the Z
  // record, or the zero page for a frame the point skips), the point's (u_l, 0) tail, and -- stereo -- a second layer
  // with the second record of each frame.
  const uint32_t zero16 = (uint32_t)((18ull * (uint64_t)h->n_rp + 4ull * (uint64_t)L + 4ull) / 2);   // zero page behind the Z blocks
  std::vector<uint32_t> wg_bptr(1, 0), bfirst(1, 0), bslot(1, 0), visits, slot_src;
  std::vector<int32_t> wg_f0, wg_group;
  visits.reserve(4 * gv.size());
  constexpr uint32_t kBatchSlots = kSchurBatchBytes / 144;
  auto visit_slots = [&](const GVisit& v, uint32_t base, std::vector<uint32_t>& out, uint32_t* rec) {
    // The image of a visit covers every strip frame that an ACTIVE tile of the visit touches -- row tiles that hold one of the point's
    // row frames, column tiles of the group that hold one of its column frames -- with the zero page for the frames the point does not
    // observe.  A lane's operand is then at (visit-uniform base) + (lane constant), no range test: k_schur_window.
    const int32_t fbase = v.chunk * SR - SBACK;
    constexpr int kRowTile0 = SBACK * 6 / 16;
    uint32_t rows = 0;
    int32_t A0 = INT32_MAX, A1 = -1, B0 = INT32_MAX, B1 = -1;
    for (int c = 0; c < kSchurGroupCols; ++c) {
      const uint32_t t3 = (v.bits >> (3 * c)) & 7u;
      if (!t3) continue;
      rows |= t3;
      const int t = kSchurGroupCols * v.group + c;
      B0 = std::min<int32_t>(B0, (16 * t) / 6); B1 = std::max<int32_t>(B1, (16 * t + 15) / 6);
    }
    for (int r = 0; r < 3; ++r)
      if ((rows >> r) & 1u) { const int t = kRowTile0 + r; A0 = std::min<int32_t>(A0, (16 * t) / 6); A1 = std::max<int32_t>(A1, (16 * t + 15) / 6); }
    A1 = std::min<int32_t>(A1, kSchurWindowFrames - 1); B1 = std::min<int32_t>(B1, kSchurWindowFrames - 1);
    uint32_t prim[kSchurWindowFrames], sec[kSchurWindowFrames];
    for (int i = 0; i < kSchurWindowFrames; ++i) prim[i] = sec[i] = zero16;
    for (uint32_t a = v.beg; a < v.beg + v.k; ++a) {
      if (!h->h_rp_active[a] || pose_vid[h->h_rp_pose[a]] < 0) continue;
      const int32_t fo = nat[h->h_rp_pose[a]] - fbase;
      if (fo < 0 || fo >= kSchurWindowFrames) continue;
      const uint32_t src = (uint32_t)((18ull * a + 4ull * v.l) / 2);
      if (prim[fo] == zero16) prim[fo] = src; else sec[fo] = src;
    }
    out.clear();
    int32_t slotA0, slotB0;   // slot of strip frame 0 for the row operands / the column operands (may lie before the image: only covered frames are read)
    uint32_t tail;
    const bool merged = B0 <= A1 + 1 && A0 <= B1 + 1;
    if (merged) {
      const int32_t lo = std::min(A0, B0), hi = std::max(A1, B1);
      for (int32_t fo = lo; fo <= hi; ++fo) out.push_back(prim[fo]);
      slotA0 = slotB0 = (int32_t)base - lo; tail = base + (uint32_t)(hi - lo + 1);
      out.push_back((uint32_t)((18ull * (v.beg + v.k) + 4ull * v.l) / 2));   // z_tail()
      if (v.twin) for (int32_t fo = lo; fo <= hi; ++fo) out.push_back(sec[fo]);
    } else {
      for (int32_t fo = A0; fo <= A1; ++fo) out.push_back(prim[fo]);
      slotA0 = (int32_t)base - A0; tail = base + (uint32_t)(A1 - A0 + 1); slotB0 = (int32_t)tail + 1 - B0;
      out.push_back((uint32_t)((18ull * (v.beg + v.k) + 4ull * v.l) / 2));
      for (int32_t fo = B0; fo <= B1; ++fo) out.push_back(prim[fo]);
      if (v.twin) {
        for (int32_t fo = A0; fo <= A1; ++fo) out.push_back(sec[fo]);
        out.push_back(zero16);
        for (int32_t fo = B0; fo <= B1; ++fo) out.push_back(sec[fo]);
      }
    }
    const uint32_t layer = v.twin ? (uint32_t)out.size() / 2 + (merged ? 1u : 0u) : 0u;   // slots from a record to its second-layer twin
    rec[0] = (uint32_t)(144 * slotA0);                                  // byte offset of strip frame 0 in the batch image, row operands (int32)
    rec[1] = (uint32_t)(144 * slotB0);                                  // ... column operands
    rec[2] = tail | (layer << 16);
    uint32_t cols = 0;
    for (int c = 0; c < kSchurGroupCols; ++c) if ((v.bits >> (3 * c)) & 7u) cols |= 1u << c;
    // the visit's tiles are (row tiles in use) x (column tiles in use), minus the tiles above the diagonal in the chunk's own group (a cut
    // the kernel knows at compile time): two masks instead of 15 tile bits
    rec[3] = cols | (v.twin ? 1u << 15 : 0u) | (rows << 16);
  };
  // the workgroups (slices of the work lists) are independent: ranges of them on host threads, joined in order
  struct WgRange { size_t w, we; int32_t chunk, group; };
  std::vector<WgRange> wgs;
  for (size_t q = 0; q < gv.size();) {
    size_t e = q;
    while (e < gv.size() && gv[e].chunk == gv[q].chunk && gv[e].group == gv[q].group) ++e;
    const int64_t n = (int64_t)(e - q), parts = (n + slice - 1) / slice, per = (n + parts - 1) / parts;
    for (size_t w = q; w < e; w += (size_t)per) wgs.push_back({w, std::min(e, w + (size_t)per), gv[q].chunk, gv[q].group});
    q = e;
  }
  struct BatchLists { std::vector<uint32_t> visits, slot_src, end_visit, end_slot, wg_batches; };
  const int parts2 = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), (int64_t)gv.size() / 1024));
  std::vector<BatchLists> lists_t(parts2);
  parallel_ranges((int64_t)wgs.size(), parts2, [&](int part, int64_t g0, int64_t g1) {
    BatchLists& o = lists_t[part];
    std::vector<uint32_t> vs;
    uint32_t rec[4];
    for (int64_t g = g0; g < g1; ++g) {
      uint32_t used = 0, count = 0, nb = 0;
      for (size_t t = wgs[g].w; t < wgs[g].we; ++t) {
        visit_slots(gv[t], used, vs, rec);
        if (count == (uint32_t)kSchurBatchVisits || used + (uint32_t)vs.size() > kBatchSlots) {
          o.end_visit.push_back((uint32_t)(o.visits.size() / 4)); o.end_slot.push_back((uint32_t)o.slot_src.size()); ++nb; used = 0; count = 0;
          visit_slots(gv[t], used, vs, rec);
        }
        o.visits.insert(o.visits.end(), rec, rec + 4);
        o.slot_src.insert(o.slot_src.end(), vs.begin(), vs.end());
        used += (uint32_t)vs.size(); ++count;
      }
      o.end_visit.push_back((uint32_t)(o.visits.size() / 4)); o.end_slot.push_back((uint32_t)o.slot_src.size()); ++nb;
      o.wg_batches.push_back(nb);
    }
  });
  for (const BatchLists& o : lists_t) {
    const uint32_t voff = (uint32_t)(visits.size() / 4), soff = (uint32_t)slot_src.size();
    visits.insert(visits.end(), o.visits.begin(), o.visits.end());
    slot_src.insert(slot_src.end(), o.slot_src.begin(), o.slot_src.end());
    for (size_t b = 0; b < o.end_visit.size(); ++b) { bfirst.push_back(voff + o.end_visit[b]); bslot.push_back(soff + o.end_slot[b]); }
    for (uint32_t nb : o.wg_batches) wg_bptr.push_back(wg_bptr.back() + nb);
  }
  for (const WgRange& g : wgs) { wg_f0.push_back(g.chunk * SR); wg_group.push_back(g.group); }
  h->schur_twins = any_twin ? 1 : 0;
  h->nchunks = (int64_t)wg_f0.size();
  h->npairs_window = n_window_pairs;
  std::sort(pairs.begin(), pairs.end(), [](const Pair& x, const Pair& y) { return x.key < y.key || (x.key == y.key && (x.a < y.a || (x.a == y.a && x.b < y.b))); });
  std::vector<uint32_t> blk_row, blk_col, blk_ptr, pair_a(pairs.size()), pair_b(pairs.size());
  for (size_t k = 0; k < pairs.size(); ++k) {
    // a block's pairs are cut into work items of at most kPairsPerItem (k_schur_blocks adds its sums atomically): a few long tracks in a
    // small window would otherwise leave one workgroup with thousands of pairs on the critical path
    const size_t kPairsPerItem = h->deterministic ? ((size_t)1 << 40) : 256;   // deterministic mode: one work item, hence one writer, per block
    if (k == 0 || pairs[k].key != pairs[k - 1].key || k - blk_ptr.back() >= kPairsPerItem) {
      blk_row.push_back((uint32_t)h->h_pose_row[pairs[k].key / (uint64_t)(h->nPv + 1)]);
      blk_col.push_back((uint32_t)h->h_pose_row[pairs[k].key % (uint64_t)(h->nPv + 1)]);
      blk_ptr.push_back((uint32_t)k);
    }
    pair_a[k] = pairs[k].a; pair_b[k] = pairs[k].b;
  }
  blk_ptr.push_back((uint32_t)pairs.size());
  h->nblk = (int64_t)blk_row.size();
  h->npairs = (int64_t)pairs.size();
  pairs.clear(); pairs.shrink_to_fit();

  if (stage_times)
    std::fprintf(stderr, "  schur plan: %zu visits, %zu workgroups, %zu batches (%.1f slots each), %zu slots = %.1f MB gathered per launch\n", gv.size(), wgs.size(), bslot.size() - 1,
                 (double)slot_src.size() / std::max<size_t>(1, bslot.size() - 1), slot_src.size(), 144e-6 * (double)slot_src.size());
  stage("schur batches");
  // ---- tile mask of the reduced matrix (lower triangle) and symbolic fill ----
  for (int k = 0; k < nt; ++k) mask[(size_t)k * nt + k] = 1;
  for (int64_t i = 0; i < h->n_bb; ++i) {
    if (!h->h_bb_active[i]) continue;
    const int32_t ov = obj_vid[h->h_bb_obj[i]], pv = pose_vid[h->h_bb_pose[i]];
    if (ov >= 0 && pv >= 0) { const int64_t ro = h->h_obj_row[ov], rp = h->h_pose_row[pv]; if (ro > rp) mark(ro, 7, rp, 6); else mark(rp, 6, ro, 7); }
  }
  for (int64_t i = 0; i < h->n_rl; ++i) {
    if (!h->h_rl_active[i]) continue;
    const int32_t va = pose_vid[h->h_rl_a[i]], vb = pose_vid[h->h_rl_b[i]];
    if (va >= 0 && vb >= 0 && va != vb) mark(h->h_pose_row[std::max(va, vb)], 6, h->h_pose_row[std::min(va, vb)], 6);
  }
  // object diagonal blocks may straddle tiles
  for (int64_t w = 0; w < h->nOv; ++w) mark(h->h_obj_row[w], 7, h->h_obj_row[w], 7);
  for (int64_t v = 0; v < nPv; ++v) mark(h->h_pose_row[v], 6, h->h_pose_row[v], 6);
  // the shared tail is exchanged across ranks as a dense lower-triangular block of tiles
  if (h->tail_t0 >= 0) for (int i = h->tail_t0; i < nt; ++i) for (int j = h->tail_t0; j <= i; ++j) mask[(size_t)i * nt + j] = 1;
  // symbolic fill (tile columns in increasing order) + column structure of L
  std::vector<int32_t> col_ptr(nt + 1, 0), col_i;
  for (int k = 0; k < nt; ++k) {
    const size_t beg = col_i.size();
    for (int i = k + 1; i < nt; ++i) if (mask[(size_t)i * nt + k]) col_i.push_back(i);
    for (size_t x = beg; x < col_i.size(); ++x) for (size_t y = beg; y <= x; ++y) mask[(size_t)col_i[x] * nt + col_i[y]] = 1;
    col_ptr[k + 1] = (int32_t)col_i.size();
  }
  stage("tile mask + fill");
  // levels of the tile elimination tree: k depends on every j < k with L(k,j) != 0
  std::vector<int32_t> level(nt, 0);
  int32_t nlev = 0;
  for (int k = 0; k < nt; ++k) {
    int32_t lv = 0;
    for (int j = 0; j < k; ++j) if (mask[(size_t)k * nt + j]) lv = std::max(lv, level[j] + 1);
    level[k] = lv; nlev = std::max(nlev, lv + 1);
  }
  h->tail_level0 = -1;
  if (h->tail_t0 >= 0) {
    // the shared tail is factorised after the multi-GPU exchange: its tile columns get their own, last levels
    int32_t base = 0;
    for (int k = 0; k < h->tail_t0; ++k) base = std::max(base, level[k] + 1);
    for (int k = h->tail_t0; k < nt; ++k) level[k] = base + (k - h->tail_t0);
    h->tail_level0 = base;
    nlev = base + (nt - h->tail_t0);
  }
  h->nlevels = nlev;
  std::vector<std::vector<int32_t>> by_level(nlev);
  for (int k = 0; k < nt; ++k) by_level[level[k]].push_back(k);
  std::vector<int32_t> lvl_k, trsm_ik, upd_ij, upd_kptr(1, 0), upd_k, rh_i, rh_kptr(1, 0), rh_k, job_signal, k_need_of(nt, 0);
  std::vector<uint8_t> upd_flag;
  const int kUpdChunk = h->deterministic ? (1 << 30) : std::max(1, env_int("OBVI_UPD_CHUNK", 2));   // products per update job (tuning knob; deterministic mode: a target's products are never split over jobs that would meet in atomics)
  const int64_t env_slice_max = std::getenv("OBVI_SLICE_MAX") ? std::atoi(std::getenv("OBVI_SLICE_MAX")) : 512;   // tuning knob
  // a potrf workgroup applies up to this many products of the previous level to its own diagonal tile (tuning knob)
  const size_t pre_max = (size_t)std::max(0, env_int("OBVI_PRE_MAX", 2));
  std::vector<std::vector<int32_t>> pre_of(nt);
  h->h_lvl_k_ptr.assign(nlev + 1, 0); h->h_trsm_ptr.assign(nlev + 1, 0); h->h_upd_ptr.assign(nlev + 1, 0); h->h_rh_ptr.assign(nlev + 1, 0); h->h_crit_upd.assign(nlev + 1, 0); h->h_crit_rh.assign(nlev + 1, 0); h->h_slices.assign(nlev + 1, 1);
  double flops = 0.0;
  const double t3 = (double)kTile * kTile * kTile;
  struct Trip { int32_t i, j, k; };
  std::vector<Trip> trips;
  std::vector<std::pair<int32_t, int32_t>> ik;
  int64_t n_products = 0;
  std::vector<size_t> trsm_level_begin;
  for (int l = 0; l < nlev; ++l) {
    trips.clear(); ik.clear();
    const size_t trsm_begin_of_level = trsm_ik.size() / 2;
    for (int32_t k : by_level[l]) {
      lvl_k.push_back(k);
      const int32_t b0 = col_ptr[k], b1 = col_ptr[k + 1];
      for (int32_t x = b0; x < b1; ++x) {
        trsm_ik.push_back(col_i[x]); trsm_ik.push_back(k);
        ik.push_back({col_i[x], k});
        for (int32_t y = b0; y <= x; ++y) trips.push_back({col_i[x], col_i[y], k});
      }
      const double nr = (double)(b1 - b0);
      flops += t3 / 3.0 + t3 * nr + 2.0 * t3 * (nr * (nr + 1) / 2);
    }
    trsm_level_begin.push_back(trsm_begin_of_level);
    std::sort(trips.begin(), trips.end(), [](const Trip& a, const Trip& b) { return a.i != b.i ? a.i < b.i : (a.j != b.j ? a.j < b.j : a.k < b.k); });
    // one job per target tile; a k-list longer than kUpdChunk is split over several jobs that accumulate atomically.
    // Jobs that finish the diagonal tile / right-hand-side block of a column of the next level come first and signal it
    // (k_update_potrf): that column's potrf starts while the rest of this level's updates are still running.
    struct Job { int32_t i, j; uint8_t flag; size_t t0, t1; bool crit; };
    std::vector<Job> jobs;
    for (size_t q = 0; q < trips.size();) {
      size_t e = q;
      while (e < trips.size() && trips[e].i == trips[q].i && trips[e].j == trips[q].j) ++e;
      const size_t len = e - q;
      const bool crit = trips[q].i == trips[q].j && level[trips[q].i] == l + 1;
      if (crit && len <= pre_max && l + 1 != h->tail_level0) {   // applied by the column's potrf workgroup itself (also its right-hand-side block)
        for (size_t t = q; t < e; ++t) pre_of[trips[q].i].push_back(trips[t].k);
        q = e;
        continue;
      }
      const size_t chunk = h->deterministic ? len : crit ? 1 : (size_t)kUpdChunk;   // the next level waits for the critical ones: one product per job
      const uint8_t flag = len > chunk ? 1 : 0;
      for (size_t c0 = q; c0 < e; c0 += chunk) jobs.push_back({trips[q].i, trips[q].j, flag, c0, std::min(e, c0 + chunk), crit});
      q = e;
    }
    std::stable_partition(jobs.begin(), jobs.end(), [](const Job& x) { return x.crit; });
    const int32_t sl = (int64_t)jobs.size() + (int64_t)ik.size() <= env_slice_max ? 4 : 1;   // thin level: the device is mostly idle, split every tile product
    h->h_slices[l] = sl;
    h->h_crit_upd[l] = (int32_t)std::count_if(jobs.begin(), jobs.end(), [](const Job& x) { return x.crit; });
    // XCD placement on the wide levels.  Block b is observed to run on XCD b % 8, each XCD with its own L2; the tiles L_ik of a column
    // are written by the level's trsm jobs and read by its update jobs a launch later, and across XCDs such a read goes through the
    // fabric.  Columns of one level are independent, so every column gets the XCD its own potrf ran on (which wrote L_kk^-1), its trsm jobs take
    // block indices with that residue and so do its update jobs (after the launch's leading critical jobs and potrf workgroups):
    // operands then come out of the L2 they were written to.  Queues that run dry are filled from the others (a speed matter only).
    static const bool xcd_place = env_int("OBVI_CHOL_XCD", 1) != 0;   // tuning knob
    std::vector<int32_t> xcd_of(nt, 0);
    {   // ... the XCD its potrf ran on: workgroup (leading critical jobs of the previous level's launch + rank) of that launch
      int32_t r = l > 0 ? h->h_slices[l - 1] * h->h_crit_upd[l - 1] + h->h_crit_rh[l - 1] : 0;
      for (int32_t k : by_level[l]) xcd_of[k] = (r++) % 8;
    }
    auto interleave = [&](auto& items, size_t first, int start_residue, auto&& xcd_of_item) {
      if (!xcd_place || sl != 1 || items.size() - first < 64) return;
      typedef typename std::decay<decltype(items)>::type Vec;
      std::vector<Vec> q(8);
      for (size_t x = first; x < items.size(); ++x) q[xcd_of_item(items[x])].push_back(items[x]);
      size_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0}, out = first;
      int res = start_residue;
      while (out < items.size()) {
        int pick = res;
        for (int t = 0; t < 8 && pos[pick] >= q[pick].size(); ++t) pick = (pick + 1) % 8;   // a dry queue: the next one that still has jobs
        items[out++] = q[pick][pos[pick]++];
        res = (res + 1) % 8;
      }
    };
    {
      const int32_t npk_next = l + 1 < nlev ? (int32_t)by_level[l + 1].size() : 0;
      // crit jobs come first, crit right-hand sides are not known yet at this point: they are few (<= columns of the next level) and only shift the residue on levels that have them
      interleave(jobs, (size_t)h->h_crit_upd[l], (int)((h->h_crit_upd[l] + npk_next) % 8), [&](const Job& jb) { return xcd_of[trips[jb.t0].k]; });
      std::vector<std::pair<int32_t, int32_t>> tj;
      for (size_t x = trsm_level_begin.back(); x < trsm_ik.size() / 2; ++x) tj.push_back({trsm_ik[2 * x], trsm_ik[2 * x + 1]});
      interleave(tj, 0, 0, [&](const std::pair<int32_t, int32_t>& e) { return xcd_of[e.second]; });
      for (size_t x = 0; x < tj.size(); ++x) { trsm_ik[2 * (trsm_level_begin.back() + x)] = tj[x].first; trsm_ik[2 * (trsm_level_begin.back() + x) + 1] = tj[x].second; }
    }
    h->h_crit_rh[l] = 0;
    for (const Job& jb : jobs) {
      upd_ij.push_back(jb.i); upd_ij.push_back(jb.j); upd_flag.push_back(jb.flag);
      for (size_t t = jb.t0; t < jb.t1; ++t) upd_k.push_back(trips[t].k);
      upd_kptr.push_back((int32_t)upd_k.size());
      job_signal.push_back(jb.crit ? jb.i : -1);
      if (jb.crit) k_need_of[jb.i] += sl;
    }
    n_products += (int64_t)trips.size();
    std::sort(ik.begin(), ik.end());
    std::stable_sort(ik.begin(), ik.end(), [&](const std::pair<int32_t, int32_t>& x, const std::pair<int32_t, int32_t>& y) { return (level[x.first] == l + 1) > (level[y.first] == l + 1); });
    ik.erase(std::remove_if(ik.begin(), ik.end(), [&](const std::pair<int32_t, int32_t>& x) { return level[x.first] == l + 1 && !pre_of[x.first].empty(); }), ik.end());
    for (size_t q = 0; q < ik.size(); ++q) {
      if (q == 0 || ik[q].first != ik[q - 1].first) {
        if (q != 0) rh_kptr.push_back((int32_t)rh_k.size());
        rh_i.push_back(ik[q].first);
        const bool crit = level[ik[q].first] == l + 1;
        job_signal.push_back(crit ? ik[q].first : -1);
        if (crit) { k_need_of[ik[q].first]++; h->h_crit_rh[l]++; }
      }
      rh_k.push_back(ik[q].second);
    }
    if (!ik.empty()) rh_kptr.push_back((int32_t)rh_k.size());
    h->h_lvl_k_ptr[l + 1] = (int32_t)lvl_k.size();
    h->h_trsm_ptr[l + 1] = (int32_t)(trsm_ik.size() / 2);
    h->h_upd_ptr[l + 1] = (int32_t)(upd_ij.size() / 2);
    h->h_rh_ptr[l + 1] = (int32_t)rh_i.size();
    if (std::getenv("OBVI_DEBUG_PLAN")) std::fprintf(stderr, "level %d: columns %zu (first %d) trsm %zu update jobs %zu (critical %d) products %zu slices %d\n", l, by_level[l].size(), by_level[l].empty() ? -1 : by_level[l][0], ik.size(), jobs.size(), h->h_crit_upd[l], trips.size(), sl);
  }
  stage("level jobs");
  // backward substitution, row oriented: one workgroup per tile of L.  A launch takes kBwLevels consecutive levels: a row forms the
  // y of its ancestors inside the launch itself (k_backward; chain record per row) and tiles between two rows of a launch get no
  // workgroup.  The launches are listed in the order of the forward levels and run last to first.
  std::vector<int32_t> bw_kj, bw_chains;
  // levels per launch (tuning knob; 1: one level per launch; chains of at most 7): four, or the whole tree when it has at most eight levels
  // (a sliding window: one launch instead of two)
  const int bw_levels = std::max(1, std::min(8, env_int("OBVI_BACKWARD_LEVELS", nlev <= 8 ? 8 : 4)));
  h->h_bw_ptr.assign(1, 0);
  {
    int top = nlev - 1;             // the levels are grouped from the top
    std::vector<std::pair<int, int>> groups;   // (lowest level, highest level)
    while (top >= 0) { const int lo = std::max(0, top - (bw_levels - 1)); groups.push_back({lo, top}); top = lo - 1; }
    std::reverse(groups.begin(), groups.end());
    std::vector<int32_t> chain;
    for (const auto& g : groups) {
      for (int l = g.first; l <= g.second; ++l)
        for (int32_t k : by_level[l]) {
          chain.clear();   // ancestors of k in the elimination tree (parent = first row of the column) that belong to the launch
          for (int32_t a = k; col_ptr[a] < col_ptr[a + 1] && level[col_i[col_ptr[a]]] <= g.second;) { a = col_i[col_ptr[a]]; chain.push_back(a); }
          std::reverse(chain.begin(), chain.end());   // top first
          const int32_t off = (int32_t)bw_chains.size(), n = (int32_t)chain.size();
          uint64_t bits = 0;
          for (int st = 1; st <= n; ++st) {
            const int32_t m = st < n ? chain[st] : k;
            for (int u = 0; u < st; ++u) if (mask[(size_t)chain[u] * nt + m]) bits |= 1ull << (8 * st + u);
          }
          bw_chains.push_back(n);
          bw_chains.insert(bw_chains.end(), chain.begin(), chain.end());
          bw_chains.push_back((int32_t)(uint32_t)(bits & 0xffffffffull)); bw_chains.push_back((int32_t)(uint32_t)(bits >> 32));
          bw_kj.push_back(k); bw_kj.push_back(-1); bw_kj.push_back(off);
          for (int j = 0; j < k; ++j) {
            if (!mask[(size_t)k * nt + j] || level[j] >= g.first) continue;   // a row of the same launch takes this tile's contribution itself
            bw_kj.push_back(k); bw_kj.push_back(j); bw_kj.push_back(off);
          }
        }
      h->h_bw_ptr.push_back((int32_t)(bw_kj.size() / 3));
    }
    h->nbw = (int32_t)groups.size();
  }
  {   // row structure of L (forward substitution with many right-hand sides: covariance extraction)
    std::vector<int32_t> row_ptr(nt + 1, 0), row_j;
    for (int k = 0; k < nt; ++k) {
      for (int j = 0; j < k; ++j) if (mask[(size_t)k * nt + j]) row_j.push_back(j);
      row_ptr[k + 1] = (int32_t)row_j.size();
    }
    if (row_j.empty()) row_j.push_back(0);
    h->d_row_ptr.upload(row_ptr, h->stream); h->d_row_j.upload(row_j, h->stream);
    // levels whose rows are long (separators near the root) spread a row over several workgroups: about 8 tiles each, at most 16
    const int row_tiles = std::max(1, std::getenv("OBVI_COV_ROW_TILES") ? std::atoi(std::getenv("OBVI_COV_ROW_TILES")) : 8);   // tuning knob
    h->h_row_split.assign(nlev, 1);
    for (int l = 0; l < nlev; ++l) {
      int longest = 0;
      for (int32_t k : by_level[l]) longest = std::max(longest, row_ptr[k + 1] - row_ptr[k]);
      h->h_row_split[l] = h->deterministic ? 1 : std::min(16, std::max(1, longest / row_tiles));   // (split rows meet in atomics)
    }
  }
  h->chol_flops = flops;
  h->n_trsm_jobs = (int64_t)(trsm_ik.size() / 2);
  h->n_upd_products = n_products;
  std::vector<int32_t> tiles;
  for (int i = 0; i < nt; ++i) for (int j = 0; j <= i; ++j) if (mask[(size_t)i * nt + j]) { tiles.push_back(i); tiles.push_back(j); }
  h->ntiles = (int32_t)(tiles.size() / 2);

  stage("lists");
  // ---- upload ----
  hipStream_t s = h->stream;
  h->d_pose_vid.upload(pose_vid, s); h->d_obj_vid.upload(obj_vid, s); h->d_point_var.upload(point_var, s);
  h->h_obj_vid = obj_vid;
  h->d_blk_row.upload(blk_row, s); h->d_blk_col.upload(blk_col, s); h->d_blk_ptr.upload(blk_ptr, s);
  h->d_pair_a.upload(pair_a, s); h->d_pair_b.upload(pair_b, s);
  h->d_chunk_ptr.upload(wg_bptr, s); h->d_batch_first.upload(bfirst, s); h->d_batch_slot.upload(bslot, s); h->d_chunk_points.upload(visits, s); h->d_chunk_f0.upload(wg_f0, s); h->d_chunk_group.upload(wg_group, s);
  h->d_slot_src.upload(slot_src, s); h->d_row_of_nat.upload(h->h_row_of_nat, s);
  h->d_tiles.upload(tiles, s); h->d_lvl_k.upload(lvl_k, s); h->d_trsm_ik.upload(trsm_ik, s);
  h->d_upd_ij.upload(upd_ij, s); h->d_upd_kptr.upload(upd_kptr, s); h->d_upd_k.upload(upd_k, s);
  h->d_rh_i.upload(rh_i, s); h->d_rh_kptr.upload(rh_kptr, s); h->d_rh_k.upload(rh_k, s);
  h->d_col_ptr.upload(col_ptr, s); h->d_col_i.upload(col_i, s); h->d_bw_kj.upload(bw_kj, s); h->d_bw_chains.upload(bw_chains, s); h->d_upd_flag.upload(upd_flag, s);
  {
    std::vector<int32_t> k_need(lvl_k.size());
    for (size_t x = 0; x < lvl_k.size(); ++x) k_need[x] = k_need_of[lvl_k[x]];
    std::vector<int32_t> pre_ptr(lvl_k.size() + 1, 0), pre_j;
    for (size_t x = 0; x < lvl_k.size(); ++x) { pre_j.insert(pre_j.end(), pre_of[lvl_k[x]].begin(), pre_of[lvl_k[x]].end()); pre_ptr[x + 1] = (int32_t)pre_j.size(); }
    if (pre_j.empty()) pre_j.push_back(0);
    h->d_pre_ptr.upload(pre_ptr, s); h->d_pre_j.upload(pre_j, s);
    h->d_job_signal.upload(job_signal, s); h->d_k_need.upload(k_need, s); h->d_diag_done.resize((size_t)nt + 1);
  }
  h->d_pose_row.upload(h->h_pose_row, s); h->d_obj_row.upload(h->h_obj_row, s); h->d_is_pad.upload(h->h_is_pad, s);
  {   // bounding-box factors by object and by pose (counting sorts, caller order inside a list), scratch for their blocks
    std::vector<uint32_t> optr((size_t)O + 1, 0), pptr((size_t)P + 1, 0), oidx((size_t)h->n_bb), pidx((size_t)h->n_bb);
    for (int64_t i = 0; i < h->n_bb; ++i) { optr[h->h_bb_obj[i] + 1]++; pptr[h->h_bb_pose[i] + 1]++; }
    for (int64_t o = 0; o < O; ++o) optr[o + 1] += optr[o];
    for (int64_t p = 0; p < P; ++p) pptr[p + 1] += pptr[p];
    std::vector<uint32_t> oc(optr.begin(), optr.end() - 1), pc(pptr.begin(), pptr.end() - 1);
    for (int64_t i = 0; i < h->n_bb; ++i) { oidx[oc[h->h_bb_obj[i]]++] = (uint32_t)i; pidx[pc[h->h_bb_pose[i]]++] = (uint32_t)i; }
    h->d_bbo_ptr.upload(optr, s); h->d_bbo_idx.upload(oidx, s); h->d_bbp_ptr.upload(pptr, s); h->d_bbp_idx.upload(pidx, s);
    h->d_bb_blk.resize((size_t)62 * (size_t)h->n_bb + 1);
    if (h->deterministic) {
      // priors and relative-pose factors by target block (objects, then poses), in factor order: entry = 2 slot + side
      const int64_t nsl = h->n_sp + h->n_lt + h->n_rl;
      std::vector<uint32_t> tptr((size_t)O + (size_t)P + 1, 0), tidx;
      for (int64_t i = 0; i < h->n_sp; ++i) tptr[h->h_sp_obj[i] + 1]++;
      for (int64_t i = 0; i < h->n_lt; ++i) tptr[h->h_lt_obj[i] + 1]++;
      for (int64_t i = 0; i < h->n_rl; ++i) { tptr[O + h->h_rl_a[i] + 1]++; tptr[O + h->h_rl_b[i] + 1]++; }
      for (size_t t = 0; t + 1 < tptr.size(); ++t) tptr[t + 1] += tptr[t];
      tidx.resize(tptr.back() + 1);
      std::vector<uint32_t> cur(tptr.begin(), tptr.end() - 1);
      for (int64_t i = 0; i < h->n_sp; ++i) tidx[cur[h->h_sp_obj[i]]++] = (uint32_t)(2 * i);
      for (int64_t i = 0; i < h->n_lt; ++i) tidx[cur[h->h_lt_obj[i]]++] = (uint32_t)(2 * (h->n_sp + i));
      for (int64_t i = 0; i < h->n_rl; ++i) {
        tidx[cur[O + h->h_rl_a[i]]++] = (uint32_t)(2 * (h->n_sp + h->n_lt + i));
        tidx[cur[O + h->h_rl_b[i]]++] = (uint32_t)(2 * (h->n_sp + h->n_lt + i) + 1);
      }
      h->d_smt_ptr.upload(tptr, s); h->d_smt_idx.upload(tidx, s);
      h->d_sm_blk.resize((size_t)62 * (size_t)nsl + 1);
    }
    // does any (object, pose) pair occur twice?  (inside a pose's list: the same object twice)
    h->bb_pairs_unique = 1;
    std::vector<uint32_t> objs;
    for (int64_t p = 0; p < P && h->bb_pairs_unique; ++p) {
      objs.clear();
      for (uint32_t q = pptr[p]; q < pptr[p + 1]; ++q) objs.push_back(h->h_bb_obj[pidx[q]]);
      std::sort(objs.begin(), objs.end());
      if (std::adjacent_find(objs.begin(), objs.end()) != objs.end()) h->bb_pairs_unique = 0;
    }
  }
  {
    std::vector<uint8_t> sh((size_t)h->nOv + 1, 0);
    for (int32_t ov : h->h_shared_ov) sh[ov] = 1;
    h->d_obj_shared.upload(sh, s); h->d_shared_ov.upload(h->h_shared_ov, s);
    const int64_t ntail = h->tail_t0 >= 0 ? nt - h->tail_t0 : 0;
    h->d_xbuf.resize((size_t)std::max<int64_t>(56 * (int64_t)h->h_shared_ov.size(), ntail * (ntail + 1) / 2 * kTile * kTile + ntail * kTile) + 64 + (size_t)h->world);
    h->d_xbuf2.resize((size_t)(56 * (int64_t)h->h_shared_ov.size()) + 64);
  }
  h->d_Hdiag.resize((size_t)(36 * h->nPv + 49 * h->nOv + 1));
  h->d_g.resize((size_t)h->m_canon + 1); h->d_scale.resize((size_t)h->m_canon + 1); h->d_lam.resize((size_t)h->m_canon + 1);
  h->d_S.resize((size_t)nt * nt * kTile * kTile);
  h->d_Linv.resize((size_t)nt * kTile * kTile);
  h->d_rhs.resize((size_t)m_pad); h->d_y.resize((size_t)m_pad);
  h->d_Ci.resize((size_t)6 * L + 1); h->d_u.resize((size_t)3 * L + 1); h->d_scale_l.resize((size_t)3 * L + 1); h->d_gl.resize((size_t)3 * L + 1); h->d_lam_l.resize((size_t)3 * L + 1);
  {   // z_off(): 18 per observation + (u_l, 0) per point, then a zero page (k_schur_window's source for frames a point skips)
    const size_t zdata = (size_t)18 * h->n_rp + 4 * (size_t)h->L + 4;
    h->d_Z.resize(zdata + 36);
    OBVI_HIP(hipMemsetAsync(h->d_Z.get() + zdata, 0, 36 * sizeof(double), s));
  }
  h->d_pose_c.resize((size_t)6 * P + 1); h->d_point_c.resize((size_t)3 * L + 1); h->d_obj_c.resize((size_t)7 * O + 1);
  h->d_pose_b.resize((size_t)6 * P + 1); h->d_point_b.resize((size_t)3 * L + 1); h->d_obj_b.resize((size_t)7 * O + 1);
  h->d_pc.resize(2 * ((size_t)P + 1)); h->d_pc_c.resize(2 * ((size_t)P + 1));   // records, then the field-major copy (k_pose_cache)
  finish_upload(h);  // host vectors above go out of scope
  stage("upload + allocations");
  h->dirty = false; h->mask_dirty = false; h->pc_valid = false; h->tiles_cleared = false;
  h->plan_pose_vid = pose_vid; h->plan_obj_vid = obj_vid; h->plan_point_var = point_var; h->plan_is_pad = h->h_is_pad;
  h->plan_rp_active = h->h_rp_active; h->plan_bb_active = h->h_bb_active; h->plan_sp_active = h->h_sp_active; h->plan_lt_active = h->h_lt_active; h->plan_rl_active = h->h_rl_active;
  h->live_rows = h->m_canon;
}

// Factor masks changed and nothing else (phase II of a window: offline_problem_runner.h:803-892 re-solves the phase-I problem minus
// the excluded factors).  If the active factors and the blocks they leave variable are subsets of what the plan was built for, the
// plan stays: elimination order, Schur work lists, tile structure and level jobs are those of a superset problem, masked
// observations write zero Z records, points that lost all their factors are skipped (their records are zeroed too), and the rows
// of a pose / object that dropped out become padding rows (identity).  Only the reduced-program bookkeeping is redone: O(factors).
// Returns false when the new state is not a subset (the caller then rebuilds the plan).
bool prepare_masks(obvi_ba_handle* h) {
  static const bool keep = !std::getenv("OBVI_KEEP_PLAN") || std::atoi(std::getenv("OBVI_KEEP_PLAN")) != 0;   // 0: always rebuild (parity runs)
  if (!keep) return false;
  const int64_t P = h->P, L = h->L, O = h->O;
  if ((int64_t)h->plan_pose_vid.size() != P || (int64_t)h->plan_obj_vid.size() != O || (int64_t)h->plan_point_var.size() != L) return false;
  auto subset = [](const std::vector<uint8_t>& now, const std::vector<uint8_t>& plan) {
    if (now.size() != plan.size()) return false;
    for (size_t i = 0; i < now.size(); ++i) if (now[i] && !plan[i]) return false;
    return true;
  };
  if (h->h_rp_active.size() != h->plan_rp_active.size() || !subset(h->h_bb_active, h->plan_bb_active) || !subset(h->h_sp_active, h->plan_sp_active) ||
      !subset(h->h_lt_active, h->plan_lt_active) || !subset(h->h_rl_active, h->plan_rl_active)) return false;
  // (scratch kept between calls: a session runs this once per frame)
  std::vector<uint8_t>& pose_used = h->scr_pose_used; std::vector<uint8_t>& obj_used = h->scr_obj_used; std::vector<uint8_t>& point_used = h->scr_point_used;
  pose_used.assign(P, 0); obj_used.assign(O, 0); point_used.assign(L, 0);
  int64_t nres = 0;
  {   // the observations once: subset test of their mask, residual count, blocks in use
    const uint8_t* act = h->h_rp_active.data(); const uint8_t* plan = h->plan_rp_active.data();
    const uint32_t* op = h->h_rp_pose.data(); const uint32_t* ol = h->h_rp_point.data();
    const uint8_t* pc = h->h_pose_const.data(); const uint8_t* lc = h->h_point_const.data();
    for (int64_t a = 0; a < h->n_rp; ++a) {
      if (!act[a]) continue;
      if (!plan[a]) return false;
      const uint32_t p = op[a], l = ol[a];
      const bool cp = pc[p], cl = lc[l];
      if (cp && cl) continue;
      nres += 2;
      if (!cp) pose_used[p] = 1;
      if (!cl) point_used[l] = 1;
    }
  }
  for (int64_t i = 0; i < h->n_bb; ++i) {
    if (!h->h_bb_active[i]) continue;
    const uint32_t o = h->h_bb_obj[i], p = h->h_bb_pose[i];
    const bool co = h->h_object_const[o], cp = h->h_pose_const[p];
    if (co && cp) continue;
    nres += 4;
    if (!co) obj_used[o] = 1;
    if (!cp) pose_used[p] = 1;
  }
  for (int64_t i = 0; i < h->n_sp; ++i) if (h->h_sp_active[i] && !h->h_object_const[h->h_sp_obj[i]]) { nres += 3; obj_used[h->h_sp_obj[i]] = 1; }
  for (int64_t i = 0; i < h->n_lt; ++i) if (h->h_lt_active[i] && !h->h_object_const[h->h_lt_obj[i]]) { nres += 7; obj_used[h->h_lt_obj[i]] = 1; }
  for (int64_t i = 0; i < h->n_rl; ++i) {
    if (!h->h_rl_active[i]) continue;
    const uint32_t a = h->h_rl_a[i], b = h->h_rl_b[i];
    const bool ca = h->h_pose_const[a], cb = h->h_pose_const[b];
    if (ca && cb) continue;
    nres += 6;
    if (!ca) pose_used[a] = 1;
    if (!cb) pose_used[b] = 1;
  }
  if (!h->h_is_shared.empty()) for (int64_t o = 0; o < O; ++o) if (h->h_is_shared[o]) obj_used[o] = 1;
  std::vector<int32_t>& pose_vid = h->scr_pose_vid; std::vector<int32_t>& obj_vid = h->scr_obj_vid;
  std::vector<uint8_t>& point_var = h->scr_point_var; std::vector<uint8_t>& is_pad = h->scr_is_pad;
  pose_vid.assign(P, -1); obj_vid.assign(O, -1); point_var.assign(L, 0); is_pad = h->plan_is_pad;
  int64_t nP = 0, nO = 0, nL = 0;
  for (int64_t p = 0; p < P; ++p) {
    const bool var = !h->h_pose_const[p] && pose_used[p];
    if (var && h->plan_pose_vid[p] < 0) return false;
    if (var) { pose_vid[p] = h->plan_pose_vid[p]; ++nP; }
    else if (h->plan_pose_vid[p] >= 0) { const int64_t r = h->h_pose_row[h->plan_pose_vid[p]]; for (int k = 0; k < 6; ++k) is_pad[r + k] = 1; }
  }
  for (int64_t o = 0; o < O; ++o) {
    const bool var = !h->h_object_const[o] && obj_used[o];
    if (var && h->plan_obj_vid[o] < 0) return false;
    if (var) { obj_vid[o] = h->plan_obj_vid[o]; ++nO; }
    else if (h->plan_obj_vid[o] >= 0) { const int64_t r = h->h_obj_row[h->plan_obj_vid[o]]; for (int k = 0; k < 7; ++k) is_pad[r + k] = 1; }
  }
  for (int64_t l = 0; l < L; ++l) {
    const bool var = !h->h_point_const[l] && point_used[l];
    if (var && !h->plan_point_var[l]) return false;
    if (var) { point_var[l] = 1; ++nL; }
  }
  std::vector<int32_t>& yrow = h->h_rp_yrow;
  yrow.resize((size_t)h->n_rp);
  for (int64_t a = 0; a < h->n_rp; ++a) {
    const int32_t v = h->h_rp_active[a] ? pose_vid[h->h_rp_pose[a]] : -1;
    yrow[a] = v >= 0 ? h->h_pose_row[v] : -1;
  }
  hipStream_t s = h->stream;
  h->d_rp_yrow.upload(yrow, s);
  h->d_pose_vid.upload(pose_vid, s); h->d_obj_vid.upload(obj_vid, s); h->d_point_var.upload(point_var, s); h->d_is_pad.upload(is_pad, s);
  h->h_obj_vid = obj_vid; h->h_is_pad = is_pad;
  h->nLv = nL; h->live_rows = 6 * nP + 7 * nO;
  h->num_params = h->live_rows + 3 * nL;
  h->num_residuals = nres;
  finish_upload(h);   // (the copies went through the pinned arena: nothing to wait for; the solve's first launches follow on the same stream)
  h->mask_dirty = false; h->pc_valid = false; h->tiles_cleared = false;
  return true;
}

StepClear step_clear(obvi_ba_handle* h, double fixed_cost) {
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
void record(obvi_ba_handle* h, int idx, hipStream_t on = nullptr) {
  if (h->profiling < 1) return;
  OBVI_HIP(hipEventRecord(h->ev[idx], on ? on : h->stream));
  if (idx < PH_COUNT) h->phase_on_side[idx] = on != nullptr && on != h->stream;
}
void record_end(obvi_ba_handle* h, int idx, hipStream_t on) { if (h->profiling >= 1) OBVI_HIP(hipEventRecord(h->ev_end[idx], on)); }

// One LM step on the device: linearise at the current point, assemble and solve the damped reduced
// system, form the candidate, evaluate it.  `solve` false: linearisation only (gradient norms).
void submit_step(obvi_ba_handle* h, double radius, bool first_iter, bool solve, bool keep_factor = false) {
  ApiTimer api_timer_("  LM step (submit + wait)");
  hipStream_t s = h->stream;
  const BlocksDev b = blocks_dev(h);
  const ReprojDev rp = reproj_dev(h);
  const SmallFactorsDev sf = small_dev(h);
  const ReducedDev rd = reduced_dev(h);
  const PointDev pt = point_dev(h);
  double* scal = h->d_scal.get();
  const double fixed = h->h_scal[SC_COST_FIXED];
  record(h, PH_POSE_CACHE);
  if (!h->pc_valid) launch_pose_cache(s, h->P, h->d_pose.get(), h->d_pc.get(), h->reproj_variant == OBVI_REPROJECTION_ANALYTIC);
  h->pc_valid = true;
  if (!h->tiles_cleared) launch_zero_tiles(s, rd.S, rd.nt, h->d_tiles.get(), h->ntiles, h->d_is_pad.get(), step_clear(h, fixed));
  h->tiles_cleared = false;
  const bool exchange = h->allreduce != nullptr && !h->h_shared_ov.empty();
  // Fork: the pose-side pass, the small factor families and the diagonal blocks do not depend on the point pass or the
  // Schur complement (everything they share is accumulated with atomics), so they run beside them on the side stream.
  // With a multi-GPU exchange the first collective (the shared objects' blocks) rides on the side stream too: it needs the pose pass and the
  // small factors, and only the diagonal-block kernel behind it needs its result.  Not in an instrumented solve.
  static const bool side_ok = !std::getenv("OBVI_SIDE") || std::atoi(std::getenv("OBVI_SIDE")) != 0;   // tuning knob
  const bool side = h->profiling < 2 && side_ok && !h->deterministic;   // deterministic mode: one stream, so that the kernels that add to the same tiles do so in a fixed order
  hipStream_t s2 = side ? h->stream2 : s;
  // the point pass first, alone: it and the pose-side pass stream the same observation arrays and are both HBM-bound (side by side the
  // point pass took 0.35 ms instead of 0.24); the side stream starts behind it and runs beside the Schur complement, which is bound
  // by instruction issue and LDS, not by HBM
  // ... on a big problem.  On a sliding window every kernel is a few microseconds of latency, nothing is bandwidth-bound, and the side stream
  // (pose pass + small factors + diagonal blocks + far pairs: 63 us for 50 frames) is longer than point pass + Schur complement (47 us): there
  // it forks in front of the point pass.
  const int64_t fork_early_below = std::getenv("OBVI_FORK_EARLY_BELOW") ? std::atoll(std::getenv("OBVI_FORK_EARLY_BELOW")) : 400000;   // tuning knob (observations); read per step: the tests flip it
  const bool fork_early = side && h->n_rp < fork_early_below;
  auto side_pose_pass = [&] {
    record(h, PH_POSE_PASS, s2);
    launch_pose_pass(s2, b, reproj_pose_dev(h), h->d_cams.get(), h->d_pc.get(), h->d_point.get(), rd);
    if (side) record_end(h, PH_POSE_PASS, s2);
  };
  auto side_small_factors = [&] {
    record(h, PH_SMALL, s2);
    launch_small_factors(s2, b, sf, h->d_cams.get(), h->d_pose.get(), h->d_obj.get(), rd, scal);
    if (side) record_end(h, PH_SMALL, s2);
  };
  auto side_diagonal = [&] {
    record(h, PH_DIAG, s2);
    if (exchange) {   // (1) global J^T J diagonal blocks and gradients of the shared objects
      const int32_t ns = (int32_t)h->h_shared_ov.size();
      launch_pack_shared_blocks(s2, b, rd, h->d_shared_ov.get(), ns, h->d_xbuf2.get(), 0);
      if (h->allreduce(h->allreduce_user, h->d_xbuf2.get(), 56 * (int64_t)ns, 0, s2)) throw HipError{hipErrorUnknown, "allreduce hook (shared blocks)", __FILE__, __LINE__};
      launch_pack_shared_blocks(s2, b, rd, h->d_shared_ov.get(), ns, h->d_xbuf2.get(), 1);
    }
    launch_reduced_diag(s2, b, h->d_pose.get(), h->d_obj.get(), rd, radius, first_iter ? 1 : 0, scal);
    if (side) record_end(h, PH_DIAG, s2);
  };
  auto main_schur_window = [&] {
    record(h, PH_SCHUR);
    if (solve) launch_schur_window(s, h->nchunks, h->schur_twins, b, pt, rd, h->d_row_of_nat.get(), h->d_chunk_ptr.get(), h->d_batch_first.get(), h->d_batch_slot.get(), h->d_chunk_points.get(), h->d_slot_src.get(), h->d_chunk_f0.get(), h->d_chunk_group.get());
  };
  auto schur_blocks_on = [&](hipStream_t st) {
    launch_schur_blocks(st, h->nblk, h->d_blk_row.get(), h->d_blk_col.get(), h->d_blk_ptr.get(), h->d_pair_a.get(), h->d_pair_b.get(), rp.point, pt, rd);
  };
  if (fork_early) { OBVI_HIP(hipEventRecord(h->ev_fork, s)); OBVI_HIP(hipStreamWaitEvent(s2, h->ev_fork, 0)); }
  record(h, PH_POINT_PASS);
  launch_point_pass(s, b, rp, h->d_cams.get(), h->d_pc.get(), h->d_point.get(), rd, pt, radius, first_iter ? 1 : 0, scal, h->d_wave_obs.get(), h->n_point_waves, h->d_long_points.get(), h->n_long_points);
  if (side && !fork_early) { OBVI_HIP(hipEventRecord(h->ev_fork, s)); OBVI_HIP(hipStreamWaitEvent(s2, h->ev_fork, 0)); }
  if (fork_early) {
    // a sliding window: every kernel is 5-25 us and the host needs about 5 us per launch, so the two streams are fed alternately -- behind
    // one another, the strip kernel reached its stream 19 us after the point pass had finished (of a 223 us iteration)
    side_pose_pass();
    main_schur_window();
    side_small_factors();
    record(h, PH_SCHUR_BLOCKS);
    if (solve) schur_blocks_on(s);   // (forked early, the side stream is not ordered behind the point pass whose Z records these pairs read: main stream)
    side_diagonal();
    OBVI_HIP(hipEventRecord(h->ev_join, s2));
  } else {
    side_pose_pass();
    side_small_factors();
    side_diagonal();
    // the pairs outside every strip (loop closures, very long tracks) only need the point pass: beside the strip kernel as well (both add
    // to the tile grid with atomics)
    if (side && solve) schur_blocks_on(s2);
    if (side) OBVI_HIP(hipEventRecord(h->ev_join, s2));
    main_schur_window();
    record(h, PH_SCHUR_BLOCKS);
    if (solve && !side) schur_blocks_on(s);
  }
  if (side) OBVI_HIP(hipStreamWaitEvent(s, h->ev_join, 0));   // join
  record(h, PH_CHOL);
  if (solve && h->m > 0) {
    const CholPlan plan = chol_plan(h);
    CholTimers timers{&h->ck_pool, &h->ck_tags, 0};
    CholTimers* tm = h->profiling >= 2 ? &timers : nullptr;
    if (exchange && h->tail_level0 >= 0) {
      launch_cholesky_factor(s, plan, 0, h->tail_level0, rd.S, h->d_Linv.get(), rd.rhs, scal, tm);
      // (2) the rank's own blocks are eliminated: sum the Schur complement onto the shared objects
      const int64_t ntail = h->nt - h->tail_t0;
      launch_pack_tail(s, rd, h->tail_t0, h->d_xbuf.get(), 0);
      if (h->allreduce(h->allreduce_user, h->d_xbuf.get(), ntail * (ntail + 1) / 2 * kTile * kTile + ntail * kTile, 0, s)) throw HipError{hipErrorUnknown, "allreduce hook (shared tail)", __FILE__, __LINE__};
      launch_pack_tail(s, rd, h->tail_t0, h->d_xbuf.get(), 1);
      launch_cholesky_factor(s, plan, h->tail_level0, plan.nlevels, rd.S, h->d_Linv.get(), rd.rhs, scal, tm);
    } else {
      launch_cholesky_factor(s, plan, 0, plan.nlevels, rd.S, h->d_Linv.get(), rd.rhs, scal, tm);
    }
    launch_cholesky_backward(s, plan, rd.S, h->d_Linv.get(), rd.rhs, rd.y, tm);
    h->ck_used = tm ? timers.used : 0;
  } else {
    h->ck_used = 0;
  }
  record(h, PH_BACKSUB);
  if (solve) launch_backsub_apply(s, b, rp, pt, rd, h->d_point.get(), h->d_point_c.get(), h->d_pose.get(), h->d_obj.get(), h->d_pose_c.get(), h->d_obj_c.get(), h->d_pc_c.get(), scal);
  record(h, PH_APPLY);   // (the candidate poses / objects are formed in the same launch)
  record(h, PH_COST);
  if (solve) launch_cost(s, b, reproj_pose_dev(h), sf, h->d_cams.get(), h->d_pc.get(), h->d_pose.get(), h->d_point.get(), h->d_obj.get(), h->d_pc_c.get(),
                         h->d_pose_c.get(), h->d_point_c.get(), h->d_obj_c.get(), 0, scal);
  record(h, PH_COUNT);
  if (exchange) {   // (3) every rank must take the same decision: the sums and every rank's gradient maximum in one collective
    launch_pack_scalars(s, scal, h->d_xbuf.get(), h->rank, h->world, 0);
    if (h->allreduce(h->allreduce_user, h->d_xbuf.get(), (SC_SUM_END - SC_COST) + h->world, 0, s)) throw HipError{hipErrorUnknown, "allreduce hook (scalars)", __FILE__, __LINE__};
    launch_pack_scalars(s, scal, h->d_xbuf.get(), h->rank, h->world, 1);
  }
  static const bool poll_ok = !std::getenv("OBVI_POLL_SCALARS") || std::atoi(std::getenv("OBVI_POLL_SCALARS")) != 0;   // tuning knob
  const bool poll = poll_ok && h->profiling < 1 && !keep_factor;
  // the clear of the next LM step does not depend on the accept / reject decision: it runs while the host takes it
  // (not when the caller goes on to use the factor that is in the tiles: covariance extraction) -- and its first workgroup behind the
  // tiles hands the scalar block to the host before it clears it
  if (poll) {
    h->scal_seq += 1.0;
    StepClear c = step_clear(h, fixed);
    c.pub_host = h->h_scal; c.pub_seq = h->scal_seq;
    launch_zero_tiles(s, rd.S, rd.nt, h->d_tiles.get(), h->ntiles, h->d_is_pad.get(), c); h->tiles_cleared = true;
  } else {
    OBVI_HIP(hipMemcpyAsync(h->h_scal, scal, sizeof(double) * SC_COUNT, hipMemcpyDeviceToHost, s));
    if (!keep_factor) { launch_zero_tiles(s, rd.S, rd.nt, h->d_tiles.get(), h->ntiles, h->d_is_pad.get(), step_clear(h, fixed)); h->tiles_cleared = true; }
  }
  if (poll) wait_scalars(h); else sync(h);
  if (h->h_scal[SC_WAIT_TIMEOUT] != 0.0) {
    // a scheduling event, not a numerical one: nothing the step wrote is kept (the current point is untouched, the accumulators were
    // cleared behind it), so the same step is submitted again on the schedule that cannot wait -- and the handle stays on it
    if (!h->fused_potrf) throw HipError{hipErrorLaunchTimeOut, "tile Cholesky: wait time-out on the two-launch schedule", __FILE__, __LINE__};
    h->fused_potrf = false; h->potrf_wait_timeouts++;
    submit_step(h, radius, first_iter, solve, keep_factor);
    return;
  }
  for (int p = 0; p < PH_COUNT && h->profiling >= 1; ++p) {   // phase timings are opt-in: a dozen event queries per LM iteration are not free
    float ms = 0.f;
    if (h->phase_on_side[p]) { OBVI_HIP(hipEventElapsedTime(&ms, h->ev[p], h->ev_end[p])); }
    else {
      int q = p + 1;
      while (q < PH_COUNT && h->phase_on_side[q]) ++q;   // next phase boundary on the main stream
      OBVI_HIP(hipEventElapsedTime(&ms, h->ev[p], h->ev[q]));
    }
    h->phase_ms[p] += ms;
    h->phase_launches[p] += 1;
  }
  for (int i = 1; i < h->ck_used; ++i) {   // per-kernel events of the tile Cholesky (profiling level 2)
    const int tag = h->ck_tags[i];
    if (tag < 0) continue;
    float ms = 0.f;
    OBVI_HIP(hipEventElapsedTime(&ms, h->ck_pool[i - 1], h->ck_pool[i]));
    h->ck_ms[tag] += ms; h->ck_launches[tag] += 1;
  }
}

double scal_gmax(const obvi_ba_handle* h) { double v; std::memcpy(&v, &h->h_scal[SC_GMAX_BITS], sizeof(v)); return v; }

void copy_current(obvi_ba_handle* h, DevBuf<double>& dp, DevBuf<double>& dl, DevBuf<double>& dobj) {
  hipStream_t s = h->stream;
  dp.resize((size_t)6 * h->P + 1); dl.resize((size_t)3 * h->L + 1); dobj.resize((size_t)7 * h->O + 1);
  launch_copy3(s, dp.get(), h->d_pose.get(), 6 * h->P, dl.get(), h->d_point.get(), 3 * h->L, dobj.get(), h->d_obj.get(), 7 * h->O);
}
void restore_from(obvi_ba_handle* h, const DevBuf<double>& dp, const DevBuf<double>& dl, const DevBuf<double>& dobj) {
  hipStream_t s = h->stream;
  launch_copy3(s, h->d_pose.get(), dp.get(), 6 * h->P, h->d_point.get(), dl.get(), 3 * h->L, h->d_obj.get(), dobj.get(), 7 * h->O);
  h->pc_valid = false;
}

bool check_ready(obvi_ba_handle* h) {
  if (h->h_cams.empty() && (h->n_rp > 0 || h->n_bb > 0)) return false;
  return true;
}

// 1 / std_dev^2 of every parameter prior at its parameter's place: compact reduced index for poses / objects, [L][3] for points
void upload_parameter_prior_diagonals(obvi_ba_handle* h) {
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
    else if (h->h_pp_kind[i] == 1) el[3 * b + h->h_pp_param[i]] += w;
    else if (obj_vid[b] >= 0) ec[6 * h->nPv + 7 * (int64_t)obj_vid[b] + h->h_pp_param[i]] += w;
  }
  h->d_extra_c.upload(ec, h->stream); h->d_extra_l.upload(el, h->stream);
  sync(h);
}

template <class T>
void set_mask(std::vector<uint8_t>& host, DevBuf<uint8_t>& dev, const uint8_t* mask, int64_t n, hipStream_t s, const T* perm_sorted_to_orig) {
  host.resize(n);
  for (int64_t i = 0; i < n; ++i) host[i] = mask ? (mask[perm_sorted_to_orig ? perm_sorted_to_orig[i] : i] != 0) : 1;
  dev.upload(host, s);
}

}  // namespace

namespace obvi {
hipStream_t handle_stream(obvi_ba_handle* h) { return h->stream; }
int handle_device(const obvi_ba_handle* h) { return h->device; }
int handle_fail(obvi_ba_handle* h, int code, const char* msg) { return fail(h, code, msg); }
void make_dev_cam(const double* K4, const double* ext7, DevCam* out) { make_cam(K4, ext7, out); }
}  // namespace obvi

// =========================================================================================
extern "C" {

const char* obvi_ba_version(void) { return "obvi_ba 0.1 (gfx950)"; }
const char* obvi_ba_last_error(const obvi_ba_handle* h) { return h ? h->err.c_str() : "null handle"; }

int obvi_ba_create(const obvi_ba_options* options, obvi_ba_handle** out) {
  if (!out) return OBVI_ERR_INVALID_ARGUMENT;
  ApiTimer api_timer_(__func__);
  *out = nullptr;
  if (options && options->object_block_size != 0 && options->object_block_size != 7) return OBVI_ERR_INVALID_ARGUMENT;
  if (options && options->reprojection_variant != OBVI_REPROJECTION_AUTODIFF && options->reprojection_variant != OBVI_REPROJECTION_ANALYTIC) return OBVI_ERR_INVALID_ARGUMENT;
  // OBVI_DEBUG_CREATE: where the time of a create goes, on stderr (the first one of a process also starts the HIP runtime)
  const bool create_times = std::getenv("OBVI_DEBUG_CREATE") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!create_times) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "create: %-44s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
    t_prev = now;
  };
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return OBVI_ERR_NO_DEVICE;
  lap("hipGetDeviceCount (runtime start)");
  const int dev = options ? options->device_id : 0;
  if (dev < 0 || dev >= count) return OBVI_ERR_NO_DEVICE;
  obvi_ba_handle* h = new (std::nothrow) obvi_ba_handle();
  if (!h) return OBVI_ERR_HIP;
  h->device = dev;
  if (options) { h->reproj_variant = options->reprojection_variant; h->deterministic = options->deterministic != 0; }
  if (const char* env = std::getenv("OBVI_FUSED_POTRF")) h->fused_potrf = std::atoi(env) != 0;   // 0: two launches per level from the start (CI parity run)
  if (const char* env = std::getenv("OBVI_DETERMINISTIC")) { if (std::atoi(env) != 0) h->deterministic = true; }   // every handle of the process (a session driven through a host that does not set the option)
  try {
    OBVI_HIP(hipSetDevice(dev));
    lap("hipSetDevice");
    OBVI_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    lap("first stream");
    // coherent (fine-grained): the host polls this page while the step is still running (wait_scalars); with a non-coherent mapping it
    // would see the device's write only at the end of the stream
    OBVI_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->h_scal), sizeof(double) * (SC_COUNT + 1), hipHostMallocCoherent));
    std::memset(h->h_scal, 0, sizeof(double) * (SC_COUNT + 1));
    lap("pinned scalar page");
    OBVI_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->staging.base), kStagingBytes, hipHostMallocDefault));
    h->staging.cap = kStagingBytes;
    lap("pinned staging arena");
    h->d_scal.resize(SC_COUNT);   // deterministic mode: grown by ensure_det_slots() to hold per-workgroup partial sums behind the block (ba_device.h)
    OBVI_HIP(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    for (auto& e : h->ev) OBVI_HIP(hipEventCreate(&e));
    for (auto& e : h->ev_end) OBVI_HIP(hipEventCreate(&e));
    OBVI_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming)); OBVI_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));   // ordering only: no timestamps
    lap("scalar block, second stream, events");
  } catch (const HipError&) {
    delete h;
    return OBVI_ERR_HIP;
  }
  *out = h;
  return OBVI_OK;
}

void obvi_ba_destroy(obvi_ba_handle* h) {
  if (!h) return;
  ApiTimer api_timer_(__func__);
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->stream2) (void)hipStreamSynchronize(h->stream2);
  for (auto& e : h->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : h->ev_end) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : {h->ev_fork, h->ev_join}) if (e) (void)hipEventDestroy(e);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  if (h->h_scal) (void)hipHostFree(h->h_scal);
  if (h->staging.base) (void)hipHostFree(h->staging.base);
  select_scratch_free(&h->sel_scratch);
  // DevBuf members free in ~obvi_ba_handle
  hipStream_t s = h->stream;
  delete h;
  if (s) (void)hipStreamDestroy(s);
}

int obvi_ba_reset(obvi_ba_handle* h) {
  if (!h) return OBVI_ERR_INVALID_ARGUMENT;
  // the state obvi_ba_create leaves, with the device allocations, streams, events and pinned pages kept: no cameras, blocks or factors, no
  // parameter priors, nothing shared and no exchange hook, no snapshot, no iteration records, profiling off and its sums at zero
  int rc = obvi_ba_set_parameter_priors(h, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
  if (!rc) rc = obvi_ba_set_reproj(h, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, 1.0);
  if (!rc) rc = obvi_ba_set_bbox(h, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 1.0, 1e6);
  if (!rc) rc = obvi_ba_set_shape_priors(h, 0, nullptr, nullptr, nullptr, 1.0);
  if (!rc) rc = obvi_ba_set_ltm_priors(h, 0, nullptr, nullptr, nullptr, 1.0);
  if (!rc) rc = obvi_ba_set_relpose(h, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 1.0);
  if (!rc) rc = obvi_ba_set_poses(h, 0, nullptr, nullptr);
  if (!rc) rc = obvi_ba_set_points(h, 0, nullptr, nullptr);
  if (!rc) rc = obvi_ba_set_objects(h, 0, nullptr, nullptr);
  if (!rc) rc = obvi_ba_set_cameras(h, 0, nullptr, nullptr);
  if (rc) return rc;
  OBVI_API_BEGIN
  h->h_is_shared.clear(); h->h_shared_ov.clear(); h->rank = 0; h->world = 1; h->tail_t0 = -1; h->tail_level0 = -1;
  h->allreduce = nullptr; h->allreduce_user = nullptr;
  h->have_snapshot = false; h->use_extra = false; h->pc_valid = false; h->tiles_cleared = false;
  h->iterations.clear();
  h->profiling = 0; h->ck_used = 0;
  for (auto& v : h->phase_ms) v = 0.0;
  for (auto& v : h->phase_launches) v = 0;
  for (auto& v : h->ck_ms) v = 0.0;
  for (auto& v : h->ck_launches) v = 0;
  // fused_potrf / potrf_wait_timeouts stay: a wait time-out of the fused kernel is a property of the device and runtime (dispatch order), not of
  // the problem -- a pooled handle that learned it does not pay the on-device wait again at every reuse
  h->dirty = true; h->mask_dirty = false;
  h->err.clear();
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_set_cameras(obvi_ba_handle* h, int32_t n, const double* K, const double* ext) {
  if (!h || n < 0 || (n > 0 && (!K || !ext))) return fail(h, OBVI_ERR_INVALID_ARGUMENT, "set_cameras: bad arguments");
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  h->h_cams.resize(n);
  for (int i = 0; i < n; ++i) {
    make_cam(K + 4 * i, ext + 7 * i, &h->h_cams[i]);
    if (h->reproj_variant == OBVI_REPROJECTION_ANALYTIC) h->h_cams[i].depth_min = OBVI_ANALYTIC_EPSILON;
  }
  h->d_cams.upload(h->h_cams, h->stream);
  finish_upload(h);
  bake_bbox(h);   // the bounding-box factors already uploaded follow the new intrinsics
  return OBVI_OK;
  OBVI_API_END(h)
}

static int set_blocks(obvi_ba_handle* h, int64_t n, int dim, const double* v, const uint8_t* c, int64_t* count, std::vector<uint8_t>* hc, DevBuf<double>* dv) {
  if (!h || n < 0 || (n > 0 && !v)) return fail(h, OBVI_ERR_INVALID_ARGUMENT, "set_blocks: bad arguments");
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  *count = n;
  if (c) hc->assign(c, c + n); else hc->assign(n, 0);
  dv->resize((size_t)n * dim + 1);
  h2d_async(dv->get(), v, sizeof(double) * n * dim, h->stream);
  finish_upload(h);
  h->dirty = true;
  h->have_snapshot = false;
  return OBVI_OK;
  OBVI_API_END(h)
}
int obvi_ba_set_poses(obvi_ba_handle* h, int64_t n, const double* v, const uint8_t* c) { return set_blocks(h, n, 6, v, c, h ? &h->P : nullptr, h ? &h->h_pose_const : nullptr, h ? &h->d_pose : nullptr); }
int obvi_ba_set_points(obvi_ba_handle* h, int64_t n, const double* v, const uint8_t* c) { return set_blocks(h, n, 3, v, c, h ? &h->L : nullptr, h ? &h->h_point_const : nullptr, h ? &h->d_point : nullptr); }
int obvi_ba_set_objects(obvi_ba_handle* h, int64_t n, const double* v, const uint8_t* c) { return set_blocks(h, n, 7, v, c, h ? &h->O : nullptr, h ? &h->h_object_const : nullptr, h ? &h->d_obj : nullptr); }

int obvi_ba_set_const_flags(obvi_ba_handle* h, const uint8_t* pc, const uint8_t* lc, const uint8_t* oc) {
  if (!h) return OBVI_ERR_INVALID_ARGUMENT;
  OBVI_API_BEGIN
  if (pc) h->h_pose_const.assign(pc, pc + h->P);
  if (lc) h->h_point_const.assign(lc, lc + h->L);
  if (oc) h->h_object_const.assign(oc, oc + h->O);
  h->dirty = true;
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_update_points(obvi_ba_handle* h, int64_t n, const double* xyz) {
  if (!h || n != h->L || (n > 0 && !xyz)) return fail(h, OBVI_ERR_INVALID_ARGUMENT, "update_points: size mismatch");
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  h2d_async(h->d_point.get(), xyz, sizeof(double) * 3 * n, h->stream);
  finish_upload(h);
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_set_reproj(obvi_ba_handle* h, int64_t n, const uint32_t* pose_idx, const uint32_t* point_idx, const uint16_t* cam_idx,
                       const double* pixel, const double* sigma, double sigma_scalar, double huber) {
  if (!h || n < 0 || (n > 0 && (!pose_idx || !point_idx || !pixel))) return fail(h, OBVI_ERR_INVALID_ARGUMENT, "set_reproj: bad arguments");
  if (n >= (int64_t)0xffffffffu) return fail(h, OBVI_ERR_INVALID_ARGUMENT, "set_reproj: more than 2^32-1 observations");
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  for (int64_t i = 0; i < n; ++i) {
    const int cam = cam_idx ? cam_idx[i] : 0;
    if (pose_idx[i] >= h->P || point_idx[i] >= h->L || cam >= (int)h->h_cams.size()) return fail(h, OBVI_ERR_OUT_OF_RANGE, "set_reproj: index out of range");
  }
  double t_sub = wall_s();
  auto sub = [&](const char* name) { if (ApiTimes* t = api_times()) { const double now = wall_s(); t->add(name, 1e3 * (now - t_sub)); t_sub = now; } };
  sub("    set_reproj: validate");
  // CSC by point: counting sort on the point index, then by pose inside each point.  (A sliding window calls this for every frame with
  // about the same n: the index arrays live in the handle, the device-only arrays are filled in pinned memory -- no allocation, no
  // second copy.)
  std::vector<uint32_t>& perm = h->h_rp_perm;
  std::vector<uint32_t>& ptr = h->h_point_ptr;
  std::vector<uint32_t>& cur = h->scr_cursor;
  perm.resize(n); ptr.assign(h->L + 1, 0);
  for (int64_t i = 0; i < n; ++i) ptr[point_idx[i] + 1]++;
  for (int64_t l = 0; l < h->L; ++l) ptr[l + 1] += ptr[l];
  cur.assign(ptr.begin(), ptr.end() - 1);
  for (int64_t i = 0; i < n; ++i) perm[cur[point_idx[i]]++] = (uint32_t)i;
  // ranges of points / observations on the host's worker threads -- from a few hundred thousand observations on: a window's 50 k are
  // 0.6 ms on one thread and 0.85-1.3 ms on 2-16 (waking the workers, 256 cores on two sockets passing cache lines around)
  const int threads = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), n / 131072));
  parallel_ranges(h->L, threads, [&](int, int64_t l0, int64_t l1) {
    auto before = [&](uint32_t x, uint32_t y) { return pose_idx[x] < pose_idx[y] || (pose_idx[x] == pose_idx[y] && x < y); };
    for (int64_t l = l0; l < l1; ++l)
      if (!std::is_sorted(perm.begin() + ptr[l], perm.begin() + ptr[l + 1], before)) std::sort(perm.begin() + ptr[l], perm.begin() + ptr[l + 1], before);
  });
  sub("    set_reproj: sort by point");
  h->n_rp = n; h->rp_huber = huber; h->rp_inv_on_device = false;
  h->max_rp_pose = max_index(pose_idx, n); h->max_rp_point = max_index(point_idx, n); h->max_rp_cam = cam_idx ? max_index(cam_idx, n) : (n > 0 ? 0 : -1);
  h->h_rp_pose.resize(n); h->h_rp_point.resize(n); h->h_rp_active.assign(n, 1); h->h_rp_inv.resize(n);
  hipStream_t s = h->stream;
  // the arrays only the device reads (camera, pixel, sigma) go up the way they came and are put in both observation orders by a kernel
  // (launch_reproj_gather); the host keeps and permutes the index arrays its symbolic phase reads
  if (cam_idx) h->d_raw_cam.upload(cam_idx, (size_t)n, s);
  h->d_raw_pixel.upload(reinterpret_cast<const double2*>(pixel), (size_t)n, s);
  if (sigma) h->d_raw_sigma.upload(sigma, (size_t)n, s);
  parallel_ranges(n, threads, [&](int, int64_t a0, int64_t a1) {
    for (int64_t a = a0; a < a1; ++a) {
      const uint32_t i = perm[a];
      h->h_rp_inv[i] = (uint32_t)a;
      h->h_rp_pose[a] = pose_idx[i]; h->h_rp_point[a] = point_idx[i];
    }
  });
  sub("    set_reproj: gather");
  {   // k_point_pass: the observation list cut into wavefront-sized pieces (<= 64 observations, whole points); longer tracks go to the per-point kernel
    std::vector<uint32_t>& wave_obs = h->scr_wave_obs;   // (first observation, count) per piece
    std::vector<uint32_t>& long_points = h->scr_long_points;
    wave_obs.clear(); long_points.clear();
    uint32_t start = 0, count = 0;
    int64_t lfirst = 0;
    for (int64_t l = 0; l < h->L; ++l) {
      const uint32_t k = ptr[l + 1] - ptr[l];
      if (k == 0) continue;
      if (count > 0 && (k > 64 || count + k > 64 || l - lfirst >= 64)) { wave_obs.push_back(start); wave_obs.push_back(count); count = 0; }
      if (k > 64) { long_points.push_back((uint32_t)l); continue; }
      if (count == 0) { start = ptr[l]; lfirst = l; }
      count += k;
    }
    if (count > 0) { wave_obs.push_back(start); wave_obs.push_back(count); }
    h->n_point_waves = (int64_t)wave_obs.size() / 2;
    h->n_long_points = (int64_t)long_points.size();
    h->d_wave_obs.upload(wave_obs, s); h->d_long_points.upload(long_points, s);
  }
  sub("    set_reproj: wave pieces");
  h->d_rp_pose.upload(h->h_rp_pose, s); h->d_rp_point.upload(h->h_rp_point, s); h->d_rp_perm.upload(perm, s); h->d_point_ptr.upload(ptr, s);
  h->d_rp_active.upload(h->h_rp_active, s);
  sub("    set_reproj: upload by point");
  // CSR-by-pose order for the pose-side pass (counting sort on the pose index; stable, so points ascend inside a pose)
  std::vector<uint32_t>& pptr = h->scr_pose_ptr;
  pptr.assign(h->P + 1, 0);
  h->h_rq_src.resize(n);
  for (int64_t a = 0; a < n; ++a) pptr[h->h_rp_pose[a] + 1]++;
  for (int64_t p = 0; p < h->P; ++p) pptr[p + 1] += pptr[p];
  cur.assign(pptr.begin(), pptr.end() - 1);
  for (int64_t a = 0; a < n; ++a) h->h_rq_src[cur[h->h_rp_pose[a]]++] = (uint32_t)a;
  sub("    set_reproj: by pose");
  h->d_rq_src.upload(h->h_rq_src, s); h->d_rq_pose_ptr.upload(pptr, s);
  h->d_rp_cam.resize((size_t)n); h->d_rp_pixel.resize((size_t)n); h->d_rp_sigma.resize((size_t)n);
  h->d_rq_point.resize((size_t)n); h->d_rq_cam.resize((size_t)n); h->d_rq_pixel.resize((size_t)n); h->d_rq_sigma.resize((size_t)n); h->d_rq_active.resize((size_t)n);
  launch_reproj_gather(s, n, h->d_rp_perm.get(), h->d_rq_src.get(), h->d_rp_point.get(), cam_idx ? h->d_raw_cam.get() : nullptr, h->d_raw_pixel.get(), sigma ? h->d_raw_sigma.get() : nullptr,
                       sigma_scalar, h->d_rp_cam.get(), h->d_rp_pixel.get(), h->d_rp_sigma.get(), h->d_rq_point.get(), h->d_rq_cam.get(), h->d_rq_pixel.get(), h->d_rq_sigma.get(),
                       h->d_rq_active.get());
  finish_upload(h);
  sub("    set_reproj: upload by pose + finish");
  h->dirty = true;
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_set_bbox(obvi_ba_handle* h, int64_t n, const uint32_t* obj_idx, const uint32_t* pose_idx, const uint16_t* cam_idx,
                     const double* corners, const double* cov, double huber, double invalid_err) {
  if (!h || n < 0 || (n > 0 && (!obj_idx || !pose_idx || !corners || !cov))) return fail(h, OBVI_ERR_INVALID_ARGUMENT, "set_bbox: bad arguments");
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  std::vector<uint16_t> cam(n);
  std::vector<double> m4all(16 * n);
  for (int64_t i = 0; i < n; ++i) {
    cam[i] = cam_idx ? cam_idx[i] : 0;
    if (obj_idx[i] >= h->O || pose_idx[i] >= h->P || cam[i] >= h->h_cams.size()) return fail(h, OBVI_ERR_OUT_OF_RANGE, "set_bbox: index out of range");
    if (!sym_inverse_sqrt(cov + 16 * i, 4, &m4all[16 * i])) return fail(h, OBVI_ERR_NUMERICAL, "set_bbox: covariance not SPD");
  }
  h->n_bb = n; h->bb_huber = huber; h->bb_invalid = invalid_err;
  h->h_bb_obj.assign(obj_idx, obj_idx + n); h->h_bb_pose.assign(pose_idx, pose_idx + n); h->h_bb_active.assign(n, 1);
  hipStream_t s = h->stream;
  h->max_bb_obj = max_index(obj_idx, n); h->max_bb_pose = max_index(pose_idx, n); h->max_bb_cam = max_index(cam.data(), n);
  h->h_bb_cam = cam; h->h_bb_corners.assign(corners, corners + 4 * n); h->h_bb_m4.swap(m4all);
  h->d_bb_obj.upload(h->h_bb_obj, s); h->d_bb_pose.upload(h->h_bb_pose, s); h->d_bb_cam.upload(cam, s);
  h->d_bb_active.upload(h->h_bb_active, s);
  finish_upload(h);
  bake_bbox(h);
  h->dirty = true;
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_set_shape_priors(obvi_ba_handle* h, int64_t n, const uint32_t* obj_idx, const double* mean3, const double* cov9, double huber) {
  if (!h || n < 0 || (n > 0 && (!obj_idx || !mean3 || !cov9))) return fail(h, OBVI_ERR_INVALID_ARGUMENT, "set_shape_priors: bad arguments");
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  std::vector<double> si(9 * n);
  for (int64_t i = 0; i < n; ++i) {
    if (obj_idx[i] >= h->O) return fail(h, OBVI_ERR_OUT_OF_RANGE, "set_shape_priors: index out of range");
    if (!sym_inverse_sqrt(cov9 + 9 * i, 3, &si[9 * i])) return fail(h, OBVI_ERR_NUMERICAL, "set_shape_priors: covariance not SPD");
  }
  h->n_sp = n; h->sp_huber = huber; h->max_sp_obj = max_index(obj_idx, n);
  h->h_sp_obj.assign(obj_idx, obj_idx + n); h->h_sp_active.assign(n, 1);
  hipStream_t s = h->stream;
  h->d_sp_obj.upload(h->h_sp_obj, s); h->d_sp_mean.upload(mean3, 3 * n, s); h->d_sp_sqrt_inf.upload(si, s); h->d_sp_active.upload(h->h_sp_active, s);
  finish_upload(h);
  h->dirty = true;
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_set_ltm_priors(obvi_ba_handle* h, int64_t n, const uint32_t* obj_idx, const double* mean7, const double* cov49, double huber) {
  if (!h || n < 0 || (n > 0 && (!obj_idx || !mean7 || !cov49))) return fail(h, OBVI_ERR_INVALID_ARGUMENT, "set_ltm_priors: bad arguments");
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  std::vector<double> si(49 * n);
  for (int64_t i = 0; i < n; ++i) {
    if (obj_idx[i] >= h->O) return fail(h, OBVI_ERR_OUT_OF_RANGE, "set_ltm_priors: index out of range");
    if (!sym_inverse_sqrt(cov49 + 49 * i, 7, &si[49 * i])) return fail(h, OBVI_ERR_NUMERICAL, "set_ltm_priors: covariance not SPD");
  }
  h->n_lt = n; h->lt_huber = huber; h->max_lt_obj = max_index(obj_idx, n);
  h->h_lt_obj.assign(obj_idx, obj_idx + n); h->h_lt_active.assign(n, 1);
  hipStream_t s = h->stream;
  h->d_lt_obj.upload(h->h_lt_obj, s); h->d_lt_mean.upload(mean7, 7 * n, s); h->d_lt_sqrt_inf.upload(si, s); h->d_lt_active.upload(h->h_lt_active, s);
  finish_upload(h);
  h->dirty = true;
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_set_relpose(obvi_ba_handle* h, int64_t n, const uint32_t* ia, const uint32_t* ib, const double* t3, const double* aa3,
                        const double* cov36, double huber) {
  if (!h || n < 0 || (n > 0 && (!ia || !ib || !t3 || !aa3 || !cov36))) return fail(h, OBVI_ERR_INVALID_ARGUMENT, "set_relpose: bad arguments");
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  std::vector<double> R(9 * n), si(36 * n);
  for (int64_t i = 0; i < n; ++i) {
    if (ia[i] >= h->P || ib[i] >= h->P) return fail(h, OBVI_ERR_OUT_OF_RANGE, "set_relpose: index out of range");
    // measured_pose_deviation.orientation_.toRotationMatrix() (relative_pose_factor.cpp:11-12)
    const double* a = aa3 + 3 * i;
    const double th = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    double* Ri = &R[9 * i];
    if (th > 0.0) {
      const double ux = a[0] / th, uy = a[1] / th, uz = a[2] / th, s = std::sin(th), c = std::cos(th), oc = 1.0 - c;
      Ri[0] = oc * ux * ux + c;      Ri[1] = oc * ux * uy - s * uz; Ri[2] = oc * ux * uz + s * uy;
      Ri[3] = oc * ux * uy + s * uz; Ri[4] = oc * uy * uy + c;      Ri[5] = oc * uy * uz - s * ux;
      Ri[6] = oc * ux * uz - s * uy; Ri[7] = oc * uy * uz + s * ux; Ri[8] = oc * uz * uz + c;
    } else {
      for (int k = 0; k < 9; ++k) Ri[k] = (k % 4 == 0) ? 1.0 : 0.0;
    }
    if (!sym_inverse_sqrt(cov36 + 36 * i, 6, &si[36 * i])) return fail(h, OBVI_ERR_NUMERICAL, "set_relpose: covariance not SPD");
  }
  h->n_rl = n; h->rl_huber = huber; h->max_rl_pose = std::max(max_index(ia, n), max_index(ib, n));
  h->h_rl_a.assign(ia, ia + n); h->h_rl_b.assign(ib, ib + n); h->h_rl_active.assign(n, 1);
  hipStream_t s = h->stream;
  h->d_rl_a.upload(h->h_rl_a, s); h->d_rl_b.upload(h->h_rl_b, s); h->d_rl_t.upload(t3, 3 * n, s); h->d_rl_R.upload(R, s);
  h->d_rl_sqrt_inf.upload(si, s); h->d_rl_active.upload(h->h_rl_active, s);
  finish_upload(h);
  h->dirty = true;
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_set_active_mask(obvi_ba_handle* h, int32_t type, const uint8_t* mask) {
  if (!h) return OBVI_ERR_INVALID_ARGUMENT;
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  switch (type) {
    case OBVI_FACTOR_REPROJECTION: {
      set_mask(h->h_rp_active, h->d_rp_active, mask, h->n_rp, s, h->h_rp_perm.data());
      std::vector<uint8_t> q(h->n_rp);
      for (int64_t k = 0; k < h->n_rp; ++k) q[k] = h->h_rp_active[h->h_rq_src[k]];
      h->d_rq_active.upload(q, s);
      finish_upload(h);
      break;
    }
    case OBVI_FACTOR_BBOX: set_mask<uint32_t>(h->h_bb_active, h->d_bb_active, mask, h->n_bb, s, nullptr); break;
    case OBVI_FACTOR_SHAPE_PRIOR: set_mask<uint32_t>(h->h_sp_active, h->d_sp_active, mask, h->n_sp, s, nullptr); break;
    case OBVI_FACTOR_LTM_PRIOR: set_mask<uint32_t>(h->h_lt_active, h->d_lt_active, mask, h->n_lt, s, nullptr); break;
    case OBVI_FACTOR_REL_POSE: set_mask<uint32_t>(h->h_rl_active, h->d_rl_active, mask, h->n_rl, s, nullptr); break;
    default: return fail(h, OBVI_ERR_INVALID_ARGUMENT, "set_active_mask: unknown factor type");
  }
  finish_upload(h);
  h->mask_dirty = true;   // prepare() keeps the symbolic plan if the new masks select a subset of what it was built for
  return OBVI_OK;
  OBVI_API_END(h)
}

int64_t obvi_ba_num_factors(const obvi_ba_handle* h, int32_t type) {
  if (!h) return -1;
  switch (type) {
    case OBVI_FACTOR_REPROJECTION: return h->n_rp; case OBVI_FACTOR_BBOX: return h->n_bb; case OBVI_FACTOR_SHAPE_PRIOR: return h->n_sp;
    case OBVI_FACTOR_LTM_PRIOR: return h->n_lt; case OBVI_FACTOR_REL_POSE: return h->n_rl; default: return -1;
  }
}
int64_t obvi_ba_num_residuals(const obvi_ba_handle* h) { return h ? 2 * h->n_rp + 4 * h->n_bb + 3 * h->n_sp + 7 * h->n_lt + 6 * h->n_rl : -1; }

int obvi_ba_evaluate(obvi_ba_handle* h, int32_t apply_loss, double* cost, double* residuals, double* block_sqnorm) {
  if (!h) return OBVI_ERR_INVALID_ARGUMENT;
  if (!check_ready(h)) return fail(h, OBVI_ERR_NOT_READY, "evaluate: cameras not set");
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  { const int vrc = validate_indices(h); if (vrc != OBVI_OK) return vrc; }
  prepare(h);
  hipStream_t s = h->stream;
  const int64_t nres = obvi_ba_num_residuals(h), nfac = h->n_rp + h->n_bb + h->n_sp + h->n_lt + h->n_rl;
  h->d_eval_res.resize((size_t)nres + 1); h->d_eval_sq.resize((size_t)nfac + 1);
  OBVI_HIP(hipMemsetAsync(h->d_scal.get(), 0, sizeof(double) * SC_COUNT, s));
  launch_pose_cache(s, h->P, h->d_pose.get(), h->d_pc.get(), h->reproj_variant == OBVI_REPROJECTION_ANALYTIC);
  launch_evaluate(s, blocks_dev(h), reproj_dev(h), h->d_rp_perm.get(), small_dev(h), h->d_cams.get(), h->d_pc.get(), h->d_pose.get(),
                  h->d_point.get(), h->d_obj.get(), apply_loss, h->d_eval_res.get(), h->d_eval_sq.get(), h->d_scal.get());
  if (!cost && !residuals && !block_sqnorm) return OBVI_OK;   // nothing to hand back (obvi_ba_select_outliers: its kernels follow on the same stream)
  double c = 0.0;
  OBVI_HIP(hipMemcpyAsync(&c, h->d_scal.get() + SC_COST, sizeof(double), hipMemcpyDeviceToHost, s));
  if (residuals) h->d_eval_res.download(residuals, (size_t)nres, s);
  if (block_sqnorm) h->d_eval_sq.download(block_sqnorm, (size_t)nfac, s);
  sync(h);
  if (cost) *cost = c;
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_debug_linearize(obvi_ba_handle* h, int32_t type, double* r, double* J0, double* J1) {
  if (!h || !r || !J0) return OBVI_ERR_INVALID_ARGUMENT;
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  { const int vrc = validate_indices(h); if (vrc != OBVI_OK) return vrc; }
  prepare(h);
  hipStream_t s = h->stream;
  int m, d0, d1; int64_t n;
  switch (type) {
    case OBVI_FACTOR_REPROJECTION: m = 2; d0 = 6; d1 = 3; n = h->n_rp; break;
    case OBVI_FACTOR_BBOX: m = 4; d0 = 7; d1 = 6; n = h->n_bb; break;
    case OBVI_FACTOR_SHAPE_PRIOR: m = 3; d0 = 7; d1 = 0; n = h->n_sp; break;
    case OBVI_FACTOR_LTM_PRIOR: m = 7; d0 = 7; d1 = 0; n = h->n_lt; break;
    case OBVI_FACTOR_REL_POSE: m = 6; d0 = 6; d1 = 6; n = h->n_rl; break;
    default: return fail(h, OBVI_ERR_INVALID_ARGUMENT, "debug_linearize: unknown factor type");
  }
  DevBuf<double> dr, dJ0, dJ1;
  dr.resize((size_t)n * m + 1); dJ0.resize((size_t)n * m * d0 + 1); dJ1.resize((size_t)n * m * d1 + 1);
  if (type == OBVI_FACTOR_REPROJECTION) {
    launch_pose_cache(s, h->P, h->d_pose.get(), h->d_pc.get(), h->reproj_variant == OBVI_REPROJECTION_ANALYTIC);
    launch_debug_linearize_reproj(s, reproj_dev(h), h->d_rp_perm.get(), h->d_cams.get(), h->d_pc.get(), h->d_point.get(), dr.get(), dJ0.get(), dJ1.get());
  } else {
    launch_debug_linearize_small(s, type, small_dev(h), h->d_cams.get(), h->d_pose.get(), h->d_obj.get(), dr.get(), dJ0.get(), dJ1.get());
  }
  dr.download(r, (size_t)n * m, s); dJ0.download(J0, (size_t)n * m * d0, s);
  if (J1 && d1) dJ1.download(J1, (size_t)n * m * d1, s);
  sync(h);
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_debug_reduced_system(obvi_ba_handle* h, double radius, double* lhs, double* rhs, int32_t m_cap, int32_t* m_out) {
  if (!h || !lhs || !rhs) return OBVI_ERR_INVALID_ARGUMENT;
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  { const int vrc = validate_indices(h); if (vrc != OBVI_OK) return vrc; }
  if (h->mask_dirty) h->dirty = true;   // the canonical (caller-order) view below is that of a plan built for exactly the current masks
  prepare(h);
  if (m_out) *m_out = (int32_t)h->m_canon;
  if (h->m_canon > m_cap) return fail(h, OBVI_ERR_INVALID_ARGUMENT, "debug_reduced_system: buffer too small");
  hipStream_t s = h->stream;
  const BlocksDev b = blocks_dev(h); const ReprojDev rp = reproj_dev(h); const SmallFactorsDev sf = small_dev(h);
  const ReducedDev rd = reduced_dev(h); const PointDev pt = point_dev(h);
  launch_pose_cache(s, h->P, h->d_pose.get(), h->d_pc.get(), h->reproj_variant == OBVI_REPROJECTION_ANALYTIC);
  launch_zero_tiles(s, rd.S, rd.nt, h->d_tiles.get(), h->ntiles, h->d_is_pad.get(), step_clear(h, 0.0));
  launch_point_pass(s, b, rp, h->d_cams.get(), h->d_pc.get(), h->d_point.get(), rd, pt, radius, 1, h->d_scal.get(), h->d_wave_obs.get(), h->n_point_waves, h->d_long_points.get(), h->n_long_points);
  launch_pose_pass(s, b, reproj_pose_dev(h), h->d_cams.get(), h->d_pc.get(), h->d_point.get(), rd);
  launch_small_factors(s, b, sf, h->d_cams.get(), h->d_pose.get(), h->d_obj.get(), rd, h->d_scal.get());
  launch_reduced_diag(s, b, h->d_pose.get(), h->d_obj.get(), rd, radius, 1, h->d_scal.get());
  { launch_schur_window(s, h->nchunks, h->schur_twins, b, pt, rd, h->d_row_of_nat.get(), h->d_chunk_ptr.get(), h->d_batch_first.get(), h->d_batch_slot.get(), h->d_chunk_points.get(), h->d_slot_src.get(), h->d_chunk_f0.get(), h->d_chunk_group.get());
    launch_schur_blocks(s, h->nblk, h->d_blk_row.get(), h->d_blk_col.get(), h->d_blk_ptr.get(), h->d_pair_a.get(), h->d_pair_b.get(), rp.point, pt, rd); }
  const int64_t mc = h->m_canon, nt = h->nt;
  std::vector<double> tiles((size_t)nt * nt * kTile * kTile), hr((size_t)nt * kTile);
  std::vector<int32_t> tl((size_t)2 * h->ntiles);
  h->d_S.download(tiles.data(), tiles.size(), s); h->d_rhs.download(hr.data(), hr.size(), s);
  h->d_tiles.download(tl.data(), tl.size(), s);
  sync(h);
  // tiles outside the structural mask are never written: read them as zero
  std::vector<uint8_t> mk((size_t)nt * nt, 0);
  for (int t = 0; t < h->ntiles; ++t) mk[(size_t)tl[2 * t] * nt + tl[2 * t + 1]] = 1;
  // canonical order (variable poses by index, then objects) <- rows of the tile grid (elimination order)
  for (int64_t ci = 0; ci < mc; ++ci) {
    const int64_t i = h->h_canon_row[ci];
    rhs[ci] = hr[i];
    for (int64_t cj = 0; cj < mc; ++cj) {
      const int64_t j = h->h_canon_row[cj];
      const int64_t r = std::max(i, j), c = std::min(i, j);
      const size_t tix = (size_t)(r / kTile) * nt + (c / kTile);
      lhs[ci * mc + cj] = mk[tix] ? tiles[tix * (kTile * kTile) + (r % kTile) * kTile + (c % kTile)] : 0.0;
    }
  }
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_object_covariances(obvi_ba_handle* h, int64_t n_pairs, const uint32_t* obj_a, const uint32_t* obj_b, double* cov49) {
  if (!h || n_pairs < 0 || (n_pairs > 0 && (!obj_a || !obj_b || !cov49))) return OBVI_ERR_INVALID_ARGUMENT;
  if (!check_ready(h)) return fail(h, OBVI_ERR_NOT_READY, "object_covariances: cameras not set");
  for (int64_t i = 0; i < n_pairs; ++i) if ((int64_t)obj_a[i] >= h->O || (int64_t)obj_b[i] >= h->O) return fail(h, OBVI_ERR_OUT_OF_RANGE, "object_covariances: object index out of range");
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  { const int vrc = validate_indices(h); if (vrc != OBVI_OK) return vrc; }
  prepare(h);
  std::fill(cov49, cov49 + 49 * n_pairs, 0.0);
  if (n_pairs == 0 || h->nOv == 0 || h->m == 0) return OBVI_OK;
  if (h->allreduce != nullptr && !h->h_shared_ov.empty()) return fail(h, OBVI_ERR_INVALID_ARGUMENT, "object_covariances: not available with objects shared across ranks");
  // the undamped reduced system S = J_c^T J_c - (Schur complement of the features) at the current point, factorised: one
  // LM step's linearisation and factorisation with the trust-region radius at infinity (its candidate point is not used)
  struct QuietStep {   // no phase events for this step; the caches of the LM loop do not survive it (also when a launch throws)
    obvi_ba_handle* h; int profiling;
    explicit QuietStep(obvi_ba_handle* hh) : h(hh), profiling(hh->profiling) { h->profiling = 0; h->pc_valid = false; h->tiles_cleared = false; }
    ~QuietStep() { h->profiling = profiling; h->pc_valid = false; h->tiles_cleared = false; }
  };
  upload_parameter_prior_diagonals(h);
  { QuietStep quiet(h); h->use_extra = !h->h_pp_kind.empty(); try { submit_step(h, 1e300, true, true, /*keep_factor=*/true); } catch (...) { h->use_extra = false; throw; } h->use_extra = false; }
  if (h->h_scal[SC_CHOL_FAIL] != 0.0 || h->h_scal[SC_NONFINITE] != 0.0 || !std::isfinite(h->h_scal[SC_STEPSQ]))
    return fail(h, OBVI_ERR_NUMERICAL, "object_covariances: the normal equations are rank deficient at the current estimate");
  hipStream_t s = h->stream;
  const int nslabs = (int)((7 * h->nOv + kTile - 1) / kTile);
  const int64_t ldt = (int64_t)h->nt * kTile, nrhs = (int64_t)nslabs * kTile;   // Y = L^-1 E transposed: [nrhs][ldt]
  std::vector<int32_t> slab_first(nslabs, h->nt);
  for (int64_t w = 0; w < h->nOv; ++w) {
    const int sl0 = (int)(7 * w / kTile), sl1 = (int)((7 * w + 6) / kTile);
    for (int sl = sl0; sl <= sl1; ++sl) slab_first[sl] = std::min(slab_first[sl], h->h_obj_row[w] / kTile);
  }
  std::vector<int32_t> cols(2 * n_pairs), first_row(n_pairs);
  for (int64_t i = 0; i < n_pairs; ++i) {
    const int32_t va = h->h_obj_vid[obj_a[i]], vb = h->h_obj_vid[obj_b[i]];
    cols[2 * i] = va >= 0 && vb >= 0 ? 7 * va : -1; cols[2 * i + 1] = va >= 0 && vb >= 0 ? 7 * vb : -1;
    first_row[i] = va >= 0 && vb >= 0 ? std::max(h->h_obj_row[va], h->h_obj_row[vb]) / kTile * kTile : 0;   // both columns are zero above
  }
  h->d_cov_Y.resize((size_t)(nrhs * ldt));
  OBVI_HIP(hipMemsetAsync(h->d_cov_Y.get(), 0, sizeof(double) * (size_t)(nrhs * ldt), s));
  h->d_cov_slab.upload(slab_first, s); h->d_cov_cols.upload(cols, s); h->d_cov_first.upload(first_row, s);
  h->d_cov_out.resize((size_t)(49 * n_pairs));
  const CholPlan plan = chol_plan(h);
  launch_forward_multi(s, plan, h->d_S.get(), h->d_Linv.get(), h->d_cov_Y.get(), ldt, nslabs, h->d_cov_slab.get(), h->d_obj_row.get(), (int32_t)h->nOv, h->h_row_split.data());
  launch_cov_pairs(s, h->d_cov_Y.get(), ldt, n_pairs, h->d_cov_cols.get(), h->d_cov_first.get(), h->d_cov_out.get());
  OBVI_HIP(hipGetLastError());
  h->d_cov_out.download(cov49, (size_t)(49 * n_pairs), s);
  sync(h);
  for (int64_t i = 0; i < 49 * n_pairs; ++i) if (!std::isfinite(cov49[i])) return fail(h, OBVI_ERR_NUMERICAL, "object_covariances: non-finite covariance");
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_set_parameter_priors(obvi_ba_handle* h, int64_t n, const uint8_t* kind, const uint32_t* block, const uint8_t* param, const double* mean, const double* std_dev) {
  if (!h || n < 0 || (n > 0 && (!kind || !block || !param || !mean || !std_dev))) return fail(h, OBVI_ERR_INVALID_ARGUMENT, "set_parameter_priors: bad arguments");
  OBVI_API_BEGIN
  for (int64_t i = 0; i < n; ++i) {
    const int64_t cnt = kind[i] == 0 ? h->P : kind[i] == 1 ? h->L : kind[i] == 2 ? h->O : -1;
    const int dim = kind[i] == 0 ? 6 : kind[i] == 1 ? 3 : 7;
    if (cnt < 0 || param[i] >= dim) return fail(h, OBVI_ERR_INVALID_ARGUMENT, "set_parameter_priors: unknown block kind or parameter index");
    if ((int64_t)block[i] >= cnt) return fail(h, OBVI_ERR_OUT_OF_RANGE, "set_parameter_priors: index out of range");
    if (!(std_dev[i] > 0.0) || !std::isfinite(std_dev[i]) || !std::isfinite(mean[i])) return fail(h, OBVI_ERR_NUMERICAL, "set_parameter_priors: standard deviation must be positive and finite");
  }
  h->h_pp_kind.assign(kind, kind + n); h->h_pp_block.assign(block, block + n); h->h_pp_param.assign(param, param + n);
  h->h_pp_mean.assign(mean, mean + n); h->h_pp_std.assign(std_dev, std_dev + n);
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_column_sqnorms(obvi_ba_handle* h, double* pose6, double* point3, double* object7) {
  if (!h) return OBVI_ERR_INVALID_ARGUMENT;
  if (!check_ready(h)) return fail(h, OBVI_ERR_NOT_READY, "column_sqnorms: cameras not set");
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  { const int vrc = validate_indices(h); if (vrc != OBVI_OK) return vrc; }
  prepare(h);
  // one linearisation as at iteration 0: the Jacobi scale it stores is s = 1 / (1 + sqrt(c)), c the squared column norm
  {
    struct Quiet { obvi_ba_handle* h; int profiling; explicit Quiet(obvi_ba_handle* hh) : h(hh), profiling(hh->profiling) { h->profiling = 0; h->pc_valid = false; h->tiles_cleared = false; }
                   ~Quiet() { h->profiling = profiling; h->pc_valid = false; h->tiles_cleared = false; } } quiet(h);
    if (h->num_params > 0) submit_step(h, 1e300, true, false);
  }
  std::vector<double> sc((size_t)h->m_canon + 1), sl((size_t)3 * h->L + 1);
  std::vector<int32_t> pose_vid((size_t)h->P + 1), obj_vid((size_t)h->O + 1);
  std::vector<uint8_t> point_var((size_t)h->L + 1);
  hipStream_t s = h->stream;
  if (h->m_canon) h->d_scale.download(sc.data(), (size_t)h->m_canon, s);
  if (h->L) { h->d_scale_l.download(sl.data(), (size_t)3 * h->L, s); h->d_point_var.download(point_var.data(), (size_t)h->L, s); }
  if (h->P) h->d_pose_vid.download(pose_vid.data(), (size_t)h->P, s);
  if (h->O) h->d_obj_vid.download(obj_vid.data(), (size_t)h->O, s);
  sync(h);
  auto colsq = [](double scale) { const double r = 1.0 / scale - 1.0; return r * r; };
  if (pose6) for (int64_t p = 0; p < h->P; ++p) for (int k = 0; k < 6; ++k) pose6[6 * p + k] = pose_vid[p] >= 0 ? colsq(sc[6 * (int64_t)pose_vid[p] + k]) : -1.0;
  if (point3) for (int64_t l = 0; l < h->L; ++l) for (int k = 0; k < 3; ++k) point3[3 * l + k] = point_var[l] ? colsq(sl[3 * l + k]) : -1.0;
  if (object7) for (int64_t o = 0; o < h->O; ++o) for (int k = 0; k < 7; ++k) object7[7 * o + k] = obj_vid[o] >= 0 ? colsq(sc[6 * h->nPv + 7 * (int64_t)obj_vid[o] + k]) : -1.0;
  for (size_t i = 0; i < h->h_pp_kind.size(); ++i) {
    const double w = 1.0 / (h->h_pp_std[i] * h->h_pp_std[i]);
    const int64_t b = h->h_pp_block[i];
    if (h->h_pp_kind[i] == 0 && pose6 && pose_vid[b] >= 0) pose6[6 * b + h->h_pp_param[i]] += w;
    else if (h->h_pp_kind[i] == 1 && point3 && point_var[b]) point3[3 * b + h->h_pp_param[i]] += w;
    else if (h->h_pp_kind[i] == 2 && object7 && obj_vid[b] >= 0) object7[7 * b + h->h_pp_param[i]] += w;
  }
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_solve(obvi_ba_handle* h, const obvi_solver_params* prm, obvi_summary* sum) {
  if (!h || !prm || !sum) return OBVI_ERR_INVALID_ARGUMENT;
  if (!check_ready(h)) return fail(h, OBVI_ERR_NOT_READY, "solve: cameras not set");
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  const double t_start = wall_s();
  std::memset(sum, 0, sizeof(*sum));
  h->iterations.clear();
  { const int vrc = validate_indices(h); if (vrc != OBVI_OK) return vrc; }
  prepare(h);
  h->pc_valid = false; h->tiles_cleared = false;
  const double ms0[3] = {h->phase_ms[PH_POINT_PASS] + h->phase_ms[PH_POSE_PASS] + h->phase_ms[PH_SMALL] + h->phase_ms[PH_DIAG] + h->phase_ms[PH_POSE_CACHE],
                         h->phase_ms[PH_SCHUR] + h->phase_ms[PH_SCHUR_BLOCKS] + h->phase_ms[PH_CHOL] + h->phase_ms[PH_BACKSUB] + h->phase_ms[PH_APPLY], h->phase_ms[PH_COST]};
  hipStream_t s = h->stream;

  // the state at entry: what the caller gets back if the solve ends in FAILURE (Ceres leaves the user's parameter blocks alone
  // when the solution is not usable [Ceres-doc solver.cc])
  copy_current(h, h->d_pose_e, h->d_point_e, h->d_obj_e);
  // fixed cost: residual blocks with only constant parameter blocks
  OBVI_HIP(hipMemsetAsync(h->d_scal.get(), 0, sizeof(double) * SC_COUNT, s));
  launch_pose_cache(s, h->P, h->d_pose.get(), h->d_pc.get(), h->reproj_variant == OBVI_REPROJECTION_ANALYTIC);
  launch_cost(s, blocks_dev(h), reproj_pose_dev(h), small_dev(h), h->d_cams.get(), h->d_pc.get(), h->d_pose.get(), h->d_point.get(), h->d_obj.get(),
              h->d_pc.get(), h->d_pose.get(), h->d_point.get(), h->d_obj.get(), 1, h->d_scal.get());
  if (h->allreduce != nullptr && !h->h_shared_ov.empty() &&
      h->allreduce(h->allreduce_user, h->d_scal.get() + SC_COST_FIXED, 1, 0, s)) return fail(h, OBVI_ERR_HIP, "allreduce hook (fixed cost)");
  OBVI_HIP(hipMemcpyAsync(h->h_scal, h->d_scal.get(), sizeof(double) * SC_COUNT, hipMemcpyDeviceToHost, s));
  sync(h);
  const double fixed_cost = h->h_scal[SC_COST_FIXED];
  sum->fixed_cost = fixed_cost;
  sum->num_parameters_reduced = (int32_t)h->num_params;
  sum->num_residuals_reduced = (int32_t)h->num_residuals;
  sum->reduced_system_size = (int32_t)h->live_rows;

  auto finish = [&](int term, const char* msg) {
    sum->termination_type = term;
    std::snprintf(sum->message, sizeof(sum->message), "%s", msg);
    sum->num_iterations = (int32_t)h->iterations.size();
    sum->final_cost = sum->initial_cost;  // min over iterations: non-monotonic steps [Ceres-doc solver.cc]
    for (const auto& it : h->iterations) sum->final_cost = std::min(sum->final_cost, it.cost);
    sum->is_solution_usable = (term == OBVI_CONVERGENCE || term == OBVI_NO_CONVERGENCE) ? 1 : 0;
    sum->total_time_in_seconds = wall_s() - t_start;
    sum->jacobian_evaluation_time_in_seconds = 1e-3 * (h->phase_ms[PH_POINT_PASS] + h->phase_ms[PH_POSE_PASS] + h->phase_ms[PH_SMALL] + h->phase_ms[PH_DIAG] + h->phase_ms[PH_POSE_CACHE] - ms0[0]);
    sum->linear_solver_time_in_seconds = 1e-3 * (h->phase_ms[PH_SCHUR] + h->phase_ms[PH_SCHUR_BLOCKS] + h->phase_ms[PH_CHOL] + h->phase_ms[PH_BACKSUB] + h->phase_ms[PH_APPLY] - ms0[1]);
    sum->residual_evaluation_time_in_seconds = 1e-3 * (h->phase_ms[PH_COST] - ms0[2]);
  };

  if (h->num_params == 0) {
    sum->initial_cost = fixed_cost;
    obvi_iteration_summary it; std::memset(&it, 0, sizeof(it));
    it.cost = fixed_cost; it.step_is_valid = 1; it.step_is_successful = 1;
    h->iterations.push_back(it);
    finish(OBVI_CONVERGENCE, "Function tolerance reached. No non-constant parameter blocks found.");
    return OBVI_OK;
  }

  // LevenbergMarquardtStrategy / TrustRegionStepEvaluator state
  double radius = prm->initial_trust_region_radius;
  const double max_radius = prm->max_trust_region_radius;
  double decrease_factor = 2.0;
  const double kMinRelDecrease = 1e-3, kMinRadius = 1e-32;
  const int kMaxInvalid = 5, max_nonmono = prm->allow_non_monotonic_steps ? 5 : 0;
  int num_invalid = 0, num_nonmono = 0;
  double x_cost = 0, x_norm = 0, minimum_cost = 0, current_cost = 0, reference_cost = 0, candidate_cost_ev = 0, acc_ref_model = 0, acc_cand_model = 0;
  double best_cost = 0;
  bool have_best = false;
  // The minimum-cost iterate is kept by buffer rotation, not by copying: while it IS the current point (`best_is_current`) an accepted
  // step parks the old current buffers as `best` and takes the superseded best buffers for the next candidate.
  bool best_is_current = false;

  obvi_iteration_summary it; std::memset(&it, 0, sizeof(it));
  bool pending_accept = false;   // `it` is an accepted step waiting for the gradient of its new point
  bool first = true;
  double iter_t0 = wall_s();
  submit_step(h, radius, true, true);

  // loop-top checks of TrustRegionMinimizer::FinalizeIterationAndCheckIfMinimizerCanContinue
  auto push_and_check = [&](obvi_iteration_summary& rec) -> bool {
    rec.trust_region_radius = radius;
    rec.iteration_time_in_seconds = wall_s() - iter_t0;
    iter_t0 = wall_s();
    h->iterations.push_back(rec);
    if (rec.step_is_successful) sum->num_successful_steps++; else sum->num_unsuccessful_steps++;   // iteration 0 counts as a successful step [Ceres-doc]
    if (rec.iteration >= prm->max_num_iterations) { finish(OBVI_NO_CONVERGENCE, "Maximum number of iterations reached."); return false; }
    if (rec.step_is_successful && rec.gradient_max_norm <= prm->gradient_tolerance) { finish(OBVI_CONVERGENCE, "Gradient tolerance reached."); return false; }
    if (radius < kMinRadius) { finish(OBVI_CONVERGENCE, "Minimum trust region radius reached."); return false; }
    return true;
  };

  for (;;) {
    const double* sc = h->h_scal;
    if (first || pending_accept) {
      // results of the linearisation at the (new) current point complete the pending record
      x_cost = sc[SC_COST];
      x_norm = std::sqrt(sc[SC_XSQ]);
      it.cost = x_cost + fixed_cost;
      it.gradient_max_norm = scal_gmax(h);
      it.gradient_norm = std::sqrt(sc[SC_GSQ]);
      if (first) {
        sum->initial_cost = it.cost;
        it.iteration = 0; it.step_is_valid = 1; it.step_is_successful = 1;
        minimum_cost = current_cost = reference_cost = candidate_cost_ev = x_cost;
        best_cost = x_cost;
      }
      if (!have_best || x_cost < best_cost) {
        // the minimum-cost iterate is what Ceres hands back [Ceres-doc trust_region_minimizer.cc]
        best_cost = x_cost; have_best = true;
        best_is_current = true;
      }
      first = false; pending_accept = false;
      if (!push_and_check(it)) break;
    }
    const obvi_iteration_summary prev = h->iterations.back();
    std::memset(&it, 0, sizeof(it));
    it.iteration = prev.iteration + 1;

    // ---- ComputeTrustRegionStep outcome ----
    const double model_cost_change = sc[SC_MODEL_CHANGE];
    const bool finite = sc[SC_CHOL_FAIL] == 0.0 && sc[SC_NONFINITE] == 0.0 && std::isfinite(model_cost_change) && std::isfinite(sc[SC_STEPSQ]);
    it.step_is_valid = (finite && model_cost_change > 0.0) ? 1 : 0;
    if (!it.step_is_valid) {
      if (++num_invalid >= kMaxInvalid) {
        h->iterations.push_back(it);
        finish(OBVI_FAILURE, "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps");
        break;
      }
      radius /= decrease_factor; decrease_factor *= 2.0;  // StepIsInvalid
      it.cost = x_cost + fixed_cost; it.gradient_max_norm = prev.gradient_max_norm; it.gradient_norm = prev.gradient_norm;
      if (!push_and_check(it)) break;
      submit_step(h, radius, false, true);
      continue;
    }
    num_invalid = 0;
    double cand_cost = sc[SC_COST_CAND];
    if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
    // ParameterToleranceReached
    it.step_norm = std::sqrt(sc[SC_STEPSQ]);
    if (it.step_norm <= prm->parameter_tolerance * (x_norm + prm->parameter_tolerance)) { finish(OBVI_CONVERGENCE, "Parameter tolerance reached."); break; }
    // FunctionToleranceReached
    it.cost_change = x_cost - cand_cost;
    if (std::fabs(it.cost_change) <= prm->function_tolerance * x_cost) { finish(OBVI_CONVERGENCE, "Function tolerance reached."); break; }
    // TrustRegionStepEvaluator::StepQuality
    {
      const double rel = (current_cost - cand_cost) / model_cost_change;
      const double hist = (reference_cost - cand_cost) / (acc_ref_model + model_cost_change);
      it.relative_decrease = (cand_cost >= std::numeric_limits<double>::max()) ? -std::numeric_limits<double>::max() : std::max(rel, hist);
    }
    if (it.relative_decrease > kMinRelDecrease) {
      // HandleSuccessfulStep: the candidate becomes the current point
      h->d_pose.swap(h->d_pose_c); h->d_point.swap(h->d_point_c); h->d_obj.swap(h->d_obj_c); h->d_pc.swap(h->d_pc_c);   // the candidate's pose cache comes along
      if (best_is_current) {   // the point just left is the best so far: it stays where it is, the old best buffers take the next candidate
        h->d_pose_c.swap(h->d_pose_b); h->d_point_c.swap(h->d_point_b); h->d_obj_c.swap(h->d_obj_b);
        best_is_current = false;
      }
      it.step_is_successful = 1;
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * it.relative_decrease - 1.0, 3));  // StepAccepted
      radius = std::min(max_radius, radius);
      decrease_factor = 2.0;
      current_cost = cand_cost; acc_cand_model += model_cost_change; acc_ref_model += model_cost_change;
      if (cand_cost < minimum_cost) { minimum_cost = cand_cost; num_nonmono = 0; candidate_cost_ev = cand_cost; acc_cand_model = 0.0; }
      else { ++num_nonmono; if (cand_cost > candidate_cost_ev) { candidate_cost_ev = cand_cost; acc_cand_model = 0.0; } }
      if (num_nonmono == max_nonmono) { reference_cost = candidate_cost_ev; acc_ref_model = acc_cand_model; }
      pending_accept = true;
      // gradient (and the next step) at the new point; at the iteration cap only the linearisation is needed
      submit_step(h, radius, false, it.iteration < prm->max_num_iterations);
    } else {
      it.step_is_successful = 0;
      it.cost = cand_cost + fixed_cost;   // HandleUnsuccessfulStep records the CANDIDATE's cost [Ceres-doc trust_region_minimizer.cc]
      it.gradient_max_norm = prev.gradient_max_norm; it.gradient_norm = prev.gradient_norm;
      radius /= decrease_factor; decrease_factor *= 2.0;  // StepRejected
      if (!push_and_check(it)) break;
      submit_step(h, radius, false, true);
    }
  }
  // hand back the minimum-cost iterate; after a FAILURE the state at entry
  if (sum->termination_type == OBVI_FAILURE) restore_from(h, h->d_pose_e, h->d_point_e, h->d_obj_e);
  else if (have_best && !best_is_current) { restore_from(h, h->d_pose_b, h->d_point_b, h->d_obj_b); h->pc_valid = false; }
  sync(h);
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_get_iterations(const obvi_ba_handle* h, obvi_iteration_summary* out, int32_t cap) {
  if (!h || !out) return 0;
  const int n = std::min<int>(cap, (int)h->iterations.size());
  for (int i = 0; i < n; ++i) out[i] = h->iterations[i];
  return n;
}

namespace {
// the selection over `n` block norms on the device (select_kernels.hip), the mask into the caller's memory: ONE wait for the device
// unless the threshold route has to hand over to the sort (more than 4096 distinct values sharing their top 24 bits)
void run_selection(obvi_ba_handle* h, int64_t n, const double* sq, const uint8_t* act, const uint32_t* inv, double fraction, uint8_t* mask_out, int64_t* num_excluded) {
  h->d_sel_mask.resize((size_t)n + 1);
  const char* sort_env = std::getenv("OBVI_SELECT_SORT");   // read per call: a host (or a test) may set it after the first selection of the process
  const bool sort_route = sort_env != nullptr && std::atoi(sort_env) != 0;   // route (b) always (its check)
  int n_out = 0;
  bool done = false;
  if (!sort_route) {
    const int* result_dev = nullptr;
    OBVI_HIP(select_by_threshold(h->stream, n, sq, act, inv, fraction, h->d_sel_mask.get(), &h->sel_scratch, &result_dev));
    void* pinned_mask = n ? h->staging.take((size_t)n) : nullptr;
    int* pinned_result = static_cast<int*>(h->staging.take(2 * sizeof(int)));
    int pageable_result[2] = {0, 0};
    if (n) OBVI_HIP(hipMemcpyAsync(pinned_mask ? pinned_mask : mask_out, h->d_sel_mask.get(), (size_t)n, hipMemcpyDeviceToHost, h->stream));
    OBVI_HIP(hipMemcpyAsync(pinned_result ? pinned_result : pageable_result, result_dev, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    sync(h);
    const int* result = pinned_result ? pinned_result : pageable_result;
    if (result[1] == 0) {
      if (pinned_mask) std::memcpy(mask_out, pinned_mask, (size_t)n);
      n_out = result[0];
      done = true;
    }
  }
  if (!done) {
    OBVI_HIP(select_outliers_sorted(h->stream, n, sq, act, inv, fraction, h->d_sel_mask.get(), &n_out, &h->sel_scratch));
    h->d_sel_mask.download(mask_out, (size_t)n, h->stream);
    sync(h);
  }
  if (num_excluded) *num_excluded = n_out;
}
}  // namespace

int obvi_ba_select_outliers(obvi_ba_handle* h, int32_t type, double fraction, uint8_t* mask_out, int64_t* num_excluded) {
  if (!h || !mask_out) return OBVI_ERR_INVALID_ARGUMENT;
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));   // (the evaluate below is skipped when the previous call already left the norms: this path allocates and launches as well)
  // un-robustified per-block squared norms at the current estimate (object_pose_graph_optimizer.h:682-693), kept on the device; the
  // runner selects for one factor type after the other (offline_problem_runner.h:769-800): the second call finds them in place
  const uint64_t this_call = h->api_calls;
  if (h->eval_sq_call == 0 || h->eval_sq_call + 1 != this_call) {
    const int rc = obvi_ba_evaluate(h, 0, nullptr, nullptr, nullptr);
    if (rc != OBVI_OK) return rc;
  }
  int64_t off = 0, n = 0;
  const uint8_t* act = nullptr;
  const uint32_t* inv = nullptr;
  switch (type) {
    case OBVI_FACTOR_REPROJECTION:
      off = 0; n = h->n_rp; act = h->d_rp_active.get();
      if (!h->rp_inv_on_device) { h->d_rp_inv.upload(h->h_rp_inv, h->stream); h->rp_inv_on_device = true; }
      inv = h->d_rp_inv.get();
      break;
    case OBVI_FACTOR_BBOX: off = h->n_rp; n = h->n_bb; act = h->d_bb_active.get(); break;
    case OBVI_FACTOR_SHAPE_PRIOR: off = h->n_rp + h->n_bb; n = h->n_sp; act = h->d_sp_active.get(); break;
    case OBVI_FACTOR_LTM_PRIOR: off = h->n_rp + h->n_bb + h->n_sp; n = h->n_lt; act = h->d_lt_active.get(); break;
    case OBVI_FACTOR_REL_POSE: off = h->n_rp + h->n_bb + h->n_sp + h->n_lt; n = h->n_rl; act = h->d_rl_active.get(); break;
    default: return fail(h, OBVI_ERR_INVALID_ARGUMENT, "select_outliers: unknown factor type");
  }
  run_selection(h, n, h->d_eval_sq.get() + off, act, inv, fraction, mask_out, num_excluded);
  h->eval_sq_call = h->api_calls;   // (the evaluate above counted as a call of its own)
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_debug_select(obvi_ba_handle* h, int64_t n, const double* sq, const uint8_t* active, double fraction, uint8_t* mask_out, int64_t* num_excluded) {
  if (!h || n < 0 || (n > 0 && (!sq || !mask_out))) return OBVI_ERR_INVALID_ARGUMENT;
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  DevBuf<double> d_sq; DevBuf<uint8_t> d_act;
  std::vector<uint8_t> ones;
  if (!active) { ones.assign((size_t)n, 1); active = ones.data(); }
  d_sq.upload(sq, (size_t)n, h->stream); d_act.upload(active, (size_t)n, h->stream);
  sync(h);
  run_selection(h, n, d_sq.get(), d_act.get(), nullptr, fraction, mask_out, num_excluded);
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_snapshot(obvi_ba_handle* h) {
  if (!h) return OBVI_ERR_INVALID_ARGUMENT;
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  copy_current(h, h->d_pose_s, h->d_point_s, h->d_obj_s);
  sync(h);
  h->have_snapshot = true;
  return OBVI_OK;
  OBVI_API_END(h)
}
int obvi_ba_restore(obvi_ba_handle* h) {
  if (!h) return OBVI_ERR_INVALID_ARGUMENT;
  if (!h->have_snapshot) return fail(h, OBVI_ERR_NOT_READY, "restore: no snapshot");
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  restore_from(h, h->d_pose_s, h->d_point_s, h->d_obj_s);
  sync(h);
  return OBVI_OK;
  OBVI_API_END(h)
}

static int get_blocks(obvi_ba_handle* h, const DevBuf<double>& d, int64_t n, int dim, double* out) {
  if (!h || (n > 0 && !out)) return OBVI_ERR_INVALID_ARGUMENT;
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  d.download(out, (size_t)n * dim, h->stream);
  sync(h);
  return OBVI_OK;
  OBVI_API_END(h)
}
int obvi_ba_get_poses(obvi_ba_handle* h, double* out) { return h ? get_blocks(h, h->d_pose, h->P, 6, out) : OBVI_ERR_INVALID_ARGUMENT; }
int obvi_ba_get_points(obvi_ba_handle* h, double* out) { return h ? get_blocks(h, h->d_point, h->L, 3, out) : OBVI_ERR_INVALID_ARGUMENT; }
int obvi_ba_get_objects(obvi_ba_handle* h, double* out) { return h ? get_blocks(h, h->d_obj, h->O, 7, out) : OBVI_ERR_INVALID_ARGUMENT; }

int obvi_ba_get_state(obvi_ba_handle* h, double* poses, double* points, double* objects) {
  if (!h) return OBVI_ERR_INVALID_ARGUMENT;
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  // the three copies land in the handle's pinned arena (a copy into pageable memory blocks, one after the other), ONE wait, then out
  struct Part { double* out; const DevBuf<double>* d; size_t n; void* pinned; } parts[3] = {
      {poses, &h->d_pose, (size_t)h->P * 6, nullptr}, {points, &h->d_point, (size_t)h->L * 3, nullptr}, {objects, &h->d_obj, (size_t)h->O * 7, nullptr}};
  for (Part& p : parts) {
    if (!p.out || !p.n) continue;
    p.pinned = h->staging.take(p.n * sizeof(double));
    p.d->download(p.pinned ? static_cast<double*>(p.pinned) : p.out, p.n, h->stream);
  }
  sync(h);
  for (Part& p : parts) if (p.pinned) std::memcpy(p.out, p.pinned, p.n * sizeof(double));
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_set_shared_objects(obvi_ba_handle* h, const uint8_t* is_shared, int32_t rank, int32_t world) {
  if (!h || world < 1 || rank < 0 || rank >= world) return OBVI_ERR_INVALID_ARGUMENT;
  OBVI_API_BEGIN
  if (is_shared) h->h_is_shared.assign(is_shared, is_shared + h->O); else h->h_is_shared.clear();
  h->rank = rank; h->world = world;
  h->dirty = true;
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_set_allreduce(obvi_ba_handle* h, obvi_allreduce_fn fn, void* user) {
  if (!h) return OBVI_ERR_INVALID_ARGUMENT;
  h->allreduce = fn; h->allreduce_user = user;
  return OBVI_OK;
}

int obvi_ba_set_profiling(obvi_ba_handle* h, int32_t level) {
  if (!h) return OBVI_ERR_INVALID_ARGUMENT;
  h->profiling = level;
  return OBVI_OK;
}

int obvi_ba_get_problem_stats(const obvi_ba_handle* h, double* out, int32_t cap) {
  if (!h || !out) return 0;
  int64_t act_rp = 0, act_bb = 0;
  for (uint8_t a : h->h_rp_active) act_rp += a != 0;
  for (uint8_t a : h->h_bb_active) act_bb += a != 0;
  const double v[14] = {(double)h->nPv, (double)h->nOv, (double)h->nLv, (double)h->m_canon, (double)h->nt, (double)h->nblk, (double)(h->npairs + h->npairs_window),
                        (double)h->ntiles, (double)h->n_trsm_jobs, (double)h->n_upd_products, h->chol_flops, (double)act_rp, (double)act_bb,
                        (double)h->nlevels};
  const int n = std::min<int>(cap, 14);
  for (int i = 0; i < n; ++i) out[i] = v[i];
  return n;
}

int obvi_ba_get_kernel_times(const obvi_ba_handle* h, char* names, int32_t names_cap, double* total_ms, int64_t* launches, int32_t cap) {
  if (!h || !names || !total_ms || !launches) return 0;
  int n = 0, off = 0;
  auto put = [&](const char* name, double ms, int64_t cnt) {
    const int len = (int)std::strlen(name);
    if (n >= cap || off + len + 1 > names_cap) return;
    std::memcpy(names + off, name, len + 1);
    off += len + 1;
    total_ms[n] = ms; launches[n] = cnt;
    ++n;
  };
  for (int p = 0; p < PH_COUNT; ++p) put(kPhaseNames[p], h->phase_ms[p], h->phase_launches[p]);
  static const char* kCholNames[CK_COUNT] = {"k_potrf", "k_trsm", "k_update_potrf", "k_backward"};
  for (int k = 0; k < CK_COUNT; ++k) if (h->ck_launches[k] > 0) put(kCholNames[k], h->ck_ms[k], h->ck_launches[k]);
  return n;
}

}  // extern "C"
