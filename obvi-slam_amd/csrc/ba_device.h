// ba_device.h -- HBM layout of one bundle-adjustment problem and the kernel launch entry points.
// See DESIGN.md ("Data layout in HBM") for the rationale.
#ifndef OBVI_BA_DEVICE_H_
#define OBVI_BA_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "ba_math.h"

struct obvi_ba_handle;
namespace obvi {
// what the front-end gating calls (frontend_kernels.hip) need of a handle: its device, its stream, its error slot
hipStream_t handle_stream(obvi_ba_handle* h);
int handle_device(const obvi_ba_handle* h);
int handle_fail(obvi_ba_handle* h, int code, const char* msg);
void make_dev_cam(const double* K4, const double* ext7, DevCam* out);

// Tile edge of the reduced (Schur) system.  The reduced matrix is stored as a grid of
// kTile x kTile fp64 tiles (row-major inside a tile, tiles row-major in the grid); only tiles
// in the lower triangle that are structurally non-zero after symbolic fill are ever touched.
constexpr int kTile = 64;
// k_point_pass: a wavefront assembles the part of the Z storage that belongs to its piece of the observation list (records of 18 doubles + a
// 4-double tail per point) in LDS and writes it out with coalesced 16-byte stores.  The host cuts the pieces (upload.cpp) so that an image never
// exceeds this many doubles: 64 observations + 28 points.  1264 doubles x 4 wavefronts = 40 KB per workgroup, FOUR workgroups per compute unit
// (160 KB of LDS) = 4 wavefronts per SIMD at <= 128 VGPRs; the round-1..4 layout (22 * 64 doubles, any 64 points) stopped at three (round 5).
#ifndef OBVI_POINT_IMAGE_DOUBLES
#define OBVI_POINT_IMAGE_DOUBLES 1264
#endif
constexpr int kPointImageDoubles = OBVI_POINT_IMAGE_DOUBLES;

// Slots of the device scalar block (fp64 unless noted).  One 256-byte D2H copy per LM step.
enum Scalar {
  // slots [SC_COST, SC_SUM_END) are summed across ranks in a multi-GPU solve, SC_GMAX_BITS is maximised
  SC_COST = 0,        // 0.5 sum rho(s) over the reduced program at the linearisation point
  SC_COST_CAND,       // same at the candidate point
  SC_GSQ,             // |g|^2
  SC_XSQ,             // |x|^2 over the reduced program
  SC_STEPSQ,          // |delta|^2
  SC_MODEL_CHANGE,    // -(J d)^T (r + J d / 2)
  SC_CHOL_FAIL,       // (as double) count of non-positive pivots
  SC_NONFINITE,       // (as double) count of non-finite step entries
  SC_WAIT_TIMEOUT,    // (as double) potrf workgroups of k_update_potrf whose wait for the previous level's jobs timed out: a scheduling
                      // failure, not a numerical one -- obvi_ba_solve returns OBVI_ERR_HIP (summed across ranks so that every rank does)
  SC_SUM_END,
  SC_GMAX_BITS = SC_SUM_END,  // max |g_i| (IEEE bits, via integer atomicMax; non-negative doubles order like u64)
  SC_COST_FIXED,      // residual blocks whose every parameter block is constant
  SC_TAIL_ORDER,      // multi-GPU: a 40-bit hash of this rank's order of the shared tail (as double); summed with SC_COST_FIXED at the start of a solve:
                      // every rank must find world x its own value (the tail's tiles are summed across ranks position by position)
  SC_COUNT = 32
};

// Deterministic mode (obvi_ba_options.deterministic).  The sums a solve's decisions are taken from -- costs, |g|^2, |x|^2, |delta|^2, the
// model cost change -- are then not added to the scalar block with fp64 atomics (whose order changes from run to run): workgroup b of
// a kernel stores its partial sum at scal[SC_COUNT + slot * stride + b], and a one-workgroup-per-slot kernel behind it adds them up
// in a fixed order (launch_det_reduce, which refuses a grid larger than the stride).  stride = BlocksDev.deterministic, the room the
// handle gave the slots for the problem it holds (ensure_det_slots in ba_handle.h: the largest grid, a power of two >= 4096).
// What stays atomic in this mode, and why the result is still the same from run to run:
//   - counters (failed pivots, non-finite entries: small integers) and the gradient maximum (integer max): exact in any order;
//   - the fp64 adds into the reduced system's tiles, right-hand side and diagonal (k_schur_window, k_schur_blocks, k_reduced_diag, the
//     small-factor gathers, the tile Cholesky's updates): every address has ONE writer per kernel -- the plan never cuts a strip, a block's
//     pair list, a target tile's products or a tile row over workgroups in this mode (prepare_plan), the small factors go through per-factor
//     scratch and a gather -- and all kernels run on one stream, so the adds reach an address in launch order.  The atomic is then only
//     the instruction the non-deterministic build shares; tests/test_gpu_deterministic.py holds reruns to bit-identity.
constexpr int kDetSlots = 7;
constexpr int64_t kDetMaxStride = 1 << 24;   // 2^24 workgroups per kernel: beyond any problem that fits the device
inline __host__ __device__ int det_slot_of(int sc) {
  return sc == SC_COST ? 0 : sc == SC_COST_CAND ? 1 : sc == SC_GSQ ? 2 : sc == SC_XSQ ? 3 : sc == SC_STEPSQ ? 4 : sc == SC_MODEL_CHANGE ? 5 : sc == SC_COST_FIXED ? 6 : -1;
}
inline __host__ __device__ int det_scalar_of(int slot) {
  return slot == 0 ? SC_COST : slot == 1 ? SC_COST_CAND : slot == 2 ? SC_GSQ : slot == 3 ? SC_XSQ : slot == 4 ? SC_STEPSQ : slot == 5 ? SC_MODEL_CHANGE : SC_COST_FIXED;
}
void launch_det_reduce(hipStream_t s, double* scal, int64_t nblocks, uint32_t scalar_mask /* bit sc: scalar sc was written by the kernel */, int stride);

struct ReprojDev {          // observations sorted by (point, pose): CSC by point
  int64_t n;
  const uint32_t* pose;     // [n]
  const uint32_t* point;    // [n]
  const uint16_t* cam;      // [n]
  const double2* pixel;     // [n]
  const double* sigma;      // [n]
  const uint8_t* active;    // [n]
  const int32_t* yrow;      // [n] first row of the observing pose in the reduced system, -1 if the observation is inactive or the pose constant
  const uint32_t* point_ptr;  // [L+1] offsets into the arrays above
  double huber;
};

struct ReprojPoseDev {      // the same observations sorted by (pose, point): CSR by pose, for the pose-side pass
  int64_t n;
  const uint32_t* point;    // [n]
  const uint16_t* cam;      // [n]
  const double2* pixel;     // [n]
  const double* sigma;      // [n]
  const uint8_t* active;    // [n]
  const uint32_t* pose_ptr; // [P+1]
  double huber;
};

struct BlocksDev {          // parameter blocks + reduced-program bookkeeping
  int64_t P, L, O;
  int64_t nPv, nOv;         // variable+used poses / objects
  int64_t m;                // rows of the tile grid in use (elimination order, nodes padded to tile boundaries)
  const int32_t* pose_row;  // [nPv] reduced pose index -> first row of its 6x6 diagonal block in the tile grid
  const int32_t* obj_row;   // [nOv] reduced object index -> first row of its od x od diagonal block
  const uint8_t* obj_shared; // [nOv] or NULL: object block is shared across ranks (multi-GPU exchange)
  int32_t shared_owner;     // 1 on the rank that contributes the shared objects' diagonal blocks and scalars
  const int32_t* pose_vid;  // [P]  reduced index or -1
  const int32_t* obj_vid;   // [O]
  const uint8_t* point_var; // [L]
  int32_t analytic_rotation; // the pose caches follow the analytic-Jacobian functor (make_pose_cache, ba_math.h)
  int32_t deterministic;     // obvi_ba_options.deterministic: 0, or the stride of the per-workgroup partial sums behind the scalar block that replace the fp64 atomics
  int32_t od;                // parameters of an ellipsoid block: 7 (x y z yaw dx dy dz: the reference's build) or 9 (x y z ax ay az dx dy dz; obvi_ba_options.object_block_size)
};

struct SmallFactorsDev {    // N <= ~3e4 each; arrays in caller order
  int32_t od;               // parameters of an ellipsoid block (BlocksDev.od)
  // bounding boxes
  int64_t n_bb; const uint32_t* bb_obj; const uint32_t* bb_pose; const uint16_t* bb_cam;
  const double* bb_rect; const double* bb_sqrt_inf; const uint8_t* bb_active; double bb_huber, bb_invalid;
  double* bb_blk;                                       // [n_bb][62] ([81] with the 9-parameter block) per-factor diagonal blocks (big problems: k_small_lin_lanes<true> / k_bbox_gather)
  int32_t bb_pairs_unique;                              // no (object, pose) pair occurs twice: the off-diagonal block of a factor is its own
  const uint32_t* bbo_ptr; const uint32_t* bbo_idx;     // factors by object: [O+1], [n_bb]
  const uint32_t* bbp_ptr; const uint32_t* bbp_idx;     // factors by pose:   [P+1], [n_bb]
  // deterministic mode: the priors and the relative-pose factors go the same way -- per-factor blocks into a scratch (slot = shape prior i,
  // then n_sp + LTM prior i, then n_sp + n_lt + relative-pose factor i; kBbBlk doubles: first block | second block), summed per target
  // block in list order by k_small_gather.  Targets: objects [0, O), then poses [O, O + P); entry = 2 slot + side (1: the slot's second block)
  double* sm_blk; const uint32_t* smt_ptr; const uint32_t* smt_idx;
  // shape priors
  int64_t n_sp; const uint32_t* sp_obj; const double* sp_mean; const double* sp_sqrt_inf; const uint8_t* sp_active; double sp_huber;
  // LTM priors
  int64_t n_lt; const uint32_t* lt_obj; const double* lt_mean; const double* lt_sqrt_inf; const uint8_t* lt_active; double lt_huber;
  // relative poses
  int64_t n_rl; const uint32_t* rl_a; const uint32_t* rl_b; const double* rl_t; const double* rl_R; const double* rl_sqrt_inf;
  const uint8_t* rl_active; double rl_huber;
};

struct ReducedDev {         // accumulators of the reduced system
  double* Hdiag;            // pose v: 36 doubles at 36 v; object w: od^2 doubles at 36 nPv + od^2 w (row-major, lower part used)
  double* g;                // [6 nPv + od nOv] gradient J^T r (compact index: pose v at 6v, object w at 6 nPv + od w)
  double* scale;            // same index: Jacobi scaling (fixed at iteration 0)
  double* lam;              // same index: LM damping of the unscaled normal equations
  double* S;                // tile grid
  double* rhs;              // [m_pad] right-hand side -> forward-substituted z
  double* y;                // [m_pad] solution
  int32_t nt;               // tiles per dimension
  const double* extra;      // NULL, or (same compact index as g) an addition to the diagonal of the normal equations: parameter priors of the covariance extraction
};

struct PointDev {           // per eliminated point
  double* Ci;               // [L][6]  inverse Cholesky factor of (Hll + lambda), lower-tri packed (00,10,11,20,21,22)
  double* u;                // [L][3]  Ci * g_l
  double* gl;               // [L][3]  g_l = sum rho' Jl^T r
  double* lam;              // [L][3]  LM damping of the point block (unscaled normal equations)
  double* scale;            // [L][3]  Jacobi scaling
  double* Z;                // [N_r][18] 6x3 row-major:  rho' Jp^T Jl Ci^T
  const double* extra;      // NULL, or [L][3] addition to the diagonal of H_ll (parameter priors of the covariance extraction)
};

// ---- launchers (ba_kernels.hip) --------------------------------------------------------
// obvi_ba_set_reproj: the arrays only the device reads (camera, pixel, sigma), in both observation orders (by point: a -> perm[a]; by pose:
// k -> perm[rq_src[k]]), gathered from the caller's order on the device; raw_cam / raw_sigma may be null (camera 0 / one sigma for all)
void launch_reproj_gather(hipStream_t s, int64_t n, const uint32_t* perm, const uint32_t* rq_src, const uint32_t* rp_point, const uint16_t* raw_cam,
                          const double2* raw_pixel, const double* raw_sigma, double sigma_scalar, uint16_t* cam, double2* pixel, double* sigma,
                          uint32_t* q_point, uint16_t* q_cam, double2* q_pixel, double* q_sigma, uint8_t* q_active);
void launch_pose_cache(hipStream_t s, int64_t P, const double* poses, PoseCache* out, int analytic /* obvi_ba_options.reprojection_variant == OBVI_REPROJECTION_ANALYTIC */);
void launch_point_pass(hipStream_t s, const BlocksDev& b, const ReprojDev& rp, const DevCam* cams, const PoseCache* pc, const double* points,
                       const ReducedDev& rd, const PointDev& pt, double radius, int first_iter, double* scal, const uint32_t* wave_obs, int64_t n_waves,
                       const uint32_t* long_points, int64_t n_long);
void launch_pose_pass(hipStream_t s, const BlocksDev& b, const ReprojPoseDev& rq, const DevCam* cams, const PoseCache* pc,
                      const double* points, const ReducedDev& rd);
void launch_small_factors(hipStream_t s, const BlocksDev& b, const SmallFactorsDev& sf, const DevCam* cams,
                          const double* poses, const double* objects, const ReducedDev& rd, double* scal);
void launch_reduced_diag(hipStream_t s, const BlocksDev& b, const double* poses, const double* objects,
                         const ReducedDev& rd, double radius, int first_iter, double* scal);
void launch_schur_blocks(hipStream_t s, int64_t nblk, const uint32_t* blk_row, const uint32_t* blk_col, const uint32_t* blk_ptr,
                         const uint32_t* pair_a, const uint32_t* pair_b, const uint32_t* obs_point, const PointDev& pt,
                         const ReducedDev& rd);
// strip geometry of k_schur_window, shared with the host code that builds the visit lists and decides which pairs
// the strip covers: row chunks of kSchurRows frames, columns = the kSchurWindowFrames frames ending with the chunk, cut
// into groups of kSchurGroupCols 16-wide tile columns; visits are streamed through LDS in batches of at most
// kSchurBatchVisits visits / kSchurBatchBytes of 144-byte slots
constexpr int kSchurRows = 8, kSchurWindowFrames = 40, kSchurGroupCols = 5, kSchurBatchBytes = 32768, kSchurBatchVisits = 128;
// the slot tables of the strip kernel, filled on the device (plan_kernels.hip): per visit the point, its tile bits of the (chunk, group) work list
// (bit 15: two records per frame), the slot its image starts at inside its batch and the offset of its slots from the first slot of its workgroup
struct PlanVisit { uint32_t l; uint16_t twin_bits; uint16_t base; uint32_t rel; };
void launch_plan_visit_slots(hipStream_t s, int64_t nvis, const PlanVisit* pv, const uint32_t* wg_ptr, const uint32_t* wg_slot0, int32_t nwg, const int32_t* wg_f0, const int32_t* wg_group,
                             const uint32_t* point_ptr, const uint8_t* rp_active, const uint32_t* rp_pose, const int32_t* frame_of_pose, uint32_t zero16, uint32_t* visits, uint32_t* slot_src);
void launch_schur_window(hipStream_t s, int64_t nwg, int has_twins, const BlocksDev& b, const PointDev& pt, const ReducedDev& rd, const int32_t* row_of_nat,
                         const uint32_t* wg_bptr, const uint32_t* bfirst, const uint32_t* bslot, const uint32_t* visits, const uint32_t* slot_src,
                         const int32_t* wg_f0, const int32_t* wg_group);
// point back-substitution and the candidate poses / objects (with the candidate's pose cache, both layouts of launch_pose_cache), one launch
void launch_backsub_apply(hipStream_t s, const BlocksDev& b, const ReprojDev& rp, const PointDev& pt, const ReducedDev& rd, const double* points,
                          double* points_cand, const double* poses, const double* objects, double* poses_cand, double* objects_cand, PoseCache* pc_cand, double* scal);
// trial-point cost + model cost change.  mode 0: cost at (poses,points,objects) into SC_COST_CAND and
// model change of the step (cand - current); mode 1: cost only, split into SC_COST / SC_COST_FIXED.
void launch_cost(hipStream_t s, const BlocksDev& b, const ReprojPoseDev& rq, const SmallFactorsDev& sf, const DevCam* cams,
                 const PoseCache* pc_cur, const double* poses_cur, const double* points_cur, const double* objects_cur,
                 const PoseCache* pc_cand, const double* poses_cand, const double* points_cand, const double* objects_cand,
                 int mode, double* scal);
// problem->Evaluate: raw / robustified residuals of every factor in caller order
void launch_evaluate(hipStream_t s, const BlocksDev& b, const ReprojDev& rp, const uint32_t* rp_perm, const SmallFactorsDev& sf,
                     const DevCam* cams, const PoseCache* pc, const double* poses, const double* points, const double* objects,
                     int apply_loss, double* residuals, double* sqnorm, double* scal);
void launch_debug_linearize_reproj(hipStream_t s, const ReprojDev& rp, const uint32_t* rp_perm, const DevCam* cams,
                                   const PoseCache* pc, const double* points, double* r, double* J0, double* J1);
void launch_debug_linearize_small(hipStream_t s, int factor_type, const SmallFactorsDev& sf, const DevCam* cams,
                                  const double* poses, const double* objects, double* r, double* J0, double* J1);
void launch_fill(hipStream_t s, double* p, int64_t n, double v);
// dst row i = src row map[i], rows of three doubles (features between the caller's numbering and the internal one)
void launch_permute_rows3(hipStream_t s, double* dst, const double* src, const uint32_t* map, int64_t n);
// multi-GPU exchange buffers: shared objects' (Hdiag 49 | g 7) and the trailing tiles [t0, nt) + rhs rows
void launch_pack_shared_blocks(hipStream_t s, const BlocksDev& b, const ReducedDev& rd, const int32_t* shared_ov, int32_t n_shared, double* buf, int unpack);
void launch_pack_tail(hipStream_t s, const ReducedDev& rd, int32_t t0, double* buf, int unpack);
// scalar sums [SC_COST, SC_SUM_END) + one gradient-maximum slot per rank: (SC_SUM_END - SC_COST) + world doubles, one all-reduce (sum)
void launch_pack_scalars(hipStream_t s, double* scal, double* buf, int32_t rank, int32_t world, int unpack);

// ---- outlier selection (select_kernels.hip) ---------------------------------------------
struct SelectScratch {
  void *tbl = nullptr, *aux = nullptr;   // hash table (keys, then representatives); histograms, counters, result, carry between rounds, candidates
  size_t tbl_slots = 0;
};
// launches only; *result_dev -> {number excluded, r} on the device: r = 0 finished, r > 0: call again with round = r (more than 4096 distinct values in the last bin; r <= 2)
hipError_t select_by_threshold(hipStream_t s, int64_t n, const double* sq, const uint8_t* active, const uint32_t* inv, double fraction,
                               uint8_t* mask_out, SelectScratch* scratch, const int** result_dev, int round = 0);
void select_scratch_free(SelectScratch* sc);

// ---- tile Cholesky (chol_kernels.hip) --------------------------------------------------
// Symbolic factorisation at tile granularity, level-scheduled: tile columns in one level of the
// tile elimination tree do not depend on each other and are processed by the same launches.
// *_ptr arrays indexed by level live on the host (they size the launches); the rest is on the device.
struct CholPlan {
  int32_t nt;
  int32_t nlevels;
  int32_t deterministic;      // column-oriented backward substitution without atomics (one launch per level)
  int32_t fused_potrf;        // 1: k_update_potrf (a level's updates + the next level's potrf in one grid, potrf workgroups wait for their jobs); 0: two launches
  const int32_t* lvl_k_ptr;   // host [nlevels+1]   tile columns of each level
  const int32_t* lvl_k;       // device
  const int32_t* trsm_ptr;    // host [nlevels+1]   trsm jobs (i,k), k in the level
  const int32_t* trsm_ik;     // device, 2 per job
  const int32_t* upd_ptr;     // host [nlevels+1]   update jobs: one per target tile (i,j) and level
  const int32_t* upd_ij;      // device, 2 per job
  const int32_t* upd_kptr;    // device [jobs+1]    the level's columns k contributing to the target
  const int32_t* upd_k;       // device
  const uint8_t* upd_flag;    // device [jobs]      1: the target's k-list is split over several jobs -> atomic accumulation
  const int32_t* rh_ptr;      // host [nlevels+1]   right-hand-side jobs: one per target tile row i and level
  const int32_t* rh_i;        // device
  const int32_t* rh_kptr;     // device [jobs+1]
  const int32_t* rh_k;        // device
  const int32_t* job_signal;  // device [update jobs + rhs jobs, per level: updates then rhs]  tile column of the next level whose potrf waits for the job, or -1
  const int32_t* k_need;      // device, parallel to lvl_k: number of jobs of the previous level that column's potrf waits for
  const int32_t* pre_ptr;     // device, parallel to lvl_k (+1): tile columns j of the previous level whose product L_kj L_kj^T (and L_kj z_j)
  const int32_t* pre_j;       // device                  the column's potrf workgroup applies itself (no job, no wait)
  int32_t* diag_done;         // device [nt] counters
  const int32_t* crit_upd;    // host [nlevels]     number of signalling update / right-hand-side jobs of the level (first in their lists)
  const int32_t* crit_rh;     // host [nlevels]
  const int32_t* slices;      // host [nlevels]     1, or 4 on thin levels (update / trsm tile products split into four row slices)
  const int32_t* col_ptr;     // device [nt+1]      column structure of L: rows i > k with L(i,k) != 0
  const int32_t* col_i;       // device
  int32_t nbw;                // launches of the backward substitution (several levels each)
  const int32_t* bw_ptr;      // host [nbw+1]       backward substitution workgroups of each launch, in the order of the forward levels
  const int32_t* bw_kj;       // device, 3 per workgroup: tile (k,j) of row k, j < k (j = -1: the workgroup that stores y_k), offset of row k's chain record
  const int32_t* bw_chains;   // device             chain records: n, the n ancestors of the row inside the launch (top first), 2 words of tile presence bits
  const int32_t* row_ptr;     // device [nt+1]      row structure of L: columns j < k with L(k,j) != 0 (forward substitution with many right-hand sides)
  const int32_t* row_j;       // device
};
void launch_copy3(hipStream_t s, double* d0, const double* s0, int64_t n0, double* d1, const double* s1, int64_t n1, double* d2, const double* s2, int64_t n2);
// small accumulators cleared at the start of an LM step, together with the tiles (one launch)
struct StepClear {
  double* hdiag; int64_t n_hdiag;
  double* g; int64_t n_g;
  double* rhs; int64_t n_rhs;
  int32_t* diag_done; int64_t n_done;   // potrf counters of k_update_potrf
  double* scal; int64_t n_scal;         // scalar block; scal[fixed_slot] = fixed_cost
  int64_t fixed_slot; double fixed_cost;
  int64_t n_max;
  double* pub_host; double pub_seq;      // not null: the workgroup that clears the scalar block first writes it to this pinned page, then pub_seq behind it (the LM loop's read-back)
};
void launch_zero_tiles(hipStream_t s, double* S, int32_t nt, const int32_t* tile_list, int32_t ntiles, const uint8_t* is_pad_row, const StepClear& c);
// optional per-kernel timing (profiling level 2): an event is recorded after every launch, tagged with the kernel class
enum CholKernel { CK_POTRF = 0, CK_TRSM, CK_UPDATE, CK_BACKWARD, CK_COUNT };
struct CholTimers {
  std::vector<hipEvent_t>* pool;   // grown on demand
  std::vector<int>* tags;          // tag of the kernel that ENDS at event i (event 0 = start, tag -1)
  int used;
};
void launch_cholesky_factor(hipStream_t s, const CholPlan& plan, int level0, int level1, double* S, double* Linv, double* rhs, double* scal, CholTimers* timers = nullptr);
void launch_cholesky_backward(hipStream_t s, const CholPlan& plan, const double* S, const double* Linv, double* rhs /* z in, overwritten */, double* y, CholTimers* timers = nullptr);

// covariance blocks: Y = L^-1 E for the unit vectors of every variable object's rows, kept transposed (Yt row-major
// [64 nslabs][ldt = 64 nt], cleared by the caller), then od x od blocks Yt[ca..] Yt[cb..]^T for pairs of row offsets (cols: 2 per
// pair, negative: zero block)
void launch_forward_multi(hipStream_t s, const CholPlan& plan, const double* S, const double* Linv, double* Yt, int64_t ldt, int nslabs,
                          const int32_t* slab_first, const int32_t* obj_row, int32_t nOv, const int32_t* row_split /* host [nlevels]: workgroups per row, or null */, int od = 7);
void launch_cov_pairs(hipStream_t s, const double* Yt, int64_t ldt, int64_t n_pairs, const int32_t* cols, const int32_t* first_row, double* out, int od = 7);

}  // namespace obvi
#endif  // OBVI_BA_DEVICE_H_
