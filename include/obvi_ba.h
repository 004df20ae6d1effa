/*
 * obvi_ba.h -- C ABI of libobvi_ba, the MI355X (gfx950) bundle-adjustment backend
 * that replaces the Ceres path behind ObVi-SLAM's optimisation entry points.
 *
 * Boundary being replaced (reference file:line, all under /root/reference):
 *   - ObjectPoseGraphOptimizer::buildPoseGraphOptimization
 *       include/refactoring/optimization/object_pose_graph_optimizer.h:126-632
 *     -> obvi_ba_set_* (flat problem upload) + obvi_ba_set_active_mask
 *   - ObjectPoseGraphOptimizer::solveOptimization  (ceres::Solve, SPARSE_SCHUR, HuberLoss, LM)
 *       include/refactoring/optimization/object_pose_graph_optimizer.h:634-707
 *     -> obvi_ba_solve  (+ obvi_ba_evaluate for the apply_loss_function=false pass at :682-693)
 *   - pose_graph->makeCopyDeepCopyValues / setValuesFromAnotherPoseGraph
 *       include/refactoring/optimization/object_pose_graph.h:1025-1121
 *     -> obvi_ba_snapshot / obvi_ba_restore
 *   - two-phase outlier selection  include/refactoring/offline/offline_problem_runner.h:689-800
 *     -> obvi_ba_select_outliers
 *   - long-term-map covariance extraction (ceres::Covariance::Compute + GetCovarianceBlock on object blocks)
 *       src/refactoring/long_term_map/long_term_object_map_extraction.cpp:419-433,
 *       include/refactoring/long_term_map/long_term_object_map_extraction.h:318-340, 499-513
 *     -> obvi_ba_object_covariances
 *
 * Conventions
 *   - every pointer argument is a HOST pointer owned by the caller; set_* copies to
 *     the device, get_* copies back.  No torch / HIP types cross this boundary.
 *   - all floating point is IEEE binary64 (the reference is fp64 throughout,
 *     include/refactoring/types/vslam_basic_types_refactor.h:42).
 *   - return value: 0 on success, negative obvi_status on error.  Nothing throws or aborts.
 *   - one handle == one GPU == one HIP stream.  A handle is not re-entrant.
 *   - parameter block layouts are the reference's raw blocks:
 *       pose   [tx ty tz ax ay az]      (vslam_types_conversion.h:13-21)
 *       point  [x y z]
 *       object [x y z yaw dx dy dz]     (vslam_obj_opt_types_refactor.h:15-21,
 *                                        CONSTRAIN_ELLIPSOID_ORIENTATION=ON, CMakeLists.txt:8-15)
 *   - camera extrinsics are T_robot<-camera as [qx qy qz qw tx ty tz]
 *     (reprojection_cost_functor.h:155-158); intrinsics [fx fy cx cy].
 */
#ifndef OBVI_BA_H_
#define OBVI_BA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct obvi_ba_handle obvi_ba_handle;

typedef enum {
  OBVI_OK = 0,
  OBVI_ERR_INVALID_ARGUMENT = -1,
  OBVI_ERR_NO_DEVICE = -2,       /* HIP runtime/device unavailable: the product never falls back to CPU */
  OBVI_ERR_HIP = -3,             /* a HIP API call failed, a host exception, or a scheduling time-out inside the tile Cholesky; see obvi_ba_last_error */
  OBVI_ERR_OUT_OF_RANGE = -4,    /* an index array refers to a block that was not uploaded */
  OBVI_ERR_NOT_READY = -5,       /* solve/evaluate before the problem was uploaded */
  OBVI_ERR_NUMERICAL = -6        /* non-finite value / non-SPD matrix in a set_* covariance */
} obvi_status;

/* Factor type ids: identical to the reference's FactorType constants
 * (low_level_feature_pose_graph.h:18-23, object_pose_graph.h:18-20). */
enum {
  OBVI_FACTOR_REPROJECTION = 0,  /* kReprojectionErrorFactorTypeId  */
  OBVI_FACTOR_BBOX = 2,          /* kObjectObservationFactorTypeId  */
  OBVI_FACTOR_SHAPE_PRIOR = 3,   /* kShapeDimPriorFactorTypeId      */
  OBVI_FACTOR_LTM_PRIOR = 4,     /* kLongTermMapFactorTypeId        */
  OBVI_FACTOR_REL_POSE = 5       /* kPairwiseRobotPoseFactorTypeId  */
};

/* Residual block sizes per factor type: offline_problem_runner.h:697-719. */
enum {
  OBVI_RESIDUAL_DIM_REPROJECTION = 2,
  OBVI_RESIDUAL_DIM_BBOX = 4,
  OBVI_RESIDUAL_DIM_SHAPE_PRIOR = 3,
  OBVI_RESIDUAL_DIM_LTM_PRIOR = 7,
  OBVI_RESIDUAL_DIM_REL_POSE = 6
};

/* Which of the reference's two reprojection functors the handle evaluates (residual_creator.h:251-264 instantiates the first;
 * the second is the symforce-generated one north_star names). */
enum {
  OBVI_REPROJECTION_AUTODIFF = 0,  /* ReprojectionCostFunctor (reprojection_cost_functor.h:56-93, vslam_math_util.h:347-394): no depth clamp,
                                      constant rotation (zero d/d aa) below |aa| = 1e-8 -- what the reference binary runs */
  OBVI_REPROJECTION_ANALYTIC = 1   /* ReprojectionCostFunctorAnalyticJacobian::Evaluate (reprojection_cost_functor_analytic_jacobian.h:59-566 ==
                                      symforce/reprojectionResidual_with_jacobians012.h:37-521): depth z <- max(z, 1e-15) (:160) with the sign-gated
                                      derivative of the clamp (:289-292), rotation smooth through aa = 0 (epsilon inside the norm, :63-70) */
};

typedef struct {
  int32_t device_id;             /* HIP device ordinal (LOCAL_RANK for one-process-per-GPU) */
  int32_t object_block_size;     /* parameters of an ellipsoid block, called `od` below.  0 or 7: (x y z yaw dx dy dz), the reference's build
                                  * (CONSTRAIN_ELLIPSOID_ORIENTATION, CMakeLists.txt:8-15).  9: (x y z ax ay az dx dy dz), the unconstrained block of
                                  * vslam_obj_opt_types_refactor.h:15-21 / ellipsoid_utils.h:217-229 (`#else`), rotation VectorToAxisAngle(ax ay az).  Every
                                  * "[7]" / "[49]" of an object in this header reads [od] / [od*od] on a handle created with 9. */
  int32_t reprojection_variant;  /* OBVI_REPROJECTION_* (0 = the production functor) */
  int32_t deterministic;         /* != 0: every cross-workgroup accumulation of a solve runs in a fixed order (no floating-point atomics whose
                                    order can change between runs): two solves of the same problem are bit-identical, as Ceres is at a fixed
                                    num_threads (object_pose_graph_optimizer.h:664).  Slower; for parity and regression runs. */
  int32_t reserved[4];
} obvi_ba_options;

/* Mirrors pose_graph_optimization::OptimizationSolverParams
 * (optimization_solver_params.h:10-30) -- the knobs solveOptimization copies into
 * ceres::Solver::Options (object_pose_graph_optimizer.h:662-672).  Everything else is
 * the Ceres default: LM trust region, Jacobi scaling, min/max LM diagonal 1e-6/1e32,
 * min_relative_decrease 1e-3, max_consecutive_nonmonotonic_steps 5. */
typedef struct {
  int32_t max_num_iterations;
  int32_t allow_non_monotonic_steps;
  double function_tolerance;
  double gradient_tolerance;
  double parameter_tolerance;
  double initial_trust_region_radius;
  double max_trust_region_radius;
} obvi_solver_params;

/* ceres::TerminationType order (CONVERGENCE, NO_CONVERGENCE, FAILURE). */
enum { OBVI_CONVERGENCE = 0, OBVI_NO_CONVERGENCE = 1, OBVI_FAILURE = 2 };

/* Fields of ceres::IterationSummary consumed by IterationLogger
 * (include/debugging/optimization_logger.h:58-81). */
typedef struct {
  int32_t iteration;
  int32_t step_is_valid;
  int32_t step_is_successful;
  int32_t reserved;
  double cost;                /* accepted step: cost of the new point; rejected step: cost of the CANDIDATE; invalid step: cost of the current point (+ fixed cost) */
  double cost_change;
  double gradient_max_norm;
  double gradient_norm;
  double step_norm;
  double relative_decrease;
  double trust_region_radius;
  double iteration_time_in_seconds;
} obvi_iteration_summary;

/* Fields of ceres::Solver::Summary consumed by OptimizationLogger
 * (include/debugging/optimization_logger.h:192-203) and solveOptimization (:698-706). */
typedef struct {
  int32_t termination_type;
  int32_t is_solution_usable;        /* Summary::IsSolutionUsable(); 0 (FAILURE): the parameter blocks are handed back as they were at entry */
  int32_t num_iterations;            /* == iterations.size(): includes the iteration-0 record */
  int32_t num_successful_steps;      /* iteration 0 counts as a successful step (successful + unsuccessful == num_iterations) */
  int32_t num_unsuccessful_steps;
  int32_t num_parameters_reduced;    /* scalar parameters in non-constant, used blocks */
  int32_t num_residuals_reduced;
  int32_t reduced_system_size;       /* rows of the Schur complement (poses + objects) */
  double initial_cost;
  double final_cost;
  double fixed_cost;
  double total_time_in_seconds;
  double linear_solver_time_in_seconds;
  double jacobian_evaluation_time_in_seconds;
  double residual_evaluation_time_in_seconds;
  char message[160];
} obvi_summary;

/* ---- lifetime ---------------------------------------------------------------------- */
int obvi_ba_create(const obvi_ba_options* options, obvi_ba_handle** out);
void obvi_ba_destroy(obvi_ba_handle* h);
/* Back to the state obvi_ba_create left -- no cameras, blocks, factors or parameter priors, nothing shared (rank 0 of 1) and no exchange
 * hook, no snapshot, no iteration records, profiling off -- with the device memory, streams and pinned pages kept: what a pool of handles
 * calls before it hands one to its next user (the reference builds a fresh ceres::Problem per optimisation, offline_problem_runner.h:164-166;
 * creating a handle costs ~25 ms). */
int obvi_ba_reset(obvi_ba_handle* h);
const char* obvi_ba_last_error(const obvi_ba_handle* h);
const char* obvi_ba_version(void);

/* ---- parameter blocks  (getPosePointers low_level_feature_pose_graph.h:315-321,
 *      getFeaturePointers :692-700, getObjectParamPointers object_pose_graph.h:495-503;
 *      constness: object_pose_graph_optimizer.h:424-472, 532-603) --------------------- */
int obvi_ba_set_cameras(obvi_ba_handle* h, int32_t n, const double* fx_fy_cx_cy /*[n][4]*/,
                        const double* ext_qxyzw_t /*[n][7]*/);
int obvi_ba_set_poses(obvi_ba_handle* h, int64_t n, const double* pose6 /*[n][6]*/,
                      const uint8_t* is_const /*[n] or NULL = all variable*/);
int obvi_ba_set_points(obvi_ba_handle* h, int64_t n, const double* xyz /*[n][3]*/,
                       const uint8_t* is_const);
int obvi_ba_set_objects(obvi_ba_handle* h, int64_t n, const double* ell7 /*[n][od]*/,
                        const uint8_t* is_const);
/* change constness without re-uploading values (window slide / PGO stage) */
int obvi_ba_set_const_flags(obvi_ba_handle* h, const uint8_t* pose_const, const uint8_t* point_const,
                            const uint8_t* object_const /* any may be NULL = unchanged */);

/* ---- factors ----------------------------------------------------------------------- */
/* ReprojectionErrorFactor -> ReprojectionCostFunctor (reprojection_cost_functor.h:56-93,
 * .cpp:5-17) wrapped in HuberLoss(huber) (residual_creator.h:251-264). */
int obvi_ba_set_reproj(obvi_ba_handle* h, int64_t n, const uint32_t* pose_idx, const uint32_t* point_idx,
                       const uint16_t* cam_idx, const double* pixel_xy /*[n][2]*/,
                       const double* sigma /*[n] or NULL*/, double sigma_scalar, double huber);
/* ObjectObservationFactor -> BoundingBoxFactor (bounding_box_factor.h:68-136, .cpp:7-40);
 * corners (min_x,max_x,min_y,max_y) px (vslam_obj_opt_types_refactor.h:184-191). */
int obvi_ba_set_bbox(obvi_ba_handle* h, int64_t n, const uint32_t* obj_idx, const uint32_t* pose_idx,
                     const uint16_t* cam_idx, const double* corners /*[n][4]*/,
                     const double* cov /*[n][16] row-major*/, double huber, double invalid_ellipse_error);
/* ShapeDimPriorFactor -> ShapePriorFactor (shape_prior_factor.h:46-61, .cpp:7-11). */
int obvi_ba_set_shape_priors(obvi_ba_handle* h, int64_t n, const uint32_t* obj_idx,
                             const double* mean3, const double* cov9, double huber);
/* LTM prior -> IndependentObjectMapFactor (independent_object_map_factor.h:21-33, .cpp:7-11;
 * long_term_map_factor_creator.h:265-322). */
int obvi_ba_set_ltm_priors(obvi_ba_handle* h, int64_t n, const uint32_t* obj_idx,
                           const double* mean7 /*[n][od]*/, const double* cov49 /*[n][od*od]*/, double huber);
/* RelPoseFactor -> RelativePoseFactor (relative_pose_factor.h:32-61, .cpp:7-19); measured
 * relative pose given as translation + axis-angle vector of Pose3D::orientation_. */
int obvi_ba_set_relpose(obvi_ba_handle* h, int64_t n, const uint32_t* pose_idx_a,
                        const uint32_t* pose_idx_b, const double* t3, const double* aa3,
                        const double* cov36, double huber);
/* excluded_feature_factor_types_and_ids (object_pose_graph_optimizer.h:133-135): mask[i]==0
 * drops factor i of that type from the problem; NULL = all active. */
int obvi_ba_set_active_mask(obvi_ba_handle* h, int32_t factor_type, const uint8_t* mask);

/* ---- evaluate / solve -------------------------------------------------------------- */
/* problem->Evaluate(apply_loss_function, ...) (object_pose_graph_optimizer.h:682-693).
 * residuals: concatenation in factor-type order 0,2,3,4,5, each factor its
 * OBVI_RESIDUAL_DIM_* entries, inactive factors written as 0; block_sqnorm: one entry per
 * factor in the same order (the per-block sum of squares the runner recomputes at
 * offline_problem_runner.h:721-733).  Either may be NULL. */
int obvi_ba_evaluate(obvi_ba_handle* h, int32_t apply_loss, double* cost, double* residuals,
                     double* block_sqnorm);
int64_t obvi_ba_num_residuals(const obvi_ba_handle* h);
int64_t obvi_ba_num_factors(const obvi_ba_handle* h, int32_t factor_type);

int obvi_ba_solve(obvi_ba_handle* h, const obvi_solver_params* params, obvi_summary* summary);
/* copies min(cap, summary.num_iterations) records of the last solve; returns the count */
int obvi_ba_get_iterations(const obvi_ba_handle* h, obvi_iteration_summary* out, int32_t cap);

/* per factor type: mask_out[i]=0 for the active factors that carry the (size_t)(d * fraction) largest DISTINCT values of the
 * un-robustified squared residual at the current estimate, d = number of distinct values among the active factors, 1 otherwise:
 * the reference keys a std::map by the residual value, so equal residuals overwrite each other and the count is taken on the
 * de-duplicated set (offline_problem_runner.h:769-800).  Runs on the device. */
int obvi_ba_select_outliers(obvi_ba_handle* h, int32_t factor_type, double fraction,
                            uint8_t* mask_out, int64_t* num_excluded);

/* Covariance blocks of pairs of object (ellipsoid) blocks at the current estimate: cov49[i] = the 7x7 block
 * (rows: obj_a[i], columns: obj_b[i], row-major) of (J^T J)^-1 over all non-constant parameter blocks, J with the
 * loss functions applied and no damping -- what ceres::Covariance::Compute(blocks, problem) followed by
 * GetCovarianceBlock(obj_a, obj_b) returns (long_term_object_map_extraction.cpp:419-433, .h:318-340, 499-513;
 * the independent-ellipsoids extractor asks for (o, o) pairs only).  Blocks of a constant or unobserved object are zero.
 * A rank-deficient problem (free gauge, unobserved feature) fails with OBVI_ERR_NUMERICAL, as Covariance::Compute
 * fails on it.  Computed on the device from the tile Cholesky factor of the undamped reduced system. */
int obvi_ba_object_covariances(obvi_ba_handle* h, int64_t n_pairs, const uint32_t* obj_a, const uint32_t* obj_b,
                               double* cov49 /*[n_pairs][od*od]*/);

/* ParameterPrior (include/refactoring/factors/parameter_prior.h:17-50): residual (block[param_idx] - mean) / std_dev, no loss
 * function.  The reference adds such factors in one place only -- to the problem the long-term-map covariance is extracted from,
 * for parameters whose Jacobian columns are numerically zero, with mean = the current estimate
 * (src/refactoring/long_term_map/long_term_object_map_extraction.cpp:764-927).  Accordingly they take part in
 * obvi_ba_object_covariances and obvi_ba_column_sqnorms (their Jacobian 1 / std_dev in the parameter's column; the residual is zero
 * at the mean) and NOT in obvi_ba_solve / obvi_ba_evaluate.  block_kind: 0 pose, 1 point, 2 object.  n = 0 clears them. */
int obvi_ba_set_parameter_priors(obvi_ba_handle* h, int64_t n, const uint8_t* block_kind, const uint32_t* block_idx,
                                 const uint8_t* param_idx, const double* mean, const double* std_dev);
/* Squared column norms of the robustified Jacobian at the current estimate, one per scalar parameter (what findRankDeficiencies
 * accumulates from the CRS Jacobian, long_term_object_map_extraction.cpp:585-608; parameter priors included); -1 for the parameters
 * of a block that is constant or touched by no active factor.  Any output may be NULL. */
int obvi_ba_column_sqnorms(obvi_ba_handle* h, double* pose6 /*[P][6]*/, double* point3 /*[L][3]*/, double* object7 /*[O][od]*/);

/* ---- state ------------------------------------------------------------------------- */
int obvi_ba_snapshot(obvi_ba_handle* h);
int obvi_ba_restore(obvi_ba_handle* h);
int obvi_ba_get_poses(obvi_ba_handle* h, double* out /*[n][6]*/);
int obvi_ba_get_points(obvi_ba_handle* h, double* out /*[n][3]*/);
int obvi_ba_get_objects(obvi_ba_handle* h, double* out /*[n][od]*/);
/* all three at once (a null pointer skips one): what the reference reads in place after Solve() -- Ceres has updated the pose graph's
 * parameter blocks (object_pose_graph_optimizer.h:664-667) -- with one wait for the device instead of three */
int obvi_ba_get_state(obvi_ba_handle* h, double* poses /*[n][6]*/, double* points /*[n][3]*/, double* objects /*[n][od]*/);
/* overwrite values only (feature re-attachment after PGO,
 * pose_graph_plus_objects_optimizer.h:238-283) */
int obvi_ba_update_points(obvi_ba_handle* h, int64_t n, const double* xyz);
/* the same for all three kinds of block at once (a null pointer keeps one; counts as uploaded): values only -- constness, factors and the
 * symbolic plan stay.  With obvi_ba_prepare this lets a caller upload and plan a window AHEAD, on a second handle and a second host
 * thread, while the previous window is still being solved, and hand over the values once they exist (the start values of a window are
 * the previous window's result: offline_problem_runner.h:183-229).  A snapshot taken before is dropped. */
int obvi_ba_update_state(obvi_ba_handle* h, const double* poses /*[n][6]*/, const double* points /*[n][3]*/, const double* objects /*[n][od]*/);
/* the symbolic phase of what has been uploaded (elimination order, Schur work lists, tile plan), now instead of inside the first
 * obvi_ba_solve / obvi_ba_evaluate: structure only, reads no parameter value (exception: the order of SHARED objects follows the (x, y)
 * given to obvi_ba_set_objects).  Returns when the plan is on the device. */
int obvi_ba_prepare(obvi_ba_handle* h);

/* ---- multi-GPU (no reference counterpart) -------------------------------
 * Independent windows / sessions, one per rank, that share object blocks.  Every rank uploads ALL shared objects
 * (same indices, same initial values) plus its own poses, points and observations; object-only factors (shape / LTM
 * priors) of a shared object are uploaded by exactly one rank.  Shared objects are eliminated last on every rank; per
 * LM step the library calls `fn` three times on the handle's stream (and once at the start of a solve, op SUM on 2 doubles: the job's fixed cost and a hash of this
 * rank's order of the shared objects -- the shared tail follows the objects' UPLOADED positions, so a rank whose shared objects carry other values is refused with
 * OBVI_ERR_INVALID_ARGUMENT instead of having its tiles summed against the wrong objects):
 *   (1) op SUM  on the packed J^T J diagonal blocks and gradients of the shared objects   (n_shared * (od*od + od) doubles: 56 each, 90 with the 9-parameter block)
 *   (2) op SUM  on the trailing shared-object tiles of the reduced system + right-hand side, after the rank's own
 *               poses / points / private objects have been eliminated
 *   (3) op SUM  on the scalar block (costs, model change, step and gradient norms) followed by one slot per rank that carries
 *               that rank's gradient maximum (9 + world doubles), so that every rank takes the same accept / reject decision.
 * (1) is issued on the handle's side stream -- it overlaps the Schur complement of the rank's own blocks --, (2) and (3) on its main
 * stream, always in this order on every rank.  `fn` sums (op 0) or maximises (op 1) `count_f64` doubles in place across ranks --
 * ncclAllReduce on `stream` (a hipStream_t) in a C++ host, torch.distributed.all_reduce from Python.  Non-zero return aborts the solve. */
typedef int (*obvi_allreduce_fn)(void* user, void* device_buf, int64_t count_f64, int32_t op, void* stream);
int obvi_ba_set_allreduce(obvi_ba_handle* h, obvi_allreduce_fn fn, void* user);
/* is_shared[i] != 0: object i is shared across ranks.  rank / world: this handle's position in the job. */
int obvi_ba_set_shared_objects(obvi_ba_handle* h, const uint8_t* is_shared /*[n objects] or NULL*/, int32_t rank, int32_t world);

/* ---- test / profiling hooks (parity tests call these through the C ABI) ------------ */
/* raw (un-robustified) residual and Jacobians of every factor of one type at the current
 * estimate, row-major: J0 w.r.t. the first block of the factor, J1 the second
 *   type 0: r[n][2], J0 = d/dpose [n][2][6], J1 = d/dpoint  [n][2][3]
 *   type 2: r[n][4], J0 = d/dobject [n][4][od], J1 = d/dpose [n][4][6]
 *   type 3: r[n][3], J0 [n][3][od]         type 4: r[n][od], J0 [n][od][od]
 *   type 5: r[n][6], J0 = d/dpose_a [n][6][6], J1 = d/dpose_b [n][6][6]  */
int obvi_ba_debug_linearize(obvi_ba_handle* h, int32_t factor_type, double* r, double* J0, double* J1);
/* the selection rule of obvi_ba_select_outliers (offline_problem_runner.h:769-800) on block norms given by the caller: mask_out[i] = 0
 * for the excluded ones, 1 for the kept active ones, 0 for inactive ones; active may be NULL (all). */
int obvi_ba_debug_select(obvi_ba_handle* h, int64_t n, const double* sq, const uint8_t* active, double fraction, uint8_t* mask_out, int64_t* num_excluded);
/* dense reduced (Schur) system at the current estimate for LM diagonal 1/radius:
 * lhs [m][m] row-major symmetric, rhs [m]; order = variable poses then variable objects. */
int obvi_ba_debug_reduced_system(obvi_ba_handle* h, double radius, double* lhs, double* rhs, int32_t m_cap,
                                 int32_t* m_out);
/* structure of the reduced program after the last evaluate/solve, as doubles:
 * [0] variable poses [1] variable objects [2] eliminated points [3] reduced rows m [4] tiles per dim
 * [5] Schur 6x6 blocks [6] Schur observation pairs [7] non-zero tiles after fill [8] trsm tile jobs
 * [9] update tile jobs [10] flops of one tile-Cholesky factorisation [11] active reprojection obs
 * [12] active bbox obs [13] levels of the tile elimination tree [14] host threads of the symbolic phase (pool + caller; OBVI_HOST_THREADS, default
 * min(16, usable CPUs)) [15] usable CPUs of the process (affinity mask and cgroup quota) [16] LM steps that were re-run because a workgroup of the
 * fused level kernel of the tile Cholesky timed out waiting for its jobs [17] 1 while the handle is on the fused schedule.  Returns the number of entries written. */
int obvi_ba_get_problem_stats(const obvi_ba_handle* h, double* out, int32_t cap);
/* level 0 (default): no events; level 1: one HIP event pair per phase of an LM step (about 40 us of host and device time per
 * iteration: a dozen records and as many elapsed-time queries); level 2: additionally an event after every launch of the
 * tile Cholesky so that obvi_ba_get_kernel_times also lists k_potrf / k_trsm / k_update_potrf / k_backward, everything on
 * one stream (costs a few percent: use a separate, un-timed solve). */
int obvi_ba_set_profiling(obvi_ba_handle* h, int32_t level);
/* per-kernel device timings of the last solve (ms, HIP events on the handle's stream):
 * names is a NUL-separated list; returns number of entries. */
int obvi_ba_get_kernel_times(const obvi_ba_handle* h, char* names, int32_t names_cap, double* total_ms,
                             int64_t* launches, int32_t cap);

/* What this device delivers for the two resources the kernels are priced against (the public peaks "to be re-measured on
 * the box with a triad and a DGEMM microbenchmark"): HBM bandwidth of a triad / copy / read over 1 GiB arrays, and the fp64 matrix rate of
 * v_mfma_f64_16x16x4_f64 -- the bare issue rate (operands in registers) and the 64x64x64 tile product of the tile Cholesky fed from LDS.
 * About 50 ms; allocates 3 GiB for the duration of the call. */
typedef struct {
  double hbm_triad_gbs, hbm_copy_gbs, hbm_read_gbs;
  double mfma_f64_issue_tflops, mfma_f64_tile_tflops;
  double clock_mhz;
  int32_t compute_units;
  int32_t reserved;
} obvi_measured_peaks;
int obvi_ba_measure_peaks(obvi_ba_handle* h, obvi_measured_peaks* out);

#ifdef __cplusplus
}
#endif
#endif /* OBVI_BA_H_ */
