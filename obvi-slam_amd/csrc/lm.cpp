// lm.cpp -- one Levenberg-Marquardt step on the device (submit_step) and the trust-region loop around it (obvi_ba_solve)  (include/obvi_ba.h; shared state and helpers: ba_handle.h)
//
// Trust-region logic: [Ceres-doc] TrustRegionMinimizer / LevenbergMarquardtStrategy / TrustRegionStepEvaluator with the options the reference sets at
// include/refactoring/optimization/object_pose_graph_optimizer.h:651-672 and Ceres defaults otherwise (Jacobi scaling, min/max LM diagonal 1e-6/1e32,
// min_relative_decrease 1e-3, max 5 consecutive non-monotonic / invalid steps, min trust-region radius 1e-32).
#include "ba_handle.h"

namespace obvi_lib {

// Solves running in this process right now (obvi_ba_solve on as many handles, a host thread each).  Window-sized solves are chains of short launches; the runtime's
// four hardware queues carry about four such chains at once, and a handle that forks a side stream takes two of them: measured with K sessions in one process
// (profiles/r05_concurrent_sessions.txt), 8 sessions reach 603 frames/s together with one stream per handle against 481 with two.  So a window-sized step forks its
// side stream only while at most OBVI_SIDE_MAX_SOLVERS solves (default 2) are in flight.
static std::atomic<int> g_active_solves{0};
struct ActiveSolve { ActiveSolve() { g_active_solves.fetch_add(1, std::memory_order_relaxed); } ~ActiveSolve() { g_active_solves.fetch_sub(1, std::memory_order_relaxed); } };

// One LM step on the device: linearise at the current point, assemble and solve the damped reduced
// system, form the candidate, evaluate it.  `solve` false: linearisation only (gradient norms).
void submit_step(obvi_ba_handle* h, double radius, bool first_iter, bool solve, bool keep_factor) {
  ApiTimer api_timer_("  LM step (submit + wait)");
  const double t_submit0 = api_times() ? wall_s() : 0.0;
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
  static const int side_max_solvers = std::getenv("OBVI_SIDE_MAX_SOLVERS") ? std::atoi(std::getenv("OBVI_SIDE_MAX_SOLVERS")) : 2;   // tuning knob (see g_active_solves)
  const int64_t fork_early_below = std::getenv("OBVI_FORK_EARLY_BELOW") ? std::atoll(std::getenv("OBVI_FORK_EARLY_BELOW")) : 400000;   // tuning knob (observations); read per step: the tests flip it
  const bool crowded_window = h->n_rp < fork_early_below && g_active_solves.load(std::memory_order_relaxed) > side_max_solvers;
  const bool side = h->profiling < 2 && side_ok && !h->deterministic && !crowded_window;   // deterministic mode: one stream, so that the kernels that add to the same tiles do so in a fixed order
  hipStream_t s2 = side ? h->stream2 : s;
  // the point pass first, alone: it and the pose-side pass stream the same observation arrays and are both HBM-bound (side by side the
  // point pass took 0.35 ms instead of 0.24); the side stream starts behind it and runs beside the Schur complement, which is bound
  // by instruction issue and LDS, not by HBM
  // ... on a big problem.  On a sliding window every kernel is a few microseconds of latency, nothing is bandwidth-bound, and the side stream
  // (pose pass + small factors + diagonal blocks + far pairs: 63 us for 50 frames) is longer than point pass + Schur complement (47 us): there
  // it forks in front of the point pass.
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
      if (h->allreduce(h->allreduce_user, h->d_xbuf2.get(), (int64_t)(h->od * h->od + h->od) * ns, 0, s2)) throw HipError{hipErrorUnknown, "allreduce hook (shared blocks)", __FILE__, __LINE__};
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
  if (ApiTimes* t = api_times()) t->add("    LM step: host time until everything is enqueued", 1e3 * (wall_s() - t_submit0));   // the part that threads of one process share the runtime for
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

}  // namespace obvi_lib

extern "C" {

int obvi_ba_solve(obvi_ba_handle* h, const obvi_solver_params* prm, obvi_summary* sum) {
  if (!h || !prm || !sum) return OBVI_ERR_INVALID_ARGUMENT;
  if (!check_ready(h)) return fail(h, OBVI_ERR_NOT_READY, "solve: cameras not set");
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  obvi_lib::ActiveSolve active_solve_;
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
  const bool exchanging = h->allreduce != nullptr && !h->h_shared_ov.empty();
  if (exchanging) {
    // the fixed cost of the job, and -- in the same collective -- proof that every rank lays the shared tail out alike: the order follows the shared objects'
    // uploaded positions (plan.cpp), which the contract says are the same on every rank; a host that breaks it must not get tiles summed across positions
    static_assert(SC_TAIL_ORDER == SC_COST_FIXED + 1, "summed together");
    h2d_async(h->d_scal.get() + SC_TAIL_ORDER, &h->tail_order_hash, sizeof(double), s);
    if (h->allreduce(h->allreduce_user, h->d_scal.get() + SC_COST_FIXED, 2, 0, s)) return fail(h, OBVI_ERR_HIP, "allreduce hook (fixed cost)");
  }
  OBVI_HIP(hipMemcpyAsync(h->h_scal, h->d_scal.get(), sizeof(double) * SC_COUNT, hipMemcpyDeviceToHost, s));
  sync(h);
  if (exchanging && h->h_scal[SC_TAIL_ORDER] != (double)h->world * h->tail_order_hash)
    return fail(h, OBVI_ERR_INVALID_ARGUMENT, "solve: the ranks order the shared objects differently -- every rank must upload the shared objects with the same indices and the same values (include/obvi_ba.h, multi-GPU)");
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

}  // extern "C"
