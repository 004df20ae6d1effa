// abi.cpp -- lifetime of a handle, evaluate, debug and profiling hooks, covariance extraction, outlier selection, state read-back, multi-GPU switches  (include/obvi_ba.h; shared state and helpers: ba_handle.h)
//
// One handle == one GPU == one HIP stream pair.  There is no CPU compute path in this library: without a HIP device obvi_ba_create fails with
// OBVI_ERR_NO_DEVICE.
#include "ba_handle.h"

namespace obvi {
hipStream_t handle_stream(obvi_ba_handle* h) { return h->stream; }
int handle_device(const obvi_ba_handle* h) { return h->device; }
int handle_fail(obvi_ba_handle* h, int code, const char* msg) { return fail(h, code, msg); }
void make_dev_cam(const double* K4, const double* ext7, DevCam* out) { make_cam(K4, ext7, out); }
}  // namespace obvi

extern "C" {

const char* obvi_ba_version(void) { return "obvi_ba 0.1 (gfx950)"; }
const char* obvi_ba_last_error(const obvi_ba_handle* h) { return h ? h->err.c_str() : "null handle"; }

int obvi_ba_create(const obvi_ba_options* options, obvi_ba_handle** out) {
  if (!out) return OBVI_ERR_INVALID_ARGUMENT;
  ApiTimer api_timer_(__func__);
  *out = nullptr;
  if (options && options->object_block_size != 0 && options->object_block_size != 7 && options->object_block_size != 9) return OBVI_ERR_INVALID_ARGUMENT;   // 7: yaw only (the reference's build); 9: axis-angle
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
  if (options) { h->reproj_variant = options->reprojection_variant; h->deterministic = options->deterministic != 0; if (options->object_block_size == 9) h->od = 9; }
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
    // The side stream (pose pass, small factor families, diagonal blocks, far pairs: beside the Schur strips).  OBVI_SIDE_CUS = n (tuning knob,
    // round 5 A/B): its queue may only use n of the device's compute units (the first n bits of the CU mask: the driver deals the bits over the
    // XCDs, n / 8 per XCD), leaving the others to the strip kernel alone.  Default: no mask (measured: EXPERIMENTS.md round 5).
    int side_cus = 0;
    if (const char* env = std::getenv("OBVI_SIDE_CUS")) side_cus = std::atoi(env);
    hipDeviceProp_t prop;
    if (side_cus > 0 && hipGetDeviceProperties(&prop, dev) == hipSuccess && side_cus < prop.multiProcessorCount) {
      std::vector<uint32_t> mask((size_t)(prop.multiProcessorCount + 31) / 32, 0u);
      for (int c = 0; c < side_cus; ++c) mask[(size_t)c / 32] |= 1u << (c % 32);
      OBVI_HIP(hipExtStreamCreateWithCUMask(&h->stream2, (uint32_t)mask.size(), mask.data()));
    } else if (const char* pr = std::getenv("OBVI_SIDE_PRIORITY")) {
      // (tuning knob, round 5 A/B) the side stream at another dispatch priority than the main stream: 1 = lowest, -1 = highest the device offers
      int lo = 0, hi = 0;
      OBVI_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      OBVI_HIP(hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, std::atoi(pr) > 0 ? lo : hi));
    } else {
      OBVI_HIP(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    }
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

int obvi_ba_prepare(obvi_ba_handle* h) {
  if (!h) return OBVI_ERR_INVALID_ARGUMENT;
  if (!check_ready(h)) return fail(h, OBVI_ERR_NOT_READY, "prepare: cameras not set");
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  { const int vrc = validate_indices(h); if (vrc != OBVI_OK) return vrc; }
  prepare(h);
  sync(h);   // the plan's uploads have left the host arrays they were issued from
  return OBVI_OK;
  OBVI_API_END(h)
}

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
    case OBVI_FACTOR_BBOX: m = 4; d0 = h->od; d1 = 6; n = h->n_bb; break;
    case OBVI_FACTOR_SHAPE_PRIOR: m = 3; d0 = h->od; d1 = 0; n = h->n_sp; break;
    case OBVI_FACTOR_LTM_PRIOR: m = h->od; d0 = h->od; d1 = 0; n = h->n_lt; break;
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
  const int od = h->od, od2 = od * od;
  std::fill(cov49, cov49 + od2 * n_pairs, 0.0);
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
  const int nslabs = (int)((od * h->nOv + kTile - 1) / kTile);
  const int64_t ldt = (int64_t)h->nt * kTile, nrhs = (int64_t)nslabs * kTile;   // Y = L^-1 E transposed: [nrhs][ldt]
  std::vector<int32_t> slab_first(nslabs, h->nt);
  for (int64_t w = 0; w < h->nOv; ++w) {
    const int sl0 = (int)(od * w / kTile), sl1 = (int)((od * w + od - 1) / kTile);
    for (int sl = sl0; sl <= sl1; ++sl) slab_first[sl] = std::min(slab_first[sl], h->h_obj_row[w] / kTile);
  }
  std::vector<int32_t> cols(2 * n_pairs), first_row(n_pairs);
  for (int64_t i = 0; i < n_pairs; ++i) {
    const int32_t va = h->h_obj_vid[obj_a[i]], vb = h->h_obj_vid[obj_b[i]];
    cols[2 * i] = va >= 0 && vb >= 0 ? od * va : -1; cols[2 * i + 1] = va >= 0 && vb >= 0 ? od * vb : -1;
    first_row[i] = va >= 0 && vb >= 0 ? std::max(h->h_obj_row[va], h->h_obj_row[vb]) / kTile * kTile : 0;   // both columns are zero above
  }
  h->d_cov_Y.resize((size_t)(nrhs * ldt));
  OBVI_HIP(hipMemsetAsync(h->d_cov_Y.get(), 0, sizeof(double) * (size_t)(nrhs * ldt), s));
  h->d_cov_slab.upload(slab_first, s); h->d_cov_cols.upload(cols, s); h->d_cov_first.upload(first_row, s);
  h->d_cov_out.resize((size_t)(od2 * n_pairs));
  const CholPlan plan = chol_plan(h);
  launch_forward_multi(s, plan, h->d_S.get(), h->d_Linv.get(), h->d_cov_Y.get(), ldt, nslabs, h->d_cov_slab.get(), h->d_obj_row.get(), (int32_t)h->nOv, h->h_row_split.data(), od);
  launch_cov_pairs(s, h->d_cov_Y.get(), ldt, n_pairs, h->d_cov_cols.get(), h->d_cov_first.get(), h->d_cov_out.get(), od);
  OBVI_HIP(hipGetLastError());
  h->d_cov_out.download(cov49, (size_t)(od2 * n_pairs), s);
  sync(h);
  for (int64_t i = 0; i < od2 * n_pairs; ++i) if (!std::isfinite(cov49[i])) return fail(h, OBVI_ERR_NUMERICAL, "object_covariances: non-finite covariance");
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_set_parameter_priors(obvi_ba_handle* h, int64_t n, const uint8_t* kind, const uint32_t* block, const uint8_t* param, const double* mean, const double* std_dev) {
  if (!h || n < 0 || (n > 0 && (!kind || !block || !param || !mean || !std_dev))) return fail(h, OBVI_ERR_INVALID_ARGUMENT, "set_parameter_priors: bad arguments");
  OBVI_API_BEGIN
  for (int64_t i = 0; i < n; ++i) {
    const int64_t cnt = kind[i] == 0 ? h->P : kind[i] == 1 ? h->L : kind[i] == 2 ? h->O : -1;
    const int dim = kind[i] == 0 ? 6 : kind[i] == 1 ? 3 : h->od;
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
  if (point3) for (int64_t l = 0; l < h->L; ++l) { const int64_t li = pt_internal(h, l); for (int k = 0; k < 3; ++k) point3[3 * l + k] = point_var[li] ? colsq(sl[3 * li + k]) : -1.0; }
  if (object7) for (int64_t o = 0; o < h->O; ++o) for (int k = 0; k < h->od; ++k) object7[h->od * o + k] = obj_vid[o] >= 0 ? colsq(sc[6 * h->nPv + h->od * (int64_t)obj_vid[o] + k]) : -1.0;
  for (size_t i = 0; i < h->h_pp_kind.size(); ++i) {
    const double w = 1.0 / (h->h_pp_std[i] * h->h_pp_std[i]);
    const int64_t b = h->h_pp_block[i];
    if (h->h_pp_kind[i] == 0 && pose6 && pose_vid[b] >= 0) pose6[6 * b + h->h_pp_param[i]] += w;
    else if (h->h_pp_kind[i] == 1 && point3 && point_var[pt_internal(h, b)]) point3[3 * b + h->h_pp_param[i]] += w;
    else if (h->h_pp_kind[i] == 2 && object7 && obj_vid[b] >= 0) object7[h->od * b + h->h_pp_param[i]] += w;
  }
  return OBVI_OK;
  OBVI_API_END(h)
}

}  // extern "C"

namespace {
// the selection over `n` block norms on the device (select_kernels.hip), the mask into the caller's memory: ONE wait for the device
void run_selection(obvi_ba_handle* h, int64_t n, const double* sq, const uint8_t* act, const uint32_t* inv, double fraction, uint8_t* mask_out, int64_t* num_excluded) {
  h->d_sel_mask.resize((size_t)n + 1);
  int n_out = 0;
  // one round is four launches and one wait; a further round only when more than 4096 distinct values share the open bin's bits (24 more bits each: at most three rounds)
  for (int round = 0; round <= 2; ++round) {
    const int* result_dev = nullptr;
    OBVI_HIP(select_by_threshold(h->stream, n, sq, act, inv, fraction, h->d_sel_mask.get(), &h->sel_scratch, &result_dev, round));
    void* pinned_mask = n ? h->staging.take((size_t)n) : nullptr;
    int* pinned_result = static_cast<int*>(h->staging.take(2 * sizeof(int)));
    int pageable_result[2] = {0, 0};
    if (n) OBVI_HIP(hipMemcpyAsync(pinned_mask ? pinned_mask : mask_out, h->d_sel_mask.get(), (size_t)n, hipMemcpyDeviceToHost, h->stream));
    OBVI_HIP(hipMemcpyAsync(pinned_result ? pinned_result : pageable_result, result_dev, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    sync(h);
    const int* result = pinned_result ? pinned_result : pageable_result;
    n_out = result[0];
    if (result[1] == 0) {
      if (pinned_mask) std::memcpy(mask_out, pinned_mask, (size_t)n);
      break;
    }
    if (result[1] != round + 1 || round == 2) throw HipError{hipErrorUnknown, "outlier selection: the radix select did not finish", __FILE__, __LINE__};
  }
  if (num_excluded) *num_excluded = n_out;
}
}  // namespace

extern "C" {

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
int obvi_ba_get_points(obvi_ba_handle* h, double* out) {
  if (!h || (h->L > 0 && !out)) return OBVI_ERR_INVALID_ARGUMENT;
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  const double* src = points_in_caller_order(h, h->d_point);
  if (h->L > 0) OBVI_HIP(hipMemcpyAsync(out, src, sizeof(double) * 3 * (size_t)h->L, hipMemcpyDeviceToHost, h->stream));
  sync(h);
  return OBVI_OK;
  OBVI_API_END(h)
}
int obvi_ba_get_objects(obvi_ba_handle* h, double* out) { return h ? get_blocks(h, h->d_obj, h->O, h->od, out) : OBVI_ERR_INVALID_ARGUMENT; }

int obvi_ba_get_state(obvi_ba_handle* h, double* poses, double* points, double* objects) {
  if (!h) return OBVI_ERR_INVALID_ARGUMENT;
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  // the three copies land in the handle's pinned arena (a copy into pageable memory blocks, one after the other), ONE wait, then out
  struct Part { double* out; const DevBuf<double>* d; size_t n; void* pinned; } parts[3] = {
      {poses, &h->d_pose, (size_t)h->P * 6, nullptr}, {points, &h->d_point, (size_t)h->L * 3, nullptr}, {objects, &h->d_obj, (size_t)h->O * (size_t)h->od, nullptr}};
  for (Part& p : parts) {
    if (!p.out || !p.n) continue;
    p.pinned = h->staging.take(p.n * sizeof(double));
    double* dst = p.pinned ? static_cast<double*>(p.pinned) : p.out;
    if (p.d == &h->d_point) OBVI_HIP(hipMemcpyAsync(dst, points_in_caller_order(h, h->d_point), p.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));   // (the caller's feature numbering)
    else p.d->download(dst, p.n, h->stream);
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
  const double v[18] = {(double)h->nPv, (double)h->nOv, (double)h->nLv, (double)h->m_canon, (double)h->nt, (double)h->nblk, (double)(h->npairs + h->npairs_window),
                        (double)h->ntiles, (double)h->n_trsm_jobs, (double)h->n_upd_products, h->chol_flops, (double)act_rp, (double)act_bb,
                        (double)h->nlevels, (double)host_threads(), (double)usable_cpus(),
                        (double)h->potrf_wait_timeouts, h->fused_potrf ? 1.0 : 0.0};   // [16] LM steps re-run because a potrf workgroup of the fused level kernel timed out waiting for its jobs, [17] the fused schedule is still on
  const int n = std::min<int>(cap, 18);
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
