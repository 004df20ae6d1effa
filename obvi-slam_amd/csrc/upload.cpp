// upload.cpp -- obvi_ba_set_*: parameter blocks, factor families, masks -- host mirrors, sorting into both observation orders, device copies  (include/obvi_ba.h; shared state and helpers: ba_handle.h)
#include "ba_handle.h"

extern "C" {

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
// the caller's constness flags of the features (just assigned to h_point_const in the caller's order) into the internal order
static void point_flags_to_internal(obvi_ba_handle* h) {
  if (!pt_mapped(h)) return;
  std::vector<uint8_t> in((size_t)h->L);
  for (int64_t l = 0; l < h->L; ++l) in[h->h_pt_new_of_old[(size_t)l]] = h->h_point_const[(size_t)l];
  h->h_point_const.swap(in);
}
int obvi_ba_set_points(obvi_ba_handle* h, int64_t n, const double* v, const uint8_t* c) {
  // (the numbering belongs to the observations: it stays until obvi_ba_set_reproj makes another.  A feature count it does not fit leaves the values in the caller's
  //  order -- the handle is then in the error state validate_indices reports -- and the same count again, or new observations, repair it)
  const int rc = set_blocks(h, n, 3, v, c, h ? &h->L : nullptr, h ? &h->h_point_const : nullptr, h ? &h->d_point : nullptr);
  if (rc == OBVI_OK) h->pt_map_applied = !h->h_pt_new_of_old.empty() && (int64_t)h->h_pt_new_of_old.size() == n;
  if (rc == OBVI_OK && h->pt_map_applied) {
    OBVI_API_BEGIN
    points_to_internal(h, h->d_point); point_flags_to_internal(h);
    sync(h);
    OBVI_API_END(h)
  }
  return rc;
}
int obvi_ba_set_objects(obvi_ba_handle* h, int64_t n, const double* v, const uint8_t* c) {
  const int rc = set_blocks(h, n, h ? h->od : 7, v, c, h ? &h->O : nullptr, h ? &h->h_object_const : nullptr, h ? &h->d_obj : nullptr);
  if (rc == OBVI_OK) {   // where the objects are, for the order of the shared tail (every rank uploads the shared objects with the same values: include/obvi_ba.h)
    h->h_obj_xy.resize((size_t)2 * (size_t)n);
    for (int64_t o = 0; o < n; ++o) { h->h_obj_xy[2 * o] = v[h->od * o]; h->h_obj_xy[2 * o + 1] = v[h->od * o + 1]; }
  }
  return rc;
}

int obvi_ba_set_const_flags(obvi_ba_handle* h, const uint8_t* pc, const uint8_t* lc, const uint8_t* oc) {
  if (!h) return OBVI_ERR_INVALID_ARGUMENT;
  OBVI_API_BEGIN
  if (pc) h->h_pose_const.assign(pc, pc + h->P);
  if (lc) { h->h_point_const.assign(lc, lc + h->L); point_flags_to_internal(h); }
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
  points_to_internal(h, h->d_point);
  finish_upload(h);
  return OBVI_OK;
  OBVI_API_END(h)
}

int obvi_ba_update_state(obvi_ba_handle* h, const double* poses, const double* points, const double* objects) {
  if (!h) return OBVI_ERR_INVALID_ARGUMENT;
  OBVI_API_BEGIN
  OBVI_HIP(hipSetDevice(h->device));
  if (poses && h->P > 0) h2d_async(h->d_pose.get(), poses, sizeof(double) * 6 * h->P, h->stream);
  if (points && h->L > 0) { h2d_async(h->d_point.get(), points, sizeof(double) * 3 * h->L, h->stream); points_to_internal(h, h->d_point); }
  if (objects && h->O > 0) {
    h2d_async(h->d_obj.get(), objects, sizeof(double) * h->od * h->O, h->stream);
    // The order of the shared tail follows the shared objects' (x, y) AS UPLOADED (plan.cpp), and every rank derives it from its own copy: values that arrive
    // here -- a handle planned ahead with placeholders, obvi_ba_prepare + obvi_ba_update_state -- are "as uploaded" too.  If a shared object moved, the key is
    // refreshed and the plan rebuilt at the next prepare / solve, so that this rank orders the tail like a rank that got the same values through
    // obvi_ba_set_objects (such a handle used to keep the placeholder order and was then refused by the order check of the solve).
    if (!h->h_is_shared.empty() && (int64_t)h->h_obj_xy.size() == 2 * h->O) {
      bool moved = false;
      for (int64_t o = 0; o < h->O; ++o) {
        if (!h->h_is_shared[o]) continue;
        if (h->h_obj_xy[2 * o] != objects[h->od * o] || h->h_obj_xy[2 * o + 1] != objects[h->od * o + 1]) moved = true;
      }
      if (moved) h->dirty = true;
    }
    if ((int64_t)h->h_obj_xy.size() == 2 * h->O) for (int64_t o = 0; o < h->O; ++o) { h->h_obj_xy[2 * o] = objects[h->od * o]; h->h_obj_xy[2 * o + 1] = objects[h->od * o + 1]; }
  }
  finish_upload(h);
  h->have_snapshot = false;
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
  // Internal feature numbering (ba_handle.h): by first observing pose, for problems big enough that the gathers of the per-observation kernels are what they wait for
  // (a sliding window is bound by its launches, and its host time matters more: it keeps the caller's numbering).  OBVI_POINT_RENUMBER_MIN: observations from which on
  // (default 2^18; 0: never).
  {
    const char* env = std::getenv("OBVI_POINT_RENUMBER_MIN");
    const int64_t min_obs = env ? std::atoll(env) : ((int64_t)1 << 18);
    std::vector<uint32_t> old_of_new;
    bool identity = true;
    if (min_obs > 0 && n >= min_obs && h->L > 1) {
      std::vector<uint32_t> first((size_t)h->L, (uint32_t)h->P);
      for (int64_t i = 0; i < n; ++i) first[point_idx[i]] = std::min(first[point_idx[i]], pose_idx[i]);
      std::vector<uint32_t> at((size_t)h->P + 2, 0);   // counting sort by the first pose (stable: ties and unobserved features keep the caller's order)
      for (int64_t l = 0; l < h->L; ++l) at[first[l] + 1]++;
      for (int64_t p = 0; p <= h->P; ++p) at[p + 1] += at[p];
      old_of_new.resize((size_t)h->L);
      for (int64_t l = 0; l < h->L; ++l) old_of_new[at[first[l]]++] = (uint32_t)l;
      for (int64_t l = 0; l < h->L && identity; ++l) identity = old_of_new[l] == (uint32_t)l;
    }
    if (identity) old_of_new.clear();
    const bool from_internal = pt_mapped(h);   // else: the values are in the caller's order (no numbering yet, or one that was not applied to the last obvi_ba_set_points)
    if (from_internal ? old_of_new != h->h_pt_old_of_new : !old_of_new.empty()) {
      // the features' values and flags are in the order of the numbering in force: move them to the new one (row `new` <- row of the same feature in the old order)
      std::vector<uint32_t> map((size_t)h->L);
      for (int64_t nw = 0; nw < h->L; ++nw) { const uint32_t old = old_of_new.empty() ? (uint32_t)nw : old_of_new[nw]; map[nw] = from_internal ? h->h_pt_new_of_old[old] : old; }
      h->d_pt_map_tmp.upload(map, h->stream);
      h->d_pt_tmp.resize((size_t)3 * h->L + 1);
      launch_permute_rows3(h->stream, h->d_pt_tmp.get(), h->d_point.get(), h->d_pt_map_tmp.get(), h->L);
      h->d_point.swap(h->d_pt_tmp);
      std::vector<uint8_t> flags((size_t)h->L);
      for (int64_t nw = 0; nw < h->L; ++nw) flags[nw] = h->h_point_const[map[nw]];
      h->h_point_const.swap(flags);
      h->h_pt_old_of_new = old_of_new;
      h->h_pt_new_of_old.assign(old_of_new.size(), 0);
      for (size_t nw = 0; nw < old_of_new.size(); ++nw) h->h_pt_new_of_old[old_of_new[nw]] = (uint32_t)nw;
      if (!old_of_new.empty()) { h->d_pt_old_of_new.upload(h->h_pt_old_of_new, h->stream); h->d_pt_new_of_old.upload(h->h_pt_new_of_old, h->stream); }
      sync(h);                 // (`map` is a local)
      h->have_snapshot = false;   // a snapshot taken under the old numbering cannot be restored into the new one
    } else if (!from_internal) {
      h->h_pt_old_of_new.clear(); h->h_pt_new_of_old.clear();   // the caller's order stays (a numbering that was never applied is dropped)
    }
    h->pt_map_applied = !h->h_pt_new_of_old.empty();
  }
  std::vector<uint32_t>& pidx = h->scr_point_internal;   // the observations' feature indices in the internal numbering
  pidx.resize(n);
  if (h->h_pt_new_of_old.empty()) { for (int64_t i = 0; i < n; ++i) pidx[i] = point_idx[i]; }
  else { for (int64_t i = 0; i < n; ++i) pidx[i] = h->h_pt_new_of_old[point_idx[i]]; }
  std::vector<uint32_t>& perm = h->h_rp_perm;
  std::vector<uint32_t>& ptr = h->h_point_ptr;
  std::vector<uint32_t>& cur = h->scr_cursor;
  perm.resize(n); ptr.assign(h->L + 1, 0);
  for (int64_t i = 0; i < n; ++i) ptr[pidx[i] + 1]++;
  for (int64_t l = 0; l < h->L; ++l) ptr[l + 1] += ptr[l];
  cur.assign(ptr.begin(), ptr.end() - 1);
  for (int64_t i = 0; i < n; ++i) perm[cur[pidx[i]]++] = (uint32_t)i;
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
      h->h_rp_pose[a] = pose_idx[i]; h->h_rp_point[a] = pidx[i];
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
      // (the piece's image in LDS: 18 doubles per observation + 4 per point index it spans, empty ones included; ba_device.h)
      if (count > 0 && (k > 64 || count + k > 64 || l - lfirst >= 64 || 18 * (int64_t)(count + k) + 4 * (l - lfirst + 1) > kPointImageDoubles)) { wave_obs.push_back(start); wave_obs.push_back(count); count = 0; }
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
  const int od = h->od, od2 = od * od;
  std::vector<double> si((size_t)od2 * n);
  for (int64_t i = 0; i < n; ++i) {
    if (obj_idx[i] >= h->O) return fail(h, OBVI_ERR_OUT_OF_RANGE, "set_ltm_priors: index out of range");
    if (!sym_inverse_sqrt(cov49 + od2 * i, od, &si[od2 * i])) return fail(h, OBVI_ERR_NUMERICAL, "set_ltm_priors: covariance not SPD");
  }
  h->n_lt = n; h->lt_huber = huber; h->max_lt_obj = max_index(obj_idx, n);
  h->h_lt_obj.assign(obj_idx, obj_idx + n); h->h_lt_active.assign(n, 1);
  hipStream_t s = h->stream;
  h->d_lt_obj.upload(h->h_lt_obj, s); h->d_lt_mean.upload(mean7, od * n, s); h->d_lt_sqrt_inf.upload(si, s); h->d_lt_active.upload(h->h_lt_active, s);
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
int64_t obvi_ba_num_residuals(const obvi_ba_handle* h) { return h ? 2 * h->n_rp + 4 * h->n_bb + 3 * h->n_sp + h->od * h->n_lt + 6 * h->n_rl : -1; }

}  // extern "C"
