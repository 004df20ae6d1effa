// plan.cpp -- the symbolic phase of a problem: reduced program, elimination order, Schur work lists, tile plan with fill and level jobs; mask-only re-plan  (include/obvi_ba.h; shared state and helpers: ba_handle.h)
#include "ba_handle.h"

namespace obvi_lib {

void prepare(obvi_ba_handle* h) {
  prepare_plan(h);
  ensure_det_slots(h);
}

// The symbolic phase, stage by stage.  One object per call of prepare_plan(): the members are what one stage leaves for the next (everything else is local
// to its stage); the stages run in the order of run().  OBVI_DEBUG_PREPARE: stage times on stderr.
struct PlanBuilder {
  explicit PlanBuilder(obvi_ba_handle* handle) : h(handle), P(handle->P), L(handle->L), O(handle->O) {}
  void run() {
    reduced_program();         stage("reduced program");
    elimination_order();       stage("ordering");
    schur_pairs_and_visits();  stage("schur pairs / visits");
    schur_batches();
    pair_blocks();             stage("schur batches");
    tile_mask_and_fill();      stage("tile mask + fill");
    level_jobs();              stage("level jobs");
    substitution_lists();      stage("lists");
    upload_and_allocate();     stage("upload + allocations");
    remember_what_the_plan_was_built_for();
  }

  obvi_ba_handle* const h;
  const int64_t P, L, O;
  // reduced program
  int64_t nres = 0, nPv = 0;
  std::vector<int32_t> pose_vid, obj_vid, nat;   // variable id of a pose / an object (-1: constant or unused); nat: rank among the variable poses in frame order
  std::vector<uint8_t> point_var;
  // elimination order
  int32_t nt = 0;
  int64_t m_pad = 0;
  // Schur complement work lists
  static constexpr int32_t SR = kSchurRows, SBACK = kSchurWindowFrames - kSchurRows;
  struct Pair { uint64_t key; uint32_t a, b; };
  struct Visit { int32_t chunk; uint32_t l, beg, k; bool twin; uint64_t tiles; };
  std::vector<uint8_t> mask;                   // nt x nt tiles of the reduced matrix, lower triangle
  std::vector<Pair> pairs;                     // observation pairs outside the strips (k_schur_blocks)
  std::vector<std::vector<Visit>> visits_t;    // the visits, in point order: one list per range of points (they are never merged: the counting sort reads the ranges)
  int64_t n_window_pairs = 0, max_visits = 0;
  bool any_twin = false, pair_bitmap = false, slots_on_host = false;
  uint32_t zero16 = 0;
  std::vector<uint32_t> wg_bptr, bfirst, bslot, visits, slot_src, plan_wg_ptr, plan_wg_slot0, blk_row, blk_col, blk_ptr, pair_a, pair_b;
  std::vector<int32_t> wg_f0, wg_group;
  std::vector<PlanVisit> plan_visits;
  size_t total_slots = 0;
  // tile plan
  std::vector<int32_t> col_ptr, col_i, level;
  int32_t nlev = 0;
  std::vector<std::vector<int32_t>> by_level, pre_of;
  std::vector<int32_t> lvl_k, trsm_ik, upd_ij, upd_kptr, upd_k, rh_i, rh_kptr, rh_k, job_signal, k_need_of, bw_kj, bw_chains, tiles;
  std::vector<uint8_t> upd_flag;
  double flops = 0.0;
  int64_t n_products = 0;

  const bool stage_times = std::getenv("OBVI_DEBUG_PREPARE") != nullptr;
  std::chrono::steady_clock::time_point t_prev = std::chrono::steady_clock::now();
  void stage(const char* name) {
    if (!stage_times) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "prepare: %-28s %8.2f ms\n", name, std::chrono::duration<double, std::milli>(t - t_prev).count());
    t_prev = t;
  }
  static int env_int(const char* name, int dflt) { const char* v = std::getenv(name); return v ? std::atoi(v) : dflt; }
  void mark(int64_t row, int dr, int64_t col, int dc) {   // the tiles a (dr x dc) block at (row, col) touches
    const int t0 = (int)(row / kTile), t1 = (int)((row + dr - 1) / kTile), c0 = (int)(col / kTile), c1 = (int)((col + dc - 1) / kTile);
    for (int ti = t0; ti <= t1; ++ti) for (int tj = c0; tj <= c1; ++tj) if (ti >= tj) mask[(size_t)ti * nt + tj] = 1;
  }

  // ---- which blocks are variables of this solve, how many residuals: the reduced program as Ceres would report it
  void reduced_program() {
    std::vector<uint8_t> pose_used(P, 0), obj_used(O, 0), point_used(L, 0);
    nres = 0;
    {   // ranges of observations on the host threads: the flags are idempotent byte stores of 1 (relaxed atomics: ranges share poses, and a point at a range's edge)
      const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), h->n_rp / 65536));
      std::vector<int64_t> nres_t(parts, 0);
      std::vector<std::vector<uint8_t>> pose_used_t(parts);   // per range: every range sees every pose (sixteen threads storing into the same P bytes were slower than one)
      parallel_ranges(h->n_rp, parts, [&](int part, int64_t a0, int64_t a1) {
        int64_t n = 0;
        std::vector<uint8_t>& pu = pose_used_t[part];
        pu.assign((size_t)P, 0);
        for (int64_t a = a0; a < a1; ++a) {
          if (!h->h_rp_active[a]) continue;
          const uint32_t p = h->h_rp_pose[a], l = h->h_rp_point[a];
          const bool cp = h->h_pose_const[p], cl = h->h_point_const[l];
          if (cp && cl) continue;
          n += 2;
          if (!cp) pu[p] = 1;
          if (!cl) __atomic_store_n(&point_used[l], (uint8_t)1, __ATOMIC_RELAXED);   // the observations are in point order: ranges meet in one point at most
        }
        nres_t[part] = n;
      });
      for (int64_t n : nres_t) nres += n;
      for (const auto& pu : pose_used_t) for (int64_t p = 0; p < P && !pu.empty(); ++p) pose_used[p] |= pu[p];
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
    for (int64_t i = 0; i < h->n_lt; ++i) if (h->h_lt_active[i] && !h->h_object_const[h->h_lt_obj[i]]) { nres += h->od; obj_used[h->h_lt_obj[i]] = 1; }
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
    pose_vid.assign((size_t)P, -1); obj_vid.assign((size_t)O, -1);
    point_var.assign((size_t)L, 0);
    h->nPv = h->nOv = h->nLv = 0;
    nat.assign((size_t)P, -1);           // rank among the variable poses in pose-index (frame) order
    for (int64_t p = 0; p < P; ++p) if (!h->h_pose_const[p] && pose_used[p]) nat[p] = (int32_t)h->nPv++;
    for (int64_t o = 0; o < O; ++o) if (!h->h_object_const[o] && obj_used[o]) obj_vid[o] = (int32_t)h->nOv++;   // provisional: index order
    for (int64_t l = 0; l < L; ++l) if (!h->h_point_const[l] && point_used[l]) { point_var[l] = 1; h->nLv++; }
    nPv = h->nPv;

  }

  void elimination_order() {
    // ---- elimination order: nested dissection of the frame chain, objects inside the tree.  reach[f] = largest
    //      frame rank f couples to through a shared point or an odometry factor; a separator
    //      [s0, s1) with s1 > reach of everything left of s0 decouples the two sides.  Cut positions
    //      are multiples of 32 poses (= 3 tiles) so tree nodes never share a tile.
    {
      std::vector<int32_t> reach(nPv);
      for (int64_t f = 0; f < nPv; ++f) reach[f] = (int32_t)f;
      {   // ranges of points on the host threads, every range with a reach array of its own (nPv integers), joined by maximum
        const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), L / 4096));
        std::vector<std::vector<int32_t>> reach_t(parts);
        parallel_ranges(L, parts, [&](int part, int64_t l0, int64_t l1) {
          std::vector<int32_t>& r = reach_t[part];
          r.assign((size_t)nPv, -1);
          for (int64_t l = l0; l < l1; ++l) {
            if (!point_var[l]) continue;
            int32_t hi = -1;
            for (uint32_t a = h->h_point_ptr[l]; a < h->h_point_ptr[l + 1]; ++a) {
              if (!h->h_rp_active[a]) continue;
              const int32_t f = nat[h->h_rp_pose[a]];
              if (f >= 0) hi = std::max(hi, f);
            }
            if (hi < 0) continue;
            for (uint32_t a = h->h_point_ptr[l]; a < h->h_point_ptr[l + 1]; ++a) {   // every frame of the track couples to its last one
              if (!h->h_rp_active[a]) continue;
              const int32_t f = nat[h->h_rp_pose[a]];
              if (f >= 0) r[f] = std::max(r[f], hi);
            }
          }
        });
        for (const auto& r : reach_t) for (int64_t f = 0; f < nPv && !r.empty(); ++f) reach[f] = std::max(reach[f], r[f]);
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
      std::vector<int64_t> tail_objs;                                   // shared across ranks: eliminated last, in an order every rank derives alike (below)
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
        for (int64_t o : objs) { obj_vid[o] = next_obj; h->h_obj_row[next_obj++] = (int32_t)row; row += h->od; }
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
      if (tail_objs.size() > 1 && (int64_t)h->h_obj_xy.size() == 2 * O && (!std::getenv("OBVI_TAIL_SPATIAL") || std::atoi(std::getenv("OBVI_TAIL_SPATIAL")) != 0)) {
        // Order of the shared tail (round 5).  Every rank must lay the shared objects out in the SAME order (the tail's tiles are summed across ranks), so the
        // order can only depend on what all ranks share: the objects' index and their uploaded values.  Object-index order (rounds 2-4) is arbitrary with
        // respect to the trajectory, so every pose tile column coupled with every object tile row of the tail (9 objects to a row: each row holds one that
        // some frame of the column sees): config #5, 16 sessions fused: 322 k tile products per factorisation.  A Hilbert curve over the objects' (x, y) as
        // uploaded puts objects that are seen together next to each other: a pose column then meets the few tail rows of its surroundings (116 k products
        // with the objects in first-observing-frame order of a single-rank problem).  Ties: object index.
        double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
        for (int64_t o : tail_objs) { x0 = std::min(x0, h->h_obj_xy[2 * o]); x1 = std::max(x1, h->h_obj_xy[2 * o]); y0 = std::min(y0, h->h_obj_xy[2 * o + 1]); y1 = std::max(y1, h->h_obj_xy[2 * o + 1]); }
        const double span = std::max(std::max(x1 - x0, y1 - y0), 1e-12);
        auto hilbert = [](uint32_t x, uint32_t y) {   // index of (x, y) on the 2^16 x 2^16 Hilbert curve
          uint64_t d = 0;
          for (uint32_t s = 1u << 15; s > 0; s >>= 1) {
            const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
            d += (uint64_t)s * (uint64_t)s * ((3u * rx) ^ ry);
            if (ry == 0) { if (rx == 1) { x = 65535u - x; y = 65535u - y; } std::swap(x, y); }
          }
          return d;
        };
        std::vector<std::pair<uint64_t, int64_t>> keyed;
        keyed.reserve(tail_objs.size());
        for (int64_t o : tail_objs) {
          const double fx = (h->h_obj_xy[2 * o] - x0) / span, fy = (h->h_obj_xy[2 * o + 1] - y0) / span;
          const bool finite = std::isfinite(fx) && std::isfinite(fy);
          const uint32_t qx = finite ? (uint32_t)std::min(65535.0, std::max(0.0, fx * 65535.0)) : 0u, qy = finite ? (uint32_t)std::min(65535.0, std::max(0.0, fy * 65535.0)) : 0u;
          keyed.emplace_back(hilbert(qx, qy), o);
        }
        std::sort(keyed.begin(), keyed.end());
        for (size_t i = 0; i < keyed.size(); ++i) tail_objs[i] = keyed[i].second;
      }
      if (!tail_objs.empty()) {
        row = ((row + kTile - 1) / kTile) * kTile;
        h->tail_t0 = (int32_t)(row / kTile);
        place_node(0, 0, tail_objs);
        for (int64_t o : tail_objs) h->h_shared_ov.push_back(obj_vid[o]);
        uint64_t hsh = 1469598103934665603ull;   // the order as this rank derived it: compared across ranks at the start of every solve (lm.cpp)
        for (int64_t o : tail_objs) { hsh ^= (uint64_t)o; hsh *= 1099511628211ull; }
        h->tail_order_hash = (double)(hsh >> 24);
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
    h->m_canon = 6 * nPv + h->od * h->nOv;
    h->h_canon_row.resize(h->m_canon);
    for (int64_t p = 0; p < P; ++p) if (nat[p] >= 0) for (int k = 0; k < 6; ++k) h->h_canon_row[6 * (int64_t)nat[p] + k] = (int64_t)h->h_pose_row[pose_vid[p]] + k;
    {
      int64_t rank = 0;   // canonical order = object index order
      for (int64_t o = 0; o < O; ++o) if (obj_vid[o] >= 0) { for (int k = 0; k < h->od; ++k) h->h_canon_row[6 * nPv + h->od * rank + k] = (int64_t)h->h_obj_row[obj_vid[o]] + k; ++rank; }
    }
    h->num_params = h->m_canon + 3 * h->nLv;
    h->num_residuals = nres;
    nt = h->nt;
    m_pad = (int64_t)nt * kTile;

  }

  void schur_pairs_and_visits() {
    // ---- Schur complement work lists.  k_schur_window takes every ordered observation pair (i >= j) of a point whose
    //      frame distance is below the window's offset count; a point is visited once per row chunk that holds one of
    //      its observations.  The remaining pairs (a, b) with row(a) >= row(b) go to k_schur_blocks grouped by 6x6
    //      block.  The tile mask gets every block.
    mask.assign((size_t)nt * (size_t)nt, 0);
    pairs.clear(); visits_t.clear();
    n_window_pairs = 0; any_twin = false;
    // pose pairs that share a point: collected in a bitmap (one store per pair of sightings) and turned into tile marks once per
    // pose pair afterwards -- a point contributes k (k + 1) / 2 pairs and most of them repeat
    pair_bitmap = h->nPv <= env_int("OBVI_PAIR_BITMAP_MAX", 8192);   // 64 MB at most; beyond it the tile marks are made pair by pair (tuning knob)
    std::vector<uint8_t> pose_pair(pair_bitmap ? (size_t)h->nPv * (size_t)h->nPv : 0, 0);
    {
      // points are independent: ranges of points on host threads (the bitmap is shared: every writer stores the same 1), lists joined in
      // point order.  Without the bitmap the tile marks go straight into the mask: one thread.
      const int64_t grain = std::max(1, env_int("OBVI_PLAN_GRAIN", 256));   // points per range (tuning knob)
      const int parts = pair_bitmap ? (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), L / grain)) : 1;   // the workers exist (host_pool): a range of a few hundred points is worth handing out
      std::vector<std::vector<Pair>> pairs_t(parts);
      visits_t.assign(parts, {});
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
        n_window_pairs += window_pairs_t[t]; any_twin = any_twin || twin_t[t];
        std::vector<Pair>().swap(pairs_t[t]);
      }
    }
    if (pair_bitmap)   // rows of the bitmap on the host threads: the marks are idempotent byte stores of 1 (relaxed atomics: two pose pairs may share a tile)
      parallel_ranges(h->nPv, (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), h->nPv / 64)), [&](int, int64_t r0, int64_t r1) {
        for (int64_t hi = r0; hi < r1; ++hi) {
          const uint8_t* row = &pose_pair[(size_t)hi * (size_t)h->nPv];
          const int64_t row_hi = h->h_pose_row[hi];
          for (int64_t lo = 0; lo <= hi; ++lo) {
            if (!row[lo]) continue;
            const int64_t row_lo = h->h_pose_row[lo];
            const int t0 = (int)(row_hi / kTile), t1 = (int)((row_hi + 5) / kTile), c0 = (int)(row_lo / kTile), c1 = (int)((row_lo + 5) / kTile);
            for (int ti = t0; ti <= t1; ++ti) for (int tj = c0; tj <= c1; ++tj) if (ti >= tj) __atomic_store_n(&mask[(size_t)ti * nt + tj], (uint8_t)1, __ATOMIC_RELAXED);
          }
        }
      });
  }

  void schur_batches() {
    // one work list per (chunk, column group): the visits with a tile in that group
    constexpr int kGroups = (kSchurWindowFrames * 6 / 16) / kSchurGroupCols, kGroupBits = 3 * kSchurGroupCols;
    struct GVisit { int32_t chunk, group; uint32_t l, beg, k; bool twin; uint32_t bits; };
    std::vector<GVisit> gv;   // the visits per (chunk, column group) work list
    {
      // ordered by chunk, then by group descending, visits of a list in point order: a counting sort over the (chunk, group) buckets -- counted and
      // scattered range by range on the host threads (the ranges are in point order: bucket by bucket, range after range, it is the serial sort)
      const int nparts = (int)visits_t.size();
      const size_t nbuckets = ((size_t)(h->nPv / SR) + 2) * kGroups;
      auto bucket = [&](int32_t chunk, int g) { return (size_t)chunk * kGroups + (size_t)(kGroups - 1 - g); };
      auto group_bits = [&](const Visit& v, int g) { return (uint32_t)(v.tiles >> (kGroupBits * g)) & ((1u << kGroupBits) - 1u); };
      std::vector<std::vector<uint32_t>> cursor_t(nparts);
      parallel_ranges(nparts, nparts, [&](int, int64_t p0, int64_t p1) {
        for (int64_t p = p0; p < p1; ++p) {
          std::vector<uint32_t>& c = cursor_t[p];
          c.assign(nbuckets, 0);
          for (const Visit& v : visits_t[p])
            for (int g = 0; g < kGroups; ++g) if (group_bits(v, g)) ++c[bucket(v.chunk, g)];
        }
      });
      size_t total = 0;
      for (size_t b2 = 0; b2 < nbuckets; ++b2)
        for (int p = 0; p < nparts; ++p) { const uint32_t n = cursor_t[p][b2]; cursor_t[p][b2] = (uint32_t)total; total += n; }
      gv.resize(total);
      parallel_ranges(nparts, nparts, [&](int, int64_t p0, int64_t p1) {
        for (int64_t p = p0; p < p1; ++p) {
          std::vector<uint32_t>& c = cursor_t[p];
          for (const Visit& v : visits_t[p])
            for (int g = 0; g < kGroups; ++g) {
              const uint32_t bits = group_bits(v, g);
              if (bits) gv[c[bucket(v.chunk, g)]++] = {v.chunk, g, v.l, v.beg, v.k, v.twin, bits};
            }
          std::vector<Visit>().swap(visits_t[p]);
        }
      });
    }
    max_visits = std::max(8, env_int("OBVI_SCHUR_VISITS", 1 << 20));   // visits per workgroup (tuning knob)
    // slices of a work list: enough workgroups to fill the device on small problems, at most max_visits visits each
    // (deterministic mode: a work list is never cut -- one workgroup, hence one writer, per strip)
    const int64_t slice = h->deterministic ? ((int64_t)1 << 40) : std::min<int64_t>(max_visits, std::max<int64_t>(64, (int64_t)gv.size() / env_int("OBVI_SCHUR_WGS", 1536)));
    // per workgroup: batches of visits that fit the kernel's LDS buffer.  A visit is laid out as consecutive 144-byte
    // slots: one per strip frame over the range of its row frames and of its column frames in the group (source: the Z
    // record, or the zero page for a frame the point skips), the point's (u_l, 0) tail, and -- stereo -- a second layer
    // with the second record of each frame.
    zero16 = (uint32_t)((18ull * (uint64_t)h->n_rp + 4ull * (uint64_t)L + 4ull) / 2);   // zero page behind the Z blocks
    wg_bptr.assign(1, 0); bfirst.assign(1, 0); bslot.assign(1, 0); visits.clear(); slot_src.clear();
    wg_f0.clear(); wg_group.clear();

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
    // Who fills the slot tables.  Default (round 5): the DEVICE (plan_kernels.hip: one lane per visit walks the point's observations).  What is left for the
    // host is to deal the visits to batches, which needs a visit's slot COUNT only -- a function of its tile bits, its group and the twin flag.
    // OBVI_PLAN_SLOTS_ON_HOST=1: the host fills them as rounds 1-4 did (the check: both give the same tables).
    slots_on_host = std::getenv("OBVI_PLAN_SLOTS_ON_HOST") && std::atoi(std::getenv("OBVI_PLAN_SLOTS_ON_HOST")) != 0;
    auto visit_slot_count = [&](const GVisit& v) -> uint32_t {
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
      const bool merged = B0 <= A1 + 1 && A0 <= B1 + 1;
      if (merged) { const uint32_t span = (uint32_t)(std::max(A1, B1) - std::min(A0, B0) + 1); return span + 1 + (v.twin ? span : 0u); }
      const uint32_t one = (uint32_t)(A1 - A0 + 1) + 1 + (uint32_t)(B1 - B0 + 1);
      return v.twin ? 2 * one : one;
    };
    struct BatchLists { std::vector<uint32_t> visits, slot_src, end_visit, end_slot, wg_batches, wg_slots; };
    const int parts2 = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), (int64_t)gv.size() / (4 * (int64_t)std::max(1, env_int("OBVI_PLAN_GRAIN", 256)))));
    std::vector<BatchLists> lists_t(parts2);
    plan_visits.assign(slots_on_host ? 0 : gv.size(), PlanVisit{});
    parallel_ranges((int64_t)wgs.size(), parts2, [&](int part, int64_t g0, int64_t g1) {
      BatchLists& o = lists_t[part];
      std::vector<uint32_t> vs;
      uint32_t rec[4];
      uint32_t nvis_part = 0, nslots_part = 0;   // visits / slots of this part so far (device route: the lists themselves are not kept)
      for (int64_t g = g0; g < g1; ++g) {
        uint32_t used = 0, count = 0, nb = 0, wg_slots = 0;
        for (size_t t = wgs[g].w; t < wgs[g].we; ++t) {
          const uint32_t n = slots_on_host ? (visit_slots(gv[t], used, vs, rec), (uint32_t)vs.size()) : visit_slot_count(gv[t]);
          if (count == (uint32_t)kSchurBatchVisits || used + n > kBatchSlots) {
            o.end_visit.push_back(nvis_part); o.end_slot.push_back(nslots_part); ++nb; used = 0; count = 0;
            if (slots_on_host) visit_slots(gv[t], used, vs, rec);
          }
          if (slots_on_host) { o.visits.insert(o.visits.end(), rec, rec + 4); o.slot_src.insert(o.slot_src.end(), vs.begin(), vs.end()); }
          else plan_visits[t] = PlanVisit{gv[t].l, (uint16_t)(gv[t].bits | (gv[t].twin ? 0x8000u : 0u)), (uint16_t)used, wg_slots};
          used += n; ++count; ++nvis_part; nslots_part += n; wg_slots += n;
        }
        o.end_visit.push_back(nvis_part); o.end_slot.push_back(nslots_part); ++nb;
        o.wg_batches.push_back(nb); o.wg_slots.push_back(wg_slots);
      }
    });
    plan_wg_ptr.clear(); plan_wg_slot0.clear();
    total_slots = 0;
    {
      uint32_t voff = 0, soff = 0;
      for (const BatchLists& o : lists_t) {
        if (slots_on_host) { visits.insert(visits.end(), o.visits.begin(), o.visits.end()); slot_src.insert(slot_src.end(), o.slot_src.begin(), o.slot_src.end()); }
        for (size_t b = 0; b < o.end_visit.size(); ++b) { bfirst.push_back(voff + o.end_visit[b]); bslot.push_back(soff + o.end_slot[b]); }
        uint32_t s0 = soff;
        for (size_t w = 0; w < o.wg_batches.size(); ++w) { wg_bptr.push_back(wg_bptr.back() + o.wg_batches[w]); plan_wg_slot0.push_back(s0); s0 += o.wg_slots[w]; }
        if (!o.end_visit.empty()) { voff += o.end_visit.back(); soff += o.end_slot.back(); }
      }
      total_slots = soff;
    }
    for (const WgRange& g : wgs) plan_wg_ptr.push_back((uint32_t)g.w);
    for (const WgRange& g : wgs) { wg_f0.push_back(g.chunk * SR); wg_group.push_back(g.group); }
    h->schur_twins = any_twin ? 1 : 0;
    h->nchunks = (int64_t)wg_f0.size();
    h->npairs_window = n_window_pairs;

    if (stage_times)
      std::fprintf(stderr, "  schur plan: %zu visits, %zu workgroups, %zu batches (%.1f slots each), %zu slots = %.1f MB gathered per launch\n", gv.size(), wgs.size(), bslot.size() - 1,
                   (double)total_slots / std::max<size_t>(1, bslot.size() - 1), total_slots, 144e-6 * (double)total_slots);
  }

  // the pairs outside the strips, grouped by 6x6 block of the reduced matrix (k_schur_blocks)
  void pair_blocks() {
    std::sort(pairs.begin(), pairs.end(), [](const Pair& x, const Pair& y) { return x.key < y.key || (x.key == y.key && (x.a < y.a || (x.a == y.a && x.b < y.b))); });
    blk_row.clear(); blk_col.clear(); blk_ptr.clear(); pair_a.assign(pairs.size(), 0); pair_b.assign(pairs.size(), 0);
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
  }

  void tile_mask_and_fill() {
    // ---- tile mask of the reduced matrix (lower triangle) and symbolic fill ----
    for (int k = 0; k < nt; ++k) mask[(size_t)k * nt + k] = 1;
    for (int64_t i = 0; i < h->n_bb; ++i) {
      if (!h->h_bb_active[i]) continue;
      const int32_t ov = obj_vid[h->h_bb_obj[i]], pv = pose_vid[h->h_bb_pose[i]];
      if (ov >= 0 && pv >= 0) { const int64_t ro = h->h_obj_row[ov], rp = h->h_pose_row[pv]; if (ro > rp) mark(ro, h->od, rp, 6); else mark(rp, 6, ro, h->od); }
    }
    for (int64_t i = 0; i < h->n_rl; ++i) {
      if (!h->h_rl_active[i]) continue;
      const int32_t va = pose_vid[h->h_rl_a[i]], vb = pose_vid[h->h_rl_b[i]];
      if (va >= 0 && vb >= 0 && va != vb) mark(h->h_pose_row[std::max(va, vb)], 6, h->h_pose_row[std::min(va, vb)], 6);
    }
    // object diagonal blocks may straddle tiles
    for (int64_t w = 0; w < h->nOv; ++w) mark(h->h_obj_row[w], h->od, h->h_obj_row[w], h->od);
    for (int64_t v = 0; v < nPv; ++v) mark(h->h_pose_row[v], 6, h->h_pose_row[v], 6);
    // the shared tail is exchanged across ranks as a dense lower-triangular block of tiles
    if (h->tail_t0 >= 0) for (int i = h->tail_t0; i < nt; ++i) for (int j = h->tail_t0; j <= i; ++j) mask[(size_t)i * nt + j] = 1;
    // symbolic fill (tile columns in increasing order) + column structure of L
    col_ptr.assign((size_t)nt + 1, 0); col_i.clear();
    for (int k = 0; k < nt; ++k) {
      const size_t beg = col_i.size();
      for (int i = k + 1; i < nt; ++i) if (mask[(size_t)i * nt + k]) col_i.push_back(i);
      for (size_t x = beg; x < col_i.size(); ++x) for (size_t y = beg; y <= x; ++y) mask[(size_t)col_i[x] * nt + col_i[y]] = 1;
      col_ptr[k + 1] = (int32_t)col_i.size();
    }
  }

  void level_jobs() {
    // levels of the tile elimination tree: k depends on every j < k with L(k,j) != 0
    level.assign((size_t)nt, 0);
    nlev = 0;
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
    by_level.assign((size_t)nlev, {});
    for (int k = 0; k < nt; ++k) by_level[level[k]].push_back(k);
    lvl_k.clear(); trsm_ik.clear(); upd_ij.clear(); upd_kptr.assign(1, 0); upd_k.clear(); rh_i.clear(); rh_kptr.assign(1, 0); rh_k.clear(); job_signal.clear(); k_need_of.assign((size_t)nt, 0);
    upd_flag.clear();
    const int kUpdChunk = h->deterministic ? (1 << 30) : std::max(1, env_int("OBVI_UPD_CHUNK", 2));   // products per update job (tuning knob; deterministic mode: a target's products are never split over jobs that would meet in atomics)
    const int64_t env_slice_max = std::getenv("OBVI_SLICE_MAX") ? std::atoi(std::getenv("OBVI_SLICE_MAX")) : 512;   // tuning knob
    // a potrf workgroup applies up to this many products of the previous level to its own diagonal tile (tuning knob)
    const size_t pre_max = (size_t)std::max(0, env_int("OBVI_PRE_MAX", 2));
    pre_of.assign((size_t)nt, {});
    h->h_lvl_k_ptr.assign(nlev + 1, 0); h->h_trsm_ptr.assign(nlev + 1, 0); h->h_upd_ptr.assign(nlev + 1, 0); h->h_rh_ptr.assign(nlev + 1, 0); h->h_crit_upd.assign(nlev + 1, 0); h->h_crit_rh.assign(nlev + 1, 0); h->h_slices.assign(nlev + 1, 1);
    flops = 0.0;
    const double t3 = (double)kTile * kTile * kTile;
    struct Trip { int32_t i, j, k; };
    std::vector<Trip> trips;
    std::vector<std::pair<int32_t, int32_t>> ik;
    n_products = 0;
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
  }

  void substitution_lists() {
    // backward substitution, row oriented: one workgroup per tile of L.  A launch takes kBwLevels consecutive levels: a row forms the
    // y of its ancestors inside the launch itself (k_backward; chain record per row) and tiles between two rows of a launch get no
    // workgroup.  The launches are listed in the order of the forward levels and run last to first.
    bw_kj.clear(); bw_chains.clear();
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
    tiles.clear();
    for (int i = 0; i < nt; ++i) for (int j = 0; j <= i; ++j) if (mask[(size_t)i * nt + j]) { tiles.push_back(i); tiles.push_back(j); }
    h->ntiles = (int32_t)(tiles.size() / 2);
  }

  void upload_and_allocate() {
    // ---- upload ----
    hipStream_t s = h->stream;
    std::vector<int32_t> frame_of_pose;   // (device-side slot fill: alive until finish_upload())
    h->d_pose_vid.upload(pose_vid, s); h->d_obj_vid.upload(obj_vid, s); h->d_point_var.upload(point_var, s);
    h->h_obj_vid = obj_vid;
    h->d_blk_row.upload(blk_row, s); h->d_blk_col.upload(blk_col, s); h->d_blk_ptr.upload(blk_ptr, s);
    h->d_pair_a.upload(pair_a, s); h->d_pair_b.upload(pair_b, s);
    h->d_chunk_ptr.upload(wg_bptr, s); h->d_batch_first.upload(bfirst, s); h->d_batch_slot.upload(bslot, s); h->d_chunk_f0.upload(wg_f0, s); h->d_chunk_group.upload(wg_group, s);
    if (slots_on_host) {
      h->d_chunk_points.upload(visits, s); h->d_slot_src.upload(slot_src, s);
    } else {
      // the two big tables are written where they are read: a lane per visit (plan_kernels.hip) from 12 bytes per visit instead of 60
      frame_of_pose.assign((size_t)P + 1, -1);
      for (int64_t pz = 0; pz < P; ++pz) if (pose_vid[pz] >= 0) frame_of_pose[pz] = nat[pz];
      h->d_plan_frame.upload(frame_of_pose, s); h->d_plan_visits.upload(plan_visits, s); h->d_plan_wg_ptr.upload(plan_wg_ptr, s); h->d_plan_wg_slot0.upload(plan_wg_slot0, s);
      h->d_chunk_points.resize(4 * plan_visits.size() + 4); h->d_slot_src.resize(total_slots + 4);
      launch_plan_visit_slots(s, (int64_t)plan_visits.size(), h->d_plan_visits.get(), h->d_plan_wg_ptr.get(), h->d_plan_wg_slot0.get(), (int32_t)plan_wg_ptr.size(), h->d_chunk_f0.get(), h->d_chunk_group.get(),
                              h->d_point_ptr.get(), h->d_rp_active.get(), h->d_rp_pose.get(), h->d_plan_frame.get(), zero16, h->d_chunk_points.get(), h->d_slot_src.get());
      // (the kernel's inputs went through the pinned arena, or -- too big for it -- straight from vectors of this function: finish_upload() at its end waits then)
    }
    h->d_row_of_nat.upload(h->h_row_of_nat, s);
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
      h->d_bb_blk.resize((size_t)(h->od * (h->od + 1) / 2 + h->od + 27) * (size_t)h->n_bb + 1);   // BbSlot<OD>::kBlk: 62, 81
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
      h->d_xbuf.resize((size_t)std::max<int64_t>((int64_t)(h->od * h->od + h->od) * (int64_t)h->h_shared_ov.size(), ntail * (ntail + 1) / 2 * kTile * kTile + ntail * kTile) + 64 + (size_t)h->world);
      h->d_xbuf2.resize((size_t)((int64_t)(h->od * h->od + h->od) * (int64_t)h->h_shared_ov.size()) + 64);
    }
    h->d_Hdiag.resize((size_t)(36 * h->nPv + h->od * h->od * h->nOv + 1));
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
    h->d_pose_c.resize((size_t)6 * P + 1); h->d_point_c.resize((size_t)3 * L + 1); h->d_obj_c.resize((size_t)h->od * O + 1);
    h->d_pose_b.resize((size_t)6 * P + 1); h->d_point_b.resize((size_t)3 * L + 1); h->d_obj_b.resize((size_t)h->od * O + 1);
    h->d_pc.resize(2 * ((size_t)P + 1)); h->d_pc_c.resize(2 * ((size_t)P + 1));   // records, then the field-major copy (k_pose_cache)
    finish_upload(h);  // host vectors above go out of scope
  }

  void remember_what_the_plan_was_built_for() {
    h->dirty = false; h->mask_dirty = false; h->pc_valid = false; h->tiles_cleared = false;
    h->plan_pose_vid = pose_vid; h->plan_obj_vid = obj_vid; h->plan_point_var = point_var; h->plan_is_pad = h->h_is_pad;
    h->plan_rp_active = h->h_rp_active; h->plan_bb_active = h->h_bb_active; h->plan_sp_active = h->h_sp_active; h->plan_lt_active = h->h_lt_active; h->plan_rl_active = h->h_rl_active;
    h->live_rows = h->m_canon;
  }
};

void prepare_plan(obvi_ba_handle* h) {
  if (!h->dirty && !h->mask_dirty) return;
  ApiTimer api_timer_(h->dirty ? "  prepare (symbolic phase)" : "  prepare (masks only)");
  if (!h->dirty && h->mask_dirty && prepare_masks(h)) return;
  PlanBuilder(h).run();
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
  for (int64_t i = 0; i < h->n_lt; ++i) if (h->h_lt_active[i] && !h->h_object_const[h->h_lt_obj[i]]) { nres += h->od; obj_used[h->h_lt_obj[i]] = 1; }
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
    else if (h->plan_obj_vid[o] >= 0) { const int64_t r = h->h_obj_row[h->plan_obj_vid[o]]; for (int k = 0; k < h->od; ++k) is_pad[r + k] = 1; }
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
  h->nLv = nL; h->live_rows = 6 * nP + h->od * nO;
  h->num_params = h->live_rows + 3 * nL;
  h->num_residuals = nres;
  finish_upload(h);   // (the copies went through the pinned arena: nothing to wait for; the solve's first launches follow on the same stream)
  h->mask_dirty = false; h->pc_valid = false; h->tiles_cleared = false;
  return true;
}

}  // namespace obvi_lib
