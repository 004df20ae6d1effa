// plan_kernels.hip -- the part of the symbolic phase that runs on the device (round 5): the slot tables of the Schur strip kernel.
//
// k_schur_window gathers, per visit (point x row chunk x column group), the point's Z records of every strip frame an active tile touches into
// LDS: the plan holds, per visit, four words (where the row / column operands start in the batch image, the tail slot, tile masks) and, per slot,
// the source of its 144 bytes.  Config #3: 0.8 M visits, 8.65 M slots.  Until round 4 the host filled both tables (a walk over every visit's
// observations: 95 ms on one thread, 26 ms on sixteen) and uploaded 48 MB; the walk is the same for every visit and independent of the others
// once the host has dealt the visits to batches (sizes depend on a visit's tile bits alone), so it runs here: one lane per visit.
// Same arithmetic, same order as plan.cpp's host version (OBVI_PLAN_SLOTS_ON_HOST=1), which stays as the check.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ba_device.h"

namespace obvi {
namespace {

constexpr int W = kSchurWindowFrames, SR = kSchurRows, SBACK = W - SR, kRowTile0 = SBACK * 6 / 16;

__global__ void __launch_bounds__(256) k_plan_visit_slots(int64_t nvis, const PlanVisit* __restrict__ pv, const uint32_t* __restrict__ wg_ptr, const uint32_t* __restrict__ wg_slot0, int32_t nwg,
                                                         const int32_t* __restrict__ wg_f0, const int32_t* __restrict__ wg_group, const uint32_t* __restrict__ point_ptr,
                                                         const uint8_t* __restrict__ rp_active, const uint32_t* __restrict__ rp_pose, const int32_t* __restrict__ frame_of_pose,
                                                         uint32_t zero16, uint32_t* __restrict__ visits, uint32_t* __restrict__ slot_src) {
  const int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (t >= nvis) return;
  // the workgroup (slice of a (chunk, group) work list) the visit belongs to: the last g with wg_ptr[g] <= t
  int32_t lo_g = 0, hi_g = nwg - 1;
  while (lo_g < hi_g) { const int32_t mid = (lo_g + hi_g + 1) >> 1; if ((int64_t)wg_ptr[mid] <= t) lo_g = mid; else hi_g = mid - 1; }
  const int32_t g = lo_g, group = wg_group[g], fbase = wg_f0[g] - SBACK;
  const PlanVisit v = pv[t];
  const uint32_t bits = v.twin_bits & 0x7fffu;
  const bool twin = (v.twin_bits & 0x8000u) != 0;
  const uint32_t base = v.base;
  uint32_t rows = 0, cols = 0;
  int32_t A0 = INT32_MAX, A1 = -1, B0 = INT32_MAX, B1 = -1;
#pragma unroll
  for (int c = 0; c < kSchurGroupCols; ++c) {
    const uint32_t t3 = (bits >> (3 * c)) & 7u;
    if (!t3) continue;
    rows |= t3; cols |= 1u << c;
    const int tc = kSchurGroupCols * group + c;
    B0 = min(B0, (16 * tc) / 6); B1 = max(B1, (16 * tc + 15) / 6);
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
    if ((rows >> r) & 1u) { const int tr = kRowTile0 + r; A0 = min(A0, (16 * tr) / 6); A1 = max(A1, (16 * tr + 15) / 6); }
  A1 = min(A1, W - 1); B1 = min(B1, W - 1);
  uint32_t prim[W], sec[W];
#pragma unroll
  for (int i = 0; i < W; ++i) prim[i] = sec[i] = zero16;
  const uint32_t beg = point_ptr[v.l], end = point_ptr[v.l + 1];
  for (uint32_t a = beg; a < end; ++a) {
    if (!rp_active[a]) continue;
    const int32_t f = frame_of_pose[rp_pose[a]];
    if (f < 0) continue;
    const int32_t fo = f - fbase;
    if (fo < 0 || fo >= W) continue;
    const uint32_t src = (uint32_t)((18ull * a + 4ull * v.l) / 2);
    if (prim[fo] == zero16) prim[fo] = src; else sec[fo] = src;
  }
  uint32_t* out = slot_src + (size_t)wg_slot0[g] + v.rel;
  const uint32_t tail_src = (uint32_t)((18ull * end + 4ull * v.l) / 2);   // z_tail(): (u_l, 0) behind the point's records
  int32_t slotA0, slotB0;
  uint32_t tail, n = 0;
  const bool merged = B0 <= A1 + 1 && A0 <= B1 + 1;
  if (merged) {
    const int32_t lo = min(A0, B0), hi = max(A1, B1);
    for (int32_t fo = lo; fo <= hi; ++fo) out[n++] = prim[fo];
    slotA0 = slotB0 = (int32_t)base - lo; tail = base + (uint32_t)(hi - lo + 1);
    out[n++] = tail_src;
    if (twin) for (int32_t fo = lo; fo <= hi; ++fo) out[n++] = sec[fo];
  } else {
    for (int32_t fo = A0; fo <= A1; ++fo) out[n++] = prim[fo];
    slotA0 = (int32_t)base - A0; tail = base + (uint32_t)(A1 - A0 + 1); slotB0 = (int32_t)tail + 1 - B0;
    out[n++] = tail_src;
    for (int32_t fo = B0; fo <= B1; ++fo) out[n++] = prim[fo];
    if (twin) {
      for (int32_t fo = A0; fo <= A1; ++fo) out[n++] = sec[fo];
      out[n++] = zero16;
      for (int32_t fo = B0; fo <= B1; ++fo) out[n++] = sec[fo];
    }
  }
  const uint32_t layer = twin ? n / 2 + (merged ? 1u : 0u) : 0u;   // slots from a record to its second-layer twin
  uint32_t* rec = visits + 4 * (size_t)t;
  rec[0] = (uint32_t)(144 * slotA0);
  rec[1] = (uint32_t)(144 * slotB0);
  rec[2] = tail | (layer << 16);
  rec[3] = cols | (twin ? 1u << 15 : 0u) | (rows << 16);
}

}  // namespace

void launch_plan_visit_slots(hipStream_t s, int64_t nvis, const PlanVisit* pv, const uint32_t* wg_ptr, const uint32_t* wg_slot0, int32_t nwg, const int32_t* wg_f0, const int32_t* wg_group,
                             const uint32_t* point_ptr, const uint8_t* rp_active, const uint32_t* rp_pose, const int32_t* frame_of_pose, uint32_t zero16, uint32_t* visits, uint32_t* slot_src) {
  if (nvis > 0) hipLaunchKernelGGL(k_plan_visit_slots, dim3((unsigned)((nvis + 255) / 256)), dim3(256), 0, s, nvis, pv, wg_ptr, wg_slot0, nwg, wg_f0, wg_group, point_ptr, rp_active, rp_pose,
                                   frame_of_pose, zero16, visits, slot_src);
}

}  // namespace obvi
