// select_kernels.hip -- K8: two-phase outlier selection on the device.
// Reference semantics (include/refactoring/offline/offline_problem_runner.h:769-800): the un-robustified squared
// norms of one factor type are put in a std::map keyed by the value in descending order (equal values collapse into
// one entry), n_outliers = floor(map.size() * fraction), the first n_outliers entries are excluded.
//   1. compact (squared norm, factor index) of the active factors
//   2. radix sort by key, descending            (rocPRIM through hipCUB)
//   3. head-of-run flags -> exclusive scan = rank among the distinct values; distinct count
//   4. mask[i] = 0 for the run heads with rank < floor(distinct * fraction)
// Which member of a run of equal values represents it is unspecified in the reference (unordered_map iteration
// order); here it is the one with the highest factor index, the same rule the oracle uses.
#include <hipcub/hipcub.hpp>

#include "ba_device.h"

namespace obvi {
namespace {

__global__ void k_gather_active(int64_t n, const double* __restrict__ sq, const uint8_t* __restrict__ active, const uint32_t* __restrict__ inv,
                                unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals, uint8_t* __restrict__ mask, int* __restrict__ count) {
  // key = (double bits of the non-negative squared norm) -- orders like the double; ties broken by the index packed below
  const int64_t i = blockIdx.x * 256LL + threadIdx.x;
  if (i >= n) return;
  const bool a = active[inv ? inv[i] : i] != 0;
  mask[i] = a ? 1 : 0;
  if (a) {
    const int slot = atomicAdd(count, 1);
    keys[slot] = (unsigned long long)__double_as_longlong(sq[i]);
    vals[slot] = (uint32_t)i;
  }
}
// after the sort: equal keys are adjacent but in arbitrary index order; a run's representative = highest index
__global__ void k_run_heads(int n, const unsigned long long* __restrict__ keys, int* __restrict__ head) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}
__global__ void k_mark(int n, double fraction, const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals, const int* __restrict__ head,
                       const int* __restrict__ rank, uint8_t* __restrict__ mask, int* __restrict__ n_excluded) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int distinct = rank[n - 1] + head[n - 1];
  const int n_out = (int)((size_t)((double)distinct * fraction));
  if (i == 0) *n_excluded = n_out;
  if (!head[i] || rank[i] >= n_out) return;
  // representative of the run: the highest index among the equal keys
  uint32_t best = vals[i];
  for (int j = i + 1; j < n && keys[j] == keys[i]; ++j) best = max(best, vals[j]);
  mask[best] = 0;
}

}  // namespace

// sq: per-factor squared norms of one type (device, caller order); active/inv: activity flags (indexed through inv if
// given); mask_out: device [n].  Returns the number excluded through *n_excluded_host.  tmp buffers are grown as needed.
hipError_t select_outliers_device(hipStream_t s, int64_t n, const double* sq, const uint8_t* active, const uint32_t* inv, double fraction,
                                  uint8_t* mask_out, int* n_excluded_host, SelectScratch* scratch) {
  if (n == 0) { *n_excluded_host = 0; return hipSuccess; }
  hipError_t e;
  auto grow = [&](void** p, size_t* cap, size_t bytes) -> hipError_t {
    if (bytes <= *cap) return hipSuccess;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;                       // a failed allocation must not leave a dangling pointer / stale capacity behind
    const size_t want = bytes + bytes / 4 + 256;
    const hipError_t rc = hipMalloc(p, want);
    if (rc != hipSuccess) { *p = nullptr; return rc; }
    *cap = want;
    return hipSuccess;
  };
  if ((e = grow(&scratch->keys_in, &scratch->cap_keys_in, n * 8)) != hipSuccess) return e;
  if ((e = grow(&scratch->keys_out, &scratch->cap_keys_out, n * 8)) != hipSuccess) return e;
  if ((e = grow(&scratch->vals_in, &scratch->cap_vals_in, n * 4)) != hipSuccess) return e;
  if ((e = grow(&scratch->vals_out, &scratch->cap_vals_out, n * 4)) != hipSuccess) return e;
  if ((e = grow(&scratch->head, &scratch->cap_head, n * 4)) != hipSuccess) return e;
  if ((e = grow(&scratch->rank, &scratch->cap_rank, n * 4)) != hipSuccess) return e;
  if ((e = grow(&scratch->counters, &scratch->cap_counters, 64)) != hipSuccess) return e;
  int* counters = static_cast<int*>(scratch->counters);
  if ((e = hipMemsetAsync(counters, 0, 64, s)) != hipSuccess) return e;
  auto* keys_in = static_cast<unsigned long long*>(scratch->keys_in);
  auto* keys_out = static_cast<unsigned long long*>(scratch->keys_out);
  auto* vals_in = static_cast<uint32_t*>(scratch->vals_in);
  auto* vals_out = static_cast<uint32_t*>(scratch->vals_out);
  hipLaunchKernelGGL(k_gather_active, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, sq, active, inv, keys_in, vals_in, mask_out, counters);
  int n_act = 0;
  if ((e = hipMemcpyAsync(&n_act, counters, sizeof(int), hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
  if (n_act == 0) { *n_excluded_host = 0; return hipSuccess; }
  size_t tmp_bytes = 0;
  if ((e = hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n_act, 0, 64, s)) != hipSuccess) return e;
  size_t scan_bytes = 0;
  if ((e = hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, static_cast<int*>(scratch->head), static_cast<int*>(scratch->rank), n_act, s)) != hipSuccess) return e;
  if ((e = grow(&scratch->tmp, &scratch->cap_tmp, std::max(tmp_bytes, scan_bytes))) != hipSuccess) return e;
  if ((e = hipcub::DeviceRadixSort::SortPairsDescending(scratch->tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n_act, 0, 64, s)) != hipSuccess) return e;
  hipLaunchKernelGGL(k_run_heads, dim3((unsigned)((n_act + 255) / 256)), dim3(256), 0, s, n_act, keys_out, static_cast<int*>(scratch->head));
  if ((e = hipcub::DeviceScan::ExclusiveSum(scratch->tmp, scan_bytes, static_cast<int*>(scratch->head), static_cast<int*>(scratch->rank), n_act, s)) != hipSuccess) return e;
  hipLaunchKernelGGL(k_mark, dim3((unsigned)((n_act + 255) / 256)), dim3(256), 0, s, n_act, fraction, keys_out, vals_out, static_cast<int*>(scratch->head),
                     static_cast<int*>(scratch->rank), mask_out, counters + 1);
  if ((e = hipMemcpyAsync(n_excluded_host, counters + 1, sizeof(int), hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
  return hipStreamSynchronize(s);
}

void select_scratch_free(SelectScratch* sc) {
  void** ps[] = {&sc->keys_in, &sc->keys_out, &sc->vals_in, &sc->vals_out, &sc->head, &sc->rank, &sc->counters, &sc->tmp};
  for (void** p : ps) if (*p) { (void)hipFree(*p); *p = nullptr; }
}

}  // namespace obvi
