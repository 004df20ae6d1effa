// select_kernels.hip -- K8: two-phase outlier selection on the device.
// Reference semantics (include/refactoring/offline/offline_problem_runner.h:769-800): the un-robustified squared
// norms of one factor type are put in a std::map keyed by the value in descending order (equal values collapse into
// one entry), n_outliers = floor(map.size() * fraction), the first n_outliers entries are excluded.
// Which member of a run of equal values represents it is unspecified in the reference (unordered_map iteration
// order); here it is the one with the highest factor index, the same rule the oracle uses.
//
// Two routes to the same mask:
//  (a) select_by_threshold -- four launches, no sort.  The values go into a hash table keyed by their bits (equal values meet in one
//      slot, which remembers the highest factor index: the table holds the std::map's entries), with a histogram of the keys' top 12 bits
//      (sign + exponent).  n_out = floor(entries * fraction); a radix select over the histogram finds the exponent bin the n_out-th
//      largest entry lies in, a second pass over the table excludes every entry above that bin and makes the histogram of the next 12
//      bits inside it, a third excludes above the second bin and collects the (few) entries inside it, and one workgroup ranks those.
//      The passes leave the table empty and the histograms at zero for the next call.
//  (b) select_by_sort -- (squared norm, index) of the active factors, radix sort descending (rocPRIM through hipCUB), head-of-run flags,
//      exclusive scan = rank among the distinct values, mask[i] = 0 for the run heads with rank < n_out.  About 25 launches and two waits
//      for the device; kept for the case (a) cannot finish (more than kSelCandCap distinct values share their top 24 bits) and as its
//      check (OBVI_SELECT_SORT=1; tests/test_gpu_parity.py runs both on the same inputs).
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


// ---- (a) ------------------------------------------------------------------------------------------------------------------------
constexpr unsigned long long kSelEmpty = ~0ull;
constexpr int kSelBins1 = 2048, kSelBins2 = 4096, kSelCandCap = 4096;
// counters (ints): [0] entries of the table (distinct values), [1] candidates collected, [2] excluded (result), [3] candidate overflow
__device__ __forceinline__ unsigned long long sel_key(double v) {
  unsigned long long k = (unsigned long long)__double_as_longlong(v);
  if (k > 0x7ff0000000000000ull) k = 0x7ff8000000000000ull;   // NaN (either sign) and anything negative: one value above +inf
  return k;
}
__device__ __forceinline__ uint32_t sel_hash(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return (uint32_t)k;
}
__device__ __forceinline__ int sel_bin1(unsigned long long k) { return (int)(k >> 52); }            // < 2048: the sign bit is clear
__device__ __forceinline__ int sel_bin2(unsigned long long k) { return (int)((k >> 40) & 0xfffu); }

__global__ void __launch_bounds__(256) k_sel_insert(int64_t n, const double* __restrict__ sq, const uint8_t* __restrict__ active, const uint32_t* __restrict__ inv,
                                                    unsigned long long* __restrict__ tkeys, uint32_t* __restrict__ trep, uint32_t tmask, int* __restrict__ hist1,
                                                    int* __restrict__ counters, uint8_t* __restrict__ mask) {
  __shared__ int lh[kSelBins1];
  __shared__ int fresh;
  for (int b = threadIdx.x; b < kSelBins1; b += 256) lh[b] = 0;
  if (threadIdx.x == 0) fresh = 0;
  __syncthreads();
  const int64_t i = blockIdx.x * 256LL + threadIdx.x;
  if (i < n) {
    const bool a = active[inv ? inv[i] : i] != 0;
    mask[i] = a ? 1 : 0;
    if (a) {
      const unsigned long long key = sel_key(sq[i]);
      uint32_t slot = sel_hash(key) & tmask;
      for (;;) {
        const unsigned long long prev = atomicCAS(&tkeys[slot], kSelEmpty, key);
        if (prev == kSelEmpty) { atomicAdd(&lh[sel_bin1(key)], 1); atomicAdd(&fresh, 1); }
        if (prev == kSelEmpty || prev == key) { atomicMax(&trep[slot], (uint32_t)i); break; }
        slot = (slot + 1) & tmask;
      }
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < kSelBins1; b += 256) if (lh[b]) atomicAdd(&hist1[b], lh[b]);
  if (threadIdx.x == 0 && fresh) atomicAdd(&counters[0], fresh);
}

// The bin, counted from the top, in which the `want`-th largest entry lies, found by the whole workgroup (kThreads threads, every one of
// them must call): *bin, and how many of that bin's entries are still wanted (*inside, >= 1).  want == 0: *bin = kBins (every entry lies
// below it), *inside = 0.  Thread t owns the kBins / kThreads bins below kBins - (kBins / kThreads) t; a workgroup-wide inclusive scan of the
// per-thread counts tells the one thread whose range holds the target to walk its bins.
template <int kBins, int kThreads>
__device__ void sel_find_bin(const int* __restrict__ hist, int want, int* sm /* [kThreads / 64 + 2] */, int* bin, int* inside) {
  constexpr int kPer = kBins / kThreads, kWaves = kThreads / 64;
  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  int mine = 0;
  for (int k = 0; k < kPer; ++k) mine += hist[kBins - 1 - (kPer * t + k)];
  int incl = mine;
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
  if (t == 0) { sm[kWaves] = kBins; sm[kWaves + 1] = 0; }
  if (lane == 63) sm[wave] = incl;
  __syncthreads();
  int before = 0;
  for (int w = 0; w < wave; ++w) before += sm[w];
  incl += before;
  const int excl = incl - mine;
  if (want > 0 && excl < want && want <= incl) {
    int above = excl;
    for (int k = 0; k < kPer; ++k) {
      const int bb = kBins - 1 - (kPer * t + k), c = hist[bb];
      if (above + c >= want) { sm[kWaves] = bb; sm[kWaves + 1] = want - above; break; }
      above += c;
    }
  }
  __syncthreads();
  *bin = sm[kWaves]; *inside = sm[kWaves + 1];
  __syncthreads();
}
__device__ __forceinline__ int sel_n_out(const int* counters, double fraction) { return (int)((long long)((double)counters[0] * fraction)); }

// pass over the table: entries above the first bin go; histogram of the next 12 bits inside it
__global__ void __launch_bounds__(256) k_sel_pass2(uint32_t tsize, const unsigned long long* __restrict__ tkeys, const uint32_t* __restrict__ trep, const int* __restrict__ hist1,
                                                   int* __restrict__ hist2, const int* __restrict__ counters, double fraction, uint8_t* __restrict__ mask) {
  __shared__ int sm[6];
  __shared__ int lh[kSelBins2];
  for (int b = threadIdx.x; b < kSelBins2; b += 256) lh[b] = 0;
  int b1, in1;
  sel_find_bin<kSelBins1, 256>(hist1, sel_n_out(counters, fraction), sm, &b1, &in1);
  const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
  bool any = false;
  if (slot < tsize) {
    const unsigned long long key = tkeys[slot];
    if (key != kSelEmpty) {
      const int e = sel_bin1(key);
      if (e > b1) mask[trep[slot]] = 0;
      else if (e == b1) { atomicAdd(&lh[sel_bin2(key)], 1); any = true; }
    }
  }
  if (__syncthreads_or(any)) for (int b = threadIdx.x; b < kSelBins2; b += 256) if (lh[b]) atomicAdd(&hist2[b], lh[b]);
}
// second pass: entries of the first bin above the second bin go, those inside it are collected; the table is left empty
__global__ void __launch_bounds__(256) k_sel_pass3(uint32_t tsize, unsigned long long* __restrict__ tkeys, uint32_t* __restrict__ trep, const int* __restrict__ hist1,
                                                   const int* __restrict__ hist2, int* __restrict__ counters, double fraction, uint8_t* __restrict__ mask,
                                                   unsigned long long* __restrict__ cand_key, uint32_t* __restrict__ cand_rep) {
  __shared__ int sm[6];
  int b1, in1, b2, in2;
  sel_find_bin<kSelBins1, 256>(hist1, sel_n_out(counters, fraction), sm, &b1, &in1);
  sel_find_bin<kSelBins2, 256>(hist2, in1, sm, &b2, &in2);
  const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
  if (slot >= tsize) return;
  const unsigned long long key = tkeys[slot];
  if (key == kSelEmpty) return;
  const uint32_t rep = trep[slot];
  tkeys[slot] = kSelEmpty; trep[slot] = 0;
  if (sel_bin1(key) != b1) return;
  const int m = sel_bin2(key);
  if (m > b2) mask[rep] = 0;
  else if (m == b2) {
    const int at = atomicAdd(&counters[1], 1);
    if (at < kSelCandCap) { cand_key[at] = key; cand_rep[at] = rep; } else counters[3] = 1;
  }
}
// the last bin's entries, ranked by one workgroup; the result; everything back to zero for the next call
__global__ void __launch_bounds__(1024) k_sel_final(int* __restrict__ hist1, int* __restrict__ hist2, int* __restrict__ counters, double fraction,
                                                    const unsigned long long* __restrict__ cand_key, const uint32_t* __restrict__ cand_rep, uint8_t* __restrict__ mask, int* __restrict__ result) {
  __shared__ int sm[18];
  __shared__ unsigned long long keys[kSelCandCap];
  const int n_out = sel_n_out(counters, fraction);
  const bool overflow = counters[3] != 0;
  const int nc = min(counters[1], kSelCandCap);
  int b1, in1, b2, want;
  sel_find_bin<kSelBins1, 1024>(hist1, n_out, sm, &b1, &in1);
  sel_find_bin<kSelBins2, 1024>(hist2, in1, sm, &b2, &want);
  if (!overflow && want > 0) {
    for (int c = threadIdx.x; c < nc; c += 1024) keys[c] = cand_key[c];
    __syncthreads();
    for (int c = threadIdx.x; c < nc; c += 1024) {
      const unsigned long long k = keys[c];
      int larger = 0;
      for (int j = 0; j < nc; ++j) larger += keys[j] > k ? 1 : 0;   // the keys of the table are distinct: a rank, no ties
      if (larger < want) mask[cand_rep[c]] = 0;
    }
  }
  __syncthreads();   // everybody has read the histograms and the counters
  for (int b = threadIdx.x; b < kSelBins1; b += 1024) hist1[b] = 0;
  for (int b = threadIdx.x; b < kSelBins2; b += 1024) hist2[b] = 0;
  if (threadIdx.x == 0) { result[0] = n_out; result[1] = overflow ? 1 : 0; counters[0] = 0; counters[1] = 0; counters[2] = 0; counters[3] = 0; }
}

}  // namespace

// sq: per-factor squared norms of one type (device, caller order); active/inv: activity flags (indexed through inv if
// given); mask_out: device [n].  (b) returns the number excluded through *n_excluded_host (it waits for the device twice).  Buffers are grown as needed.
static hipError_t select_by_sort(hipStream_t s, int64_t n, const double* sq, const uint8_t* active, const uint32_t* inv, double fraction,
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

// (a): launches only.  *result_dev -> two ints on the device: the number excluded, and 1 if the last bin held more than kSelCandCap
// entries (then the mask is not finished: take route (b)).
hipError_t select_by_threshold(hipStream_t s, int64_t n, const double* sq, const uint8_t* active, const uint32_t* inv, double fraction,
                               uint8_t* mask_out, SelectScratch* scratch, const int** result_dev) {
  hipError_t e;
  // table: a power of two of at least twice the values, so that probe sequences stay short
  size_t slots = 1024;
  while (slots < 2 * (size_t)std::max<int64_t>(n, 1)) slots *= 2;
  if (slots > scratch->tbl_slots) {
    if (scratch->tbl) (void)hipFree(scratch->tbl);
    scratch->tbl = nullptr; scratch->tbl_slots = 0;
    if ((e = hipMalloc(&scratch->tbl, slots * 12)) != hipSuccess) { scratch->tbl = nullptr; return e; }
    scratch->tbl_slots = slots;
    if ((e = hipMemsetAsync(scratch->tbl, 0xff, slots * 8, s)) != hipSuccess) return e;                                   // keys: empty
    if ((e = hipMemsetAsync(static_cast<char*>(scratch->tbl) + slots * 8, 0, slots * 4, s)) != hipSuccess) return e;      // representatives
  }
  constexpr size_t kAuxInts = kSelBins1 + kSelBins2 + 8;
  constexpr size_t kAuxBytes = kAuxInts * 4 + (size_t)kSelCandCap * 12;
  if (!scratch->aux) {
    if ((e = hipMalloc(&scratch->aux, kAuxBytes)) != hipSuccess) { scratch->aux = nullptr; return e; }
    if ((e = hipMemsetAsync(scratch->aux, 0, kAuxBytes, s)) != hipSuccess) return e;
  }
  // every call leaves the table empty and the histograms / counters at zero (k_sel_pass3, k_sel_final); they are sized for the largest n so far
  const size_t tsize = scratch->tbl_slots;
  auto* tkeys = static_cast<unsigned long long*>(scratch->tbl);
  auto* trep = reinterpret_cast<uint32_t*>(static_cast<char*>(scratch->tbl) + tsize * 8);
  int* hist1 = static_cast<int*>(scratch->aux);
  int* hist2 = hist1 + kSelBins1;
  int* counters = hist2 + kSelBins2;
  int* result = counters + 4;
  auto* cand_key = reinterpret_cast<unsigned long long*>(static_cast<char*>(scratch->aux) + kAuxInts * 4);
  auto* cand_rep = reinterpret_cast<uint32_t*>(cand_key + kSelCandCap);
  *result_dev = result;
  if (n == 0) return hipMemsetAsync(result, 0, 8, s);
  const unsigned tgrid = (unsigned)((tsize + 255) / 256);
  hipLaunchKernelGGL(k_sel_insert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, sq, active, inv, tkeys, trep, (uint32_t)(tsize - 1), hist1, counters, mask_out);
  hipLaunchKernelGGL(k_sel_pass2, dim3(tgrid), dim3(256), 0, s, (uint32_t)tsize, tkeys, trep, hist1, hist2, counters, fraction, mask_out);
  hipLaunchKernelGGL(k_sel_pass3, dim3(tgrid), dim3(256), 0, s, (uint32_t)tsize, tkeys, trep, hist1, hist2, counters, fraction, mask_out, cand_key, cand_rep);
  hipLaunchKernelGGL(k_sel_final, dim3(1), dim3(1024), 0, s, hist1, hist2, counters, fraction, cand_key, cand_rep, mask_out, result);
  return hipGetLastError();
}

hipError_t select_outliers_sorted(hipStream_t s, int64_t n, const double* sq, const uint8_t* active, const uint32_t* inv, double fraction,
                                  uint8_t* mask_out, int* n_excluded_host, SelectScratch* scratch) {
  return select_by_sort(s, n, sq, active, inv, fraction, mask_out, n_excluded_host, scratch);
}

void select_scratch_free(SelectScratch* sc) {
  void** ps[] = {&sc->keys_in, &sc->keys_out, &sc->vals_in, &sc->vals_out, &sc->head, &sc->rank, &sc->counters, &sc->tmp, &sc->tbl, &sc->aux};
  for (void** p : ps) if (*p) { (void)hipFree(*p); *p = nullptr; }
}

}  // namespace obvi
