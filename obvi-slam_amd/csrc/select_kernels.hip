// select_kernels.hip -- K8: two-phase outlier selection on the device.
// Reference semantics (include/refactoring/offline/offline_problem_runner.h:769-800): the un-robustified squared
// norms of one factor type are put in a std::map keyed by the value in descending order (equal values collapse into
// one entry), n_outliers = floor(map.size() * fraction), the first n_outliers entries are excluded.
// Which member of a run of equal values represents it is unspecified in the reference (unordered_map iteration
// order); here it is the one with the highest factor index, the same rule the oracle uses.
//
// select_by_threshold -- four launches a round, no sort, no library primitive.  The values go into a hash table keyed by their bits (equal values meet in one
//      slot, which remembers the highest factor index: the table holds the std::map's entries), with a histogram of the keys' top 12 bits
//      (sign + exponent).  n_out = floor(entries * fraction); a radix select over the histogram finds the exponent bin the n_out-th
//      largest entry lies in, a second pass over the table excludes every entry above that bin and makes the histogram of the next 12
//      bits inside it, a third excludes above the second bin and collects the (few) entries inside it, and one workgroup ranks those.
//      The passes leave the table empty and the histograms at zero for the next call.
//      More than kSelCandCap distinct values sharing their top 24 bits (round 6; rounds 1-5 handed over to hipCUB's radix sort here): the last workgroup leaves
//      (prefix, how many of that bin are still wanted) behind and the caller runs the same four launches once more on the entries with that prefix, their keys
//      shifted left by 24 bits -- 24 + 24 + 16 bits: at most three rounds, one wait for the device per round.
#include "ba_device.h"

namespace obvi {
namespace {

constexpr unsigned long long kSelEmpty = ~0ull;
constexpr int kSelBins1 = 4096, kSelBins2 = 4096, kSelCandCap = 4096;   // (the first level of round 0 uses 2048 of its bins: the sign bit is clear)
// counters (ints): [0] entries of the table (distinct values), [1] candidates collected, [2] excluded (result), [3] candidate overflow
// carry (ints, survives a call): [0] round in flight (0: none), [1] entries of the open bin still wanted, [2] n_out of the selection, [3..4] the open bin's prefix (top 24 r bits, as a 64-bit value)
struct SelRound { int round; };   // 0: the whole table; r >= 1: only the entries whose top 24 r bits equal the carried prefix, keys shifted left by 24 r
__device__ __forceinline__ bool sel_in_round(unsigned long long key, int round, const int* carry, unsigned long long* sub) {
  if (round == 0) { *sub = key; return true; }
  const unsigned long long prefix = ((unsigned long long)(uint32_t)carry[4] << 32) | (uint32_t)carry[3];
  const int bits = 24 * round;
  if ((key >> (64 - bits)) != prefix) return false;
  *sub = key << bits;
  return true;
}
__device__ __forceinline__ unsigned long long sel_key(double v) {
  unsigned long long k = (unsigned long long)__double_as_longlong(v);
  if (k > 0x7ff0000000000000ull) k = 0x7ff8000000000000ull;   // NaN (either sign) and anything negative: one value above +inf
  return k;
}
__device__ __forceinline__ uint32_t sel_hash(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return (uint32_t)k;
}
__device__ __forceinline__ int sel_bin1(unsigned long long k) { return (int)(k >> 52); }            // round 0: < 2048, the sign bit is clear
__device__ __forceinline__ int sel_bin2(unsigned long long k) { return (int)((k >> 40) & 0xfffu); }

__global__ void __launch_bounds__(256) k_sel_insert(int64_t n, const double* __restrict__ sq, const uint8_t* __restrict__ active, const uint32_t* __restrict__ inv,
                                                    unsigned long long* __restrict__ tkeys, uint32_t* __restrict__ trep, uint32_t tmask, int* __restrict__ hist1,
                                                    int* __restrict__ counters, uint8_t* __restrict__ mask, int round, const int* __restrict__ carry) {
  __shared__ int lh[kSelBins1];
  __shared__ int fresh;
  for (int b = threadIdx.x; b < kSelBins1; b += 256) lh[b] = 0;
  if (threadIdx.x == 0) fresh = 0;
  __syncthreads();
  const int64_t i = blockIdx.x * 256LL + threadIdx.x;
  if (i < n) {
    const bool a = active[inv ? inv[i] : i] != 0;
    if (round == 0) mask[i] = a ? 1 : 0;            // (a later round only looks at the open bin: everything else has its answer)
    unsigned long long key = 0;
    if (a && sel_in_round(sel_key(sq[i]), round, carry, &key)) {
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
__device__ __forceinline__ int sel_n_out(const int* counters, double fraction, int round, const int* carry) { return round == 0 ? (int)((long long)((double)counters[0] * fraction)) : carry[1]; }

// pass over the table: entries above the first bin go; histogram of the next 12 bits inside it
__global__ void __launch_bounds__(256) k_sel_pass2(uint32_t tsize, const unsigned long long* __restrict__ tkeys, const uint32_t* __restrict__ trep, const int* __restrict__ hist1,
                                                   int* __restrict__ hist2, const int* __restrict__ counters, double fraction, uint8_t* __restrict__ mask, int round, const int* __restrict__ carry) {
  __shared__ int sm[6];
  __shared__ int lh[kSelBins2];
  for (int b = threadIdx.x; b < kSelBins2; b += 256) lh[b] = 0;
  int b1, in1;
  sel_find_bin<kSelBins1, 256>(hist1, sel_n_out(counters, fraction, round, carry), sm, &b1, &in1);
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
                                                   unsigned long long* __restrict__ cand_key, uint32_t* __restrict__ cand_rep, int round, const int* __restrict__ carry) {
  __shared__ int sm[6];
  int b1, in1, b2, in2;
  sel_find_bin<kSelBins1, 256>(hist1, sel_n_out(counters, fraction, round, carry), sm, &b1, &in1);
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
                                                    const unsigned long long* __restrict__ cand_key, const uint32_t* __restrict__ cand_rep, uint8_t* __restrict__ mask, int* __restrict__ result,
                                                    int round, int* __restrict__ carry) {
  __shared__ int sm[18];
  __shared__ unsigned long long keys[kSelCandCap];
  const int n_out = sel_n_out(counters, fraction, round, carry);
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
  if (threadIdx.x == 0) {
    const int total = round == 0 ? n_out : carry[2];          // floor(entries * fraction) of the whole selection
    if (overflow) {
      // the open bin goes to the next round: its prefix grows by (b1, b2), and `want` of its entries are still to be excluded
      const unsigned long long old = round == 0 ? 0ull : (((unsigned long long)(uint32_t)carry[4] << 32) | (uint32_t)carry[3]);
      const unsigned long long prefix = (old << 24) | ((unsigned long long)b1 << 12) | (unsigned long long)b2;
      carry[0] = round + 1; carry[1] = want; carry[2] = total; carry[3] = (int)(uint32_t)(prefix & 0xffffffffull); carry[4] = (int)(uint32_t)(prefix >> 32);
    } else {
      carry[0] = 0;
    }
    result[0] = total; result[1] = overflow ? round + 1 : 0; counters[0] = 0; counters[1] = 0; counters[2] = 0; counters[3] = 0;
  }
}

}  // namespace

// sq: per-factor squared norms of one type (device, caller order); active/inv: activity flags (indexed through inv if given); mask_out: device [n].
// Launches only.  *result_dev -> two ints on the device: the number excluded, and r > 0 if the last bin held more than kSelCandCap entries: the mask is then not
// finished and the caller runs round r (the same launches on that bin alone); r <= 2.
hipError_t select_by_threshold(hipStream_t s, int64_t n, const double* sq, const uint8_t* active, const uint32_t* inv, double fraction,
                               uint8_t* mask_out, SelectScratch* scratch, const int** result_dev, int round) {
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
  constexpr size_t kAuxInts = kSelBins1 + kSelBins2 + 16;
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
  int* carry = counters + 8;
  auto* cand_key = reinterpret_cast<unsigned long long*>(static_cast<char*>(scratch->aux) + kAuxInts * 4);
  auto* cand_rep = reinterpret_cast<uint32_t*>(cand_key + kSelCandCap);
  *result_dev = result;
  if (n == 0) return hipMemsetAsync(result, 0, 8, s);
  const unsigned tgrid = (unsigned)((tsize + 255) / 256);
  hipLaunchKernelGGL(k_sel_insert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, sq, active, inv, tkeys, trep, (uint32_t)(tsize - 1), hist1, counters, mask_out, round, carry);
  hipLaunchKernelGGL(k_sel_pass2, dim3(tgrid), dim3(256), 0, s, (uint32_t)tsize, tkeys, trep, hist1, hist2, counters, fraction, mask_out, round, carry);
  hipLaunchKernelGGL(k_sel_pass3, dim3(tgrid), dim3(256), 0, s, (uint32_t)tsize, tkeys, trep, hist1, hist2, counters, fraction, mask_out, cand_key, cand_rep, round, carry);
  hipLaunchKernelGGL(k_sel_final, dim3(1), dim3(1024), 0, s, hist1, hist2, counters, fraction, cand_key, cand_rep, mask_out, result, round, carry);
  return hipGetLastError();
}

void select_scratch_free(SelectScratch* sc) {
  void** ps[] = {&sc->tbl, &sc->aux};
  for (void** p : ps) if (*p) { (void)hipFree(*p); *p = nullptr; }
}

}  // namespace obvi
