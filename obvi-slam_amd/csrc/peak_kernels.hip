// peak_kernels.hip -- obvi_ba_measure_peaks (include/obvi_ba.h, profiling hooks): what THIS device delivers for the two resources the
// kernels of the path are priced against ("peaks ... to be re-measured on the box with a triad and a DGEMM microbenchmark;
// report both fractions").  bench.py calls it in-process and reports every roofline fraction against the public figure AND against these.
//
//   HBM     triad   a[i] = b[i] + s c[i]   over three arrays of `bytes` each (16-byte loads / stores, grid-stride): 3 x bytes moved
//           copy    a[i] = b[i]                                                                                   2 x bytes moved
//           read    sum of b[i] (no store)                                                                        1 x bytes moved
//   MFMA    issue   every wavefront runs a chain of independent v_mfma_f64_16x16x4_f64 (8 accumulators, operands in registers): the
//                   rate the matrix pipes issue at, 2048 flop per instruction
//           tile    the product of the tile Cholesky's update jobs (chol_kernels.hip: C += A B^T, 64x64x64, both operands in LDS with
//                   leading dimension 66, four wavefronts, 64 instructions each) repeated on operands staged once: what a 64x64-tile
//                   GEMM built from this instruction reaches when it only has to feed the pipes from LDS
#include <algorithm>
#include <vector>

#include "ba_device.h"
#include "host_util.h"
#include "../../include/obvi_ba.h"

namespace obvi {
namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_peak_triad(double2* __restrict__ a, const double2* __restrict__ b, const double2* __restrict__ c, int64_t n2, double s) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
    const double2 x = b[i], y = c[i];
    a[i] = make_double2(x.x + s * y.x, x.y + s * y.y);
  }
}
__global__ void __launch_bounds__(256) k_peak_copy(double2* __restrict__ a, const double2* __restrict__ b, int64_t n2) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) a[i] = b[i];
}
__global__ void __launch_bounds__(256) k_peak_read(const double2* __restrict__ b, int64_t n2, double* out) {
  double s = 0.0;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) { const double2 v = b[i]; s += v.x + v.y; }
  if (s == 1.2345e300) out[0] = s;
}

// the accumulator tied in place in VGPRs (asm): the builtin form lets the compiler park the accumulators in AGPRs and copy all of them
// back and forth around every trip of the loop, which measures the copies
__device__ __forceinline__ void mfma_acc(f64x4& acc, double a, double b) {
  asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
constexpr int kIssueAcc = 8;
__global__ void __launch_bounds__(256) k_peak_mfma_issue(double* out, int iters, double a0, double b0) {
  f64x4 acc[kIssueAcc];
#pragma unroll
  for (int i = 0; i < kIssueAcc; ++i) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
  const double a = a0 + 1e-9 * (threadIdx.x & 63), b = b0 - 1e-9 * (threadIdx.x & 63);
  asm volatile("s_nop 4" ::: "memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < kIssueAcc; ++i) mfma_acc(acc[i], a, b);
  }
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // MFMA result -> VALU read
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < kIssueAcc; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1.2345e300) out[0] = s;
}

constexpr int T = kTile, LDM = T + 2;
__global__ void __launch_bounds__(256) k_peak_mfma_tile(const double* __restrict__ tiles, double* out, int products) {
  __shared__ double A[T * LDM];
  __shared__ double B[T * LDM];
  for (int e = threadIdx.x; e < T * T; e += 256) { A[(e / T) * LDM + e % T] = tiles[e]; B[(e / T) * LDM + e % T] = tiles[T * T + e]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r16 = lane & 15, kq = lane >> 4;
  f64x4 acc[4] = {};
  const double* Bp = B + (16 * wv + r16) * LDM + kq;
  const double* Ap = A + r16 * LDM + kq;
  for (int p = 0; p < products; ++p) {
#pragma unroll 4
    for (int k0 = 0; k0 < T; k0 += 4) {
      const double bv = Bp[k0];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ap[16 * rt * LDM + k0], bv, acc[rt], 0, 0, 0);
    }
  }
  double s = 0.0;
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) s += acc[rt][0] + acc[rt][1] + acc[rt][2] + acc[rt][3];
  if (s == 1.2345e300) out[0] = s;
}

template <class F>
double best_ms(hipStream_t s, int reps, F&& launch) {
  hipEvent_t a, b;
  OBVI_HIP(hipEventCreate(&a)); OBVI_HIP(hipEventCreate(&b));
  launch();   // warm-up
  OBVI_HIP(hipStreamSynchronize(s));
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    OBVI_HIP(hipEventRecord(a, s));
    launch();
    OBVI_HIP(hipEventRecord(b, s));
    OBVI_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    OBVI_HIP(hipEventElapsedTime(&ms, a, b));
    best = std::min(best, ms);
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  return best;
}

}  // namespace
}  // namespace obvi

using namespace obvi;  // NOLINT

extern "C" int obvi_ba_measure_peaks(obvi_ba_handle* h, obvi_measured_peaks* out) {
  if (!h || !out) return OBVI_ERR_INVALID_ARGUMENT;
  try {
    OBVI_HIP(hipSetDevice(handle_device(h)));
    hipStream_t s = handle_stream(h);
    hipDeviceProp_t prop;
    OBVI_HIP(hipGetDeviceProperties(&prop, handle_device(h)));
    const int cus = prop.multiProcessorCount;
    // ---- HBM: three arrays of 1 GiB (far beyond the 256 MB last-level cache)
    const int64_t bytes = (int64_t)1 << 30, n2 = bytes / 16;
    DevBuf<double> a, b, c, o;
    a.resize((size_t)(bytes / 8)); b.resize((size_t)(bytes / 8)); c.resize((size_t)(bytes / 8)); o.resize(8);
    OBVI_HIP(hipMemsetAsync(a.get(), 0, bytes, s)); OBVI_HIP(hipMemsetAsync(b.get(), 0, bytes, s)); OBVI_HIP(hipMemsetAsync(c.get(), 0, bytes, s));
    const unsigned grid = (unsigned)(cus * 32);
    double2 *pa = reinterpret_cast<double2*>(a.get()), *pb = reinterpret_cast<double2*>(b.get()), *pc = reinterpret_cast<double2*>(c.get());
    const double t_triad = best_ms(s, 5, [&] { hipLaunchKernelGGL(k_peak_triad, dim3(grid), dim3(256), 0, s, pa, pb, pc, n2, 0.5); });
    const double t_copy = best_ms(s, 5, [&] { hipLaunchKernelGGL(k_peak_copy, dim3(grid), dim3(256), 0, s, pa, pb, n2); });
    const double t_read = best_ms(s, 5, [&] { hipLaunchKernelGGL(k_peak_read, dim3(grid), dim3(256), 0, s, pb, n2, o.get()); });
    out->hbm_triad_gbs = 3.0 * bytes / (1e-3 * t_triad) / 1e9;
    out->hbm_copy_gbs = 2.0 * bytes / (1e-3 * t_copy) / 1e9;
    out->hbm_read_gbs = 1.0 * bytes / (1e-3 * t_read) / 1e9;
    // ---- fp64 MFMA: 4 wavefronts per workgroup (one per SIMD), 2 workgroups per compute unit
    const int iters = 4096;
    const unsigned g2 = (unsigned)(cus * 2);
    const double t_issue = best_ms(s, 5, [&] { hipLaunchKernelGGL(k_peak_mfma_issue, dim3(g2), dim3(256), 0, s, o.get(), iters, 1.0, 0.5); });
    out->mfma_f64_issue_tflops = (double)g2 * 4.0 * iters * kIssueAcc * 2048.0 / (1e-3 * t_issue) / 1e12;
    std::vector<double> tiles(2 * T * T);
    for (size_t i = 0; i < tiles.size(); ++i) tiles[i] = 1e-3 * (double)((i * 2654435761u) % 1000u);
    DevBuf<double> dt;
    dt.upload(tiles, s);
    OBVI_HIP(hipStreamSynchronize(s));
    const int products = 256;
    const double t_tile = best_ms(s, 5, [&] { hipLaunchKernelGGL(k_peak_mfma_tile, dim3(g2), dim3(256), 0, s, dt.get(), o.get(), products); });
    out->mfma_f64_tile_tflops = (double)g2 * products * 2.0 * T * T * T / (1e-3 * t_tile) / 1e12;
    out->compute_units = cus;
    out->clock_mhz = prop.clockRate / 1000.0;
    OBVI_HIP(hipStreamSynchronize(s));
    return OBVI_OK;
  } catch (const HipError& e) {
    return handle_fail(h, OBVI_ERR_HIP, e.what);
  } catch (...) {
    return handle_fail(h, OBVI_ERR_HIP, "measure_peaks: host exception");
  }
}
