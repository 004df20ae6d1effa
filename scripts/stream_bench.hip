// Streaming-read patterns over a Z-sized array (3 M records of 144 B): what the memory system gives each access shape.
// Build (repo root): hipcc --offload-arch=gfx950 -O3 -o scripts/stream_bench scripts/stream_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kRec = 18;   // doubles per record
// (a) coalesced: lane i of the grid reads piece i, i + stride, ...
__global__ void __launch_bounds__(256) k_coalesced(const double2* __restrict__ z, int64_t n, double* out) {
  double s = 0;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) { const double2 v = z[e]; s += v.x + v.y; }
  if (s == 1.2345) out[0] = s;
}
// (b) a lane per feature, per records each, nine 16-byte loads per record
__global__ void __launch_bounds__(256) k_lane_per_feature(const double2* __restrict__ z, int64_t L, int per, double* out) {
  const int64_t l = blockIdx.x * 256ll + threadIdx.x;
  if (l >= L) return;
  const double2* p = z + l * per * 9;
  double s = 0;
  for (int a = 0; a < per; ++a) {
#pragma unroll
    for (int x = 0; x < 9; ++x) { const double2 v = p[a * 9 + x]; s += v.x + v.y; }
  }
  if (s == 1.2345) out[0] = s;
}
// (c) G lanes per feature, pieces of the feature's run dealt round-robin, plus the row lookup and two y gathers
template <int G, bool GATHER>
__global__ void __launch_bounds__(256) k_pieces(const double2* __restrict__ z, const int32_t* __restrict__ yrow, const double* __restrict__ y, int64_t L, int per, double* out) {
  const uint32_t g = threadIdx.x % G;
  double s = 0;
  for (int64_t l = (blockIdx.x * 256ll + threadIdx.x) / G; l < L; l += (int64_t)gridDim.x * (256 / G)) {
    const double2* p = z + l * per * 9;
    const uint32_t n = 9 * per;
    for (uint32_t e = g; e < n; e += G) {
      const double2 v = p[e];
      if (GATHER) {
        const uint32_t rec = e / 9, w = 2 * (e - 9 * rec);
        const int32_t yr = yrow[l * per + rec];
        s += v.x * y[yr + w / 3] + v.y * y[yr + (w + 1) / 3];
      } else s += v.x + v.y;
    }
  }
  if (s == 1.2345) out[0] = s;
}
template <class F> static void timeit(const char* name, double bytes, F f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("%-40s %8.1f us  %6.2f TB/s\n", name, 1e3 * ms / reps, bytes / (1e-3 * ms / reps) / 1e12);
}
int main() {
  const int64_t L = 50000; const int per = 60, P = 500;
  const int64_t nrec = L * per, npieces = nrec * 9;
  double2* z; int32_t* yrow; double *y, *out;
  CK(hipMalloc(&z, npieces * 16)); CK(hipMemset(z, 0, npieces * 16));
  CK(hipMalloc(&y, 6 * P * 8 + 64)); CK(hipMemset(y, 0, 6 * P * 8 + 64)); CK(hipMalloc(&out, 8));
  std::vector<int32_t> h(nrec);
  for (int64_t l = 0; l < L; ++l) { const int p0 = (int)((l * 7919) % (P - per)); for (int a = 0; a < per; ++a) h[l * per + a] = 6 * (p0 + a); }
  CK(hipMalloc(&yrow, nrec * 4)); CK(hipMemcpy(yrow, h.data(), nrec * 4, hipMemcpyHostToDevice));
  const double bytes = npieces * 16.0;
  for (int grid : {1024, 2048, 4096, 16384}) {
    char nm[64]; snprintf(nm, 64, "coalesced, %d workgroups", grid);
    timeit(nm, bytes, [&] { hipLaunchKernelGGL(k_coalesced, dim3(grid), dim3(256), 0, 0, z, npieces, out); });
  }
  timeit("lane per feature (196 workgroups)", bytes, [&] { hipLaunchKernelGGL(k_lane_per_feature, dim3((L + 255) / 256), dim3(256), 0, 0, z, L, per, out); });
  for (int grid : {1024, 2048, 4096}) {
    char nm[64];
    snprintf(nm, 64, "64 lanes/feature, no gathers, %d wg", grid);
    timeit(nm, bytes, [&] { hipLaunchKernelGGL((k_pieces<64, false>), dim3(grid), dim3(256), 0, 0, z, yrow, y, L, per, out); });
    snprintf(nm, 64, "64 lanes/feature, gathers, %d wg", grid);
    timeit(nm, bytes, [&] { hipLaunchKernelGGL((k_pieces<64, true>), dim3(grid), dim3(256), 0, 0, z, yrow, y, L, per, out); });
    snprintf(nm, 64, "16 lanes/feature, gathers, %d wg", grid);
    timeit(nm, bytes, [&] { hipLaunchKernelGGL((k_pieces<16, true>), dim3(grid), dim3(256), 0, 0, z, yrow, y, L, per, out); });
  }
  return 0;
}
