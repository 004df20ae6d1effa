#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long n, long long* out) { const long long t0 = clock64(); long long t; do { t = clock64(); } while (t - t0 < n); out[0] = t - t0; out[1] = wall_clock64(); }
__global__ void fmas(double* s, double y, long long* out) { double x = 1.0 + threadIdx.x; const long long t0 = clock64(); const long long w0 = wall_clock64();
  for (int i = 0; i < 100000; ++i) { x = fma(x, y, 1.0); x = fma(x, y, 1.0); x = fma(x, y, 1.0); x = fma(x, y, 1.0); x = fma(x, y, 1.0); x = fma(x, y, 1.0); x = fma(x, y, 1.0); x = fma(x, y, 1.0); x = fma(x, y, 1.0); x = fma(x, y, 1.0); }
  s[threadIdx.x] = x; out[0] = clock64() - t0; out[1] = wall_clock64() - w0; }
int main() {
  long long* d; double* s; hipMalloc(&d, 64); hipMalloc(&s, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0); hipLaunchKernelGGL(spin, dim3(1), dim3(1), 0, 0, 20000000LL, d); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("spin: %lld ticks in %.3f ms -> %.1f MHz\n", h[0], ms, h[0] / ms / 1e3);
    hipEventRecord(e0, 0); hipLaunchKernelGGL(fmas, dim3(1), dim3(64), 0, 0, s, 0.5, d); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("1M dependent fp64 fma: %lld ticks, %lld wall ticks (100 MHz), %.3f ms -> %.2f ns per fma, %.2f ticks per fma\n", h[0], h[1], ms, ms * 1e6 / 1e6, h[0] / 1e6);
  }
  return 0;
}
