// fp64 MFMA probes (dev tool): v_mfma_f64_16x16x4_f64 dependent / independent issue, result-to-VALU latency
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ long long tick() { long long t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory"); return t; }
__global__ void k(long long* out, double* sink, double x0) {
  const int lane = threadIdx.x & 63;
  double a = x0 + lane * 1e-3, b = x0 * 0.5 + lane * 1e-4;
  f64x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  long long t0, t1;
  t0 = tick();
#pragma unroll
  for (int i = 0; i < 32; ++i) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
  asm volatile("" : "+v"(c0));
  t1 = tick(); if (threadIdx.x == 0) out[0] = t1 - t0;
  t0 = tick();
#pragma unroll
  for (int i = 0; i < 8; ++i) { c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0); }
  asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
  t1 = tick(); if (threadIdx.x == 0) out[1] = t1 - t0;
  // mfma -> valu -> mfma chain
  t0 = tick();
#pragma unroll
  for (int i = 0; i < 16; ++i) { c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); a = c0[0] * 0.5 + a; }
  asm volatile("" : "+v"(c0), "+v"(a));
  t1 = tick(); if (threadIdx.x == 0) out[2] = t1 - t0;
  sink[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + a;
}
int main() {
  long long* d; double* s; hipMalloc(&d, 128); hipMalloc(&s, 8 * 1024);
  for (int threads : {64, 512}) {
    for (int rep = 0; rep < 3000; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, d, s, 1.000001);
    hipDeviceSynchronize();
    long long h[16]; hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
    printf("threads %3d: dependent mfma %.1f | 4 independent accumulators %.1f | mfma + dependent fma %.1f  (cycles per mfma)\n", threads, h[0] / 32.0, h[1] / 32.0, h[2] / 16.0);
  }
  return 0;
}
