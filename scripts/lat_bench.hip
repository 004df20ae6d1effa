// instruction latency probes for gfx950 (dev tool): single wavefront, clock64 around unrolled sequences
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ long long tick(double& dep) { long long t; asm volatile("s_nop 0" : "+v"(dep)); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory"); asm volatile("s_nop 0" : "+v"(dep)); return t; }
#define TICK() tick2(x, a0, a1, a2, a3, a4, a5, a6, a7)
__device__ __forceinline__ void pin(double& x, double& a0, double& a1, double& a2, double& a3, double& a4, double& a5, double& a6, double& a7) { asm volatile("s_nop 0" : "+v"(x), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
__device__ __forceinline__ long long tick2(double& x, double& a0, double& a1, double& a2, double& a3, double& a4, double& a5, double& a6, double& a7) { pin(x, a0, a1, a2, a3, a4, a5, a6, a7); const long long t = tick(x); pin(x, a0, a1, a2, a3, a4, a5, a6, a7); return t; }
__global__ void k(long long* out, double* sink, double x0) {
  __shared__ double sh[512];
  const int lane = threadIdx.x;
  double x = x0 + lane * 1e-9, y = x0 * 0.5, a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
  long long t0, t1;
  // (a) dependent fp64 fma chain, 64 ops
  t0 = TICK();
#pragma unroll
  for (int i = 0; i < 64; ++i) x = fma(x, y, 1.0);
  t1 = TICK(); if (lane == 0) out[0] = t1 - t0;
  // (b) 8 independent chains x 16 = 128 ops
  t0 = TICK();
#pragma unroll
  for (int i = 0; i < 16; ++i) { a0 = fma(a0, y, 1.0); a1 = fma(a1, y, 1.0); a2 = fma(a2, y, 1.0); a3 = fma(a3, y, 1.0); a4 = fma(a4, y, 1.0); a5 = fma(a5, y, 1.0); a6 = fma(a6, y, 1.0); a7 = fma(a7, y, 1.0); }
  t1 = TICK(); if (lane == 0) out[1] = t1 - t0;
  x += a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  // (c) dependent rsq chain 16
  t0 = TICK();
#pragma unroll
  for (int i = 0; i < 16; ++i) x = __builtin_amdgcn_rsq(x) + 1.0;
  t1 = TICK(); if (lane == 0) out[2] = t1 - t0;   // rsq + add per iteration
  // (d) readlane -> use chain 16
  t0 = TICK();
#pragma unroll
  for (int i = 0; i < 16; ++i) { const int lo = __builtin_amdgcn_readlane(__double2loint(x), 5), hi = __builtin_amdgcn_readlane(__double2hiint(x), 5); x = fma(__hiloint2double(hi, lo), 0.5, x); }
  t1 = TICK(); if (lane == 0) out[3] = t1 - t0;
  // (e) LDS write -> read (other lane) round trip chain 16
  t0 = TICK();
#pragma unroll
  for (int i = 0; i < 16; ++i) { sh[lane] = x; __builtin_amdgcn_wave_barrier(); x = sh[(lane + 1) & 63] * 0.5 + 1.0; __builtin_amdgcn_wave_barrier(); }
  t1 = TICK(); if (lane == 0) out[4] = t1 - t0;
  // (f) __syncthreads x16 (all waves of the block)
  t0 = TICK();
#pragma unroll
  for (int i = 0; i < 16; ++i) __syncthreads();
  t1 = TICK(); if (lane == 0) out[5] = t1 - t0;
  // (g) ds_bpermute chain 16
  t0 = TICK();
#pragma unroll
  for (int i = 0; i < 16; ++i) { x = __shfl(x, (lane + 3) & 63, 64) + 1.0; }
  t1 = TICK(); if (lane == 0) out[6] = t1 - t0;
  // (h) dependent fp64 fma chain with exec = 1 lane
  if (lane == 7) {
    t0 = TICK();
#pragma unroll
    for (int i = 0; i < 64; ++i) x = fma(x, y, 1.0);
    t1 = TICK(); out[7] = t1 - t0;
  }
  // (i) 4 independent chains x 16
  t0 = TICK();
#pragma unroll
  for (int i = 0; i < 16; ++i) { a0 = fma(a0, y, 1.0); a1 = fma(a1, y, 1.0); a2 = fma(a2, y, 1.0); a3 = fma(a3, y, 1.0); }
  t1 = TICK(); if (lane == 0) out[8] = t1 - t0;
  // (j) 2 independent chains x 32
  t0 = TICK();
#pragma unroll
  for (int i = 0; i < 32; ++i) { a4 = fma(a4, y, 1.0); a5 = fma(a5, y, 1.0); }
  t1 = TICK(); if (lane == 0) out[9] = t1 - t0;
  // (k) dependent mul chain
  t0 = TICK();
#pragma unroll
  for (int i = 0; i < 64; ++i) a6 = a6 * y;
  t1 = TICK(); if (lane == 0) out[10] = t1 - t0;
  sink[threadIdx.x] = x + a0 + a1 + a2 + a3 + a4 + a5 + a6;
}
int main() {
  long long* d; double* s; hipMalloc(&d, 128); hipMalloc(&s, 8 * 512);
  for (int threads : {64, 512}) {
    hipMemset(d, 0, 128);
    for (int rep = 0; rep < 3000; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, d, s, 1.000001);
    hipDeviceSynchronize();
    long long h[16]; hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
    printf("threads %d: dep fma %.1f | 8-way indep fma %.1f | rsq+add %.1f | readlane x2+fma %.1f | lds wr->rd %.1f | syncthreads %.1f | shfl+add %.1f | dep fma 1 lane %.1f | 4-way %.1f | 2-way %.1f | dep mul %.1f  (cycles per op)\n", threads,
           h[0] / 64.0, h[1] / 128.0, h[2] / 16.0, h[3] / 16.0, h[4] / 16.0, h[5] / 16.0, h[6] / 16.0, h[7] / 64.0, h[8] / 64.0, h[9] / 64.0, h[10] / 64.0);
  }
  return 0;
}
