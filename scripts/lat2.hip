// fp64 issue-rate probes, part 2 (dev tool): operand patterns of v_fma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ long long tick(double& dep) { long long t; asm volatile("s_nop 0" : "+v"(dep)); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory"); asm volatile("s_nop 0" : "+v"(dep)); return t; }
#define REP16(X) X X X X X X X X X X X X X X X X
__global__ void k(long long* out, double* sink, double x0) {
  const int lane = threadIdx.x;
  double x = x0 + lane * 1e-9, y = x0 * 0.5, z = x0 * 0.25, a = x + 1, b = x + 2, c = x + 3, d = x + 4;
  long long t0, t1;
  // (0) x = fma(x, y, 1.0): 2 vgpr sources
  t0 = tick(x); REP16(asm volatile("v_fma_f64 %0, %0, %1, 1.0" : "+v"(x) : "v"(y));) REP16(asm volatile("v_fma_f64 %0, %0, %1, 1.0" : "+v"(x) : "v"(y));) t1 = tick(x); if (lane == 0) out[0] = t1 - t0;
  // (1) x = fma(x, y, z): 3 vgpr sources
  t0 = tick(x); REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));) REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));) t1 = tick(x); if (lane == 0) out[1] = t1 - t0;
  // (2) independent: a = fma(y, z, a) ... 4 accumulators, 3 vgpr sources
  t0 = tick(x); REP16(asm volatile("v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(y), "v"(z));) t1 = tick(x); if (lane == 0) out[2] = t1 - t0;
  // (3) v_fmac_f64 (VOP2): a += y * z, 4 accumulators
  t0 = tick(x); REP16(asm volatile("v_fmac_f64 %0, %4, %5\n\tv_fmac_f64 %1, %4, %5\n\tv_fmac_f64 %2, %4, %5\n\tv_fmac_f64 %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(y), "v"(z));) t1 = tick(x); if (lane == 0) out[3] = t1 - t0;
  // (4) v_mul_f64 independent 4
  t0 = tick(x); REP16(asm volatile("v_mul_f64 %0, %4, %5\n\tv_mul_f64 %1, %4, %5\n\tv_mul_f64 %2, %4, %5\n\tv_mul_f64 %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(y), "v"(z));) t1 = tick(x); if (lane == 0) out[4] = t1 - t0;
  // (5) v_add_f64 independent 4
  t0 = tick(x); REP16(asm volatile("v_add_f64 %0, %4, %0\n\tv_add_f64 %1, %4, %1\n\tv_add_f64 %2, %4, %2\n\tv_add_f64 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(y), "v"(z));) t1 = tick(x); if (lane == 0) out[5] = t1 - t0;
  // (6) fma with negated source (neg modifier) 3 vgpr
  t0 = tick(x); REP16(asm volatile("v_fma_f64 %0, -%4, %5, %0\n\tv_fma_f64 %1, -%4, %5, %1\n\tv_fma_f64 %2, -%4, %5, %2\n\tv_fma_f64 %3, -%4, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(y), "v"(z));) t1 = tick(x); if (lane == 0) out[6] = t1 - t0;
  // (7) v_mov_b64 independent
  t0 = tick(x); REP16(asm volatile("v_mov_b64 %0, %4\n\tv_mov_b64 %1, %5\n\tv_mov_b64 %2, %4\n\tv_mov_b64 %3, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(y), "v"(z));) t1 = tick(x); if (lane == 0) out[7] = t1 - t0;
  // (8) fp32 fma independent 4 (reference)
  float fa = (float)x, fb = fa + 1, fc = fa + 2, fd = fa + 3, fy = (float)y, fz = (float)z;
  t0 = tick(x); REP16(asm volatile("v_fma_f32 %0, %4, %5, %0\n\tv_fma_f32 %1, %4, %5, %1\n\tv_fma_f32 %2, %4, %5, %2\n\tv_fma_f32 %3, %4, %5, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(fy), "v"(fz));) t1 = tick(x); if (lane == 0) out[8] = t1 - t0;
  sink[threadIdx.x] = x + a + b + c + d + fa + fb + fc + fd;
}
int main() {
  long long* d; double* s; hipMalloc(&d, 128); hipMalloc(&s, 8 * 1024);
  for (int threads : {64, 128, 512}) {
    for (int rep = 0; rep < 3000; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, d, s, 1.000001);
    hipDeviceSynchronize();
    long long h[16]; hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
    printf("threads %3d: dep fma 2src %.1f | dep fma 3src %.1f | indep fma 3src %.1f | fmac %.1f | mul %.1f | add %.1f | fma neg %.1f | mov_b64 %.1f | fma_f32 %.1f  (cycles per instruction)\n", threads,
           h[0] / 32.0, h[1] / 32.0, h[2] / 64.0, h[3] / 64.0, h[4] / 64.0, h[5] / 64.0, h[6] / 64.0, h[7] / 64.0, h[8] / 64.0);
  }
  return 0;
}
