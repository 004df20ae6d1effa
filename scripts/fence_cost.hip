// cost of agent-scope release / acquire fences around tile-sized writes (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void __launch_bounds__(256) k(double* tiles, int* flags, int mode, int nprod) {
  // producers: blocks [0, nprod): write a 32 KB tile, (mode&1) fence + flag.  consumers: blocks [nprod, ...): (mode&2) wait for the flag of tile (b % nprod) + acquire fence; read the tile
  const int b = blockIdx.x;
  if (b < nprod) {
    double* t = tiles + (size_t)b * 4096;
    for (int e = threadIdx.x; e < 4096; e += 256) t[e] = t[e] * 0.5 + 1.0;
    if (mode & 1) { __threadfence(); __syncthreads(); if (threadIdx.x == 0) atomicAdd(flags + b, 1); }
  } else {
    const int src = b % nprod;
    if (mode & 2) {
      if (threadIdx.x == 0) { int spins = 0; while (__hip_atomic_load(flags + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 1 && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(4); }
      __syncthreads();
      __threadfence();
    }
    const double* t = tiles + (size_t)src * 4096;
    double s = 0;
    for (int e = threadIdx.x; e < 4096; e += 256) s += t[e];
    if (s == 123.456) tiles[0] = s;
  }
}
int main() {
  const int maxp = 2048;
  double* d; int* f; hipMalloc(&d, (size_t)maxp * 4096 * 8); hipMalloc(&f, maxp * 4); hipMemset(d, 0, (size_t)maxp * 4096 * 8);
  for (int nprod : {32, 256, 2048}) for (int ncons : {0, 4}) for (int mode : {0, 1, 3}) {
    if (ncons == 0 && mode == 3) continue;
    float best = 1e9;
    for (int rep = 0; rep < 20; ++rep) {
      hipMemset(f, 0, maxp * 4);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k, dim3(nprod * (1 + ncons)), dim3(256), 0, 0, d, f, mode, nprod);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
      hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("producers %4d consumers/tile %d mode %d (%s): %.1f us\n", nprod, ncons, mode, mode == 0 ? "no fences" : mode == 1 ? "release only" : "release + wait + acquire", best * 1e3);
  }
  return 0;
}
