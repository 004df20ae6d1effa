// dev tool: does the SPAN of the tile grid matter?  Jobs that read two 32 KB tiles and read-modify-write a third, tiles taken from a compact
// pool vs scattered over a grid as large as config #3's (nt = 241: 1.9 GB).  hipcc --offload-arch=gfx950 -O3 scripts/tlb_probe.hip -o scripts/tlb_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
__global__ void __launch_bounds__(256) k_job(double* S, const long long* __restrict__ tiles, int njobs) {
  __shared__ double sh[2 * 4096 + 64];
  const int g = blockIdx.x;
  const double2* A = reinterpret_cast<const double2*>(S + tiles[3 * g] * 4096);
  const double2* B = reinterpret_cast<const double2*>(S + tiles[3 * g + 1] * 4096);
  double2* C = reinterpret_cast<double2*>(S + tiles[3 * g + 2] * 4096);
  for (int e = threadIdx.x; e < 2048; e += 256) { const double2 a = A[e], b = B[e]; sh[2 * e] = a.x; sh[2 * e + 1] = a.y; sh[4096 + 2 * e] = b.x; sh[4096 + 2 * e + 1] = b.y; }
  __syncthreads();
  double s = 0.0;
  for (int k = 0; k < 64; ++k) s += sh[(threadIdx.x * 17 + k * 64) & 4095] * sh[4096 + ((threadIdx.x * 29 + k * 64) & 4095)];
  for (int e = threadIdx.x; e < 2048; e += 256) { double2 c = C[e]; c.x -= s; c.y -= s; C[e] = c; }
}
int main() {
  const long long grid_tiles = 241LL * 241LL;   // 58 081 tiles = 1.9 GB
  double* S; hipMalloc(&S, grid_tiles * 4096 * 8); hipMemset(S, 0, grid_tiles * 4096 * 8);
  std::mt19937_64 rng(3);
  const int njobs = 1755, nused = 6000;
  for (int mode = 0; mode < 3; ++mode) {
    // mode 0: the used tiles are the first 6 000 of the buffer; 1: spread evenly over the grid; 2: random positions in the grid
    std::vector<long long> pool(nused);
    for (int i = 0; i < nused; ++i) pool[i] = mode == 0 ? i : mode == 1 ? (long long)i * (grid_tiles / nused) : (long long)(rng() % grid_tiles);
    std::vector<long long> t(3 * njobs);
    std::vector<int> perm(nused); for (int i = 0; i < nused; ++i) perm[i] = i; std::shuffle(perm.begin(), perm.end(), rng);
    for (int g = 0; g < njobs; ++g) { t[3 * g] = pool[rng() % 400]; t[3 * g + 1] = pool[rng() % 400]; t[3 * g + 2] = pool[perm[400 + g]]; }   // operands from 400 shared tiles, targets distinct
    long long* dt; hipMalloc(&dt, t.size() * 8); hipMemcpy(dt, t.data(), t.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(a, 0); hipLaunchKernelGGL(k_job, dim3(njobs), dim3(256), 0, 0, S, dt, njobs); hipEventRecord(b, 0); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); best = std::min(best, ms);
    }
    printf("mode %d (%s): %d jobs  %.1f us\n", mode, mode == 0 ? "compact pool" : mode == 1 ? "spread over 1.9 GB" : "random in 1.9 GB", njobs, best * 1e3);
    hipFree(dt);
  }
  return 0;
}
