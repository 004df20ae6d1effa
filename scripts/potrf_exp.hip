// experimental potrf variants against k_potrf (dev tool, not part of the product):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -w scripts/potrf_exp.hip -o scripts/potrf_exp
// clocks of one chosen thread of workgroup 0: time attributed to the code section that ends at marker i
__device__ long long g_acc[8];
__device__ int g_probe;
#define OBVI_PH8_DECL long long acc8_[6] = {0, 0, 0, 0, 0, 0}, last8_ = 0
#define OBVI_PH8(i) do { const long long now_ = clock64(); acc8_[i] += now_ - last8_; last8_ = now_; } while (0)
#define OBVI_PH8_END do { if ((int)threadIdx.x == g_probe && blockIdx.x == 0) for (int q_ = 1; q_ < 6; ++q_) g_acc[q_] = acc8_[q_]; } while (0)
#include "../obvi-slam_amd/csrc/chol_kernels.hip"
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>
using namespace obvi;

namespace obvi { namespace {
__global__ void __launch_bounds__(512) k_potrf2(double* S, int nt, const int32_t* __restrict__ klist, double* Linv_all, double* rhs, double* scal) {
  __shared__ double smem[kPotrf2Lds];
  potrf2_tile(smem, S, nt, klist[blockIdx.x], Linv_all, rhs, scal);
}
} }

int main() {
  const int nt = 32, T = 64;
  std::vector<double> S((size_t)nt * nt * T * T, 0.0), rhs(nt * T, 1.0);
  std::mt19937_64 rng(1);
  std::normal_distribution<double> nd;
  for (int k = 0; k < nt; ++k) {
    std::vector<double> A(T * T);
    for (auto& v : A) v = nd(rng);
    double* t = &S[((size_t)k * nt + k) * T * T];
    for (int i = 0; i < T; ++i) for (int j = 0; j < T; ++j) { double s = 0; for (int q = 0; q < T; ++q) s += A[i * T + q] * A[j * T + q]; t[i * T + j] = s + (i == j ? T : 0); }
  }
  double *dS, *dL, *dr, *dscal; int32_t* dk;
  hipMalloc(&dS, S.size() * 8); hipMalloc(&dL, (size_t)nt * T * T * 8); hipMalloc(&dr, nt * T * 8); hipMalloc(&dscal, 256); hipMalloc(&dk, nt * 4);
  std::vector<int32_t> kl(nt); for (int i = 0; i < nt; ++i) kl[i] = i;
  hipMemcpy(dk, kl.data(), nt * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int variant = 0; variant < 3; ++variant) {
    for (int n : {1, 32}) {
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dr, rhs.data(), nt * T * 8, hipMemcpyHostToDevice); hipMemset(dscal, 0, 256);
        hipEventRecord(e0, 0);
        if (variant == 0) hipLaunchKernelGGL(k_potrf, dim3(n), dim3(512), 0, 0, dS, nt, dk, dL, dr, dscal);
        else if (variant == 2) hipLaunchKernelGGL(k_potrf2, dim3(n), dim3(512), 0, 0, dS, nt, dk, dL, dr, dscal);
        else continue;
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
      }
      printf("variant %d  n=%2d  %.1f us\n", variant, n, best * 1e3);
    }
    std::vector<double> Lh(S.size()), Li((size_t)nt * T * T), z(nt * T);
    hipMemcpy(Lh.data(), dS, S.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(Li.data(), dL, Li.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(z.data(), dr, z.size() * 8, hipMemcpyDeviceToHost);
    double e_llt = 0, e_inv = 0, e_z = 0, e_up = 0;
    for (int k = 0; k < nt; ++k) {
      const double* A = &S[((size_t)k * nt + k) * T * T]; const double* L = &Lh[((size_t)k * nt + k) * T * T]; const double* W = &Li[(size_t)k * T * T];
      for (int i = 0; i < T; ++i) for (int j = 0; j < T; ++j) {
        double s = 0, t = 0; for (int q = 0; q < T; ++q) { s += L[i * T + q] * L[j * T + q]; t += L[i * T + q] * W[q * T + j]; }
        e_llt = std::max(e_llt, std::fabs(s - A[i * T + j]) / T); e_inv = std::max(e_inv, std::fabs(t - (i == j)));
        if (j > i) e_up = std::max(e_up, std::max(std::fabs(L[i * T + j]), std::fabs(W[i * T + j])));
      }
      for (int i = 0; i < T; ++i) { double s = 0; for (int q = 0; q < T; ++q) s += L[i * T + q] * z[k * T + q]; e_z = std::max(e_z, std::fabs(s - 1.0)); }
    }
    std::vector<double> sc(32); hipMemcpy(sc.data(), dscal, 256, hipMemcpyDeviceToHost);
    printf("variant %d  chol_fail %g   max |L L^T - A|/64 %.2e   max |L W - I| %.2e   max |L z - b| %.2e   upper %.1e\n", variant, sc[SC_CHOL_FAIL], e_llt, e_inv, e_z, e_up);
  }
  for (int probe : {0, 255, 256, 511, 34}) {
    long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_acc), z, sizeof(z)); hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(int));
    hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_potrf2, dim3(1), dim3(512), 0, 0, dS, nt, dk, dL, dr, dscal);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(z, HIP_SYMBOL(g_acc), sizeof(z));
    printf("thread %3d: work %lld  barrier wait %lld  (cycles, sum over the steps)\n", probe, z[1], z[2]);
  }
  return 0;
}
