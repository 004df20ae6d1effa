// micro-benchmark / phase timing of k_potrf (dev tool, not part of the product)
#include "../obvi-slam_amd/csrc/chol_kernels.hip"
#include <cstdio>
#include <vector>
#include <random>
using namespace obvi;
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
  for (int n : {1, 4, 16, 32}) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dr, rhs.data(), nt * T * 8, hipMemcpyHostToDevice); hipMemset(dscal, 0, 256);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_potrf, dim3(n), dim3(512), 0, 0, dS, nt, dk, dL, dr, dscal);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    printf("n=%2d  %.1f us\n", n, best * 1e3);
  }
  std::vector<double> sc(32); hipMemcpy(sc.data(), dscal, 256, hipMemcpyDeviceToHost); printf("chol_fail %g\n", sc[SC_CHOL_FAIL]);
  return 0;
}
