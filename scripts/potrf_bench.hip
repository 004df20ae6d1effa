// micro-benchmark + self-check of k_potrf (dev tool, not part of the product):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics scripts/potrf_bench.hip -o scripts/potrf_bench
// phase clocks of one probe thread of workgroup 0, kept in registers: per-phase sums over the 16 panel steps
__device__ long long g_ph[4], g_mark[8], g_step[32];
__device__ int g_probe;
enum { idx_ph1 = 0, idx_ph2 = 1, idx_ph3 = 2 };
#ifndef NOPROBE
#define OBVI_TICK(i) OBVI_TICK_##i
#define OBVI_TICK_0 long long acc_[3] = {0, 0, 0}, last_ = 0; int si_ = 0
#define OBVI_TICK_1 last_ = clock64()
#define OBVI_TICK_2 do { if ((int)threadIdx.x == g_probe && blockIdx.x == 0) { g_ph[0] = acc_[0]; g_ph[1] = acc_[1]; g_ph[2] = acc_[2]; } } while (0)
#define OBVI_TICK_3
#define OBVI_TICK_4
#define OBVI_MARK(i) do { if ((int)threadIdx.x == g_probe && blockIdx.x == 0) g_mark[i] = clock64(); } while (0)
#define OBVI_PH(var) do { const long long now_ = clock64(); acc_[idx_##var] += now_ - last_; if ((int)threadIdx.x == g_probe && blockIdx.x == 0) g_step[si_] = now_ - last_; ++si_; last_ = now_; } while (0)
#endif
#include "../obvi-slam_amd/csrc/chol_kernels.hip"
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>
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
  std::vector<double> Lh(S.size()), Li((size_t)nt * T * T), z(nt * T);
  hipMemcpy(Lh.data(), dS, S.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(Li.data(), dL, Li.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(z.data(), dr, z.size() * 8, hipMemcpyDeviceToHost);
  double e_llt = 0, e_inv = 0, e_z = 0;
  for (int k = 0; k < nt; ++k) {
    const double* A = &S[((size_t)k * nt + k) * T * T]; const double* L = &Lh[((size_t)k * nt + k) * T * T]; const double* W = &Li[(size_t)k * T * T];
    for (int i = 0; i < T; ++i) for (int j = 0; j < T; ++j) {
      double s = 0, t = 0; for (int q = 0; q < T; ++q) { s += L[i * T + q] * L[j * T + q]; t += L[i * T + q] * W[q * T + j]; }
      e_llt = std::max(e_llt, std::fabs(s - A[i * T + j]) / T); e_inv = std::max(e_inv, std::fabs(t - (i == j)));
    }
    for (int i = 0; i < T; ++i) { double s = 0; for (int q = 0; q < T; ++q) s += L[i * T + q] * z[k * T + q]; e_z = std::max(e_z, std::fabs(s - 1.0)); }
  }
  for (int probe : {0, 64, 128, 255, 256, 320, 384, 511}) {
    long long ph[4];
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(int));
    hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_potrf, dim3(1), dim3(512), 0, 0, dS, nt, dk, dL, dr, dscal);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_ph), sizeof(ph));
    { long long mk[8]; hipMemcpyFromSymbol(mk, HIP_SYMBOL(g_mark), sizeof(mk)); printf("            start->loop %lld   loop %lld   loop end->L stored %lld   ->W stored %lld   ->end %lld\n", mk[1] - mk[0], mk[2] - mk[1], mk[3] - mk[2], mk[4] - mk[3], mk[5] - mk[4]); }
    { long long st[32]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_step), sizeof(st)); printf("            per step work/wait:"); for (int i = 0; i < 32; i += 2) printf(" %lld/%lld", st[i], st[i + 1]); printf("\n"); }
    printf("thread %3d: phase 1 %lld   phase 2 %lld   phase 3 %lld   (cycles, sum over 16 steps: panel + barrier / trailing / publish + diagonal block + barrier)\n", probe, ph[0], ph[1], ph[2]);
  }
  std::vector<double> sc(32); hipMemcpy(sc.data(), dscal, 256, hipMemcpyDeviceToHost);
  {   // a tile that is not positive definite (one pivot <= 0 at column 21, nothing non-finite on the way in) and one with a NaN: both must raise the flag
    for (int which = 0; which < 2; ++which) {
      std::vector<double> Sb(S.begin(), S.begin() + (size_t)T * T);
      if (which == 0) Sb[21 * T + 21] = -1.0; else Sb[40 * T + 3] = Sb[3 * T + 40] = std::nan("");
      hipMemcpy(dS, Sb.data(), Sb.size() * 8, hipMemcpyHostToDevice); hipMemset(dscal, 0, 256);
      hipLaunchKernelGGL(k_potrf, dim3(1), dim3(512), 0, 0, dS, nt, dk, dL, dr, dscal);
      std::vector<double> sb(32); hipMemcpy(sb.data(), dscal, 256, hipMemcpyDeviceToHost);
      printf("%s tile: chol_fail %g (expected > 0)\n", which == 0 ? "indefinite" : "NaN", sb[SC_CHOL_FAIL]);
    }
  }
  printf("chol_fail %g   max |L L^T - A|/64 %.2e   max |L W - I| %.2e   max |L z - b| %.2e\n", sc[SC_CHOL_FAIL], e_llt, e_inv, e_z);
  return 0;
}
