// launch-gap probe (dev tool): a chain of N small dependent kernels, stream launches vs one hipGraph launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k(double* x, int spin) { double v = x[threadIdx.x]; for (int i = 0; i < spin; ++i) v = v * 1.0000001 + 1e-9; x[threadIdx.x] = v; }
int main() {
  double* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int N = 120;
  for (int spin : {10, 2000}) {
    for (int wgs : {1, 512}) {
      auto run_stream = [&]() { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, s, d, spin); };
      run_stream(); hipStreamSynchronize(s);
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < 20; ++r) { run_stream(); hipStreamSynchronize(s); }
      double us_stream = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20;
      hipGraph_t g; hipGraphExec_t ge;
      hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal); run_stream(); hipStreamEndCapture(s, &g);
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      hipGraphLaunch(ge, s); hipStreamSynchronize(s);
      t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < 20; ++r) { hipGraphLaunch(ge, s); hipStreamSynchronize(s); }
      double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20;
      hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, s, d, spin * N); hipStreamSynchronize(s);
      t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < 20; ++r) { hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, s, d, spin * N); hipStreamSynchronize(s); }
      double us_one = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20;
      printf("one kernel with the same work %.1f us -> gap per kernel: stream %.2f us, graph %.2f us | ", us_one, (us_stream - us_one) / N, (us_graph - us_one) / N);
      printf("spin %4d wgs %3d: %d kernels  stream %.1f us (%.2f per kernel)   graph %.1f us (%.2f per kernel)\n", spin, wgs, N, us_stream, us_stream / N, us_graph, us_graph / N);
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
  }
  return 0;
}
