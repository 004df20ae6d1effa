// chol_kernels.hip -- K5: exact solve of the reduced camera+object system S y = b by a
// right-looking tile Cholesky (S = L L^T, 64x64 fp64 tiles) that only visits tiles which are
// structurally non-zero after symbolic fill (ba_device.h: CholPlan).  This is the role Ceres'
// sparse Cholesky of the Schur complement plays behind SPARSE_SCHUR
// (object_pose_graph_optimizer.h:665) [Ceres-doc]; the elimination order (poses in frame
// order, objects last) keeps the pose part banded and the object part a dense border.
//
// Per step k:  potrf(k)   L_kk, L_kk^-1, z_k = L_kk^-1 b_k             (1 workgroup)
//              trsm(k)    L_ik = S_ik L_kk^-T ; b_i -= L_ik z_k        (1 workgroup per tile)
//              update(k)  S_ij -= L_ik L_jk^T                          (1 workgroup per tile pair)
// then backward: y_k = L_kk^-T (z_k - sum_{i>k} L_ik^T y_i), right-looking over k descending.
#include "ba_device.h"

namespace obvi {
namespace {

constexpr int T = kTile;        // 64
constexpr int LD = T + 1;       // LDS leading dimension (bank-conflict padding)
constexpr int kThreads = 256;

__device__ __forceinline__ double* tile_ptr(double* S, int nt, int i, int j) { return S + ((int64_t)i * nt + j) * (T * T); }

__global__ void __launch_bounds__(kThreads) k_zero_tiles(double* S, int nt, const int32_t* __restrict__ tiles, int64_t m) {
  const int ti = tiles[2 * blockIdx.x], tj = tiles[2 * blockIdx.x + 1];
  double* t = tile_ptr(S, nt, ti, tj);
  for (int e = threadIdx.x; e < T * T; e += kThreads) {
    double v = 0.0;
    if (ti == tj) { const int r = e / T, c = e % T; if (r == c && (int64_t)ti * T + r >= m) v = 1.0; }  // identity on padding rows
    t[e] = v;
  }
}

__global__ void __launch_bounds__(kThreads) k_potrf(double* S, int nt, int k, double* Linv_all, double* rhs, double* scal) {
  __shared__ double A[T * LD];
  __shared__ double B[T * LD];
  __shared__ double zsh[T];
  double* tile = tile_ptr(S, nt, k, k);
  const int tid = threadIdx.x;
  for (int e = tid; e < T * T; e += kThreads) { const int r = e / T, c = e % T; A[r * LD + c] = tile[e]; }
  if (tid < T) zsh[tid] = rhs[(int64_t)k * T + tid];
  bool bad = false;
  for (int j = 0; j < T; ++j) {
    __syncthreads();
    double d = A[j * LD + j];
    if (!(d > 0.0)) { bad = true; d = 1.0; }
    const double sq = sqrt(d), inv = 1.0 / sq;
    __syncthreads();
    if (tid < T) {
      if (tid > j) A[tid * LD + j] *= inv;
      else if (tid == j) A[j * LD + j] = sq;
    }
    __syncthreads();
    // trailing update of the lower triangle: (r, c) with j < c <= r
    const int nrem = T - 1 - j;
    for (int e = tid; e < nrem * nrem; e += kThreads) {
      const int r = j + 1 + e / nrem, c = j + 1 + e % nrem;
      if (c <= r) A[r * LD + c] -= A[r * LD + j] * A[c * LD + j];
    }
  }
  __syncthreads();
  if (bad && tid == 0) unsafeAtomicAdd(scal + SC_CHOL_FAIL, 1.0);
  // B = L^-1 (lower): thread c solves L x = e_c
  for (int e = tid; e < T * LD; e += kThreads) B[e] = 0.0;
  __syncthreads();
  if (tid < T) {
    const int c = tid;
    for (int i = c; i < T; ++i) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int t = c; t < i; ++t) s -= A[i * LD + t] * B[t * LD + c];
      B[i * LD + c] = s / A[i * LD + i];
    }
  }
  __syncthreads();
  double* Li = Linv_all + (int64_t)k * (T * T);
  for (int e = tid; e < T * T; e += kThreads) {
    const int r = e / T, c = e % T;
    tile[e] = (c <= r) ? A[r * LD + c] : 0.0;
    Li[e] = B[r * LD + c];
  }
  // z_k = L^-1 b_k
  if (tid < T) {
    double s = 0.0;
    for (int c = 0; c <= tid; ++c) s += B[tid * LD + c] * zsh[c];
    rhs[(int64_t)k * T + tid] = s;
  }
}

// out[r][c] = sum_t A[r][t] * B[c][t]   (both operands staged in LDS, 4x4 outputs per thread)
__device__ __forceinline__ void tile_abt(const double* A, const double* B, double acc[4][4]) {
  const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  for (int t = 0; t < T; ++t) {
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = A[(ty * 4 + i) * LD + t]; b[i] = B[(tx * 4 + i) * LD + t]; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
  }
}

__global__ void __launch_bounds__(kThreads) k_trsm(double* S, int nt, int k, const int32_t* __restrict__ rows, const double* __restrict__ Linv_all, double* rhs) {
  __shared__ double A[T * LD];
  __shared__ double B[T * LD];
  __shared__ double zsh[T];
  const int i = rows[blockIdx.x];
  double* tile = tile_ptr(S, nt, i, k);
  const double* Li = Linv_all + (int64_t)k * (T * T);
  const int tid = threadIdx.x;
  for (int e = tid; e < T * T; e += kThreads) { const int r = e / T, c = e % T; A[r * LD + c] = tile[e]; B[r * LD + c] = Li[e]; }
  if (tid < T) zsh[tid] = rhs[(int64_t)k * T + tid];
  __syncthreads();
  double acc[4][4];
  tile_abt(A, B, acc);   // X = S_ik * Linv^T
  __syncthreads();
  const int ty = tid / 16, tx = tid % 16;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) { A[(ty * 4 + a) * LD + tx * 4 + b] = acc[a][b]; tile[(ty * 4 + a) * T + tx * 4 + b] = acc[a][b]; }
  __syncthreads();
  if (tid < T) {
    double s = 0.0;
    for (int c = 0; c < T; ++c) s += A[tid * LD + c] * zsh[c];
    rhs[(int64_t)i * T + tid] -= s;
  }
}

__global__ void __launch_bounds__(kThreads) k_update(double* S, int nt, int k, const int32_t* __restrict__ jobs) {
  __shared__ double A[T * LD];
  __shared__ double B[T * LD];
  const int i = jobs[2 * blockIdx.x], j = jobs[2 * blockIdx.x + 1];
  const double* Xi = tile_ptr(S, nt, i, k);
  const double* Xj = tile_ptr(S, nt, j, k);
  double* C = tile_ptr(S, nt, i, j);
  const int tid = threadIdx.x;
  for (int e = tid; e < T * T; e += kThreads) { const int r = e / T, c = e % T; A[r * LD + c] = Xi[e]; B[r * LD + c] = Xj[e]; }
  __syncthreads();
  double acc[4][4];
  tile_abt(A, B, acc);
  const int ty = tid / 16, tx = tid % 16;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) C[(ty * 4 + a) * T + tx * 4 + b] -= acc[a][b];
}

// backward step k: every workgroup recomputes y_k = L_kk^-T z_k (64x64 mat-vec), workgroup g < n
// applies z_j -= L_kj^T y_k for its tile j, the last workgroup stores y_k.
__global__ void __launch_bounds__(kThreads) k_backward(const double* S, int nt, int k, const int32_t* __restrict__ cols, int n, const double* __restrict__ Linv_all,
                                                      double* rhs, double* y) {
  __shared__ double ysh[T];
  __shared__ double part[4][T];
  const int tid = threadIdx.x;
  const double* Li = Linv_all + (int64_t)k * (T * T);
  {
    // y[c] = sum_{i>=c} Linv[i][c] z[i] ; 4 row-slices of 16 rows
    const int c = tid % T, q = tid / T;
    double s = 0.0;
    for (int i = q * 16; i < q * 16 + 16; ++i) if (i >= c) s += Li[i * T + c] * rhs[(int64_t)k * T + i];
    part[q][c] = s;
  }
  __syncthreads();
  if (tid < T) ysh[tid] = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
  __syncthreads();
  if ((int)blockIdx.x == n) {
    if (tid < T) y[(int64_t)k * T + tid] = ysh[tid];
    return;
  }
  const int j = cols[blockIdx.x];
  const double* X = tile_ptr(const_cast<double*>(S), nt, k, j);
  {
    const int c = tid % T, q = tid / T;
    double s = 0.0;
    for (int r = q * 16; r < q * 16 + 16; ++r) s += X[r * T + c] * ysh[r];
    __syncthreads();
    part[q][c] = s;
  }
  __syncthreads();
  if (tid < T) rhs[(int64_t)j * T + tid] -= part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
}

}  // namespace

void launch_zero_tiles(hipStream_t s, double* S, int32_t nt, const int32_t* tile_list, int32_t ntiles, int64_t m) {
  if (ntiles > 0) hipLaunchKernelGGL(k_zero_tiles, dim3(ntiles), dim3(kThreads), 0, s, S, nt, tile_list, m);
}

void launch_cholesky_solve(hipStream_t s, const CholPlan& plan, double* S, double* Linv, double* rhs, double* y, double* scal) {
  const int nt = plan.nt;
  for (int k = 0; k < nt; ++k) {
    hipLaunchKernelGGL(k_potrf, dim3(1), dim3(kThreads), 0, s, S, nt, k, Linv, rhs, scal);
    const int ntr = plan.trsm_ptr[k + 1] - plan.trsm_ptr[k];
    if (ntr > 0) hipLaunchKernelGGL(k_trsm, dim3(ntr), dim3(kThreads), 0, s, S, nt, k, plan.trsm_i + plan.trsm_ptr[k], Linv, rhs);
    const int nup = plan.upd_ptr[k + 1] - plan.upd_ptr[k];
    if (nup > 0) hipLaunchKernelGGL(k_update, dim3(nup), dim3(kThreads), 0, s, S, nt, k, plan.upd_ij + 2 * (int64_t)plan.upd_ptr[k]);
  }
  for (int k = nt - 1; k >= 0; --k) {
    const int nb = plan.back_ptr[k + 1] - plan.back_ptr[k];
    hipLaunchKernelGGL(k_backward, dim3(nb + 1), dim3(kThreads), 0, s, S, nt, k, plan.back_j + plan.back_ptr[k], nb, Linv, rhs, y);
  }
}

}  // namespace obvi
