// chol_kernels.hip -- K5: exact solve of the reduced camera+object system S y = b by a
// level-scheduled tile Cholesky (S = L L^T, 64x64 fp64 tiles).  Only tiles that are structurally
// non-zero after symbolic fill are visited, and tile columns that do not depend on each other
// (same level of the tile elimination tree) are processed by the same launch.  This is the role
// Ceres' sparse Cholesky of the Schur complement plays behind SPARSE_SCHUR
// (object_pose_graph_optimizer.h:665) [Ceres-doc]; because the factorisation is exact the
// elimination order (plan.cpp: nested dissection of the pose chain, objects inside the tree) changes the
// result only by round-off.
//
// Per level, two launches:
//   k_trsm          L_ik = S_ik L_kk^-T                                          (tile product with L_kk^-1, matrix cores)
//   k_update_potrf  S_ij -= sum_k L_ik L_jk^T ; b_i -= sum_k L_ik z_k            (1 workgroup per target tile / row block)
//                   and, in the same grid, for the columns of the NEXT level:  L_kk, L_kk^-1, z_k = L_kk^-1 b_k
//                   (potrf_mfma_tile: 4-column panels on the matrix cores, one workgroup per column)
// then backward over levels descending, row oriented: y_k = L_kk^-T t_k, t_j -= L_kj^T y_k (k_backward, 1 workgroup per tile).
#include <algorithm>
#include <cstdlib>
#include "ba_device.h"

namespace obvi {
namespace {

constexpr int T = kTile;        // 64
constexpr int LD = T + 1;       // LDS leading dimension (bank-conflict padding)
constexpr int kThreads = 256;

// 1/sqrt(p): hardware v_rsq_f64 seed + one Newton step in its cubic (Halley-type) form: e = 1 - p y^2,
// y <- y (1 + e/2 + 3 e^2/8): relative error ~ (5/16) e^3, i.e. full fp64 accuracy from the >= 20-bit seed with one
// dependent chain of 5 operations instead of the 7 of two quadratic steps (the 4x4 pivot chain is the critical path of potrf)
__device__ __forceinline__ double fast_rsqrt(double p) {
  const double y = __builtin_amdgcn_rsq(p);
  const double e = fma(-p * y, y, 1.0);
  const double c = fma(e, 0.375, 0.5);   // 1/2 + 3 e / 8
  return fma(y * e, c, y);
}
// sum over the 16 lanes of a DPP row (lanes 16 q .. 16 q + 15), in the row's last lane: four row shifts on the vector ALU instead of four
// ds_bpermute round trips
template <int CTRL>
__device__ __forceinline__ double dpp_row_shr(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum_to_last(double v) {
  v += dpp_row_shr<0x111>(v);
  v += dpp_row_shr<0x112>(v);
  v += dpp_row_shr<0x114>(v);
  v += dpp_row_shr<0x118>(v);
  return v;
}
// sum over the four lanes of a quad, in all four (DPP quad permutes)
__device__ __forceinline__ double quad_sum(double v) {
  v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xf, 0xf, true));   // quad_perm:[1,0,3,2]
  v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x4E, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x4E, 0xf, 0xf, true));   // quad_perm:[2,3,0,1]
  return v;
}
__device__ __forceinline__ double* tile_ptr(double* S, int nt, int i, int j) { return S + ((int64_t)i * nt + j) * (T * T); }

// Start of an LM step, one launch: the first workgroups clear the small accumulators of the step (diagonal blocks, gradient, right-hand
// side, potrf counters, scalar block), the others zero the structurally non-zero tiles (identity on padding rows).
__global__ void __launch_bounds__(kThreads) k_zero_tiles(double* S, int nt, const int32_t* __restrict__ tiles, int ntiles, const uint8_t* __restrict__ is_pad_row, StepClear c, int extra) {
  if ((int)blockIdx.x < extra) {
    if (c.pub_host != nullptr && blockIdx.x == 0) {
      // the scalar block of the step that just ended goes to the host before it is cleared (the threads below clear the entries they
      // published): one launch less between the trial cost and the host's accept / reject decision.  Workgroup 0: the first to start.
      volatile double* host = c.pub_host;
      if ((int64_t)threadIdx.x < c.n_scal) host[threadIdx.x] = c.scal[threadIdx.x];
      __threadfence_system();
      __syncthreads();
      if (threadIdx.x == 0) { __threadfence_system(); host[c.n_scal] = c.pub_seq; }
    }
    const int64_t stride = (int64_t)extra * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < c.n_max; i += stride) {
      if (i < c.n_hdiag) c.hdiag[i] = 0.0;
      if (i < c.n_g) c.g[i] = 0.0;
      if (i < c.n_rhs) c.rhs[i] = 0.0;
      if (i < c.n_done) c.diag_done[i] = 0;
      if (i < c.n_scal) c.scal[i] = i == c.fixed_slot ? c.fixed_cost : 0.0;
    }
    return;
  }
  const int tb = (int)blockIdx.x - extra;
  const int ti = tiles[2 * tb], tj = tiles[2 * tb + 1];
  double* t = tile_ptr(S, nt, ti, tj);
  for (int e = threadIdx.x; e < T * T; e += kThreads) {
    double v = 0.0;
    if (ti == tj) {
      const int r = e / T, c = e % T;
      if (r == c && is_pad_row[(int64_t)ti * T + r]) v = 1.0;
    }
    t[e] = v;
  }
}

#ifndef OBVI_TICK
#define OBVI_TICK(i)
#define OBVI_PH(var)
#endif
#ifndef OBVI_MARK
#define OBVI_MARK(i)
#endif

// ---------------------------------------------------------------------------------------
// 64x64x64 fp64 tile product  C += A * B^T  on the matrix cores: v_mfma_f64_16x16x4_f64.
// Both operands are row-major tiles staged in LDS with leading dimension LDM = 66 doubles
// (bank = (4 row + 2 col) mod 64 for the 16-row x 2-col footprint of a 32-lane group: conflict-free).
// Wavefront w owns output columns [16w, 16w+16); acc[rt] is the 16x16 tile of rows [16rt, 16rt+16).
// Fragment layout (cdna_hip_programming.md 3): A: lane l -> A[l&15][l>>4]; B: lane l -> B[l>>4][l&15];
// D: lane l, reg r -> D[(l>>4) + 4r][l&15].
// ---------------------------------------------------------------------------------------
constexpr int LDM = T + 2;
typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void tile_abt_mfma(const double* A, const double* B, f64x4 acc[4]) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const double* Bp = B + (16 * wv + r16) * LDM + kq;
  const double* Ap = A + r16 * LDM + kq;
#pragma unroll
  for (int k0 = 0; k0 < T; k0 += 4) {
    const double bv = Bp[k0];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const double av = Ap[16 * rt * LDM + k0];
      acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[rt], 0, 0, 0);
    }
  }
}
// A tile (or 16 rows of one) travels global -> registers -> LDS in two separate phases: every 16-byte load of a thread first, then the LDS
// writes.  (Written as one loop of load + store, the compiler clustered five of a tile's eight loads and waited for each of the other
// three on its own: four memory latencies per tile, eight for the two operands of a product.)
struct TileRegs { double2 v[T * T / 2 / kThreads]; };    // 8 x 16 bytes per thread of a 256-thread group
__device__ __forceinline__ void tile_fetch(TileRegs& r, const double* __restrict__ src) {
#pragma unroll
  for (int x = 0; x < T * T / 2 / kThreads; ++x) r.v[x] = reinterpret_cast<const double2*>(src)[threadIdx.x + kThreads * x];
}
__device__ __forceinline__ void tile_put(double* dst, const TileRegs& r) {   // row-major tile -> LDS with leading dimension LDM
#pragma unroll
  for (int x = 0; x < T * T / 2 / kThreads; ++x) {
    const int e = threadIdx.x + kThreads * x, row = e / (T / 2), c2 = e % (T / 2);
    dst[row * LDM + 2 * c2] = r.v[x].x; dst[row * LDM + 2 * c2 + 1] = r.v[x].y;
  }
}
__device__ __forceinline__ void stage_tile(double* dst, const double* __restrict__ src) {
  TileRegs r;
  tile_fetch(r, src);
  tile_put(dst, r);
}
// two tiles: all sixteen loads in flight together
__device__ __forceinline__ void stage_tiles(double* dstA, const double* __restrict__ srcA, double* dstB, const double* __restrict__ srcB) {
  TileRegs ra, rb;
  tile_fetch(ra, srcA);
  tile_fetch(rb, srcB);
  tile_put(dstA, ra);
  tile_put(dstB, rb);
}

// Row slice q (16 rows) of  C += A * B^T:  As holds rows [16 q, 16 q + 16) of A, B the whole tile; wavefront w computes the
// 16x16 output tile of columns [16 w, 16 w + 16).  Used on thin levels, where a tile product is latency-critical and the
// device is mostly idle: four workgroups share one product.
__device__ __forceinline__ void tile_abt_mfma_rows(const double* As, const double* B, f64x4& acc) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const double* Ap = As + r16 * LDM + kq;
  const double* Bp = B + (16 * wv + r16) * LDM + kq;
#pragma unroll
  for (int k0 = 0; k0 < T; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ap[k0], Bp[k0], acc, 0, 0, 0);
}
__device__ __forceinline__ void stage_rows16(double* dst, const double* __restrict__ src) {   // 16 rows of a row-major tile -> LDS (LDM)
  for (int e = threadIdx.x; e < 16 * T / 2; e += kThreads) {
    const int r = e / (T / 2), c2 = e % (T / 2);
    const double2 v = reinterpret_cast<const double2*>(src)[e];
    dst[r * LDM + 2 * c2] = v.x; dst[r * LDM + 2 * c2 + 1] = v.y;
  }
}

// ---------------------------------------------------------------------------------------
// potrf of one 64x64 tile with the trailing updates on the matrix cores.
// A right-looking panel of 4 columns is exactly the K of v_mfma_f64_16x16x4_f64, so per panel step kb:
//   (a) the 4x4 diagonal block is factorised and its factor inverted (D)                        [scalar, the serial part]
//   (b) the solved panel X = P D^T (P = current panel columns) is never stored as a matrix: a wavefront that needs the rows of
//       tile row J as an MFMA operand forms them itself, lane (k, n) <- X[16 J + n][k] = row k of D times the four panel values of
//       row 16 J + n (four fma; the A and B fragment layout alike), so nothing goes through LDS in between
//   (c) trailing matrix  A -= X X^T  and  Acc -= X W  (W rows = D Acc rows, four fma per lane, again directly in B fragment
//       layout):  one MFMA per 16x16 tile, finished rows / columns masked out of the operands
// The trailing matrices stay in registers in the MFMA accumulator layout (a wavefront owns a tile row; wavefronts 0-3 A,
// wavefronts 4-7 the accumulator of L^-1).  At the end of a step the 4 columns (rows) the next step needs are published to
// LDS in operand layout (row-major [64][4]: a fragment is 64 consecutive doubles).
// (a) is off the other wavefronts' path (look-ahead): the diagonal block of step kb + 1 after the update of step kb is
// W - X X^T with W its 4x4 values BEFORE that update (published, with the panel, during step kb - 1) and X its four rows of
// the panel times D_kb^T -- 24 fma on 16 lanes -- so one wavefront (accumulator row 0, idle after the first four steps) forms
// and factorises it during step kb, while the others run the update, and D_kb+1 is in LDS when step kb + 1 starts.
// The loop's critical path is that wavefront's chain D_kb -> D_kb+1 (about 130 dependent fp64 instructions) or the heaviest
// tile row, whichever is longer (scripts/potrf_bench.hip: 17.0 -> 15.4 us for one tile with the look-ahead, 14.2 us with (b) as fma instead of
// an MFMA per tile row and the epilogue's row sums on DPP moves).
// ONE workgroup barrier per panel step; P, W, D and the published accumulator rows alternate between two buffers.
// Fragment layout: A: lane l -> A[l&15][l>>4]; B: lane l -> B[l>>4][l&15]; D: lane l, reg r -> D[(l>>4) + 4r][l&15].
// ---------------------------------------------------------------------------------------
constexpr int kPotrfMfmaLds = T * LD + 2 * (T * 4) + 2 * (T * 4) + 32 + T + 32;   // doubles
struct PotrfLds { double *Lsh, *Psh2, *AR2, *Dsh2, *Wsh2; };
// 4x4 Cholesky of a diagonal block and the inverse of its factor.  2x2 block pivots: for the pivot block
// (p q; q r) the reciprocal square roots of p and of p r - q^2 are independent, so two columns cost one rsqrt latency:
// l00 = p i0, l10 = q i0, 1/l11 = rsqrt(det) l00  (det has the same cancellation as r - l10^2).
// A non-positive pivot is flagged and poisons the tile (NaN); the step is then rejected on the host.
// Only the inverse is handed on (its strict upper part stays zero from the start): the block's own factor falls out of the panel
// solve, L_kk = A_kk D^T, with the rows below it.  d = (D00, D10, D11, D20, D21, D22, D30, D31, D32, D33).
struct Diag4 { double b00, b10, b11, b20, b21, b22, b30, b31, b32, b33; };
__device__ __forceinline__ void potrf_factor_block(const Diag4& B, double (&d)[10], double& bad) {
  const double p = B.b00, q = B.b10, r = B.b11;
  const double det = fma(p, r, -(q * q));
  const double i0 = fast_rsqrt(p), id = fast_rsqrt(det);
  const double l00 = p * i0, l10 = q * i0, i1 = id * l00;
  const double l20 = B.b20 * i0, l30 = B.b30 * i0;
  const double l21 = fma(-l20, l10, B.b21) * i1, l31 = fma(-l30, l10, B.b31) * i1;
  const double p2 = fma(-l21, l21, fma(-l20, l20, B.b22)), q2 = fma(-l31, l21, fma(-l30, l20, B.b32)), r2 = fma(-l31, l31, fma(-l30, l30, B.b33));
  const double det2 = fma(p2, r2, -(q2 * q2));
  const double i2 = fast_rsqrt(p2), id2 = fast_rsqrt(det2);
  const double l22 = p2 * i2, l32 = q2 * i2, i3 = id2 * l22;
  bad = fmin(bad, fmin(fmin(p, det), fmin(p2, det2)));   // smallest pivot quantity so far; fmin skips a NaN, the product below does not
  const double d10 = -l10 * i0 * i1, d21 = -l21 * i1 * i2, d32 = -l32 * i2 * i3;
  const double d20 = -(l20 * i0 + l21 * d10) * i2, d31 = -(l31 * i1 + l32 * d21) * i3;
  const double d30 = -(l30 * i0 + l31 * d10 + l32 * d20) * i3;
  bad = fma(0.0, i3, bad);   // i3 is NaN whenever a pivot quantity was
  d[0] = i0; d[1] = d10; d[2] = i1; d[3] = d20; d[4] = d21; d[5] = i2; d[6] = d30; d[7] = d31; d[8] = d32; d[9] = i3;
}
__device__ __forceinline__ double lane_value(double v, int src) {   // v of lane src, wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
// The look-ahead wavefront's duty of panel step kb: the inverse factor D of diagonal block kb + 1 while the other wavefronts run the
// trailing update of panel kb.  The block after that update is  W - X X^T  with W the block's 4x4 values before the update (published
// by its owner during step kb - 1), X = (rows of the block in panel kb) D_kb^T, D_kb in this wavefront's registers (d) from the step
// before.  Lane 4 a + b forms entry (a, b); the ten entries of the lower triangle are then made wave-uniform and every lane runs the
// 4x4 factorisation (no divergence); lane 0 publishes D.  Prow == nullptr: the block is W itself (block 0, before the loop).
__device__ __forceinline__ void potrf_lookahead(const double* Prow, const double* W, double* Dout, double (&d)[10], double& bad) {
  const int lane = threadIdx.x & 63, a = (lane >> 2) & 3, b = lane & 3;
  double bp = W[a * 4 + b];
  if (Prow) {
    const double* pa = Prow + a * 4;
    const double* pb = Prow + b * 4;
    const double a0 = pa[0], a1 = pa[1], a2 = pa[2], a3 = pa[3], b0 = pb[0], b1 = pb[1], b2 = pb[2], b3 = pb[3];
    const double xa0 = a0 * d[0], xa1 = fma(a1, d[2], a0 * d[1]), xa2 = fma(a2, d[5], fma(a1, d[4], a0 * d[3])), xa3 = fma(a3, d[9], fma(a2, d[8], fma(a1, d[7], a0 * d[6])));
    const double xb0 = b0 * d[0], xb1 = fma(b1, d[2], b0 * d[1]), xb2 = fma(b2, d[5], fma(b1, d[4], b0 * d[3])), xb3 = fma(b3, d[9], fma(b2, d[8], fma(b1, d[7], b0 * d[6])));
    bp = fma(-xa3, xb3, fma(-xa2, xb2, fma(-xa1, xb1, fma(-xa0, xb0, bp))));
  }
  Diag4 B;
  B.b00 = lane_value(bp, 0); B.b10 = lane_value(bp, 4); B.b11 = lane_value(bp, 5); B.b20 = lane_value(bp, 8); B.b21 = lane_value(bp, 9);
  B.b22 = lane_value(bp, 10); B.b30 = lane_value(bp, 12); B.b31 = lane_value(bp, 13); B.b32 = lane_value(bp, 14); B.b33 = lane_value(bp, 15);
  potrf_factor_block(B, d, bad);
  if (lane == 0) { Dout[0] = d[0]; Dout[4] = d[1]; Dout[5] = d[2]; Dout[8] = d[3]; Dout[9] = d[4]; Dout[10] = d[5]; Dout[12] = d[6]; Dout[13] = d[7]; Dout[14] = d[8]; Dout[15] = d[9]; }
}
// the panel loop of one wavefront: tile row I of A (FAC) or of the accumulator of L^-1
template <bool FAC>
__device__ __forceinline__ void potrf_mfma_wave(const PotrfLds& s, const int I, f64x4 (&acc)[4], double (&dreg)[10], double& bad) {
  const int lane = threadIdx.x & 63, q = lane >> 4, c = lane & 15;
  OBVI_TICK(0);
  OBVI_TICK(1);
#pragma unroll 1
  for (int Ik = 0; Ik < 4; ++Ik) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int kb = 4 * Ik + m, done = 4 * kb + 3;   // rows / columns <= done are finished after this panel
      const int par = m & 1;                            // = kb & 1
      const double* P = s.Psh2 + par * (T * 4);
      const double* D = s.Dsh2 + par * 16;
      double* Pn = s.Psh2 + (1 - par) * (T * 4);
      // the look-ahead wavefront (accumulator row 0: one tile during the first four steps, nothing afterwards) first: its chain
      // D_kb -> D_kb+1 is the critical path of the loop
      if (!FAC && I == 0 && kb + 1 < 16) potrf_lookahead(P + 4 * (kb + 1) * 4, s.Wsh2 + par * 16, s.Dsh2 + (1 - par) * 16, dreg, bad);
      const bool live = I > Ik || (I == Ik && m < 3);   // the tile row still has rows below the diagonal block
      const int Jn = m == 3 ? Ik + 1 : Ik, mn = (m + 1) & 3;   // tile column / column block of the next panel
      // operand of tile row J for the trailing updates: lane (q, c) <- X[16 J + c][q], X = P D^T: row q of D (lower triangular, zeros stored)
      // times the four panel values of row 16 J + c -- four fma per lane.  (As an MFMA with D padded to 16 x 4 the solve cost a 64-cycle
      // slot of the matrix pipe and 17 wait states before its result could be masked, per tile and step.)
      const double dq0 = D[q * 4], dq1 = D[q * 4 + 1], dq2 = D[q * 4 + 2], dq3 = D[q * 4 + 3];
      auto solved_rows = [&](int J) -> double {
        const double* p4 = P + (16 * J + c) * 4;
        return fma(dq3, p4[3], fma(dq2, p4[2], fma(dq1, p4[1], dq0 * p4[0])));
      };
      const bool live2 = I >= Ik;   // ... or the rows of the diagonal block itself (their part of X = P D^T is L_kk)
      double xI = 0.0, an = 0.0;
      if (FAC && live2) {
        xI = solved_rows(I);
        an = (16 * I + c > done) ? -xI : 0.0;
      }
      if (FAC) {
        if (live) {
#pragma unroll
          for (int J = 3; J >= 0; --J) if (J <= I && (J > Ik || (J == Ik && m < 3))) {   // the tile with the next panel's columns is among the first
            const double xJ = J == I ? xI : solved_rows(J);
            const double bv = (16 * J + c > done) ? xJ : 0.0;
            acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(an, bv, acc[J], 0, 0, 0);
          }
        }
        if (kb + 1 < 16) {
          if (I >= Jn && (c >> 2) == mn) {
#pragma unroll
            for (int J = 0; J < 4; ++J)
              if (J == Jn) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Pn[(16 * I + q + 4 * r) * 4 + (c & 3)] = acc[J][r];
              }
          }
        }
        if (kb + 2 < 16) {   // the 4x4 diagonal block after next, as it stands after this panel's update: the look-ahead wavefront's W of the next step
          const int m2 = (m + 2) & 3;
          const int J2 = Ik + (m >= 2 ? 1 : 0);
          if (I == J2 && (c >> 2) == m2) {
            double* Wn = s.Wsh2 + (1 - par) * 16;
#pragma unroll
            for (int J = 0; J < 4; ++J)
              if (J == J2) Wn[q * 4 + (c & 3)] = acc[J][m2];
          }
        }
        if (live2 && 16 * I + c >= 4 * kb) s.Lsh[(16 * I + c) * LD + 4 * kb + q] = xI;   // the panel's part of L incl. the diagonal block (read at the end)
      } else if (I >= Ik) {
        // rows kb of W = D Acc(rows kb): lane (q, c) forms W[4 kb + q][16 J + c], the B operand it needs; the wavefront that
        // owns these rows keeps them
        const double* AR = s.AR2 + par * (T * 4);
        const double d0 = dq0, d1 = dq1, d2 = dq2, d3 = dq3;
        double ar[4][4];
#pragma unroll
        for (int J = 0; J < 4; ++J)
#pragma unroll
          for (int t = 0; t < 4; ++t) ar[J][t] = AR[(16 * J + c) * 4 + t];
        if (live) {
          xI = solved_rows(I);
          an = (16 * I + c > done) ? -xI : 0.0;
        }
#pragma unroll
        for (int J = 0; J < 4; ++J) if (J <= Ik) {
          const double w = fma(d3, ar[J][3], fma(d2, ar[J][2], fma(d1, ar[J][1], d0 * ar[J][0])));
          if (I == Ik) acc[J][m] = w;
          if (live) acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(an, w, acc[J], 0, 0, 0);
        }
        if (kb + 1 < 16 && I == Jn) {
          double* ARn = s.AR2 + (1 - par) * (T * 4);
#pragma unroll
          for (int J = 0; J < 4; ++J) if (J <= I) ARn[(16 * J + c) * 4 + q] = acc[J][mn];
        }
      }
      OBVI_PH(ph1);
      __syncthreads();
      OBVI_PH(ph2);
    }
  }
  OBVI_TICK(2);
}
template <bool FAC>
__device__ __forceinline__ void potrf_mfma_rows(const int I, double* smem, double* S, int nt, int k, double* Linv_all, double* rhs, double* scal, const double* pre_tile, const double* pre_z) {
  PotrfLds s;
  s.Lsh = smem;                 // L, row-major (LD)
  s.Psh2 = s.Lsh + T * LD;      // 2 x [64][4] current values of the panel columns, all rows (alternating with the panel index)
  s.AR2 = s.Psh2 + 2 * T * 4;   // 2 x [64][4] AR[c][t] = row 4kb+t of the accumulator of W, column c
  s.Dsh2 = s.AR2 + 2 * T * 4;   // 2 x 4x4 inverse of L_kk
  double* zsh = s.Dsh2 + 32;
  s.Wsh2 = zsh + T;             // 2 x 4x4 diagonal block the look-ahead wavefront factorises next (values before the running panel's update)
  double* tile = tile_ptr(S, nt, k, k);
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr bool fac = FAC;
  OBVI_MARK(0);
  const int q = lane >> 4, c = lane & 15;
  f64x4 acc[4];
#pragma unroll
  for (int J = 0; J < 4; ++J)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * I + q + 4 * r, col = 16 * J + c;
      double v = 0.0;
      if (fac) {
        if (J <= I) {
          const int rr = (J == I && col > row) ? col : row, cc = (J == I && col > row) ? row : col;   // diagonal tiles are kept symmetric
          v = pre_tile ? pre_tile[rr * (T + 2) + cc] : tile[rr * T + cc];
        }
      } else v = row == col ? 1.0 : 0.0;
      acc[J][r] = v;
    }
  double zv = 0.0;
  if (tid < T) zv = pre_z ? pre_z[tid] : rhs[(int64_t)k * T + tid];
  if (pre_tile) __syncthreads();
  for (int e = tid; e < 2 * T * 4; e += 512) s.AR2[e] = 0.0;
  if (tid < 32) s.Dsh2[tid] = 0.0;
  if (fac && c < 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s.Psh2[(16 * I + q + 4 * r) * 4 + c] = acc[0][r];
  }
  if (fac && I == 0 && c < 8) s.Wsh2[(c < 4 ? 16 : 0) + q * 4 + (c & 3)] = acc[0][c >> 2];   // diagonal blocks 0 (second buffer) and 1 (first: step 0 reads it)
  double bad = 1.0, dreg[10] = {};
  __syncthreads();
  if (!fac && I == 0) {
    potrf_lookahead(nullptr, s.Wsh2 + 16, s.Dsh2, dreg, bad);
    s.AR2[c * 4 + q] = acc[0][0];
  }
  if (tid < T) zsh[tid] = zv;   // (only read behind the panel loop; stored here so that the wait for its load does not sit in front of the tile's loads)
  __syncthreads();
  OBVI_MARK(1);
  potrf_mfma_wave<FAC>(s, I, acc, dreg, bad);
  OBVI_MARK(2);
  if (!(bad > 0.0) && lane == 0) unsafeAtomicAdd(scal + SC_CHOL_FAIL, 1.0);   // (every lane of the look-ahead wavefront holds the same verdict)
  OBVI_MARK(3);
  if (fac) {   // L: coalesced from LDS; the strict upper part was never written (or holds round-off of the diagonal blocks): zeros
    double lv[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) { const int e = tid + 256 * it; lv[it] = (e & 63) <= (e >> 6) ? s.Lsh[(e >> 6) * LD + (e & 63)] : 0.0; }   // all reads first
#pragma unroll
    for (int it = 0; it < 16; ++it) tile[tid + 256 * it] = lv[it];
  } else {
    // L^-1 from the accumulator registers, and z_k = L^-1 b_k without staging L^-1: every lane multiplies its elements with
    // b and the 16 lanes that share a row add up
    double* Li = Linv_all + (int64_t)k * (T * T);
    double zp[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int J = 0; J < 4; ++J) {
      const double bz = zsh[16 * J + c];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + q + 4 * r, col = 16 * J + c;
        const double v = (col <= row) ? acc[J][r] : 0.0;
        Li[row * T + col] = v;
        zp[r] = fma(v, bz, zp[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double v = zp[r];
      v = row16_sum_to_last(v);   // the 16 lanes that share the row (a DPP row): the sum lands in lane c == 15
      if (c == 15) rhs[(int64_t)k * T + 16 * I + q + 4 * r] = v;
    }
  }
  OBVI_MARK(4);
  OBVI_MARK(5);
}

__device__ __forceinline__ void potrf_mfma_tile(double* smem, double* S, int nt, int k, double* Linv_all, double* rhs, double* scal, const double* pre_tile = nullptr, const double* pre_z = nullptr) {
  // wavefronts 0-3: tile rows 0-3 of A; wavefronts 4-7: tile rows 0, 3, 2, 1 of the accumulator of L^-1.  Wavefronts w and w + 4 share a
  // SIMD: the look-ahead wavefront (accumulator row 0) sits beside tile row 0 of A, which is finished after four steps, and the heavy
  // rows of the two halves do not meet.  Two instruction streams (A / accumulator), tile row wave-uniform; both execute the same barriers.
  const int wvi = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wvi == 4) __builtin_amdgcn_s_setprio(3);   // the look-ahead wavefront's chain is the critical path of the panel loop
  if (wvi < 4) potrf_mfma_rows<true>(wvi, smem, S, nt, k, Linv_all, rhs, scal, pre_tile, pre_z);
  else potrf_mfma_rows<false>((8 - wvi) & 3, smem, S, nt, k, Linv_all, rhs, scal, pre_tile, pre_z);
}

__global__ void __launch_bounds__(512) k_potrf(double* S, int nt, const int32_t* __restrict__ klist, double* Linv_all, double* rhs, double* scal) {
  __shared__ double smem[kPotrfMfmaLds];
  potrf_mfma_tile(smem, S, nt, klist[blockIdx.x], Linv_all, rhs, scal);
}

__global__ void __launch_bounds__(kThreads) k_trsm(double* S, int nt, const int32_t* __restrict__ jobs, const double* __restrict__ Linv_all, int slices) {
  __shared__ double A[T * LDM];
  __shared__ double B[T * LDM];
  const int job = blockIdx.x / slices, q = blockIdx.x % slices;
  const int i = jobs[2 * job], k = jobs[2 * job + 1];
  double* tile = tile_ptr(S, nt, i, k);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (slices == 1) {
    stage_tiles(A, tile, B, Linv_all + (int64_t)k * (T * T));
    __syncthreads();
    f64x4 acc[4] = {};
    tile_abt_mfma(A, B, acc);   // X = S_ik * Linv^T
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[(16 * rt + (lane >> 4) + 4 * r) * T + 16 * wv + (lane & 15)] = acc[rt][r];
  } else {   // rows [16 q, 16 q + 16) of X: this workgroup reads and writes only these rows of the tile
    { TileRegs rb; tile_fetch(rb, Linv_all + (int64_t)k * (T * T)); stage_rows16(A, tile + 16 * q * T); tile_put(B, rb); }
    __syncthreads();
    f64x4 acc = {};
    tile_abt_mfma_rows(A, B, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) tile[(16 * q + (lane >> 4) + 4 * r) * T + 16 * wv + (lane & 15)] = acc[r];
  }
}

// job g < n_upd: tile target (i,j): S_ij -= sum_{k in list} L_ik L_jk^T
// job g >= n_upd: rhs target i:     b_i  -= sum_{k in list} L_ik z_k
// A target whose k-list was split over several jobs (flag != 0) accumulates with fp64 hardware atomics.
// The first 256 threads of the workgroup work; any others have left before the call.
struct UpdateJobs {
  int n_upd;
  const int32_t* upd_ij; const int32_t* upd_kptr; const int32_t* upd_k; const uint8_t* upd_flag;
  const int32_t* rh_i; const int32_t* rh_kptr; const int32_t* rh_k;
};
__device__ __forceinline__ void update_job(double* smem, double* S, int nt, const UpdateJobs& u, int g, double* rhs, int slice = -1) {
  double* A = smem;
  double* B = smem + T * LDM;
  const int tid = threadIdx.x;
  if (g < u.n_upd && slice >= 0) {   // rows [16 slice, 16 slice + 16) of the target
    // (the job's scalars once, in front: read inside the loop or behind it they are a load and a full wait each -- the compiler cannot
    // know that the tile stores do not alias them)
    const int i = u.upd_ij[2 * g], j = u.upd_ij[2 * g + 1], q0 = u.upd_kptr[g], q1 = u.upd_kptr[g + 1];
    const bool atomic = u.upd_flag[g] != 0;
    f64x4 acc = {};
    for (int q = q0; q < q1; ++q) {
      const int k = u.upd_k[q];
      __syncthreads();
      { TileRegs rb; tile_fetch(rb, tile_ptr(S, nt, j, k)); stage_rows16(A, tile_ptr(S, nt, i, k) + 16 * slice * T); tile_put(B, rb); }
      __syncthreads();
      tile_abt_mfma_rows(A, B, acc);
    }
    double* C = tile_ptr(S, nt, i, j);
    const int lane = tid & 63, wv = tid >> 6;
    double* c[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = &C[(16 * slice + (lane >> 4) + 4 * r) * T + 16 * wv + (lane & 15)];
    if (atomic) {
#pragma unroll
      for (int r = 0; r < 4; ++r) unsafeAtomicAdd(c[r], -acc[r]);
    } else {   // the four reads together, then the four writes
      double v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = *c[r];
#pragma unroll
      for (int r = 0; r < 4; ++r) *c[r] = v[r] - acc[r];
    }
    return;
  }
  if (g < u.n_upd) {
    const int i = u.upd_ij[2 * g], j = u.upd_ij[2 * g + 1], q0 = u.upd_kptr[g], q1 = u.upd_kptr[g + 1];
    const bool atomic = u.upd_flag[g] != 0;
    f64x4 acc[4] = {};
    for (int q = q0; q < q1; ++q) {
      const int k = u.upd_k[q];
      __syncthreads();
      if (i != j) stage_tiles(A, tile_ptr(S, nt, i, k), B, tile_ptr(S, nt, j, k));
      else stage_tile(A, tile_ptr(S, nt, i, k));
      __syncthreads();
      tile_abt_mfma(A, i != j ? B : A, acc);
    }
    double* C = tile_ptr(S, nt, i, j);
    const int lane = tid & 63, wv = tid >> 6;
    if (atomic) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) unsafeAtomicAdd(&C[(16 * rt + (lane >> 4) + 4 * r) * T + 16 * wv + (lane & 15)], -acc[rt][r]);
    } else {
      double v[4][4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[rt][r] = C[(16 * rt + (lane >> 4) + 4 * r) * T + 16 * wv + (lane & 15)];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) C[(16 * rt + (lane >> 4) + 4 * r) * T + 16 * wv + (lane & 15)] = v[rt][r] - acc[rt][r];
    }
  } else {
    const int h = g - u.n_upd;
    const int i = u.rh_i[h];
    // thread (r = tid/4, part = tid%4): 16 columns each
    const int r = tid >> 2, part = tid & 3;
    double s = 0.0;
    const int q0 = u.rh_kptr[h], q1 = u.rh_kptr[h + 1];
    for (int q = q0; q < q1; ++q) {
      const int k = u.rh_k[q];
      const double* X = tile_ptr(S, nt, i, k) + r * T + part * 16;
      const double* z = rhs + (int64_t)k * T + part * 16;
#pragma unroll
      for (int c = 0; c < 16; ++c) s += X[c] * z[c];
    }
    s = quad_sum(s);
    if (part == 0) rhs[(int64_t)i * T + r] -= s;
  }
}
__global__ void __launch_bounds__(kThreads) k_update(double* S, int nt, UpdateJobs u, double* rhs, int slices) {
  __shared__ double smem[2 * T * LDM];
  const int b = blockIdx.x, nu = slices * u.n_upd;
  if (b < nu) update_job(smem, S, nt, u, b / slices, rhs, slices > 1 ? b % slices : -1);
  else update_job(smem, S, nt, u, u.n_upd + (b - nu), rhs);
}

// potrf of tile column k preceded by the few products of the previous level that finish its diagonal tile and right-hand-side
// block (pre list, may be empty): wavefronts 0-3 form A_kk - sum L_kj L_kj^T on the matrix cores, wavefronts 4-7 z_k - sum L_kj z_j;
// the results stay in LDS and the factorisation starts from there.  512 threads, smem = 2 T LDM + T doubles.
// one 16x16 tile (rows of tile row rt, columns of tile row w) of  A A^T, A a 64x64 tile in LDS (LDM): 16 MFMAs
__device__ __forceinline__ void syrk_tile16(const double* A, int rt, int w, f64x4& acc) {
  const int lane = threadIdx.x & 63, r16 = lane & 15, kq = lane >> 4;
  const double* Ap = A + (16 * rt + r16) * LDM + kq;
  const double* Bp = A + (16 * w + r16) * LDM + kq;
#pragma unroll
  for (int k0 = 0; k0 < T; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ap[k0], Bp[k0], acc, 0, 0, 0);
}
__device__ __forceinline__ void potrf_column(double* smem, double* S, int nt, int k, int pb, int pe, const int32_t* __restrict__ pre_j, double* Linv_all, double* rhs, double* scal) {
  if (pe == pb) { potrf_mfma_tile(smem, S, nt, k, Linv_all, rhs, scal); return; }
  double* A = smem;
  double* Ct = smem + T * LDM;
  double* zpre = smem + 2 * T * LDM;
  const int tid = threadIdx.x, wvi = tid >> 6, lane = tid & 63;
  // The factorisation reads the lower triangle of A_kk - sum L_kj L_kj^T only: its ten 16x16 tiles are spread over the eight wavefronts
  // (tiles v and v + 8 of the list below: two tiles for wavefronts 0 and 1, one for the others; all four wavefronts of the first half used to
  // form whole block columns, four tiles in the first), wavefronts 4-7 also take z_k - sum L_kj z_j.
  //   tile list (row tile, column tile): (0,0) (1,0) (2,0) (3,0) (1,1) (2,1) (3,1) (2,2) (3,2) (3,3)
  constexpr int kRt[10] = {0, 1, 2, 3, 1, 2, 3, 2, 3, 3}, kW[10] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3};
  int rt0 = 0, w0 = 0, rt1 = 0, w1 = 0;
#pragma unroll
  for (int t = 0; t < 10; ++t) { if (t == wvi) { rt0 = kRt[t]; w0 = kW[t]; } if (t == wvi + 8) { rt1 = kRt[t]; w1 = kW[t]; } }
  const bool two = wvi < 2;
  f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  double zs = 0.0;
  for (int q = pb; q < pe; ++q) {
    const int j = pre_j[q];
    __syncthreads();
    if (tid < kThreads) stage_tile(A, tile_ptr(S, nt, k, j));
    __syncthreads();
    syrk_tile16(A, rt0, w0, acc0);
    if (two) syrk_tile16(A, rt1, w1, acc1);
    if (tid >= kThreads) {
      const int r = (tid - kThreads) >> 2, part = tid & 3;
      const double* z = rhs + (int64_t)j * T + part * 16;
#pragma unroll
      for (int c = 0; c < 16; ++c) zs += A[r * LDM + part * 16 + c] * z[c];
    }
  }
  {
    const double* C = tile_ptr(S, nt, k, k);
    const int q4 = lane >> 4, c16 = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * rt0 + q4 + 4 * r, col = 16 * w0 + c16;
      Ct[row * LDM + col] = C[row * T + col] - acc0[r];
    }
    if (two) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * rt1 + q4 + 4 * r, col = 16 * w1 + c16;
        Ct[row * LDM + col] = C[row * T + col] - acc1[r];
      }
    }
  }
  if (tid >= kThreads) {
    const int r = (tid - kThreads) >> 2, part = tid & 3;
    zs = quad_sum(zs);
    if (part == 0) zpre[r] = rhs[(int64_t)k * T + r] - zs;
  }
  __syncthreads();
  potrf_mfma_tile(smem, S, nt, k, Linv_all, rhs, scal, Ct, zpre);
}
// the potrf half of k_update_potrf as its own launch (OBVI_FUSED_POTRF=0: update jobs of a level, then this; no waiting inside a launch)
__global__ void __launch_bounds__(512) k_potrf_pre(double* S, int nt, const int32_t* __restrict__ klist, const int32_t* __restrict__ pre_ptr, const int32_t* __restrict__ pre_j,
                                                  double* Linv_all, double* rhs, double* scal) {
  __shared__ double smem[2 * T * LDM + T];
  potrf_column(smem, S, nt, klist[blockIdx.x], pre_ptr[blockIdx.x], pre_ptr[blockIdx.x + 1], pre_j, Linv_all, rhs, scal);
}

// One launch for the updates of level l and the potrf of level l+1.  Grid order: the update / right-hand-side jobs whose
// target is the diagonal tile or right-hand-side block of a tile column of level l+1 ("critical": n_crit_upd + n_crit_rh,
// first in their job lists), then the potrf workgroups of level l+1, then all other jobs of level l.  A critical job bumps
// its column's counter when its result is out (release); the column's potrf workgroup starts as soon as the counter
// reaches the number of such jobs (acquire) and runs while the rest of level l is still being updated.
// No deadlock: the jobs waited for have lower block indices and never wait themselves.  The wait is bounded (a lost
// signal becomes a failed step).
__global__ void __launch_bounds__(512) k_update_potrf(double* S, int nt, UpdateJobs u, int n_rh, int n_crit_upd, int n_crit_rh, int n_potrf, int slices,
                                                     const int32_t* __restrict__ job_signal, const int32_t* __restrict__ klist,
                                                     const int32_t* __restrict__ k_need, const int32_t* __restrict__ pre_ptr, const int32_t* __restrict__ pre_j,
                                                     int32_t* done, double* Linv_all, double* rhs, double* scal) {
  __shared__ double smem[2 * T * LDM + T];
  static_assert(kPotrfMfmaLds <= 2 * T * LDM, "potrf fits the update buffers");
  const int b = blockIdx.x, n_crit = slices * n_crit_upd + n_crit_rh;
  if (b >= n_crit && b < n_crit + n_potrf) {
    const int k = klist[b - n_crit], need = k_need[b - n_crit];
    if (need > 0) {
      if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(done + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
          __builtin_amdgcn_s_sleep(8);
          // a lost or late signal (the jobs waited for have lower block indices, but HIP does not promise dispatch order) is not a
          // numerical event: it is reported through its own scalar and obvi_ba_solve returns OBVI_ERR_HIP
          if (++spins > (1 << 22)) { unsafeAtomicAdd(scal + SC_WAIT_TIMEOUT, 1.0); unsafeAtomicAdd(scal + SC_CHOL_FAIL, 1.0); break; }
        }
      }
      __syncthreads();
      __threadfence();
    }
    potrf_column(smem, S, nt, k, pre_ptr[b - n_crit], pre_ptr[b - n_crit + 1], pre_j, Linv_all, rhs, scal);
    return;
  }
  if (threadIdx.x >= kThreads) return;
  // job id g: updates [0, n_upd), right-hand sides [n_upd, n_upd + n_rh); with slices == 4 every update job is four workgroups
  int g, slice = -1;
  const int nc = slices * n_crit_upd;
  if (b < nc) { g = b / slices; if (slices > 1) slice = b % slices; }
  else if (b < nc + n_crit_rh) g = u.n_upd + (b - nc);
  else {
    const int r = b - nc - n_crit_rh - n_potrf, nr = slices * (u.n_upd - n_crit_upd);
    if (r < nr) { g = n_crit_upd + r / slices; if (slices > 1) slice = r % slices; }
    else g = u.n_upd + n_crit_rh + (r - nr);
  }
  update_job(smem, S, nt, u, g, rhs, slice);
  const int sig = job_signal[g];
  if (sig >= 0) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(done + sig, 1);
  }
  (void)n_rh;
}

// backward substitution (L^T y = z), row oriented, one workgroup per tile of the launch's tile rows.  t starts as z and is updated
// in place:  every workgroup of row k forms y_k = L_kk^-T t_k; the workgroup of tile (k,j) subtracts L_kj^T y_k from t_j, the
// workgroup with j = -1 stores y_k.
// A launch covers SEVERAL levels of the elimination tree (a separator's chain of tile columns is one launch instead of 4-6): a row k
// whose ancestors a_0 > ... > a_{n-1} (top first, a_{n-1} its parent) sit in the same launch does not wait for launch boundaries to
// receive their contributions -- its workgroups walk the chain themselves from values that are final when the launch starts:
//     y_a0 = L^-T t_a0,   y_as = L^-T (t_as - sum_{u<s} L_{au,as}^T y_au),   ...,   y_k = L^-T (t_k - sum_u L_{au,k}^T y_au)
// (the work is redundant across the workgroups of a row and across rows; the device is idle at this point and a step is two
// reductions, a launch boundary is several microseconds).  Tiles between two rows of one launch get no workgroup.
// Chain record: n, a_0 .. a_{n-1}, then the presence bits of the tiles (a_u, a_s) / (a_u, k) (bit 8 s + u, step n = row k), 2 words.
constexpr int kBackThreads = 512, kBackChain = 7;
__global__ void __launch_bounds__(kBackThreads) k_backward(const double* __restrict__ S, int nt, const int32_t* __restrict__ kj, const int32_t* __restrict__ chains,
                                                         const double* __restrict__ Linv_all, double* t, double* y) {
  constexpr int Q = kBackThreads / T, R = T / Q;   // row slices, rows per slice
  __shared__ double part[Q][T];
  __shared__ double ych[kBackChain + 1][T], vsh[T];
  const int k = kj[3 * blockIdx.x], j = kj[3 * blockIdx.x + 1];
  const int32_t* ch = chains + kj[3 * blockIdx.x + 2];
  const int n = ch[0];
  const uint64_t bits = (uint64_t)(uint32_t)ch[1 + n] | ((uint64_t)(uint32_t)ch[2 + n] << 32);
  const int tid = threadIdx.x, c = tid % T, q = tid / T;
  auto reduce = [&](double s) -> double {   // sum over the row slices; valid for tid < T (column tid)
    part[q][c] = s;
    __syncthreads();
    double a = 0.0;
    if (tid < T) {
#pragma unroll
      for (int i = 0; i < Q; ++i) a += part[i][tid];
    }
    return a;
  };
  // the tile of the workgroup itself: in flight from the start
  double x[R];
  if (j >= 0) {
    const double* X = tile_ptr(const_cast<double*>(S), nt, k, j) + (q * R) * T + c;
#pragma unroll
    for (int r = 0; r < R; ++r) x[r] = X[r * T];
  }
  if (n <= 3) {
    // Short chains (the default of four levels per launch): everything the chain will read -- the L^-1 rows of every step, the tiles between
    // the ancestors, the t blocks -- is requested before the first step, so that a step is two reductions instead of a memory latency and two
    // reductions.  Slot of the tile (ancestor u, step st): st (st - 1) / 2 + u.
    double wv[4][R], xv[6][R], tv[4];
#pragma unroll
    for (int st = 0; st < 4; ++st)
      if (st <= n) {
        const int m = st < n ? ch[1 + st] : k;
        const double* Li = Linv_all + (int64_t)m * (T * T) + (q * R) * T + c;
#pragma unroll
        for (int r = 0; r < R; ++r) wv[st][r] = Li[r * T];
        tv[st] = tid < T ? t[(int64_t)m * T + tid] : 0.0;
#pragma unroll
        for (int u = 0; u < 3; ++u)
          if (u < st && ((bits >> (8 * st + u)) & 1ull)) {
            const double* Xu = tile_ptr(const_cast<double*>(S), nt, ch[1 + u], m) + (q * R) * T + c;
#pragma unroll
            for (int r = 0; r < R; ++r) xv[st * (st - 1) / 2 + u][r] = Xu[r * T];
          }
      }
#pragma unroll
    for (int st = 0; st < 4; ++st)
      if (st <= n) {
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < 3; ++u)
          if (u < st && ((bits >> (8 * st + u)) & 1ull)) {
#pragma unroll
            for (int r = 0; r < R; ++r) s += xv[st * (st - 1) / 2 + u][r] * ych[u][q * R + r];
          }
        if (st > 0) {
          const double a = reduce(s);
          if (tid < T) vsh[tid] = tv[st] - a;
          __syncthreads();
        } else {
          if (tid < T) vsh[tid] = tv[0];
          __syncthreads();
        }
        s = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) s += wv[st][r] * vsh[q * R + r];   // L^-1 is lower triangular with an explicit zero upper part
        const double a = reduce(s);
        if (tid < T) { ych[st][tid] = a; if (st == n && j < 0) y[(int64_t)k * T + tid] = a; }
        __syncthreads();
      }
  } else
  for (int st = 0; st <= n; ++st) {
    const int m = st < n ? ch[1 + st] : k;
    // all tiles of the step are loaded before the first is used
    double w[R], xu[kBackChain][R];
    const double* Li = Linv_all + (int64_t)m * (T * T) + (q * R) * T + c;
#pragma unroll
    for (int r = 0; r < R; ++r) w[r] = Li[r * T];
#pragma unroll
    for (int u = 0; u < kBackChain; ++u)
      if (u < st && ((bits >> (8 * st + u)) & 1ull)) {
        const double* Xu = tile_ptr(const_cast<double*>(S), nt, ch[1 + u], m) + (q * R) * T + c;
#pragma unroll
        for (int r = 0; r < R; ++r) xu[u][r] = Xu[r * T];
      }
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < kBackChain; ++u)
      if (u < st && ((bits >> (8 * st + u)) & 1ull)) {
#pragma unroll
        for (int r = 0; r < R; ++r) s += xu[u][r] * ych[u][q * R + r];
      }
    if (st > 0) {
      const double a = reduce(s);
      if (tid < T) vsh[tid] = t[(int64_t)m * T + tid] - a;
      __syncthreads();
    }
    s = 0.0;
    const double* vm = st > 0 ? &vsh[q * R] : t + (int64_t)m * T + q * R;
#pragma unroll
    for (int r = 0; r < R; ++r) s += w[r] * vm[r];   // L^-1 is lower triangular with an explicit zero upper part
    const double a = reduce(s);
    if (tid < T) { ych[st][tid] = a; if (st == n && j < 0) y[(int64_t)k * T + tid] = a; }
    __syncthreads();
  }
  if (j < 0) return;
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r) s += x[r] * ych[n][q * R + r];
  const double a = reduce(s);
  if (tid < T) unsafeAtomicAdd(t + (int64_t)j * T + tid, -a);
}

// Deterministic mode (obvi_ba_options.deterministic): column-oriented backward substitution, one workgroup per tile column k of a level,
// levels descending:  y_k = L_kk^-T (z_k - sum_{i in col(k)} L_ik^T y_i)  with the column's tiles walked in list order -- every y_i
// belongs to a higher level and is final.  One writer per block of y, no atomics (k_backward above lets the rows of a launch subtract
// from t_j atomically, in whatever order they finish); a launch per level instead of one per four.
__global__ void __launch_bounds__(kThreads) k_backward_det(const double* __restrict__ S, int nt, const int32_t* __restrict__ klist, const int32_t* __restrict__ col_ptr,
                                                         const int32_t* __restrict__ col_i, const double* __restrict__ Linv_all, const double* __restrict__ z, double* y) {
  constexpr int Q = kThreads / T, R = T / Q;   // row slices, rows per slice
  __shared__ double part[Q][T];
  __shared__ double v[T];
  const int k = klist[blockIdx.x];
  const int tid = threadIdx.x, c = tid % T, q = tid / T;
  double s = 0.0;
  for (int e = col_ptr[k]; e < col_ptr[k + 1]; ++e) {
    const int i = col_i[e];
    const double* X = tile_ptr(const_cast<double*>(S), nt, i, k) + (q * R) * T + c;
    const double* yi = y + (int64_t)i * T + q * R;
#pragma unroll
    for (int r = 0; r < R; ++r) s += X[r * T] * yi[r];
  }
  part[q][c] = s;
  __syncthreads();
  if (tid < T) {
    double a = 0.0;
#pragma unroll
    for (int i = 0; i < Q; ++i) a += part[i][tid];
    v[tid] = z[(int64_t)k * T + tid] - a;
  }
  __syncthreads();
  const double* Li = Linv_all + (int64_t)k * (T * T) + (q * R) * T + c;
  s = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r) s += Li[r * T] * v[q * R + r];   // L^-1 carries an explicit zero upper part
  part[q][c] = s;
  __syncthreads();
  if (tid < T) {
    double a = 0.0;
#pragma unroll
    for (int i = 0; i < Q; ++i) a += part[i][tid];
    y[(int64_t)k * T + tid] = a;
  }
}

// ---------------------------------------------------------------------------------------
// Covariance blocks (obvi_ba_object_covariances): forward substitution with many right-hand sides on the factor
// that is already in the tiles, Y = L^-1 E (E: the unit vectors of the object rows), then blocks of S^-1 = Y^T Y.
// Y is kept transposed, Yt[64 nslabs][ldt = 64 nt] row-major: right-hand side c of slab `sl` is row 64 sl + c, so the
// substitution of tile row k has the shape of the factorisation's update,
//     Yt_k = (E_k - sum_j Yt_j L_kj^T) L_kk^-T          (64x64 blocks, C += A B^T on the matrix cores, tile_abt_mfma)
// with A = the (slab, j) block of Yt and B = the tile L_kj, both staged in LDS.  One workgroup per (tile row k of the
// level, slab); the tiles (k, j) of a row belong to earlier levels.  Objects are numbered in elimination order, so a slab
// is zero left of the first row of its first object: slab_first[sl] = that tile; work left of it is skipped (Yt is
// cleared beforehand).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_block_ld(double* dst, const double* src, int64_t ld) {   // 64x64 block of a row-major matrix -> LDS (LDM)
  for (int e = threadIdx.x; e < T * T / 2; e += kThreads) {
    const int r = e / (T / 2), c2 = e % (T / 2);
    const double2 v = *reinterpret_cast<const double2*>(src + (int64_t)r * ld + 2 * c2);
    dst[r * LDM + 2 * c2] = v.x; dst[r * LDM + 2 * c2 + 1] = v.y;
  }
}
// nsplit = 1: the whole row in one workgroup.  nsplit > 1 (levels with long rows, i.e. the separators near the root, where
// a level has few rows of 100+ tiles): workgroup z takes every nsplit-th tile of the row and subtracts its partial sum from
// E_k atomically; k_forward_multi_diag then applies L_kk^-T.
__global__ void __launch_bounds__(kThreads) k_forward_multi(const double* __restrict__ S, int nt, const int32_t* __restrict__ lvl_k,
                                                           const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ row_j,
                                                           const double* __restrict__ Linv_all, double* Yt, int64_t ldt,
                                                           const int32_t* __restrict__ slab_first, int nsplit) {
  __shared__ double smem[2 * T * LDM];
  double* A = smem;
  double* B = smem + T * LDM;
  const int k = lvl_k[blockIdx.x], sl = blockIdx.y;
  const int first = slab_first[sl];
  if (k < first) return;                          // uniform per workgroup
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double* Yslab = Yt + (int64_t)sl * T * ldt;
  f64x4 acc[4] = {};
  bool any = false;
  for (int e = row_ptr[k] + (int)blockIdx.z; e < row_ptr[k + 1]; e += nsplit) {
    const int j = row_j[e];
    if (j < first) continue;                      // Yt_j is zero in this slab
    any = true;
    __syncthreads();
    stage_block_ld(A, Yslab + (int64_t)j * T, ldt);
    stage_tile(B, tile_ptr(const_cast<double*>(S), nt, k, j));
    __syncthreads();
    tile_abt_mfma(A, B, acc);
  }
  double* Yk = Yslab + (int64_t)k * T;
  if (nsplit > 1) {
    if (!any) return;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) unsafeAtomicAdd(&Yk[(int64_t)(16 * rt + (lane >> 4) + 4 * r) * ldt + 16 * wv + (lane & 15)], -acc[rt][r]);
    return;
  }
  // T = E_k - sum: from the accumulator layout straight into the A operand of the product with L_kk^-T
  __syncthreads();
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * rt + (lane >> 4) + 4 * r, col = 16 * wv + (lane & 15);
      A[row * LDM + col] = Yk[(int64_t)row * ldt + col] - acc[rt][r];
    }
  stage_tile(B, Linv_all + (int64_t)k * (T * T));   // L^-1 is stored with an explicit zero upper part
  __syncthreads();
  f64x4 out[4] = {};
  tile_abt_mfma(A, B, out);
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) Yk[(int64_t)(16 * rt + (lane >> 4) + 4 * r) * ldt + 16 * wv + (lane & 15)] = out[rt][r];
}
__global__ void __launch_bounds__(kThreads) k_forward_multi_diag(int nt, const int32_t* __restrict__ lvl_k, const double* __restrict__ Linv_all, double* Yt, int64_t ldt,
                                                                const int32_t* __restrict__ slab_first) {
  __shared__ double smem[2 * T * LDM];
  double* A = smem;
  double* B = smem + T * LDM;
  const int k = lvl_k[blockIdx.x], sl = blockIdx.y;
  if (k < slab_first[sl]) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double* Yk = Yt + (int64_t)sl * T * ldt + (int64_t)k * T;
  stage_block_ld(A, Yk, ldt);
  stage_tile(B, Linv_all + (int64_t)k * (T * T));
  __syncthreads();
  f64x4 out[4] = {};
  tile_abt_mfma(A, B, out);
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) Yk[(int64_t)(16 * rt + (lane >> 4) + 4 * r) * ldt + 16 * wv + (lane & 15)] = out[rt][r];
}

// unit right-hand sides: right-hand side od ov + a has its one in row obj_row[ov] + a (od = 7, or 9 for the unconstrained ellipsoid block)
__global__ void __launch_bounds__(64) k_cov_seed(double* Yt, int64_t ldt, const int32_t* __restrict__ obj_row, int32_t nOv, int od) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= od * nOv) return;
  Yt[(int64_t)t * ldt + obj_row[t / od] + t % od] = 1.0;
}

// cov[p][r][k] = Yt[ca + r] . Yt[cb + k] (rows of Yt from first_row on); one workgroup per pair; cols[2p] < 0: zero block
constexpr int kCovThreads = 256;
template <int OD>
__global__ void __launch_bounds__(kCovThreads) k_cov_pairs(const double* __restrict__ Yt, int64_t ldt, const int32_t* __restrict__ cols,
                                                          const int32_t* __restrict__ first_row, double* __restrict__ out) {
  __shared__ double red[kCovThreads / 64][OD * OD];
  const int p = blockIdx.x, ca = cols[2 * p], cb = cols[2 * p + 1];
  double acc[OD * OD];
#pragma unroll
  for (int i = 0; i < OD * OD; ++i) acc[i] = 0.0;
  if (ca >= 0 && cb >= 0) {
    const double* ya = Yt + (int64_t)ca * ldt;
    const double* yb = Yt + (int64_t)cb * ldt;
    for (int64_t x = first_row[p] + threadIdx.x; x < ldt; x += kCovThreads) {
      double b[OD];
#pragma unroll
      for (int k = 0; k < OD; ++k) b[k] = yb[k * ldt + x];
#pragma unroll
      for (int r = 0; r < OD; ++r) {
        const double a = ya[r * ldt + x];
#pragma unroll
        for (int k = 0; k < OD; ++k) acc[OD * r + k] = fma(a, b[k], acc[OD * r + k]);
      }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < OD * OD; ++i) {
    double v = acc[i];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wv][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < OD * OD) {
    double v = 0.0;
    for (int w = 0; w < kCovThreads / 64; ++w) v += red[w][threadIdx.x];
    out[OD * OD * (int64_t)p + threadIdx.x] = v;
  }
}

}  // namespace

void launch_zero_tiles(hipStream_t s, double* S, int32_t nt, const int32_t* tile_list, int32_t ntiles, const uint8_t* is_pad_row, const StepClear& c) {
  const int extra = (int)std::max<int64_t>(c.pub_host ? 1 : 0, std::min<int64_t>(64, (c.n_max + kThreads - 1) / kThreads));
  if (ntiles + extra > 0) hipLaunchKernelGGL(k_zero_tiles, dim3(ntiles + extra), dim3(kThreads), 0, s, S, nt, tile_list, ntiles, is_pad_row, c, extra);
}

static void tick(hipStream_t s, CholTimers* t, int tag) {
  if (!t) return;
  if ((size_t)t->used >= t->pool->size()) { hipEvent_t e; (void)hipEventCreate(&e); t->pool->push_back(e); t->tags->push_back(-1); }
  (*t->tags)[t->used] = tag;
  (void)hipEventRecord((*t->pool)[t->used], s);
  t->used++;
}

// Factorisation of the levels [l0, l1) (forward phase), and the backward substitution over all levels.
// The multi-GPU exchange sits between the levels of a rank's own blocks and the levels of the shared tail.
void launch_cholesky_factor(hipStream_t s, const CholPlan& p, int l0, int l1, double* S, double* Linv, double* rhs, double* scal, CholTimers* timers) {
  const int nt = p.nt;
  if (l0 >= l1) return;
  tick(s, timers, -1);
  // the first level of the range has nothing to wait for inside a launch
  hipLaunchKernelGGL(k_potrf, dim3(p.lvl_k_ptr[l0 + 1] - p.lvl_k_ptr[l0]), dim3(512), 0, s, S, nt, p.lvl_k + p.lvl_k_ptr[l0], Linv, rhs, scal);
  tick(s, timers, CK_POTRF);
  for (int l = l0; l < l1; ++l) {
    const int ntr = p.trsm_ptr[l + 1] - p.trsm_ptr[l];
    const int sl = p.slices[l];   // 4 on thin levels: a tile product is shared by four workgroups
    if (ntr > 0) { hipLaunchKernelGGL(k_trsm, dim3(ntr * sl), dim3(kThreads), 0, s, S, nt, p.trsm_ik + 2 * (int64_t)p.trsm_ptr[l], Linv, sl); tick(s, timers, CK_TRSM); }
    const int nup = p.upd_ptr[l + 1] - p.upd_ptr[l], nrh = p.rh_ptr[l + 1] - p.rh_ptr[l];
    const int npk = l + 1 < l1 ? p.lvl_k_ptr[l + 2] - p.lvl_k_ptr[l + 1] : 0;
    UpdateJobs u{nup, p.upd_ij + 2 * (int64_t)p.upd_ptr[l], p.upd_kptr + p.upd_ptr[l], p.upd_k, p.upd_flag + p.upd_ptr[l],
                 p.rh_i + p.rh_ptr[l], p.rh_kptr + p.rh_ptr[l], p.rh_k};
    const bool fused = p.fused_potrf != 0;   // 0: two launches per level, nothing waits inside a launch (handle state: OBVI_FUSED_POTRF=0, or after a wait time-out)
    if (npk > 0 && !fused) {
      if (nup + nrh > 0) { hipLaunchKernelGGL(k_update, dim3(sl * nup + nrh), dim3(kThreads), 0, s, S, nt, u, rhs, sl); tick(s, timers, CK_UPDATE); }
      hipLaunchKernelGGL(k_potrf_pre, dim3(npk), dim3(512), 0, s, S, nt, p.lvl_k + p.lvl_k_ptr[l + 1], p.pre_ptr + p.lvl_k_ptr[l + 1], p.pre_j, Linv, rhs, scal);
      tick(s, timers, CK_POTRF);
    } else if (npk > 0) {
      hipLaunchKernelGGL(k_update_potrf, dim3(sl * nup + nrh + npk), dim3(512), 0, s, S, nt, u, nrh, p.crit_upd[l], p.crit_rh[l], npk, sl,
                         p.job_signal + p.upd_ptr[l] + p.rh_ptr[l], p.lvl_k + p.lvl_k_ptr[l + 1], p.k_need + p.lvl_k_ptr[l + 1], p.pre_ptr + p.lvl_k_ptr[l + 1], p.pre_j, p.diag_done, Linv, rhs, scal);
      tick(s, timers, CK_UPDATE);
    } else if (nup + nrh > 0) {
      hipLaunchKernelGGL(k_update, dim3(sl * nup + nrh), dim3(kThreads), 0, s, S, nt, u, rhs, sl);
      tick(s, timers, CK_UPDATE);
    }
  }
}
void launch_cholesky_backward(hipStream_t s, const CholPlan& p, const double* S, const double* Linv, double* rhs, double* y, CholTimers* timers) {
  const int nt = p.nt;
  tick(s, timers, -1);
  if (p.deterministic) {
    for (int l = p.nlevels - 1; l >= 0; --l) {
      const int nk = p.lvl_k_ptr[l + 1] - p.lvl_k_ptr[l];
      if (nk > 0) hipLaunchKernelGGL(k_backward_det, dim3(nk), dim3(kThreads), 0, s, S, nt, p.lvl_k + p.lvl_k_ptr[l], p.col_ptr, p.col_i, Linv, rhs, y);
      tick(s, timers, CK_BACKWARD);
    }
    return;
  }
  for (int l = p.nbw - 1; l >= 0; --l) {
    const int nwg = p.bw_ptr[l + 1] - p.bw_ptr[l];
    if (nwg > 0) hipLaunchKernelGGL(k_backward, dim3(nwg), dim3(kBackThreads), 0, s, S, nt, p.bw_kj + 3 * (int64_t)p.bw_ptr[l], p.bw_chains, Linv, rhs, y);
    tick(s, timers, CK_BACKWARD);
  }
}

void launch_forward_multi(hipStream_t s, const CholPlan& p, const double* S, const double* Linv, double* Yt, int64_t ldt, int nslabs,
                          const int32_t* slab_first, const int32_t* obj_row, int32_t nOv, const int32_t* row_split, int od) {
  if (nOv <= 0 || nslabs <= 0) return;
  hipLaunchKernelGGL(k_cov_seed, dim3((od * nOv + 63) / 64), dim3(64), 0, s, Yt, ldt, obj_row, nOv, od);
  for (int l = 0; l < p.nlevels; ++l) {
    const int nk = p.lvl_k_ptr[l + 1] - p.lvl_k_ptr[l];
    if (nk <= 0) continue;
    const int nsplit = row_split ? row_split[l] : 1;
    hipLaunchKernelGGL(k_forward_multi, dim3(nk, nslabs, nsplit), dim3(kThreads), 0, s, S, p.nt, p.lvl_k + p.lvl_k_ptr[l], p.row_ptr, p.row_j, Linv, Yt, ldt, slab_first, nsplit);
    if (nsplit > 1) hipLaunchKernelGGL(k_forward_multi_diag, dim3(nk, nslabs), dim3(kThreads), 0, s, p.nt, p.lvl_k + p.lvl_k_ptr[l], Linv, Yt, ldt, slab_first);
  }
}
void launch_cov_pairs(hipStream_t s, const double* Yt, int64_t ldt, int64_t n_pairs, const int32_t* cols, const int32_t* first_row, double* out, int od) {
  if (n_pairs <= 0) return;
  if (od == 9) hipLaunchKernelGGL(k_cov_pairs<9>, dim3((unsigned)n_pairs), dim3(kCovThreads), 0, s, Yt, ldt, cols, first_row, out);
  else hipLaunchKernelGGL(k_cov_pairs<7>, dim3((unsigned)n_pairs), dim3(kCovThreads), 0, s, Yt, ldt, cols, first_row, out);
}

}  // namespace obvi
