// ba_kernels.hip -- gfx950 kernels of the LM step outside the reduced-system factorisation:
//   K0 pose_cache        per-pose R^T, -R^T t, SO(3) right Jacobian
//   K1 point_pass        reprojection linearisation, one thread per eliminated point: Hll, g_l, cost,
//                        pose-side J^T J / J^T r, per-point 3x3 Cholesky of (Hll + lambda), Z = rho' Jp^T Jl C^-T
//   K2/K3 small factors  bbox / shape prior / LTM prior / relative pose (forward-mode duals)
//   K4 schur_blocks      S(p,q) -= sum_l Z_pl Z_ql^T, rhs_p -= sum_l Z_pl u_l  -- one wavefront per 6x6 block
//   K6 point_backsub     y_l = C^-T (u - sum Z^T y_p), candidate points
//   K7 cost              trial-point cost + model cost change
//   K9 apply step        candidate poses / objects
// Reference arithmetic: see ba_math.h.  Solver algebra: [Ceres-doc] SchurEliminator / LM strategy.
#include <algorithm>

#include <cstdlib>
#include <stdexcept>
#include "ba_device.h"

namespace obvi {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }

// Wavefront reductions on the data-parallel-primitive path of the vector ALU (a lane reads a neighbour's register as part of a
// v_mov: row_shr inside the rows of 16 lanes, then the two row broadcasts) instead of six rounds of ds_bpermute through the LDS crossbar:
// 12 DPP moves of 4 cycles for a double instead of 12 LDS permutes.  The total ends up in lane 63 and is handed to every lane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double v, double identity) {   // lanes without a source (or of a masked row) get `identity`
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(identity), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(identity), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double v) {   // a quad permute of a double (all four lanes of a quad are sources: no identity needed)
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ double wave_last_lane(double v) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_take<0x111, 0xf>(v, 0.0);   // row_shr:1
  v += dpp_take<0x112, 0xf>(v, 0.0);   // row_shr:2
  v += dpp_take<0x114, 0xf>(v, 0.0);   // row_shr:4
  v += dpp_take<0x118, 0xf>(v, 0.0);   // row_shr:8   -> lane 15 of a row: the row's sum
  v += dpp_take<0x142, 0xa>(v, 0.0);   // row_bcast:15 into rows 1 and 3
  v += dpp_take<0x143, 0xc>(v, 0.0);   // row_bcast:31 into rows 2 and 3  -> lane 63: the wavefront's sum
  return wave_last_lane(v);
}
__device__ __forceinline__ double wave_max(double v) {
  const double ninf = -__builtin_huge_val();
  v = fmax(v, dpp_take<0x111, 0xf>(v, ninf));
  v = fmax(v, dpp_take<0x112, 0xf>(v, ninf));
  v = fmax(v, dpp_take<0x114, 0xf>(v, ninf));
  v = fmax(v, dpp_take<0x118, 0xf>(v, ninf));
  v = fmax(v, dpp_take<0x142, 0xa>(v, ninf));
  v = fmax(v, dpp_take<0x143, 0xc>(v, ninf));
  return wave_last_lane(v);
}
// a workgroup's total for scalar `sc`, one thread per workgroup: an fp64 atomic, or -- deterministic mode, ba_device.h -- the
// workgroup's slot of the partial sums behind the scalar block (every workgroup of the grid must get here: the slots are not cleared).
// `det` is BlocksDev.deterministic: 0, or the number of workgroups each slot has room for
__device__ __forceinline__ void scal_add(double* scal, int det, int sc, double t) {
  if (det) scal[SC_COUNT + (int64_t)det_slot_of(sc) * det + blockIdx.x] = t;
  else if (t != 0.0) atomic_add_f64(scal + sc, t);
}
// block-wide sum -> one atomic (or one partial-sum slot) per block
__device__ __forceinline__ void block_accumulate(double v, double* scal, int sc, int det = 0) {
  __shared__ double sm[kBlock / 64];
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < kBlock / 64; ++i) t += sm[i];
    scal_add(scal, det, sc, t);
  }
}
__device__ __forceinline__ void block_accumulate_max(double v, double* dst_bits) {
  __shared__ double smx[kBlock / 64];
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smx[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < kBlock / 64; ++i) t = fmax(t, smx[i]);
    if (t > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(dst_bits), (unsigned long long)__double_as_longlong(t));
  }
}

// Z storage: the 18-double record of observation a of point l sits at 18 a + 4 l, and behind a point's last record come
// 4 doubles (u_l, 0): one contiguous, 16-byte aligned block per point = everything k_schur_window needs of it.
__device__ __forceinline__ int64_t z_off(int64_t a, int64_t l) { return 18 * a + 4 * l; }
__device__ __forceinline__ int64_t z_tail(int64_t end_obs, int64_t l) { return 18 * end_obs + 4 * l; }
__device__ __forceinline__ double* S_at(double* S, int32_t nt, int64_t i, int64_t j) {
  return S + ((i / kTile) * (int64_t)nt + (j / kTile)) * (kTile * kTile) + (i % kTile) * kTile + (j % kTile);
}

// 1/sqrt(p): hardware v_rsq_f64 seed + two Newton steps (<= 1-2 ulp for normal positive p; NaN/inf for p <= 0, which the caller flags)
__device__ __forceinline__ double rsqrt_f64(double p) {
  double y = __builtin_amdgcn_rsq(p);
  double e = fma(-p * y, y, 1.0);
  y = fma(y * 0.5, e, y);
  e = fma(-p * y, y, 1.0);
  y = fma(y * 0.5, e, y);
  return y;
}
// block-wide sums of 4 values and maximum of a 5th with one barrier pair -> one atomic each per block
// (deterministic mode: the three sums go to the partial-sum slots; the failure count and the maximum are exact in any order)
__device__ __forceinline__ void block_accumulate5(double* scal, int det, double v0, int s0, double v1, int s1, double v2, int s2, double v3, double* d3, double vmax, double* dmax_bits) {
  __shared__ double sm5[kBlock / 64][5];
  v0 = wave_sum(v0); v1 = wave_sum(v1); v2 = wave_sum(v2); v3 = wave_sum(v3); vmax = wave_max(vmax);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { sm5[w][0] = v0; sm5[w][1] = v1; sm5[w][2] = v2; sm5[w][3] = v3; sm5[w][4] = vmax; }
  __syncthreads();
  if (threadIdx.x < 5) {
    double t = 0.0;
    for (int i = 0; i < kBlock / 64; ++i) t = threadIdx.x < 4 ? t + sm5[i][threadIdx.x] : fmax(t, sm5[i][4]);
    if (threadIdx.x < 3) scal_add(scal, det, threadIdx.x == 0 ? s0 : threadIdx.x == 1 ? s1 : s2, t);
    else if (threadIdx.x == 3) { if (t != 0.0) atomic_add_f64(d3, t); }
    else if (t > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(dmax_bits), (unsigned long long)__double_as_longlong(t));
  }
}

__device__ __forceinline__ double lm_lambda(double colsq, double scale, double radius) {
  // LevenbergMarquardtStrategy::ComputeStep [Ceres-doc]: D^2 = clamp(diag(Js^T Js), 1e-6, 1e32) / radius on
  // the Jacobi-scaled Jacobian Js = J diag(scale); in unscaled variables the damping is D^2 / scale^2.
  const double d = fmin(fmax(colsq * scale * scale, 1e-6), 1e32);
  return d / radius / (scale * scale);
}

// ---------------------------------------------------------------------------------------

// obvi_ba_set_reproj: camera / pixel / sigma of every observation in the two orders the kernels stream them in (launch_reproj_gather)
__global__ void __launch_bounds__(kBlock) k_reproj_gather(int64_t n, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ rq_src, const uint32_t* __restrict__ rp_point,
                                                         const uint16_t* __restrict__ raw_cam, const double2* __restrict__ raw_pixel, const double* __restrict__ raw_sigma,
                                                         double sigma_scalar, uint16_t* __restrict__ cam, double2* __restrict__ pixel, double* __restrict__ sigma,
                                                         uint32_t* __restrict__ q_point, uint16_t* __restrict__ q_cam, double2* __restrict__ q_pixel, double* __restrict__ q_sigma,
                                                         uint8_t* __restrict__ q_active) {
  const int64_t a = blockIdx.x * (int64_t)kBlock + threadIdx.x;
  if (a >= n) return;
  const uint32_t b = rq_src[a];
  const uint32_t i = perm[a], j = perm[b];
  cam[a] = raw_cam ? raw_cam[i] : (uint16_t)0; pixel[a] = raw_pixel[i]; sigma[a] = raw_sigma ? raw_sigma[i] : sigma_scalar;
  q_point[a] = rp_point[b]; q_cam[a] = raw_cam ? raw_cam[j] : (uint16_t)0; q_pixel[a] = raw_pixel[j]; q_sigma[a] = raw_sigma ? raw_sigma[j] : sigma_scalar; q_active[a] = 1;
}
__global__ void __launch_bounds__(kBlock) k_pose_cache(int64_t P, const double* __restrict__ poses, PoseCache* __restrict__ out, int analytic) {
  const int64_t p = blockIdx.x * (int64_t)kBlock + threadIdx.x;
  if (p >= P) return;
  double pose[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) pose[k] = poses[6 * p + k];
  PoseCache pc;
  make_pose_cache(pose, &pc, analytic != 0);
  out[p] = pc;
  // second copy, field-major, behind the records (out must hold 2 (P + 1) records): k_point_pass gathers the cache per observation,
  // and the lanes of a point's run look at consecutive poses -- field-major makes those loads coalesce
  double* soa = reinterpret_cast<double*>(out + P + 1);
  const double* f = reinterpret_cast<const double*>(&pc);
#pragma unroll
  for (int k = 0; k < 21; ++k) soa[k * P + p] = f[k];
}

// ---------------------------------------------------------------------------------------
// K1 (long tracks).  One thread per point, for the points with more observations than a wavefront has lanes
// (list built by the host; usually empty).  Same arithmetic as k_point_pass below.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_point_pass_long(BlocksDev b, ReprojDev rp, const DevCam* __restrict__ cams,
                                                           const PoseCache* __restrict__ pc, const double* __restrict__ points,
                                                           ReducedDev rd, PointDev pt, double radius, int first_iter, double* scal,
                                                           const uint32_t* __restrict__ long_points, int64_t n_long) {
  const int64_t idx = blockIdx.x * (int64_t)kBlock + threadIdx.x;
  double cost = 0.0, gsq = 0.0, gmax = 0.0, xsq = 0.0, fail = 0.0;
  if (idx < n_long) {
    const int64_t l = long_points[idx];
    const uint32_t beg = rp.point_ptr[l], end = rp.point_ptr[l + 1];
    const bool lvar = b.point_var[l] != 0;
    const double X[3] = {points[3 * l], points[3 * l + 1], points[3 * l + 2]};
    double h00 = 0, h10 = 0, h11 = 0, h20 = 0, h21 = 0, h22 = 0, g0 = 0, g1 = 0, g2 = 0;
    for (uint32_t a = beg; a < end; ++a) {
      if (!rp.active[a]) continue;
      const uint32_t p = rp.pose[a];
      const int32_t vid = b.pose_vid[p];
      if (vid < 0 && !lvar) continue;  // all-constant residual block: fixed cost, not part of the reduced program
      const double2 px = rp.pixel[a];
      double r[2], Jp[12], Jl[6];
      reproj_eval<true>(pc[p], cams[rp.cam[a]], X, px.x, px.y, rp.sigma[a], r, Jp, Jl);
      double rho0, w;
      huber_eval(r[0] * r[0] + r[1] * r[1], rp.huber, &rho0, &w);
      cost += 0.5 * rho0;
      if (lvar) {
        h00 += w * (Jl[0] * Jl[0] + Jl[3] * Jl[3]); h10 += w * (Jl[1] * Jl[0] + Jl[4] * Jl[3]); h11 += w * (Jl[1] * Jl[1] + Jl[4] * Jl[4]);
        h20 += w * (Jl[2] * Jl[0] + Jl[5] * Jl[3]); h21 += w * (Jl[2] * Jl[1] + Jl[5] * Jl[4]); h22 += w * (Jl[2] * Jl[2] + Jl[5] * Jl[5]);
        g0 += w * (Jl[0] * r[0] + Jl[3] * r[1]); g1 += w * (Jl[1] * r[0] + Jl[4] * r[1]); g2 += w * (Jl[2] * r[0] + Jl[5] * r[1]);
      }
    }
    if (lvar) {
      xsq = X[0] * X[0] + X[1] * X[1] + X[2] * X[2];
      gsq = g0 * g0 + g1 * g1 + g2 * g2;
      gmax = fmax(fabs(g0), fmax(fabs(g1), fabs(g2)));
      double s0, s1, s2;
      if (first_iter) {
        s0 = 1.0 / (1.0 + sqrt(h00)); s1 = 1.0 / (1.0 + sqrt(h11)); s2 = 1.0 / (1.0 + sqrt(h22));
        pt.scale[3 * l] = s0; pt.scale[3 * l + 1] = s1; pt.scale[3 * l + 2] = s2;
      } else {
        s0 = pt.scale[3 * l]; s1 = pt.scale[3 * l + 1]; s2 = pt.scale[3 * l + 2];
      }
      double lam0 = lm_lambda(h00, s0, radius), lam1 = lm_lambda(h11, s1, radius), lam2 = lm_lambda(h22, s2, radius);
      if (pt.extra) { lam0 += pt.extra[3 * l]; lam1 += pt.extra[3 * l + 1]; lam2 += pt.extra[3 * l + 2]; }
      const double a00 = h00 + lam0, a11 = h11 + lam1, a22 = h22 + lam2;
      pt.gl[3 * l] = g0; pt.gl[3 * l + 1] = g1; pt.gl[3 * l + 2] = g2; pt.lam[3 * l] = lam0; pt.lam[3 * l + 1] = lam1; pt.lam[3 * l + 2] = lam2;
      // 3x3 Cholesky A = C C^T and Ci = C^-1
      const double c00 = sqrt(a00), c10 = h10 / c00, c20 = h20 / c00;
      const double d11 = a11 - c10 * c10;
      const double c11 = sqrt(d11), c21 = (h21 - c20 * c10) / c11;
      const double d22 = a22 - c20 * c20 - c21 * c21;
      const double c22 = sqrt(d22);
      if (!(a00 > 0.0) || !(d11 > 0.0) || !(d22 > 0.0)) fail = 1.0;
      const double i00 = 1.0 / c00, i11 = 1.0 / c11, i22 = 1.0 / c22;
      const double i10 = -c10 * i00 * i11, i21 = -c21 * i11 * i22, i20 = -(c20 * i00 + c21 * i10) * i22;
      double* Ci = pt.Ci + 6 * l;
      Ci[0] = i00; Ci[1] = i10; Ci[2] = i11; Ci[3] = i20; Ci[4] = i21; Ci[5] = i22;
      const double ul0 = i00 * g0, ul1 = i10 * g0 + i11 * g1, ul2 = i20 * g0 + i21 * g1 + i22 * g2;
      pt.u[3 * l] = ul0; pt.u[3 * l + 1] = ul1; pt.u[3 * l + 2] = ul2;
      { double* ut = pt.Z + z_tail(end, l); ut[0] = ul0; ut[1] = ul1; ut[2] = ul2; ut[3] = 0.0; }   // copy behind the point's Z records (k_schur_window)
      // second sweep: Z = rho' Jp^T (Jl Ci^T)
      for (uint32_t a = beg; a < end; ++a) {
        const uint32_t p = rp.pose[a];
        if (!rp.active[a] || b.pose_vid[p] < 0) {   // no contribution: a zero record (a plan kept across a mask change may still visit it)
          double* Z0 = pt.Z + z_off(a, l);
          for (int x = 0; x < 18; ++x) Z0[x] = 0.0;
          continue;
        }
        const double2 px = rp.pixel[a];
        double r[2], Jp[12], Jl[6];
        reproj_eval<true>(pc[p], cams[rp.cam[a]], X, px.x, px.y, rp.sigma[a], r, Jp, Jl);
        double rho0, w;
        huber_eval(r[0] * r[0] + r[1] * r[1], rp.huber, &rho0, &w);
        const double m00 = Jl[0] * i00, m01 = Jl[0] * i10 + Jl[1] * i11, m02 = Jl[0] * i20 + Jl[1] * i21 + Jl[2] * i22;
        const double m10 = Jl[3] * i00, m11 = Jl[3] * i10 + Jl[4] * i11, m12 = Jl[3] * i20 + Jl[4] * i21 + Jl[5] * i22;
        double* Z = pt.Z + z_off(a, l);
#pragma unroll
        for (int x = 0; x < 6; ++x) {
          Z[3 * x] = w * (Jp[x] * m00 + Jp[6 + x] * m10);
          Z[3 * x + 1] = w * (Jp[x] * m01 + Jp[6 + x] * m11);
          Z[3 * x + 2] = w * (Jp[x] * m02 + Jp[6 + x] * m12);
        }
      }
    } else {   // not variable: zero records and tail (see k_point_pass)
      for (uint32_t a = beg; a < end; ++a) { double* Z = pt.Z + z_off(a, l); for (int x = 0; x < 18; ++x) Z[x] = 0.0; }
      double* ut = pt.Z + z_tail(end, l); ut[0] = ut[1] = ut[2] = ut[3] = 0.0;
      pt.u[3 * l] = 0.0; pt.u[3 * l + 1] = 0.0; pt.u[3 * l + 2] = 0.0;
    }
  }
  block_accumulate(cost, scal, SC_COST, b.deterministic);
  block_accumulate(gsq, scal, SC_GSQ, b.deterministic);
  block_accumulate(xsq, scal, SC_XSQ, b.deterministic);
  block_accumulate(fail, scal, SC_CHOL_FAIL);
  block_accumulate_max(gmax, scal + SC_GMAX_BITS);
}

// ---------------------------------------------------------------------------------------
// K1.  Point side of the reprojection linearisation, one lane per observation.  The observations are stored by point
// (CSC), so the lanes of a wavefront read 64 consecutive records (coalesced) and the observations of a point are a
// run of consecutive lanes; the host cuts the observation list into wavefront-sized pieces at point boundaries
// (first observation and count per piece, at most 64, whole points).  Per lane: residual + closed-form Jacobians + Huber once
// (the one-thread-per-point form evaluated them twice).  Per point: H_ll = sum rho' Jl^T Jl, g_l = sum rho' Jl^T r
// by a segmented scan over the run's lanes (shuffles); every lane forms the
// 3x3 Cholesky of H_ll + lambda and its own Z = rho' Jp^T Jl C^-T.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock, kPointImageDoubles <= 1264 ? 4 : 3) k_point_pass(BlocksDev b, ReprojDev rp, const DevCam* __restrict__ cams,
                                                      const PoseCache* __restrict__ pc, const double* __restrict__ points,
                                                      ReducedDev rd, PointDev pt, double radius, int first_iter, double* scal,
                                                      const uint32_t* __restrict__ wave_obs, int64_t n_waves) {
  // per wavefront: the image of the wavefront's part of the Z storage (records + tails, 18 n + 4 points <= kPointImageDoubles: the host cuts
  // the pieces accordingly), written out with coalesced 16-byte stores
  __shared__ __attribute__((aligned(16))) double ex[kBlock / 64][kPointImageDoubles];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t gw = blockIdx.x * (int64_t)(kBlock / 64) + wv;
  double cost = 0.0, gsq = 0.0, gmax = 0.0, xsq = 0.0, fail = 0.0;
  if (gw < n_waves) {
    const uint32_t a0 = wave_obs[2 * gw], n = wave_obs[2 * gw + 1];
    const uint32_t l0 = rp.point[a0], l1 = rp.point[a0 + n - 1];   // first and last point of the piece (host: l1 - l0 < 64)
    const bool have = (uint32_t)lane < n;
    const uint32_t a = a0 + (have ? (uint32_t)lane : 0u);
    const uint32_t l = rp.point[a], p = rp.pose[a];
    const uint32_t beg = rp.point_ptr[l], end = rp.point_ptr[l + 1];
    const bool lvar = b.point_var[l] != 0;
    const int32_t vid = b.pose_vid[p];
    const bool live = have && rp.active[a] && (vid >= 0 || lvar);   // else: inactive, or an all-constant residual block (fixed cost)
    const double X[3] = {points[3 * (int64_t)l], points[3 * (int64_t)l + 1], points[3 * (int64_t)l + 2]};
    double r[2] = {0.0, 0.0}, Jp[12], Jl[6], w = 0.0;
#pragma unroll
    for (int i = 0; i < 12; ++i) Jp[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) Jl[i] = 0.0;
    if (live) {
      const double2 px = rp.pixel[a];
      PoseCache cache;   // from the field-major copy (k_pose_cache)
      {
        const double* soa = reinterpret_cast<const double*>(pc + b.P + 1) + p;
        double* f = reinterpret_cast<double*>(&cache);
#pragma unroll
        for (int k = 0; k < 21; ++k) f[k] = soa[k * b.P];
      }
      reproj_eval<true>(cache, cams[rp.cam[a]], X, px.x, px.y, rp.sigma[a], r, Jp, Jl);
      double rho0;
      huber_eval(r[0] * r[0] + r[1] * r[1], rp.huber, &rho0, &w);
      cost = 0.5 * rho0;
    }
    const bool sum = live && lvar;
    // per point: sums over its run of lanes (segmented inclusive scan, then everybody reads the run's last lane)
    const int first = (int)(beg - a0), len = (int)(end - beg);
    const bool head = have && a == beg;
    double t[9];
    t[0] = sum ? w * (Jl[0] * Jl[0] + Jl[3] * Jl[3]) : 0.0;
    t[1] = sum ? w * (Jl[1] * Jl[0] + Jl[4] * Jl[3]) : 0.0;
    t[2] = sum ? w * (Jl[1] * Jl[1] + Jl[4] * Jl[4]) : 0.0;
    t[3] = sum ? w * (Jl[2] * Jl[0] + Jl[5] * Jl[3]) : 0.0;
    t[4] = sum ? w * (Jl[2] * Jl[1] + Jl[5] * Jl[4]) : 0.0;
    t[5] = sum ? w * (Jl[2] * Jl[2] + Jl[5] * Jl[5]) : 0.0;
    t[6] = sum ? w * (Jl[0] * r[0] + Jl[3] * r[1]) : 0.0;
    t[7] = sum ? w * (Jl[1] * r[0] + Jl[4] * r[1]) : 0.0;
    t[8] = sum ? w * (Jl[2] * r[0] + Jl[5] * r[1]) : 0.0;
    const int seg0 = have ? first : lane;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const bool take = lane - d >= seg0;
#pragma unroll
      for (int q = 0; q < 9; ++q) { const double up = __shfl_up(t[q], d, 64); t[q] += take ? up : 0.0; }
    }
    const int last = have ? min(first + len - 1, 63) : lane;
#pragma unroll
    for (int q = 0; q < 9; ++q) t[q] = __shfl(t[q], last, 64);
    double* img = &ex[wv][0];          // img[18 (a - a0) + 4 (l - l0)] <-> pt.Z[z_off(a, l)]
    if (have && lvar) {
      const double h00 = t[0], h10 = t[1], h11 = t[2], h20 = t[3], h21 = t[4], h22 = t[5], g0 = t[6], g1 = t[7], g2 = t[8];
      double s0, s1, s2;
      if (first_iter) { s0 = 1.0 / (1.0 + sqrt(h00)); s1 = 1.0 / (1.0 + sqrt(h11)); s2 = 1.0 / (1.0 + sqrt(h22)); }
      else { s0 = pt.scale[3 * (int64_t)l]; s1 = pt.scale[3 * (int64_t)l + 1]; s2 = pt.scale[3 * (int64_t)l + 2]; }
      double lam0 = lm_lambda(h00, s0, radius), lam1 = lm_lambda(h11, s1, radius), lam2 = lm_lambda(h22, s2, radius);
      if (pt.extra) { lam0 += pt.extra[3 * (int64_t)l]; lam1 += pt.extra[3 * (int64_t)l + 1]; lam2 += pt.extra[3 * (int64_t)l + 2]; }
      const double a00 = h00 + lam0, a11 = h11 + lam1, a22 = h22 + lam2;
      // 3x3 Cholesky A = C C^T and Ci = C^-1, division-free: i_kk = rsqrt(pivot)
      const double i00 = rsqrt_f64(a00), c10 = h10 * i00, c20 = h20 * i00;
      const double d11 = a11 - c10 * c10;
      const double i11 = rsqrt_f64(d11), c21 = (h21 - c20 * c10) * i11;
      const double d22 = a22 - c20 * c20 - c21 * c21;
      const double i22 = rsqrt_f64(d22);
      const double i10 = -c10 * i00 * i11, i21 = -c21 * i11 * i22, i20 = -(c20 * i00 + c21 * i10) * i22;
      if (head) {
        if (!(a00 > 0.0) || !(d11 > 0.0) || !(d22 > 0.0)) fail = 1.0;
        xsq = X[0] * X[0] + X[1] * X[1] + X[2] * X[2];
        gsq = g0 * g0 + g1 * g1 + g2 * g2;
        gmax = fmax(fabs(g0), fmax(fabs(g1), fabs(g2)));
        if (first_iter) { pt.scale[3 * (int64_t)l] = s0; pt.scale[3 * (int64_t)l + 1] = s1; pt.scale[3 * (int64_t)l + 2] = s2; }
        pt.gl[3 * (int64_t)l] = g0; pt.gl[3 * (int64_t)l + 1] = g1; pt.gl[3 * (int64_t)l + 2] = g2;
        pt.lam[3 * (int64_t)l] = lam0; pt.lam[3 * (int64_t)l + 1] = lam1; pt.lam[3 * (int64_t)l + 2] = lam2;
        double* Ci = pt.Ci + 6 * (int64_t)l;
        Ci[0] = i00; Ci[1] = i10; Ci[2] = i11; Ci[3] = i20; Ci[4] = i21; Ci[5] = i22;
        const double ul0 = i00 * g0, ul1 = i10 * g0 + i11 * g1, ul2 = i20 * g0 + i21 * g1 + i22 * g2;
        pt.u[3 * (int64_t)l] = ul0; pt.u[3 * (int64_t)l + 1] = ul1; pt.u[3 * (int64_t)l + 2] = ul2;
        double* ut = img + 18 * (first + len) + 4 * (int)(l - l0);   // (u_l, 0) behind the point's Z records (k_schur_window)
        ut[0] = ul0; ut[1] = ul1; ut[2] = ul2; ut[3] = 0.0;
      }
      // Z = rho' Jp^T (Jl Ci^T); w = 0 for an observation without a Z record (inactive / constant pose): its slot is never read
      const double wz = (live && vid >= 0) ? w : 0.0;
      const double m00 = Jl[0] * i00, m01 = Jl[0] * i10 + Jl[1] * i11, m02 = Jl[0] * i20 + Jl[1] * i21 + Jl[2] * i22;
      const double m10 = Jl[3] * i00, m11 = Jl[3] * i10 + Jl[4] * i11, m12 = Jl[3] * i20 + Jl[4] * i21 + Jl[5] * i22;
      double* Z = img + 18 * lane + 4 * (int)(l - l0);
#pragma unroll
      for (int x = 0; x < 6; ++x) {
        Z[3 * x] = wz * (Jp[x] * m00 + Jp[6 + x] * m10);
        Z[3 * x + 1] = wz * (Jp[x] * m01 + Jp[6 + x] * m11);
        Z[3 * x + 2] = wz * (Jp[x] * m02 + Jp[6 + x] * m12);
      }
    } else if (have) {
      // a point that is not variable (constant, or -- under a plan kept across a mask change -- one that lost all its factors): zero
      // records and a zero tail, because the plan's Schur work lists may still visit it
      double* Z = img + 18 * lane + 4 * (int)(l - l0);
#pragma unroll
      for (int x = 0; x < 18; ++x) Z[x] = 0.0;
      if (head) {
        double* ut = img + 18 * (first + len) + 4 * (int)(l - l0); ut[0] = ut[1] = ut[2] = ut[3] = 0.0;
        pt.u[3 * (int64_t)l] = 0.0; pt.u[3 * (int64_t)l + 1] = 0.0; pt.u[3 * (int64_t)l + 2] = 0.0;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // write the image: contiguous in global memory, 16 bytes per lane per store
    const int total2 = (18 * (int)n + 4 * (int)(l1 - l0 + 1)) / 2;
    double2* dst = reinterpret_cast<double2*>(pt.Z + z_off(a0, l0));
    const double2* src = reinterpret_cast<const double2*>(img);
    for (int i = lane; i < total2; i += 64) dst[i] = src[i];
  }
  block_accumulate5(scal, b.deterministic, cost, SC_COST, gsq, SC_GSQ, xsq, SC_XSQ, fail, scal + SC_CHOL_FAIL, gmax, scal + SC_GMAX_BITS);
}

// ---------------------------------------------------------------------------------------
// K1b.  Pose side of the reprojection linearisation: one workgroup per pose, its observations
// contiguous (CSR by pose copy of the observation arrays).  Each thread accumulates the 21 unique
// entries of rho' Jp^T Jp and the 6 of rho' Jp^T r over its observations; one deterministic
// wavefront/LDS reduction per pose, no atomics.
// ---------------------------------------------------------------------------------------
template <bool PIPE>
__global__ void __launch_bounds__(kBlock) k_pose_pass(BlocksDev b, ReprojPoseDev rq, const DevCam* __restrict__ cams, const PoseCache* __restrict__ pc,
                                                     const double* __restrict__ points, ReducedDev rd, int slices) {
  // slices > 1 (a sliding window: tens of poses with a thousand sightings each): `slices` workgroups share a pose, so that a thread has one
  // or two sightings instead of a chain of dependent gathers, and add their sums atomically
  const int64_t p = blockIdx.x / slices;
  const int sl = blockIdx.x % slices;
  const int32_t vid = b.pose_vid[p];
  if (vid < 0) return;   // uniform per workgroup
  __shared__ double red[kBlock / 64][27];
  const PoseCache cache = pc[p];
  double acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.0;
  const uint32_t beg = rq.pose_ptr[p], end = rq.pose_ptr[p + 1];
  if (PIPE) {
    // (loads in two rounds, the next sighting's first round in flight during this one's arithmetic: a window's pose pass is one sighting per thread and three dependent loads deep otherwise)
    struct ByK { uint32_t l; double2 px; double sg; uint16_t cam; uint8_t act; };
    auto by_k = [&](uint32_t k) { ByK o; o.act = rq.active[k]; o.l = rq.point[k]; o.px = rq.pixel[k]; o.cam = rq.cam[k]; o.sg = rq.sigma[k]; return o; };
    uint32_t k = beg + sl * kBlock + threadIdx.x;
    const uint32_t kstep = kBlock * slices;
    ByK cur = {};
    if (k < end) cur = by_k(k);
    for (; k < end; k += kstep) {
      ByK nxt = {};
      if (k + kstep < end) nxt = by_k(k + kstep);
      const ByK o = cur;
      cur = nxt;
      const uint32_t l = o.l;
      const double X[3] = {points[3 * (int64_t)l], points[3 * (int64_t)l + 1], points[3 * (int64_t)l + 2]};
      double r[2], Jp[12], Jl[6];
      reproj_eval<true>(cache, cams[o.cam], X, o.px.x, o.px.y, o.sg, r, Jp, Jl);
      double rho0, w;
      huber_eval(r[0] * r[0] + r[1] * r[1], rq.huber, &rho0, &w);
      if (o.act) {   // (a masked sighting is evaluated -- its loads were in flight anyway -- and not added: its Jacobian may be non-finite)
        int e = 0;
  #pragma unroll
        for (int x = 0; x < 6; ++x) {
  #pragma unroll
          for (int y = 0; y <= x; ++y) acc[e++] += w * (Jp[x] * Jp[y] + Jp[6 + x] * Jp[6 + y]);
        }
  #pragma unroll
        for (int x = 0; x < 6; ++x) acc[21 + x] += w * (Jp[x] * r[0] + Jp[6 + x] * r[1]);
      }
    }
  } else {
    for (uint32_t k = beg + sl * kBlock + threadIdx.x; k < end; k += kBlock * slices) {
      if (!rq.active[k]) continue;
      const uint32_t l = rq.point[k];
      const double X[3] = {points[3 * (int64_t)l], points[3 * (int64_t)l + 1], points[3 * (int64_t)l + 2]};
      const double2 px = rq.pixel[k];
      double r[2], Jp[12], Jl[6];
      reproj_eval<true>(cache, cams[rq.cam[k]], X, px.x, px.y, rq.sigma[k], r, Jp, Jl);
      double rho0, w;
      huber_eval(r[0] * r[0] + r[1] * r[1], rq.huber, &rho0, &w);
      int e = 0;
  #pragma unroll
      for (int x = 0; x < 6; ++x) {
  #pragma unroll
        for (int y = 0; y <= x; ++y) acc[e++] += w * (Jp[x] * Jp[y] + Jp[6 + x] * Jp[6 + y]);
      }
  #pragma unroll
      for (int x = 0; x < 6; ++x) acc[21 + x] += w * (Jp[x] * r[0] + Jp[6 + x] * r[1]);
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) red[wv][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    double t = 0.0;
    for (int i = 0; i < kBlock / 64; ++i) t += red[i][threadIdx.x];
    const int k = threadIdx.x;
    if (k < 21) {
      // packed lower-triangular index -> (x, y)
      int x = 0, base = 0;
      while (base + x + 1 <= k) { base += x + 1; ++x; }
      const int y = k - base;
      if (slices > 1) atomic_add_f64(&rd.Hdiag[36 * (int64_t)vid + 6 * x + y], t); else rd.Hdiag[36 * (int64_t)vid + 6 * x + y] += t;
    } else {
      if (slices > 1) atomic_add_f64(&rd.g[6 * (int64_t)vid + (k - 21)], t); else rd.g[6 * (int64_t)vid + (k - 21)] += t;
    }
  }
}

// ---------------------------------------------------------------------------------------
// K2/K3.  Small factor families (k_small_lin_lanes below): 16 lanes per bounding-box / relative-pose factor, a lane per prior.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void add_diag_block(double* Hd, double* gd, int d, const double* J, const double* r, int m, double w) {
  for (int x = 0; x < d; ++x) {
    for (int y = 0; y <= x; ++y) {
      double acc = 0.0;
      for (int a = 0; a < m; ++a) acc += J[d * a + x] * J[d * a + y];
      atomic_add_f64(Hd + d * x + y, w * acc);
    }
    double acc = 0.0;
    for (int a = 0; a < m; ++a) acc += J[d * a + x] * r[a];
    atomic_add_f64(gd + x, w * acc);
  }
}

// deterministic mode: the same block, lower-packed (H: d (d + 1) / 2 entries, then g: d) into the factor's scratch slot, no atomics
__device__ __forceinline__ void store_diag_block(double* slot, int d, const double* J, const double* r, int m, double w) {
  int e = 0;
  for (int x = 0; x < d; ++x)
    for (int y = 0; y <= x; ++y) {
      double acc = 0.0;
      for (int a = 0; a < m; ++a) acc += J[d * a + x] * J[d * a + y];
      slot[e++] = w * acc;
    }
  for (int x = 0; x < d; ++x) {
    double acc = 0.0;
    for (int a = 0; a < m; ++a) acc += J[d * a + x] * r[a];
    slot[e++] = w * acc;
  }
}
constexpr int kSmBlk = 62, kSmSecond = 35;   // scratch slot of a prior / relative-pose factor: first block at 0 (an object's: 35 entries, 54 for the 9-parameter block), a relative-pose factor's second block at 35

__device__ __forceinline__ void object_priors_lin(int64_t block, const BlocksDev& b, const SmallFactorsDev& sf, const double* __restrict__ objects, const ReducedDev& rd, double* scal) {
  const int64_t t = block * 64LL + threadIdx.x;
  const int od = b.od;   // 7, or 9 for the unconstrained ellipsoid block
  double cost = 0.0;
  bool stored = false;
  if (t < sf.n_sp) {
    const int64_t i = t;
    if (sf.sp_active[i]) {
      const uint32_t o = sf.sp_obj[i];
      const int32_t ov = b.obj_vid[o];
      if (ov >= 0) {
        double r[3], J[27];
        shape_prior_eval(objects + od * (int64_t)o, sf.sp_mean + 3 * i, sf.sp_sqrt_inf + 9 * i, r, J, od);
        double rho0, w;
        huber_eval(r[0] * r[0] + r[1] * r[1] + r[2] * r[2], sf.sp_huber, &rho0, &w);
        cost = 0.5 * rho0;
        if (b.deterministic) { store_diag_block(sf.sm_blk + (int64_t)kSmBlk * t, od, J, r, 3, w); stored = true; }
        else add_diag_block(rd.Hdiag + 36 * b.nPv + od * od * (int64_t)ov, rd.g + 6 * b.nPv + od * (int64_t)ov, od, J, r, 3, w);
      }
    }
  } else if (t < sf.n_sp + sf.n_lt) {
    const int64_t i = t - sf.n_sp;
    if (sf.lt_active[i]) {
      const uint32_t o = sf.lt_obj[i];
      const int32_t ov = b.obj_vid[o];
      if (ov >= 0) {
        double r[9], J[81];
        ltm_prior_eval(objects + od * (int64_t)o, sf.lt_mean + od * i, sf.lt_sqrt_inf + od * od * i, r, J, od);
        double s = 0.0;
        for (int a = 0; a < od; ++a) s += r[a] * r[a];
        double rho0, w;
        huber_eval(s, sf.lt_huber, &rho0, &w);
        cost = 0.5 * rho0;
        if (b.deterministic) { store_diag_block(sf.sm_blk + (int64_t)kSmBlk * t, od, J, r, od, w); stored = true; }
        else add_diag_block(rd.Hdiag + 36 * b.nPv + od * od * (int64_t)ov, rd.g + 6 * b.nPv + od * (int64_t)ov, od, J, r, od, w);
      }
    }
  }
  if (b.deterministic && !stored && t < sf.n_sp + sf.n_lt) {   // an inactive factor / a constant object: a zero block (k_small_gather reads every slot of its lists)
    double* slot = sf.sm_blk + (int64_t)kSmBlk * t;
    for (int e = 0; e < od * (od + 1) / 2 + od; ++e) slot[e] = 0.0;
  }
  cost = wave_sum(cost);
  if (threadIdx.x == 0) scal_add(scal, b.deterministic, SC_COST, cost);
}

// Small problems (a sliding window holds a few hundred of these factors): the thread-per-factor kernels above are then a handful of
// wavefronts whose single-lane dual arithmetic (13 or 12 directions) is the latency of the whole side stream.  Here a factor takes
// 16 lanes: lane `dir` evaluates the residual with a one-direction dual, so the Jacobian column of a parameter lives in its lane;
// rows of J^T J are formed from the group's columns (wave shuffles) and added by the lane that owns the row.  (On big problems the
// thread-per-factor kernels win: see DESIGN.md.)
// STORE (big problems): nothing is added atomically except the cost -- the factor's blocks (w J^T J, w J^T r) go to its slot of the
// scratch (kBbBlk doubles: H_oo lower-packed 28 | g_o 7 | H_pp lower-packed 21 | g_p 6), and k_bbox_gather sums the slots per
// object and per pose; the off-diagonal 7x6 block goes straight into its tile.  Tens of thousands of factors would otherwise mean millions of fp64 atomics (32-byte memory-side
// transactions each): that, not the dual arithmetic, was the duration of the thread-per-factor kernel this replaces.
// (9-parameter ellipsoid block: H_oo 45 | g_o 9 | H_pp 21 | g_p 6 = 81 doubles per slot, 15 of the 16 lanes carry a direction)
template <int OD> struct BbSlot { static constexpr int kHoo = 0, kGo = OD * (OD + 1) / 2, kHpp = kGo + OD, kGp = kHpp + 21, kBlk = kGp + 6; };
constexpr int kBbBlk = BbSlot<7>::kBlk, kBbGo = BbSlot<7>::kGo, kBbHpp = BbSlot<7>::kHpp, kBbGp = BbSlot<7>::kGp;
static_assert(kBbBlk == 62 && kBbGo == 28 && kBbHpp == 35 && kBbGp == 56 && BbSlot<9>::kBlk == 81, "slot layout");
template <bool STORE, int OD>
__device__ __forceinline__ void bbox_lin_lanes(int64_t block, const BlocksDev& b, const SmallFactorsDev& sf, const DevCam* __restrict__ cams, const double* __restrict__ poses,
                                               const double* __restrict__ objects, const ReducedDev& rd, double* scal) {
  const int64_t i = block * 4LL + (threadIdx.x >> 4);
  const int dir = threadIdx.x & 15, base = threadIdx.x & 48;
  uint32_t o = 0, p = 0;
  int32_t ov = -1, pv = -1;
  if (i < sf.n_bb && sf.bb_active[i]) { o = sf.bb_obj[i]; p = sf.bb_pose[i]; ov = b.obj_vid[o]; pv = b.pose_vid[p]; }
  const bool work = ov >= 0 || pv >= 0;
  double r[4] = {0.0, 0.0, 0.0, 0.0}, J[4] = {0.0, 0.0, 0.0, 0.0}, cost = 0.0, w = 0.0;
  if (work) {
    Dual<1> res[4];
    bbox_eval_n<1, OD>(objects + OD * (int64_t)o, poses + 6 * (int64_t)p, cams[sf.bb_cam[i]], sf.bb_rect + 4 * i, sf.bb_sqrt_inf + 16 * i, sf.bb_invalid, res, dir);
    for (int a = 0; a < 4; ++a) { r[a] = res[a].v; J[a] = res[a].d[0]; }
    double rho0;
    huber_eval(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3], sf.bb_huber, &rho0, &w);
    if (dir == 0) cost = 0.5 * rho0;
  }
  double Jk[4][OD + 6];                                   // the factor's whole Jacobian, column k from lane base + k
#pragma unroll
  for (int k = 0; k < OD + 6; ++k)
#pragma unroll
    for (int a = 0; a < 4; ++a) Jk[a][k] = __shfl(J[a], base + k, 64);
  if (STORE) {
    typedef BbSlot<OD> SL;
    double* slot = sf.bb_blk + (int64_t)SL::kBlk * i;
    if (work && dir < OD && ov >= 0) {
      const int x = dir;
#pragma unroll
      for (int y = 0; y < OD; ++y) {
        if (y > x) continue;
        double acc = 0.0;
        for (int a = 0; a < 4; ++a) acc += J[a] * Jk[a][y];
        slot[SL::kHoo + x * (x + 1) / 2 + y] = w * acc;
      }
      double acc = 0.0;
      for (int a = 0; a < 4; ++a) acc += J[a] * r[a];
      slot[SL::kGo + x] = w * acc;
      if (pv >= 0) {
        // the 7x6 block between the object and the pose belongs to this factor alone when no (object, pose) pair occurs twice
        // (host: bb_pairs_unique; false only with several cameras seeing the object from one frame): plain stores into the cleared tile
        const int64_t orow = b.obj_row[ov], prow = b.pose_row[pv];
        const bool obj_low = orow > prow;
#pragma unroll
        for (int y = 0; y < 6; ++y) {
          double acc2 = 0.0;
          for (int a = 0; a < 4; ++a) acc2 += J[a] * Jk[a][OD + y];
          double* dst = obj_low ? S_at(rd.S, rd.nt, orow + x, prow + y) : S_at(rd.S, rd.nt, prow + y, orow + x);
          if (sf.bb_pairs_unique) *dst = w * acc2; else atomic_add_f64(dst, w * acc2);
        }
      }
    } else if (work && dir >= OD && dir < OD + 6 && pv >= 0) {
      const int x = dir - OD;
#pragma unroll
      for (int y = 0; y < 6; ++y) {
        if (y > x) continue;
        double acc = 0.0;
        for (int a = 0; a < 4; ++a) acc += J[a] * Jk[a][OD + y];
        slot[SL::kHpp + x * (x + 1) / 2 + y] = w * acc;
      }
      double acc = 0.0;
      for (int a = 0; a < 4; ++a) acc += J[a] * r[a];
      slot[SL::kGp + x] = w * acc;
    }
  } else if (work && dir < OD && ov >= 0) {
    const int x = dir;
    double* Hd = rd.Hdiag + 36 * b.nPv + OD * OD * (int64_t)ov;
#pragma unroll
    for (int y = 0; y < OD; ++y) {
      if (y > x) continue;
      double acc = 0.0;
      for (int a = 0; a < 4; ++a) acc += J[a] * Jk[a][y];
      atomic_add_f64(Hd + OD * x + y, w * acc);
    }
    double acc = 0.0;
    for (int a = 0; a < 4; ++a) acc += J[a] * r[a];
    atomic_add_f64(rd.g + 6 * b.nPv + OD * (int64_t)ov + x, w * acc);
    if (pv >= 0) {
      const int64_t orow = b.obj_row[ov], prow = b.pose_row[pv];
      const bool obj_low = orow > prow;
#pragma unroll
      for (int y = 0; y < 6; ++y) {
        double acc2 = 0.0;
        for (int a = 0; a < 4; ++a) acc2 += J[a] * Jk[a][OD + y];
        atomic_add_f64(obj_low ? S_at(rd.S, rd.nt, orow + x, prow + y) : S_at(rd.S, rd.nt, prow + y, orow + x), w * acc2);
      }
    }
  } else if (work && dir >= OD && dir < OD + 6 && pv >= 0) {
    const int x = dir - OD;
    double* Hd = rd.Hdiag + 36 * (int64_t)pv;
#pragma unroll
    for (int y = 0; y < 6; ++y) {
      if (y > x) continue;
      double acc = 0.0;
      for (int a = 0; a < 4; ++a) acc += J[a] * Jk[a][OD + y];
      atomic_add_f64(Hd + 6 * x + y, w * acc);
    }
    double acc = 0.0;
    for (int a = 0; a < 4; ++a) acc += J[a] * r[a];
    atomic_add_f64(rd.g + 6 * (int64_t)pv + x, w * acc);
  }
  cost = wave_sum(cost);
  if (threadIdx.x == 0) scal_add(scal, b.deterministic, SC_COST, cost);
}
// lanes 0..5 own the columns of the first pose, 6..11 of the second
__device__ __forceinline__ void relpose_lin_lanes(int64_t block, const BlocksDev& b, const SmallFactorsDev& sf, const double* __restrict__ poses, const ReducedDev& rd, double* scal) {
  const int64_t i = block * 4LL + (threadIdx.x >> 4);
  const int dir = threadIdx.x & 15, base = threadIdx.x & 48;
  uint32_t pa = 0, pb = 0;
  int32_t va = -1, vb = -1;
  if (i < sf.n_rl && sf.rl_active[i]) { pa = sf.rl_a[i]; pb = sf.rl_b[i]; va = b.pose_vid[pa]; vb = b.pose_vid[pb]; }
  const bool work = va >= 0 || vb >= 0;
  double r[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, J[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, cost = 0.0, w = 0.0;
  if (work) {
    Dual<1> res[6];
    relpose_eval_n<1>(poses + 6 * (int64_t)pa, poses + 6 * (int64_t)pb, sf.rl_t + 3 * i, sf.rl_R + 9 * i, sf.rl_sqrt_inf + 36 * i, res, dir);
    double s = 0.0;
    for (int a = 0; a < 6; ++a) { r[a] = res[a].v; J[a] = res[a].d[0]; s += r[a] * r[a]; }
    double rho0;
    huber_eval(s, sf.rl_huber, &rho0, &w);
    if (dir == 0) cost = 0.5 * rho0;
  }
  double Jk[6][12];
#pragma unroll
  for (int k = 0; k < 12; ++k)
#pragma unroll
    for (int a = 0; a < 6; ++a) Jk[a][k] = __shfl(J[a], base + k, 64);
  const bool first = dir < 6;
  const int32_t vid = first ? va : vb;
  if (b.deterministic && i < sf.n_rl && dir < 12) {
    // row x of the lane's diagonal block into the factor's scratch slot (lower-packed 21 | g 6; second pose at kSmSecond); zeros when the
    // factor is inactive or that pose is constant: k_small_gather reads every slot of its lists
    const int x = first ? dir : dir - 6;
    double* slot = sf.sm_blk + (int64_t)kSmBlk * (sf.n_sp + sf.n_lt + i) + (first ? 0 : kSmSecond);
    const double on = (work && vid >= 0) ? w : 0.0;
#pragma unroll
    for (int y = 0; y < 6; ++y) {
      if (y > x) continue;
      double acc = 0.0;
      for (int a = 0; a < 6; ++a) acc += J[a] * (first ? Jk[a][y] : Jk[a][6 + y]);
      slot[x * (x + 1) / 2 + y] = on * acc;
    }
    double acc = 0.0;
    for (int a = 0; a < 6; ++a) acc += J[a] * r[a];
    slot[21 + x] = on * acc;
  }
  if (work && dir < 12 && vid >= 0) {
    const int x = first ? dir : dir - 6;
    if (!b.deterministic) {
      double* Hd = rd.Hdiag + 36 * (int64_t)vid;
#pragma unroll
      for (int y = 0; y < 6; ++y) {
        if (y > x) continue;
        double acc = 0.0;
        for (int a = 0; a < 6; ++a) acc += J[a] * (first ? Jk[a][y] : Jk[a][6 + y]);
        atomic_add_f64(Hd + 6 * x + y, w * acc);
      }
      double acc = 0.0;
      for (int a = 0; a < 6; ++a) acc += J[a] * r[a];
      atomic_add_f64(rd.g + 6 * (int64_t)vid + x, w * acc);
    }
    if (va >= 0 && vb >= 0 && va != vb) {
      const int64_t ra = b.pose_row[va], rb = b.pose_row[vb];
      const bool b_low = rb > ra;  // lower triangle: the later-eliminated block is the row, and its lanes add the block
      if (b_low != first) {
        const int64_t row = b_low ? rb : ra, col = b_low ? ra : rb;
#pragma unroll
        for (int y = 0; y < 6; ++y) {
          double acc2 = 0.0;
          for (int a = 0; a < 6; ++a) acc2 += J[a] * (first ? Jk[a][6 + y] : Jk[a][y]);
          atomic_add_f64(S_at(rd.S, rd.nt, row + x, col + y), w * acc2);
        }
      }
    }
  }
  cost = wave_sum(cost);
  if (threadIdx.x == 0) scal_add(scal, b.deterministic, SC_COST, cost);
}

// the three small-factor families of a small problem in one launch (at this size an iteration's first half is bound by the host's launches)
template <bool STORE, int OD>
__global__ void __launch_bounds__(64) k_small_lin_lanes(BlocksDev b, SmallFactorsDev sf, const DevCam* __restrict__ cams, const double* __restrict__ poses,
                                                       const double* __restrict__ objects, ReducedDev rd, double* scal, int nb_bbox, int nb_priors) {
  const int blk = blockIdx.x;
  if (blk < nb_bbox) bbox_lin_lanes<STORE, OD>(blk, b, sf, cams, poses, objects, rd, scal);
  else if (blk < nb_bbox + nb_priors) object_priors_lin(blk - nb_bbox, b, sf, objects, rd, scal);
  else relpose_lin_lanes(blk - nb_bbox - nb_priors, b, sf, poses, rd, scal);
}

// The sums behind k_small_lin_lanes<true>: the diagonal blocks.  Workgroups [0, O): one per object, its factors' H_oo | g_o (CSR by
// object; an object of the global problem has ~100 factors, so the list is cut into 7 slices of 36 lanes and the loop is unrolled:
// the loads of a slot go through two indirections).  The rest: one wavefront per pose, H_pp | g_p of its ~10 factors (CSR by pose).
// One writer per block: plain read-modify-write, a fixed summation order.  Runs after the kernels that add to the diagonal blocks
// atomically (same stream).
__global__ void __launch_bounds__(kBlock) k_bbox_gather(BlocksDev b, SmallFactorsDev sf, ReducedDev rd) {
  const int od = b.od, nho = od * (od + 1) / 2, ne = nho + od;                          // 7: 28 + 7 = 35 entries per slot;  9: 45 + 9 = 54
  const int blk = nho + od + 27, hpp = nho + od;                                         // slot size and the offset of H_pp | g_p (BbSlot<OD>)
  if ((int64_t)blockIdx.x < b.O) {
    __shared__ double part[7][64];
    const int64_t o = blockIdx.x;
    const int32_t ov = b.obj_vid[o];
    if (ov < 0) return;   // uniform per workgroup
    const int width = od == 7 ? 36 : 64, nslice = od == 7 ? 7 : 4;                      // lanes per slice x slices <= 256
    const int slice = threadIdx.x / width, e = threadIdx.x % width;
    const uint32_t q0 = sf.bbo_ptr[o], q1 = sf.bbo_ptr[o + 1];
    if (slice < nslice) {
      double acc = 0.0;
      if (e < ne) {
#pragma unroll 4
        for (uint32_t q = q0 + slice; q < q1; q += nslice) {
          const uint32_t f = sf.bbo_idx[q];
          const double v = sf.bb_blk[(int64_t)blk * f + e];
          acc += sf.bb_active[f] ? v : 0.0;
        }
      }
      part[slice][e] = acc;
    }
    __syncthreads();
    if ((int)threadIdx.x < ne) {
      const int lane = threadIdx.x;
      double acc = 0.0;
      for (int i = 0; i < nslice; ++i) acc += part[i][lane];
      if (lane < nho) {
        int x = 0, base = 0;
        while (base + x + 1 <= lane) { base += x + 1; ++x; }
        rd.Hdiag[36 * b.nPv + od * od * (int64_t)ov + od * x + (lane - base)] += acc;
      } else {
        rd.g[6 * b.nPv + od * (int64_t)ov + (lane - nho)] += acc;
      }
    }
    return;
  }
  const int lane = threadIdx.x & 63;
  const int64_t p = ((int64_t)blockIdx.x - b.O) * (kBlock / 64) + (threadIdx.x >> 6);
  if (p >= b.P) return;
  const int32_t pv = b.pose_vid[p];
  if (pv < 0 || lane >= 27) return;
  const uint32_t q0 = sf.bbp_ptr[p], q1 = sf.bbp_ptr[p + 1];
  if (q1 == q0) return;
  double acc = 0.0;
#pragma unroll 4
  for (uint32_t q = q0; q < q1; ++q) {
    const uint32_t f = sf.bbp_idx[q];
    const double v = sf.bb_blk[(int64_t)blk * f + hpp + lane];
    acc += sf.bb_active[f] ? v : 0.0;
  }
  if (lane < 21) {
    int x = 0, base = 0;
    while (base + x + 1 <= lane) { base += x + 1; ++x; }
    rd.Hdiag[36 * (int64_t)pv + 6 * x + (lane - base)] += acc;
  } else {
    rd.g[6 * (int64_t)pv + (lane - 21)] += acc;
  }
}

// deterministic mode: the sums behind the priors' and relative-pose factors' scratch slots.  One wavefront per target block (objects,
// then poses), lane e = entry e of the lower-packed block | gradient; the target's list is walked in order: a fixed summation order,
// one writer per block (plain read-modify-write behind the kernels that ran before on the stream).
__global__ void __launch_bounds__(kBlock) k_small_gather(BlocksDev b, SmallFactorsDev sf, ReducedDev rd) {
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (t >= b.O + b.P) return;
  const bool is_obj = t < b.O;
  const int32_t vid = is_obj ? b.obj_vid[t] : b.pose_vid[t - b.O];
  const int d = is_obj ? b.od : 6, nh = d * (d + 1) / 2;
  if (vid < 0 || lane >= nh + d) return;
  const uint32_t q0 = sf.smt_ptr[t], q1 = sf.smt_ptr[t + 1];
  if (q1 == q0) return;
  double acc = 0.0;
  for (uint32_t q = q0; q < q1; ++q) {
    const uint32_t e = sf.smt_idx[q];
    acc += sf.sm_blk[(int64_t)kSmBlk * (e >> 1) + ((e & 1u) ? kSmSecond : 0) + lane];
  }
  double* Hd = is_obj ? rd.Hdiag + 36 * b.nPv + d * d * (int64_t)vid : rd.Hdiag + 36 * (int64_t)vid;
  double* gd = is_obj ? rd.g + 6 * b.nPv + d * (int64_t)vid : rd.g + 6 * (int64_t)vid;
  if (lane < nh) {
    int x = 0, base = 0;
    while (base + x + 1 <= lane) { base += x + 1; ++x; }
    Hd[d * x + (lane - base)] += acc;
  } else {
    gd[lane - nh] += acc;
  }
}

// diagonal blocks of the reduced system: scaling, damping, gradient norms, |x|^2
__global__ void __launch_bounds__(kBlock) k_reduced_diag(BlocksDev b, const double* __restrict__ poses, const double* __restrict__ objects,
                                                        ReducedDev rd, double radius, int first_iter, double* scal) {
  // 8 threads per block (16 with the 9-parameter ellipsoid block): thread k handles row k of the block
  const int sh = b.od > 8 ? 4 : 3;
  const int64_t t = (blockIdx.x * (int64_t)kBlock + threadIdx.x) >> sh;
  const int k = threadIdx.x & ((1 << sh) - 1);
  double gsq = 0.0, gmax = 0.0, xsq = 0.0;
  const int64_t nblk = b.P + b.O;
  if (t < nblk) {
    const bool is_pose = t < b.P;
    const int64_t idx = is_pose ? t : t - b.P;
    const int32_t vid = is_pose ? b.pose_vid[idx] : b.obj_vid[idx];
    const int d = is_pose ? 6 : b.od;
    if (vid >= 0 && k < d) {
      const int64_t crow = is_pose ? 6 * (int64_t)vid : 6 * b.nPv + d * (int64_t)vid;   // compact index (g, scale, lam)
      const int64_t row = is_pose ? b.pose_row[vid] : b.obj_row[vid];                     // row of the tile grid (S, rhs, y)
      const double* Hd = is_pose ? rd.Hdiag + 36 * (int64_t)vid : rd.Hdiag + 36 * b.nPv + d * d * (int64_t)vid;
      const double* x = is_pose ? poses + 6 * idx : objects + d * idx;
      // a shared object's (already globally summed) diagonal block, gradient and norms are contributed by one rank only
      const bool contribute = is_pose || b.obj_shared == nullptr || !b.obj_shared[vid] || b.shared_owner;
      const double c = Hd[d * k + k];
      double s;
      if (first_iter) { s = 1.0 / (1.0 + sqrt(c)); rd.scale[crow + k] = s; } else { s = rd.scale[crow + k]; }
      const double lam = lm_lambda(c, s, radius) + (rd.extra ? rd.extra[crow + k] : 0.0);
      rd.lam[crow + k] = lam;
      if (contribute) {
        // atomics: the Schur complement kernels subtract from the same tiles and right-hand side, possibly at the same time (side stream)
        for (int y = 0; y <= k; ++y) atomic_add_f64(S_at(rd.S, rd.nt, row + k, row + y), Hd[d * k + y] + (y == k ? lam : 0.0));
        const double g = rd.g[crow + k];
        atomic_add_f64(rd.rhs + row + k, g);
        gsq = g * g; gmax = fabs(g); xsq = x[k] * x[k];
      }
    }
  }
  block_accumulate(gsq, scal, SC_GSQ, b.deterministic);
  block_accumulate(xsq, scal, SC_XSQ, b.deterministic);
  block_accumulate_max(gmax, scal + SC_GMAX_BITS);
}

// ---------------------------------------------------------------------------------------
// K4.  One wavefront per 6x6 block (p,q) of the Schur complement; the block's pair list
// (observation a of pose p, observation b of pose q, same point) is contiguous.
// lanes 0..35: element (x,y); lanes 36..41 on diagonal blocks: rhs component x.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_schur_blocks(int64_t nblk, const uint32_t* __restrict__ blk_row, const uint32_t* __restrict__ blk_col,
                                                        const uint32_t* __restrict__ blk_ptr, const uint32_t* __restrict__ pair_a,
                                                        const uint32_t* __restrict__ pair_b, const uint32_t* __restrict__ obs_point,
                                                        PointDev pt, ReducedDev rd) {
  // One workgroup per 6x6 block.  Per pass the 4 wavefronts stage 16 pairs each: the two 6x3 Z
  // blocks of a pair (36 doubles) are fetched with coalesced 8-byte loads into LDS, then lanes
  // 0..35 of each wavefront accumulate element (x,y); lanes 36..41 the rhs on diagonal blocks.
  constexpr int kPairsPerWave = 16;
  __shared__ double zsh[kBlock / 64][kPairsPerWave * 36];
  __shared__ double ush[kBlock / 64][kPairsPerWave * 3];
  __shared__ double red[kBlock / 64][42];
  // XCD-aware mapping: workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md); give each XCD a contiguous range of
  // blocks (sorted by row then column) so the Z records of a pose are re-read from that XCD's L2.
  const int64_t chunk = (nblk + 7) / 8;
  const int64_t blk = (blockIdx.x % 8) * chunk + blockIdx.x / 8;
  if (blk >= nblk) return;   // uniform per workgroup
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t row = blk_row[blk], col = blk_col[blk];
  const uint32_t beg = blk_ptr[blk], end = blk_ptr[blk + 1];
  const bool diag = row == col;
  const int x = lane < 36 ? lane / 6 : lane - 36, y = lane % 6;
  double acc = 0.0;
  for (uint32_t base = beg; base < end; base += (kBlock / 64) * kPairsPerWave) {
    const uint32_t wbeg = base + wv * kPairsPerWave;
    // pair indices of this wavefront: lanes 0..15 -> a, 16..31 -> b
    uint32_t idx = 0xffffffffu;
    if (lane < 32) {
      const uint32_t k = wbeg + (lane & 15);
      if (k < end) idx = lane < 16 ? pair_a[k] : pair_b[k];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int f = i * 64 + lane;             // [0, 576): pair = f / 36, half = (f % 36) / 18, e = f % 18
      const int pr = f / 36, rem = f - 36 * pr, half = rem >= 18 ? 1 : 0, e = rem - 18 * half;
      const uint32_t src = __shfl(idx, pr + 16 * half, 64);
      zsh[wv][f] = (src != 0xffffffffu) ? pt.Z[z_off(src, obs_point[src]) + e] : 0.0;
    }
    if (diag && lane < 48) {
      const int pr = lane / 3, e = lane - 3 * pr;
      const uint32_t sa = __shfl(idx, pr, 64), sb = __shfl(idx, pr + 16, 64);
      ush[wv][lane] = (sa != 0xffffffffu && sa == sb) ? pt.u[3 * (int64_t)obs_point[sa] + e] : 0.0;
    }
    __syncthreads();
    if (lane < 36) {
#pragma unroll
      for (int pr = 0; pr < kPairsPerWave; ++pr) {
        const double* Za = &zsh[wv][36 * pr + 3 * x];
        const double* Zb = &zsh[wv][36 * pr + 18 + 3 * y];
        acc += Za[0] * Zb[0] + Za[1] * Zb[1] + Za[2] * Zb[2];
      }
    } else if (diag && lane < 42) {
#pragma unroll
      for (int pr = 0; pr < kPairsPerWave; ++pr) {
        const double* Za = &zsh[wv][36 * pr + 3 * x];
        const double* u = &ush[wv][3 * pr];
        acc += Za[0] * u[0] + Za[1] * u[1] + Za[2] * u[2];
      }
    }
    __syncthreads();
  }
  if (lane < 42) red[wv][lane] = acc;
  __syncthreads();
  if (threadIdx.x < 42) {
    const int t = threadIdx.x;
    double s = 0.0;
    for (int i = 0; i < kBlock / 64; ++i) s += red[i][t];
    if (t < 36) {
      const int xx = t / 6, yy = t % 6;
      if (!diag || yy <= xx) atomic_add_f64(S_at(rd.S, rd.nt, (int64_t)row + xx, (int64_t)col + yy), -s);   // a block may be spread over several work items
    } else if (diag) {
      atomic_add_f64(rd.rhs + row + (t - 36), -s);
    }
  }
}

// ---------------------------------------------------------------------------------------
// K4a.  Schur complement on the matrix cores, point-centric, no atomics inside the loop.
// Geometry.  The variable frames (trajectory order) are cut into row chunks of kSR = 8 frames = 48 rows = 3 MFMA row
// tiles.  The strip of a chunk is the 48 x 240 part of S whose columns are the kSFr = 40 frames [f0 - 32, f0 + 8):
// 3 x 15 tiles of v_mfma_f64_16x16x4_f64, cut into 3 column groups of kSGC = 5 tile columns.  A point is "visited" by
// every (chunk, group) in which it has a row frame and a column frame; with Zrow(i) the 3-vector of matrix row i
// (row 6 fo + x  <->  Z[obs at frame fo][x][0..2]) a visit adds
//     strip(i, n) += sum_k Zrow(i)[k] * Zrow(n)[k]          (K = 3, padded to the instruction's 4)
// for the tiles the point covers (wave-uniform bits of the host-built visit record).
// Parallelism.  A workgroup owns a slice of the visits of one (chunk, group); each of its 4 wavefronts is an
// independent stream with a private 3 x 5 tile accumulator in registers (120 VGPRs, two wavefronts per SIMD) and takes
// every 4th visit.  Nothing in the loop depends on another wavefront, and nothing in it touches global memory:
// Memory.  The host lays a visit out as consecutive 144-byte slots -- the point's Z record for each strip frame from
// its first to its last row frame and column frame of the group (zeros for a frame it skips), its (u_l, 0) tail, and
// for a stereo point the second record of each frame -- and cuts a workgroup's visits into batches of <= 32 KB.  A
// batch is one gather (slot table entry -> global_load_lds_dwordx4, all 256 lanes) issued while the previous batch is
// multiplied; the table entries of the batch after that ride in registers.  An operand is then a single ds_read_b64 at
// (slot + fo - first) * 144 + 8 (3 x + (l>>4)), lanes without an operand read a zero.
// The workgroup adds its tiles to the tile grid once, at the end: the four streams' accumulators are summed through LDS, then fp64
// hardware atomics (15 tiles per workgroup).
// Pairs whose frames are further apart than the strip (loop closures, very long tracks) go to k_schur_blocks.
// ---------------------------------------------------------------------------------------
constexpr int kSR = kSchurRows, kSFr = kSchurWindowFrames, kSBack = kSFr - kSR;
constexpr int kSTR = kSR * 6 / 16, kSTC = kSFr * 6 / 16, kSDiag = kSBack * 6 / 16, kSWv = 4, kSGC = kSchurGroupCols;
static_assert(kSR * 6 % 16 == 0 && kSFr * 6 % 16 == 0 && kSBack * 6 % 16 == 0 && kSFr <= 64 && kSTC % kSGC == 0, "strip must be whole MFMA tiles");
static_assert(kSchurBatchBytes % (16 * 64 * kSWv) == 0, "a batch is whole gather instructions");
typedef double sf64x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void schur_gptr;
typedef __attribute__((address_space(3))) void schur_lptr;
typedef __attribute__((address_space(3))) const unsigned char schur_lds8;
typedef __attribute__((address_space(3))) const double schur_ldsd;
// global -> LDS gather of 16 bytes per lane: lane l's bytes land at lds_base + 16 l (lds_base wave-uniform).  Issued as asm so
// that the compiler does not order every later LDS read of the *other* batch buffer behind it (it would wait vmcnt(0));
// the batch loop waits for it explicitly before its barrier.  M0 recipe: cdna_hip_programming.md 5.7.
__device__ __forceinline__ void gather16_to_lds(const void* gsrc, uint32_t lds_base) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
__device__ __forceinline__ uint32_t lds_address(const void* p) { return (uint32_t)(uintptr_t)(schur_lptr*)p; }
// Accumulating MFMA with the accumulator tied in place (VGPR form).  Written as asm because the builtin form of a
// *conditional* MFMA leaves the choice of C/D registers to the allocator across the join.
// Wait states (cdna_hip_programming.md 5.7 item 2): 2 after a VALU write of an operand (s_nop 1); none between MFMAs
// chained through C; the write-out below pads the MFMA -> VALU read itself.
__device__ __forceinline__ void mfma_f64_acc(sf64x4& acc, double a, double bv) {
  asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(bv));
}

#ifndef OBVI_SCHUR_VISIT_GROUP
#define OBVI_SCHUR_VISIT_GROUP 2
#endif
template <bool TWIN>
__global__ void __launch_bounds__(64 * kSWv) k_schur_window(BlocksDev b, PointDev pt, ReducedDev rd, const int32_t* __restrict__ row_of_nat,
                                                          const uint32_t* __restrict__ wg_bptr, const uint32_t* __restrict__ bfirst,
                                                          const uint32_t* __restrict__ bslot, const uint4* __restrict__ visits,
                                                          const uint32_t* __restrict__ slot_src, const int32_t* __restrict__ wg_f0,
                                                          const int32_t* __restrict__ wg_group) {
  // two batch buffers as separate objects, each addressed statically (the loop below is unrolled by two): the compiler then
  // knows that the reads of one do not alias the gather in flight into the other and does not wait for it
  __shared__ __attribute__((aligned(16))) unsigned char zbuf0[kSchurBatchBytes], zbuf1[kSchurBatchBytes];
  __shared__ __attribute__((aligned(16))) uint4 recbuf0[kSchurBatchVisits], recbuf1[kSchurBatchVisits];
  __shared__ __attribute__((aligned(16))) double zero2[2];
  __shared__ int32_t rown[kSFr];
  constexpr int kIters = kSchurBatchBytes / 16 / (64 * kSWv);   // gather instructions per lane per batch
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int32_t f0 = wg_f0[blockIdx.x], fbase = f0 - kSBack, cbase = kSGC * wg_group[blockIdx.x];
  const bool with_rhs = cbase + kSGC == kSTC;   // the group holding the chunk's own frames sees every visit of the chunk once
  if (tid < kSFr) { const int32_t f = fbase + tid; rown[tid] = (f >= 0 && f < b.nPv) ? row_of_nat[f] : -1; }
  if (tid < 2) zero2[tid] = 0.0;
  // per-lane operand addresses: an operand of row tile r / column tile cbase + c sits at (byte offset of strip frame 0 in the batch image:
  // visit-uniform, from the record) + 144 (frame offset of the lane's matrix row) + (offset in the record) -- the host lays a visit out so
  // that every frame an active tile touches has a slot (zeros where the point has no observation), hence no range test.  Lanes kq = 3 pad
  // K = 3 to the instruction's 4: multiplier 0 and the address of a zero.  One v_mad per operand; constants per batch buffer.
  const uint32_t kmul = kq < 3 ? 1u : 0u;
  uint32_t cA[2][kSTR], cB[2][kSGC];
#pragma unroll
  for (int bf = 0; bf < 2; ++bf) {
    const uint32_t img = lds_address(bf ? &zbuf1[0] : &zbuf0[0]), zr = lds_address(&zero2[0]);
#pragma unroll
    for (int r = 0; r < kSTR; ++r) { const int row = 16 * (kSDiag + r) + m; cA[bf][r] = kq < 3 ? img + 144u * (uint32_t)(row / 6) + 8u * (uint32_t)((row % 6) * 3 + kq) : zr; }
#pragma unroll
    for (int c = 0; c < kSGC; ++c) { const int col = 16 * (cbase + c) + m; cB[bf][c] = kq < 3 ? img + 144u * (uint32_t)(col / 6) + 8u * (uint32_t)((col % 6) * 3 + kq) : zr; }
  }
  sf64x4 acc[kSGC][kSTR];
#pragma unroll
  for (int c = 0; c < kSGC; ++c)
#pragma unroll
    for (int r = 0; r < kSTR; ++r) acc[c][r] = sf64x4{0.0, 0.0, 0.0, 0.0};
  double racc[kSTR] = {};   // partial of (Z u) for row 16 r + m, component kq

  // ---- batch gather: chunk q (16 bytes) of the batch image comes from slot_src[slot0 + q / 9] + q % 9
  uint32_t src_next[kIters];   // table entries of the batch to stream next, loaded one batch ahead
  auto load_table = [&](uint32_t bi) {
    const uint32_t s0 = bslot[bi], ns = bslot[bi + 1] - s0;
#pragma unroll
    for (int i = 0; i < kIters; ++i) {
      const uint32_t slot = ((uint32_t)(64 * kSWv * i) + (uint32_t)tid) / 9u;
      src_next[i] = slot < ns ? slot_src[s0 + slot] : 0xffffffffu;
    }
  };
  using Buf0 = std::integral_constant<int, 0>;
  using Buf1 = std::integral_constant<int, 1>;
  auto stream_batch = [&](uint32_t bi, auto which) {   // global -> LDS, asynchronous (vmcnt); the LDS image is lane-linear
    unsigned char* zb_ = decltype(which)::value ? zbuf1 : zbuf0;
    uint4* rb_ = decltype(which)::value ? recbuf1 : recbuf0;
#pragma unroll
    for (int i = 0; i < kIters; ++i)
      if (src_next[i] != 0xffffffffu) {
        const uint32_t q = (uint32_t)(64 * kSWv * i) + (uint32_t)tid;
        gather16_to_lds(reinterpret_cast<const unsigned char*>(pt.Z) + 16ull * (src_next[i] + q % 9u), lds_address(zb_) + 16u * (uint32_t)(64 * kSWv * i + 64 * wv));
      }
    const uint32_t vb = bfirst[bi], nv = bfirst[bi + 1] - vb;
    if (wv == 0)
      for (uint32_t c0 = 0; c0 < nv; c0 += 64)
        if (c0 + lane < nv) gather16_to_lds(visits + vb + c0 + lane, lds_address(rb_) + 16u * c0);
  };

  // ---- one visit.  Record: x = byte offset of strip frame 0 in the batch image for the row operands (int32), y = the same for the
  //      group's column operands, z = tail slot | distance to the second layer << 16, w = column tiles of the group in use (5 bits) | stereo << 15 | row tiles in use << 16
  // A visit in two halves -- load_ops issues every LDS read of the visit (row operands, the operands of the active column tiles,
  // (u_l, 0)), multiply does the rest -- so that a wavefront can take its visits two at a time: both records, then both sets of
  // operands, are read together and the read latency (exposed at two wavefronts per SIMD) is paid once per pair.
  struct Ops { double a[kSTR]; double b[kSGC]; double ul; uint32_t bits; };
  auto load_ops = [&](const uint32_t vx, const uint32_t vy, const uint32_t vz, const uint32_t vw, auto which) -> Ops {
    constexpr int bf = decltype(which)::value;
    schur_lds8* zimg = bf ? (schur_lds8*)(&zbuf1[0]) : (schur_lds8*)(&zbuf0[0]);
    const bool twin = TWIN && ((vw >> 15) & 1u);
    const uint32_t layer2 = 144u * (vz >> 16) * kmul;
    auto lds_f64 = [](uint32_t addr) -> double { return *reinterpret_cast<schur_ldsd*>((schur_lds8*)(uintptr_t)addr); };
    auto operand = [&](uint32_t base, uint32_t lane_const) -> double {
      const uint32_t at = base * kmul + lane_const;
      double val = lds_f64(at);
      if (TWIN && twin) val += lds_f64(at + layer2);
      return val;
    };
    Ops o;
    o.bits = vw;
    // every operand is read whether its tile is active or not (a read is an add and a ds_read_b64; testing the tile bits first cost five
    // scalar instructions per operand): an inactive tile's address may lie anywhere -- beyond the allocation LDS returns zero -- and its
    // value is never used (multiply() touches the row / column tiles of the visit's masks only)
#pragma unroll
    for (int r = 0; r < kSTR; ++r) o.a[r] = operand(vx, cA[bf][r]);
    o.ul = with_rhs ? *reinterpret_cast<schur_ldsd*>(zimg + (144u * (vz & 0xffffu) + 8u * kq)) : 0.0;   // (u_l, 0)
#pragma unroll
    for (int c = 0; c < kSGC; ++c) o.b[c] = operand(vy, cB[bf][c]);
    return o;
  };
  // The active tiles of a visit are (row tiles holding one of the point's row frames) x (column tiles of the group holding one of its
  // column frames), minus -- in the group that holds the chunk's own frames -- the tiles above the diagonal: a rectangle given by two
  // masks (a column bit, then the row bits of the columns in use).
  constexpr int kCut = kSDiag - (kSTC - kSGC);   // diagonal group: column tile c of the group lies at or below row tile r iff c <= r + kCut
  auto multiply = [&](const Ops& o) {
    const uint32_t R = (o.bits >> 16) & 7u;
    if (with_rhs) {
#pragma unroll
      for (int r = 0; r < kSTR; ++r) if (R & (1u << r)) racc[r] += o.a[r] * o.ul;
    }
#pragma unroll
    for (int c = 0; c < kSGC; ++c) {
      if ((o.bits >> c) & 1u) {
        // rows of this column tile: the visit's row mask, minus -- chunk's own group only -- the row tiles above the diagonal
        uint32_t Rc = R;
        if (c > kCut) Rc = with_rhs ? (R & ~((1u << (c - kCut)) - 1u)) : R;
#pragma unroll
        for (int r = 0; r < kSTR; ++r)
          if (Rc & (1u << r)) mfma_f64_acc(acc[c][r], o.a[r], o.b[c]);
      }
    }
  };

  const uint32_t b0 = wg_bptr[blockIdx.x], b1 = wg_bptr[blockIdx.x + 1];
  if (b0 < b1) { load_table(b0); stream_batch(b0, Buf0{}); }
  if (b0 + 1 < b1) load_table(b0 + 1);
  auto batch = [&](uint32_t bi, auto which, auto other) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's part of batch bi has landed, its successor's table entries too
    __syncthreads();                                    // ... everybody's; and the other buffer is free
    if (bi + 1 < b1) { stream_batch(bi + 1, other); if (bi + 2 < b1) load_table(bi + 2); }
    const uint32_t nv = bfirst[bi + 1] - bfirst[bi];
    const uint4* rb_ = decltype(which)::value ? recbuf1 : recbuf0;
    auto ops_of = [&](uint32_t x) {
      const uint4 rv = rb_[x];
      return load_ops(__builtin_amdgcn_readfirstlane(rv.x), __builtin_amdgcn_readfirstlane(rv.y), __builtin_amdgcn_readfirstlane(rv.z), __builtin_amdgcn_readfirstlane(rv.w), which);
    };
    constexpr int kVisitGroup = OBVI_SCHUR_VISIT_GROUP;   // visits a wavefront takes together
    uint32_t i = (uint32_t)wv;
    for (; i + (kVisitGroup - 1) * kSWv < nv; i += kVisitGroup * kSWv) {
      Ops o[kVisitGroup];
#pragma unroll
      for (int v = 0; v < kVisitGroup; ++v) o[v] = ops_of(i + v * kSWv);
#pragma unroll
      for (int v = 0; v < kVisitGroup; ++v) multiply(o[v]);
    }
    for (; i < nv; i += kSWv) multiply(ops_of(i));
  };
  for (uint32_t bi = b0; bi < b1; bi += 2) {
    batch(bi, Buf0{}, Buf1{});
    if (bi + 1 < b1) batch(bi + 1, Buf1{}, Buf0{});
  }

  // ---- add the workgroup's tiles to the tile grid: tile (r, cbase + c), lane, register q -> row 16 r + kq + 4 q, column 16 (cbase + c) + m.
  //      The four wavefronts hold private accumulators of the same strip: they are summed through LDS first (the batch buffers are free),
  //      tile column by tile column, so that the strip costs one set of atomics instead of four.
  static_assert(kSWv * kSTR * 4 * 64 * sizeof(double) <= kSchurBatchBytes && (kSTR * 4) % kSWv == 0, "one tile column of every wavefront fits a batch buffer");
  double* red = reinterpret_cast<double*>(&zbuf0[0]);   // [wavefront][r][q][lane]
  __syncthreads();
#pragma unroll
  for (int c = 0; c < kSGC; ++c) {
#pragma unroll
    for (int r = 0; r < kSTR; ++r) {
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[c][r]));   // 16-pass MFMA result -> VALU read: 18 wait states
#pragma unroll
      for (int q = 0; q < 4; ++q) red[((wv * kSTR + r) * 4 + q) * 64 + lane] = acc[c][r][q];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kSTR * 4 / kSWv; ++i) {
      const int pq = (kSTR * 4 / kSWv) * wv + i, r = pq >> 2, q = pq & 3;   // wave-uniform
      double v = 0.0;
#pragma unroll
      for (int w = 0; w < kSWv; ++w) v += red[((w * kSTR + r) * 4 + q) * 64 + lane];
      if (v != 0.0) {
        const int row = 16 * r + kq + 4 * q, col = 16 * (cbase + c) + m;
        const int fp = kSBack + row / 6, x = row % 6, fq = col / 6, y = col % 6;
        if (!(fq > fp || (fq == fp && y > x))) {
          const int64_t rp_ = rown[fp], rq_ = rown[fq];
          if (rp_ >= 0 && rq_ >= 0) {
            if (fq == fp) atomic_add_f64(S_at(rd.S, rd.nt, rp_ + x, rp_ + y), -v);
            else if (rp_ > rq_) atomic_add_f64(S_at(rd.S, rd.nt, rp_ + x, rq_ + y), -v);
            else atomic_add_f64(S_at(rd.S, rd.nt, rq_ + y, rp_ + x), -v);
          }
        }
      }
    }
    __syncthreads();
  }
  if (with_rhs) {   // uniform per workgroup
    // the four streams' partial sums of (Z u) meet in LDS and are added by one wavefront, in a fixed order: one atomic per row and
    // workgroup instead of four whose order would change from run to run
#pragma unroll
    for (int r = 0; r < kSTR; ++r) {
      double tot = racc[r];
      tot += __shfl_xor(tot, 16, 64);
      tot += __shfl_xor(tot, 32, 64);
      if (kq == 0) red[(wv * kSTR + r) * 16 + m] = tot;
    }
    __syncthreads();
    if (wv == 0 && kq == 0) {
#pragma unroll
      for (int r = 0; r < kSTR; ++r) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < kSWv; ++w) tot += red[(w * kSTR + r) * 16 + m];
        const int row = 16 * r + m;
        const int64_t rp_ = rown[kSBack + row / 6];
        if (tot != 0.0 && rp_ >= 0) atomic_add_f64(rd.rhs + rp_ + row % 6, -tot);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// K6.  Back-substitution of the eliminated points, candidate point.
// ---------------------------------------------------------------------------------------
// G lanes share a feature: lane g takes its sightings beg + g, beg + g + G, ... and the G partial sums meet in a butterfly: a lane per
// feature walks all of a feature's records one dependent round trip after the other (300 k features x 10 sightings: 169 us with one
// lane, 116 us with eight).  A workgroup takes feature groups block, block + stride, ... and adds its three sums to the step's scalars
// once: they are single addresses of one cache line, and a same-line atomic costs ~40 ns (12 500 workgroups were measured at 1.5 ms).
template <int G>
__device__ __forceinline__ void point_backsub_block(int64_t block, int64_t stride, const BlocksDev& b, const ReprojDev& rp, const PointDev& pt, const ReducedDev& rd,
                                                    const double* __restrict__ points, double* __restrict__ points_cand, double* scal) {
  const uint32_t g = threadIdx.x % G;
  double stepsq = 0.0, bad = 0.0, model = 0.0;
  for (int64_t l = (block * (int64_t)kBlock + threadIdx.x) / G; l < b.L; l += stride * (kBlock / G)) {   // uniform over the G lanes of a feature
    if (b.point_var[l]) {
      double t0 = 0.0, t1 = 0.0, t2 = 0.0;
      const uint32_t end = rp.point_ptr[l + 1];
      // Branch-free: an observation of a constant pose (yr < 0) reads y[0..5] and multiplies by zero instead of skipping, so that the row
      // lookup and the record (whose address does not depend on it) are requested together: one exposed latency less per observation.
      // 16-byte loads: z_off() is even, and a pose's rows start at an even row of the tile grid.
      for (uint32_t a = rp.point_ptr[l] + g; a < end; a += G) {
        const int32_t yr = rp.yrow[a];   // one lookup instead of active -> pose -> variable id -> row
        const double2* Z2 = reinterpret_cast<const double2*>(pt.Z + z_off(a, l));
        const double2* y2 = reinterpret_cast<const double2*>(rd.y + (yr < 0 ? 0 : yr));
        double Z[18], y[6];
#pragma unroll
        for (int x = 0; x < 9; ++x) { const double2 v = Z2[x]; Z[2 * x] = v.x; Z[2 * x + 1] = v.y; }
#pragma unroll
        for (int x = 0; x < 3; ++x) { const double2 v = y2[x]; y[2 * x] = v.x; y[2 * x + 1] = v.y; }
        const double on = yr < 0 ? 0.0 : 1.0;
#pragma unroll
        for (int x = 0; x < 6; ++x) { const double yx = on * y[x]; t0 -= Z[3 * x] * yx; t1 -= Z[3 * x + 1] * yx; t2 -= Z[3 * x + 2] * yx; }
      }
#pragma unroll
      for (int m = 1; m < G; m <<= 1) {   // butterfly over the G lanes of the feature: inside a quad by DPP quad permutes, beyond it through ds_bpermute
        if (m == 1) { t0 += dpp_quad<0xB1>(t0); t1 += dpp_quad<0xB1>(t1); t2 += dpp_quad<0xB1>(t2); }        // quad_perm:[1,0,3,2]
        else if (m == 2) { t0 += dpp_quad<0x4E>(t0); t1 += dpp_quad<0x4E>(t1); t2 += dpp_quad<0x4E>(t2); }   // quad_perm:[2,3,0,1]
        else { t0 += __shfl_xor(t0, m); t1 += __shfl_xor(t1, m); t2 += __shfl_xor(t2, m); }
      }
      if (g == 0) {
        t0 += pt.u[3 * l]; t1 += pt.u[3 * l + 1]; t2 += pt.u[3 * l + 2];
        const double* Ci = pt.Ci + 6 * l;
        // y_l = Ci^T t ; delta = -y_l
        const double d0 = -(Ci[0] * t0 + Ci[1] * t1 + Ci[3] * t2), d1 = -(Ci[2] * t1 + Ci[4] * t2), d2 = -(Ci[5] * t2);
        if (!isfinite(d0) || !isfinite(d1) || !isfinite(d2)) bad = 1.0;
        points_cand[3 * l] = points[3 * l] + d0; points_cand[3 * l + 1] = points[3 * l + 1] + d1; points_cand[3 * l + 2] = points[3 * l + 2] + d2;
        stepsq += d0 * d0 + d1 * d1 + d2 * d2;
        model += 0.5 * (pt.lam[3 * l] * d0 * d0 + pt.lam[3 * l + 1] * d1 * d1 + pt.lam[3 * l + 2] * d2 * d2 - (pt.gl[3 * l] * d0 + pt.gl[3 * l + 1] * d1 + pt.gl[3 * l + 2] * d2));
      }
    } else if (g == 0) {
      points_cand[3 * l] = points[3 * l]; points_cand[3 * l + 1] = points[3 * l + 1]; points_cand[3 * l + 2] = points[3 * l + 2];
    }
  }
  block_accumulate(stepsq, scal, SC_STEPSQ, b.deterministic);
  block_accumulate(bad, scal, SC_NONFINITE);
  block_accumulate(model, scal, SC_MODEL_CHANGE, b.deterministic);
}

__device__ __forceinline__ void apply_reduced_step_block(int64_t block, const BlocksDev& b, const ReducedDev& rd, const double* __restrict__ poses, const double* __restrict__ objects,
                                                         double* __restrict__ poses_cand, double* __restrict__ objects_cand, PoseCache* __restrict__ pc_cand, double* scal) {
  const int64_t t = block * (int64_t)kBlock + threadIdx.x;
  double stepsq = 0.0, bad = 0.0, model = 0.0;
  if (t < b.P + b.O) {
    const bool is_pose = t < b.P;
    const int64_t idx = is_pose ? t : t - b.P;
    const int d = is_pose ? 6 : b.od;
    const int32_t vid = is_pose ? b.pose_vid[idx] : b.obj_vid[idx];
    const double* x = is_pose ? poses + 6 * idx : objects + d * idx;
    double* xc = is_pose ? poses_cand + 6 * idx : objects_cand + d * idx;
    const int64_t row = vid < 0 ? 0 : (is_pose ? b.pose_row[vid] : b.obj_row[vid]);
    const int64_t ci = vid < 0 ? 0 : (is_pose ? 6 * (int64_t)vid : 6 * (int64_t)b.nPv + d * (int64_t)vid);   // compact index of g, lam
    const bool count = is_pose || vid < 0 || b.obj_shared == nullptr || !b.obj_shared[vid] || b.shared_owner;
    for (int k = 0; k < d; ++k) {
      double v = x[k];
      if (vid >= 0) {
        const double dlt = -rd.y[row + k];
        if (!isfinite(dlt)) bad = 1.0;
        v += dlt;
        if (count) { stepsq += dlt * dlt; model += 0.5 * dlt * (rd.lam[ci + k] * dlt - rd.g[ci + k]); }
      }
      xc[k] = v;
    }
    if (is_pose && pc_cand != nullptr) {   // the candidate's pose cache (what k_pose_cache would write), while the pose is in registers
      double pose[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) pose[k] = xc[k];
      PoseCache pc;
      make_pose_cache(pose, &pc, b.analytic_rotation != 0);
      pc_cand[idx] = pc;
      double* soa = reinterpret_cast<double*>(pc_cand + b.P + 1);
      const double* f = reinterpret_cast<const double*>(&pc);
#pragma unroll
      for (int k = 0; k < 21; ++k) soa[k * b.P + idx] = f[k];
    }
  }
  block_accumulate(stepsq, scal, SC_STEPSQ, b.deterministic);
  block_accumulate(bad, scal, SC_NONFINITE);
  block_accumulate(model, scal, SC_MODEL_CHANGE, b.deterministic);
}

// K6 + K9 in one launch (both only read y): workgroups [0, n_point_blocks) back-substitute the features, the rest form the candidate poses / objects
template <int G>
__global__ void __launch_bounds__(kBlock) k_backsub_apply(BlocksDev b, ReprojDev rp, PointDev pt, ReducedDev rd, const double* __restrict__ points, double* __restrict__ points_cand,
                                                         const double* __restrict__ poses, const double* __restrict__ objects, double* __restrict__ poses_cand,
                                                         double* __restrict__ objects_cand, PoseCache* __restrict__ pc_cand, int n_point_blocks, double* scal) {
  if ((int)blockIdx.x < n_point_blocks) point_backsub_block<G>(blockIdx.x, n_point_blocks, b, rp, pt, rd, points, points_cand, scal);
  else apply_reduced_step_block((int64_t)blockIdx.x - n_point_blocks, b, rd, poses, objects, poses_cand, objects_cand, pc_cand, scal);
}

// ---------------------------------------------------------------------------------------
// K7.  Cost at the trial point (mode 0) / cost of the all-constant residual blocks (mode 1).
// The model cost change  -(J d)^T (r + J d/2)  of [Ceres-doc: TrustRegionMinimizer::ComputeTrustRegionStep] is not
// re-derived from the Jacobian here: with d the exact solution of (J^T J + D) d = -g it equals  -d^T g / 2 + d^T D d / 2,
// which k_point_backsub and k_apply_reduced_step accumulate from quantities they already hold.
// ---------------------------------------------------------------------------------------
// One workgroup per pose over the pose-ordered copy of the observations: the pose cache is uniform per workgroup (a gather
// of it per observation, in point order, costs more L2 bandwidth than everything else the kernel reads).
template <bool PIPE>
__device__ __forceinline__ void cost_reproj_block(int64_t p, const BlocksDev& b, const ReprojPoseDev& rq, const DevCam* __restrict__ cams,
                                                  const PoseCache* __restrict__ pc, const double* __restrict__ points, int mode, double* scal) {
  const bool pose_var = b.pose_vid[p] >= 0;
  const PoseCache cache = pc[p];
  double cost = 0.0;
  const uint32_t beg = rq.pose_ptr[p], end = rq.pose_ptr[p + 1];
  if (PIPE) {
    // A window (tens of poses, a thousand sightings each): a thread's 4-5 sightings are a chain of four dependent loads each (flag -> feature
    // index -> its flag -> its position) and there are too few wavefronts to hide it.  Two rounds instead -- everything indexed by k, then
    // everything indexed by the feature -- and the first round of the NEXT sighting in flight while this one is evaluated; an inactive or
    // unwanted sighting is evaluated and not added (its data is valid, only masked).  (On the 2 000-pose problem this form is 5 us slower.)
    struct ByK { uint32_t l; double2 px; double sg; uint16_t cam; uint8_t act; };
    auto by_k = [&](uint32_t k) { ByK o; o.act = rq.active[k]; o.l = rq.point[k]; o.px = rq.pixel[k]; o.cam = rq.cam[k]; o.sg = rq.sigma[k]; return o; };
    uint32_t k = beg + threadIdx.x;
    ByK cur = {};
    if (k < end) cur = by_k(k);
    while (k < end) {
      const uint32_t kn = k + kBlock;
      ByK nxt = {};
      if (kn < end) nxt = by_k(kn);
      const uint32_t l = cur.l;
      const bool var = pose_var || b.point_var[l] != 0;
      const double X[3] = {points[3 * (int64_t)l], points[3 * (int64_t)l + 1], points[3 * (int64_t)l + 2]};
      double r[2], rho0, w;
      reproj_eval<false>(cache, cams[cur.cam], X, cur.px.x, cur.px.y, cur.sg, r, nullptr, nullptr);
      huber_eval(r[0] * r[0] + r[1] * r[1], rq.huber, &rho0, &w);
      if (cur.act && var == (mode == 0)) cost += 0.5 * rho0;
      cur = nxt; k = kn;
    }
  } else {
    for (uint32_t k = beg + threadIdx.x; k < end; k += kBlock) {
      if (!rq.active[k]) continue;
      const uint32_t l = rq.point[k];
      const bool var = pose_var || b.point_var[l] != 0;
      if (var != (mode == 0)) continue;
      const double X[3] = {points[3 * (int64_t)l], points[3 * (int64_t)l + 1], points[3 * (int64_t)l + 2]};
      const double2 px = rq.pixel[k];
      double r[2], rho0, w;
      reproj_eval<false>(cache, cams[rq.cam[k]], X, px.x, px.y, rq.sigma[k], r, nullptr, nullptr);
      huber_eval(r[0] * r[0] + r[1] * r[1], rq.huber, &rho0, &w);
      cost += 0.5 * rho0;
    }
  }
  block_accumulate(cost, scal, mode == 0 ? SC_COST_CAND : SC_COST_FIXED, b.deterministic);
}

// small factors: one kernel, thread ranges [bbox | shape | ltm | relpose]
__device__ __forceinline__ void cost_small_block(int64_t block, const BlocksDev& b, const SmallFactorsDev& sf, const DevCam* __restrict__ cams,
                                                 const double* __restrict__ poses, const double* __restrict__ objects, int mode, double* scal) {
  int64_t t = block * (int64_t)kBlock + threadIdx.x;
  const int od = b.od;
  double cost = 0.0, rho0, w;
  if (t < sf.n_bb) {
    const int64_t i = t;
    const uint32_t o = sf.bb_obj[i], p = sf.bb_pose[i];
    if (sf.bb_active[i] && (b.obj_vid[o] >= 0 || b.pose_vid[p] >= 0) == (mode == 0)) {
      Dual<1> res[4];
      bbox_eval_1(od, objects + od * (int64_t)o, poses + 6 * (int64_t)p, cams[sf.bb_cam[i]], sf.bb_rect + 4 * i, sf.bb_sqrt_inf + 16 * i, sf.bb_invalid, res);
      huber_eval(res[0].v * res[0].v + res[1].v * res[1].v + res[2].v * res[2].v + res[3].v * res[3].v, sf.bb_huber, &rho0, &w);
      cost = 0.5 * rho0;
    }
  } else if ((t -= sf.n_bb) < sf.n_sp) {
    const int64_t i = t;
    const uint32_t o = sf.sp_obj[i];
    if (sf.sp_active[i] && (b.obj_vid[o] >= 0) == (mode == 0)) {
      double r[3];
      shape_prior_eval(objects + od * (int64_t)o, sf.sp_mean + 3 * i, sf.sp_sqrt_inf + 9 * i, r, nullptr, od);
      huber_eval(r[0] * r[0] + r[1] * r[1] + r[2] * r[2], sf.sp_huber, &rho0, &w);
      cost = 0.5 * rho0;
    }
  } else if ((t -= sf.n_sp) < sf.n_lt) {
    const int64_t i = t;
    const uint32_t o = sf.lt_obj[i];
    if (sf.lt_active[i] && (b.obj_vid[o] >= 0) == (mode == 0)) {
      double r[9], s = 0.0;
      ltm_prior_eval(objects + od * (int64_t)o, sf.lt_mean + od * i, sf.lt_sqrt_inf + od * od * i, r, nullptr, od);
      for (int a = 0; a < od; ++a) s += r[a] * r[a];
      huber_eval(s, sf.lt_huber, &rho0, &w);
      cost = 0.5 * rho0;
    }
  } else if ((t -= sf.n_lt) < sf.n_rl) {
    const int64_t i = t;
    const uint32_t pa = sf.rl_a[i], pb = sf.rl_b[i];
    if (sf.rl_active[i] && (b.pose_vid[pa] >= 0 || b.pose_vid[pb] >= 0) == (mode == 0)) {
      Dual<1> res[6];
      double s = 0.0;
      relpose_eval_n<1>(poses + 6 * (int64_t)pa, poses + 6 * (int64_t)pb, sf.rl_t + 3 * i, sf.rl_R + 9 * i, sf.rl_sqrt_inf + 36 * i, res);
      for (int a = 0; a < 6; ++a) s += res[a].v * res[a].v;
      huber_eval(s, sf.rl_huber, &rho0, &w);
      cost = 0.5 * rho0;
    }
  }
  block_accumulate(cost, scal, mode == 0 ? SC_COST_CAND : SC_COST_FIXED, b.deterministic);
}
// one launch: workgroups [0, n_pose_blocks) take the reprojection factors of a pose, the rest the small factor families
template <bool PIPE>
__global__ void __launch_bounds__(kBlock) k_cost(BlocksDev b, ReprojPoseDev rq, SmallFactorsDev sf, const DevCam* __restrict__ cams, const PoseCache* __restrict__ pc,
                                                const double* __restrict__ poses, const double* __restrict__ points, const double* __restrict__ objects, int mode,
                                                int n_pose_blocks, double* scal) {
  if ((int)blockIdx.x < n_pose_blocks) cost_reproj_block<PIPE>(blockIdx.x, b, rq, cams, pc, points, mode, scal);
  else cost_small_block((int64_t)blockIdx.x - n_pose_blocks, b, sf, cams, poses, objects, mode, scal);
}

// ---------------------------------------------------------------------------------------
// Problem::Evaluate -- every active factor, caller order
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_eval_reproj(ReprojDev rp, const uint32_t* __restrict__ perm, const DevCam* __restrict__ cams,
                                                       const PoseCache* __restrict__ pc, const double* __restrict__ points, int apply_loss,
                                                       double* residuals, double* sqnorm, double* scal, int det) {
  const int64_t a = blockIdx.x * (int64_t)kBlock + threadIdx.x;
  double cost = 0.0;
  if (a < rp.n) {
    const uint32_t orig = perm[a];
    double r[2] = {0.0, 0.0}, s = 0.0;
    if (rp.active[a]) {
      const uint32_t l = rp.point[a];
      const double X[3] = {points[3 * (int64_t)l], points[3 * (int64_t)l + 1], points[3 * (int64_t)l + 2]};
      const double2 px = rp.pixel[a];
      reproj_eval<false>(pc[rp.pose[a]], cams[rp.cam[a]], X, px.x, px.y, rp.sigma[a], r, nullptr, nullptr);
      s = r[0] * r[0] + r[1] * r[1];
      if (apply_loss) {
        double rho0, w;
        huber_eval(s, rp.huber, &rho0, &w);
        cost = 0.5 * rho0;
        const double sw = sqrt(w);
        r[0] *= sw; r[1] *= sw;
      } else {
        cost = 0.5 * s;
      }
    }
    if (residuals) { residuals[2 * (int64_t)orig] = r[0]; residuals[2 * (int64_t)orig + 1] = r[1]; }
    if (sqnorm) sqnorm[orig] = s;
  }
  block_accumulate(cost, scal, SC_COST, det);
}

template <int M>
__device__ __forceinline__ double finish_eval(double* r, double huber, int apply_loss, double* residuals, double* sqnorm, int64_t i, bool active) {
  double s = 0.0, cost = 0.0;
  if (active) {
    for (int a = 0; a < M; ++a) s += r[a] * r[a];
    if (apply_loss) {
      double rho0, w;
      huber_eval(s, huber, &rho0, &w);
      cost = 0.5 * rho0;
      const double sw = sqrt(w);
      for (int a = 0; a < M; ++a) r[a] *= sw;
    } else {
      cost = 0.5 * s;
    }
  } else {
    for (int a = 0; a < M; ++a) r[a] = 0.0;
  }
  if (residuals) for (int a = 0; a < M; ++a) residuals[M * i + a] = r[a];
  if (sqnorm) sqnorm[i] = s;
  return cost;
}

__global__ void __launch_bounds__(64) k_eval_small(SmallFactorsDev sf, const DevCam* __restrict__ cams, const double* __restrict__ poses,
                                                  const double* __restrict__ objects, int apply_loss, double* res_bb, double* sq_bb,
                                                  double* res_sp, double* sq_sp, double* res_lt, double* sq_lt, double* res_rl, double* sq_rl, double* scal, int det) {
  int64_t t = blockIdx.x * 64LL + threadIdx.x;
  const int od = sf.od;
  double cost = 0.0;
  if (t < sf.n_bb) {
    const int64_t i = t;
    double r[4] = {0, 0, 0, 0};
    const bool act = sf.bb_active[i] != 0;
    if (act) {
      if (od == 7) {   // (the 13-direction form: the values every earlier round's fixtures were taken with)
        D13 res[4];
        bbox_eval(objects + 7 * (int64_t)sf.bb_obj[i], poses + 6 * (int64_t)sf.bb_pose[i], cams[sf.bb_cam[i]], sf.bb_rect + 4 * i, sf.bb_sqrt_inf + 16 * i, sf.bb_invalid, res);
        for (int a = 0; a < 4; ++a) r[a] = res[a].v;
      } else {
        Dual<1> res[4];
        bbox_eval_n<1, 9>(objects + 9 * (int64_t)sf.bb_obj[i], poses + 6 * (int64_t)sf.bb_pose[i], cams[sf.bb_cam[i]], sf.bb_rect + 4 * i, sf.bb_sqrt_inf + 16 * i, sf.bb_invalid, res);
        for (int a = 0; a < 4; ++a) r[a] = res[a].v;
      }
    }
    cost = finish_eval<4>(r, sf.bb_huber, apply_loss, res_bb, sq_bb, i, act);
  } else if ((t -= sf.n_bb) < sf.n_sp) {
    const int64_t i = t;
    double r[3] = {0, 0, 0};
    const bool act = sf.sp_active[i] != 0;
    if (act) shape_prior_eval(objects + od * (int64_t)sf.sp_obj[i], sf.sp_mean + 3 * i, sf.sp_sqrt_inf + 9 * i, r, nullptr, od);
    cost = finish_eval<3>(r, sf.sp_huber, apply_loss, res_sp, sq_sp, i, act);
  } else if ((t -= sf.n_sp) < sf.n_lt) {
    const int64_t i = t;
    double r[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bool act = sf.lt_active[i] != 0;
    if (act) ltm_prior_eval(objects + od * (int64_t)sf.lt_obj[i], sf.lt_mean + od * i, sf.lt_sqrt_inf + od * od * i, r, nullptr, od);
    cost = od == 7 ? finish_eval<7>(r, sf.lt_huber, apply_loss, res_lt, sq_lt, i, act) : finish_eval<9>(r, sf.lt_huber, apply_loss, res_lt, sq_lt, i, act);
  } else if ((t -= sf.n_lt) < sf.n_rl) {
    const int64_t i = t;
    double r[6] = {0, 0, 0, 0, 0, 0};
    const bool act = sf.rl_active[i] != 0;
    if (act) {
      D12 res[6];
      relpose_eval(poses + 6 * (int64_t)sf.rl_a[i], poses + 6 * (int64_t)sf.rl_b[i], sf.rl_t + 3 * i, sf.rl_R + 9 * i, sf.rl_sqrt_inf + 36 * i, res);
      for (int a = 0; a < 6; ++a) r[a] = res[a].v;
    }
    cost = finish_eval<6>(r, sf.rl_huber, apply_loss, res_rl, sq_rl, i, act);
  }
  cost = wave_sum(cost);
  if (threadIdx.x == 0) scal_add(scal, det, SC_COST, cost);
}

// deterministic mode: the partial sums the workgroups of the previous kernel left behind the scalar block, added up in a fixed order (one
// workgroup per scalar: strided per-thread sums, then a fixed tree) and added to the scalar -- plain read-modify-write, stream order
__global__ void __launch_bounds__(kBlock) k_det_reduce(double* scal, int64_t nblocks, uint32_t scalar_mask, int stride) {
  const int slot = blockIdx.x, sc = det_scalar_of(slot);
  if (!((scalar_mask >> sc) & 1u)) return;
  __shared__ double sm[kBlock];
  const double* part = scal + SC_COUNT + (int64_t)slot * stride;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < nblocks; i += kBlock) acc += part[i];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int w = kBlock / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) scal[sc] += sm[0];
}

__global__ void __launch_bounds__(kBlock) k_debug_lin_reproj(ReprojDev rp, const uint32_t* __restrict__ perm, const DevCam* __restrict__ cams,
                                                            const PoseCache* __restrict__ pc, const double* __restrict__ points, double* r_out, double* J0, double* J1) {
  const int64_t a = blockIdx.x * (int64_t)kBlock + threadIdx.x;
  if (a >= rp.n) return;
  const int64_t o = perm[a];
  const uint32_t l = rp.point[a];
  const double X[3] = {points[3 * (int64_t)l], points[3 * (int64_t)l + 1], points[3 * (int64_t)l + 2]};
  const double2 px = rp.pixel[a];
  double r[2], Jp[12], Jl[6];
  reproj_eval<true>(pc[rp.pose[a]], cams[rp.cam[a]], X, px.x, px.y, rp.sigma[a], r, Jp, Jl);
  r_out[2 * o] = r[0]; r_out[2 * o + 1] = r[1];
  for (int k = 0; k < 12; ++k) J0[12 * o + k] = Jp[k];
  for (int k = 0; k < 6; ++k) J1[6 * o + k] = Jl[k];
}

__global__ void __launch_bounds__(64) k_debug_lin_small(int type, SmallFactorsDev sf, const DevCam* __restrict__ cams, const double* __restrict__ poses,
                                                       const double* __restrict__ objects, double* r_out, double* J0, double* J1) {
  const int64_t i = blockIdx.x * 64LL + threadIdx.x;
  const int od = sf.od;
  if (type == 2 && i < sf.n_bb && od == 7) {
    D13 res[4];
    bbox_eval(objects + 7 * (int64_t)sf.bb_obj[i], poses + 6 * (int64_t)sf.bb_pose[i], cams[sf.bb_cam[i]], sf.bb_rect + 4 * i, sf.bb_sqrt_inf + 16 * i, sf.bb_invalid, res);
    for (int a = 0; a < 4; ++a) {
      r_out[4 * i + a] = res[a].v;
      for (int k = 0; k < 7; ++k) J0[28 * i + 7 * a + k] = res[a].d[k];
      for (int k = 0; k < 6; ++k) J1[24 * i + 6 * a + k] = res[a].d[7 + k];
    }
  } else if (type == 2 && i < sf.n_bb) {
    Dual<15> res[4];
    bbox_eval_n<15, 9>(objects + 9 * (int64_t)sf.bb_obj[i], poses + 6 * (int64_t)sf.bb_pose[i], cams[sf.bb_cam[i]], sf.bb_rect + 4 * i, sf.bb_sqrt_inf + 16 * i, sf.bb_invalid, res);
    for (int a = 0; a < 4; ++a) {
      r_out[4 * i + a] = res[a].v;
      for (int k = 0; k < 9; ++k) J0[36 * i + 9 * a + k] = res[a].d[k];
      for (int k = 0; k < 6; ++k) J1[24 * i + 6 * a + k] = res[a].d[9 + k];
    }
  } else if (type == 3 && i < sf.n_sp) {
    double r[3], J[27];
    shape_prior_eval(objects + od * (int64_t)sf.sp_obj[i], sf.sp_mean + 3 * i, sf.sp_sqrt_inf + 9 * i, r, J, od);
    for (int a = 0; a < 3; ++a) r_out[3 * i + a] = r[a];
    for (int k = 0; k < 3 * od; ++k) J0[3 * od * i + k] = J[k];
  } else if (type == 4 && i < sf.n_lt) {
    double r[9], J[81];
    ltm_prior_eval(objects + od * (int64_t)sf.lt_obj[i], sf.lt_mean + od * i, sf.lt_sqrt_inf + od * od * i, r, J, od);
    for (int a = 0; a < od; ++a) r_out[od * i + a] = r[a];
    for (int k = 0; k < od * od; ++k) J0[od * od * i + k] = J[k];
  } else if (type == 5 && i < sf.n_rl) {
    D12 res[6];
    relpose_eval(poses + 6 * (int64_t)sf.rl_a[i], poses + 6 * (int64_t)sf.rl_b[i], sf.rl_t + 3 * i, sf.rl_R + 9 * i, sf.rl_sqrt_inf + 36 * i, res);
    for (int a = 0; a < 6; ++a) {
      r_out[6 * i + a] = res[a].v;
      for (int k = 0; k < 6; ++k) { J0[36 * i + 6 * a + k] = res[a].d[k]; J1[36 * i + 6 * a + k] = res[a].d[6 + k]; }
    }
  }
}

__global__ void __launch_bounds__(kBlock) k_fill(double* p, int64_t n, double v) {
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) p[i] = v;
}

// multi-GPU exchange (1): (Hdiag od^2 | g od) of the shared objects <-> contiguous buffer (56 doubles per object; 90 with the 9-parameter block)
__global__ void __launch_bounds__(128) k_pack_shared_blocks(BlocksDev b, ReducedDev rd, const int32_t* __restrict__ shared_ov, int32_t n_shared, double* buf, int unpack) {
  const int o = blockIdx.x, od = b.od, nh = od * od, n = nh + od;
  if (o >= n_shared || (int)threadIdx.x >= n) return;
  const int32_t ov = shared_ov[o];
  double* src = (int)threadIdx.x < nh ? rd.Hdiag + 36 * b.nPv + nh * (int64_t)ov + threadIdx.x : rd.g + 6 * b.nPv + od * (int64_t)ov + (threadIdx.x - nh);
  if (unpack) *src = buf[n * (int64_t)o + threadIdx.x]; else buf[n * (int64_t)o + threadIdx.x] = *src;
}
// multi-GPU exchange (2): lower tiles (i >= j >= t0) of the tile grid, then rhs rows [t0*64, nt*64)
__global__ void __launch_bounds__(kBlock) k_pack_tail(ReducedDev rd, int32_t t0, double* buf, int unpack) {
  const int nt = rd.nt, ntail = nt - t0;
  const int ntiles = ntail * (ntail + 1) / 2;
  const int job = blockIdx.x;
  if (job < ntiles) {
    int i = (int)((sqrtf(8.0f * (float)job + 1.0f) - 1.0f) * 0.5f);
    while ((i + 1) * (i + 2) / 2 <= job) ++i;
    while (i * (i + 1) / 2 > job) --i;
    const int j = job - i * (i + 1) / 2;
    double* tile = rd.S + ((int64_t)(t0 + i) * nt + (t0 + j)) * (kTile * kTile);
    double* dst = buf + (int64_t)job * (kTile * kTile);
    for (int e = threadIdx.x; e < kTile * kTile; e += kBlock) { if (unpack) tile[e] = dst[e]; else dst[e] = tile[e]; }
  } else {
    double* dst = buf + (int64_t)ntiles * (kTile * kTile);
    for (int e = threadIdx.x; e < ntail * kTile; e += kBlock) { if (unpack) rd.rhs[(int64_t)t0 * kTile + e] = dst[e]; else dst[e] = rd.rhs[(int64_t)t0 * kTile + e]; }
  }
}

inline unsigned grid_for(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

}  // namespace

// =========================================================================================
void launch_det_reduce(hipStream_t s, double* scal, int64_t nblocks, uint32_t scalar_mask, int stride) {
  if (nblocks > stride) throw std::logic_error("deterministic mode: a grid larger than the partial-sum slots (ensure_det_slots() in ba_handle.h undercounts)");
  if (nblocks > 0) hipLaunchKernelGGL(k_det_reduce, dim3(kDetSlots), dim3(kBlock), 0, s, scal, nblocks, scalar_mask, stride);
}
#define OBVI_SC(x) (1u << (x))
void launch_reproj_gather(hipStream_t s, int64_t n, const uint32_t* perm, const uint32_t* rq_src, const uint32_t* rp_point, const uint16_t* raw_cam,
                          const double2* raw_pixel, const double* raw_sigma, double sigma_scalar, uint16_t* cam, double2* pixel, double* sigma,
                          uint32_t* q_point, uint16_t* q_cam, double2* q_pixel, double* q_sigma, uint8_t* q_active) {
  if (n > 0) hipLaunchKernelGGL(k_reproj_gather, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, n, perm, rq_src, rp_point, raw_cam, raw_pixel, raw_sigma, sigma_scalar, cam, pixel, sigma,
                                q_point, q_cam, q_pixel, q_sigma, q_active);
}
void launch_pose_cache(hipStream_t s, int64_t P, const double* poses, PoseCache* out, int analytic) {
  if (P > 0) hipLaunchKernelGGL(k_pose_cache, dim3(grid_for(P, kBlock)), dim3(kBlock), 0, s, P, poses, out, analytic);
}
void launch_point_pass(hipStream_t s, const BlocksDev& b, const ReprojDev& rp, const DevCam* cams, const PoseCache* pc, const double* points,
                       const ReducedDev& rd, const PointDev& pt, double radius, int first_iter, double* scal, const uint32_t* wave_obs, int64_t n_waves,
                       const uint32_t* long_points, int64_t n_long) {
  if (n_waves > 0) {
    hipLaunchKernelGGL(k_point_pass, dim3(grid_for(n_waves, kBlock / 64)), dim3(kBlock), 0, s, b, rp, cams, pc, points, rd, pt, radius, first_iter, scal, wave_obs, n_waves);
    if (b.deterministic) launch_det_reduce(s, scal, grid_for(n_waves, kBlock / 64), OBVI_SC(SC_COST) | OBVI_SC(SC_GSQ) | OBVI_SC(SC_XSQ), b.deterministic);
  }
  if (n_long > 0) {
    hipLaunchKernelGGL(k_point_pass_long, dim3(grid_for(n_long, kBlock)), dim3(kBlock), 0, s, b, rp, cams, pc, points, rd, pt, radius, first_iter, scal, long_points, n_long);
    if (b.deterministic) launch_det_reduce(s, scal, grid_for(n_long, kBlock), OBVI_SC(SC_COST) | OBVI_SC(SC_GSQ) | OBVI_SC(SC_XSQ), b.deterministic);
  }
}
void launch_pose_pass(hipStream_t s, const BlocksDev& b, const ReprojPoseDev& rq, const DevCam* cams, const PoseCache* pc, const double* points,
                      const ReducedDev& rd) {
  if (b.P <= 0 || rq.n <= 0) return;
  // a pose's workgroup walks its sightings 256 at a time; with few poses (a window) that loop is the latency of the launch: cut it
  // (not in the deterministic mode: one writer per block)
  static const int max_slices = std::getenv("OBVI_POSE_PASS_SLICES") ? std::atoi(std::getenv("OBVI_POSE_PASS_SLICES")) : 8;   // tuning knob
  const int64_t per_pose = (rq.n + b.P - 1) / b.P;
  int slices = 1;
  static const int64_t slice_below = std::getenv("OBVI_POSE_PASS_SLICE_BELOW") ? std::atoll(std::getenv("OBVI_POSE_PASS_SLICE_BELOW")) : 256;   // tuning knob (poses)
  if (!b.deterministic && b.P <= slice_below) slices = (int)std::max<int64_t>(1, std::min<int64_t>(max_slices, (per_pose + kBlock - 1) / kBlock));
  // few poses: the loads of a sighting in two rounds with the next sighting's first round in flight (150 registers); many poses: the plain loop
  // (120 registers: beside the strip kernel the side stream is otherwise the longer one -- 2.02 vs 1.95 ms per LM iteration)
  if (b.P <= slice_below) hipLaunchKernelGGL(k_pose_pass<true>, dim3((unsigned)(b.P * slices)), dim3(kBlock), 0, s, b, rq, cams, pc, points, rd, slices);
  else hipLaunchKernelGGL(k_pose_pass<false>, dim3((unsigned)(b.P * slices)), dim3(kBlock), 0, s, b, rq, cams, pc, points, rd, slices);
}
void launch_small_factors(hipStream_t s, const BlocksDev& b, const SmallFactorsDev& sf, const DevCam* cams, const double* poses,
                          const double* objects, const ReducedDev& rd, double* scal) {
  // few factors (a sliding window): 16 lanes per factor, the latency of a handful of wavefronts is the whole side stream; one launch
  const int64_t lanes_below = std::getenv("OBVI_SMALL_LANES_BELOW") ? std::atoll(std::getenv("OBVI_SMALL_LANES_BELOW")) : 4096;   // tuning knob
  const int nb_bbox = (int)grid_for(sf.n_bb, 4), nb_priors = (int)grid_for(sf.n_sp + sf.n_lt, 64), nb_rel = (int)grid_for(sf.n_rl, 4);
  if (nb_bbox + nb_priors + nb_rel == 0) return;
  // (the bounding-box lanes are compiled per ellipsoid block size: 13 or 15 directions on the factor's 16 lanes)
#define OBVI_SMALL_LIN(STORE) do { if (b.od == 9) hipLaunchKernelGGL((k_small_lin_lanes<STORE, 9>), dim3(nb_bbox + nb_priors + nb_rel), dim3(64), 0, s, b, sf, cams, poses, objects, rd, scal, nb_bbox, nb_priors); \
                                   else hipLaunchKernelGGL((k_small_lin_lanes<STORE, 7>), dim3(nb_bbox + nb_priors + nb_rel), dim3(64), 0, s, b, sf, cams, poses, objects, rd, scal, nb_bbox, nb_priors); } while (0)
  if (b.deterministic) {
    // no fp64 atomics on the diagonal blocks: every factor leaves its blocks in a scratch slot, the gathers add them per target in list order
    OBVI_SMALL_LIN(true);
    launch_det_reduce(s, scal, nb_bbox + nb_priors + nb_rel, OBVI_SC(SC_COST), b.deterministic);
    if (sf.n_bb > 0) hipLaunchKernelGGL(k_bbox_gather, dim3((unsigned)b.O + grid_for(b.P, kBlock / 64)), dim3(kBlock), 0, s, b, sf, rd);
    if (sf.n_sp + sf.n_lt + sf.n_rl > 0) hipLaunchKernelGGL(k_small_gather, dim3(grid_for(b.O + b.P, kBlock / 64)), dim3(kBlock), 0, s, b, sf, rd);
    return;
  }
  if (sf.n_bb < lanes_below) {
    OBVI_SMALL_LIN(false);
    return;
  }
  // many bounding boxes: per-factor blocks into the scratch, then one wavefront per object / pose sums them (no atomics); the priors and
  // the relative-pose factors ride in the first launch, 16 lanes per factor at every size (a thread per factor left the 2 000 odometry
  // factors of the global problem as 32 wavefronts dragging a 12-direction dual: 250 us)
  OBVI_SMALL_LIN(true);
  hipLaunchKernelGGL(k_bbox_gather, dim3((unsigned)b.O + grid_for(b.P, kBlock / 64)), dim3(kBlock), 0, s, b, sf, rd);
#undef OBVI_SMALL_LIN
}
void launch_reduced_diag(hipStream_t s, const BlocksDev& b, const double* poses, const double* objects, const ReducedDev& rd, double radius,
                         int first_iter, double* scal) {
  if (b.P + b.O > 0) {
    const int per = b.od > 8 ? 16 : 8;   // threads per diagonal block
    hipLaunchKernelGGL(k_reduced_diag, dim3(grid_for(per * (b.P + b.O), kBlock)), dim3(kBlock), 0, s, b, poses, objects, rd, radius, first_iter, scal);
    if (b.deterministic) launch_det_reduce(s, scal, grid_for(per * (b.P + b.O), kBlock), OBVI_SC(SC_GSQ) | OBVI_SC(SC_XSQ), b.deterministic);
  }
}
void launch_schur_blocks(hipStream_t s, int64_t nblk, const uint32_t* blk_row, const uint32_t* blk_col, const uint32_t* blk_ptr,
                         const uint32_t* pair_a, const uint32_t* pair_b, const uint32_t* obs_point, const PointDev& pt, const ReducedDev& rd) {
  if (nblk > 0) hipLaunchKernelGGL(k_schur_blocks, dim3((unsigned)(8 * ((nblk + 7) / 8))), dim3(kBlock), 0, s, nblk, blk_row, blk_col, blk_ptr, pair_a, pair_b, obs_point, pt, rd);
}
void launch_schur_window(hipStream_t s, int64_t nwg, int has_twins, const BlocksDev& b, const PointDev& pt, const ReducedDev& rd, const int32_t* row_of_nat,
                         const uint32_t* wg_bptr, const uint32_t* bfirst, const uint32_t* bslot, const uint32_t* visits, const uint32_t* slot_src,
                         const int32_t* wg_f0, const int32_t* wg_group) {
  if (nwg <= 0) return;
  const uint4* v = reinterpret_cast<const uint4*>(visits);
  if (has_twins) hipLaunchKernelGGL(k_schur_window<true>, dim3((unsigned)nwg), dim3(64 * kSWv), 0, s, b, pt, rd, row_of_nat, wg_bptr, bfirst, bslot, v, slot_src, wg_f0, wg_group);
  else hipLaunchKernelGGL(k_schur_window<false>, dim3((unsigned)nwg), dim3(64 * kSWv), 0, s, b, pt, rd, row_of_nat, wg_bptr, bfirst, bslot, v, slot_src, wg_f0, wg_group);
}
void launch_backsub_apply(hipStream_t s, const BlocksDev& b, const ReprojDev& rp, const PointDev& pt, const ReducedDev& rd, const double* points,
                          double* points_cand, const double* poses, const double* objects, double* poses_cand, double* objects_cand, PoseCache* pc_cand, double* scal) {
  // lanes per feature: enough that the features' sightings spread over the chip, no more than a feature has sightings to hand out.
  // Measured on 300 k features x 10 sightings (us): 1 lane 169, 2 133, 4 120, 8 116, 16 146, 32 156 -- past 8 the wavefronts' record lines
  // push each other out of the 32 KB vector cache between the nine loads of a record.
  const char* env = getenv("OBVI_BACKSUB_LANES");   // per launch: the tests flip it inside one process
  const int forced = env ? atoi(env) : 0;
  const int64_t per = b.L > 0 ? rp.n / b.L : 0;
  int G = per >= 32 ? 8 : per >= 8 ? 4 : per >= 4 ? 2 : 1;
  if (forced == 1 || forced == 2 || forced == 4 || forced == 8 || forced == 16 || forced == 32) G = forced;
  const int n_point_blocks = (int)std::min<int64_t>(grid_for(b.L * G, kBlock), 2048);   // 8 per CU, each walks its share (flat between 512 and 2048)
  const unsigned grid = (unsigned)n_point_blocks + grid_for(b.P + b.O, kBlock);
  if (grid == 0) return;
#define OBVI_BACKSUB(GG) hipLaunchKernelGGL(k_backsub_apply<GG>, dim3(grid), dim3(kBlock), 0, s, b, rp, pt, rd, points, points_cand, poses, objects, poses_cand, objects_cand, pc_cand, n_point_blocks, scal)
  switch (G) { case 32: OBVI_BACKSUB(32); break; case 16: OBVI_BACKSUB(16); break; case 8: OBVI_BACKSUB(8); break; case 4: OBVI_BACKSUB(4); break; case 2: OBVI_BACKSUB(2); break; default: OBVI_BACKSUB(1); }
#undef OBVI_BACKSUB
  if (b.deterministic) launch_det_reduce(s, scal, grid, OBVI_SC(SC_STEPSQ) | OBVI_SC(SC_MODEL_CHANGE), b.deterministic);
}
void launch_cost(hipStream_t s, const BlocksDev& b, const ReprojPoseDev& rq, const SmallFactorsDev& sf, const DevCam* cams, const PoseCache* pc_cur,
                 const double* poses_cur, const double* points_cur, const double* objects_cur, const PoseCache* pc_cand, const double* poses_cand,
                 const double* points_cand, const double* objects_cand, int mode, double* scal) {
  // mode 0: cost of the variable residual blocks at the candidate; mode 1: cost of the all-constant blocks at the current point
  const PoseCache* pc = mode == 0 ? pc_cand : pc_cur;
  const double* poses = mode == 0 ? poses_cand : poses_cur;
  const double* points = mode == 0 ? points_cand : points_cur;
  const double* objects = mode == 0 ? objects_cand : objects_cur;
  const int n_pose_blocks = rq.n > 0 && b.P > 0 ? (int)b.P : 0;
  const int64_t ns = sf.n_bb + sf.n_sp + sf.n_lt + sf.n_rl;
  const unsigned grid = (unsigned)n_pose_blocks + grid_for(ns, kBlock);
  if (grid > 0) {
    if (b.P <= 256) hipLaunchKernelGGL(k_cost<true>, dim3(grid), dim3(kBlock), 0, s, b, rq, sf, cams, pc, poses, points, objects, mode, n_pose_blocks, scal);   // (few poses: cost_reproj_block)
    else hipLaunchKernelGGL(k_cost<false>, dim3(grid), dim3(kBlock), 0, s, b, rq, sf, cams, pc, poses, points, objects, mode, n_pose_blocks, scal);
    if (b.deterministic) launch_det_reduce(s, scal, grid, mode == 0 ? OBVI_SC(SC_COST_CAND) : OBVI_SC(SC_COST_FIXED), b.deterministic);
  }
}
void launch_evaluate(hipStream_t s, const BlocksDev& b, const ReprojDev& rp, const uint32_t* rp_perm, const SmallFactorsDev& sf, const DevCam* cams,
                     const PoseCache* pc, const double* poses, const double* points, const double* objects, int apply_loss, double* residuals,
                     double* sqnorm, double* scal) {
  if (rp.n > 0) {
    hipLaunchKernelGGL(k_eval_reproj, dim3(grid_for(rp.n, kBlock)), dim3(kBlock), 0, s, rp, rp_perm, cams, pc, points, apply_loss, residuals, sqnorm, scal, b.deterministic);
    if (b.deterministic) launch_det_reduce(s, scal, grid_for(rp.n, kBlock), OBVI_SC(SC_COST), b.deterministic);
  }
  const int64_t ns = sf.n_bb + sf.n_sp + sf.n_lt + sf.n_rl;
  if (ns > 0) {
    double* r_bb = residuals ? residuals + 2 * rp.n : nullptr;
    double* r_sp = residuals ? r_bb + 4 * sf.n_bb : nullptr;
    double* r_lt = residuals ? r_sp + 3 * sf.n_sp : nullptr;
    double* r_rl = residuals ? r_lt + sf.od * sf.n_lt : nullptr;
    double* q_bb = sqnorm ? sqnorm + rp.n : nullptr;
    double* q_sp = sqnorm ? q_bb + sf.n_bb : nullptr;
    double* q_lt = sqnorm ? q_sp + sf.n_sp : nullptr;
    double* q_rl = sqnorm ? q_lt + sf.n_lt : nullptr;
    hipLaunchKernelGGL(k_eval_small, dim3(grid_for(ns, 64)), dim3(64), 0, s, sf, cams, poses, objects, apply_loss, r_bb, q_bb, r_sp, q_sp, r_lt, q_lt, r_rl, q_rl, scal, b.deterministic);
    if (b.deterministic) launch_det_reduce(s, scal, grid_for(ns, 64), OBVI_SC(SC_COST), b.deterministic);
  }
}
void launch_debug_linearize_reproj(hipStream_t s, const ReprojDev& rp, const uint32_t* rp_perm, const DevCam* cams, const PoseCache* pc,
                                   const double* points, double* r, double* J0, double* J1) {
  if (rp.n > 0) hipLaunchKernelGGL(k_debug_lin_reproj, dim3(grid_for(rp.n, kBlock)), dim3(kBlock), 0, s, rp, rp_perm, cams, pc, points, r, J0, J1);
}
void launch_debug_linearize_small(hipStream_t s, int factor_type, const SmallFactorsDev& sf, const DevCam* cams, const double* poses,
                                  const double* objects, double* r, double* J0, double* J1) {
  const int64_t n = factor_type == 2 ? sf.n_bb : factor_type == 3 ? sf.n_sp : factor_type == 4 ? sf.n_lt : sf.n_rl;
  if (n > 0) hipLaunchKernelGGL(k_debug_lin_small, dim3(grid_for(n, 64)), dim3(64), 0, s, factor_type, sf, cams, poses, objects, r, J0, J1);
}
__global__ void __launch_bounds__(kBlock) k_permute_rows3(double* __restrict__ dst, const double* __restrict__ src, const uint32_t* __restrict__ map, int64_t n) {
  const int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t j = map[i];
  dst[3 * i] = src[3 * j]; dst[3 * i + 1] = src[3 * j + 1]; dst[3 * i + 2] = src[3 * j + 2];
}
void launch_permute_rows3(hipStream_t s, double* dst, const double* src, const uint32_t* map, int64_t n) {
  if (n > 0) hipLaunchKernelGGL(k_permute_rows3, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, dst, src, map, n);
}
void launch_pack_shared_blocks(hipStream_t s, const BlocksDev& b, const ReducedDev& rd, const int32_t* shared_ov, int32_t n_shared, double* buf, int unpack) {
  if (n_shared > 0) hipLaunchKernelGGL(k_pack_shared_blocks, dim3(n_shared), dim3(128), 0, s, b, rd, shared_ov, n_shared, buf, unpack);
}
// multi-GPU: the scalar block's sums and the gradient maximum in ONE all-reduce (sum): buf = [SC_COST, SC_SUM_END) | one slot per rank
// holding that rank's maximum (zero elsewhere), so that after the sum every rank sees every maximum and takes the largest itself
__global__ void __launch_bounds__(64) k_pack_scalars(double* scal, double* buf, int32_t rank, int32_t world, int unpack) {
  const int t = threadIdx.x;
  constexpr int n = SC_SUM_END - SC_COST;
  if (!unpack) {
    for (int i = t; i < n + world; i += 64) buf[i] = i < n ? scal[SC_COST + i] : (i - n == rank ? scal[SC_GMAX_BITS] : 0.0);
  } else {
    for (int i = t; i < n; i += 64) scal[SC_COST + i] = buf[i];
    if (t == 0) {
      double m = 0.0;   // non-negative doubles: the value whose bits the slot holds
      for (int r = 0; r < world; ++r) m = fmax(m, buf[n + r]);
      scal[SC_GMAX_BITS] = m;
    }
  }
}
void launch_pack_scalars(hipStream_t s, double* scal, double* buf, int32_t rank, int32_t world, int unpack) {
  hipLaunchKernelGGL(k_pack_scalars, dim3(1), dim3(64), 0, s, scal, buf, rank, world, unpack);
}
void launch_pack_tail(hipStream_t s, const ReducedDev& rd, int32_t t0, double* buf, int unpack) {
  const int ntail = rd.nt - t0;
  if (ntail > 0) hipLaunchKernelGGL(k_pack_tail, dim3(ntail * (ntail + 1) / 2 + 1), dim3(kBlock), 0, s, rd, t0, buf, unpack);
}
// three arrays in one launch (the parameter blocks: poses, points, objects)
__global__ void __launch_bounds__(kBlock) k_copy3(double* d0, const double* s0, int64_t n0, double* d1, const double* s1, int64_t n1, double* d2, const double* s2, int64_t n2) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n0 + n1 + n2; i += stride) {
    if (i < n0) d0[i] = s0[i];
    else if (i < n0 + n1) d1[i - n0] = s1[i - n0];
    else d2[i - n0 - n1] = s2[i - n0 - n1];
  }
}
void launch_copy3(hipStream_t s, double* d0, const double* s0, int64_t n0, double* d1, const double* s1, int64_t n1, double* d2, const double* s2, int64_t n2) {
  const int64_t n = n0 + n1 + n2;
  if (n > 0) hipLaunchKernelGGL(k_copy3, dim3((unsigned)std::min<int64_t>((n + kBlock - 1) / kBlock, 2048)), dim3(kBlock), 0, s, d0, s0, n0, d1, s1, n1, d2, s2, n2);
}
void launch_fill(hipStream_t s, double* p, int64_t n, double v) {
  if (n > 0) hipLaunchKernelGGL(k_fill, dim3((unsigned)std::min<int64_t>((n + kBlock - 1) / kBlock, 4096)), dim3(kBlock), 0, s, p, n, v);
}

}  // namespace obvi
