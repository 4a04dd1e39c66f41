// Factor kernels: NonlinearFactorGraph::linearize / error over one bucket = (factor type, noise kind).
// One lane per factor, FP64 VALU, variables read AoS (a camera is one 120-B row, a point one 24-B row),
// whitened Jacobians [A1 A2 b] written column-major per factor like the reference's VerticalBlockMatrix.
// The SFM bucket (the 1M-factor one) stages its 26-double Jacobian rows through LDS so that a wave
// writes 64 x 208 B contiguously.
//   a2  NonlinearFactorGraph::linearize            gtsam/nonlinear/NonlinearFactorGraph.cpp:239-278
//   a3  GeneralSFMFactor::linearize/evaluateError  gtsam/slam/GeneralSFMFactor.h:127-177
//   a4  projection chain                           gtsam/geometry/CalibratedCamera.cpp:27-46,88-94,116-135,
//                                                  gtsam/geometry/Cal3Bundler.cpp:64-90, PinholePose.h:90-109
//   a5  NoiseModelFactor::linearize, BetweenFactor gtsam/nonlinear/NonlinearFactor.cpp:152-184, slam/BetweenFactor.h:111-124
//   a6  PriorFactor, GenericProjectionFactor       gtsam/nonlinear/PriorFactor.h:98-102, slam/ProjectionFactor.h:138-165
//   a7  NoiseModelFactor::error                    gtsam/nonlinear/NonlinearFactor.cpp:138-149
#pragma once
#include <hip/hip_runtime.h>

#include "geometry_dev.hpp"

namespace lmgpu {

struct BucketDev {
  int32_t type;
  int32_t n;
  int32_t noise_kind;     // 0 unit, 2 diag (inverse sigmas), 3 gauss (R row-major)
  const int32_t* vidx;    // n x arity : index of each variable inside its type array
  const double* meas;     // n x meas doubles
  const double* noise;    // n x (rows | rows*rows) or null
  double* J;              // n x rows*(cols) whitened [A1 A2 b], col-major per factor
  const int32_t* epos;    // n : position in the error buffer (= rank of the factor by graph index)
  int32_t robust;         // lmgpu_robust_kind: noiseModel::Robust around the Gaussian model (0 = none)
  double rk;              // its tuning constant
  const int32_t* sel;     // null: all n factors of the bucket; else n indices into the bucket (ISAM2 relinearizes a subset,
                          // gtsam/nonlinear/ISAM2.cpp:66-114)
};

struct ValuesDev {
  const double* v[6];  // POSE2 [n][3], POSE3 [n][12], POINT3 [n][3], CAM [n][15], POINT2 [n][2], CAL3_S2 [n][5]
};

// whiten Jl (col-major M x COLS) in place
template <int M, int COLS>
__device__ __forceinline__ void whiten_block(double* Jl, int noise_kind, const double* nz) {
  if (noise_kind == 2) {
#pragma unroll
    for (int r = 0; r < M; r++) {
      const double s = nz[r];
#pragma unroll
      for (int c = 0; c < COLS; c++) Jl[c * M + r] *= s;
    }
  } else if (noise_kind == 3) {
#pragma unroll
    for (int c = 0; c < COLS; c++) {
      double t[M];
#pragma unroll
      for (int r = 0; r < M; r++) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < M; k++) s += nz[r * M + k] * Jl[c * M + k];
        t[r] = s;
      }
#pragma unroll
      for (int r = 0; r < M; r++) Jl[c * M + r] = t[r];
    }
  }
}

// m-estimators (gtsam/linear/LossFunctions.cpp), kinds as in lmgpu_robust_kind; d = ||whitened error|| >= 0
__device__ __forceinline__ double robust_weight(int kind, double k, double d) {
  switch (kind) {
    case 1: return 1.0 / (1.0 + d / k);                      // Fair :146-148
    case 2: return (d <= k) ? 1.0 : k / d;                   // Huber :179-182
    case 3: return (k * k) / (k * k + d * d);                // Cauchy :217-219
    case 4: {                                                // Tukey :250-256
      const double t = 1.0 - d * d / (k * k);
      return (d <= k) ? t * t : 0.0;
    }
    case 5: return exp(-(d * d) / (k * k));                  // Welsch :289-292
    case 6: {                                                // Geman-McClure :320-325
      const double c2 = k * k, c2e = c2 + d * d;
      return c2 * c2 / (c2e * c2e);
    }
    case 7: {                                                // DCS :354-362
      const double e2 = d * d, w = 2.0 * k / (k + e2);
      return (e2 > k) ? w * w : 1.0;
    }
    case 8: return (d <= k) ? 0.0 : (d - k) / d;             // L2WithDeadZone :400-407
    default: return 1.0;
  }
}
__device__ __forceinline__ double robust_loss(int kind, double k, double d) {
  switch (kind) {
    case 1: return k * k * (d / k - log1p(d / k));           // :150-155
    case 2: return (d <= k) ? d * d / 2 : k * (d - k / 2);   // :184-191
    case 3: return k * k * log1p(d * d / (k * k)) * 0.5;     // :221-224
    case 4: {                                                // :258-266
      const double t = 1.0 - d * d / (k * k);
      return (d <= k) ? k * k * (1 - t * t * t) / 6.0 : k * k / 6.0;
    }
    case 5: return k * k * 0.5 * -expm1(-(d * d) / (k * k)); // :294-297
    case 6: return 0.5 * (k * k * d * d) / (k * k + d * d);  // :327-331
    case 7: {                                                // :365-373
      const double e2 = d * d;
      return (k * k * e2 + k * e2 * e2) / ((e2 + k) * (e2 + k));
    }
    case 8: return (d < k) ? 0.0 : 0.5 * (k - d) * (k - d);  // :409-412
    default: return 0.5 * d * d;
  }
}

// Robust::WhitenSystem after the Gaussian whitening: [A b] *= sqrt(weight(||b||))  (Block reweighting, LossFunctions.cpp:61-76)
template <int M, int COLS>
__device__ __forceinline__ void robust_reweight(double* Jl, int kind, double k) {
  double s = 0;
#pragma unroll
  for (int r = 0; r < M; r++) s += Jl[(COLS - 1) * M + r] * Jl[(COLS - 1) * M + r];
  const double w = sqrt(robust_weight(kind, k, sqrt(s)));
#pragma unroll
  for (int i = 0; i < M * COLS; i++) Jl[i] *= w;
}

// NoiseModelFactor::error: loss(squaredMahalanobisDistance) = 0.5 d^2 (Gaussian) or rho(d) (Robust, NoiseModel.h:717-725)
template <int M>
__device__ __forceinline__ double whitened_half_sq(double* e, int noise_kind, const double* nz, int robust = 0, double rk = 0.0) {
  whiten_block<M, 1>(e, noise_kind, nz);
  double s = 0;
#pragma unroll
  for (int r = 0; r < M; r++) s += e[r] * e[r];
  if (robust) return robust_loss(robust, rk, sqrt(s));
  return 0.5 * s;
}

// ---------------------------------------------------------------- SFM projection math (shared by linearize and error)
// returns false on cheirality.  pi = projected pixel; H1 (2x9) / H2 (2x3) row-major when JAC.
template <bool JAC>
__device__ __forceinline__ bool sfm_project(const double* cam, const double* pt, double* pi, double* H1, double* H2) {
  const D3 d{pt[0] - cam[9], pt[1] - cam[10], pt[2] - cam[11]};
  // q = R^T (p - t)   Pose3::transformTo gtsam/geometry/Pose3.cpp:413-430
  const double qx = cam[0] * d.x + cam[3] * d.y + cam[6] * d.z;
  const double qy = cam[1] * d.x + cam[4] * d.y + cam[7] * d.z;
  const double qz = cam[2] * d.x + cam[5] * d.y + cam[8] * d.z;
  if (qz <= 0) return false;
  const double dz = 1.0 / qz;
  const double u = qx * dz, v = qy * dz;
  const double f = cam[12], k1 = cam[13], k2 = cam[14];
  const double r = u * u + v * v;
  const double g = 1. + (k1 + k2 * r) * r;
  pi[0] = f * (g * u);
  pi[1] = f * (g * v);
  if (JAC) {
    const double a = 2. * (k1 + 2. * k2 * r);
    const double Dp00 = f * (g + a * u * u), Dp01 = f * (a * u * v), Dp11 = f * (g + a * v * v);
    // Dpn_pose  CalibratedCamera.cpp:27-34
    const double uv = u * v, uu = u * u, vv = v * v;
    const double P0[6] = {uv, -1 - uu, v, -dz, 0, dz * u};
    const double P1[6] = {1 + vv, -uv, -u, 0, -dz, dz * v};
#pragma unroll
    for (int j = 0; j < 6; j++) {
      H1[j] = Dp00 * P0[j] + Dp01 * P1[j];
      H1[9 + j] = Dp01 * P0[j] + Dp11 * P1[j];
    }
    // Dcal  Cal3Bundler.cpp:77-80
    const double rx = r * u, ry = r * v;
    H1[6] = g * u; H1[7] = f * rx; H1[8] = f * r * rx;
    H1[15] = g * v; H1[16] = f * ry; H1[17] = f * r * ry;
    // Dpn_point = d * [Rt0 - u Rt2; Rt1 - v Rt2]  (Rt = R^T: Rt(i,j) = cam[3j+i])  CalibratedCamera.cpp:37-46
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double rt0 = cam[3 * j + 0], rt1 = cam[3 * j + 1], rt2 = cam[3 * j + 2];
      const double a0 = (rt0 - u * rt2) * dz, a1 = (rt1 - v * rt2) * dz;
      H2[j] = Dp00 * a0 + Dp01 * a1;
      H2[3 + j] = Dp01 * a0 + Dp11 * a1;
    }
  }
  return true;
}

// LINEARIZE: one lane per factor, Jacobian rows staged through LDS for contiguous 208-B-per-factor stores.
__global__ __launch_bounds__(256) void sfm_linearize_kernel(BucketDev b, ValuesDev vals) {
  __shared__ double stage[256 * 27];  // 27 = 26 + 1 pad (bank spread)
  const int tid = threadIdx.x;
  const int f = blockIdx.x * 256 + tid;
  double Jl[26];
  if (f < b.n) {
    const int ci = b.vidx[2 * f], pi_ = b.vidx[2 * f + 1];
    double cam[15], pt[3];
    const double* cp = vals.v[3] + (size_t)ci * 15;
#pragma unroll
    for (int i = 0; i < 15; i++) cam[i] = cp[i];
    const double* pp = vals.v[2] + (size_t)pi_ * 3;
    pt[0] = pp[0]; pt[1] = pp[1]; pt[2] = pp[2];
    const double zx = b.meas[2 * f], zy = b.meas[2 * f + 1];
    double pix[2], H1[18], H2[6];
    const bool ok = sfm_project<true>(cam, pt, pix, H1, H2);
    if (ok) {
#pragma unroll
      for (int c = 0; c < 9; c++) {
        Jl[2 * c] = H1[c];
        Jl[2 * c + 1] = H1[9 + c];
      }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        Jl[18 + 2 * c] = H2[c];
        Jl[18 + 2 * c + 1] = H2[3 + c];
      }
      Jl[24] = zx - pix[0];
      Jl[25] = zy - pix[1];
    } else {  // cheirality: zero Jacobian, zero b  (GeneralSFMFactor.h:153-157)
#pragma unroll
      for (int i = 0; i < 26; i++) Jl[i] = 0.0;
    }
    if (b.noise_kind != 0) whiten_block<2, 13>(Jl, b.noise_kind, b.noise + (size_t)f * (b.noise_kind == 2 ? 2 : 4));
    if (b.robust) robust_reweight<2, 13>(Jl, b.robust, b.rk);
  } else {
#pragma unroll
    for (int i = 0; i < 26; i++) Jl[i] = 0.0;
  }
#pragma unroll
  for (int i = 0; i < 26; i++) stage[tid * 27 + i] = Jl[i];
  __syncthreads();
  // block writes 256*26 doubles contiguously
  const size_t base = (size_t)blockIdx.x * 256 * 26;
  const size_t total = (size_t)b.n * 26;
#pragma unroll
  for (int k = 0; k < 26; k++) {
    const int lin = k * 256 + tid;          // linear index inside the block's 256x26 tile
    const int ff = lin / 26, cc = lin - ff * 26;
    const size_t g = base + lin;
    if (g < total) b.J[g] = stage[ff * 27 + cc];
  }
}

// the same for a SUBSET of the bucket (b.sel): one lane per selected factor, direct stores (the subset is not contiguous)
__global__ __launch_bounds__(256) void sfm_linearize_sel_kernel(BucketDev b, ValuesDev vals) {
  const int fi = blockIdx.x * 256 + threadIdx.x;
  if (fi >= b.n) return;
  const int f = b.sel[fi];
  const int ci = b.vidx[2 * f], pi_ = b.vidx[2 * f + 1];
  double cam[15], pt[3], Jl[26];
  const double* cp = vals.v[3] + (size_t)ci * 15;
#pragma unroll
  for (int i = 0; i < 15; i++) cam[i] = cp[i];
  const double* pp = vals.v[2] + (size_t)pi_ * 3;
  pt[0] = pp[0]; pt[1] = pp[1]; pt[2] = pp[2];
  double pix[2], H1[18], H2[6];
  if (sfm_project<true>(cam, pt, pix, H1, H2)) {
#pragma unroll
    for (int c = 0; c < 9; c++) {
      Jl[2 * c] = H1[c];
      Jl[2 * c + 1] = H1[9 + c];
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      Jl[18 + 2 * c] = H2[c];
      Jl[18 + 2 * c + 1] = H2[3 + c];
    }
    Jl[24] = b.meas[2 * f] - pix[0];
    Jl[25] = b.meas[2 * f + 1] - pix[1];
  } else {
#pragma unroll
    for (int i = 0; i < 26; i++) Jl[i] = 0.0;
  }
  if (b.noise_kind != 0) whiten_block<2, 13>(Jl, b.noise_kind, b.noise + (size_t)f * (b.noise_kind == 2 ? 2 : 4));
  if (b.robust) robust_reweight<2, 13>(Jl, b.robust, b.rk);
  double* out = b.J + (size_t)f * 26;
#pragma unroll
  for (int i = 0; i < 26; i++) out[i] = Jl[i];
}

__global__ __launch_bounds__(256) void sfm_error_kernel(BucketDev b, ValuesDev vals, double* __restrict__ ebuf) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= b.n) return;
  const int ci = b.vidx[2 * f], pi_ = b.vidx[2 * f + 1];
  double cam[15], pt[3];
  const double* cp = vals.v[3] + (size_t)ci * 15;
#pragma unroll
  for (int i = 0; i < 15; i++) cam[i] = cp[i];
  const double* pp = vals.v[2] + (size_t)pi_ * 3;
  pt[0] = pp[0]; pt[1] = pp[1]; pt[2] = pp[2];
  double pix[2], e[2];
  const bool ok = sfm_project<false>(cam, pt, pix, nullptr, nullptr);
  if (ok) {
    e[0] = pix[0] - b.meas[2 * f];
    e[1] = pix[1] - b.meas[2 * f + 1];
  } else {
    e[0] = e[1] = 0.0;
  }
  ebuf[b.epos[f]] = whitened_half_sq<2>(e, b.noise_kind, b.noise ? b.noise + (size_t)f * (b.noise_kind == 2 ? 2 : 4) : nullptr, b.robust, b.rk);
}

// ---------------------------------------------------------------- generic per-type evaluation
// Each returns the unwhitened error e[M] and (JAC) row-major H1 [M x D0], H2 [M x D1].
template <bool JAC>
__device__ __forceinline__ void eval_between_pose3(const double* m, const double* v0, const double* v1, double* e, double* H1, double* H2) {
  const P3 p1 = load_pose3(v0), p2 = load_pose3(v1), z = load_pose3(m);
  const P3 h = between3(p1, p2);
  if (JAC) {
    // H1 = -Ad(h^-1) = -[R 0; [t]x R  R] of h^-1   (Lie.h:63-69, Pose3.cpp:69-75)
    const P3 hi = inverse3(h);
    const double S[9] = {0, -hi.t.z, hi.t.y, hi.t.z, 0, -hi.t.x, -hi.t.y, hi.t.x, 0};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const double rij = hi.R.m[3 * i + j];
        const double aij = S[3 * i] * hi.R.m[j] + S[3 * i + 1] * hi.R.m[3 + j] + S[3 * i + 2] * hi.R.m[6 + j];
        H1[6 * i + j] = -rij;
        H1[6 * i + 3 + j] = 0.0;
        H1[6 * (i + 3) + j] = -aij;
        H1[6 * (i + 3) + 3 + j] = -rij;
      }
#pragma unroll
    for (int i = 0; i < 36; i++) H2[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) H2[7 * i] = 1.0;
  }
  // Local(measured, h) = Logmap(measured^-1 h)   (BetweenFactor.h:122, Lie.h:136-138; GTSAM_POSE3_EXPMAP)
  pose3_logmap(between3(z, h), e);
}

template <bool JAC>
__device__ __forceinline__ void eval_between_pose2(const double* m, const double* v0, const double* v1, double* e, double* H1, double* H2) {
  const P2 p1 = pose2_from(v0[0], v0[1], v0[2]), p2 = pose2_from(v1[0], v1[1], v1[2]);
  const P2 h = compose2(inverse2(p1), p2);
  if (JAC) {
    const P2 hi = inverse2(h);  // AdjointMap gtsam/geometry/Pose2.cpp:125-135
    H1[0] = -hi.c; H1[1] = hi.s; H1[2] = -hi.y;
    H1[3] = -hi.s; H1[4] = -hi.c; H1[5] = hi.x;
    H1[6] = 0.0; H1[7] = 0.0; H1[8] = -1.0;
#pragma unroll
    for (int i = 0; i < 9; i++) H2[i] = 0.0;
    H2[0] = H2[4] = H2[8] = 1.0;
  }
  const P2 z = pose2_from(m[0], m[1], m[2]);
  const P2 d = compose2(inverse2(z), h);  // ChartAtOrigin::Local = (x, y, theta)  Pose2.cpp:112-121
  e[0] = d.x; e[1] = d.y; e[2] = atan2(d.s, d.c);
}

// BearingRangeFactor<Pose2, Point2>  gtsam/sam/BearingRangeFactor.h:33-77 (ExpressionFactor: error = -Local(value, measured),
// Jacobians of the value, gtsam/nonlinear/ExpressionFactor.h:104-115): Pose2::bearing gtsam/geometry/Pose2.cpp:260-271 over
// transformTo :222-229 and Rot2::relativeBearing gtsam/geometry/Rot2.cpp:134-145; Pose2::range Pose2.cpp:285-299 over norm2
// gtsam/geometry/Point2.cpp:27-36.  m = (bearing angle, range); v0 = pose (x, y, theta); v1 = landmark (x, y).
template <bool JAC>
__device__ __forceinline__ void eval_bearing_range_2d(const double* m, const double* v0, const double* v1, double* e, double* H1, double* H2) {
  const double c = cos(v0[2]), sn = sin(v0[2]);
  const double dx = v1[0] - v0[0], dy = v1[1] - v0[1];
  const double qx = c * dx + sn * dy, qy = -sn * dx + c * dy;
  const double d2 = qx * qx + qy * qy, n = sqrt(d2);
  double hb0 = 0.0, hb1 = 0.0, cb = 1.0, sb = 0.0;
  if (fabs(n) > 1e-5) {
    hb0 = -qy / d2;
    hb1 = qx / d2;
    cb = qx / n;
    sb = qy / n;
  }
  const double r = sqrt(dx * dx + dy * dy);
  double hr0 = 1.0, hr1 = 1.0;
  if (fabs(r) > 1e-10) {
    hr0 = dx / r;
    hr1 = dy / r;
  }
  if (JAC) {
    H1[0] = -hb0;
    H1[1] = -hb1;
    H1[2] = hb0 * qy - hb1 * qx;
    H1[3] = hr0 * -c + hr1 * -sn;
    H1[4] = hr0 * sn + hr1 * -c;
    H1[5] = 0.0;
    H2[0] = hb0 * c + hb1 * -sn;
    H2[1] = hb0 * sn + hb1 * c;
    H2[2] = hr0;
    H2[3] = hr1;
  }
  const double cm = cos(m[0]), sm = sin(m[0]);
  e[0] = -atan2(cb * sm - sb * cm, cb * cm + sb * sm);
  e[1] = r - m[1];
}

template <int D>
__device__ __forceinline__ void set_identity(double* H) {
#pragma unroll
  for (int i = 0; i < D * D; i++) H[i] = 0.0;
#pragma unroll
  for (int i = 0; i < D; i++) H[(D + 1) * i] = 1.0;
}

template <bool JAC>
__device__ __forceinline__ void eval_prior_pose2(const double* m, const double* v0, double* e, double* H1) {
  const P2 x = pose2_from(v0[0], v0[1], v0[2]), z = pose2_from(m[0], m[1], m[2]);
  const P2 d = compose2(inverse2(x), z);
  e[0] = -d.x; e[1] = -d.y; e[2] = -atan2(d.s, d.c);
  if (JAC) set_identity<3>(H1);
}
template <bool JAC>
__device__ __forceinline__ void eval_prior_pose3(const double* m, const double* v0, double* e, double* H1) {
  const P3 x = load_pose3(v0), z = load_pose3(m);
  pose3_logmap(between3(x, z), e);
#pragma unroll
  for (int i = 0; i < 6; i++) e[i] = -e[i];
  if (JAC) set_identity<6>(H1);
}
template <bool JAC>
__device__ __forceinline__ void eval_prior_point3(const double* m, const double* v0, double* e, double* H1) {
#pragma unroll
  for (int i = 0; i < 3; i++) e[i] = -(m[i] - v0[i]);
  if (JAC) set_identity<3>(H1);
}
template <bool JAC>
__device__ __forceinline__ void eval_prior_cam(const double* m, const double* v0, double* e, double* H1) {
  // PinholeCamera::localCoordinates gtsam/geometry/PinholeCamera.h:206-211
  const P3 x = load_pose3(v0), z = load_pose3(m);
  pose3_logmap(between3(x, z), e);
  e[6] = m[12] - v0[12];
  e[7] = m[13] - v0[13];
  e[8] = m[14] - v0[14];
#pragma unroll
  for (int i = 0; i < 9; i++) e[i] = -e[i];
  if (JAC) set_identity<9>(H1);
}
template <bool JAC>
__device__ __forceinline__ void eval_projection(const double* m, const double* v0, const double* v1, double* e, double* H1, double* H2) {
  // GenericProjectionFactor (no body_P_sensor), Cal3_S2::uncalibrate gtsam/geometry/Cal3_S2.cpp:44-50
  const double fx = m[2], fy = m[3], s = m[4], u0 = m[5], v0c = m[6];
  const D3 d{v1[0] - v0[9], v1[1] - v0[10], v1[2] - v0[11]};
  const double qx = v0[0] * d.x + v0[3] * d.y + v0[6] * d.z;
  const double qy = v0[1] * d.x + v0[4] * d.y + v0[7] * d.z;
  const double qz = v0[2] * d.x + v0[5] * d.y + v0[8] * d.z;
  if (qz <= 0) {  // cheirality, throwCheirality_=false: zero Jacobians, error = 2 fx  (ProjectionFactor.h:156-165)
    if (JAC) {
#pragma unroll
      for (int i = 0; i < 12; i++) H1[i] = 0.0;
#pragma unroll
      for (int i = 0; i < 6; i++) H2[i] = 0.0;
    }
    e[0] = e[1] = 2.0 * fx;
    return;
  }
  const double dz = 1.0 / qz;
  const double u = qx * dz, v = qy * dz;
  if (JAC) {
    const double uv = u * v, uu = u * u, vv = v * v;
    const double P0[6] = {uv, -1 - uu, v, -dz, 0, dz * u};
    const double P1[6] = {1 + vv, -uv, -u, 0, -dz, dz * v};
#pragma unroll
    for (int j = 0; j < 6; j++) {
      H1[j] = fx * P0[j] + s * P1[j];
      H1[6 + j] = fy * P1[j];
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double rt0 = v0[3 * j + 0], rt1 = v0[3 * j + 1], rt2 = v0[3 * j + 2];
      const double a0 = (rt0 - u * rt2) * dz, a1 = (rt1 - v * rt2) * dz;
      H2[j] = fx * a0 + s * a1;
      H2[3 + j] = fy * a1;
    }
  }
  e[0] = fx * u + s * v + u0 - m[0];
  e[1] = fy * v + v0c - m[1];
}

template <bool JAC>
__device__ __forceinline__ void eval_prior_cal3_s2(const double* m, const double* v0, double* e, double* H1) {
  // PriorFactor<Cal3_S2>: -Local(x, prior) with Cal3_S2::localCoordinates = T2.vector() - vector() (Cal3_S2.h:118-119)
#pragma unroll
  for (int i = 0; i < 5; i++) e[i] = -(m[i] - v0[i]);
  if (JAC) set_identity<5>(H1);
}

// GeneralSFMFactor2<Cal3_S2> (gtsam/slam/GeneralSFMFactor.h:208-262): one lane per factor, three variables (Pose3, Point3, Cal3_S2).
// error = PinholeCamera<Cal3_S2>(pose, K).project(point) - z with H1 (2x6), H2 (2x3) as GenericProjectionFactor's (PinholePose chain) and
// H3 = Cal3_S2::uncalibrate's Dcal = [u 0 v 1 0; 0 v 0 0 1] at the intrinsic point (u, v) (gtsam/geometry/Cal3_S2.cpp:44-50).  A point
// behind the camera: the reference catches the CheiralityException, zeroes H1..H3 and returns a ZERO error (:251-260).
// [A1 A2 A3 b] is written column-major, 2 x 15.
template <bool JAC>
__global__ __launch_bounds__(128) void sfm2_factor_kernel(BucketDev b, ValuesDev vals, double* __restrict__ ebuf) {
  const int fi = blockIdx.x * blockDim.x + threadIdx.x;
  if (fi >= b.n) return;
  const int f = b.sel ? b.sel[fi] : fi;
  constexpr int M = 2, COLS = 15;
  double m[7], v0[12], v1[3];
  const double* mp = b.meas + (size_t)f * 2;
  const double* p0 = vals.v[1] + (size_t)b.vidx[3 * f] * 12;
  const double* p1 = vals.v[2] + (size_t)b.vidx[3 * f + 1] * 3;
  const double* p2 = vals.v[5] + (size_t)b.vidx[3 * f + 2] * 5;
  m[0] = mp[0];
  m[1] = mp[1];
#pragma unroll
  for (int i = 0; i < 5; i++) m[2 + i] = p2[i];
#pragma unroll
  for (int i = 0; i < 12; i++) v0[i] = p0[i];
#pragma unroll
  for (int i = 0; i < 3; i++) v1[i] = p1[i];
  double e[2], H1[12], H2[6], H3[10];
  const D3 d{v1[0] - v0[9], v1[1] - v0[10], v1[2] - v0[11]};
  const double qz = v0[2] * d.x + v0[5] * d.y + v0[8] * d.z;
  if (qz <= 0) {
    e[0] = e[1] = 0.0;
#pragma unroll
    for (int i = 0; i < 12; i++) H1[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) H2[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 10; i++) H3[i] = 0.0;
  } else {
    eval_projection<JAC>(m, v0, v1, e, H1, H2);
    const double dz = 1.0 / qz;
    const double u = (v0[0] * d.x + v0[3] * d.y + v0[6] * d.z) * dz, v = (v0[1] * d.x + v0[4] * d.y + v0[7] * d.z) * dz;
    H3[0] = u; H3[1] = 0.0; H3[2] = v; H3[3] = 1.0; H3[4] = 0.0;
    H3[5] = 0.0; H3[6] = v; H3[7] = 0.0; H3[8] = 0.0; H3[9] = 1.0;
  }
  const double* nz = b.noise ? b.noise + (size_t)f * (b.noise_kind == 2 ? M : M * M) : nullptr;
  if (JAC) {
    double Jl[M * COLS];
#pragma unroll
    for (int r = 0; r < M; r++) {
#pragma unroll
      for (int c = 0; c < 6; c++) Jl[c * M + r] = H1[r * 6 + c];
#pragma unroll
      for (int c = 0; c < 3; c++) Jl[(6 + c) * M + r] = H2[r * 3 + c];
#pragma unroll
      for (int c = 0; c < 5; c++) Jl[(9 + c) * M + r] = H3[r * 5 + c];
      Jl[14 * M + r] = -e[r];
    }
    whiten_block<M, COLS>(Jl, b.noise_kind, nz);
    if (b.robust) robust_reweight<M, COLS>(Jl, b.robust, b.rk);
    double* out = b.J + (size_t)f * (M * COLS);
#pragma unroll
    for (int i = 0; i < M * COLS; i++) out[i] = Jl[i];
  } else {
    ebuf[b.epos[f]] = whitened_half_sq<M>(e, b.noise_kind, nz, b.robust, b.rk);
  }
}

// Generic bucket kernel.  TYPE selects the evaluator; M rows, D0/D1 tangent dims, S0/S1 stored doubles, T0/T1 value types.
template <int TYPE, int M, int D0, int D1, int ML, int T0, int S0, int T1, int S1, bool JAC>
__device__ __forceinline__ void generic_factor_body(const BucketDev& b, const ValuesDev& vals, double* __restrict__ ebuf, const int fi) {
  if (fi >= b.n) return;
  const int f = b.sel ? b.sel[fi] : fi;
  constexpr int AR = (D1 > 0) ? 2 : 1;
  constexpr int COLS = D0 + D1 + 1;
  double m[ML], v0[S0], v1[S1 > 0 ? S1 : 1];
  const double* mp = b.meas + (size_t)f * ML;
#pragma unroll
  for (int i = 0; i < ML; i++) m[i] = mp[i];
  const double* p0 = vals.v[T0] + (size_t)b.vidx[AR * f] * S0;
#pragma unroll
  for (int i = 0; i < S0; i++) v0[i] = p0[i];
  if (D1 > 0) {
    const double* p1 = vals.v[T1 >= 0 ? T1 : 0] + (size_t)b.vidx[AR * f + 1] * S1;
#pragma unroll
    for (int i = 0; i < S1; i++) v1[i] = p1[i];
  }
  double e[M], H1[JAC ? M * D0 : 1], H2[(JAC && D1 > 0) ? M * D1 : 1];
  if (TYPE == 1) eval_between_pose2<JAC>(m, v0, v1, e, H1, H2);
  if (TYPE == 2) eval_between_pose3<JAC>(m, v0, v1, e, H1, H2);
  if (TYPE == 3) eval_prior_pose2<JAC>(m, v0, e, H1);
  if (TYPE == 4) eval_prior_pose3<JAC>(m, v0, e, H1);
  if (TYPE == 5) eval_prior_point3<JAC>(m, v0, e, H1);
  if (TYPE == 6) eval_prior_cam<JAC>(m, v0, e, H1);
  if (TYPE == 7) eval_projection<JAC>(m, v0, v1, e, H1, H2);
  if (TYPE == 9) eval_bearing_range_2d<JAC>(m, v0, v1, e, H1, H2);
  if (TYPE == 11) eval_prior_cal3_s2<JAC>(m, v0, e, H1);
  if (TYPE == 8) {
    // GenericProjectionFactor with body_P_sensor (ProjectionFactor.h:142-149): camera pose = pose o sensor; H1 = H1_cam Ad(sensor^-1)
    const P3 sensor = load_pose3(m + 7);
    double vc[12];
    store_pose3(compose3(load_pose3(v0), sensor), vc);
    double Hc[JAC ? 12 : 1];
    eval_projection<JAC>(m, vc, v1, e, Hc, H2);
    if (JAC) {
      const P3 hi = inverse3(sensor);  // AdjointMap = [R 0; [t]x R, R]  (Pose3.cpp:69-75)
      const double S[9] = {0, -hi.t.z, hi.t.y, hi.t.z, 0, -hi.t.x, -hi.t.y, hi.t.x, 0};
      double Ad[36];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const double rij = hi.R.m[3 * i + j];
          Ad[6 * i + j] = rij;
          Ad[6 * i + 3 + j] = 0.0;
          Ad[6 * (i + 3) + j] = S[3 * i] * hi.R.m[j] + S[3 * i + 1] * hi.R.m[3 + j] + S[3 * i + 2] * hi.R.m[6 + j];
          Ad[6 * (i + 3) + 3 + j] = rij;
        }
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 6; j++) {
          double v = 0;
#pragma unroll
          for (int k = 0; k < 6; k++) v += Hc[6 * r + k] * Ad[6 * k + j];
          H1[6 * r + j] = v;
        }
    }
  }
  const double* nz = b.noise ? b.noise + (size_t)f * (b.noise_kind == 2 ? M : M * M) : nullptr;
  if (JAC) {
    double Jl[M * COLS];
#pragma unroll
    for (int r = 0; r < M; r++) {
#pragma unroll
      for (int c = 0; c < D0; c++) Jl[c * M + r] = H1[r * D0 + c];
      if (D1 > 0) {
#pragma unroll
        for (int c = 0; c < D1; c++) Jl[(D0 + c) * M + r] = H2[r * D1 + c];
      }
      Jl[(D0 + D1) * M + r] = -e[r];
    }
    whiten_block<M, COLS>(Jl, b.noise_kind, nz);
    if (b.robust) robust_reweight<M, COLS>(Jl, b.robust, b.rk);
    double* out = b.J + (size_t)f * (M * COLS);
#pragma unroll
    for (int i = 0; i < M * COLS; i++) out[i] = Jl[i];
  } else {
    ebuf[b.epos[f]] = whitened_half_sq<M>(e, b.noise_kind, nz, b.robust, b.rk);
  }
}

template <int TYPE, int M, int D0, int D1, int ML, int T0, int S0, int T1, int S1, bool JAC>
__global__ __launch_bounds__(128) void generic_factor_kernel(BucketDev b, ValuesDev vals, double* __restrict__ ebuf) {
  generic_factor_body<TYPE, M, D0, D1, ML, T0, S0, T1, S1, JAC>(b, vals, ebuf, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}

// Linearization of SEVERAL buckets in one launch (ISAM2: an update relinearizes two or three factor types, each a launch of a few
// workgroups of its own before): block blockIdx.x belongs to bucket k with first[k] <= blockIdx.x < first[k + 1]; 128 threads per block
// like the per-bucket launches.  (The SFM types have kernels of their own shape and stay apart.)
#define LIN_MULTI_MAX 6
struct MultiLin {
  BucketDev b[LIN_MULTI_MAX];
  int32_t first[LIN_MULTI_MAX + 1];
  int32_t nb;
};
__global__ __launch_bounds__(128) void linearize_multi_kernel(MultiLin ml, ValuesDev vals) {
  int k = 0;
  while (k + 1 < ml.nb && (int)blockIdx.x >= ml.first[k + 1]) k++;
  const BucketDev& b = ml.b[k];
  const int fi = ((int)blockIdx.x - ml.first[k]) * 128 + (int)threadIdx.x;
  double* nob = nullptr;
  switch (b.type) {
    case 1: generic_factor_body<1, 3, 3, 3, 3, 0, 3, 0, 3, true>(b, vals, nob, fi); break;
    case 2: generic_factor_body<2, 6, 6, 6, 12, 1, 12, 1, 12, true>(b, vals, nob, fi); break;
    case 3: generic_factor_body<3, 3, 3, 0, 3, 0, 3, -1, 0, true>(b, vals, nob, fi); break;
    case 4: generic_factor_body<4, 6, 6, 0, 12, 1, 12, -1, 0, true>(b, vals, nob, fi); break;
    case 5: generic_factor_body<5, 3, 3, 0, 3, 2, 3, -1, 0, true>(b, vals, nob, fi); break;
    case 6: generic_factor_body<6, 9, 9, 0, 15, 3, 15, -1, 0, true>(b, vals, nob, fi); break;
    case 7: generic_factor_body<7, 2, 6, 3, 7, 1, 12, 2, 3, true>(b, vals, nob, fi); break;
    case 8: generic_factor_body<8, 2, 6, 3, 19, 1, 12, 2, 3, true>(b, vals, nob, fi); break;
    case 9: generic_factor_body<9, 2, 3, 2, 2, 0, 3, 4, 2, true>(b, vals, nob, fi); break;
    case 11: generic_factor_body<11, 5, 5, 0, 5, 5, 5, -1, 0, true>(b, vals, nob, fi); break;
    default: break;
  }
}


// ---------------------------------------------------------------- linear error  (a8)
// GaussianFactorGraph::error gtsam/linear/GaussianFactorGraph.cpp:71-78, JacobianFactor::error :509-514
// per factor: 0.5||A x - b||^2 (x = delta) and 0.5||b||^2 (x = 0)
struct FacDesc {
  int64_t joff;   // offset of the factor's [A|b] in the pool
  int32_t x0, x1; // scalar offsets of its variables in delta (x1 = -1 unary)
  int16_t rows, d0, d1, d2;  // d2 > 0: a third variable (GeneralSFMFactor2), its offset in x2
  int32_t x2, pad;
};
// local column q of the factor's [A1 A2 A3 b] -> offset of that scalar in delta (q < d0 + d1 + d2)
__device__ __forceinline__ int fac_xoff(const FacDesc& d, int q) {
  return q < d.d0 ? d.x0 + q : (q < d.d0 + d.d1 ? d.x1 + (q - d.d0) : d.x2 + (q - d.d0 - d.d1));
}

// One lane per factor.  When the 64 factors of a wave are of one shape and stored back to back (always, inside a bucket),
// their [A b] blocks are first copied to LDS as ONE contiguous, coalesced stream (a lane-per-factor read would touch 64
// different cache lines per load instruction and fetch every line several times).
#define LINERR_MAX_SZ 32
__global__ __launch_bounds__(256) void linear_error_kernel(const FacDesc* __restrict__ fd, int nfac, const double* __restrict__ pool,
                                                            const double* __restrict__ delta, double* __restrict__ e0buf,
                                                            double* __restrict__ e1buf) {
  __shared__ double stage[4][64 * LINERR_MAX_SZ + 64];
  const int f = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool valid = f < nfac;
  const FacDesc d = fd[valid ? f : nfac - 1];
  const int m = d.rows, cols = d.d0 + d.d1 + d.d2, sz = m * (cols + 1);
  const long long j0 = __shfl((long long)d.joff, 0, 64);
  const int sz0 = __shfl(sz, 0, 64);
  const int cnt = min(64, nfac - (blockIdx.x * 256 + wave * 64));  // valid lanes of this wave (<= 0: none)
  const bool fast = __all(!valid || (sz == sz0 && sz0 <= LINERR_MAX_SZ && d.joff == j0 + (long long)lane * sz0));
  const double* J = pool + d.joff;
  if (fast && cnt > 0) {
    const double* src = pool + j0;
    double* dst = stage[wave];
    const int pitch = sz0 | 1;  // odd pitch: lanes spread over the banks
    // eight loads in flight, then their stores (left as one load -> wait -> store per iteration by the compiler otherwise)
    const int total = cnt * sz0;
    for (int i0 = lane; i0 < total; i0 += 512) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = (i0 + 64 * u < total) ? src[i0 + 64 * u] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = i0 + 64 * u;
        if (i < total) {
          const int q = i / sz0;
          dst[q * pitch + (i - q * sz0)] = v[u];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    J = dst + lane * pitch;
  }
  if (!valid) return;
  // the factor's columns of delta once, all loads in flight together (they sat inside the row loop: one dependent load per product)
  double dx[12];
  const bool small = cols <= 12;  // (every two-variable factor; a three-variable one takes the loop below)
#pragma unroll
  for (int c = 0; c < 12; c++) dx[c] = delta[(c < d.d0) ? d.x0 + c : ((c < cols && small) ? d.x1 + (c - d.d0) : d.x0)];
  double s0 = 0, s1 = 0;
  for (int r = 0; r < m; r++) {
    const double bb = J[cols * m + r];
    double e = -bb;
    if (small) {
#pragma unroll
      for (int c = 0; c < 12; c++)
        if (c < cols) e += J[c * m + r] * dx[c];
    } else {
      for (int c = 0; c < cols; c++) e += J[c * m + r] * delta[fac_xoff(d, c)];
    }
    s0 += bb * bb;
    s1 += e * e;
  }
  e0buf[f] = 0.5 * s0;
  e1buf[f] = 0.5 * s1;
}

// ---------------------------------------------------------------- hessianDiagonal (a9)
// one thread per variable scalar; CSR var -> (factor, position)
__global__ __launch_bounds__(256) void hessian_diag_kernel(int ntot, const int32_t* __restrict__ scalar_var, const int32_t* __restrict__ scalar_col,
                                                            const int32_t* __restrict__ vi_ptr, const int32_t* __restrict__ vi_fac,
                                                            const int8_t* __restrict__ vi_pos, const FacDesc* __restrict__ fd,
                                                            const double* __restrict__ pool, double* __restrict__ diag) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ntot) return;
  const int v = scalar_var[i], c = scalar_col[i];
  double s = 0;
  for (int k = vi_ptr[v]; k < vi_ptr[v + 1]; k++) {
    const FacDesc d = fd[vi_fac[k]];
    const int col = (vi_pos[k] == 0) ? c : (vi_pos[k] == 1 ? d.d0 + c : d.d0 + d.d1 + c);
    const double* J = pool + d.joff + (size_t)col * d.rows;
    for (int r = 0; r < d.rows; r++) s += J[r] * J[r];
  }
  diag[i] = s;
}

// ---------------------------------------------------------------- deterministic sum of a buffer
__global__ __launch_bounds__(256) void reduce_stage1(const double* __restrict__ buf, int n, double* __restrict__ partial) {
  __shared__ double sh[256];
  double s = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) s += buf[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(256) void reduce_stage2(const double* __restrict__ partial, int n, double* __restrict__ out) {
  __shared__ double sh[256];
  double s = 0;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sh[0];
}

// ---------------------------------------------------------------- retract (a15)
// Values::retract gtsam/nonlinear/Values.cpp:53-64; one thread per variable of a type.
// sel (may be null): the n variables to retract as indices into the type array (ISAM2's retractMasked, gtsam/nonlinear/ISAM2.cpp:465);
// cur == out is allowed (every lane reads its variable before it writes it).
__device__ __forceinline__ void retract_body(int type, int n, const double* cur, double* out, const int32_t* __restrict__ xoff,
                                             const double* __restrict__ delta, const int32_t* __restrict__ sel, const int li) {
  if (li >= n) return;
  const int i = sel ? sel[li] : li;
  const double* d = delta + xoff[i];
  if (type == 0) {  // Pose2: compose(Pose2(d0,d1,d2))  gtsam/geometry/Pose2.cpp:100-110
    const double* v = cur + (size_t)i * 3;
    const P2 c = compose2(pose2_from(v[0], v[1], v[2]), pose2_from(d[0], d[1], d[2]));
    double* o = out + (size_t)i * 3;
    o[0] = c.x; o[1] = c.y; o[2] = atan2(c.s, c.c);
  } else if (type == 1) {
    const P3 p = load_pose3(cur + (size_t)i * 12);
    store_pose3(compose3(p, pose3_expmap(d)), out + (size_t)i * 12);
  } else if (type == 2) {
    const double* v = cur + (size_t)i * 3;
    double* o = out + (size_t)i * 3;
    o[0] = v[0] + d[0]; o[1] = v[1] + d[1]; o[2] = v[2] + d[2];
  } else if (type == 4) {  // Point2: vector space
    const double* v = cur + (size_t)i * 2;
    double* o = out + (size_t)i * 2;
    o[0] = v[0] + d[0]; o[1] = v[1] + d[1];
  } else if (type == 5) {  // Cal3_S2::retract gtsam/geometry/Cal3_S2.h:113-115: vector() + d
    const double* v = cur + (size_t)i * 5;
    double* o = out + (size_t)i * 5;
#pragma unroll
    for (int k = 0; k < 5; k++) o[k] = v[k] + d[k];
  } else {  // PinholeCamera::retract gtsam/geometry/PinholeCamera.h:197-203
    const double* v = cur + (size_t)i * 15;
    double* o = out + (size_t)i * 15;
    const P3 p = load_pose3(v);
    store_pose3(compose3(p, pose3_expmap(d)), o);
    o[12] = v[12] + d[6]; o[13] = v[13] + d[7]; o[14] = v[14] + d[8];
  }
}
__global__ __launch_bounds__(256) void retract_kernel(int type, int n, const double* cur, double* out, const int32_t* __restrict__ xoff,
                                                       const double* __restrict__ delta, const int32_t* __restrict__ sel = nullptr) {
  retract_body(type, n, cur, out, xoff, delta, sel, (int)(blockIdx.x * 256 + threadIdx.x));
}
// the variable types of one masked retraction in one launch (ISAM2 relinearizes poses and points in the same update): blockIdx.y = entry
struct RetractMulti {
  int32_t type[6], n[6];
  const double* cur[6];
  double* out[6];
  const int32_t* xoff[6];
  const int32_t* sel[6];
};
__global__ __launch_bounds__(256) void retract_multi_kernel(RetractMulti rm, const double* __restrict__ delta) {
  const int k = blockIdx.y;
  retract_body(rm.type[k], rm.n[k], rm.cur[k], rm.out[k], rm.xoff[k], delta, rm.sel[k], (int)(blockIdx.x * 256 + threadIdx.x));
}

}  // namespace lmgpu
