// Device-side geometry for the factor kernels (FP64 VALU).  Product code: independent of oracle/.
// Follows the reference's formulas; citations relative to the reference tree.
#pragma once
#include <hip/hip_runtime.h>

namespace lmgpu {

struct D3 {
  double x, y, z;
};
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 operator*(double s, D3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot3(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ D3 cross3(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// row-major 3x3
struct R3 {
  double m[9];
};
__device__ __forceinline__ D3 rot(const R3& R, D3 v) {
  return {R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z, R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z};
}
__device__ __forceinline__ D3 unrot(const R3& R, D3 v) {  // R^T v
  return {R.m[0] * v.x + R.m[3] * v.y + R.m[6] * v.z, R.m[1] * v.x + R.m[4] * v.y + R.m[7] * v.z, R.m[2] * v.x + R.m[5] * v.y + R.m[8] * v.z};
}
__device__ __forceinline__ R3 mul3(const R3& a, const R3& b) {
  R3 c;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) c.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
  return c;
}
__device__ __forceinline__ R3 mul3_tn(const R3& a, const R3& b) {  // a^T b
  R3 c;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) c.m[3 * i + j] = a.m[i] * b.m[j] + a.m[3 + i] * b.m[3 + j] + a.m[6 + i] * b.m[6 + j];
  return c;
}

struct P3 {
  R3 R;
  D3 t;
};
__device__ __forceinline__ P3 load_pose3(const double* v) {
  P3 p;
#pragma unroll
  for (int i = 0; i < 9; i++) p.R.m[i] = v[i];
  p.t = {v[9], v[10], v[11]};
  return p;
}
__device__ __forceinline__ void store_pose3(const P3& p, double* v) {
#pragma unroll
  for (int i = 0; i < 9; i++) v[i] = p.R.m[i];
  v[9] = p.t.x;
  v[10] = p.t.y;
  v[11] = p.t.z;
}
// Pose3::operator*  /  Pose3::inverse gtsam/geometry/Pose3.cpp:61-64 / inverse(a)*b fused (LieGroup::between, Lie.h:63-69)
__device__ __forceinline__ P3 compose3(const P3& a, const P3& b) { return {mul3(a.R, b.R), a.t + rot(a.R, b.t)}; }
__device__ __forceinline__ P3 between3(const P3& a, const P3& b) {
  // a^-1 = (Ra^T, Ra^T(-ta));  a^-1 b = (Ra^T Rb, Ra^T(-ta) + Ra^T tb)
  P3 r;
  r.R = mul3_tn(a.R, b.R);
  D3 nt = unrot(a.R, D3{-a.t.x, -a.t.y, -a.t.z});
  r.t = nt + unrot(a.R, b.t);
  return r;
}
__device__ __forceinline__ P3 inverse3(const P3& a) {
  P3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r.R.m[3 * i + j] = a.R.m[3 * j + i];
  r.t = unrot(a.R, D3{-a.t.x, -a.t.y, -a.t.z});
  return r;
}

// so3::ExpmapFunctor / DexpFunctor gtsam/geometry/SO3.cpp:61-112 ; Pose3::Expmap gtsam/geometry/Pose3.cpp:217-255
__device__ inline P3 pose3_expmap(const double* xi) {
  const D3 w{xi[0], xi[1], xi[2]}, v{xi[3], xi[4], xi[5]};
  const double theta2 = dot3(w, w);
  const bool nearZero = (theta2 <= 1e-5) || (theta2 <= 2.220446049250313e-16);
  double A, B, C;
  if (!nearZero) {
    const double theta = sqrt(theta2);
    const double sin_theta = sin(theta);
    A = sin_theta / theta;
    const double s2 = sin(theta / 2.0);
    const double one_minus_cos = 2.0 * s2 * s2;
    B = one_minus_cos / theta2;
    C = (1 - A) / theta2;
  } else {
    A = 1.0 - theta2 * (1.0 / 6.0);
    B = 0.5 - theta2 * (1.0 / 24.0);
    C = (1.0 / 6.0) - theta2 * (1.0 / 120.0);
  }
  // W = skew(w), WW = W*W ; R = I + A W + B WW
  const double W[9] = {0, -w.z, w.y, w.z, 0, -w.x, -w.y, w.x, 0};
  P3 T;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double ww = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
      T.R.m[3 * i + j] = (i == j ? 1.0 : 0.0) + A * W[3 * i + j] + B * ww;
    }
  const D3 Wv = cross3(w, v);
  const D3 WWv = cross3(w, Wv);
  T.t = v + B * Wv + C * WWv;
  return T;
}

// SO3::Logmap gtsam/geometry/SO3.cpp:299-375
__device__ inline D3 so3_logmap(const R3& R) {
  const double R11 = R.m[0], R12 = R.m[1], R13 = R.m[2];
  const double R21 = R.m[3], R22 = R.m[4], R23 = R.m[5];
  const double R31 = R.m[6], R32 = R.m[7], R33 = R.m[8];
  const double tr = R11 + R22 + R33;
  const double PI = 3.14159265358979323846;
  if (tr + 1.0 < 1e-3) {
    double W, Q1, Q2, Q3;
    int perm;
    if (R33 > R22 && R33 > R11) {
      W = R21 - R12; Q1 = 2.0 + 2.0 * R33; Q2 = R31 + R13; Q3 = R23 + R32; perm = 0;
    } else if (R22 > R11) {
      W = R13 - R31; Q1 = 2.0 + 2.0 * R22; Q2 = R23 + R32; Q3 = R12 + R21; perm = 1;
    } else {
      W = R32 - R23; Q1 = 2.0 + 2.0 * R11; Q2 = R12 + R21; Q3 = R31 + R13; perm = 2;
    }
    const double r = sqrt(Q1);
    const double one_over_r = 1 / r;
    const double norm = sqrt(Q1 * Q1 + Q2 * Q2 + Q3 * Q3 + W * W);
    const double sgn_w = W < 0 ? -1.0 : 1.0;
    const double mag = PI - (2 * sgn_w * W) / norm;
    const double scale = 0.5 * one_over_r * mag;
    const double s = sgn_w * scale;
    if (perm == 0) return {s * Q2, s * Q3, s * Q1};
    if (perm == 1) return {s * Q3, s * Q1, s * Q2};
    return {s * Q1, s * Q2, s * Q3};
  }
  double magnitude;
  const double tr_3 = tr - 3.0;
  if (tr_3 < -1e-6) {
    const double theta = acos((tr - 1.0) / 2.0);
    magnitude = theta / (2.0 * sin(theta));
  } else {
    magnitude = 0.5 - tr_3 / 12.0 + tr_3 * tr_3 / 60.0;
  }
  return {magnitude * (R32 - R23), magnitude * (R13 - R31), magnitude * (R21 - R12)};
}

// Pose3::Logmap gtsam/geometry/Pose3.cpp:258-278
__device__ inline void pose3_logmap(const P3& p, double* out) {
  const D3 w = so3_logmap(p.R);
  const D3 T = p.t;
  const double t = sqrt(dot3(w, w));
  out[0] = w.x; out[1] = w.y; out[2] = w.z;
  if (t < 1e-10) {
    out[3] = T.x; out[4] = T.y; out[5] = T.z;
  } else {
    const D3 a{w.x / t, w.y / t, w.z / t};
    const double Tan = tan(0.5 * t);
    const D3 WT = cross3(a, T);
    const D3 WWT = cross3(a, WT);
    const D3 u = T - (0.5 * t) * WT + (1 - t / (2. * Tan)) * WWT;
    out[3] = u.x; out[4] = u.y; out[5] = u.z;
  }
}

// Pose2 as (x, y, c, s)
struct P2 {
  double x, y, c, s;
};
__device__ __forceinline__ P2 pose2_from(double x, double y, double th) { return {x, y, cos(th), sin(th)}; }
// Rot2::normalize gtsam/geometry/Rot2.cpp:56-64 ; Pose2::operator* gtsam/geometry/Pose2.h:141-143
__device__ __forceinline__ P2 compose2(const P2& a, const P2& b) {
  P2 r;
  r.c = a.c * b.c - a.s * b.s;
  r.s = a.s * b.c + a.c * b.s;
  double scale = r.c * r.c + r.s * r.s;
  if (fabs(scale - 1.0) > 1e-10) {
    scale = 1 / sqrt(scale);
    r.c *= scale;
    r.s *= scale;
  }
  r.x = a.x + (a.c * b.x - a.s * b.y);
  r.y = a.y + (a.s * b.x + a.c * b.y);
  return r;
}
// Pose2::inverse gtsam/geometry/Pose2.cpp:202-204
__device__ __forceinline__ P2 inverse2(const P2& a) {
  P2 r;
  r.c = a.c;
  r.s = -a.s;
  const double px = -a.x, py = -a.y;
  r.x = a.c * px + a.s * py;
  r.y = -a.s * px + a.c * py;
  return r;
}

}  // namespace lmgpu
