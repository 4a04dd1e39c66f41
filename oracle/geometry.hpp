// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
// CPU restatement of the geometry used on GTSAM's LM hot path.  Every function
// cites the reference file:line (relative to /root/reference) it follows.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
#pragma once
#include <cmath>
#include <cstring>
#include <limits>

namespace orc {

struct V3 { double x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// 3x3 matrix, row-major m[3*i+j]
struct M3 {
  double m[9];
  double& operator()(int i, int j) { return m[3 * i + j]; }
  double operator()(int i, int j) const { return m[3 * i + j]; }
};
inline M3 I3() { return {{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
inline M3 mul(const M3& a, const M3& b) {
  M3 c;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += a(i, k) * b(k, j);
      c(i, j) = s;
    }
  return c;
}
inline M3 transpose(const M3& a) {
  M3 c;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) c(i, j) = a(j, i);
  return c;
}
inline V3 mul(const M3& a, V3 v) {
  return {a(0, 0) * v.x + a(0, 1) * v.y + a(0, 2) * v.z, a(1, 0) * v.x + a(1, 1) * v.y + a(1, 2) * v.z,
          a(2, 0) * v.x + a(2, 1) * v.y + a(2, 2) * v.z};
}
inline M3 skew(double wx, double wy, double wz) { return {{0, -wz, wy, wz, 0, -wx, -wy, wx, 0}}; }

// ---- SO(3): gtsam/geometry/SO3.cpp:61-112 (ExpmapFunctor / DexpFunctor) ----
struct ExpmapFunctor {
  double theta2, theta;
  M3 W, WW;
  bool nearZero;
  double A, B;
  ExpmapFunctor(V3 omega, bool nearZeroApprox = false) {
    theta2 = dot(omega, omega);
    theta = std::sqrt(theta2);
    W = skew(omega.x, omega.y, omega.z);
    WW = mul(W, W);
    // init(): SO3.cpp:61-76
    nearZero = nearZeroApprox || (theta2 <= std::numeric_limits<double>::epsilon());
    if (!nearZero) {
      const double sin_theta = std::sin(theta);
      A = sin_theta / theta;
      const double s2 = std::sin(theta / 2.0);
      const double one_minus_cos = 2.0 * s2 * s2;
      B = one_minus_cos / theta2;
    } else {
      A = 1.0 - theta2 * (1.0 / 6.0);
      B = 0.5 - theta2 * (1.0 / 24.0);
    }
  }
  // SO3.cpp:95  I + A W + B WW
  M3 expmap() const {
    M3 R = I3();
    for (int i = 0; i < 9; i++) R.m[i] += A * W.m[i] + B * WW.m[i];
    return R;
  }
};

struct DexpFunctor : ExpmapFunctor {
  V3 omega;
  double C;
  DexpFunctor(V3 w, bool nearZeroApprox = false) : ExpmapFunctor(w, nearZeroApprox), omega(w) {
    // SO3.cpp:97-112 (only C is needed by applyLeftJacobian without Jacobians)
    if (!nearZero)
      C = (1 - A) / theta2;
    else
      C = (1.0 / 6.0) - theta2 * (1.0 / 120.0);
  }
  // SO3.cpp:165-174 applyLeftJacobian (value only): v + B (w x v) + C (w x (w x v))
  V3 applyLeftJacobian(V3 v) const {
    V3 Wv = cross(omega, v);
    V3 WWv = cross(omega, Wv);
    return v + B * Wv + C * WWv;
  }
};

// SO3::Logmap, gtsam/geometry/SO3.cpp:299-375
inline V3 so3_logmap(const M3& R) {
  const double R11 = R(0, 0), R12 = R(0, 1), R13 = R(0, 2);
  const double R21 = R(1, 0), R22 = R(1, 1), R23 = R(1, 2);
  const double R31 = R(2, 0), R32 = R(2, 1), R33 = R(2, 2);
  const double tr = R11 + R22 + R33;
  V3 omega;
  if (tr + 1.0 < 1e-3) {
    if (R33 > R22 && R33 > R11) {
      const double W = R21 - R12, Q1 = 2.0 + 2.0 * R33, Q2 = R31 + R13, Q3 = R23 + R32;
      const double r = std::sqrt(Q1), one_over_r = 1 / r;
      const double norm = std::sqrt(Q1 * Q1 + Q2 * Q2 + Q3 * Q3 + W * W);
      const double sgn_w = W < 0 ? -1.0 : 1.0;
      const double mag = M_PI - (2 * sgn_w * W) / norm;
      const double scale = 0.5 * one_over_r * mag;
      omega = (sgn_w * scale) * V3{Q2, Q3, Q1};
    } else if (R22 > R11) {
      const double W = R13 - R31, Q1 = 2.0 + 2.0 * R22, Q2 = R23 + R32, Q3 = R12 + R21;
      const double r = std::sqrt(Q1), one_over_r = 1 / r;
      const double norm = std::sqrt(Q1 * Q1 + Q2 * Q2 + Q3 * Q3 + W * W);
      const double sgn_w = W < 0 ? -1.0 : 1.0;
      const double mag = M_PI - (2 * sgn_w * W) / norm;
      const double scale = 0.5 * one_over_r * mag;
      omega = (sgn_w * scale) * V3{Q3, Q1, Q2};
    } else {
      const double W = R32 - R23, Q1 = 2.0 + 2.0 * R11, Q2 = R12 + R21, Q3 = R31 + R13;
      const double r = std::sqrt(Q1), one_over_r = 1 / r;
      const double norm = std::sqrt(Q1 * Q1 + Q2 * Q2 + Q3 * Q3 + W * W);
      const double sgn_w = W < 0 ? -1.0 : 1.0;
      const double mag = M_PI - (2 * sgn_w * W) / norm;
      const double scale = 0.5 * one_over_r * mag;
      omega = (sgn_w * scale) * V3{Q1, Q2, Q3};
    }
  } else {
    double magnitude;
    const double tr_3 = tr - 3.0;
    if (tr_3 < -1e-6) {
      double theta = std::acos((tr - 1.0) / 2.0);
      magnitude = theta / (2.0 * std::sin(theta));
    } else {
      magnitude = 0.5 - tr_3 / 12.0 + tr_3 * tr_3 / 60.0;
    }
    omega = magnitude * V3{R32 - R23, R13 - R31, R21 - R12};
  }
  return omega;
}

// ---- Pose3 (R row-major, t) ----
struct Pose3 {
  M3 R;
  V3 t;
};
// Pose3::operator* (compose): R1 R2, t1 + R1 t2  (gtsam/geometry/Pose3.h operator*)
inline Pose3 compose(const Pose3& a, const Pose3& b) { return {mul(a.R, b.R), a.t + mul(a.R, b.t)}; }
// Pose3::inverse, gtsam/geometry/Pose3.cpp:61-64
inline Pose3 inverse(const Pose3& a) {
  M3 Rt = transpose(a.R);
  return {Rt, mul(Rt, V3{-a.t.x, -a.t.y, -a.t.z})};
}
// Pose3::AdjointMap, gtsam/geometry/Pose3.cpp:69-75 -> 6x6 row-major [R 0; [t]x R, R]
inline void adjointMap(const Pose3& p, double adj[36]) {
  M3 A = mul(skew(p.t.x, p.t.y, p.t.z), p.R);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      adj[6 * i + j] = p.R(i, j);
      adj[6 * i + 3 + j] = 0.0;
      adj[6 * (i + 3) + j] = A(i, j);
      adj[6 * (i + 3) + 3 + j] = p.R(i, j);
    }
}
// Pose3::Expmap, gtsam/geometry/Pose3.cpp:217-255 (value only)
inline Pose3 pose3_expmap(const double xi[6]) {
  V3 w{xi[0], xi[1], xi[2]}, v{xi[3], xi[4], xi[5]};
  const bool nearZero = (dot(w, w) <= 1e-5);
  DexpFunctor local(w, nearZero);
  Pose3 T;
  T.R = local.expmap();
  T.t = local.applyLeftJacobian(v);
  return T;
}
// Pose3::Logmap, gtsam/geometry/Pose3.cpp:258-278
inline void pose3_logmap(const Pose3& p, double out[6]) {
  V3 w = so3_logmap(p.R);
  V3 T = p.t;
  const double t = std::sqrt(dot(w, w));
  if (t < 1e-10) {
    out[0] = w.x; out[1] = w.y; out[2] = w.z; out[3] = T.x; out[4] = T.y; out[5] = T.z;
  } else {
    M3 W = skew(w.x / t, w.y / t, w.z / t);
    const double Tan = std::tan(0.5 * t);
    V3 WT = mul(W, T);
    V3 u = T - (0.5 * t) * WT + (1 - t / (2. * Tan)) * mul(W, WT);
    out[0] = w.x; out[1] = w.y; out[2] = w.z; out[3] = u.x; out[4] = u.y; out[5] = u.z;
  }
}
// LieGroup::retract = compose(Expmap(v))  (gtsam/base/Lie.h:131-133, GTSAM_POSE3_EXPMAP=ON)
inline Pose3 pose3_retract(const Pose3& p, const double xi[6]) { return compose(p, pose3_expmap(xi)); }
// LieGroup::localCoordinates = Logmap(between)  (gtsam/base/Lie.h:136-138)
inline void pose3_local(const Pose3& a, const Pose3& b, double out[6]) { pose3_logmap(compose(inverse(a), b), out); }

// ---- Pose2 ---- stored as (x, y, theta); Rot2 kept as (c,s) when composing (Rot2.h:119-121 normalises)
struct Pose2 {
  double x, y, c, s;
};
inline Pose2 pose2_from(double x, double y, double th) { return {x, y, std::cos(th), std::sin(th)}; }
inline double pose2_theta(const Pose2& p) { return std::atan2(p.s, p.c); }
// Rot2::fromCosSin → normalize(), gtsam/geometry/Rot2.cpp:27-30,56-64
inline void rot2_normalize(double& c, double& s) {
  double scale = c * c + s * s;
  if (std::abs(scale - 1.0) > 1e-10) {
    scale = 1 / std::sqrt(scale);
    c *= scale;
    s *= scale;
  }
}
// Pose2::operator*, gtsam/geometry/Pose2.h:141-143
inline Pose2 compose(const Pose2& a, const Pose2& b) {
  Pose2 r;
  r.c = a.c * b.c - a.s * b.s;
  r.s = a.s * b.c + a.c * b.s;
  rot2_normalize(r.c, r.s);
  r.x = a.x + (a.c * b.x - a.s * b.y);
  r.y = a.y + (a.s * b.x + a.c * b.y);
  return r;
}
// Pose2::inverse, gtsam/geometry/Pose2.cpp:202-204 : (R^T, R^T(-t))
inline Pose2 inverse(const Pose2& a) {
  Pose2 r;
  r.c = a.c;
  r.s = -a.s;
  const double px = -a.x, py = -a.y;
  r.x = a.c * px + a.s * py;   // unrotate: R^T p
  r.y = -a.s * px + a.c * py;
  return r;
}
// Pose2::AdjointMap, gtsam/geometry/Pose2.cpp:125-135 (row-major 3x3)
inline void adjointMap(const Pose2& p, double adj[9]) {
  adj[0] = p.c; adj[1] = -p.s; adj[2] = p.y;
  adj[3] = p.s; adj[4] = p.c; adj[5] = -p.x;
  adj[6] = 0; adj[7] = 0; adj[8] = 1;
}

// ---- Cal3Bundler::uncalibrate, gtsam/geometry/Cal3Bundler.cpp:64-90 ----
struct Cal3Bundler {
  double f, k1, k2, u0, v0;
};
// Dcal 2x3 row-major, Dp 2x2 row-major
inline void cal3bundler_uncalibrate(const Cal3Bundler& K, double x, double y, double out[2], double* Dcal, double* Dp) {
  const double r = x * x + y * y;
  const double g = 1. + (K.k1 + K.k2 * r) * r;
  const double u = g * x, v = g * y;
  const double f_ = K.f;
  if (Dcal) {
    double rx = r * x, ry = r * y;
    Dcal[0] = u; Dcal[1] = f_ * rx; Dcal[2] = f_ * r * rx;
    Dcal[3] = v; Dcal[4] = f_ * ry; Dcal[5] = f_ * r * ry;
  }
  if (Dp) {
    const double a = 2. * (K.k1 + 2. * K.k2 * r);
    const double axx = a * x * x, axy = a * x * y, ayy = a * y * y;
    Dp[0] = g + axx; Dp[1] = axy; Dp[2] = axy; Dp[3] = g + ayy;
    for (int i = 0; i < 4; i++) Dp[i] *= f_;
  }
  out[0] = K.u0 + f_ * u;
  out[1] = K.v0 + f_ * v;
}

// PinholeBase::project2 with Dpose (2x6) / Dpoint (2x3), row-major.
// gtsam/geometry/CalibratedCamera.cpp:27-46,88-94,116-135; Pose3::transformTo Pose3.cpp:413-430
// returns false on cheirality (q.z <= 0) -- GTSAM_THROW_CHEIRALITY_EXCEPTION=ON
inline bool pinhole_project2(const Pose3& pose, V3 point, double pn[2], double* Dpose, double* Dpoint) {
  M3 Rt = transpose(pose.R);
  V3 q = mul(Rt, point - pose.t);
  if (q.z <= 0) return false;
  const double d = 1.0 / q.z;
  const double u = q.x * d, v = q.y * d;
  pn[0] = u;
  pn[1] = v;
  if (Dpose) {
    double uv = u * v, uu = u * u, vv = v * v;
    const double D[12] = {uv, -1 - uu, v, -d, 0, d * u, 1 + vv, -uv, -u, 0, -d, d * v};
    std::memcpy(Dpose, D, sizeof(D));
  }
  if (Dpoint) {
    Dpoint[0] = d * (Rt(0, 0) - u * Rt(2, 0)); Dpoint[1] = d * (Rt(0, 1) - u * Rt(2, 1)); Dpoint[2] = d * (Rt(0, 2) - u * Rt(2, 2));
    Dpoint[3] = d * (Rt(1, 0) - v * Rt(2, 0)); Dpoint[4] = d * (Rt(1, 1) - v * Rt(2, 1)); Dpoint[5] = d * (Rt(1, 2) - v * Rt(2, 2));
  }
  return true;
}

}  // namespace orc
