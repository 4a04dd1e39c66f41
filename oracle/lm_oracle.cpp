// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement (plain C++17, single thread, no Eigen) of the reference's
// Levenberg-Marquardt inner loop: NonlinearFactorGraph::linearize / error,
// eliminateMultifrontal with EliminateCholesky, GaussianBayesTree::optimize,
// Values::retract and LevenbergMarquardtOptimizer::iterate/tryLambda.
// Citations are file:line relative to /root/reference.
//
// Parity pinning: checked in tests/ against the reference's own known answers
// (gtsam/base/tests/testCholesky.cpp, gtsam/linear/tests/testHessianFactor.cpp,
// tests/testGeneralSFMFactorB.cpp, gtsam/geometry/tests/testCal3Bundler.cpp, ...).
// The reference itself is unbuildable here without its CMake-generated headers
// (gtsam/config.h, gtsam/dllexport.h); only the vendored plain-C
// CCOLAMD / METIS sources are compiled into oracle/_ref (oracle/Makefile).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
#include <algorithm>
#include <atomic>
#include <cassert>
#include <cmath>
#include <functional>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <list>
#include <malloc.h>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <stdexcept>
#include <thread>
#include <vector>

#include "geometry.hpp"

namespace orc {

typedef uint64_t Key;

// Column-major dense matrix (gtsam/base/Matrix.h:39: Eigen::MatrixXd is column-major)
struct Mat {
  int r = 0, c = 0;
  std::vector<double> a;
  Mat() {}
  Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
  double& operator()(int i, int j) { return a[(size_t)j * r + i]; }
  double operator()(int i, int j) const { return a[(size_t)j * r + i]; }
};

enum VarType { POSE2 = 0, POSE3 = 1, POINT3 = 2, CAM_BUNDLER = 3, POINT2 = 4, CAL3_S2 = 5 };
static const int kVarDim[6] = {3, 6, 3, 9, 2, 5};
static const int kVarStore[6] = {3, 12, 3, 17, 2, 5};  // packed value doubles (oracle keeps u0,v0 with the camera); CAL3_S2 = (fx, fy, s, u0, v0)

enum FactorType {
  F_SFM = 0,            // GeneralSFMFactor<PinholeCamera<Cal3Bundler>,Point3>  (cam, point), meas 2
  F_BETWEEN_POSE2 = 1,  // BetweenFactor<Pose2>  meas 3 (x,y,theta)
  F_BETWEEN_POSE3 = 2,  // BetweenFactor<Pose3>  meas 12 (R row-major, t)
  F_PRIOR_POSE2 = 3,
  F_PRIOR_POSE3 = 4,
  F_PRIOR_POINT3 = 5,
  F_PRIOR_CAM = 6,      // PriorFactor<PinholeCamera<Cal3Bundler>>  meas 17
  F_PROJECTION = 7,     // GenericProjectionFactor<Pose3,Point3,Cal3_S2> meas 2 + K(fx,fy,s,u0,v0)
  F_PROJECTION_BPS = 8, // the same with body_P_sensor: meas 2 + K 5 + sensor pose (R row-major 9, t 3)
  F_BEARING_RANGE_2D = 9,  // BearingRangeFactor<Pose2,Point2>  (pose, landmark), meas 2 (bearing angle, range)
  F_SFM2 = 10,             // GeneralSFMFactor2<Cal3_S2>  (pose, point, calibration), meas 2   gtsam/slam/GeneralSFMFactor.h:208-262
  F_PRIOR_CAL3_S2 = 11     // PriorFactor<Cal3_S2>  meas 5 (fx, fy, s, u0, v0)
};
static const int kNumFactorTypes = 12;
static const int kFactorArity[12] = {2, 2, 2, 1, 1, 1, 1, 2, 2, 2, 3, 1};
static const int kFactorRows[12] = {2, 3, 6, 3, 6, 3, 9, 2, 2, 2, 2, 5};
static const int kFactorMeas[12] = {2, 3, 12, 3, 12, 3, 17, 7, 19, 2, 2, 5};

enum NoiseKind { N_UNIT = 0, N_ISO = 1, N_DIAG = 2, N_GAUSS = 3 };

struct Value {
  int type;
  double v[17];
};

struct Factor {
  int type;
  Key keys[3];
  double meas[19];
  int noise_kind;
  std::vector<double> noise;  // ISO: sigma ; DIAG: sigmas[m] ; GAUSS: R m x m row-major (sqrt information)
  int robust = 0;             // noiseModel::Robust around the Gaussian model: m-estimator id (0 = none), see robust_weight
  double rk = 0.0;            // its tuning constant
};

// Linear (Gaussian) factor: either Jacobian [A1 A2 .. b] (already whitened) or Hessian
struct GFactor {
  std::vector<Key> keys;
  std::vector<int> dims;
  bool hessian = false;
  Mat Ab;    // Jacobian: m x (sum dims + 1)
  Mat info;  // Hessian: (sum dims + 1)^2, upper triangle valid
  bool empty() const { return keys.empty(); }
};

typedef std::map<Key, Value> Values;
typedef std::map<Key, std::vector<double>> VectorValues;

// ------------------------------------------------------------------ noise
// whiten a vector / rows of a matrix: Gaussian R*v (NoiseModel.cpp:160-186), Diagonal v.*invsigmas
// (:314-332), Isotropic v*invsigma (:616-665), Unit identity.
static void whiten_rows(const Factor& f, int m, double* data, int ncols, int ld /*col-major ld*/) {
  if (f.noise_kind == N_UNIT) return;
  if (f.noise_kind == N_ISO) {
    const double inv = 1.0 / f.noise[0];
    for (int j = 0; j < ncols; j++)
      for (int i = 0; i < m; i++) data[(size_t)j * ld + i] *= inv;
  } else if (f.noise_kind == N_DIAG) {
    for (int j = 0; j < ncols; j++)
      for (int i = 0; i < m; i++) data[(size_t)j * ld + i] *= (1.0 / f.noise[i]);
  } else {
    std::vector<double> tmp(m);
    for (int j = 0; j < ncols; j++) {
      double* col = data + (size_t)j * ld;
      for (int i = 0; i < m; i++) {
        double s = 0;
        for (int k = 0; k < m; k++) s += f.noise[(size_t)i * m + k] * col[k];
        tmp[i] = s;
      }
      for (int i = 0; i < m; i++) col[i] = tmp[i];
    }
  }
}

// ------------------------------------------------------------------ values helpers
static Pose3 as_pose3(const double* v) {
  Pose3 p;
  std::memcpy(p.R.m, v, 9 * sizeof(double));
  p.t = {v[9], v[10], v[11]};
  return p;
}
static void store_pose3(const Pose3& p, double* v) {
  std::memcpy(v, p.R.m, 9 * sizeof(double));
  v[9] = p.t.x; v[10] = p.t.y; v[11] = p.t.z;
}

// Values::retract per variable (gtsam/nonlinear/Values.cpp:53-64,99-101)
static Value retract(const Value& x, const double* d) {
  Value r = x;
  switch (x.type) {
    case POSE2: {
      // Pose2 retract = compose(Pose2(v0,v1,v2))  gtsam/geometry/Pose2.cpp:100-110
      Pose2 a = pose2_from(x.v[0], x.v[1], x.v[2]);
      Pose2 b = pose2_from(d[0], d[1], d[2]);
      Pose2 c = compose(a, b);
      r.v[0] = c.x; r.v[1] = c.y; r.v[2] = pose2_theta(c);
      break;
    }
    case POSE3: {
      store_pose3(pose3_retract(as_pose3(x.v), d), r.v);
      break;
    }
    case POINT3:
      for (int i = 0; i < 3; i++) r.v[i] = x.v[i] + d[i];
      break;
    case POINT2:
      for (int i = 0; i < 2; i++) r.v[i] = x.v[i] + d[i];
      break;
    case CAL3_S2:  // Cal3_S2::retract gtsam/geometry/Cal3_S2.h:113-115: Cal3_S2(vector() + d)
      for (int i = 0; i < 5; i++) r.v[i] = x.v[i] + d[i];
      break;
    case CAM_BUNDLER: {
      // PinholeCamera::retract gtsam/geometry/PinholeCamera.h:197-203; Cal3Bundler.h:134-136
      store_pose3(pose3_retract(as_pose3(x.v), d), r.v);
      r.v[12] = x.v[12] + d[6];
      r.v[13] = x.v[13] + d[7];
      r.v[14] = x.v[14] + d[8];
      break;
    }
  }
  return r;
}

// ------------------------------------------------------------------ factors: unwhitened error + Jacobians
// H1, H2, H3: row-major m x dim (nullptr to skip).  Returns error e (size m).
static void evaluate_error(const Factor& f, const Values& vals, double* e, double* H1, double* H2, double* H3 = nullptr) {
  switch (f.type) {
    case F_SFM2: {
      // GeneralSFMFactor2::evaluateError gtsam/slam/GeneralSFMFactor.h:245-262: PinholeCamera<Cal3_S2>(pose, calib).project(point, H1, H2, H3)
      // - measured; the CheiralityException is caught: zero Jacobians and a ZERO error (:251-260).  PinholeCamera::project
      // (gtsam/geometry/PinholeCamera.h:228-240 via PinholeBaseK): pn = project2(pose, point), pi = K.uncalibrate(pn, Dcal, Dpi_pn),
      // Dpose = Dpi_pn * Dpn_pose, Dpoint = Dpi_pn * Dpn_point.  Cal3_S2::uncalibrate gtsam/geometry/Cal3_S2.cpp:44-50:
      // Dcal = [x 0 y 1 0; 0 y 0 0 1], Dp = [fx s; 0 fy].
      const Value& po = vals.at(f.keys[0]);
      const Value& pt = vals.at(f.keys[1]);
      const Value& kv = vals.at(f.keys[2]);
      const Pose3 pose = as_pose3(po.v);
      const double fx = kv.v[0], fy = kv.v[1], s = kv.v[2], u0 = kv.v[3], v0 = kv.v[4];
      double pn[2], Dpose[12], Dpoint[6];
      bool ok = pinhole_project2(pose, V3{pt.v[0], pt.v[1], pt.v[2]}, pn, H1 ? Dpose : nullptr, H2 ? Dpoint : nullptr);
      if (!ok) {
        if (H1) std::memset(H1, 0, 12 * sizeof(double));
        if (H2) std::memset(H2, 0, 6 * sizeof(double));
        if (H3) std::memset(H3, 0, 10 * sizeof(double));
        e[0] = e[1] = 0.0;
        return;
      }
      if (H1)
        for (int j = 0; j < 6; j++) {
          H1[j] = fx * Dpose[j] + s * Dpose[6 + j];
          H1[6 + j] = fy * Dpose[6 + j];
        }
      if (H2)
        for (int j = 0; j < 3; j++) {
          H2[j] = fx * Dpoint[j] + s * Dpoint[3 + j];
          H2[3 + j] = fy * Dpoint[3 + j];
        }
      if (H3) {
        const double D[10] = {pn[0], 0.0, pn[1], 1.0, 0.0, 0.0, pn[1], 0.0, 0.0, 1.0};
        std::memcpy(H3, D, sizeof(D));
      }
      e[0] = fx * pn[0] + s * pn[1] + u0 - f.meas[0];
      e[1] = fy * pn[1] + v0 - f.meas[1];
      return;
    }
    case F_PRIOR_CAL3_S2: {
      // PriorFactor::evaluateError gtsam/nonlinear/PriorFactor.h:98-102 with Cal3_S2::localCoordinates (Cal3_S2.h:118-119): -(prior - x), H = I
      const Value& a = vals.at(f.keys[0]);
      for (int i = 0; i < 5; i++) e[i] = -(f.meas[i] - a.v[i]);
      if (H1) {
        std::memset(H1, 0, 25 * sizeof(double));
        for (int i = 0; i < 5; i++) H1[6 * i] = 1.0;
      }
      return;
    }
    case F_SFM: {
      // GeneralSFMFactor::evaluateError gtsam/slam/GeneralSFMFactor.h:127-138
      const Value& cam = vals.at(f.keys[0]);
      const Value& pt = vals.at(f.keys[1]);
      Pose3 pose = as_pose3(cam.v);
      Cal3Bundler K{cam.v[12], cam.v[13], cam.v[14], cam.v[15], cam.v[16]};
      double pn[2], Dpose[12], Dpoint[6];
      bool ok = pinhole_project2(pose, V3{pt.v[0], pt.v[1], pt.v[2]}, pn, H1 ? Dpose : nullptr, H2 || H1 ? Dpoint : nullptr);
      if (!ok) {
        if (H1) std::memset(H1, 0, 18 * sizeof(double));
        if (H2) std::memset(H2, 0, 6 * sizeof(double));
        e[0] = e[1] = 0.0;
        return;
      }
      double pi[2], Dcal[6], Dp[4];
      cal3bundler_uncalibrate(K, pn[0], pn[1], pi, H1 ? Dcal : nullptr, (H1 || H2) ? Dp : nullptr);
      // chain rule, gtsam/geometry/PinholePose.h:103-106; [Dpose Dcal] gtsam/geometry/PinholeCamera.h:228-240
      if (H1) {
        for (int i = 0; i < 2; i++) {
          for (int j = 0; j < 6; j++) H1[9 * i + j] = Dp[2 * i] * Dpose[j] + Dp[2 * i + 1] * Dpose[6 + j];
          for (int j = 0; j < 3; j++) H1[9 * i + 6 + j] = Dcal[3 * i + j];
        }
      }
      if (H2) {
        for (int i = 0; i < 2; i++)
          for (int j = 0; j < 3; j++) H2[3 * i + j] = Dp[2 * i] * Dpoint[j] + Dp[2 * i + 1] * Dpoint[3 + j];
      }
      e[0] = pi[0] - f.meas[0];
      e[1] = pi[1] - f.meas[1];
      return;
    }
    case F_PROJECTION:
    case F_PROJECTION_BPS: {
      // GenericProjectionFactor::evaluateError gtsam/slam/ProjectionFactor.h:138-165; with body_P_sensor (:142-149) the camera is
      // pose.compose(body_P_sensor) and H1 is chained with H0 = d compose / d pose = AdjointMap(body_P_sensor^-1) (Lie.h:63-69)
      const Value& po = vals.at(f.keys[0]);
      const Value& pt = vals.at(f.keys[1]);
      Pose3 pose = as_pose3(po.v);
      double H0[36];
      const bool bps = f.type == F_PROJECTION_BPS;
      if (bps) {
        const Pose3 sensor = as_pose3(f.meas + 7);
        pose = compose(pose, sensor);
        adjointMap(inverse(sensor), H0);
      }
      const double fx = f.meas[2], fy = f.meas[3], s = f.meas[4], u0 = f.meas[5], v0 = f.meas[6];
      double pn[2], Dpose[12], Dpoint[6];
      bool ok = pinhole_project2(pose, V3{pt.v[0], pt.v[1], pt.v[2]}, pn, H1 ? Dpose : nullptr, H2 ? Dpoint : nullptr);
      if (!ok) {
        if (H1) std::memset(H1, 0, 12 * sizeof(double));
        if (H2) std::memset(H2, 0, 6 * sizeof(double));
        e[0] = e[1] = 2.0 * fx;
        return;
      }
      // Cal3_S2::uncalibrate gtsam/geometry/Cal3_S2.cpp:44-50 ; Dp = [fx s; 0 fy]
      if (H1) {
        double Hc[12];
        for (int j = 0; j < 6; j++) {
          Hc[j] = fx * Dpose[j] + s * Dpose[6 + j];
          Hc[6 + j] = fy * Dpose[6 + j];
        }
        for (int r = 0; r < 2; r++)
          for (int j = 0; j < 6; j++) {
            if (!bps) {
              H1[6 * r + j] = Hc[6 * r + j];
            } else {
              double v = 0;
              for (int k = 0; k < 6; k++) v += Hc[6 * r + k] * H0[6 * k + j];
              H1[6 * r + j] = v;
            }
          }
      }
      if (H2)
        for (int j = 0; j < 3; j++) {
          H2[j] = fx * Dpoint[j] + s * Dpoint[3 + j];
          H2[3 + j] = fy * Dpoint[3 + j];
        }
      e[0] = fx * pn[0] + s * pn[1] + u0 - f.meas[0];
      e[1] = fy * pn[1] + v0 - f.meas[1];
      return;
    }
    case F_BETWEEN_POSE3: {
      // BetweenFactor::evaluateError gtsam/slam/BetweenFactor.h:111-124 (fast path: Local Jacobian NOT applied)
      // LieGroup::between gtsam/base/Lie.h:63-69 : H1 = -Ad(h^-1), H2 = I
      Pose3 p1 = as_pose3(vals.at(f.keys[0]).v), p2 = as_pose3(vals.at(f.keys[1]).v);
      Pose3 h = compose(inverse(p1), p2);
      if (H1) {
        double adj[36];
        adjointMap(inverse(h), adj);
        for (int i = 0; i < 36; i++) H1[i] = -adj[i];
      }
      if (H2) {
        std::memset(H2, 0, 36 * sizeof(double));
        for (int i = 0; i < 6; i++) H2[7 * i] = 1.0;
      }
      Pose3 z = as_pose3(f.meas);
      pose3_local(z, h, e);
      return;
    }
    case F_BETWEEN_POSE2: {
      const Value& a = vals.at(f.keys[0]);
      const Value& b = vals.at(f.keys[1]);
      Pose2 p1 = pose2_from(a.v[0], a.v[1], a.v[2]), p2 = pose2_from(b.v[0], b.v[1], b.v[2]);
      Pose2 h = compose(inverse(p1), p2);
      if (H1) {
        double adj[9];
        adjointMap(inverse(h), adj);
        for (int i = 0; i < 9; i++) H1[i] = -adj[i];
      }
      if (H2) {
        std::memset(H2, 0, 9 * sizeof(double));
        H2[0] = H2[4] = H2[8] = 1.0;
      }
      // Local(measured, h) = ChartAtOrigin::Local(measured^-1 h) = (x, y, theta)  Pose2.cpp:112-121
      Pose2 z = pose2_from(f.meas[0], f.meas[1], f.meas[2]);
      Pose2 d = compose(inverse(z), h);
      e[0] = d.x; e[1] = d.y; e[2] = pose2_theta(d);
      return;
    }
    case F_BEARING_RANGE_2D: {
      // BearingRangeFactor<Pose2,Point2> gtsam/sam/BearingRangeFactor.h:33-77 = ExpressionFactorN over BearingRange::Measure;
      // unwhitenedError = -Local(value, measured) (gtsam/nonlinear/ExpressionFactor.h:104-115) with the Jacobians of the value:
      // Pose2::bearing gtsam/geometry/Pose2.cpp:260-271 (transformTo :222-229, Rot2::unrotate, Rot2::relativeBearing
      // gtsam/geometry/Rot2.cpp:134-145) and Pose2::range Pose2.cpp:285-299 (norm2 gtsam/geometry/Point2.cpp:27-36).
      // (The reference orders the factor's keys by Key value; the oracle keeps (pose, landmark) -- the blocks are the same.)
      const Value& a = vals.at(f.keys[0]);
      const Value& l = vals.at(f.keys[1]);
      const double c = std::cos(a.v[2]), sn = std::sin(a.v[2]);
      const double dx = l.v[0] - a.v[0], dy = l.v[1] - a.v[1];
      const double qx = c * dx + sn * dy, qy = -sn * dx + c * dy;  // transformTo
      const double d2 = qx * qx + qy * qy, n = std::sqrt(d2);
      double Hb[2] = {0.0, 0.0}, cb = 1.0, sb = 0.0;  // relativeBearing
      if (std::abs(n) > 1e-5) {
        Hb[0] = -qy / d2;
        Hb[1] = qx / d2;
        cb = qx / n;
        sb = qy / n;
      }
      // bearing row: Hb * [-1 0 qy; 0 -1 -qx]  and  Hb * R^T
      const double r = std::sqrt(dx * dx + dy * dy);
      double Hr[2] = {1.0, 1.0};
      if (std::abs(r) > 1e-10) {
        Hr[0] = dx / r;
        Hr[1] = dy / r;
      }
      if (H1) {
        H1[0] = -Hb[0];
        H1[1] = -Hb[1];
        H1[2] = Hb[0] * qy - Hb[1] * qx;
        H1[3] = Hr[0] * -c + Hr[1] * -sn;
        H1[4] = Hr[0] * sn + Hr[1] * -c;
        H1[5] = 0.0;
      }
      if (H2) {
        H2[0] = Hb[0] * c + Hb[1] * -sn;
        H2[1] = Hb[0] * sn + Hb[1] * c;
        H2[2] = Hr[0];
        H2[3] = Hr[1];
      }
      // -Local(value, measured): Rot2 Local = Logmap(value^-1 * measured) = theta of (cb, sb)^-1 * (cm, sm)
      const double cm = std::cos(f.meas[0]), sm = std::sin(f.meas[0]);
      e[0] = -std::atan2(cb * sm - sb * cm, cb * cm + sb * sm);
      e[1] = r - f.meas[1];
      return;
    }
    case F_PRIOR_POSE2: {
      // PriorFactor::evaluateError gtsam/nonlinear/PriorFactor.h:98-102 : -Local(x, prior), H = I
      const Value& a = vals.at(f.keys[0]);
      Pose2 x = pose2_from(a.v[0], a.v[1], a.v[2]);
      Pose2 z = pose2_from(f.meas[0], f.meas[1], f.meas[2]);
      Pose2 d = compose(inverse(x), z);
      e[0] = -d.x; e[1] = -d.y; e[2] = -pose2_theta(d);
      if (H1) {
        std::memset(H1, 0, 9 * sizeof(double));
        H1[0] = H1[4] = H1[8] = 1.0;
      }
      return;
    }
    case F_PRIOR_POSE3: {
      Pose3 x = as_pose3(vals.at(f.keys[0]).v), z = as_pose3(f.meas);
      pose3_local(x, z, e);
      for (int i = 0; i < 6; i++) e[i] = -e[i];
      if (H1) {
        std::memset(H1, 0, 36 * sizeof(double));
        for (int i = 0; i < 6; i++) H1[7 * i] = 1.0;
      }
      return;
    }
    case F_PRIOR_POINT3: {
      const Value& a = vals.at(f.keys[0]);
      for (int i = 0; i < 3; i++) e[i] = -(f.meas[i] - a.v[i]);
      if (H1) {
        std::memset(H1, 0, 9 * sizeof(double));
        H1[0] = H1[4] = H1[8] = 1.0;
      }
      return;
    }
    case F_PRIOR_CAM: {
      // PinholeCamera::localCoordinates gtsam/geometry/PinholeCamera.h:206-211
      const Value& a = vals.at(f.keys[0]);
      Pose3 x = as_pose3(a.v), z = as_pose3(f.meas);
      pose3_local(x, z, e);
      e[6] = f.meas[12] - a.v[12];
      e[7] = f.meas[13] - a.v[13];
      e[8] = f.meas[14] - a.v[14];
      for (int i = 0; i < 9; i++) e[i] = -e[i];
      if (H1) {
        std::memset(H1, 0, 81 * sizeof(double));
        for (int i = 0; i < 9; i++) H1[10 * i] = 1.0;
      }
      return;
    }
  }
  throw std::runtime_error("unknown factor type");
}

// m-estimators, gtsam/linear/LossFunctions.cpp: weight(distance) and loss(distance) of
// 1 Fair :146-155, 2 Huber :179-191, 3 Cauchy :217-224, 4 Tukey :250-266, 5 Welsch :289-297, 6 GemanMcClure :320-331,
// 7 DCS :354-373, 8 L2WithDeadZone :400-412; distance = ||whitened error|| >= 0 (noiseModel::Robust, NoiseModel.h:717-725)
static double robust_weight(int kind, double k, double d) {
  switch (kind) {
    case 1: return 1.0 / (1.0 + std::abs(d) / k);
    case 2: return (std::abs(d) <= k) ? 1.0 : k / std::abs(d);
    case 3: return (k * k) / (k * k + d * d);
    case 4: {
      if (std::abs(d) <= k) {
        const double t = 1.0 - d * d / (k * k);
        return t * t;
      }
      return 0.0;
    }
    case 5: return std::exp(-(d * d) / (k * k));
    case 6: {
      const double c2 = k * k, c4 = c2 * c2, c2e = c2 + d * d;
      return c4 / (c2e * c2e);
    }
    case 7: {
      const double e2 = d * d;
      if (e2 > k) {
        const double w = 2.0 * k / (k + e2);
        return w * w;
      }
      return 1.0;
    }
    case 8: return (std::abs(d) <= k) ? 0.0 : (d > k ? (-k + d) / d : (k + d) / d);
    default: return 1.0;
  }
}
static double robust_loss(int kind, double k, double d) {
  const double a = std::abs(d);
  switch (kind) {
    case 1: return k * k * (a / k - std::log1p(a / k));
    case 2: return (a <= k) ? d * d / 2 : k * (a - k / 2);
    case 3: return k * k * std::log1p(d * d / (k * k)) * 0.5;
    case 4: {
      if (a <= k) {
        const double t = 1.0 - d * d / (k * k);
        return k * k * (1 - t * t * t) / 6.0;
      }
      return k * k / 6.0;
    }
    case 5: return k * k * 0.5 * -std::expm1(-(d * d) / (k * k));
    case 6: return 0.5 * (k * k * d * d) / (k * k + d * d);
    case 7: {
      const double e2 = d * d, e4 = e2 * e2, c2 = k * k;
      return (c2 * e2 + k * e4) / ((e2 + k) * (e2 + k));
    }
    case 8: return (a < k) ? 0.0 : 0.5 * (k - a) * (k - a);
    default: return 0.5 * d * d;
  }
}

// NoiseModelFactor::error gtsam/nonlinear/NonlinearFactor.cpp:138-149: noiseModel->loss(squaredMahalanobisDistance(e)):
// 0.5 * ||whiten(e)||^2 for a Gaussian model, rho(||whiten(e)||) for noiseModel::Robust (NoiseModel.h:717-725)
static double factor_error(const Factor& f, const Values& vals) {
  const int m = kFactorRows[f.type];
  double e[9];
  evaluate_error(f, vals, e, nullptr, nullptr);
  whiten_rows(f, m, e, 1, m);
  double s = 0;
  for (int i = 0; i < m; i++) s += e[i] * e[i];
  if (f.robust) return robust_loss(f.robust, f.rk, std::sqrt(s));
  return 0.5 * s;
}

// NoiseModelFactor::linearize gtsam/nonlinear/NonlinearFactor.cpp:152-184 and
// GeneralSFMFactor::linearize gtsam/slam/GeneralSFMFactor.h:141-177 (same result, see
// gtsam/slam/tests/testGeneralSFMFactor.cpp:439-491): A = whiten(H), b = whiten(-e)
static GFactor linearize_factor(const Factor& f, const Values& vals) {
  const int m = kFactorRows[f.type];
  const int ar = kFactorArity[f.type];
  GFactor g;
  int tot = 0;
  for (int j = 0; j < ar; j++) {
    g.keys.push_back(f.keys[j]);
    g.dims.push_back(kVarDim[vals.at(f.keys[j]).type]);
    tot += g.dims.back();
  }
  double e[9], H1[81], H2[54], H3[10];
  evaluate_error(f, vals, e, H1, ar > 1 ? H2 : nullptr, ar > 2 ? H3 : nullptr);
  g.Ab = Mat(m, tot + 1);
  for (int i = 0; i < m; i++) {
    for (int j = 0; j < g.dims[0]; j++) g.Ab(i, j) = H1[i * g.dims[0] + j];
    if (ar > 1)
      for (int j = 0; j < g.dims[1]; j++) g.Ab(i, g.dims[0] + j) = H2[i * g.dims[1] + j];
    if (ar > 2)
      for (int j = 0; j < g.dims[2]; j++) g.Ab(i, g.dims[0] + g.dims[1] + j) = H3[i * g.dims[2] + j];
    g.Ab(i, tot) = -e[i];
  }
  whiten_rows(f, m, g.Ab.a.data(), tot + 1, m);
  if (f.robust) {  // Robust::WhitenSystem (NoiseModel.cpp:705-723) = whiten, then mEstimator::Base::reweight (Block scheme,
                   // LossFunctions.cpp:61-76): every entry of [A b] times sqrt(weight(||b||))
    double s = 0;
    for (int i = 0; i < m; i++) s += g.Ab(i, tot) * g.Ab(i, tot);
    const double w = std::sqrt(robust_weight(f.robust, f.rk, std::sqrt(s)));
    for (auto& x : g.Ab.a) x *= w;
  }
  return g;
}

// ------------------------------------------------------------------ dense partial Cholesky
// choleskyPartial gtsam/base/cholesky.cpp:108-159.  ABC col-major n x n, upper triangle used.
// Upper Cholesky of the frontal block (Eigen LLT<Upper> numerics up to rounding order), then
// S = R^-T B, C -= S^T S (upper only), then the pivot-exponent test.
// Small fronts: unblocked, row by row of R.  Fronts with >= 256 frontal columns: the same elimination in panels of 128 rows with a
// register-tiled rank-128 update, as Eigen's LLT does it (llt_inplace::blocked, Eigen/src/Cholesky/LLT.h: panel, triangular solve,
// rankUpdate): the unblocked form streams the whole trailing matrix once per pivot and ran the 9001-column root of the headline
// workload at 0.8 GFLOP/s, 6-7 times slower than the reference's own Eigen code on the same machine -- not a fair CPU stand-in.
static void chol_unblocked_rows(double* ABC, int ld, int k0, int k1, int upd_end, int jend, bool* ok) {
  // rows k0 .. k1-1 of R; the trailing update of every pivot is applied to rows k+1 .. upd_end-1, columns up to jend
  auto A = [&](int i, int j) -> double& { return ABC[(size_t)j * ld + i]; };
  for (int k = k0; k < k1; k++) {
    const double x = A(k, k);
    if (!(x > 0.0)) {
      if (x <= 0.0) {
        *ok = false;
        return;
      }
    }
    const double rkk = std::sqrt(x);
    A(k, k) = rkk;
    const double inv = 1.0 / rkk;
    for (int j = k + 1; j < jend; j++) A(k, j) *= inv;
    for (int j = k + 1; j < jend; j++) {
      const double rkj = A(k, j);
      if (rkj == 0.0) continue;
      double* col = &A(0, j);
      const int iend = std::min(j, upd_end - 1);
      for (int i = k + 1; i <= iend; i++) col[i] -= A(k, i) * rkj;
    }
  }
}
// C[i][j] -= sum_{k in [k0, k1)} P[k][i] P[k][j] for r0 <= i <= j < n (upper), P = rows k0..k1-1 of the same matrix.
// `pack` = the panel copied once per call into groups of four columns, [(i - r0) / 4][k][4] (4 KB per group, contiguous), so that the
// 4 x 4 register tile is two 2-wide vectors of i per column j, updated with a broadcast P[k][j] -- the shape of Eigen's SSE2 product
// kernel with its packed operands (the reference is built without -march=native: BASELINE.md section 2).
typedef double v2d __attribute__((vector_size(16)));
static void syrk_upper_panel(double* ABC, int ld, int k0, int k1, int r0, int n, const double* pack, int part = 0, int nparts = 1) {
  const int kb = k1 - k0;
  for (int j0 = r0 + 4 * part; j0 < n; j0 += 4 * nparts) {  // column tiles dealt round-robin: the work of a tile grows with its column
    const int jn = std::min(4, n - j0);
    for (int i0 = r0; i0 <= j0; i0 += 4) {
      const int in = std::min(4, n - i0);
      v2d c00 = {0, 0}, c01 = {0, 0}, c10 = {0, 0}, c11 = {0, 0}, c20 = {0, 0}, c21 = {0, 0}, c30 = {0, 0}, c31 = {0, 0};
      const double* pa = pack + (size_t)((i0 - r0) >> 2) * kb * 4;  // (the last group is padded with zeros)
      const double* pb = pack + (size_t)((j0 - r0) >> 2) * kb * 4;
      for (int k = 0; k < kb; k++, pa += 4, pb += 4) {
        v2d a0, a1;
        __builtin_memcpy(&a0, pa, 16);
        __builtin_memcpy(&a1, pa + 2, 16);
        const v2d b0 = {pb[0], pb[0]}, b1 = {pb[1], pb[1]}, b2 = {pb[2], pb[2]}, b3 = {pb[3], pb[3]};
        c00 += a0 * b0; c01 += a1 * b0;
        c10 += a0 * b1; c11 += a1 * b1;
        c20 += a0 * b2; c21 += a1 * b2;
        c30 += a0 * b3; c31 += a1 * b3;
      }
      const double acc[4][4] = {{c00[0], c00[1], c01[0], c01[1]}, {c10[0], c10[1], c11[0], c11[1]}, {c20[0], c20[1], c21[0], c21[1]},
                                {c30[0], c30[1], c31[0], c31[1]}};  // [j][i]
      for (int bj = 0; bj < jn; bj++)
        for (int ai = 0; ai < in; ai++)
          if (i0 + ai <= j0 + bj) ABC[(size_t)(j0 + bj) * ld + i0 + ai] -= acc[bj][ai];
    }
  }
}
static int chol_threads = 1;  // orc_set_dense_threads: threads of the blocked dense factorisation's updates
static bool cholesky_partial(double* ABC, int ld, int n, int nFrontal) {
  if (nFrontal == 0) return true;
  auto A = [&](int i, int j) -> double& { return ABC[(size_t)j * ld + i]; };
  bool ok = true;
  if (nFrontal < 256) {
    chol_unblocked_rows(ABC, ld, 0, nFrontal, n, n, &ok);  // right-looking over the whole matrix, pivot by pivot
  } else {
    const int NB = 128;
    std::vector<double> pack;
    for (int k0 = 0; k0 < nFrontal && ok; k0 += NB) {
      const int k1 = std::min(nFrontal, k0 + NB);
      chol_unblocked_rows(ABC, ld, k0, k1, k1, n, &ok);  // the panel's rows, complete over all columns (diagonal block + triangular solve)
      if (ok && k1 < n) {
        // all-cores leg (orc_set_threads > 1): the rank-128 update of a large trailing matrix is shared out over the threads.  This
        // goes BEYOND what the reference does (Eigen's LLT runs on one thread whatever GTSAM's TBB setting) and is stated where
        // the number is reported; with one thread this is the plain serial update.
        const int nt = (chol_threads > 1 && (long)(n - k1) * (n - k1) > 512L * 512L) ? chol_threads : 1;
        const int m = n - k1, kb = k1 - k0;
        pack.assign((size_t)kb * ((m + 3) & ~3), 0.0);
        for (int i = 0; i < m; i++)
          for (int k = k0; k < k1; k++) pack[((size_t)(i >> 2) * kb + (k - k0)) * 4 + (i & 3)] = A(k, k1 + i);
        const double* pk = pack.data();
        if (nt == 1) {
          syrk_upper_panel(ABC, ld, k0, k1, k1, n, pk);
        } else {
          std::vector<std::thread> pool;
          for (int t = 0; t < nt; t++) pool.emplace_back([=]() { syrk_upper_panel(ABC, ld, k0, k1, k1, n, pk, t, nt); });
          for (auto& th : pool) th.join();
        }
      }
    }
  }
  if (!ok) return false;
  if (nFrontal >= 2) {
    int exp2, exp1;
    (void)std::frexp(A(nFrontal - 2, nFrontal - 2), &exp2);
    (void)std::frexp(A(nFrontal - 1, nFrontal - 1), &exp1);
    return (exp2 - exp1 < 12);
  } else {
    int exp1;
    (void)std::frexp(A(0, 0), &exp1);
    return (exp1 > -12);
  }
}

// ------------------------------------------------------------------ symbolic
// VariableIndex (gtsam/inference/VariableIndex-inl.h:27-49): key -> factor indices ascending
typedef std::map<Key, std::vector<size_t>> VariableIndex;

struct ENode {
  Key key;
  std::vector<size_t> factors;              // indices into the gaussian graph
  std::vector<std::shared_ptr<ENode>> children;
};

// EliminationTree ctor gtsam/inference/EliminationTree-inst.h:78-156
static std::vector<std::shared_ptr<ENode>> build_etree(const std::vector<std::vector<Key>>& fkeys, const VariableIndex& vi,
                                                        const std::vector<Key>& order) {
  const size_t m = fkeys.size(), n = order.size();
  const size_t none = std::numeric_limits<size_t>::max();
  std::vector<std::shared_ptr<ENode>> nodes(n);
  std::vector<size_t> parents(n, none), prevCol(m, none);
  for (size_t j = 0; j < n; j++) {
    auto it = vi.find(order[j]);
    if (it == vi.end()) throw std::invalid_argument("EliminationTree: ordering contains variables not in the graph");
    auto node = std::make_shared<ENode>();
    node->key = order[j];
    for (size_t i : it->second) {
      if (prevCol[i] != none) {
        size_t r = prevCol[i];
        while (parents[r] != none) r = parents[r];
        if (r != j) {
          parents[r] = j;
          node->children.push_back(nodes[r]);
        }
      } else {
        node->factors.push_back(i);
      }
      prevCol[i] = j;
    }
    nodes[j] = node;
  }
  std::vector<std::shared_ptr<ENode>> roots;
  for (size_t j = 0; j < n; j++)
    if (parents[j] == none) roots.push_back(nodes[j]);
  return roots;
}

struct JNode {
  std::vector<Key> orderedFrontalKeys;
  std::vector<size_t> factors;
  std::vector<std::shared_ptr<JNode>> children;
};

// JunctionTree ctor + ConstructorTraversalVisitorPostAlg2 (gtsam/inference/JunctionTree-inst.h:65-153),
// Cluster::merge / mergeChildren (gtsam/inference/ClusterTree-inst.h:45-96).
// Returns the separator (symbolic "parents" of the conditional on ETree node's key) as sorted key set.
static std::vector<Key> jt_visit(const std::shared_ptr<ENode>& en, const std::vector<std::vector<Key>>& fkeys,
                                 std::shared_ptr<JNode>& out) {
  auto node = std::make_shared<JNode>();
  node->orderedFrontalKeys.push_back(en->key);
  node->factors = en->factors;
  std::vector<std::vector<Key>> childSeps;
  for (auto& ch : en->children) {
    std::shared_ptr<JNode> cj;
    childSeps.push_back(jt_visit(ch, fkeys, cj));
    node->children.push_back(cj);
  }
  // symbolic elimination of en->key over own factors + child separator factors
  std::vector<Key> all;
  for (size_t f : en->factors)
    for (Key k : fkeys[f]) all.push_back(k);
  for (auto& s : childSeps)
    for (Key k : s) all.push_back(k);
  std::sort(all.begin(), all.end());
  all.erase(std::unique(all.begin(), all.end()), all.end());
  std::vector<Key> sep;
  for (Key k : all)
    if (k != en->key) sep.push_back(k);
  const size_t myNrParents = sep.size();
  const size_t nrChildren = node->children.size();
  std::vector<bool> merge(nrChildren, false);
  size_t myNrFrontals = 1;
  for (size_t i = 0; i < nrChildren; i++) {
    if (myNrParents + myNrFrontals == childSeps[i].size()) {
      myNrFrontals += node->children[i]->orderedFrontalKeys.size();
      merge[i] = true;
    }
  }
  // mergeChildren
  auto oldChildren = node->children;
  node->children.clear();
  for (size_t i = 0; i < nrChildren; i++) {
    auto& child = oldChildren[i];
    if (merge[i]) {
      node->orderedFrontalKeys.insert(node->orderedFrontalKeys.end(), child->orderedFrontalKeys.rbegin(),
                                      child->orderedFrontalKeys.rend());
      node->factors.insert(node->factors.end(), child->factors.begin(), child->factors.end());
      node->children.insert(node->children.end(), child->children.begin(), child->children.end());
    } else {
      node->children.push_back(child);
    }
  }
  std::reverse(node->orderedFrontalKeys.begin(), node->orderedFrontalKeys.end());
  out = node;
  return sep;
}

// ------------------------------------------------------------------ numeric elimination
struct Clique {
  std::vector<Key> keys;  // frontals then separator (Scatter order)
  std::vector<int> dims;
  int nFrontal = 0;       // number of frontal keys
  Mat RSd;                // nf x (n) : [R S d], strictly-lower zeroed
  std::vector<int> children;
  int parent = -1;
};

struct BayesTree {
  std::vector<Clique> cliques;  // post-order (children before parents)
  std::vector<int> roots;
};

struct Indeterminate : std::runtime_error {
  Key key;
  Indeterminate(Key k) : std::runtime_error("IndeterminantLinearSystemException"), key(k) {}
};

// JacobianFactor::updateHessian / BinaryJacobianFactor::updateHessian / HessianFactor::updateHessian
// (gtsam/linear/JacobianFactor.cpp:586-624, BinaryJacobianFactor.h:51-83, HessianFactor.cpp:349-373):
// info(upper) += [A b]^T [A b] at the slots of this factor's keys.
static void update_hessian(const GFactor& f, const std::vector<Key>& infoKeys, const std::vector<int>& offs, Mat& info) {
  const int nv = (int)f.keys.size();
  std::vector<int> start(nv + 1), dim(nv + 1), loc(nv + 1);
  int o = 0;
  for (int j = 0; j < nv; j++) {
    size_t slot = std::find(infoKeys.begin(), infoKeys.end(), f.keys[j]) - infoKeys.begin();
    if (slot == infoKeys.size()) throw std::runtime_error("update_hessian: key not in scatter");
    start[j] = offs[slot];
    dim[j] = f.dims[j];
    loc[j] = o;
    o += f.dims[j];
  }
  start[nv] = offs[infoKeys.size()];
  dim[nv] = 1;
  loc[nv] = o;
  const int m = f.hessian ? 0 : f.Ab.r;
  for (int bj = 0; bj <= nv; bj++)
    for (int bi = 0; bi <= bj; bi++) {
      for (int cj = 0; cj < dim[bj]; cj++)
        for (int ci = 0; ci < dim[bi]; ci++) {
          if (bi == bj && ci > cj) continue;
          double v;
          if (f.hessian) {
            v = f.info(loc[bi] + ci, loc[bj] + cj);
          } else {
            v = 0;
            for (int r = 0; r < m; r++) v += f.Ab(r, loc[bi] + ci) * f.Ab(r, loc[bj] + cj);
          }
          int gi = start[bi] + ci, gj = start[bj] + cj;
          if (gi > gj) std::swap(gi, gj);  // SymmetricBlockMatrix.h:230-236 transposes into the upper triangle
          info(gi, gj) += v;
        }
    }
}

// EliminateCholesky gtsam/linear/HessianFactor.cpp:515-535 with Scatter (gtsam/linear/Scatter.cpp:39-73)
static void eliminate_clique(const std::vector<const GFactor*>& gathered, const std::vector<Key>& frontals,
                             const std::map<Key, int>& keyDim, Clique& cq, GFactor& separator) {
  // Scatter: frontal keys in the given order, then all other keys sorted
  std::vector<Key> keys = frontals, rest;
  for (auto* g : gathered)
    for (Key k : g->keys)
      if (std::find(frontals.begin(), frontals.end(), k) == frontals.end()) rest.push_back(k);
  std::sort(rest.begin(), rest.end());
  rest.erase(std::unique(rest.begin(), rest.end()), rest.end());
  keys.insert(keys.end(), rest.begin(), rest.end());
  std::vector<int> dims, offs;
  int n = 0;
  for (Key k : keys) {
    dims.push_back(keyDim.at(k));
    offs.push_back(n);
    n += dims.back();
  }
  offs.push_back(n);
  n += 1;
  static thread_local Mat info;  // reused per thread: a fresh (n x n) allocation per clique is page-fault bound (BAL: 70 KB x 100 000)
  info.r = info.c = n;
  info.a.assign((size_t)n * n, 0.0);
  for (auto* g : gathered) update_hessian(*g, keys, offs, info);
  int nf = 0;
  for (size_t i = 0; i < frontals.size(); i++) nf += dims[i];
  if (!cholesky_partial(info.a.data(), n, n, nf)) throw Indeterminate(frontals.front());
  // split (gtsam/base/SymmetricBlockMatrix.cpp:93-107)
  cq.keys = keys;
  cq.dims = dims;
  cq.nFrontal = (int)frontals.size();
  cq.RSd = Mat(nf, n);
  for (int j = 0; j < n; j++)
    for (int i = 0; i < nf && i <= j; i++) cq.RSd(i, j) = info(i, j);
  separator = GFactor();
  separator.hessian = true;
  for (size_t i = frontals.size(); i < keys.size(); i++) {
    separator.keys.push_back(keys[i]);
    separator.dims.push_back(dims[i]);
  }
  const int ns = n - nf;
  separator.info = Mat(ns, ns);
  for (int j = 0; j < ns; j++)
    for (int i = 0; i <= j; i++) separator.info(i, j) = info(nf + i, nf + j);
}

static double now_s();
// all-cores leg of the CPU baseline (orc_set_threads > 1): the subtrees below a clique with many children are eliminated by a
// pool of threads, each into a Bayes tree of its own, and spliced into the result in child order afterwards -- the same cliques
// in the same post-order.  This is the reference's TBB subtree parallelism (gtsam/base/treeTraversal/parallelTraversalTasks.h:78-93,
// threshold gtsam/inference/ClusterTree-inst.h:299-300); the dense work inside a clique stays single-threaded, as Eigen's LLT is there.
static int g_threads = 1;

static int eliminate_tree(const std::shared_ptr<JNode>& node, const std::vector<const GFactor*>& graph,
                          const std::map<Key, int>& keyDim, BayesTree& bt, GFactor& sepOut) {
  // post-order: children first (gtsam/inference/ClusterTree-inst.h:219-266)
  std::vector<GFactor> childFactors(node->children.size());
  std::vector<int> childIdx;
  if (g_threads > 1 && node->children.size() >= 64) {
    const size_t nc = node->children.size();
    std::vector<BayesTree> local(nc);
    std::vector<int> localRoot(nc, -1);
    std::atomic<size_t> next(0);
    std::exception_ptr failure;
    std::mutex mu;
    auto work = [&]() {
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= nc) return;
        try {
          localRoot[i] = eliminate_tree(node->children[i], graph, keyDim, local[i], childFactors[i]);
        } catch (...) {
          std::lock_guard<std::mutex> lk(mu);
          if (!failure) failure = std::current_exception();
        }
      }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < g_threads; t++) pool.emplace_back(work);
    for (auto& t : pool) t.join();
    if (failure) std::rethrow_exception(failure);
    for (size_t i = 0; i < nc; i++) {  // splice in child order: indices shift by the cliques already there
      const int base = (int)bt.cliques.size();
      for (Clique& c : local[i].cliques) {
        for (int& ch : c.children) ch += base;
        if (c.parent >= 0) c.parent += base;
        bt.cliques.push_back(std::move(c));
      }
      childIdx.push_back(base + localRoot[i]);
    }
  } else {
    for (size_t i = 0; i < node->children.size(); i++) childIdx.push_back(eliminate_tree(node->children[i], graph, keyDim, bt, childFactors[i]));
  }
  std::vector<const GFactor*> gathered;
  for (size_t f : node->factors) gathered.push_back(graph[f]);
  for (auto& cf : childFactors)
    if (!cf.empty()) gathered.push_back(&cf);
  Clique cq;
  const double tq = (getenv("ORC_TRACE") && node->children.size() >= 64) ? now_s() : 0.0;
  eliminate_clique(gathered, node->orderedFrontalKeys, keyDim, cq, sepOut);
  if (tq > 0.0) std::fprintf(stderr, "clique with %zu children: own elimination %.3f s\n", node->children.size(), now_s() - tq);
  cq.children = childIdx;
  int me = (int)bt.cliques.size();
  bt.cliques.push_back(std::move(cq));
  for (int c : childIdx) bt.cliques[c].parent = me;
  return me;
}

// optimizeBayesTree gtsam/linear/linearAlgorithms-inst.h:54-155
static void backsub(const BayesTree& bt, VectorValues& x) {
  for (int ci = (int)bt.cliques.size() - 1; ci >= 0; ci--) {  // reverse post-order = parents before children
    const Clique& c = bt.cliques[ci];
    const int nf = c.RSd.r, n = c.RSd.c;
    std::vector<double> rhs(nf);
    for (int i = 0; i < nf; i++) rhs[i] = c.RSd(i, n - 1);
    int col = nf;
    for (size_t k = c.nFrontal; k < c.keys.size(); k++) {
      const auto& xs = x.at(c.keys[k]);
      for (int d = 0; d < c.dims[k]; d++, col++)
        for (int i = 0; i < nf; i++) rhs[i] -= c.RSd(i, col) * xs[d];
    }
    for (int i = nf - 1; i >= 0; i--) {
      double s = rhs[i];
      for (int j = i + 1; j < nf; j++) s -= c.RSd(i, j) * rhs[j];
      rhs[i] = s / c.RSd(i, i);
    }
    for (int i = 0; i < nf; i++)
      if (std::isnan(rhs[i])) throw Indeterminate(c.keys.front());
    int o = 0;
    for (int k = 0; k < c.nFrontal; k++) {
      x[c.keys[k]] = std::vector<double>(rhs.begin() + o, rhs.begin() + o + c.dims[k]);
      o += c.dims[k];
    }
  }
}

// ------------------------------------------------------------------ problem / optimizer state
struct LMParams {
  // gtsam/nonlinear/LevenbergMarquardtParams.h:69-82 (legacy defaults)
  int maxIterations = 100;
  double relativeErrorTol = 1e-5, absoluteErrorTol = 1e-5, errorTol = 0.0;
  double lambdaInitial = 1e-5, lambdaFactor = 10.0, lambdaUpperBound = 1e5, lambdaLowerBound = 0.0;
  double minModelFidelity = 1e-3;
  int diagonalDamping = 0, useFixedLambdaFactor = 1;
  double minDiagonal = 1e-6, maxDiagonal = 1e32;
};

struct Problem {
  Values values;
  std::vector<Factor> factors;
  std::vector<Key> ordering;
  // linear state
  std::vector<GFactor> linear;  // one per nonlinear factor
  BayesTree bt;
  VectorValues delta;
  // LM state (gtsam/nonlinear/internal/LevenbergMarquardtState.h:42-157)
  double lambda = 1e-5, currentFactor = 10.0, error = 0.0;
  int iterations = 0, totalInner = 0;
  std::vector<double> trace;  // per inner iteration: lambda, newError, modelFidelity, accepted, solved
  // timing (seconds) of last iterate
  double t_linearize = 0, t_eliminate = 0, t_backsub = 0;
};

static double graph_error(const Problem& p, const Values& v) {
  // NonlinearFactorGraph::error gtsam/nonlinear/NonlinearFactorGraph.cpp:170-179 (factor index order)
  double total = 0;
  for (auto& f : p.factors) total += factor_error(f, v);
  return total;
}

static void linearize(Problem& p) {
  p.linear.clear();
  if (g_threads > 1 && p.factors.size() >= 4096) {  // NonlinearFactorGraph::linearize with TBB: parallel_for over the factors (:246-261)
    p.linear.resize(p.factors.size());
    std::atomic<size_t> next(0);
    auto work = [&]() {
      for (;;) {
        const size_t b = next.fetch_add(1024);
        if (b >= p.factors.size()) return;
        for (size_t i = b; i < std::min(p.factors.size(), b + 1024); i++) p.linear[i] = linearize_factor(p.factors[i], p.values);
      }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < g_threads; t++) pool.emplace_back(work);
    for (auto& t : pool) t.join();
    return;
  }
  p.linear.reserve(p.factors.size());
  for (auto& f : p.factors) p.linear.push_back(linearize_factor(f, p.values));
}

// GaussianFactorGraph::error gtsam/linear/GaussianFactorGraph.cpp:71-78, JacobianFactor::error :509-514
static double linear_error(const std::vector<GFactor>& g, const VectorValues& x) {
  double total = 0;
  for (auto& f : g) {
    const int m = f.Ab.r, n = f.Ab.c;
    double s = 0;
    for (int i = 0; i < m; i++) {
      double e = -f.Ab(i, n - 1);
      int col = 0;
      for (size_t k = 0; k < f.keys.size(); k++) {
        const auto& xv = x.at(f.keys[k]);
        for (int d = 0; d < f.dims[k]; d++, col++) e += f.Ab(i, col) * xv[d];
      }
      s += e * e;
    }
    total += 0.5 * s;
  }
  return total;
}

// hessianDiagonal gtsam/linear/GaussianFactorGraph.cpp:279-287
static VectorValues hessian_diagonal(const Problem& p) {
  VectorValues d;
  for (auto& kv : p.values) d[kv.first] = std::vector<double>(kVarDim[kv.second.type], 0.0);
  for (auto& f : p.linear) {
    int col = 0;
    for (size_t k = 0; k < f.keys.size(); k++) {
      auto& dv = d[f.keys[k]];
      for (int c = 0; c < f.dims[k]; c++, col++) {
        double s = 0;
        for (int i = 0; i < f.Ab.r; i++) s += f.Ab(i, col) * f.Ab(i, col);
        dv[c] += s;
      }
    }
  }
  return d;
}

static double now_s();

// buildDampedSystem + solve.  LevenbergMarquardtState.h:125-156, NonlinearOptimizer.cpp:132-146,
// GaussianFactorGraph::optimize -> eliminateMultifrontal (EliminateableFactorGraph-inst.h:123-146)
static void solve_damped(Problem& p, double lambda, const VectorValues* sqrtHessianDiagonal) {
  // the damped graph = the linear factors followed by one prior per variable (no copies of the linear factors)
  std::vector<GFactor> priors;
  priors.reserve(p.values.size());
  const double sigma = 1.0 / std::sqrt(lambda);
  for (auto& kv : p.values) {
    if (!(lambda > 0.0)) break;  // Gauss-Newton: the plain linearized graph (GaussNewtonOptimizer.cpp:44-66), no damping priors
    const int dim = kVarDim[kv.second.type];
    GFactor g;
    g.keys = {kv.first};
    g.dims = {dim};
    g.Ab = Mat(dim, dim + 1);
    for (int i = 0; i < dim; i++) {
      double a = sqrtHessianDiagonal ? sqrtHessianDiagonal->at(kv.first)[i] : 1.0;
      g.Ab(i, i) = a / sigma;  // JacobianFactor with Isotropic sigma: whitened at updateHessian (JacobianFactor.cpp:769-776)
    }
    priors.push_back(std::move(g));
  }
  std::vector<const GFactor*> damped;
  damped.reserve(p.linear.size() + priors.size());
  for (auto& g : p.linear) damped.push_back(&g);
  for (auto& g : priors) damped.push_back(&g);
  std::vector<std::vector<Key>> fkeys;
  fkeys.reserve(damped.size());
  for (auto* g : damped) fkeys.push_back(g->keys);
  VariableIndex vi;
  for (size_t i = 0; i < fkeys.size(); i++)
    for (Key k : fkeys[i]) vi[k].push_back(i);
  std::map<Key, int> keyDim;
  for (auto& kv : p.values) keyDim[kv.first] = kVarDim[kv.second.type];
  double t0 = now_s();
  auto roots = build_etree(fkeys, vi, p.ordering);
  const double ta = now_s();
  double tj = 0, te = 0;
  p.bt = BayesTree();
  for (auto& r : roots) {
    std::shared_ptr<JNode> jr;
    double tq = now_s();
    jt_visit(r, fkeys, jr);
    tj += now_s() - tq;
    tq = now_s();
    GFactor rem;
    int idx = eliminate_tree(jr, damped, keyDim, p.bt, rem);
    te += now_s() - tq;
    p.bt.roots.push_back(idx);
  }
  if (getenv("ORC_TRACE")) std::fprintf(stderr, "etree %.3f jt %.3f elim %.3f\n", ta - t0, tj, te);
  double t1 = now_s();
  p.delta.clear();
  backsub(p.bt, p.delta);
  double t2 = now_s();
  p.t_eliminate = t1 - t0;
  p.t_backsub = t2 - t1;
}

static Values retract_all(const Values& v, const VectorValues& d) {
  Values r;
  for (auto& kv : v) r[kv.first] = retract(kv.second, d.at(kv.first).data());
  return r;
}

// LevenbergMarquardtOptimizer::tryLambda gtsam/nonlinear/LevenbergMarquardtOptimizer.cpp:121-270
static bool try_lambda(Problem& p, const LMParams& prm, const VectorValues* sqrtHD) {
  double modelFidelity = 0.0;
  bool step_is_successful = false, stopSearchingLambda = false;
  double newError = std::numeric_limits<double>::infinity(), costChange = 0.0;
  Values newValues;
  bool solved;
  try {
    solve_damped(p, p.lambda, sqrtHD);
    solved = true;
  } catch (const Indeterminate&) {
    solved = false;
  }
  const double lambdaTried = p.lambda;
  if (solved) {
    VectorValues zero;
    for (auto& kv : p.delta) zero[kv.first] = std::vector<double>(kv.second.size(), 0.0);
    double oldLin = linear_error(p.linear, zero);
    double newLin = linear_error(p.linear, p.delta);
    double linearizedCostChange = oldLin - newLin;
    if (linearizedCostChange >= 0) {
      newValues = retract_all(p.values, p.delta);
      newError = graph_error(p, newValues);
      costChange = p.error - newError;
      if (linearizedCostChange > std::numeric_limits<double>::epsilon() * oldLin) {
        modelFidelity = costChange / linearizedCostChange;
        step_is_successful = modelFidelity > prm.minModelFidelity;
      }
      double minAbsoluteTolerance = prm.relativeErrorTol * p.error;
      if (std::abs(costChange) < minAbsoluteTolerance) stopSearchingLambda = true;
    }
  }
  p.trace.push_back(lambdaTried);
  p.trace.push_back(newError);
  p.trace.push_back(modelFidelity);
  p.trace.push_back(step_is_successful ? 1.0 : 0.0);
  p.trace.push_back(solved ? 1.0 : 0.0);
  if (step_is_successful) {
    // decreaseLambda LevenbergMarquardtState.h:80-93
    double newLambda = p.lambda, newFactor = p.currentFactor;
    if (prm.useFixedLambdaFactor) {
      newLambda /= p.currentFactor;
    } else {
      newLambda *= std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * modelFidelity - 1.0, 3));
      newFactor = 2.0 * p.currentFactor;
    }
    newLambda = std::max(prm.lambdaLowerBound, newLambda);
    p.values = std::move(newValues);
    p.error = newError;
    p.lambda = newLambda;
    p.currentFactor = newFactor;
    p.iterations += 1;
    p.totalInner += 1;
    return true;
  } else if (!stopSearchingLambda) {
    // increaseLambda :70-76
    p.lambda *= p.currentFactor;
    p.totalInner += 1;
    if (!prm.useFixedLambdaFactor) p.currentFactor *= 2.0;
    if (p.lambda >= prm.lambdaUpperBound) return true;
    return false;
  } else {
    return true;
  }
}

// LevenbergMarquardtOptimizer::iterate :273-308
static void lm_iterate(Problem& p, const LMParams& prm) {
  double t0 = now_s();
  linearize(p);
  p.t_linearize = now_s() - t0;
  VectorValues sqrtHD;
  if (prm.diagonalDamping) {
    sqrtHD = hessian_diagonal(p);
    for (auto& kv : sqrtHD)
      for (auto& x : kv.second) x = std::sqrt(std::min(std::max(x, prm.minDiagonal), prm.maxDiagonal));
  }
  while (!try_lambda(p, prm, prm.diagonalDamping ? &sqrtHD : nullptr)) {
  }
}

// GaussNewtonOptimizer::iterate gtsam/nonlinear/GaussNewtonOptimizer.cpp:44-66: linearize, solve, retract, new error;
// an IndeterminantLinearSystemException propagates to the caller (returned as false here)
static bool gn_iterate(Problem& p) {
  linearize(p);
  try {
    solve_damped(p, 0.0, nullptr);
  } catch (const Indeterminate&) {
    return false;
  }
  p.values = retract_all(p.values, p.delta);
  p.error = graph_error(p, p.values);
  p.iterations += 1;
  return true;
}

// ---- Dogleg (gtsam/nonlinear/DoglegOptimizer.cpp:84-126, DoglegOptimizerImpl.h:139-254, DoglegOptimizerImpl.cpp:26-91)
// The Bayes tree as a GaussianFactorGraph of unit-noise Jacobian factors [R S | d] (GaussianBayesTree.cpp:73-92).
// e_c = [R S] x - alpha d for every clique; returns sum ||e_c||^2
static double bt_residual_sq(const BayesTree& bt, const VectorValues& x, double alpha) {
  double tot = 0;
  for (const Clique& c : bt.cliques) {
    const int nf = c.RSd.r, n = c.RSd.c;
    std::vector<double> e(nf);
    for (int i = 0; i < nf; i++) e[i] = -alpha * c.RSd(i, n - 1);
    int col = 0;
    for (size_t k = 0; k < c.keys.size(); k++) {
      const auto& xk = x.at(c.keys[k]);
      for (int d = 0; d < c.dims[k]; d++, col++)
        for (int i = 0; i < nf; i++) e[i] += c.RSd(i, col) * xk[d];
    }
    for (int i = 0; i < nf; i++) tot += e[i] * e[i];
  }
  return tot;
}
// GaussianFactorGraph::gradientAtZero (GaussianFactorGraph.cpp:369-378; JacobianFactor.cpp:716-724): g = - sum [R S]^T d
static VectorValues bt_gradient_at_zero(const BayesTree& bt, const VectorValues& like) {
  VectorValues g;
  for (auto& kv : like) g[kv.first] = std::vector<double>(kv.second.size(), 0.0);
  for (const Clique& c : bt.cliques) {
    const int nf = c.RSd.r, n = c.RSd.c;
    int col = 0;
    for (size_t k = 0; k < c.keys.size(); k++) {
      auto& gk = g.at(c.keys[k]);
      for (int d = 0; d < c.dims[k]; d++, col++)
        for (int i = 0; i < nf; i++) gk[d] -= c.RSd(i, col) * c.RSd(i, n - 1);
    }
  }
  return g;
}
static double vv_dot(const VectorValues& a, const VectorValues& b) {
  double s = 0;
  for (auto& kv : a) {
    const auto& y = b.at(kv.first);
    for (size_t i = 0; i < kv.second.size(); i++) s += kv.second[i] * y[i];
  }
  return s;
}
// DoglegOptimizerImpl::ComputeDoglegPoint / ComputeBlend (DoglegOptimizerImpl.cpp:26-91)
static VectorValues dogleg_point(double delta, const VectorValues& dx_u, const VectorValues& dx_n) {
  const double deltaSq = delta * delta, uu = vv_dot(dx_u, dx_u), nn = vv_dot(dx_n, dx_n);
  VectorValues out = dx_n;
  if (deltaSq < uu) {
    const double f = std::sqrt(deltaSq / uu);
    out = dx_u;
    for (auto& kv : out)
      for (auto& x : kv.second) x *= f;
  } else if (deltaSq < nn) {
    const double un = vv_dot(dx_u, dx_n);
    const double a = uu - 2. * un + nn, b = 2. * (un - uu), c = uu - delta * delta;
    const double sq = std::sqrt(b * b - 4 * a * c);
    const double tau1 = (-b + sq) / (2. * a), tau2 = (-b - sq) / (2. * a);
    const double eps = std::numeric_limits<double>::epsilon();
    const double tau = (-eps <= tau1 && tau1 <= 1.0 + eps) ? tau1 : tau2;
    for (auto& kv : out) {
      const auto& u = dx_u.at(kv.first);
      for (size_t i = 0; i < kv.second.size(); i++) kv.second[i] = (1. - tau) * u[i] + tau * kv.second[i];
    }
  }
  return out;
}
// DoglegOptimizer::iterate with multifrontal elimination and ONE_STEP_PER_ITERATION; p.lambda carries the trust radius Delta
static bool dl_iterate(Problem& p) {
  linearize(p);
  try {
    solve_damped(p, 0.0, nullptr);  // Bayes tree in p.bt, Newton step in p.delta
  } catch (const Indeterminate&) {
    return false;
  }
  const VectorValues dx_n = p.delta;
  VectorValues dx_u = bt_gradient_at_zero(p.bt, dx_n);
  const double gg = vv_dot(dx_u, dx_u);
  const double step = -gg / bt_residual_sq(p.bt, dx_u, 0.0);  // GaussianFactorGraph::optimizeGradientSearch :381-406
  for (auto& kv : dx_u)
    for (auto& x : kv.second) x *= step;
  VectorValues zero = dx_n;
  for (auto& kv : zero)
    for (auto& x : kv.second) x = 0.0;
  const double M_error = 0.5 * bt_residual_sq(p.bt, zero, 1.0);
  double delta = p.lambda;
  const double f_error = p.error;
  VectorValues dx_d;
  double new_f = f_error;
  bool stay = true;
  while (stay) {
    dx_d = dogleg_point(delta, dx_u, dx_n);
    const Values x_d = retract_all(p.values, dx_d);
    new_f = graph_error(p, x_d);
    const double new_M = 0.5 * bt_residual_sq(p.bt, dx_d, 1.0);
    const double rho = (std::abs(f_error - new_f) < 1e-15 || std::abs(M_error - new_M) < 1e-15) ? 0.5 : (f_error - new_f) / (M_error - new_M);
    if (rho >= 0.75) {
      const double nrm = std::sqrt(vv_dot(dx_d, dx_d));
      delta = std::max(delta, 3.0 * nrm);
      stay = false;
    } else if (rho >= 0.25) {
      stay = false;
    } else if (rho >= 0.0) {
      if (delta > 1e-5) delta *= 0.5;
      stay = false;  // ONE_STEP_PER_ITERATION
    } else {  // f increased (or NaN): halve until it does not
      if (delta > 1e-5) {
        delta *= 0.5;
        stay = true;
      } else {
        for (auto& kv : dx_d)
          for (auto& x : kv.second) x = 0.0;
        new_f = f_error;
        stay = false;
      }
    }
  }
  p.values = retract_all(p.values, dx_d);
  p.error = new_f;
  p.lambda = delta;
  p.iterations += 1;
  p.delta = dx_d;
  return true;
}

// checkConvergence gtsam/nonlinear/NonlinearOptimizer.cpp:182-231
static bool check_convergence(double relTol, double absTol, double errTol, double currentError, double newError) {
  if (newError <= errTol) return true;
  double absoluteDecrease = currentError - newError;
  double relativeDecrease = absoluteDecrease / currentError;
  return (relTol && (relativeDecrease <= relTol)) || (absoluteDecrease <= absTol);
}

// NonlinearOptimizer::defaultOptimize gtsam/nonlinear/NonlinearOptimizer.cpp:62-117
static void lm_optimize(Problem& p, const LMParams& prm) {
  double currentError = p.error;
  if (currentError <= prm.errorTol) return;
  if (p.iterations >= prm.maxIterations) return;
  double newError = currentError;
  do {
    currentError = newError;
    lm_iterate(p, prm);
    newError = p.error;
  } while (p.iterations < prm.maxIterations &&
           !check_convergence(prm.relativeErrorTol, prm.absoluteErrorTol, prm.errorTol, currentError, newError) &&
           std::isfinite(currentError));
}

}  // namespace orc

#include <chrono>
namespace orc {
static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace orc

// =================================================================== C interface (ctypes)
#include "isam2_oracle.hpp"

using namespace orc;
extern "C" {

void* orc_create() { return new Problem(); }
void orc_destroy(void* h) { delete (Problem*)h; }

int orc_add_variable(void* h, uint64_t key, int type, const double* value) {
  auto* p = (Problem*)h;
  if (type < 0 || type > 5) return 2;
  Value v;
  v.type = type;
  std::memset(v.v, 0, sizeof(v.v));
  std::memcpy(v.v, value, kVarStore[type] * sizeof(double));
  p->values[key] = v;
  return 0;
}

int orc_add_factor(void* h, int type, const uint64_t* keys, const double* meas, int noise_kind, const double* noise) {
  auto* p = (Problem*)h;
  if (type < 0 || type >= kNumFactorTypes) return 2;
  Factor f;
  f.type = type;
  f.keys[0] = keys[0];
  f.keys[1] = kFactorArity[type] > 1 ? keys[1] : 0;
  f.keys[2] = kFactorArity[type] > 2 ? keys[2] : 0;
  std::memset(f.meas, 0, sizeof(f.meas));
  std::memcpy(f.meas, meas, kFactorMeas[type] * sizeof(double));
  f.noise_kind = noise_kind;
  const int m = kFactorRows[type];
  if (noise_kind == N_ISO) f.noise.assign(noise, noise + 1);
  if (noise_kind == N_DIAG) f.noise.assign(noise, noise + m);
  if (noise_kind == N_GAUSS) f.noise.assign(noise, noise + m * m);
  p->factors.push_back(f);
  return 0;
}

// tap: weight and loss of an m-estimator (pinned against gtsam/linear/tests/testNoiseModel.cpp:460-617)
int orc_robust(int kind, double k, double distance, double* out2) {
  out2[0] = robust_weight(kind, k, distance);
  out2[1] = robust_loss(kind, k, distance);
  return 0;
}

int orc_set_factor_robust(void* h, int index, int kind, double k) {
  auto* p = (Problem*)h;
  if (index < 0 || index >= (int)p->factors.size() || kind < 0 || kind > 8) return 2;
  p->factors[index].robust = kind;
  p->factors[index].rk = k;
  return 0;
}

int orc_set_ordering(void* h, int n, const uint64_t* keys) {
  auto* p = (Problem*)h;
  p->ordering.assign(keys, keys + n);
  return 0;
}

// threads of the all-cores baseline leg (1 = the plain single-thread restatement); returns the value in effect
// threads inside the dense factorisation of a large front (1 = like the reference, whose Eigen LLT is single-threaded)
int orc_set_dense_threads(int n) {
  chol_threads = std::max(1, n);
  return chol_threads;
}
int orc_set_threads(int n) {
  g_threads = std::max(1, n);
  if (g_threads > 1) {
    // every clique allocates its (n x n) separator factor; with glibc's default arena growth that is one mprotect / page-fault
    // storm per thread, serialised in the kernel (measured: no speed-up at 8 threads).  Keep freed memory and grow in large steps.
    mallopt(M_TOP_PAD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 0x7fffffff);
  }
  return g_threads;
}

int orc_num_variables(void* h) { return (int)((Problem*)h)->values.size(); }
int orc_num_factors(void* h) { return (int)((Problem*)h)->factors.size(); }
int orc_total_dim(void* h) {
  int n = 0;
  for (auto& kv : ((Problem*)h)->values) n += kVarDim[kv.second.type];
  return n;
}

// values in key-sorted order, packed kVarStore doubles each
int orc_get_values(void* h, double* out) {
  for (auto& kv : ((Problem*)h)->values) {
    std::memcpy(out, kv.second.v, kVarStore[kv.second.type] * sizeof(double));
    out += kVarStore[kv.second.type];
  }
  return 0;
}

double orc_error(void* h) {
  auto* p = (Problem*)h;
  return graph_error(*p, p->values);
}

int orc_linearize(void* h) {
  linearize(*(Problem*)h);
  return 0;
}

// Jacobian of factor i: col-major m x (sum dims + 1)  (the VerticalBlockMatrix of the JacobianFactor)
int orc_get_jacobian(void* h, int i, double* out, int* rows, int* cols) {
  auto* p = (Problem*)h;
  const GFactor& g = p->linear.at(i);
  *rows = g.Ab.r;
  *cols = g.Ab.c;
  if (out) std::memcpy(out, g.Ab.a.data(), g.Ab.a.size() * sizeof(double));
  return 0;
}

// one damped solve; delta_out packed in key-sorted order.  returns 0 ok, 1 indeterminate
int orc_solve(void* h, double lambda, int diagonal_damping, double min_diag, double max_diag, double* delta_out,
              double* lin_err0, double* lin_err1) {
  auto* p = (Problem*)h;
  VectorValues sqrtHD;
  if (diagonal_damping) {
    sqrtHD = hessian_diagonal(*p);
    for (auto& kv : sqrtHD)
      for (auto& x : kv.second) x = std::sqrt(std::min(std::max(x, min_diag), max_diag));
  }
  try {
    solve_damped(*p, lambda, diagonal_damping ? &sqrtHD : nullptr);
  } catch (const Indeterminate&) {
    return 1;
  }
  if (delta_out)
    for (auto& kv : p->values) {
      auto& d = p->delta.at(kv.first);
      std::memcpy(delta_out, d.data(), d.size() * sizeof(double));
      delta_out += d.size();
    }
  VectorValues zero;
  for (auto& kv : p->delta) zero[kv.first] = std::vector<double>(kv.second.size(), 0.0);
  if (lin_err0) *lin_err0 = linear_error(p->linear, zero);
  if (lin_err1) *lin_err1 = linear_error(p->linear, p->delta);
  return 0;
}

// Marginals::marginalCovariance (gtsam/nonlinear/Marginals.cpp:124-127) of one variable at the current values:
// the constructor linearizes the graph at the solution (:28-33); marginalInformation (:109-121) is the information matrix of
// BayesTree::marginalFactor(variable), i.e. the Schur complement of the whole information matrix A^T A onto the variable, and
// marginalCovariance its inverse -- the variable's diagonal block of (A^T A)^-1.  The reference reaches it through Bayes-tree
// shortcuts; this restatement assembles the information matrix densely and inverts it by Cholesky, which is the same
// quantity for every ordering (sizes the oracle is used at: total dimension up to a few thousand).
// out: dim x dim row-major.  Returns 0 ok, 1 singular information matrix (the reference throws IndeterminantLinearSystemException).
int orc_marginal_covariance(void* h, uint64_t key, double* out) {
  auto* p = (Problem*)h;
  linearize(*p);
  std::map<Key, int> off;
  int n = 0;
  for (auto& kv : p->values) {
    off[kv.first] = n;
    n += kVarDim[kv.second.type];
  }
  if (!off.count(key)) return 2;
  std::vector<double> L((size_t)n * n, 0.0);  // information matrix, then its Cholesky factor (lower, row-major)
  for (auto& f : p->linear) {
    if (f.hessian) return 3;
    std::vector<int> col;  // global column of every Jacobian column
    for (size_t k = 0; k < f.keys.size(); k++)
      for (int c = 0; c < f.dims[k]; c++) col.push_back(off.at(f.keys[k]) + c);
    for (size_t a = 0; a < col.size(); a++)
      for (size_t b = 0; b < col.size(); b++) {
        double sum = 0;
        for (int i = 0; i < f.Ab.r; i++) sum += f.Ab(i, (int)a) * f.Ab(i, (int)b);
        L[(size_t)col[a] * n + col[b]] += sum;
      }
  }
  for (int j = 0; j < n; j++) {
    double d = L[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
    if (!(d > 0.0)) return 1;
    d = std::sqrt(d);
    L[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double v = L[(size_t)i * n + j];
      for (int k = 0; k < j; k++) v -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
      L[(size_t)i * n + j] = v / d;
    }
  }
  const int x0 = off.at(key), dim = kVarDim[p->values.at(key).type];
  std::vector<double> x(n);
  for (int c = 0; c < dim; c++) {  // column c of the block: L L^T x = e_{x0 + c}
    std::fill(x.begin(), x.end(), 0.0);
    x[x0 + c] = 1.0;
    for (int i = 0; i < n; i++) {
      double v = x[i];
      for (int k = 0; k < i; k++) v -= L[(size_t)i * n + k] * x[k];
      x[i] = v / L[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
      double v = x[i];
      for (int k = i + 1; k < n; k++) v -= L[(size_t)k * n + i] * x[k];
      x[i] = v / L[(size_t)i * n + i];
    }
    for (int r = 0; r < dim; r++) out[(size_t)r * dim + c] = x[x0 + r];
  }
  return 0;
}

int orc_hessian_diagonal(void* h, double* out) {
  auto* p = (Problem*)h;
  auto d = hessian_diagonal(*p);
  for (auto& kv : p->values) {
    auto& v = d.at(kv.first);
    std::memcpy(out, v.data(), v.size() * sizeof(double));
    out += v.size();
  }
  return 0;
}

int orc_retract(void* h, const double* delta) {
  auto* p = (Problem*)h;
  VectorValues d;
  for (auto& kv : p->values) {
    int dim = kVarDim[kv.second.type];
    d[kv.first] = std::vector<double>(delta, delta + dim);
    delta += dim;
  }
  p->values = retract_all(p->values, d);
  return 0;
}

// Bayes tree taps
int orc_num_cliques(void* h) { return (int)((Problem*)h)->bt.cliques.size(); }
// sizes: nkeys, nfrontal_keys, nf (rows), n (cols), parent
int orc_clique_info(void* h, int i, int* info5) {
  const Clique& c = ((Problem*)h)->bt.cliques.at(i);
  info5[0] = (int)c.keys.size();
  info5[1] = c.nFrontal;
  info5[2] = c.RSd.r;
  info5[3] = c.RSd.c;
  info5[4] = c.parent;
  return 0;
}
int orc_clique_get(void* h, int i, uint64_t* keys, double* RSd_colmajor) {
  const Clique& c = ((Problem*)h)->bt.cliques.at(i);
  if (keys) std::memcpy(keys, c.keys.data(), c.keys.size() * sizeof(uint64_t));
  if (RSd_colmajor) std::memcpy(RSd_colmajor, c.RSd.a.data(), c.RSd.a.size() * sizeof(double));
  return 0;
}

struct orc_lm_params {
  int maxIterations;
  double relativeErrorTol, absoluteErrorTol, errorTol;
  double lambdaInitial, lambdaFactor, lambdaUpperBound, lambdaLowerBound;
  double minModelFidelity;
  int diagonalDamping, useFixedLambdaFactor;
  double minDiagonal, maxDiagonal;
};
static LMParams to_params(const orc_lm_params* q) {
  LMParams p;
  p.maxIterations = q->maxIterations;
  p.relativeErrorTol = q->relativeErrorTol;
  p.absoluteErrorTol = q->absoluteErrorTol;
  p.errorTol = q->errorTol;
  p.lambdaInitial = q->lambdaInitial;
  p.lambdaFactor = q->lambdaFactor;
  p.lambdaUpperBound = q->lambdaUpperBound;
  p.lambdaLowerBound = q->lambdaLowerBound;
  p.minModelFidelity = q->minModelFidelity;
  p.diagonalDamping = q->diagonalDamping;
  p.useFixedLambdaFactor = q->useFixedLambdaFactor;
  p.minDiagonal = q->minDiagonal;
  p.maxDiagonal = q->maxDiagonal;
  return p;
}

// LevenbergMarquardtOptimizer ctor: state = (values, graph.error(values), lambdaInitial, lambdaFactor)
int orc_lm_init(void* h, const orc_lm_params* q) {
  auto* p = (Problem*)h;
  p->error = graph_error(*p, p->values);
  p->lambda = q->lambdaInitial;
  p->currentFactor = q->lambdaFactor;
  p->iterations = 0;
  p->totalInner = 0;
  p->trace.clear();
  return 0;
}
int orc_lm_iterate(void* h, const orc_lm_params* q) {
  lm_iterate(*(Problem*)h, to_params(q));
  return 0;
}
int orc_lm_optimize(void* h, const orc_lm_params* q) {
  lm_optimize(*(Problem*)h, to_params(q));
  return 0;
}
int orc_gn_iterate(void* h) { return gn_iterate(*(Problem*)h) ? 0 : 1; }
// NonlinearOptimizer::defaultOptimize with GaussNewtonOptimizer::iterate
int orc_gn_optimize(void* h, const orc_lm_params* q) {
  auto& p = *(Problem*)h;
  const LMParams prm = to_params(q);
  double currentError = p.error;
  if (currentError <= prm.errorTol || p.iterations >= prm.maxIterations) return 0;
  double newError = currentError;
  do {
    currentError = newError;
    if (!gn_iterate(p)) return 1;
    newError = p.error;
  } while (p.iterations < prm.maxIterations &&
           !check_convergence(prm.relativeErrorTol, prm.absoluteErrorTol, prm.errorTol, currentError, newError) && std::isfinite(currentError));
  return 0;
}
// DoglegOptimizer: state.lambda = trust radius Delta (DoglegParams::deltaInitial, default 1.0)
int orc_dl_init(void* h, double deltaInitial) {
  auto* p = (Problem*)h;
  p->error = graph_error(*p, p->values);
  p->lambda = deltaInitial;
  p->iterations = 0;
  p->totalInner = 0;
  return 0;
}
int orc_dl_iterate(void* h) { return dl_iterate(*(Problem*)h) ? 0 : 1; }
int orc_dl_optimize(void* h, const orc_lm_params* q) {
  auto& p = *(Problem*)h;
  const LMParams prm = to_params(q);
  double currentError = p.error;
  if (currentError <= prm.errorTol || p.iterations >= prm.maxIterations) return 0;
  double newError = currentError;
  do {
    currentError = newError;
    if (!dl_iterate(p)) return 1;
    newError = p.error;
  } while (p.iterations < prm.maxIterations &&
           !check_convergence(prm.relativeErrorTol, prm.absoluteErrorTol, prm.errorTol, currentError, newError) && std::isfinite(currentError));
  return 0;
}
// taps: steepest-descent point and Newton point of the current values (ordering order), and the dogleg blend
int orc_dl_points(void* h, double* xu_out, double* xn_out) {
  auto& p = *(Problem*)h;
  linearize(p);
  try {
    solve_damped(p, 0.0, nullptr);
  } catch (const Indeterminate&) {
    return 1;
  }
  VectorValues dx_u = bt_gradient_at_zero(p.bt, p.delta);
  const double step = -vv_dot(dx_u, dx_u) / bt_residual_sq(p.bt, dx_u, 0.0);
  int o = 0;
  for (Key k : p.ordering) {
    const auto& u = dx_u.at(k);
    const auto& nn = p.delta.at(k);
    for (size_t i = 0; i < u.size(); i++, o++) {
      xu_out[o] = step * u[i];
      xn_out[o] = nn[i];
    }
  }
  return 0;
}
int orc_dogleg_point(int n, const double* xu, const double* xn, double delta, double* out) {
  VectorValues u, v;
  u[0] = std::vector<double>(xu, xu + n);
  v[0] = std::vector<double>(xn, xn + n);
  const VectorValues d = dogleg_point(delta, u, v);
  std::memcpy(out, d.at(0).data(), n * sizeof(double));
  return 0;
}
// last step taken (LM / GN: the accepted or last tried delta; Dogleg: dx_d), in ordering order
int orc_get_delta(void* h, double* out) {
  auto* p = (Problem*)h;
  int o = 0;
  for (Key k : p->ordering) {
    const auto& v = p->delta.at(k);
    for (double x : v) out[o++] = x;
  }
  return 0;
}
// state: error, lambda, iterations, totalInner, currentFactor
int orc_lm_state(void* h, double* out5) {
  auto* p = (Problem*)h;
  out5[0] = p->error;
  out5[1] = p->lambda;
  out5[2] = p->iterations;
  out5[3] = p->totalInner;
  out5[4] = p->currentFactor;
  return 0;
}
int orc_lm_trace_len(void* h) { return (int)((Problem*)h)->trace.size(); }
int orc_lm_trace(void* h, double* out) {
  auto* p = (Problem*)h;
  std::memcpy(out, p->trace.data(), p->trace.size() * sizeof(double));
  return 0;
}
int orc_timings(void* h, double* out3) {
  auto* p = (Problem*)h;
  out3[0] = p->t_linearize;
  out3[1] = p->t_eliminate;
  out3[2] = p->t_backsub;
  return 0;
}

// ---- low-level taps used to pin the oracle against the reference's golden vectors ----
// choleskyPartial on a col-major n x n matrix (upper triangle), in place.  returns 1 on success.
int orc_cholesky_partial(double* ABC, int n, int nFrontal) { return cholesky_partial(ABC, n, n, nFrontal) ? 1 : 0; }

// Linear (Gaussian) factor graph built directly: Jacobian factors [A|b] with diagonal sigmas, or Hessian
// factors, then eliminated with a given ordering.  Used for testHessianFactor / testGaussianBayesTree vectors.
struct LinearProblem {
  std::vector<GFactor> graph;
  std::map<Key, int> keyDim;
  BayesTree bt;
  VectorValues x;
  std::vector<GFactor> remaining;
};
void* orc_linear_create() { return new LinearProblem(); }
void orc_linear_destroy(void* h) { delete (LinearProblem*)h; }
// A: col-major m x (sum dims), b: m, sigmas: m or null (unit).  Whitening as JacobianFactor::whiten (JacobianFactor.cpp:769-776)
int orc_linear_add_jacobian(void* h, int nkeys, const uint64_t* keys, const int* dims, int m, const double* A, const double* b,
                            const double* sigmas) {
  auto* p = (LinearProblem*)h;
  GFactor g;
  int tot = 0;
  for (int i = 0; i < nkeys; i++) {
    g.keys.push_back(keys[i]);
    g.dims.push_back(dims[i]);
    p->keyDim[keys[i]] = dims[i];
    tot += dims[i];
  }
  g.Ab = Mat(m, tot + 1);
  for (int j = 0; j < tot; j++)
    for (int i = 0; i < m; i++) g.Ab(i, j) = A[(size_t)j * m + i] / (sigmas ? sigmas[i] : 1.0);
  for (int i = 0; i < m; i++) g.Ab(i, tot) = b[i] / (sigmas ? sigmas[i] : 1.0);
  p->graph.push_back(g);
  return 0;
}
// info: col-major (sum dims + 1)^2 augmented information matrix [G g; g' f] (upper triangle read)
int orc_linear_add_hessian(void* h, int nkeys, const uint64_t* keys, const int* dims, const double* info) {
  auto* p = (LinearProblem*)h;
  GFactor g;
  g.hessian = true;
  int tot = 1;
  for (int i = 0; i < nkeys; i++) {
    g.keys.push_back(keys[i]);
    g.dims.push_back(dims[i]);
    p->keyDim[keys[i]] = dims[i];
    tot += dims[i];
  }
  g.info = Mat(tot, tot);
  std::memcpy(g.info.a.data(), info, sizeof(double) * tot * tot);
  p->graph.push_back(g);
  return 0;
}
// EliminateCholesky of the whole graph on the given frontal keys (one dense step).  Outputs: keys in Scatter
// order, [R S d] (col-major nf x n) and the remaining separator information (col-major ns x ns, upper).
int orc_linear_eliminate_dense(void* h, int nfrontal, const uint64_t* frontal, int* nkeys_out, uint64_t* keys_out, int* nf_out,
                               int* n_out, double* RSd, double* sep_info) {
  auto* p = (LinearProblem*)h;
  std::vector<const GFactor*> gathered;
  for (auto& g : p->graph) gathered.push_back(&g);
  std::vector<Key> fr(frontal, frontal + nfrontal);
  Clique cq;
  GFactor sep;
  try {
    eliminate_clique(gathered, fr, p->keyDim, cq, sep);
  } catch (const Indeterminate&) {
    return 1;
  }
  *nkeys_out = (int)cq.keys.size();
  std::memcpy(keys_out, cq.keys.data(), cq.keys.size() * sizeof(uint64_t));
  *nf_out = cq.RSd.r;
  *n_out = cq.RSd.c;
  std::memcpy(RSd, cq.RSd.a.data(), cq.RSd.a.size() * sizeof(double));
  if (sep_info) std::memcpy(sep_info, sep.info.a.data(), sep.info.a.size() * sizeof(double));
  return 0;
}
// multifrontal solve of the linear graph with the given ordering; x packed in key-sorted order
int orc_linear_optimize(void* h, int n, const uint64_t* ordering, double* x_out) {
  auto* p = (LinearProblem*)h;
  std::vector<std::vector<Key>> fkeys;
  for (auto& g : p->graph) fkeys.push_back(g.keys);
  VariableIndex vi;
  for (size_t i = 0; i < fkeys.size(); i++)
    for (Key k : fkeys[i]) vi[k].push_back(i);
  std::vector<Key> order(ordering, ordering + n);
  try {
    auto roots = build_etree(fkeys, vi, order);
    p->bt = BayesTree();
    for (auto& r : roots) {
      std::shared_ptr<JNode> jr;
      jt_visit(r, fkeys, jr);
      GFactor rem;
      std::vector<const GFactor*> gp;
      for (auto& g : p->graph) gp.push_back(&g);
      p->bt.roots.push_back(eliminate_tree(jr, gp, p->keyDim, p->bt, rem));
    }
    p->x.clear();
    backsub(p->bt, p->x);
  } catch (const Indeterminate&) {
    return 1;
  }
  for (auto& kv : p->keyDim) {
    auto& v = p->x.at(kv.first);
    std::memcpy(x_out, v.data(), v.size() * sizeof(double));
    x_out += v.size();
  }
  return 0;
}
int orc_linear_num_cliques(void* h) { return (int)((LinearProblem*)h)->bt.cliques.size(); }
int orc_linear_clique_info(void* h, int i, int* info5) {
  const Clique& c = ((LinearProblem*)h)->bt.cliques.at(i);
  info5[0] = (int)c.keys.size();
  info5[1] = c.nFrontal;
  info5[2] = c.RSd.r;
  info5[3] = c.RSd.c;
  info5[4] = c.parent;
  return 0;
}
int orc_linear_clique_get(void* h, int i, uint64_t* keys, double* RSd_colmajor) {
  const Clique& c = ((LinearProblem*)h)->bt.cliques.at(i);
  if (keys) std::memcpy(keys, c.keys.data(), c.keys.size() * sizeof(uint64_t));
  if (RSd_colmajor) std::memcpy(RSd_colmajor, c.RSd.a.data(), c.RSd.a.size() * sizeof(double));
  return 0;
}

// geometry taps for golden tests (testCal3Bundler, testPose3, testSO3)
void orc_cal3bundler_uncalibrate(const double K[5], double x, double y, double out[2], double Dcal[6], double Dp[4]) {
  Cal3Bundler k{K[0], K[1], K[2], K[3], K[4]};
  cal3bundler_uncalibrate(k, x, y, out, Dcal, Dp);
}
void orc_pose3_expmap(const double xi[6], double out12[12]) { store_pose3(pose3_expmap(xi), out12); }
void orc_pose3_logmap(const double in12[12], double xi[6]) { pose3_logmap(as_pose3(in12), xi); }
void orc_rot3_expmap(const double w[3], double R[9]) {
  M3 r = ExpmapFunctor(V3{w[0], w[1], w[2]}).expmap();
  std::memcpy(R, r.m, sizeof(r.m));
}
void orc_rot3_logmap(const double R[9], double w[3]) {
  M3 r;
  std::memcpy(r.m, R, sizeof(r.m));
  V3 o = so3_logmap(r);
  w[0] = o.x; w[1] = o.y; w[2] = o.z;
}
// unwhitened error + Jacobians of a single factor (row-major H1, H2)
int orc_factor_evaluate(void* h, int i, double* e, double* H1, double* H2) {
  auto* p = (Problem*)h;
  evaluate_error(p->factors.at(i), p->values, e, H1, H2);
  return 0;
}
// the same for a three-variable factor
int orc_factor_evaluate3(void* h, int i, double* e, double* H1, double* H2, double* H3) {
  auto* p = (Problem*)h;
  evaluate_error(p->factors.at(i), p->values, e, H1, H2, H3);
  return 0;
}
}

// =================================================================== ISAM2 (isam2_oracle.hpp)
namespace orc {
struct ISAM2Handle {
  ISAM2 S;
  std::vector<ICliquePtr> snap;      // depth-first snapshot of the Bayes tree for the taps below
  std::vector<int> snapParent;
};
}  // namespace orc

extern "C" {

void* orc_isam2_create(double relinearizeThreshold, int relinearizeSkip, int enableRelinearization, double wildfireThreshold,
                       orc::ccolamd_fn cb) {
  auto* h = new ISAM2Handle();
  h->S.relinearizeThreshold = relinearizeThreshold;
  h->S.relinearizeSkip = relinearizeSkip;
  h->S.enableRelinearization = enableRelinearization != 0;
  h->S.wildfireThreshold = wildfireThreshold;
  h->S.ccolamd = cb;
  return h;
}
void orc_isam2_destroy(void* h) { delete (ISAM2Handle*)h; }
// ISAM2Params::relinearizeThreshold = FastMap<char, Vector> (n entries: character, dimension, values back to back; n = 0: the double again)
// and ISAM2Params::enablePartialRelinearizationCheck
void orc_isam2_set_thresholds(void* h, int n, const unsigned char* chrs, const int* dims, const double* values) {
  auto& S = ((ISAM2Handle*)h)->S;
  S.relinearizeThresholds.clear();
  for (int i = 0; i < n; i++) {
    S.relinearizeThresholds[chrs[i]] = std::vector<double>(values, values + dims[i]);
    values += dims[i];
  }
}
// ISAM2Params::evaluateNonlinearError; ISAM2Result::errorBefore / errorAfter of the last update
void orc_isam2_set_evaluate_error(void* h, int enable) { ((ISAM2Handle*)h)->S.evaluateNonlinearError = enable != 0; }
void orc_isam2_errors(void* h, double* before, double* after) {
  *before = ((ISAM2Handle*)h)->S.errorBefore;
  *after = ((ISAM2Handle*)h)->S.errorAfter;
}
// getFactorsUnsafe().error(values): which = 0 calculateEstimate(), 2 the linearization point
double orc_isam2_error(void* h, int which) {
  auto& S = ((ISAM2Handle*)h)->S;
  return isam2_graph_error(S, which == 2 ? S.theta : isam2_calculate_estimate(S, false));
}
// ISAM2Params::optimizationParams = ISAM2DoglegParams(initialDelta, wildfireThreshold, adaptationMode) (before the first update)
void orc_isam2_set_dogleg(void* h, double initialDelta, double wildfireThreshold, int adaptationMode) {
  auto& S = ((ISAM2Handle*)h)->S;
  S.dogleg = true;
  S.doglegDelta = initialDelta;
  S.doglegWildfireThreshold = wildfireThreshold;
  S.doglegAdaptationMode = adaptationMode;
}
double orc_isam2_dogleg_delta(void* h) { return ((ISAM2Handle*)h)->S.doglegDelta; }
void orc_isam2_set_partial_check(void* h, int enable) { ((ISAM2Handle*)h)->S.enablePartialRelinearizationCheck = enable != 0; }

int orc_isam2_add_variable(void* h, uint64_t key, int type, const double* value) {
  auto& S = ((ISAM2Handle*)h)->S;
  if (type < 0 || type > 5) return 2;
  Value v;
  v.type = type;
  std::memset(v.v, 0, sizeof(v.v));
  std::memcpy(v.v, value, kVarStore[type] * sizeof(double));
  S.newTheta[key] = v;
  return 0;
}

int orc_isam2_add_factor(void* h, int type, const uint64_t* keys, const double* meas, int noise_kind, const double* noise) {
  auto& S = ((ISAM2Handle*)h)->S;
  if (type < 0 || type >= kNumFactorTypes) return 2;
  Factor f;
  f.type = type;
  f.keys[0] = keys[0];
  f.keys[1] = kFactorArity[type] > 1 ? keys[1] : 0;
  f.keys[2] = kFactorArity[type] > 2 ? keys[2] : 0;
  std::memset(f.meas, 0, sizeof(f.meas));
  std::memcpy(f.meas, meas, kFactorMeas[type] * sizeof(double));
  f.noise_kind = noise_kind;
  const int m = kFactorRows[type];
  if (noise_kind == N_ISO) f.noise.assign(noise, noise + 1);
  if (noise_kind == N_DIAG) f.noise.assign(noise, noise + m);
  if (noise_kind == N_GAUSS) f.noise.assign(noise, noise + m * m);
  S.newFactors.push_back(f);
  return 0;
}
// the same with noiseModel::Robust(mEstimator(k), model) around the factor's Gaussian model (gtsam/linear/NoiseModel.h:663-760)
int orc_isam2_add_factor_robust(void* h, int type, const uint64_t* keys, const double* meas, int noise_kind, const double* noise, int robust, double rk) {
  const int rc = orc_isam2_add_factor(h, type, keys, meas, noise_kind, noise);
  if (rc) return rc;
  auto& S = ((ISAM2Handle*)h)->S;
  S.newFactors.back().robust = robust;
  S.newFactors.back().rk = rk;
  return 0;
}

// result5: variablesRelinearized, variablesReeliminated, factorsRecalculated, cliques, batch.  1 = indeterminate system
static int isam2_update_guarded(void* h, const orc::ISAM2UpdateParams& up, int* result5) {
  auto& S = ((ISAM2Handle*)h)->S;
  try {
    const ISAM2Result r = isam2_update(S, up);
    if (result5) {
      result5[0] = r.variablesRelinearized;
      result5[1] = r.variablesReeliminated;
      result5[2] = r.factorsRecalculated;
      result5[3] = r.cliques;
      result5[4] = r.batch;
    }
  } catch (const Indeterminate&) {
    return 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "orc_isam2_update: %s\n", e.what());
    return 2;
  }
  return 0;
}
int orc_isam2_update(void* h, int force_relinearize, int* result5) {
  orc::ISAM2UpdateParams up;
  up.force_relinearize = force_relinearize != 0;
  return isam2_update_guarded(h, up, result5);
}
// ISAM2::update(newFactors, newTheta, ISAM2UpdateParams) (gtsam/nonlinear/ISAM2.h:146-186): has_constrained = the optional is engaged
int orc_isam2_update_with(void* h, int n_remove, const uint64_t* remove_idx, int has_constrained, int n_constrained, const uint64_t* ckeys,
                          const int* cgroups, int n_norelin, const uint64_t* norelin, int n_extra, const uint64_t* extra, int force_relinearize,
                          int forceFullSolve, int* result5) {
  orc::ISAM2UpdateParams up;
  up.removeFactorIndices.assign(remove_idx, remove_idx + n_remove);
  up.hasConstrainedKeys = has_constrained != 0;
  for (int i = 0; i < n_constrained; i++) up.constrainedKeys[ckeys[i]] = cgroups[i];
  up.noRelinKeys.assign(norelin, norelin + n_norelin);
  up.extraReelimKeys.assign(extra, extra + n_extra);
  up.force_relinearize = force_relinearize != 0;
  up.forceFullSolve = forceFullSolve != 0;
  return isam2_update_guarded(h, up, result5);
}
// ISAM2Result::unusedKeys of the last update (ascending); returns their number
int orc_isam2_unused_keys(void* h, uint64_t* keys_out) {
  auto& S = ((ISAM2Handle*)h)->S;
  if (keys_out) std::copy(S.lastUnusedKeys.begin(), S.lastUnusedKeys.end(), keys_out);
  return (int)S.lastUnusedKeys.size();
}
// 1 when slot i of the factor list still holds a factor
int orc_isam2_factor_exists(void* h, int i) {
  auto& S = ((ISAM2Handle*)h)->S;
  return i >= 0 && i < (int)S.nonlinearFactors.size() && !S.removedFactor[i];
}

// ISAM2Params::findUnusedFactorSlots (ISAM2Params.h:225)
void orc_isam2_set_find_unused_slots(void* h, int enable) { ((ISAM2Handle*)h)->S.findUnusedFactorSlots = enable != 0; }

// ISAM2::marginalizeLeaves(leafKeys, &marginalFactorsIndices, &deletedFactorsIndices) (ISAM2.h:198-222).  The index outputs may be null
// (their counts are returned in counts2: marginal, deleted).  1 = a key is not a leaf (the reference's debug-build exception; the object
// is unusable afterwards, as there), 2 = another error
int orc_isam2_marginalize_leaves(void* h, int n, const uint64_t* keys, uint64_t* marginal_idx, uint64_t* deleted_idx, int* counts2) {
  auto& S = ((ISAM2Handle*)h)->S;
  try {
    std::vector<size_t> mi, di;
    isam2_marginalize_leaves(S, std::vector<Key>(keys, keys + n), &mi, &di);
    if (marginal_idx) std::copy(mi.begin(), mi.end(), marginal_idx);
    if (deleted_idx) std::copy(di.begin(), di.end(), deleted_idx);
    if (counts2) {
      counts2[0] = (int)mi.size();
      counts2[1] = (int)di.size();
    }
  } catch (const std::runtime_error& e) {
    std::fprintf(stderr, "orc_isam2_marginalize_leaves: %s\n", e.what());
    return 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "orc_isam2_marginalize_leaves: %s\n", e.what());
    return 2;
  }
  return 0;
}
// slot i as a marginal (LinearContainerFactor) factor: returns its number of keys, -1 when the slot holds none; keys / dims / the
// augmented information matrix ((sum dims + 1)^2, column-major, both triangles filled) may be null
int orc_isam2_marginal_factor(void* h, int i, uint64_t* keys, int* dims, double* info_colmajor) {
  auto& S = ((ISAM2Handle*)h)->S;
  if (i < 0 || i >= (int)S.nonlinearFactors.size() || S.removedFactor[i] || !S.isContainer[i]) return -1;
  const GFactor& g = S.linearFactors[i];
  if (keys) std::copy(g.keys.begin(), g.keys.end(), keys);
  if (dims) std::copy(g.dims.begin(), g.dims.end(), dims);
  if (info_colmajor) {
    const int N = g.info.r;
    for (int c = 0; c < N; c++)
      for (int r = 0; r < N; r++) info_colmajor[(size_t)c * N + r] = r <= c ? g.info(r, c) : g.info(c, r);
  }
  return (int)g.keys.size();
}
// ISAM2::getFixedVariables() (ISAM2.h:259), ascending; returns their number
int orc_isam2_fixed_variables(void* h, uint64_t* keys_out) {
  auto& S = ((ISAM2Handle*)h)->S;
  if (keys_out) std::copy(S.fixedVariables.begin(), S.fixedVariables.end(), keys_out);
  return (int)S.fixedVariables.size();
}

// getFactorsUnsafe().linearize(getLinearizationPoint())->augmentedHessian() with the variables ascending by key: (D+1)^2 doubles,
// symmetric (the reference's own marginalizeLeaves checks compare this, tests/testGaussianISAM2.cpp:617-660); returns D
int orc_isam2_graph_hessian(void* h, double* out) {
  auto& S = ((ISAM2Handle*)h)->S;
  std::map<Key, int> off;
  int D = 0;
  for (auto& kv : S.theta) {
    off[kv.first] = D;
    D += kVarDim[kv.second.type];
  }
  if (!out) return D;
  const int N = D + 1;
  std::fill(out, out + (size_t)N * N, 0.0);
  for (size_t i = 0; i < S.nonlinearFactors.size(); i++) {
    if (S.removedFactor[i]) continue;
    const GFactor g = S.isContainer[i] ? S.linearFactors[i] : linearize_factor(S.nonlinearFactors[i], S.theta);
    std::vector<int> col;
    for (size_t k = 0; k < g.keys.size(); k++)
      for (int d = 0; d < g.dims[k]; d++) col.push_back(off.at(g.keys[k]) + d);
    col.push_back(D);
    const int n = (int)col.size();
    for (int a = 0; a < n; a++)
      for (int b = a; b < n; b++) {
        double v = 0;
        if (g.hessian) v = g.info(a, b);
        else
          for (int r = 0; r < g.Ab.r; r++) v += g.Ab(r, a) * g.Ab(r, b);
        out[(size_t)col[b] * N + col[a]] += v;
        if (col[a] != col[b]) out[(size_t)col[a] * N + col[b]] += v;
      }
  }
  return D;
}

int orc_isam2_num_variables(void* h) { return (int)((ISAM2Handle*)h)->S.theta.size(); }
int orc_isam2_num_factors(void* h) { return (int)((ISAM2Handle*)h)->S.nonlinearFactors.size(); }

// which: 0 = calculateEstimate, 1 = calculateBestEstimate, 2 = getLinearizationPoint.  Variables ascending by key;
// types_out / packed_out (kVarStore doubles each) may be null.
int orc_isam2_values(void* h, int which, uint64_t* keys_out, int* types_out, double* packed_out) {
  auto& S = ((ISAM2Handle*)h)->S;
  try {
    const Values v = which == 2 ? S.theta : isam2_calculate_estimate(S, which == 1);
    for (auto& kv : v) {
      if (keys_out) *keys_out++ = kv.first;
      if (types_out) *types_out++ = kv.second.type;
      if (packed_out) {
        std::memcpy(packed_out, kv.second.v, kVarStore[kv.second.type] * sizeof(double));
        packed_out += kVarStore[kv.second.type];
      }
    }
  } catch (const Indeterminate&) {
    return 1;
  }
  return 0;
}

// getDelta(), ascending by key
int orc_isam2_delta(void* h, double* out) {
  auto& S = ((ISAM2Handle*)h)->S;
  if (!S.deltaReplacedMask.empty()) isam2_update_delta(S, false);
  for (auto& kv : S.delta)
    for (double x : kv.second) *out++ = x;
  return 0;
}

// depth-first snapshot of the Bayes tree (roots in order, children in order); returns the number of cliques
int orc_isam2_snapshot(void* hh) {
  auto* h = (ISAM2Handle*)hh;
  h->snap.clear();
  h->snapParent.clear();
  for (auto& r : h->S.roots) isam2_collect(r, &h->snap);
  for (auto& c : h->snap) {
    int par = -1;
    if (auto p = c->parent.lock())
      par = (int)(std::find(h->snap.begin(), h->snap.end(), p) - h->snap.begin());
    h->snapParent.push_back(par);
  }
  return (int)h->snap.size();
}
// info5: nkeys, nfrontal keys, nf (rows), n (cols), parent (index in the snapshot, -1 root)
int orc_isam2_clique_info(void* hh, int i, int* info5) {
  auto* h = (ISAM2Handle*)hh;
  const IClique& c = *h->snap.at(i);
  info5[0] = (int)c.keys.size();
  info5[1] = c.nFrontal;
  info5[2] = c.RSd.r;
  info5[3] = c.RSd.c;
  info5[4] = h->snapParent.at(i);
  return 0;
}
int orc_isam2_clique_get(void* hh, int i, uint64_t* keys, double* RSd_colmajor) {
  const IClique& c = *((ISAM2Handle*)hh)->snap.at(i);
  if (keys) std::memcpy(keys, c.keys.data(), c.keys.size() * sizeof(uint64_t));
  if (RSd_colmajor) std::memcpy(RSd_colmajor, c.RSd.a.data(), c.RSd.a.size() * sizeof(double));
  return 0;
}

}  // extern "C"
