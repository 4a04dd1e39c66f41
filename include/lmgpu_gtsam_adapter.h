// lmgpu_gtsam_adapter.h — reference-side binding of the C ABI in lmgpu.h.
//
// Header-only adapter a GTSAM maintainer adds next to gtsam/nonlinear/LevenbergMarquardtOptimizer.h: a subclass that
// overrides the reference's own extension points (virtual iterate(), LevenbergMarquardtOptimizer.h:103; the pattern is
// the `IterativeLM` subclass of tests/testNonlinearOptimizer.cpp:507-528) and forwards the hot path to liblmgpu.so.
// It is NOT compiled in this repository (GTSAM itself is not buildable here); it documents
// exactly how the entry points of lmgpu.h bind.  Supported factor types are the ones of SURVEY section 8a; any other
// factor makes the constructor throw, so a caller can fall back to the stock optimizer explicitly.
// GaussNewtonOptimizer and DoglegOptimizer bind the same way: the same constructor body, and iterate() forwarding to
// lmgpu_gn_iterate / lmgpu_dl_iterate (the trust radius of DoglegState travels in lmgpu_lm_state::lambda).
#pragma once

#include <gtsam/geometry/Cal3Bundler.h>
#include <gtsam/geometry/PinholeCamera.h>
#include <gtsam/linear/linearExceptions.h>
#include <gtsam/nonlinear/LevenbergMarquardtOptimizer.h>
#include <gtsam/nonlinear/internal/LevenbergMarquardtState.h>
#include <gtsam/slam/BetweenFactor.h>
#include <gtsam/slam/GeneralSFMFactor.h>

#include <map>
#include <stdexcept>
#include <tuple>
#include <vector>

#include "lmgpu.h"

namespace gtsam {

class GpuLevenbergMarquardtOptimizer : public LevenbergMarquardtOptimizer {
  typedef PinholeCamera<Cal3Bundler> Camera;
  typedef GeneralSFMFactor<Camera, Point3> SfmFactor;
  lmgpu_handle* h_ = nullptr;
  std::vector<Key> slotKey_;          // slot -> key (elimination order)
  std::vector<int32_t> slotType_;
  lmgpu_lm_params cp_;

  static void check(int rc, lmgpu_handle* h, Key firstKeyOf(int)) { (void)firstKeyOf; if (rc == LMGPU_OK) return;
    throw std::runtime_error(std::string("lmgpu: ") + lmgpu_last_error(h)); }

  static void packPose3(const Pose3& p, double* v) {
    const Matrix3 R = p.rotation().matrix();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) v[3 * i + j] = R(i, j);
    v[9] = p.x(); v[10] = p.y(); v[11] = p.z();
  }

 public:
  GpuLevenbergMarquardtOptimizer(const NonlinearFactorGraph& graph, const Values& initial, const LevenbergMarquardtParams& params, int device = 0)
      : LevenbergMarquardtOptimizer(graph, initial, params) {
    // 1. variables in elimination order (params_.ordering was fixed by the base class, LevenbergMarquardtParams.h:112-117)
    const Ordering& ordering = *params_.ordering;
    std::map<Key, int32_t> slot;
    std::vector<uint64_t> keys;
    for (Key k : ordering) {
      slot[k] = (int32_t)keys.size();
      keys.push_back(k);
      const Value& v = initial.at(k);
      int32_t t;
      if (dynamic_cast<const GenericValue<Pose2>*>(&v)) t = LMGPU_POSE2;
      else if (dynamic_cast<const GenericValue<Pose3>*>(&v)) t = LMGPU_POSE3;
      else if (dynamic_cast<const GenericValue<Point3>*>(&v)) t = LMGPU_POINT3;
      else if (dynamic_cast<const GenericValue<Camera>*>(&v)) t = LMGPU_CAM_BUNDLER;
      else if (dynamic_cast<const GenericValue<Point2>*>(&v)) t = LMGPU_POINT2;
      else throw std::invalid_argument("GpuLevenbergMarquardtOptimizer: unsupported variable type");
      slotType_.push_back(t);
    }
    slotKey_.assign(keys.begin(), keys.end());
    lmgpu_config cfg{device, 0, 1, 0};
    if (lmgpu_create(&cfg, &h_) != LMGPU_OK) throw std::runtime_error("lmgpu_create failed (no HIP device?)");
    lmgpu_set_variables(h_, (int32_t)keys.size(), keys.data(), slotType_.data());

    // 2. factors, bucketed by (type, noise kind); graph index = position in the NonlinearFactorGraph
    struct B { std::vector<int32_t> gi, slots; std::vector<double> meas, noise; };
    std::map<std::tuple<int, int, int, double>, B> buckets;  // (factor type, noise kind, m-estimator, its constant)
    for (size_t i = 0; i < graph.size(); i++) {
      if (!graph[i]) continue;
      auto nm = std::dynamic_pointer_cast<NoiseModelFactor>(graph[i]);
      if (!nm) throw std::invalid_argument("GpuLevenbergMarquardtOptimizer: unsupported factor");
      // noise: Unit / Diagonal (inverse sigmas) / Gaussian (R row-major)
      int kind = LMGPU_N_UNIT; std::vector<double> nz;
      auto model = nm->noiseModel();
      // noiseModel::Robust: unwrap into (m-estimator id, constant) + the Gaussian model underneath (lmgpu_add_factor_bucket_robust);
      // only the default Block reweighting scheme is bound
      int rkind = LMGPU_ROBUST_NONE; double rk = 0.0;
      if (auto rob = std::dynamic_pointer_cast<noiseModel::Robust>(model)) {
        const auto est = rob->robust();
        if (auto e1 = std::dynamic_pointer_cast<noiseModel::mEstimator::Huber>(est)) { rkind = LMGPU_ROBUST_HUBER; rk = e1->modelParameter(); }
        else if (auto e2 = std::dynamic_pointer_cast<noiseModel::mEstimator::Cauchy>(est)) { rkind = LMGPU_ROBUST_CAUCHY; rk = e2->modelParameter(); }
        else if (auto e3 = std::dynamic_pointer_cast<noiseModel::mEstimator::Tukey>(est)) { rkind = LMGPU_ROBUST_TUKEY; rk = e3->modelParameter(); }
        else if (auto e4 = std::dynamic_pointer_cast<noiseModel::mEstimator::GemanMcClure>(est)) { rkind = LMGPU_ROBUST_GEMAN_MCCLURE; rk = e4->modelParameter(); }
        else if (auto e5 = std::dynamic_pointer_cast<noiseModel::mEstimator::Welsch>(est)) { rkind = LMGPU_ROBUST_WELSCH; rk = e5->modelParameter(); }
        else if (auto e6 = std::dynamic_pointer_cast<noiseModel::mEstimator::Fair>(est)) { rkind = LMGPU_ROBUST_FAIR; rk = e6->modelParameter(); }
        else if (auto e7 = std::dynamic_pointer_cast<noiseModel::mEstimator::DCS>(est)) { rkind = LMGPU_ROBUST_DCS; rk = e7->modelParameter(); }
        else if (auto e8 = std::dynamic_pointer_cast<noiseModel::mEstimator::L2WithDeadZone>(est)) { rkind = LMGPU_ROBUST_L2_WITH_DEAD_ZONE; rk = e8->modelParameter(); }
        else throw std::invalid_argument("GpuLevenbergMarquardtOptimizer: m-estimator not bound");
        model = rob->noise();
      }
      if (model && !model->isUnit()) {
        if (auto d = std::dynamic_pointer_cast<noiseModel::Diagonal>(model)) { kind = LMGPU_N_DIAG; const Vector s = d->invsigmas(); nz.assign(s.data(), s.data() + s.size()); }
        else if (auto g = std::dynamic_pointer_cast<noiseModel::Gaussian>(model)) { kind = LMGPU_N_GAUSS; const Matrix R = g->R(); for (int r = 0; r < R.rows(); r++) for (int c = 0; c < R.cols(); c++) nz.push_back(R(r, c)); }
        else throw std::invalid_argument("GpuLevenbergMarquardtOptimizer: unsupported noise model (constrained)");
      }
      int type; std::vector<double> m;
      if (auto f = std::dynamic_pointer_cast<SfmFactor>(graph[i])) {
        type = LMGPU_F_SFM;
        const Cal3Bundler& K = initial.at<Camera>(f->key1()).calibration();
        m = {f->measured().x() - K.px(), f->measured().y() - K.py()};  // fold the constant principal point (lmgpu.h)
      } else if (auto f3 = std::dynamic_pointer_cast<BetweenFactor<Pose3>>(graph[i])) {
        type = LMGPU_F_BETWEEN_POSE3; m.resize(12); packPose3(f3->measured(), m.data());
      } else if (auto f2 = std::dynamic_pointer_cast<BetweenFactor<Pose2>>(graph[i])) {
        type = LMGPU_F_BETWEEN_POSE2; m = {f2->measured().x(), f2->measured().y(), f2->measured().theta()};
      } else {
        // PriorFactor<T>, GenericProjectionFactor<Pose3,Point3,Cal3_S2>, BearingRangeFactor<Pose2,Point2> (measured().bearing().theta(),
        // measured().range()): same pattern (measurement packing in lmgpu.h)
        throw std::invalid_argument("GpuLevenbergMarquardtOptimizer: factor type not bound in this sketch");
      }
      B& b = buckets[std::make_tuple(type, kind, rkind, rk)];
      b.gi.push_back((int32_t)i);
      for (Key k : nm->keys()) b.slots.push_back(slot.at(k));
      b.meas.insert(b.meas.end(), m.begin(), m.end());
      b.noise.insert(b.noise.end(), nz.begin(), nz.end());
    }
    for (auto& kv : buckets)
      if (lmgpu_add_factor_bucket_robust(h_, std::get<0>(kv.first), (int32_t)kv.second.gi.size(), kv.second.gi.data(), kv.second.slots.data(),
                                         kv.second.meas.data(), std::get<1>(kv.first), kv.second.noise.empty() ? nullptr : kv.second.noise.data(),
                                         std::get<2>(kv.first), std::get<3>(kv.first)) != LMGPU_OK)
        throw std::runtime_error(lmgpu_last_error(h_));
    if (lmgpu_finalize_structure(h_) != LMGPU_OK) throw std::runtime_error(lmgpu_last_error(h_));
    uploadValues(initial);
    cp_ = lmgpu_lm_params{(int32_t)params.maxIterations, params.relativeErrorTol, params.absoluteErrorTol, params.errorTol, params.lambdaInitial,
                          params.lambdaFactor, params.lambdaUpperBound, params.lambdaLowerBound, params.minModelFidelity,
                          params.diagonalDamping, params.useFixedLambdaFactor, params.minDiagonal, params.maxDiagonal};
  }
  ~GpuLevenbergMarquardtOptimizer() override { if (h_) lmgpu_destroy(h_); }

  /// drop-in for LevenbergMarquardtOptimizer::iterate() (LevenbergMarquardtOptimizer.cpp:273-308)
  GaussianFactorGraph::shared_ptr iterate() override {
    auto cur = static_cast<const internal::LevenbergMarquardtState*>(state_.get());
    lmgpu_lm_state st{cur->error, cur->lambda, cur->currentFactor, (int32_t)cur->iterations, cur->totalNumberInnerIterations};
    const int rc = lmgpu_iterate(h_, &cp_, &st);
    if (rc != LMGPU_OK) throw std::runtime_error(lmgpu_last_error(h_));
    // new state: values come back from the device only when the caller asks for them (values()) — here eagerly:
    state_.reset(new internal::LevenbergMarquardtState(downloadValues(cur->values), st.error, st.lambda, st.currentFactor,
                                                       (unsigned)st.iterations, (unsigned)st.totalNumberInnerIterations));
    return GaussianFactorGraph::shared_ptr();  // in-tree callers ignore the returned linear graph
  }

  /// drop-in for NonlinearOptimizer::solve() (NonlinearOptimizer.cpp:132-178) on the last linearization
  VectorValues solveDamped(double lambda) {
    std::vector<double> d((size_t)lmgpu_total_dim(h_));
    double e0, e1;
    const int rc = lmgpu_solve(h_, lambda, params_.diagonalDamping, params_.minDiagonal, params_.maxDiagonal, d.data(), &e0, &e1);
    if (rc == LMGPU_INDETERMINATE) throw IndeterminantLinearSystemException(slotKey_[lmgpu_last_failed_slot(h_)]);
    if (rc != LMGPU_OK) throw std::runtime_error(lmgpu_last_error(h_));
    VectorValues out; size_t o = 0;
    static const int dim[4] = {3, 6, 3, 9};
    for (size_t s = 0; s < slotKey_.size(); s++) { out.insert(slotKey_[s], Eigen::Map<Vector>(d.data() + o, dim[slotType_[s]])); o += dim[slotType_[s]]; }
    return out;
  }

 private:
  void uploadValues(const Values& v) {
    std::vector<double> packed((size_t)lmgpu_total_store(h_)); size_t o = 0;
    for (size_t s = 0; s < slotKey_.size(); s++) {
      const Key k = slotKey_[s];
      switch (slotType_[s]) {
        case LMGPU_POSE2: { const Pose2& p = v.at<Pose2>(k); packed[o] = p.x(); packed[o + 1] = p.y(); packed[o + 2] = p.theta(); o += 3; break; }
        case LMGPU_POSE3: packPose3(v.at<Pose3>(k), &packed[o]); o += 12; break;
        case LMGPU_POINT3: { const Point3& p = v.at<Point3>(k); packed[o] = p.x(); packed[o + 1] = p.y(); packed[o + 2] = p.z(); o += 3; break; }
        case LMGPU_POINT2: { const Point2& p = v.at<Point2>(k); packed[o] = p.x(); packed[o + 1] = p.y(); o += 2; break; }
        default: { const Camera& c = v.at<Camera>(k); packPose3(c.pose(), &packed[o]); packed[o + 12] = c.calibration().fx(); packed[o + 13] = c.calibration().k1(); packed[o + 14] = c.calibration().k2(); o += 15; }
      }
    }
    if (lmgpu_set_values(h_, packed.data()) != LMGPU_OK) throw std::runtime_error(lmgpu_last_error(h_));
  }
  Values downloadValues(const Values& like) {
    std::vector<double> packed((size_t)lmgpu_total_store(h_));
    if (lmgpu_get_values(h_, packed.data()) != LMGPU_OK) throw std::runtime_error(lmgpu_last_error(h_));
    Values out; size_t o = 0;
    for (size_t s = 0; s < slotKey_.size(); s++) {
      const Key k = slotKey_[s]; const double* p = &packed[o];
      auto pose3 = [&](const double* q) { Matrix3 R; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R(i, j) = q[3 * i + j]; return Pose3(Rot3(R), Point3(q[9], q[10], q[11])); };
      switch (slotType_[s]) {
        case LMGPU_POSE2: out.insert(k, Pose2(p[0], p[1], p[2])); o += 3; break;
        case LMGPU_POSE3: out.insert(k, pose3(p)); o += 12; break;
        case LMGPU_POINT3: out.insert(k, Point3(p[0], p[1], p[2])); o += 3; break;
        case LMGPU_POINT2: out.insert(k, Point2(p[0], p[1])); o += 2; break;
        default: { const Cal3Bundler& K0 = like.at<Camera>(k).calibration(); out.insert(k, Camera(pose3(p), Cal3Bundler(p[12], p[13], p[14], K0.px(), K0.py()))); o += 15; }
      }
    }
    return out;
  }
};

}  // namespace gtsam
