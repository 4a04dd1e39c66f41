// lmgpu_gtsam_adapter.h — reference-side binding of the C ABI in lmgpu.h.
//
// Header-only adapter a GTSAM maintainer adds next to gtsam/nonlinear/LevenbergMarquardtOptimizer.h: subclasses that override
// the reference's own extension points and forward the hot path to liblmgpu.so:
//   iterate()    virtual, gtsam/nonlinear/NonlinearOptimizer.h:136, LM override LevenbergMarquardtOptimizer.h:103
//   linearize()  virtual, gtsam/nonlinear/LevenbergMarquardtOptimizer.h:113 ("can be overwritten")
//   solve()      virtual, gtsam/nonlinear/NonlinearOptimizer.h:129-130 (precedent: IterativeLM, tests/testNonlinearOptimizer.cpp:507-528)
//
// This file only EXTRACTS numbers from GTSAM objects; slot assignment, bucketing, packing, the call order of the C ABI and
// the status -> exception mapping are in lmgpu_adapter_core.hpp, which has no GTSAM include and is compiled and tested in the
// repository (tests/cpp/adapter_harness.cpp).  This file is type-checked against the reference's headers by
// tests/test_adapter_header_compiles.py (g++ -fsyntax-only; the two CMake-generated headers GTSAM's sources include are written into
// the test's temporary directory); it cannot be LINKED there (libgtsam is not buildable in the repository's container).
//
// Two modes (constructor argument):
//   WholeIterate (default)  iterate() = lmgpu_iterate: the whole tryLambda loop on device-resident data; values come back
//                           to the host once per outer iteration.
//   Piecewise               iterate() is the reference's own (LevenbergMarquardtOptimizer.cpp:273-308); only its two virtual
//                           calls are replaced: linearize() -> lmgpu_linearize (+ download of the whitened Jacobians as
//                           JacobianFactors, which tryLambda needs for linear.error), solve() -> lmgpu_solve at state lambda.
#pragma once

#include <gtsam/base/GenericValue.h>
#include <gtsam/geometry/BearingRange.h>
#include <gtsam/geometry/Cal3Bundler.h>
#include <gtsam/geometry/Cal3_S2.h>
#include <gtsam/geometry/PinholeCamera.h>
#include <gtsam/geometry/Point2.h>
#include <gtsam/geometry/Point3.h>
#include <gtsam/geometry/Pose2.h>
#include <gtsam/geometry/Pose3.h>
#include <gtsam/geometry/Rot2.h>
#include <gtsam/geometry/Rot3.h>
#include <gtsam/inference/Ordering.h>
#include <gtsam/linear/GaussianFactorGraph.h>
#include <gtsam/linear/JacobianFactor.h>
#include <gtsam/linear/NoiseModel.h>
#include <gtsam/linear/VectorValues.h>
#include <gtsam/linear/linearExceptions.h>
#include <gtsam/nonlinear/DoglegOptimizer.h>
#include <gtsam/nonlinear/GaussNewtonOptimizer.h>
#include <gtsam/nonlinear/ISAM2.h>
#include <gtsam/3rdparty/CCOLAMD/Include/ccolamd.h>
#include <gtsam/nonlinear/LevenbergMarquardtOptimizer.h>
#include <gtsam/nonlinear/NonlinearFactorGraph.h>
#include <gtsam/nonlinear/PriorFactor.h>
#include <gtsam/nonlinear/Values.h>
#include <gtsam/nonlinear/internal/LevenbergMarquardtState.h>
#include <gtsam/nonlinear/internal/NonlinearOptimizerState.h>
#include <gtsam/sam/BearingRangeFactor.h>
#include <gtsam/slam/BetweenFactor.h>
#include <gtsam/slam/GeneralSFMFactor.h>
#include <gtsam/slam/ProjectionFactor.h>

#include <algorithm>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "lmgpu_adapter_core.hpp"

namespace gtsam {

namespace lmgpu_detail {

typedef PinholeCamera<Cal3Bundler> Camera;
typedef GeneralSFMFactor<Camera, Point3> SfmFactor;
typedef GenericProjectionFactor<Pose3, Point3, Cal3_S2> ProjFactor;

inline void packPose3(const Pose3& p, double* v) {  // R row-major 9, t 3 (lmgpu.h, POSE3)
  const Matrix3 R = p.rotation().matrix();
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) v[3 * i + j] = R(i, j);
  v[9] = p.x();
  v[10] = p.y();
  v[11] = p.z();
}
inline Pose3 unpackPose3(const double* q) {
  Matrix3 R;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R(i, j) = q[3 * i + j];
  return Pose3(Rot3(R), Point3(q[9], q[10], q[11]));
}

inline int32_t variableType(const Value& v) {  // gtsam/base/GenericValue.h
  if (dynamic_cast<const GenericValue<Pose2>*>(&v)) return LMGPU_POSE2;
  if (dynamic_cast<const GenericValue<Pose3>*>(&v)) return LMGPU_POSE3;
  if (dynamic_cast<const GenericValue<Point3>*>(&v)) return LMGPU_POINT3;
  if (dynamic_cast<const GenericValue<Camera>*>(&v)) return LMGPU_CAM_BUNDLER;
  if (dynamic_cast<const GenericValue<Point2>*>(&v)) return LMGPU_POINT2;
  if (dynamic_cast<const GenericValue<Cal3_S2>*>(&v)) return LMGPU_CAL3_S2;  // self-calibration: the calibration is a variable
  throw std::invalid_argument("lmgpu adapter: unsupported variable type");
}

/// noise model -> (kind, data, m-estimator, constant); gtsam/linear/NoiseModel.h (Unit :617-660, Diagonal::invsigmas :361,
/// Gaussian::R :263, Robust::robust / noise :705-708), gtsam/linear/LossFunctions.h (modelParameter, reweightScheme :82)
struct Noise {
  int32_t kind = LMGPU_N_UNIT, robust = LMGPU_ROBUST_NONE;
  double robustK = 0.0;
  std::vector<double> data;
};
inline Noise extractNoise(SharedNoiseModel model) {
  Noise out;
  if (auto rob = std::dynamic_pointer_cast<noiseModel::Robust>(model)) {
    const auto est = rob->robust();
    if (est->reweightScheme() != noiseModel::mEstimator::Base::Block)
      throw std::invalid_argument("lmgpu adapter: only the Block reweighting scheme is bound");
    using namespace noiseModel::mEstimator;
    if (auto e1 = std::dynamic_pointer_cast<Huber>(est)) { out.robust = LMGPU_ROBUST_HUBER; out.robustK = e1->modelParameter(); }
    else if (auto e2 = std::dynamic_pointer_cast<Cauchy>(est)) { out.robust = LMGPU_ROBUST_CAUCHY; out.robustK = e2->modelParameter(); }
    else if (auto e3 = std::dynamic_pointer_cast<Tukey>(est)) { out.robust = LMGPU_ROBUST_TUKEY; out.robustK = e3->modelParameter(); }
    else if (auto e4 = std::dynamic_pointer_cast<GemanMcClure>(est)) { out.robust = LMGPU_ROBUST_GEMAN_MCCLURE; out.robustK = e4->modelParameter(); }
    else if (auto e5 = std::dynamic_pointer_cast<Welsch>(est)) { out.robust = LMGPU_ROBUST_WELSCH; out.robustK = e5->modelParameter(); }
    else if (auto e6 = std::dynamic_pointer_cast<Fair>(est)) { out.robust = LMGPU_ROBUST_FAIR; out.robustK = e6->modelParameter(); }
    else if (auto e7 = std::dynamic_pointer_cast<DCS>(est)) { out.robust = LMGPU_ROBUST_DCS; out.robustK = e7->modelParameter(); }
    else if (auto e8 = std::dynamic_pointer_cast<L2WithDeadZone>(est)) { out.robust = LMGPU_ROBUST_L2_WITH_DEAD_ZONE; out.robustK = e8->modelParameter(); }
    else throw std::invalid_argument("lmgpu adapter: m-estimator not bound");
    model = rob->noise();
  }
  if (!model || model->isUnit()) return out;
  if (model->isConstrained()) throw std::invalid_argument("lmgpu adapter: constrained noise models take the QR path (not bound)");
  if (auto d = std::dynamic_pointer_cast<noiseModel::Diagonal>(model)) {  // Isotropic is a Diagonal
    out.kind = LMGPU_N_DIAG;
    const Vector& s = d->invsigmas();
    out.data.assign(s.data(), s.data() + s.size());
  } else if (auto g = std::dynamic_pointer_cast<noiseModel::Gaussian>(model)) {
    out.kind = LMGPU_N_GAUSS;
    const Matrix R = g->R();
    for (int r = 0; r < R.rows(); r++)
      for (int c = 0; c < R.cols(); c++) out.data.push_back(R(r, c));
  } else {
    throw std::invalid_argument("lmgpu adapter: unsupported noise model");
  }
  return out;
}

/// one nonlinear factor -> (factor type, measurement doubles) in the packing lmgpu.h documents; false if the type is not bound
inline bool extractFactor(const NonlinearFactor::shared_ptr& f, const Values& initial, int32_t* type, std::vector<double>* m) {
  if (auto s = std::dynamic_pointer_cast<SfmFactor>(f)) {  // gtsam/slam/GeneralSFMFactor.h:70-206
    *type = LMGPU_F_SFM;
    const Cal3Bundler& K = initial.at<Camera>(s->key1()).calibration();
    *m = {s->measured().x() - K.px(), s->measured().y() - K.py()};  // fold the constant principal point (lmgpu.h, CAM_BUNDLER)
  } else if (auto b3 = std::dynamic_pointer_cast<BetweenFactor<Pose3>>(f)) {  // gtsam/slam/BetweenFactor.h
    *type = LMGPU_F_BETWEEN_POSE3;
    m->resize(12);
    packPose3(b3->measured(), m->data());
  } else if (auto b2 = std::dynamic_pointer_cast<BetweenFactor<Pose2>>(f)) {
    *type = LMGPU_F_BETWEEN_POSE2;
    *m = {b2->measured().x(), b2->measured().y(), b2->measured().theta()};
  } else if (auto p2 = std::dynamic_pointer_cast<PriorFactor<Pose2>>(f)) {  // gtsam/nonlinear/PriorFactor.h:104 prior()
    *type = LMGPU_F_PRIOR_POSE2;
    *m = {p2->prior().x(), p2->prior().y(), p2->prior().theta()};
  } else if (auto p3 = std::dynamic_pointer_cast<PriorFactor<Pose3>>(f)) {
    *type = LMGPU_F_PRIOR_POSE3;
    m->resize(12);
    packPose3(p3->prior(), m->data());
  } else if (auto pp = std::dynamic_pointer_cast<PriorFactor<Point3>>(f)) {
    *type = LMGPU_F_PRIOR_POINT3;
    *m = {pp->prior().x(), pp->prior().y(), pp->prior().z()};
  } else if (auto pc = std::dynamic_pointer_cast<PriorFactor<Camera>>(f)) {  // graph.addPrior(C(0), camera, ...) SFMExample_bal.cpp:67
    *type = LMGPU_F_PRIOR_CAM;
    m->resize(15);
    packPose3(pc->prior().pose(), m->data());
    (*m)[12] = pc->prior().calibration().fx();
    (*m)[13] = pc->prior().calibration().k1();
    (*m)[14] = pc->prior().calibration().k2();
  } else if (auto pr = std::dynamic_pointer_cast<ProjFactor>(f)) {  // gtsam/slam/ProjectionFactor.h:169-187
    if (pr->throwCheirality()) {
      // the reference lets CheiralityException escape from linearize in this configuration; the device path zeroes the factor
      // like the default build of GeneralSFMFactor does.  Bind only the non-throwing form.
      throw std::invalid_argument("lmgpu adapter: GenericProjectionFactor with throwCheirality is not bound");
    }
    const Cal3_S2& K = *pr->calibration();
    *m = {pr->measured().x(), pr->measured().y(), K.fx(), K.fy(), K.skew(), K.px(), K.py()};
    if (pr->body_P_sensor()) {
      *type = LMGPU_F_PROJECTION_BPS;
      m->resize(19);
      packPose3(*pr->body_P_sensor(), m->data() + 7);
    } else {
      *type = LMGPU_F_PROJECTION;
    }
  } else if (auto s2 = std::dynamic_pointer_cast<GeneralSFMFactor2<Cal3_S2>>(f)) {  // gtsam/slam/GeneralSFMFactor.h:208-262 (keys: pose, point, K)
    *type = LMGPU_F_SFM2;
    *m = {s2->measured().x(), s2->measured().y()};
  } else if (auto pk = std::dynamic_pointer_cast<PriorFactor<Cal3_S2>>(f)) {  // graph.addPrior(Symbol('K', 0), K, calNoise) SelfCalibrationExample.cpp:83
    *type = LMGPU_F_PRIOR_CAL3_S2;
    const Cal3_S2& K = pk->prior();
    *m = {K.fx(), K.fy(), K.skew(), K.px(), K.py()};
  } else if (auto br = std::dynamic_pointer_cast<BearingRangeFactor<Pose2, Point2>>(f)) {  // gtsam/sam/BearingRangeFactor.h:33-77
    *type = LMGPU_F_BEARING_RANGE_2D;
    *m = {br->measured().bearing().theta(), br->measured().range()};  // ExpressionFactor::measured :82, BearingRange.h:73-76
  } else {
    return false;
  }
  return true;
}

/// shared by the three optimizers: the device-resident problem + Values / VectorValues marshalling
class Device {
 public:
  Device(const NonlinearFactorGraph& graph, const Values& initial, const Ordering& ordering, int device) : p_(device) {
    std::vector<uint64_t> keys(ordering.begin(), ordering.end());
    std::vector<int32_t> types;
    types.reserve(keys.size());
    for (Key k : ordering) types.push_back(variableType(initial.at(k)));
    p_.setVariables(keys, types);
    for (size_t i = 0; i < graph.size(); i++) {
      if (!graph[i]) continue;  // NonlinearFactorGraph::linearize keeps null factors null (NonlinearFactorGraph.cpp:214-233)
      auto nm = std::dynamic_pointer_cast<NoiseModelFactor>(graph[i]);
      if (!nm) throw std::invalid_argument("lmgpu adapter: only NoiseModelFactors are bound");
      int32_t type;
      std::vector<double> meas;
      if (!extractFactor(graph[i], initial, &type, &meas)) throw std::invalid_argument("lmgpu adapter: factor type not bound");
      const Noise nz = extractNoise(nm->noiseModel());
      std::vector<uint64_t> fk(nm->keys().begin(), nm->keys().end());
      p_.addFactor(type, (int32_t)i, fk.data(), meas.data(), nz.kind, nz.data.empty() ? nullptr : nz.data.data(), nz.robust, nz.robustK);
    }
    p_.finalize();
    upload(initial);
  }

  lmgpu_adapter::Problem& problem() { return p_; }
  const lmgpu_adapter::Problem& problem() const { return p_; }

  void upload(const Values& v) const {
    std::vector<double> packed((size_t)p_.totalStore());
    for (size_t s = 0; s < p_.numVariables(); s++) {
      const Key k = p_.keyOfSlot((int)s);
      double* o = &packed[p_.valueOffset((int)s)];
      switch (p_.typeOfSlot((int)s)) {
        case LMGPU_POSE2: { const Pose2& q = v.at<Pose2>(k); o[0] = q.x(); o[1] = q.y(); o[2] = q.theta(); break; }
        case LMGPU_POSE3: packPose3(v.at<Pose3>(k), o); break;
        case LMGPU_POINT3: { const Point3& q = v.at<Point3>(k); o[0] = q.x(); o[1] = q.y(); o[2] = q.z(); break; }
        case LMGPU_POINT2: { const Point2& q = v.at<Point2>(k); o[0] = q.x(); o[1] = q.y(); break; }
        case LMGPU_CAL3_S2: { const Cal3_S2& q = v.at<Cal3_S2>(k); o[0] = q.fx(); o[1] = q.fy(); o[2] = q.skew(); o[3] = q.px(); o[4] = q.py(); break; }
        default: {
          const Camera& c = v.at<Camera>(k);
          packPose3(c.pose(), o);
          o[12] = c.calibration().fx(); o[13] = c.calibration().k1(); o[14] = c.calibration().k2();
        }
      }
    }
    const_cast<lmgpu_adapter::Problem&>(p_).setValues(packed);
  }

  /// `like` supplies what does not travel: the constant principal point of every Cal3Bundler
  Values download(const Values& like) const {
    const std::vector<double> packed = p_.getValues();
    Values out;
    for (size_t s = 0; s < p_.numVariables(); s++) {
      const Key k = p_.keyOfSlot((int)s);
      const double* q = &packed[p_.valueOffset((int)s)];
      switch (p_.typeOfSlot((int)s)) {
        case LMGPU_POSE2: out.insert(k, Pose2(q[0], q[1], q[2])); break;
        case LMGPU_POSE3: out.insert(k, unpackPose3(q)); break;
        case LMGPU_POINT3: out.insert(k, Point3(q[0], q[1], q[2])); break;
        case LMGPU_POINT2: out.insert(k, Point2(q[0], q[1])); break;
        case LMGPU_CAL3_S2: out.insert(k, Cal3_S2(q[0], q[1], q[2], q[3], q[4])); break;
        default: {
          const Cal3Bundler& K0 = like.at<Camera>(k).calibration();
          out.insert(k, Camera(unpackPose3(q), Cal3Bundler(q[12], q[13], q[14], K0.px(), K0.py())));
        }
      }
    }
    return out;
  }

  VectorValues toVectorValues(const std::vector<double>& packed) const {
    VectorValues out;
    for (size_t s = 0; s < p_.numVariables(); s++) {
      const int d = lmgpu_adapter::varDim(p_.typeOfSlot((int)s));
      out.insert(p_.keyOfSlot((int)s), Vector(Eigen::Map<const Vector>(packed.data() + p_.deltaOffset((int)s), d)));
    }
    return out;
  }

  /// lmgpu_solve with the reference's exception: IndeterminantLinearSystemException(first frontal key of the failing clique)
  VectorValues solve(double lambda, const LevenbergMarquardtParams* lm) const {
    try {
      const lmgpu_adapter::SolveResult r = lm ? p_.solve(lambda, lm->diagonalDamping, lm->minDiagonal, lm->maxDiagonal) : p_.solve(lambda, false, 0.0, 0.0);
      return toVectorValues(r.delta);
    } catch (const lmgpu_adapter::Indeterminate& e) {
      throw IndeterminantLinearSystemException(e.key);
    }
  }

  /// the device-resident linearization as the GaussianFactorGraph NonlinearFactorGraph::linearize returns: one JacobianFactor
  /// per factor, already whitened (unit noise model), index-preserving
  GaussianFactorGraph::shared_ptr downloadLinearGraph(const NonlinearFactorGraph& graph) const {
    const lmgpu_adapter::Problem::LinearGraph lg = p_.jacobians();  // one device copy per factor bucket
    auto out = std::make_shared<GaussianFactorGraph>();
    out->reserve(graph.size());
    size_t next = 0;
    for (size_t i = 0; i < graph.size(); i++) {
      if (!graph[i]) { out->push_back(GaussianFactor::shared_ptr()); continue; }
      if (next >= lg.graphIndex.size() || (size_t)lg.graphIndex[next] != i) throw std::logic_error("lmgpu adapter: linear graph does not match the nonlinear graph");
      const int rows = lg.rows[next], cols = lg.cols[next];
      Eigen::Map<const Matrix> M(lg.data.data() + lg.offsets[next], rows, cols);  // column-major
      next++;
      std::vector<std::pair<Key, Matrix>> terms;
      int c0 = 0;
      for (Key k : graph[i]->keys()) {
        const int d = lmgpu_adapter::varDim(p_.typeOfSlot(p_.slotOf(k)));
        terms.emplace_back(k, M.middleCols(c0, d));
        c0 += d;
      }
      out->push_back(std::make_shared<JacobianFactor>(terms, Vector(M.col(cols - 1))));
    }
    return out;
  }

 private:
  lmgpu_adapter::Problem p_;
};

inline lmgpu_lm_params toC(const LevenbergMarquardtParams& p) {  // gtsam/nonlinear/LevenbergMarquardtParams.h:35-157
  return lmgpu_lm_params{(int32_t)p.maxIterations, p.relativeErrorTol, p.absoluteErrorTol, p.errorTol, p.lambdaInitial, p.lambdaFactor,
                         p.lambdaUpperBound, p.lambdaLowerBound, p.minModelFidelity, p.diagonalDamping ? 1 : 0, p.useFixedLambdaFactor ? 1 : 0,
                         p.minDiagonal, p.maxDiagonal};
}

}  // namespace lmgpu_detail

/// drop-in for LevenbergMarquardtOptimizer: same constructor arguments, same optimize() / iterate() / lambda() / values().
/// Unsupported factors, constrained noise, Scalar-scheme or custom m-estimators make the constructor throw
/// std::invalid_argument, so that a caller keeps the stock optimizer explicitly — there is no silent fallback.
class GpuLevenbergMarquardtOptimizer : public LevenbergMarquardtOptimizer {
 public:
  enum Mode { WholeIterate, Piecewise };

  GpuLevenbergMarquardtOptimizer(const NonlinearFactorGraph& graph, const Values& initial,
                                 const LevenbergMarquardtParams& params = LevenbergMarquardtParams(), int device = 0, Mode mode = WholeIterate)
      : LevenbergMarquardtOptimizer(graph, initial, params),  // fixes params_.ordering (LevenbergMarquardtParams.h:112-117), initial error
        dev_(graph, initial, *params_.ordering, device), cp_(lmgpu_detail::toC(params_)), mode_(mode) {}

  /// LevenbergMarquardtOptimizer::iterate() (LevenbergMarquardtOptimizer.cpp:273-308)
  GaussianFactorGraph::shared_ptr iterate() override {
    if (mode_ == Piecewise) return LevenbergMarquardtOptimizer::iterate();  // calls linearize() / solve() below
    auto cur = static_cast<const internal::LevenbergMarquardtState*>(state_.get());
    lmgpu_lm_state st{cur->error, cur->lambda, cur->currentFactor, (int32_t)cur->iterations, cur->totalNumberInnerIterations};
    try {
      dev_.problem().iterate(cp_, &st);
    } catch (const lmgpu_adapter::Indeterminate& e) {
      throw IndeterminantLinearSystemException(e.key);
    }
    state_.reset(new internal::LevenbergMarquardtState(dev_.download(cur->values), st.error, st.lambda, st.currentFactor,
                                                       (unsigned)st.iterations, (unsigned)st.totalNumberInnerIterations));
    // the hook's contract (NonlinearOptimizer.h:136, LevenbergMarquardtOptimizer.cpp:281,307; read at tests/testNonlinearOptimizer.cpp:282):
    // the linearization this iteration solved -- lmgpu_iterate leaves it on the device (it linearizes once per outer iteration, before
    // the lambda loop, and the accepted step does not touch the Jacobians).  optimize() below discards the graph like defaultOptimize
    // does (NonlinearOptimizer.cpp:92) and therefore does not fetch it.
    return discardLinear_ ? GaussianFactorGraph::shared_ptr() : dev_.downloadLinearGraph(graph_);
  }

  /// NonlinearOptimizer::optimize() (NonlinearOptimizer.h:98): defaultOptimize() calls iterate() and drops what it returns; skip the download
  const Values& optimize() override {
    discardLinear_ = true;
    try {
      defaultOptimize();
    } catch (...) {
      discardLinear_ = false;
      throw;
    }
    discardLinear_ = false;
    return values();
  }

  /// LevenbergMarquardtOptimizer::linearize() (LevenbergMarquardtOptimizer.h:113): graph_.linearize(state_->values) on the device
  GaussianFactorGraph::shared_ptr linearize() const override {
    dev_.upload(state_->values);
    dev_.problem().linearize();
    return dev_.downloadLinearGraph(graph_);
  }

  /// NonlinearOptimizer::solve() (NonlinearOptimizer.h:129-130).  tryLambda hands over buildDampedSystem(linear) — the
  /// linearization lmgpu_linearize left on the device plus lambda priors — so the damped system is rebuilt on the device from
  /// the state's lambda instead of being parsed out of `damped`.
  VectorValues solve(const GaussianFactorGraph& /*damped*/, const NonlinearOptimizerParams& /*params*/) const override {
    return dev_.solve(static_cast<const internal::LevenbergMarquardtState*>(state_.get())->lambda, &params_);
  }

  /// the whitened Jacobians of the last linearization as a GaussianFactorGraph (what iterate() returns in the reference)
  GaussianFactorGraph::shared_ptr lastLinearGraph() const { return dev_.downloadLinearGraph(graph_); }

 private:
  mutable lmgpu_detail::Device dev_;
  lmgpu_lm_params cp_;
  Mode mode_;
  bool discardLinear_ = false;
};

/// drop-in for GaussNewtonOptimizer (gtsam/nonlinear/GaussNewtonOptimizer.cpp:44-66): iterate() = lmgpu_gn_iterate
class GpuGaussNewtonOptimizer : public GaussNewtonOptimizer {
 public:
  GpuGaussNewtonOptimizer(const NonlinearFactorGraph& graph, const Values& initial, const GaussNewtonParams& params = GaussNewtonParams(),
                          int device = 0)
      : GaussNewtonOptimizer(graph, initial, params), dev_(graph, initial, *params_.ordering, device) {}

  GaussianFactorGraph::shared_ptr iterate() override {
    lmgpu_lm_state st{state_->error, 0.0, 0.0, (int32_t)state_->iterations, 0};
    try {
      dev_.problem().gnIterate(&st);
    } catch (const lmgpu_adapter::Indeterminate& e) {
      throw IndeterminantLinearSystemException(e.key);
    }
    state_.reset(new internal::NonlinearOptimizerState(dev_.download(state_->values), st.error, (unsigned)st.iterations));
    return discardLinear_ ? GaussianFactorGraph::shared_ptr() : dev_.downloadLinearGraph(graph_);  // GaussNewtonOptimizer.cpp:49,65
  }

  const Values& optimize() override {  // see GpuLevenbergMarquardtOptimizer::optimize
    discardLinear_ = true;
    try {
      defaultOptimize();
    } catch (...) {
      discardLinear_ = false;
      throw;
    }
    discardLinear_ = false;
    return values();
  }

  VectorValues solve(const GaussianFactorGraph& /*linear*/, const NonlinearOptimizerParams& /*params*/) const override {
    return dev_.solve(0.0, nullptr);
  }

 private:
  mutable lmgpu_detail::Device dev_;
  bool discardLinear_ = false;
};

/// DoglegOptimizer's state type (internal::DoglegState) is private to DoglegOptimizer.cpp:54-62, so a subclass cannot replace
/// state_; the Dogleg binding is therefore a NonlinearOptimizer of its own with the same interface (getDelta(), iterate()).
class GpuDoglegOptimizer : public NonlinearOptimizer {
  struct State : public internal::NonlinearOptimizerState {
    double delta;
    State(const Values& v, double e, double d, unsigned it = 0) : internal::NonlinearOptimizerState(v, e, it), delta(d) {}
  };

 public:
  GpuDoglegOptimizer(const NonlinearFactorGraph& graph, const Values& initial, const DoglegParams& params = DoglegParams(), int device = 0)
      : NonlinearOptimizer(graph, std::unique_ptr<internal::NonlinearOptimizerState>(new State(initial, graph.error(initial), params.deltaInitial))),
        params_(params) {
    if (!params_.ordering) params_.ordering = Ordering::Create(params_.orderingType, graph);  // DoglegOptimizer.cpp:127-131
    dev_.reset(new lmgpu_detail::Device(graph, initial, *params_.ordering, device));
  }
  double getDelta() const { return static_cast<const State*>(state_.get())->delta; }

  GaussianFactorGraph::shared_ptr iterate() override {  // DoglegOptimizer.cpp:84-126
    lmgpu_lm_state st{state_->error, getDelta(), 0.0, (int32_t)state_->iterations, 0};
    try {
      dev_->problem().dlIterate(&st);  // trust radius travels in lambda (lmgpu.h)
    } catch (const lmgpu_adapter::Indeterminate& e) {
      throw IndeterminantLinearSystemException(e.key);
    }
    state_.reset(new State(dev_->download(state_->values), st.error, st.lambda, (unsigned)st.iterations));
    return discardLinear_ ? GaussianFactorGraph::shared_ptr() : dev_->downloadLinearGraph(graph_);  // DoglegOptimizer.cpp:87,122
  }

  const Values& optimize() override {  // see GpuLevenbergMarquardtOptimizer::optimize
    discardLinear_ = true;
    try {
      defaultOptimize();
    } catch (...) {
      discardLinear_ = false;
      throw;
    }
    discardLinear_ = false;
    return values();
  }

 protected:
  const NonlinearOptimizerParams& _params() const override { return params_; }
  DoglegParams params_;
  std::unique_ptr<lmgpu_detail::Device> dev_;
  bool discardLinear_ = false;
};

/// ISAM2 on the device (lmgpu_isam2_*): the same update() / calculateEstimate() calls as gtsam::ISAM2 (gtsam/nonlinear/ISAM2.h:146-260)
/// for the parameter subset the C ABI binds (Gauss-Newton or Dogleg optimisation params, relinearization threshold as a double or per Symbol
/// character, partial relinearization check, Cholesky).  Not a
/// subclass: ISAM2 is a BayesTree<ISAM2Clique> whose cliques live on the host; here the tree lives on the device and only the
/// estimate comes back.  The constrained COLAMD of recalculate() stays on this side: the callback below is the body of
/// Ordering::ColamdConstrained (gtsam/inference/Ordering.cpp:86-108) on the arrays the library hands over.
class GpuISAM2 {
 public:
  explicit GpuISAM2(const ISAM2Params& params = ISAM2Params(), int device = 0) {
    if (params.factorization != ISAM2Params::CHOLESKY || !params.cacheLinearizedFactors)
      throw std::invalid_argument("GpuISAM2: parameter set not bound (Cholesky, cached linear factors)");
    const bool byChar = std::holds_alternative<FastMap<char, Vector>>(params.relinearizeThreshold);
    const ISAM2DoglegParams* dl = std::get_if<ISAM2DoglegParams>(&params.optimizationParams);
    lmgpu_isam2_params p{byChar ? 0.1 : std::get<double>(params.relinearizeThreshold), params.relinearizeSkip, params.enableRelinearization ? 1 : 0,
                         dl ? dl->wildfireThreshold : std::get<ISAM2GaussNewtonParams>(params.optimizationParams).wildfireThreshold};
    lmgpu_config cfg{device, 0, 1, 0};
    if (lmgpu_isam2_create(&cfg, &p, &GpuISAM2::Ccolamd, nullptr, &h_) != LMGPU_OK) {
      const std::string why = h_ ? lmgpu_isam2_last_error(h_) : "lmgpu_isam2_create failed";
      if (h_) lmgpu_isam2_destroy(h_);
      throw std::runtime_error(why);
    }
    if (dl) check(lmgpu_isam2_set_dogleg(h_, dl->initialDelta, dl->wildfireThreshold, (int32_t)dl->adaptationMode));  // ISAM2Params.h:68-110
    if (byChar) {  // ISAM2Params::relinearizeThreshold as FastMap<char, Vector> (ISAM2Params.h:139-141)
      std::string chrs;
      std::vector<int32_t> dims;
      std::vector<double> values;
      for (const auto& cv : std::get<FastMap<char, Vector>>(params.relinearizeThreshold)) {
        chrs.push_back(cv.first);
        dims.push_back((int32_t)cv.second.size());
        values.insert(values.end(), cv.second.data(), cv.second.data() + cv.second.size());
      }
      check(lmgpu_isam2_set_relinearize_thresholds(h_, (int32_t)dims.size(), chrs.data(), dims.data(), values.data()));
    }
    if (params.enablePartialRelinearizationCheck) check(lmgpu_isam2_set_partial_relinearization_check(h_, 1));
    if (params.evaluateNonlinearError) check(lmgpu_isam2_set_evaluate_nonlinear_error(h_, 1));
    if (params.findUnusedFactorSlots) check(lmgpu_isam2_set_find_unused_factor_slots(h_, 1));
  }
  ~GpuISAM2() { if (h_) lmgpu_isam2_destroy(h_); }
  GpuISAM2(const GpuISAM2&) = delete;
  GpuISAM2& operator=(const GpuISAM2&) = delete;

  /// ISAM2::update(newFactors, newTheta, removeFactorIndices, constrainedKeys, noRelinKeys, extraReelimKeys, force_relinearize)
  /// (ISAM2.h:146-156, ISAM2.cpp:400-416)
  lmgpu_isam2_result update(const NonlinearFactorGraph& newFactors = NonlinearFactorGraph(), const Values& newTheta = Values(),
                            const FactorIndices& removeFactorIndices = FactorIndices(),
                            const std::optional<FastMap<Key, int>>& constrainedKeys = {}, const std::optional<FastList<Key>>& noRelinKeys = {},
                            const std::optional<FastList<Key>>& extraReelimKeys = {}, bool force_relinearize = false) {
    ISAM2UpdateParams params;
    params.constrainedKeys = constrainedKeys;
    params.extraReelimKeys = extraReelimKeys;
    params.force_relinearize = force_relinearize;
    params.noRelinKeys = noRelinKeys;
    params.removeFactorIndices = removeFactorIndices;
    return update(newFactors, newTheta, params);
  }

  /// ISAM2::update(newFactors, newTheta, const ISAM2UpdateParams&) (ISAM2.h:176-186, ISAM2.cpp:419-480).  The new factors take the
  /// indices ISAM2Result::newFactorsIndices would name: size() .. of the factor list, or its empty slots first with findUnusedFactorSlots.
  lmgpu_isam2_result update(const NonlinearFactorGraph& newFactors, const Values& newTheta, const ISAM2UpdateParams& up) {
    if (up.newAffectedKeys) throw std::invalid_argument("GpuISAM2: newAffectedKeys (smart factors) is not bound");
    std::vector<uint64_t> keys;
    std::vector<int32_t> types;
    std::vector<double> packed;
    for (const auto& kv : newTheta) {
      const int32_t t = lmgpu_detail::variableType(kv.value);
      keys.push_back(kv.key);
      types.push_back(t);
      double v[15];
      packOne(newTheta, kv.key, t, v);
      packed.insert(packed.end(), v, v + lmgpu_adapter::varStore(t));
      if (all_.exists(kv.key)) all_.erase(kv.key);  // a key that left the system (unusedKeys) may come back
      all_.insert(kv.key, kv.value);  // keeps what does not travel (Cal3Bundler's principal point) and the types for download
    }
    check(lmgpu_isam2_add_variables(h_, (int32_t)keys.size(), keys.data(), types.data(), packed.data()));
    for (size_t i = 0; i < newFactors.size(); i++) {
      if (!newFactors[i]) continue;
      auto nm = std::dynamic_pointer_cast<NoiseModelFactor>(newFactors[i]);
      int32_t type;
      std::vector<double> meas;
      if (!nm || !lmgpu_detail::extractFactor(newFactors[i], all_, &type, &meas)) throw std::invalid_argument("GpuISAM2: factor type not bound");
      const lmgpu_detail::Noise nz = lmgpu_detail::extractNoise(nm->noiseModel());
      std::vector<uint64_t> fk(nm->keys().begin(), nm->keys().end());
      check(lmgpu_isam2_add_factors_robust(h_, type, 1, fk.data(), meas.data(), nz.kind, nz.data.empty() ? nullptr : nz.data.data(), nz.robust, nz.robustK));
    }
    std::vector<uint64_t> rm(up.removeFactorIndices.begin(), up.removeFactorIndices.end()), ck, nr, ex;
    std::vector<int32_t> cg;
    if (up.constrainedKeys)
      for (const auto& kg : *up.constrainedKeys) {
        ck.push_back(kg.first);
        cg.push_back(kg.second);
      }
    if (up.noRelinKeys) nr.assign(up.noRelinKeys->begin(), up.noRelinKeys->end());
    if (up.extraReelimKeys) ex.assign(up.extraReelimKeys->begin(), up.extraReelimKeys->end());
    const lmgpu_isam2_update_params p{(int32_t)rm.size(), rm.data(), up.constrainedKeys ? 1 : 0, (int32_t)ck.size(), ck.data(), cg.data(),
                                      (int32_t)nr.size(), nr.data(), (int32_t)ex.size(), ex.data(), up.force_relinearize ? 1 : 0,
                                      up.forceFullSolve ? 1 : 0};
    lmgpu_isam2_result r{};
    check(lmgpu_isam2_update_with(h_, &p, &r));
    return r;
  }

  /// ISAM2::marginalizeLeaves(leafKeys, marginalFactorsIndices, deletedFactorsIndices) (ISAM2.h:198-222, ISAM2.cpp:487-720).  The leaf
  /// keys must have been ordered first (constrainedKeys of the update before, as IncrementalFixedLagSmoother does).  A key that is not a
  /// leaf throws std::runtime_error BEFORE anything changes (the reference checks it in debug builds only).
  void marginalizeLeaves(const FastList<Key>& leafKeys, FactorIndices* marginalFactorsIndices = nullptr, FactorIndices* deletedFactorsIndices = nullptr) {
    const std::vector<uint64_t> keys(leafKeys.begin(), leafKeys.end());
    int32_t nm = 0, nd = 0;
    check(lmgpu_isam2_marginalize_leaves(h_, (int32_t)keys.size(), keys.data(), &nm, &nd));
    std::vector<uint64_t> mi((size_t)std::max(1, nm)), di((size_t)std::max(1, nd));
    check(lmgpu_isam2_get_marginalize_result(h_, mi.data(), di.data()));
    if (marginalFactorsIndices) marginalFactorsIndices->assign(mi.begin(), mi.begin() + nm);
    if (deletedFactorsIndices) deletedFactorsIndices->assign(di.begin(), di.begin() + nd);
    for (Key k : leafKeys)
      if (all_.exists(k)) all_.erase(k);
  }
  /// ISAM2::getFixedVariables() (ISAM2.h:259)
  KeySet getFixedVariables() const {
    std::vector<uint64_t> k((size_t)std::max(1, lmgpu_isam2_get_fixed_variables(h_, nullptr)));
    const int n = lmgpu_isam2_get_fixed_variables(h_, k.data());
    return KeySet(k.begin(), k.begin() + n);
  }

  /// ISAM2::marginalCovariance(key) (ISAM2.h:253-257)
  Matrix marginalCovariance(Key key) const {
    if (!all_.exists(key)) throw std::out_of_range("GpuISAM2::marginalCovariance: unknown key");
    const int d = (int)all_.at(key).dim();
    Matrix cov(d, d);  // symmetric: the row-major block the library writes reads the same column-major
    check(lmgpu_isam2_marginal_covariance(h_, key, cov.data()));
    return cov;
  }

  /// ISAM2Result::errorBefore / errorAfter of the last update (with ISAM2Params::evaluateNonlinearError)
  std::pair<double, double> errors() const {
    double before = 0, after = 0;
    check(lmgpu_isam2_get_errors(h_, &before, &after));
    return {before, after};
  }
  /// getFactorsUnsafe().error(calculateEstimate())
  double error() const {
    double e = 0;
    check(lmgpu_isam2_error(h_, 0, &e));
    return e;
  }

  /// ISAM2Result::unusedKeys of the last update
  KeySet unusedKeys() const {
    std::vector<uint64_t> k((size_t)std::max(1, lmgpu_isam2_get_unused_keys(h_, nullptr)));
    const int n = lmgpu_isam2_get_unused_keys(h_, k.data());
    return KeySet(k.begin(), k.begin() + n);
  }

  Values calculateEstimate() const { return download(0); }
  /// ISAM2::calculateEstimate<VALUE>(Key) (ISAM2.h:231-236, ISAM2.cpp:757-760): only this variable is retracted and downloaded
  template <class VALUE>
  VALUE calculateEstimate(Key key) const {
    int32_t t = -1;
    double q[15];
    check(lmgpu_isam2_get_value(h_, 0, key, &t, q));
    Values one;
    unpackOne(key, t, q, &one);
    return one.at<VALUE>(key);
  }
  Values calculateBestEstimate() const { return download(1); }
  Values getLinearizationPoint() const { return download(2); }

 private:
  static int Ccolamd(void*, int32_t n_rows, int32_t n_cols, const int32_t* col_ptr, const int32_t* row_idx, const int32_t* cmember, int32_t* perm_out) {
    const size_t Alen = ccolamd_recommended(col_ptr[n_cols], n_rows, n_cols);
    std::vector<int> A(Alen), p(col_ptr, col_ptr + n_cols + 1), cm(cmember, cmember + n_cols);
    std::copy(row_idx, row_idx + col_ptr[n_cols], A.begin());
    double knobs[CCOLAMD_KNOBS];
    ccolamd_set_defaults(knobs);
    knobs[CCOLAMD_DENSE_ROW] = -1;
    knobs[CCOLAMD_DENSE_COL] = -1;
    int stats[CCOLAMD_STATS];
    if (ccolamd(n_rows, n_cols, (int)Alen, A.data(), p.data(), knobs, stats, cm.data()) != 1) return 0;
    std::copy(p.begin(), p.begin() + n_cols, perm_out);
    return 1;
  }
  void check(int rc) const {
    if (rc == LMGPU_OK) return;
    if (rc == LMGPU_INDETERMINATE) throw IndeterminantLinearSystemException(lmgpu_isam2_last_failed_key(h_));
    throw std::runtime_error(lmgpu_isam2_last_error(h_));
  }
  static void packOne(const Values& v, Key k, int32_t t, double* o) {
    switch (t) {
      case LMGPU_POSE2: { const Pose2& q = v.at<Pose2>(k); o[0] = q.x(); o[1] = q.y(); o[2] = q.theta(); break; }
      case LMGPU_POSE3: lmgpu_detail::packPose3(v.at<Pose3>(k), o); break;
      case LMGPU_POINT3: { const Point3& q = v.at<Point3>(k); o[0] = q.x(); o[1] = q.y(); o[2] = q.z(); break; }
      case LMGPU_POINT2: { const Point2& q = v.at<Point2>(k); o[0] = q.x(); o[1] = q.y(); break; }
        case LMGPU_CAL3_S2: { const Cal3_S2& q = v.at<Cal3_S2>(k); o[0] = q.fx(); o[1] = q.fy(); o[2] = q.skew(); o[3] = q.px(); o[4] = q.py(); break; }
      default: {
        const lmgpu_detail::Camera& c = v.at<lmgpu_detail::Camera>(k);
        lmgpu_detail::packPose3(c.pose(), o);
        o[12] = c.calibration().fx(); o[13] = c.calibration().k1(); o[14] = c.calibration().k2();
      }
    }
  }
  void unpackOne(Key k, int32_t t, const double* q, Values* out) const {
    switch (t) {
      case LMGPU_POSE2: out->insert(k, Pose2(q[0], q[1], q[2])); break;
      case LMGPU_POSE3: out->insert(k, lmgpu_detail::unpackPose3(q)); break;
      case LMGPU_POINT3: out->insert(k, Point3(q[0], q[1], q[2])); break;
      case LMGPU_POINT2: out->insert(k, Point2(q[0], q[1])); break;
      case LMGPU_CAL3_S2: out->insert(k, Cal3_S2(q[0], q[1], q[2], q[3], q[4])); break;
      default: {
        const Cal3Bundler& K0 = all_.at<lmgpu_detail::Camera>(k).calibration();
        out->insert(k, lmgpu_detail::Camera(lmgpu_detail::unpackPose3(q), Cal3Bundler(q[12], q[13], q[14], K0.px(), K0.py())));
      }
    }
  }
  Values download(int which) const {
    const int n = lmgpu_isam2_num_variables(h_);
    std::vector<uint64_t> keys((size_t)n);
    std::vector<int32_t> types((size_t)n);
    check(lmgpu_isam2_get_values(h_, 2, keys.data(), types.data(), nullptr));
    size_t tot = 0;
    for (int32_t t : types) tot += (size_t)lmgpu_adapter::varStore(t);
    std::vector<double> packed(tot);
    check(lmgpu_isam2_get_values(h_, which, nullptr, nullptr, packed.data()));
    Values out;
    const double* q = packed.data();
    for (int i = 0; i < n; i++) {
      const Key k = keys[(size_t)i];
      unpackOne(k, types[(size_t)i], q, &out);
      q += lmgpu_adapter::varStore(types[(size_t)i]);
    }
    return out;
  }

  lmgpu_isam2* h_ = nullptr;
  Values all_;  // every variable ever added, at its initial value
};

}  // namespace gtsam
