// lmgpu_adapter_core.hpp — the GTSAM-independent half of the reference-side binding of lmgpu.h.
//
// include/lmgpu_gtsam_adapter.h (the header a GTSAM maintainer adds) only EXTRACTS numbers from GTSAM objects (dynamic casts,
// accessors) and hands them to the classes below; everything that touches the C ABI — slot assignment in elimination order,
// bucketing by (factor type, noise kind, m-estimator), packing, call order, status -> exception mapping — lives here, has no
// GTSAM include and is therefore compiled and tested in this repository (tests/cpp/adapter_harness.cpp drives SFMExample_bal's
// graph through it; GTSAM itself cannot be built here).
//
// Call order (= what GpuLevenbergMarquardtOptimizer's constructor does, gtsam/nonlinear/LevenbergMarquardtOptimizer.cpp:40-62):
//   Problem p(device);
//   p.setVariables(keys in elimination order, types);        // after EnsureHasOrdering, LevenbergMarquardtParams.h:112-117
//   p.addFactor(...) for every factor of the NonlinearFactorGraph, in graph order;
//   p.finalize();                                             // buckets -> lmgpu_add_factor_bucket_robust, lmgpu_finalize_structure
//   p.setValues(packed);  p.lmInit(params);                   // State(initialValues, graph.error(initialValues), lambdaInitial, ...)
//   p.iterate(params) ...                                     // LevenbergMarquardtOptimizer::iterate
#pragma once

#include <cmath>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "lmgpu.h"

namespace lmgpu_adapter {

// ---- shape tables of include/lmgpu.h (variable / factor / noise enums)
inline int varDim(int32_t t) { static const int d[LMGPU_NUM_VAR_TYPES] = {3, 6, 3, 9, 2, 5}; return d[t]; }
inline int varStore(int32_t t) { static const int s[LMGPU_NUM_VAR_TYPES] = {3, 12, 3, 15, 2, 5}; return s[t]; }
inline int factorArity(int32_t f) { static const int a[LMGPU_NUM_FACTOR_TYPES] = {2, 2, 2, 1, 1, 1, 1, 2, 2, 2, 3, 1}; return a[f]; }
inline int factorRows(int32_t f) { static const int r[LMGPU_NUM_FACTOR_TYPES] = {2, 3, 6, 3, 6, 3, 9, 2, 2, 2, 2, 5}; return r[f]; }
inline int factorMeas(int32_t f) { static const int m[LMGPU_NUM_FACTOR_TYPES] = {2, 3, 12, 3, 12, 3, 15, 7, 19, 2, 2, 5}; return m[f]; }
inline int32_t factorVarType(int32_t f, int i) {
  static const int32_t v[LMGPU_NUM_FACTOR_TYPES][3] = {{LMGPU_CAM_BUNDLER, LMGPU_POINT3, -1}, {LMGPU_POSE2, LMGPU_POSE2, -1}, {LMGPU_POSE3, LMGPU_POSE3, -1},
                                                       {LMGPU_POSE2, -1, -1}, {LMGPU_POSE3, -1, -1}, {LMGPU_POINT3, -1, -1}, {LMGPU_CAM_BUNDLER, -1, -1},
                                                       {LMGPU_POSE3, LMGPU_POINT3, -1}, {LMGPU_POSE3, LMGPU_POINT3, -1}, {LMGPU_POSE2, LMGPU_POINT2, -1},
                                                       {LMGPU_POSE3, LMGPU_POINT3, LMGPU_CAL3_S2}, {LMGPU_CAL3_S2, -1, -1}};
  return v[f][i];
}
inline int noiseDoubles(int32_t kind, int rows) { return kind == LMGPU_N_UNIT ? 0 : (kind == LMGPU_N_DIAG ? rows : rows * rows); }

/// status 1: what the reference throws as IndeterminantLinearSystemException(key) (gtsam/linear/linearExceptions.h;
/// raised at HessianFactor.cpp:475-482 with keys.front() of the failing clique)
struct Indeterminate : std::runtime_error {
  int slot;
  uint64_t key;
  Indeterminate(int s, uint64_t k) : std::runtime_error("lmgpu: indeterminate linear system"), slot(s), key(k) {}
};
struct Error : std::runtime_error {
  int status;
  Error(int st, const std::string& what) : std::runtime_error("lmgpu: " + what), status(st) {}
};

struct SolveResult {
  std::vector<double> delta;  // packed by slot (elimination order), lmgpu_total_dim() doubles
  double linearError0 = 0, linearError = 0;
};

class Problem {
 public:
  /// device < 0: structure-only handle (symbolic analysis; every compute call throws Error(LMGPU_HIP_ERROR))
  explicit Problem(int device = 0, int rank = 0, int worldSize = 1) {
    lmgpu_config cfg{device, rank, worldSize, 0};
    if (lmgpu_create(&cfg, &h_) != LMGPU_OK) {
      const std::string why = h_ ? lmgpu_last_error(h_) : "lmgpu_create failed";
      if (h_) lmgpu_destroy(h_);
      h_ = nullptr;
      throw Error(LMGPU_HIP_ERROR, why);
    }
  }
  ~Problem() { if (h_) lmgpu_destroy(h_); }
  Problem(const Problem&) = delete;
  Problem& operator=(const Problem&) = delete;

  lmgpu_handle* handle() const { return h_; }

  /// variables in ELIMINATION order (the Ordering of NonlinearOptimizerParams::ordering); position = slot
  void setVariables(const std::vector<uint64_t>& keys, const std::vector<int32_t>& types) {
    if (keys.size() != types.size()) throw Error(LMGPU_INVALID, "keys / types size mismatch");
    keys_ = keys;
    types_ = types;
    slot_.clear();
    slot_.reserve(keys.size() * 2);
    xoff_.assign(1, 0);
    voff_.assign(1, 0);
    for (size_t i = 0; i < keys.size(); i++) {
      if (types[i] < 0 || types[i] >= LMGPU_NUM_VAR_TYPES) throw Error(LMGPU_INVALID, "unknown variable type");
      if (!slot_.emplace(keys[i], (int32_t)i).second) throw Error(LMGPU_INVALID, "duplicate key in the ordering");
      xoff_.push_back(xoff_.back() + varDim(types[i]));
      voff_.push_back(voff_.back() + varStore(types[i]));
    }
    check(lmgpu_set_variables(h_, (int32_t)keys.size(), keys_.data(), types_.data()));
  }
  size_t numVariables() const { return keys_.size(); }
  uint64_t keyOfSlot(int s) const { return keys_.at((size_t)s); }
  int32_t typeOfSlot(int s) const { return types_.at((size_t)s); }
  int32_t slotOf(uint64_t key) const {
    auto it = slot_.find(key);
    if (it == slot_.end()) throw Error(LMGPU_INVALID, "a factor references a key that is not in the ordering");
    return it->second;
  }
  /// offsets of slot s in a packed tangent vector (delta, hessian diagonal) / in the packed values
  size_t deltaOffset(int s) const { return xoff_.at((size_t)s); }
  size_t valueOffset(int s) const { return voff_.at((size_t)s); }

  /// one factor of the NonlinearFactorGraph; graphIndex = its position there (VariableIndex order and error-sum order).
  /// meas: factorMeas(type) doubles as documented in lmgpu.h; noise: noiseDoubles(kind, rows) doubles (nullptr for UNIT).
  void addFactor(int32_t type, int32_t graphIndex, const uint64_t* keys, const double* meas, int32_t noiseKind, const double* noise,
                 int32_t robustKind = LMGPU_ROBUST_NONE, double robustK = 0.0) {
    if (type < 0 || type >= LMGPU_NUM_FACTOR_TYPES) throw Error(LMGPU_INVALID, "unknown factor type");
    Bucket& b = buckets_[std::make_tuple(type, noiseKind, robustKind, robustK)];
    b.gi.push_back(graphIndex);
    for (int i = 0; i < factorArity(type); i++) {
      const int32_t s = slotOf(keys[i]);
      if (types_[(size_t)s] != factorVarType(type, i)) throw Error(LMGPU_INVALID, "variable type does not fit the factor");
      b.slots.push_back(s);
    }
    b.meas.insert(b.meas.end(), meas, meas + factorMeas(type));
    const int nn = noiseDoubles(noiseKind, factorRows(type));
    if (nn) b.noise.insert(b.noise.end(), noise, noise + nn);
  }

  void finalize() {
    for (auto& kv : buckets_) {
      const Bucket& b = kv.second;
      check(lmgpu_add_factor_bucket_robust(h_, std::get<0>(kv.first), (int32_t)b.gi.size(), b.gi.data(), b.slots.data(), b.meas.data(),
                                           std::get<1>(kv.first), b.noise.empty() ? nullptr : b.noise.data(), std::get<2>(kv.first),
                                           std::get<3>(kv.first)));
    }
    buckets_.clear();
    check(lmgpu_finalize_structure(h_));
  }

  int totalDim() const { return lmgpu_total_dim(h_); }
  int totalStore() const { return lmgpu_total_store(h_); }
  int numFronts() const { return lmgpu_num_fronts(h_); }

  // ---- values
  void setValues(const std::vector<double>& packed) {
    if ((int)packed.size() != totalStore()) throw Error(LMGPU_INVALID, "packed values have the wrong length");
    check(lmgpu_set_values(h_, packed.data()));
  }
  std::vector<double> getValues() const {
    std::vector<double> v((size_t)totalStore());
    check(lmgpu_get_values(h_, v.data()));
    return v;
  }

  // ---- the hot path, piecewise (LevenbergMarquardtOptimizer::linearize :113, NonlinearOptimizer::solve :129)
  double error() const {
    double e = 0;
    check(lmgpu_error(h_, &e));
    return e;
  }
  void linearize() const { check(lmgpu_linearize(h_)); }
  SolveResult solve(double lambda, bool diagonalDamping, double minDiagonal, double maxDiagonal) const {
    SolveResult r;
    r.delta.resize((size_t)totalDim());
    check(lmgpu_solve(h_, lambda, diagonalDamping ? 1 : 0, minDiagonal, maxDiagonal, r.delta.data(), &r.linearError0, &r.linearError));
    return r;
  }
  void retract(const std::vector<double>* delta = nullptr) const { check(lmgpu_retract(h_, delta ? delta->data() : nullptr)); }
  std::vector<double> hessianDiagonal() const {
    std::vector<double> d((size_t)totalDim());
    check(lmgpu_hessian_diagonal(h_, d.data()));
    return d;
  }
  /// whitened [A1 .. Ak b] of one factor, column-major rows x cols (JacobianFactor's VerticalBlockMatrix)
  std::vector<double> jacobian(int32_t graphIndex, int32_t* rows, int32_t* cols) const {
    check(lmgpu_get_jacobian(h_, graphIndex, nullptr, rows, cols));
    std::vector<double> out((size_t)*rows * (size_t)*cols);
    check(lmgpu_get_jacobian(h_, graphIndex, out.data(), rows, cols));
    return out;
  }

  /// the whole linearization (what iterate() returns in the reference): one copy per factor bucket instead of one per factor
  struct LinearGraph {
    std::vector<int32_t> graphIndex, rows, cols;  // ascending graph index
    std::vector<int64_t> offsets;                 // n + 1
    std::vector<double> data;                     // [A1 .. Ak b] column-major, back to back
  };
  LinearGraph jacobians() const {
    LinearGraph g;
    int32_t n = 0;
    check(lmgpu_get_jacobians(h_, &n, nullptr, nullptr, nullptr, nullptr, nullptr));
    g.graphIndex.resize((size_t)n);
    g.rows.resize((size_t)n);
    g.cols.resize((size_t)n);
    g.offsets.resize((size_t)n + 1);
    check(lmgpu_get_jacobians(h_, &n, g.graphIndex.data(), g.rows.data(), g.cols.data(), g.offsets.data(), nullptr));
    g.data.resize((size_t)g.offsets.back());
    check(lmgpu_get_jacobians(h_, &n, nullptr, nullptr, nullptr, nullptr, g.data.data()));
    return g;
  }

  // ---- the hot path, whole
  lmgpu_lm_state lmInit(const lmgpu_lm_params& p) {
    lmgpu_lm_state st{};
    check(lmgpu_lm_init(h_, &p, &st));
    return st;
  }
  void iterate(const lmgpu_lm_params& p, lmgpu_lm_state* inout) { check(lmgpu_iterate(h_, &p, inout)); }
  void optimize(const lmgpu_lm_params& p, lmgpu_lm_state* inout) { check(lmgpu_optimize(h_, &p, inout)); }
  void gnIterate(lmgpu_lm_state* inout) { check(lmgpu_gn_iterate(h_, inout)); }
  void dlIterate(lmgpu_lm_state* inout) { check(lmgpu_dl_iterate(h_, inout)); }

  /// status -> exception, the mapping the reference-side adapter relies on
  void check(int rc) const {
    if (rc == LMGPU_OK) return;
    if (rc == LMGPU_INDETERMINATE) {
      const int s = lmgpu_last_failed_slot(h_);
      throw Indeterminate(s, (s >= 0 && (size_t)s < keys_.size()) ? keys_[(size_t)s] : 0);
    }
    const char* m = lmgpu_last_error(h_);
    throw Error(rc, m ? m : "");
  }

 private:
  struct Bucket {
    std::vector<int32_t> gi, slots;
    std::vector<double> meas, noise;
  };
  lmgpu_handle* h_ = nullptr;
  std::vector<uint64_t> keys_;
  std::vector<int32_t> types_;
  std::unordered_map<uint64_t, int32_t> slot_;
  std::vector<size_t> xoff_, voff_;
  std::map<std::tuple<int32_t, int32_t, int32_t, double>, Bucket> buckets_;
};

/// NonlinearOptimizer::defaultOptimize (gtsam/nonlinear/NonlinearOptimizer.cpp:62-117) around an iterate() callable: what the
/// reference's optimize() does with the adapter's iterate() override.  iterate(state) advances state by one outer iteration.
template <class Iterate>
inline void defaultOptimize(const lmgpu_lm_params& p, lmgpu_lm_state* st, Iterate iterate) {
  double currentError = st->error;
  if (st->iterations >= p.maxIterations) return;
  if (currentError <= p.errorTol) return;
  double newError = currentError;
  do {
    currentError = newError;
    iterate(st);
    newError = st->error;
    if (newError <= p.errorTol) break;
    const double absoluteDecrease = currentError - newError, relativeDecrease = absoluteDecrease / currentError;
    if ((p.relativeErrorTol && relativeDecrease <= p.relativeErrorTol) || absoluteDecrease <= p.absoluteErrorTol) break;
  } while (st->iterations < p.maxIterations && std::isfinite(currentError));
}

}  // namespace lmgpu_adapter
