// adapter_harness — drives liblmgpu.so through include/lmgpu_adapter_core.hpp in EXACTLY the call order of the reference-side
// adapter (include/lmgpu_gtsam_adapter.h: GpuLevenbergMarquardtOptimizer's constructor, iterate(), linearize(), solve()), from
// C++, without GTSAM (which cannot be built in this repository).  The problem comes in a neutral text file written by the
// test (tests/test_adapter_harness.py) from the same graph the Python mirror holds:
//
//   VARIABLES n            then n lines:  key type store d0 d1 ...        (elimination order; `store` packed doubles, lmgpu.h)
//   FACTORS m              then m lines:  type graph_index k0 [k1] | nmeas meas... | noise_kind nnoise noise... | robust_kind robust_k
//   PARAMS                 13 numbers of lmgpu_lm_params in declaration order
//
// usage: adapter_harness <problem file> <device> <mode>      device -1 = structure only
//   mode structure : fronts (Scatter key order, frontal counts, parents) as JSON
//   mode optimize  : NonlinearOptimizer::defaultOptimize around iterate() (= what lm.optimize() does with the override)
//   mode piecewise : the reference's own iterate()/tryLambda (LevenbergMarquardtOptimizer.cpp:121-308) restated on the host with
//                    ONLY linearize() / solve() / error / retract forwarded — the Piecewise mode of the adapter
// Output: one JSON object on stdout.  Exit code 0 ok, 1 lmgpu error (message in the JSON), 2 usage.
#include <cinttypes>
#include <cstdarg>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/lmgpu_adapter_core.hpp"

using namespace lmgpu_adapter;

struct Input {
  std::vector<uint64_t> keys;
  std::vector<int32_t> types;
  std::vector<double> values;
  struct F { int32_t type, gi; uint64_t k[2]; std::vector<double> meas, noise; int32_t noiseKind, robust; double robustK; };
  std::vector<F> factors;
  lmgpu_lm_params params{};
};

static bool expect(std::istream& is, const char* word) {
  std::string w;
  return (is >> w) && w == word;
}

static bool readInput(const char* path, Input* in) {
  std::ifstream is(path);
  if (!is) return false;
  size_t n = 0, m = 0;
  if (!expect(is, "VARIABLES") || !(is >> n)) return false;
  for (size_t i = 0; i < n; i++) {
    uint64_t k; int32_t t; int store;
    if (!(is >> k >> t >> store) || store != varStore(t)) return false;
    in->keys.push_back(k);
    in->types.push_back(t);
    for (int j = 0; j < store; j++) { double d; is >> d; in->values.push_back(d); }
  }
  if (!expect(is, "FACTORS") || !(is >> m)) return false;
  for (size_t i = 0; i < m; i++) {
    Input::F f{};
    int nmeas = 0, nnoise = 0;
    if (!(is >> f.type >> f.gi)) return false;
    for (int j = 0; j < factorArity(f.type); j++) is >> f.k[j];
    is >> nmeas;
    if (nmeas != factorMeas(f.type)) return false;
    f.meas.resize((size_t)nmeas);
    for (double& d : f.meas) is >> d;
    is >> f.noiseKind >> nnoise;
    if (nnoise != noiseDoubles(f.noiseKind, factorRows(f.type))) return false;
    f.noise.resize((size_t)nnoise);
    for (double& d : f.noise) is >> d;
    is >> f.robust >> f.robustK;
    in->factors.push_back(f);
  }
  if (!expect(is, "PARAMS")) return false;
  lmgpu_lm_params& p = in->params;
  is >> p.maxIterations >> p.relativeErrorTol >> p.absoluteErrorTol >> p.errorTol >> p.lambdaInitial >> p.lambdaFactor >> p.lambdaUpperBound >>
      p.lambdaLowerBound >> p.minModelFidelity >> p.diagonalDamping >> p.useFixedLambdaFactor >> p.minDiagonal >> p.maxDiagonal;
  return (bool)is;
}

static std::string g_out;  // the JSON object is printed whole at the end (or replaced by {"exception": ...})
static void outf(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_out += buf;
}
static void printVector(const char* name, const std::vector<double>& v) {
  outf("\"%s\": [", name);
  for (size_t i = 0; i < v.size(); i++) outf("%s%.17g", i ? ", " : "", v[i]);
  outf("]");
}

// LevenbergMarquardtOptimizer::tryLambda (LevenbergMarquardtOptimizer.cpp:121-270) with the adapter's Piecewise bindings:
// solve() -> Problem::solve (which also returns linear.error(0) / linear.error(delta)), retract + graph error on the device.
static bool tryLambdaPiecewise(Problem& p, const lmgpu_lm_params& q, lmgpu_lm_state* st, const std::vector<double>& valuesAtLinearization) {
  bool systemSolvedSuccessfully = false;
  SolveResult r;
  try {
    r = p.solve(st->lambda, q.diagonalDamping != 0, q.minDiagonal, q.maxDiagonal);
    systemSolvedSuccessfully = true;
  } catch (const Indeterminate&) {
  }
  double modelFidelity = 0.0, newError = std::numeric_limits<double>::infinity(), costChange = 0.0;
  bool step_is_successful = false, stopSearchingLambda = false;
  if (systemSolvedSuccessfully) {
    const double oldLinearizedError = r.linearError0, newlinearizedError = r.linearError;
    const double linearizedCostChange = oldLinearizedError - newlinearizedError;
    if (linearizedCostChange >= 0) {
      p.retract(&r.delta);  // newValues = values.retract(delta)
      newError = p.error();
      costChange = st->error - newError;
      if (linearizedCostChange > std::numeric_limits<double>::epsilon() * oldLinearizedError) {
        modelFidelity = costChange / linearizedCostChange;
        step_is_successful = modelFidelity > q.minModelFidelity;
      }
      const double minAbsoluteTolerance = q.relativeErrorTol * st->error;
      if (std::abs(costChange) < minAbsoluteTolerance) stopSearchingLambda = true;
      if (!step_is_successful) p.setValues(valuesAtLinearization);  // the reference simply drops newValues
    }
  }
  if (step_is_successful) {  // decreaseLambda, LevenbergMarquardtState.h:80-93
    double newLambda = st->lambda, newFactor = st->currentFactor;
    if (q.useFixedLambdaFactor) {
      newLambda /= st->currentFactor;
    } else {
      newLambda *= std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * modelFidelity - 1.0, 3));
      newFactor = 2.0 * st->currentFactor;
    }
    st->lambda = std::max(q.lambdaLowerBound, newLambda);
    st->currentFactor = newFactor;
    st->error = newError;
    st->iterations += 1;
    st->totalNumberInnerIterations += 1;
    return true;
  } else if (!stopSearchingLambda) {  // increaseLambda :70-76
    st->lambda *= st->currentFactor;
    st->totalNumberInnerIterations += 1;
    if (!q.useFixedLambdaFactor) st->currentFactor *= 2.0;
    return st->lambda >= q.lambdaUpperBound;
  }
  return true;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <problem file> <device> <structure|optimize|piecewise>\n", argv[0]);
    return 2;
  }
  Input in;
  if (!readInput(argv[1], &in)) {
    std::fprintf(stderr, "cannot parse %s\n", argv[1]);
    return 2;
  }
  const int device = std::atoi(argv[2]);
  const std::string mode = argv[3];
  try {
    // ---- the adapter's constructor, step by step
    Problem p(device);
    p.setVariables(in.keys, in.types);
    for (const Input::F& f : in.factors)
      p.addFactor(f.type, f.gi, f.k, f.meas.data(), f.noiseKind, f.noise.empty() ? nullptr : f.noise.data(), f.robust, f.robustK);
    p.finalize();
    outf("{\"total_dim\": %d, \"total_store\": %d, \"num_fronts\": %d", p.totalDim(), p.totalStore(), p.numFronts());
    if (mode == "structure") {
      outf(", \"fronts\": [");
      for (int i = 0; i < p.numFronts(); i++) {
        int32_t info[8];
        p.check(lmgpu_front_info(p.handle(), i, info));
        std::vector<int32_t> slots((size_t)info[0]);
        p.check(lmgpu_get_front(p.handle(), i, slots.data(), nullptr));
        outf("%s{\"nfk\": %d, \"nf\": %d, \"n\": %d, \"parent\": %d, \"keys\": [", i ? ", " : "", info[1], info[2], info[3], info[4]);
        for (size_t j = 0; j < slots.size(); j++) outf("%s%" PRIu64, j ? ", " : "", p.keyOfSlot(slots[j]));
        outf("]}");
      }
      outf("]}\n");
      std::fputs(g_out.c_str(), stdout);
      return 0;
    }
    p.setValues(in.values);
    lmgpu_lm_state st = p.lmInit(in.params);
    outf(", \"error_initial\": %.17g", st.error);
    if (mode == "optimize") {
      defaultOptimize(in.params, &st, [&](lmgpu_lm_state* s) { p.iterate(in.params, s); });
    } else if (mode == "piecewise") {
      defaultOptimize(in.params, &st, [&](lmgpu_lm_state* s) {
        // LevenbergMarquardtOptimizer::iterate :273-308: linearize once, then tryLambda until a step is taken or LM gives up
        const std::vector<double> x0 = p.getValues();
        p.linearize();
        while (!tryLambdaPiecewise(p, in.params, s, x0)) {
        }
      });
    } else {
      std::fprintf(stderr, "unknown mode %s\n", mode.c_str());
      return 2;
    }
    {  // what iterate() returns through the adapter (Problem::jacobians, one copy per bucket) against the per-factor tap
      const Problem::LinearGraph lg = p.jacobians();
      bool same = lg.graphIndex.size() == in.factors.size();
      for (size_t i = 0; same && i < lg.graphIndex.size(); i++) {
        int32_t rows = 0, cols = 0;
        const std::vector<double> one = p.jacobian(lg.graphIndex[i], &rows, &cols);
        same = rows == lg.rows[i] && cols == lg.cols[i] && lg.offsets[i + 1] - lg.offsets[i] == (int64_t)one.size() &&
               std::equal(one.begin(), one.end(), lg.data.begin() + lg.offsets[i]);
      }
      outf(", \"linear_graph_factors\": %zu, \"linear_graph_matches_taps\": %s", lg.graphIndex.size(), same ? "true" : "false");
    }
    outf(", \"error\": %.17g, \"lambda\": %.17g, \"iterations\": %d, \"inner\": %d, ", st.error, st.lambda, st.iterations,
                st.totalNumberInnerIterations);
    printVector("values", p.getValues());
    outf("}\n");
    std::fputs(g_out.c_str(), stdout);
    return 0;
  } catch (const std::exception& e) {
    std::printf("{\"exception\": \"%s\"}\n", e.what());
    return 1;
  }
}
