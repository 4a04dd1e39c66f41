// ORACLE — TEST INFRASTRUCTURE ONLY (included by lm_oracle.cpp; see its header).
//
// CPU restatement of the reference's incremental smoother for the path of BASELINE config 5 (ISAM2 with Gauss-Newton
// optimisation, Cholesky factorisation, COLAMD ordering): pointer-based and literal like the reference.
//   ISAM2::update                      gtsam/nonlinear/ISAM2.cpp:404-480      (UpdateImpl: gtsam/nonlinear/ISAM2-impl.h:113-508)
//   ISAM2::recalculate                 gtsam/nonlinear/ISAM2.cpp:117-175      (batch :178-247, incremental :250-362)
//   ISAM2::relinearizeAffectedFactors  gtsam/nonlinear/ISAM2.cpp:66-114
//   BayesTree::removeTop / removePath / removeClique   gtsam/inference/BayesTree-inst.h:440-508
//   orphan subtrees as symbolic factors  gtsam/inference/BayesTree.h:283-304, ClusterTree-inst.h:228-236, Scatter.cpp:53-55
//   Ordering::ColamdConstrained        gtsam/inference/Ordering.cpp:50-125, 193-210 (ccolamd itself through a callback: the
//                                       test harness binds it to oracle/_ref/libccolamd_ref.so = the reference's own C source)
//   ISAM2::updateDelta / calculateEstimate  gtsam/nonlinear/ISAM2.cpp:701-781; wildfire back-substitution
//                                       gtsam/nonlinear/ISAM2Clique.cpp:56-77, 151-268, ISAM2-impl.cpp:34-77
//   ISAM2UpdateParams (gtsam/nonlinear/ISAM2UpdateParams.h:30-90): removeFactorIndices (pushBackFactors ISAM2-impl.h:141-173,
//                                       computeUnusedKeys :175-190, ISAM2::removeVariables ISAM2.cpp:385-398), constrainedKeys,
//                                       noRelinKeys, extraReelimKeys, force_relinearize, forceFullSolve
//   ISAM2Params: relinearizeThreshold as double or FastMap<char, Vector>, enablePartialRelinearizationCheck (ISAM2-impl.h:246-378)
//   ISAM2DoglegParams: ISAM2::updateDelta's Dogleg branch (ISAM2.cpp:739-779), DoglegOptimizerImpl::Iterate (DoglegOptimizerImpl.h:139-254)
//   ISAM2::marginalizeLeaves           gtsam/nonlinear/ISAM2.cpp:487-720 (BayesTree::removeSubtree gtsam/inference/BayesTree-inst.h:512-547;
//                                       LinearContainerFactor without a linearization point, gtsam/nonlinear/LinearContainerFactor.cpp:77-109;
//                                       fixedVariables_ ISAM2-impl.h:385-388; ISAM2Params::findUnusedFactorSlots, FactorGraph-inst.h:109-137)
// Not restated (not reached by the configs): QR, newAffectedKeys (smart factors).
#pragma once

namespace orc {

// int cb(n_rows, n_cols, col_ptr[n_cols + 1], row_idx[nnz], cmember[n_cols], perm_out[n_cols]): 1 on success.  The callee calls
// ccolamd with GTSAM's knobs (dense row / column detection off, Ordering.cpp:94-97).
typedef int (*ccolamd_fn)(int, int, const int*, const int*, const int*, int*);

struct IClique {
  std::vector<Key> keys;  // frontals then separator (Scatter order)
  std::vector<int> dims;
  int nFrontal = 0;
  Mat RSd;         // conditional_
  GFactor cached;  // cachedFactor_: the separator Hessian this clique passed to its parent (ISAM2Clique.cpp:35-46)
  std::vector<std::shared_ptr<IClique>> children;
  std::weak_ptr<IClique> parent;
};
typedef std::shared_ptr<IClique> ICliquePtr;

struct ISAM2Result {
  int variablesRelinearized = 0, variablesReeliminated = 0, factorsRecalculated = 0, cliques = 0, batch = 0;
};

// one entry of the linear graph handed to the elimination: a (cached) linear factor, or an orphan subtree standing in as a
// symbolic factor on its separator keys (BayesTreeOrphanWrapper)
struct IFactor {
  const GFactor* g = nullptr;
  ICliquePtr orphan;
  std::vector<Key> keys;
};

struct ISAM2 {
  // ISAM2Params (gtsam/nonlinear/ISAM2Params.h:133-246): GaussNewton(wildfireThreshold), relinearizeThreshold (double),
  // relinearizeSkip, enableRelinearization; cacheLinearizedFactors = true
  double wildfireThreshold = 0.001, relinearizeThreshold = 0.1;
  std::map<unsigned char, std::vector<double>> relinearizeThresholds;  // the FastMap<char, Vector> alternative (non-empty: in force)
  int relinearizeSkip = 10;
  bool enableRelinearization = true, enablePartialRelinearizationCheck = false;
  // ISAM2DoglegParams (ISAM2Params.h:68-110) instead of ISAM2GaussNewtonParams: initialDelta -> doglegDelta (ISAM2.cpp:41-46),
  // wildfireThreshold, adaptationMode (DoglegOptimizerImpl.h:54-58: 0 SEARCH_EACH_ITERATION, 1 SEARCH_REDUCE_ONLY, 2 ONE_STEP_PER_ITERATION)
  bool dogleg = false;
  double doglegDelta = 1.0, doglegWildfireThreshold = 1e-5;
  int doglegAdaptationMode = 0;
  VectorValues deltaNewton, RgProd;
  bool evaluateNonlinearError = false;  // ISAM2Params.h:200-203: ISAM2Result::errorBefore / errorAfter
  double errorBefore = 0, errorAfter = 0;
  ccolamd_fn ccolamd = nullptr;

  Values theta;
  VariableIndex variableIndex;
  VectorValues delta;
  std::set<Key> deltaReplacedMask;
  std::vector<Factor> nonlinearFactors;
  // a slot may hold a LinearContainerFactor (what marginalizeLeaves adds: a constant Gaussian factor on any number of keys, no linearization
  // point): isContainer[i], its factor is linearFactors[i] for good (never relinearized, error 0: LinearContainerFactor.cpp:77-79, 104-108)
  std::vector<char> isContainer;
  std::set<Key> fixedVariables;       // keys of marginal factors: never relinearized (ISAM2.cpp:693, ISAM2-impl.h:385-388)
  bool findUnusedFactorSlots = false;  // ISAM2Params.h:225
  std::vector<char> removedFactor;  // a removed factor leaves its slot behind (NonlinearFactorGraph::remove resets the pointer): indices stay
  std::vector<GFactor> linearFactors;
  std::vector<ICliquePtr> roots;
  std::map<Key, ICliquePtr> nodes;
  int update_count = 0;
  // pending input of the next update()
  std::vector<Factor> newFactors;
  Values newTheta;
  ISAM2Result last;
  std::vector<Key> lastUnusedKeys;
};

// gtsam/nonlinear/ISAM2UpdateParams.h:30-90 (without newAffectedKeys)
struct ISAM2UpdateParams {
  std::vector<size_t> removeFactorIndices;
  bool hasConstrainedKeys = false;  // std::optional: an empty map that is GIVEN still replaces the default grouping
  std::map<Key, int> constrainedKeys;
  std::vector<Key> noRelinKeys, extraReelimKeys;
  bool force_relinearize = false, forceFullSolve = false;
};

// Ordering::ColamdConstrained(variableIndex, groups) gtsam/inference/Ordering.cpp:193-210 -> :50-125
static std::vector<Key> colamd_constrained(const ISAM2& S, const VariableIndex& vi, size_t nFactors, const std::map<Key, int>& groups) {
  const size_t nVars = vi.size();
  if (nVars == 0) return {};
  if (nVars == 1) return {vi.begin()->first};
  std::vector<int> cmember(nVars, 0), p(nVars + 1, 0), A;
  std::vector<Key> keys(nVars);
  std::map<Key, size_t> keyIndices;
  size_t index = 0;
  for (auto& kf : vi) {
    for (size_t f : kf.second) A.push_back((int)f);
    p[index + 1] = (int)A.size();
    keys[index] = kf.first;
    keyIndices[kf.first] = index;
    ++index;
  }
  for (auto& g : groups) cmember[keyIndices.at(g.first)] = g.second;
  std::vector<int> perm(nVars);
  if (!S.ccolamd || S.ccolamd((int)nFactors, (int)nVars, p.data(), A.data(), cmember.data(), perm.data()) != 1)
    throw std::runtime_error("ccolamd failed");
  std::vector<Key> result(nVars);
  for (size_t j = 0; j < nVars; ++j) result[j] = keys[perm[j]];
  return result;
}

// keys of the factor in slot i (a typed factor's first `arity` keys; a container's own list)
static std::vector<Key> isam2_factor_keys(const ISAM2& S, size_t i) {
  if (i < S.isContainer.size() && S.isContainer[i]) return S.linearFactors[i].keys;
  const Factor& f = S.nonlinearFactors[i];
  return std::vector<Key>(f.keys, f.keys + kFactorArity[f.type]);
}

// BayesTree::removeClique gtsam/inference/BayesTree-inst.h:440-460
static void isam2_remove_clique(ISAM2& S, ICliquePtr clique) {
  ICliquePtr parent = clique->parent.lock();
  if (!parent) {
    auto it = std::find(S.roots.begin(), S.roots.end(), clique);
    if (it != S.roots.end()) S.roots.erase(it);
  } else {
    auto it = std::find(parent->children.begin(), parent->children.end(), clique);
    assert(it != parent->children.end());
    parent->children.erase(it);
  }
  for (auto& child : clique->children) child->parent.reset();
  for (int k = 0; k < clique->nFrontal; k++) S.nodes.erase(clique->keys[k]);
}

// BayesTree::removePath :464-486
static void isam2_remove_path(ISAM2& S, ICliquePtr clique, std::vector<ICliquePtr>* bn, std::list<ICliquePtr>* orphans) {
  if (!clique) return;
  orphans->remove(clique);
  ICliquePtr parent = clique->parent.lock();  // taken before removeClique, which does not touch the clique's own parent pointer
  isam2_remove_clique(S, clique);
  isam2_remove_path(S, parent, bn, orphans);
  orphans->insert(orphans->begin(), clique->children.begin(), clique->children.end());
  clique->children.clear();
  bn->push_back(clique);
}

// ISAM2Clique::findAll gtsam/nonlinear/ISAM2Clique.cpp:304-325
static void isam2_find_all(const ICliquePtr& c, const std::set<Key>& markedMask, std::set<Key>* keys) {
  bool found = false;
  for (size_t k = c->nFrontal; k < c->keys.size(); k++)
    if (markedMask.count(c->keys[k])) {
      found = true;
      break;
    }
  if (found)
    for (int k = 0; k < c->nFrontal; k++) keys->insert(c->keys[k]);
  for (auto& child : c->children) isam2_find_all(child, markedMask, keys);
}

// GaussianConditional::solve for one clique (gtsam/linear/GaussianConditional.cpp): x_F = R^-1 (d - S x_S)
static std::vector<double> isam2_solve_clique(const IClique& c, const VectorValues& x) {
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
  return rhs;
}

// DeltaImpl::UpdateGaussNewtonDelta gtsam/nonlinear/ISAM2-impl.cpp:47-77 with optimizeWildfireNonRecursive
// (ISAM2Clique.cpp:236-268; optimizeWildfireNode :211-234, isDirty :56-77, valuesChanged :151-158)
static void isam2_wildfire(ISAM2& S, double threshold, VectorValues& delta) {
  std::vector<ICliquePtr> stack;
  for (auto& root : S.roots) {
    std::set<Key> changed;  // one per root (optimizeWildfireNonRecursive's local)
    stack.clear();
    stack.push_back(root);
    while (!stack.empty()) {
      ICliquePtr c = stack.back();
      stack.pop_back();
      bool dirty = true;
      if (threshold > 0.0) {
        dirty = S.deltaReplacedMask.count(c->keys.front()) > 0;
        if (!dirty)
          for (size_t k = c->nFrontal; k < c->keys.size(); k++)
            if (changed.count(c->keys[k])) {
              dirty = true;
              break;
            }
      }
      if (!dirty) continue;
      const std::vector<double> sol = isam2_solve_clique(*c, delta);
      bool valuesChanged = true;
      if (threshold > 0.0 && !S.deltaReplacedMask.count(c->keys.front())) {
        double maxdiff = 0;
        int o = 0;
        for (int k = 0; k < c->nFrontal; k++)
          for (int d = 0; d < c->dims[k]; d++, o++) maxdiff = std::max(maxdiff, std::abs(delta.at(c->keys[k])[d] - sol[o]));
        valuesChanged = maxdiff >= threshold;
      }
      if (valuesChanged) {  // otherwise restoreFromOriginals: the old values stay
        int o = 0;
        for (int k = 0; k < c->nFrontal; k++) {
          auto& v = delta[c->keys[k]];
          v.assign(sol.begin() + o, sol.begin() + o + c->dims[k]);
          o += c->dims[k];
          changed.insert(c->keys[k]);
        }
      }
      for (auto& child : c->children) stack.push_back(child);
    }
  }
}

static double isam2_graph_error(const ISAM2& S, const Values& values);
static void isam2_for_each_clique(const ICliquePtr& c, const std::function<void(const IClique&)>& fn) {
  fn(*c);
  for (auto& ch : c->children) isam2_for_each_clique(ch, fn);
}
// ISAM2::error(x) = GaussianFactorGraph(*this).error(x) (ISAM2.cpp:820-823): 1/2 sum over the cliques of ||[R S] x - d||^2
static double isam2_tree_error(const ISAM2& S, const VectorValues& x) {
  double tot = 0;
  for (auto& root : S.roots)
    isam2_for_each_clique(root, [&](const IClique& c) {
      const int nf = c.RSd.r, n = c.RSd.c;
      for (int i = 0; i < nf; i++) {
        double e = -c.RSd(i, n - 1);
        int col = 0;
        for (size_t k = 0; k < c.keys.size(); k++) {
          const auto& xk = x.at(c.keys[k]);
          for (int d = 0; d < c.dims[k]; d++, col++) e += c.RSd(i, col) * xk[d];
        }
        tot += e * e;
      }
    });
  return 0.5 * tot;
}
// UpdateRgProd (ISAM2-impl.cpp:82-141): below a clique none of whose keys was replaced nothing has changed, so the walk stops there
static void isam2_update_rgprod(const ISAM2& S, const ICliquePtr& c, const VectorValues& grad, VectorValues* RgProd) {
  bool anyReplaced = false;
  for (Key j : c->keys)
    if (S.deltaReplacedMask.count(j)) {
      anyReplaced = true;
      break;
    }
  if (!anyReplaced) return;
  const int nf = c->RSd.r;
  std::vector<double> prod(nf, 0.0);
  int col = 0;
  for (size_t k = 0; k < c->keys.size(); k++) {
    const auto& gk = grad.at(c->keys[k]);
    for (int d = 0; d < c->dims[k]; d++, col++)
      for (int i = 0; i < nf; i++) prod[i] += c->RSd(i, col) * gk[d];
  }
  int o = 0;
  for (int k = 0; k < c->nFrontal; k++) {
    (*RgProd)[c->keys[k]].assign(prod.begin() + o, prod.begin() + o + c->dims[k]);
    o += c->dims[k];
  }
  for (auto& child : c->children) isam2_update_rgprod(S, child, grad, RgProd);
}
static double vv_norm2(const VectorValues& a) {
  double s = 0;
  for (auto& kv : a)
    for (double x : kv.second) s += x * x;
  return s;
}

// ISAM2::updateDelta gtsam/nonlinear/ISAM2.cpp:722-784
static void isam2_update_delta(ISAM2& S, bool forceFullSolve) {
  if (!S.dogleg) {
    isam2_wildfire(S, forceFullSolve ? 0.0 : S.wildfireThreshold, S.delta);
    S.deltaReplacedMask.clear();
    return;
  }
  // Powell's dog leg: Newton point by wildfire, steepest-descent point from the tree's gradient, one trust-region iteration
  isam2_wildfire(S, forceFullSolve ? 0.0 : S.doglegWildfireThreshold, S.deltaNewton);
  VectorValues grad;  // ISAM2::gradientAtZero :825-833: sum of the cliques' -[R S]^T d (ISAM2Clique.cpp:35-46, 328-343)
  for (auto& root : S.roots)
    isam2_for_each_clique(root, [&](const IClique& c) {
      const int nf = c.RSd.r, n = c.RSd.c;
      int col = 0;
      for (size_t k = 0; k < c.keys.size(); k++) {
        auto& gk = grad[c.keys[k]];
        gk.resize(c.dims[k], 0.0);
        for (int d = 0; d < c.dims[k]; d++, col++)
          for (int i = 0; i < nf; i++) gk[d] -= c.RSd(i, col) * c.RSd(i, n - 1);
      }
    });
  for (auto& root : S.roots) isam2_update_rgprod(S, root, grad, &S.RgProd);
  const double step = -vv_norm2(grad) / vv_norm2(S.RgProd);  // ComputeGradientSearch ISAM2-impl.cpp:144-157
  VectorValues dx_u = grad;
  for (auto& kv : dx_u)
    for (auto& x : kv.second) x *= step;
  S.deltaReplacedMask.clear();
  // DoglegOptimizerImpl::Iterate (DoglegOptimizerImpl.h:139-254) with Rd = this tree, f = the nonlinear graph, x0 = theta
  VectorValues dx_n;  // (over the variables of the tree, like dx_u: VectorValues arithmetic needs equal structure)
  for (auto& kv : dx_u) dx_n[kv.first] = S.deltaNewton.at(kv.first);
  const double f_error = isam2_graph_error(S, S.theta);
  VectorValues zero = dx_u;
  for (auto& kv : zero)
    for (auto& x : kv.second) x = 0.0;
  const double M_error = isam2_tree_error(S, zero);
  double delta = S.doglegDelta;
  const int mode = S.doglegAdaptationMode;
  enum { NONE, INCREASED_DELTA, DECREASED_DELTA } lastAction = NONE;
  VectorValues dx_d;
  bool stay = true;
  while (stay) {
    dx_d = dogleg_point(delta, dx_u, dx_n);
    Values x_d = S.theta;
    for (auto& kv : dx_d) x_d[kv.first] = retract(S.theta.at(kv.first), kv.second.data());
    const double new_f = isam2_graph_error(S, x_d);
    const double new_M = isam2_tree_error(S, dx_d);
    const double rho = (std::abs(f_error - new_f) < 1e-15 || std::abs(M_error - new_M) < 1e-15) ? 0.5 : (f_error - new_f) / (M_error - new_M);
    if (rho >= 0.75) {
      const double newDelta = std::max(delta, 3.0 * std::sqrt(vv_norm2(dx_d)));
      if (mode == 2 || mode == 1) {
        stay = false;
      } else if (std::abs(newDelta - delta) < 1e-15 || lastAction == DECREASED_DELTA) {
        stay = false;
      } else {
        stay = true;
        lastAction = INCREASED_DELTA;
      }
      delta = newDelta;
    } else if (rho >= 0.25) {
      stay = false;
    } else if (rho >= 0.0) {
      const bool hitMinimumDelta = !(delta > 1e-5);
      const double newDelta = hitMinimumDelta ? delta : 0.5 * delta;
      if (mode == 2 || lastAction == INCREASED_DELTA || hitMinimumDelta) {
        stay = false;
      } else {
        stay = true;
        lastAction = DECREASED_DELTA;
      }
      delta = newDelta;
    } else {  // the error went up (or rho is NaN, which the reference's assert only sees in debug builds)
      if (delta > 1e-5) {
        delta *= 0.5;
        stay = true;
        lastAction = DECREASED_DELTA;
      } else {
        for (auto& kv : dx_d)
          for (auto& x : kv.second) x = 0.0;
        stay = false;
      }
    }
  }
  S.doglegDelta = delta;
  for (auto& kv : dx_d) S.delta[kv.first] = kv.second;  // delta_ = doglegResult.dx_d
}

// the elimination of a junction tree into ISAM2 cliques: EliminatableClusterTree::eliminate with ISAM2Clique::setEliminationResult
// (gtsam/inference/ClusterTree-inst.h:219-266, 286-318; gtsam/nonlinear/ISAM2Clique.cpp:35-46)
static ICliquePtr isam2_eliminate_node(ISAM2& S, const std::shared_ptr<JNode>& node, const std::vector<IFactor>& graph,
                                       const std::map<Key, int>& keyDim, GFactor* sepOut) {
  auto cq = std::make_shared<IClique>();
  std::vector<GFactor> childFactors(node->children.size());
  for (size_t i = 0; i < node->children.size(); i++) {  // pre-order visitor: children in junction-tree order
    ICliquePtr ch = isam2_eliminate_node(S, node->children[i], graph, keyDim, &childFactors[i]);
    cq->children.push_back(ch);
    ch->parent = cq;
  }
  std::vector<const GFactor*> gathered;
  for (size_t f : node->factors)
    if (graph[f].g) gathered.push_back(graph[f].g);
  for (auto& cf : childFactors)
    if (!cf.empty()) gathered.push_back(&cf);
  for (size_t f : node->factors)  // orphan subtrees hang below the clique that eliminates their separator
    if (graph[f].orphan) {
      cq->children.push_back(graph[f].orphan);
      graph[f].orphan->parent = cq;
    }
  Clique tmp;
  GFactor sep;
  eliminate_clique(gathered, node->orderedFrontalKeys, keyDim, tmp, sep);
  cq->keys = tmp.keys;
  cq->dims = tmp.dims;
  cq->nFrontal = tmp.nFrontal;
  cq->RSd = tmp.RSd;
  cq->cached = sep;
  for (int k = 0; k < cq->nFrontal; k++) S.nodes[cq->keys[k]] = cq;
  *sepOut = sep;
  return cq;
}

static void isam2_eliminate(ISAM2& S, const std::vector<IFactor>& graph, const VariableIndex& vi, const std::vector<Key>& ordering) {
  std::vector<std::vector<Key>> fkeys;
  for (auto& f : graph) fkeys.push_back(f.keys);
  std::map<Key, int> keyDim;
  for (auto& kv : S.theta) keyDim[kv.first] = kVarDim[kv.second.type];
  auto eroots = build_etree(fkeys, vi, ordering);
  for (auto& r : eroots) {
    std::shared_ptr<JNode> jr;
    jt_visit(r, fkeys, jr);
    GFactor rem;
    S.roots.push_back(isam2_eliminate_node(S, jr, graph, keyDim, &rem));
  }
}

static int isam2_count_cliques(const ICliquePtr& c) {
  int n = 1;
  for (auto& ch : c->children) n += isam2_count_cliques(ch);
  return n;
}

// is the variable's delta above its relinearization threshold?  double: infinity norm >= threshold (ISAM2-impl.h:287, 361);
// FastMap<char, Vector>: any |delta_i| > threshold_i of the vector registered for the key's Symbol character (:252-268, 365-377)
static bool isam2_above_threshold(const ISAM2& S, Key key, const std::vector<double>& d, double scalarThreshold) {
  if (S.relinearizeThresholds.empty()) {
    double m = 0;
    for (double x : d) m = std::max(m, std::abs(x));
    return m >= scalarThreshold;
  }
  auto it = S.relinearizeThresholds.find((unsigned char)(key >> 56));
  if (it == S.relinearizeThresholds.end()) throw std::invalid_argument("ISAM2: no relinearization threshold for this Symbol character");
  if (it->second.size() != d.size())
    throw std::invalid_argument("Relinearization threshold vector dimensionality does not match actual variable dimensionality");
  for (size_t i = 0; i < d.size(); i++)
    if (std::abs(d[i]) > it->second[i]) return true;
  return false;
}

// CheckRelinearizationRecursiveDouble / Map, ISAM2-impl.h:246-300: every key of the clique's conditional (frontals AND parents) is
// checked; the children are only visited when one of them was above the threshold
static void isam2_check_relin_partial(const ISAM2& S, const ICliquePtr& c, std::set<Key>* relinKeys) {
  bool relinearize = false;
  for (Key var : c->keys)
    if (isam2_above_threshold(S, var, S.delta.at(var), S.relinearizeThreshold)) {
      relinKeys->insert(var);
      relinearize = true;
    }
  if (relinearize)
    for (auto& child : c->children) isam2_check_relin_partial(S, child, relinKeys);
}

static Values isam2_calculate_estimate(ISAM2& S, bool best);
static double isam2_graph_error(const ISAM2& S, const Values& values);

// ISAM2::update gtsam/nonlinear/ISAM2.cpp:419-480
static ISAM2Result isam2_update(ISAM2& S, const ISAM2UpdateParams& up) {
  S.update_count += 1;
  ISAM2Result result;
  std::vector<Factor> newFactors;
  newFactors.swap(S.newFactors);
  Values newTheta;
  newTheta.swap(S.newTheta);
  // addVariables :365-384
  for (auto& kv : newTheta) {
    if (S.theta.count(kv.first)) throw std::invalid_argument("ISAM2: variable already exists");
    S.theta[kv.first] = kv.second;
    S.delta[kv.first] = std::vector<double>(kVarDim[kv.second.type], 0.0);
    S.deltaNewton[kv.first] = S.delta[kv.first];
    S.RgProd[kv.first] = S.delta[kv.first];
  }
  const bool relinNeeded = up.force_relinearize || (S.enableRelinearization && S.relinearizeSkip > 0 && S.update_count % S.relinearizeSkip == 0);
  if (relinNeeded) isam2_update_delta(S, up.forceFullSolve);
  // 1. pushBackFactors (ISAM2-impl.h:141-173): FactorGraph::add_factors (FactorGraph-inst.h:109-137) -- the indices continue the list, or
  //    (findUnusedFactorSlots) the new factors fill the empty slots from the front; then the removals
  std::vector<size_t> newFactorsIndices;
  {
    size_t i = 0;
    for (auto& f : newFactors) {
      if (S.findUnusedFactorSlots) {
        while (i < S.nonlinearFactors.size() && !S.removedFactor[i]) ++i;
      } else {
        i = S.nonlinearFactors.size();
      }
      if (i >= S.nonlinearFactors.size()) {
        S.nonlinearFactors.push_back(f);
        S.removedFactor.push_back(0);
        S.isContainer.push_back(0);
        S.linearFactors.push_back(GFactor());
        i = S.nonlinearFactors.size() - 1;
      } else {
        S.nonlinearFactors[i] = f;
        S.removedFactor[i] = 0;
        S.isContainer[i] = 0;
      }
      newFactorsIndices.push_back(i);
    }
  }
  const std::set<size_t> newIndexSet(newFactorsIndices.begin(), newFactorsIndices.end());
  std::set<Key> keysWithRemovedFactors;
  for (size_t index : up.removeFactorIndices) {
    if (index >= S.nonlinearFactors.size() || newIndexSet.count(index)) throw std::invalid_argument("ISAM2: removeFactorIndices out of range");
    if (S.removedFactor[index]) continue;  // an empty slot: nothing to take out of the variable index
    for (Key key : isam2_factor_keys(S, index)) {
      keysWithRemovedFactors.insert(key);
      auto& entries = S.variableIndex.at(key);  // VariableIndex::remove (VariableIndex-inl.h:52-80): the key keeps its (emptier) list
      entries.erase(std::find(entries.begin(), entries.end(), index));
    }
    S.removedFactor[index] = 1;
    S.isContainer[index] = 0;
    S.linearFactors[index] = GFactor();
  }
  // computeUnusedKeys :175-190: keys whose last factor went and which no new factor mentions
  std::set<Key> newFactorKeys, unusedKeys;
  for (auto& f : newFactors)
    for (int k = 0; k < kFactorArity[f.type]; k++) newFactorKeys.insert(f.keys[k]);
  for (Key key : keysWithRemovedFactors)
    if (S.variableIndex.at(key).empty() && !newFactorKeys.count(key)) unusedKeys.insert(key);
  // 2. errorBefore (ISAM2.cpp:444-446): the graph WITH the new factors at calculateEstimate(), which brings delta up to date first
  if (S.evaluateNonlinearError) S.errorBefore = isam2_graph_error(S, isam2_calculate_estimate(S, false));
  // 3. gatherInvolvedKeys :199-226: keys of the new factors, of the removed factors, extraReelimKeys; updateKeys :228-244:
  //    observedKeys = the marked keys that stay in the system
  std::set<Key> markedKeys = newFactorKeys;
  markedKeys.insert(keysWithRemovedFactors.begin(), keysWithRemovedFactors.end());
  markedKeys.insert(up.extraReelimKeys.begin(), up.extraReelimKeys.end());
  std::vector<Key> observedKeys;
  for (Key k : markedKeys)
    if (!unusedKeys.count(k)) observedKeys.push_back(k);
  std::set<Key> relinKeys;
  if (relinNeeded) {
    // 4. gatherRelinearizeKeys :367-399: CheckRelinearizationFull (:353-383, max |delta_j| >= threshold) minus noRelinKeys
    if (up.forceFullSolve) {  // CheckRelinearizationFull(delta, 0.0): everything
      for (auto& kd : S.delta) relinKeys.insert(kd.first);
    } else if (S.enablePartialRelinearizationCheck) {  // CheckRelinearizationPartial :315-331
      for (auto& root : S.roots) isam2_check_relin_partial(S, root, &relinKeys);
    } else {  // CheckRelinearizationFull :344-378
      for (auto& kd : S.delta)
        if (isam2_above_threshold(S, kd.first, kd.second, S.relinearizeThreshold)) relinKeys.insert(kd.first);
    }
    for (Key k : S.fixedVariables) relinKeys.erase(k);  // "keys whose linearization points are fixed" (ISAM2-impl.h:385-388)
    for (Key k : up.noRelinKeys) relinKeys.erase(k);
    markedKeys.insert(relinKeys.begin(), relinKeys.end());
    if (!relinKeys.empty()) {
      // 5. findFluid (:431-451), 6. theta.retractMasked(delta, relinKeys)
      for (auto& root : S.roots) isam2_find_all(root, relinKeys, &markedKeys);
      for (Key k : relinKeys) S.theta[k] = retract(S.theta.at(k), S.delta.at(k).data());
    }
    result.variablesRelinearized = (int)markedKeys.size();
  }
  // 7. linearizeNewFactors (:454-468) + augmentVariableIndex
  for (size_t i : newFactorsIndices) {
    S.linearFactors[i] = linearize_factor(S.nonlinearFactors[i], S.theta);
    for (int k = 0; k < kFactorArity[S.nonlinearFactors[i].type]; k++) {  // VariableIndex::augment appends (VariableIndex-inl.h:27-49)
      S.variableIndex[S.nonlinearFactors[i].keys[k]].push_back(i);
    }
  }
  // 8. recalculate (ISAM2.cpp:117-175)
  if (!markedKeys.empty() || !observedKeys.empty()) {
    std::vector<ICliquePtr> affectedBayesNet;
    std::list<ICliquePtr> orphans;
    for (Key j : markedKeys) {  // removeTop BayesTree-inst.h:491-508
      auto node = S.nodes.find(j);
      if (node != S.nodes.end()) isam2_remove_path(S, node->second, &affectedBayesNet, &orphans);
    }
    std::vector<Key> affectedKeys;
    for (auto& c : affectedBayesNet)
      for (int k = 0; k < c->nFrontal; k++) affectedKeys.push_back(c->keys[k]);
    std::set<Key> affectedKeysSet;
    if ((double)affectedKeys.size() >= (double)S.theta.size() * 0.65) {
      // ---- recalculateBatch :178-247
      result.batch = 1;
      VariableIndex affectedFactorsVarIndex = S.variableIndex;
      for (Key key : unusedKeys) affectedFactorsVarIndex.erase(key);
      for (auto& kv : affectedFactorsVarIndex) affectedKeysSet.insert(kv.first);
      std::vector<Key> order;
      if (up.hasConstrainedKeys) {
        order = colamd_constrained(S, affectedFactorsVarIndex, S.nonlinearFactors.size(), up.constrainedKeys);
      } else if (S.theta.size() > observedKeys.size()) {
        std::map<Key, int> groups;
        for (Key var : observedKeys) groups[var] = 1;
        order = colamd_constrained(S, affectedFactorsVarIndex, S.nonlinearFactors.size(), groups);
      } else {
        order = colamd_constrained(S, affectedFactorsVarIndex, S.nonlinearFactors.size(), {});
      }
      std::vector<IFactor> graph(S.linearFactors.size());  // an empty slot stays an entry without keys (the indices are the row ids)
      for (size_t i = 0; i < S.nonlinearFactors.size(); i++) {
        if (S.removedFactor[i]) continue;
        if (!S.isContainer[i]) S.linearFactors[i] = linearize_factor(S.nonlinearFactors[i], S.theta);  // (a container linearizes to itself)
        graph[i].g = &S.linearFactors[i];
        graph[i].keys = S.linearFactors[i].keys;
      }
      S.roots.clear();
      S.nodes.clear();
      isam2_eliminate(S, graph, affectedFactorsVarIndex, order);
      result.variablesReeliminated = (int)affectedKeysSet.size();
      result.factorsRecalculated = (int)S.nonlinearFactors.size();
    } else {
      // ---- recalculateIncremental :250-362
      std::vector<Key> affectedAndNewKeys = affectedKeys;
      affectedAndNewKeys.insert(affectedAndNewKeys.end(), observedKeys.begin(), observedKeys.end());
      // relinearizeAffectedFactors :66-114
      std::set<size_t> candidates;
      for (Key key : affectedAndNewKeys)
        for (size_t f : S.variableIndex.at(key)) candidates.insert(f);
      const std::set<Key> inSet(affectedAndNewKeys.begin(), affectedAndNewKeys.end());
      std::vector<IFactor> factors;
      for (size_t idx : candidates) {
        bool inside = true, useCachedLinear = true;
        const Factor& nf = S.nonlinearFactors[idx];
        for (Key key : isam2_factor_keys(S, idx)) {
          if (!inSet.count(key)) {
            inside = false;
            break;
          }
          if (relinKeys.count(key)) useCachedLinear = false;
        }
        if (!inside) continue;
        if (!useCachedLinear && !S.isContainer[idx]) S.linearFactors[idx] = linearize_factor(nf, S.theta);
        IFactor f;
        f.g = &S.linearFactors[idx];
        f.keys = S.linearFactors[idx].keys;
        factors.push_back(f);
      }
      result.variablesReeliminated = (int)affectedAndNewKeys.size();
      result.factorsRecalculated = (int)factors.size();
      for (auto& orphan : orphans) {  // GetCachedBoundaryFactors ISAM2-impl.h:499-509
        IFactor f;
        f.g = &orphan->cached;
        f.keys = orphan->cached.keys;
        factors.push_back(f);
      }
      for (auto& orphan : orphans) {  // BayesTreeOrphanWrapper: keys = the orphan's separator
        IFactor f;
        f.orphan = orphan;
        f.keys.assign(orphan->keys.begin() + orphan->nFrontal, orphan->keys.end());
        factors.push_back(f);
      }
      affectedKeysSet.insert(markedKeys.begin(), markedKeys.end());
      affectedKeysSet.insert(affectedKeys.begin(), affectedKeys.end());
      VariableIndex affectedFactorsVarIndex;
      for (size_t i = 0; i < factors.size(); i++)
        for (Key k : factors[i].keys) affectedFactorsVarIndex[k].push_back(i);
      std::map<Key, int> constraintGroups;
      if (up.hasConstrainedKeys) {
        constraintGroups = up.constrainedKeys;
      } else {
        const int group = observedKeys.size() < affectedFactorsVarIndex.size() ? 1 : 0;
        for (Key var : observedKeys) constraintGroups.emplace(var, group);
      }
      for (auto it = constraintGroups.begin(); it != constraintGroups.end();) {  // "Remove unaffected keys from the constraints"
        if (unusedKeys.count(it->first) || !affectedKeysSet.count(it->first)) it = constraintGroups.erase(it);
        else ++it;
      }
      const std::vector<Key> ordering = colamd_constrained(S, affectedFactorsVarIndex, factors.size(), constraintGroups);
      isam2_eliminate(S, factors, affectedFactorsVarIndex, ordering);
    }
    S.deltaReplacedMask.insert(affectedKeysSet.begin(), affectedKeysSet.end());
  }
  // removeVariables ISAM2.cpp:385-398
  for (Key key : unusedKeys) {
    S.variableIndex.erase(key);
    S.delta.erase(key);
    S.deltaNewton.erase(key);
    S.RgProd.erase(key);
    S.deltaReplacedMask.erase(key);
    S.nodes.erase(key);
    S.theta.erase(key);
    S.fixedVariables.erase(key);
  }
  S.lastUnusedKeys.assign(unusedKeys.begin(), unusedKeys.end());
  if (S.evaluateNonlinearError) S.errorAfter = isam2_graph_error(S, isam2_calculate_estimate(S, false));  // ISAM2.cpp:481-483
  result.cliques = 0;
  for (auto& r : S.roots) result.cliques += isam2_count_cliques(r);
  S.last = result;
  return result;
}

// nonlinearFactors_.error(values) over the factors still in the graph (NonlinearFactorGraph.cpp:170-179: empty slots are skipped)
static double isam2_graph_error(const ISAM2& S, const Values& values) {
  double total = 0;
  for (size_t i = 0; i < S.nonlinearFactors.size(); i++)
    if (!S.removedFactor[i] && !S.isContainer[i]) total += factor_error(S.nonlinearFactors[i], values);  // (a container without a linearization point: 0)
  return total;
}

// ISAM2::calculateEstimate :748-754 (getDelta :776-779)
static Values isam2_calculate_estimate(ISAM2& S, bool best) {
  if (best) isam2_update_delta(S, true);  // calculateBestEstimate :763-766
  else if (!S.deltaReplacedMask.empty()) isam2_update_delta(S, false);
  Values out;
  for (auto& kv : S.theta) out[kv.first] = retract(kv.second, S.delta.at(kv.first).data());
  return out;
}

// BayesTree::removeSubtree gtsam/inference/BayesTree-inst.h:512-547: the clique leaves its parent (or the roots); it and everything below it
// leave the nodes index; returned breadth-first like the reference's list
static std::vector<ICliquePtr> isam2_remove_subtree(ISAM2& S, const ICliquePtr& subtree) {
  std::vector<ICliquePtr> cliques{subtree};
  ICliquePtr parent = subtree->parent.lock();
  if (parent) parent->children.erase(std::find(parent->children.begin(), parent->children.end(), subtree));
  else S.roots.erase(std::find(S.roots.begin(), S.roots.end(), subtree));
  for (size_t i = 0; i < cliques.size(); i++) {
    ICliquePtr c = cliques[i];
    for (auto& child : c->children) cliques.push_back(child);
    for (int k = 0; k < c->nFrontal; k++) S.nodes.erase(c->keys[k]);
    c->parent.reset();
    c->children.clear();
  }
  return cliques;
}

// ISAM2::marginalizeLeaves gtsam/nonlinear/ISAM2.cpp:487-720.  marginalFactorsIndices / deletedFactorsIndices may be null.
static void isam2_marginalize_leaves(ISAM2& S, const std::vector<Key>& leafKeysList, std::vector<size_t>* marginalFactorsIndices,
                                     std::vector<size_t>* deletedFactorsIndices) {
  const std::set<Key> leafKeys(leafKeysList.begin(), leafKeysList.end());
  std::map<Key, std::vector<GFactor>> marginalFactors;  // front key of a clique -> marginals passed up to it
  std::set<Key> leafKeysRemoved;
  std::set<size_t> factorIndicesToRemove;
  std::map<Key, int> keyDim;
  for (auto& kv : S.theta) keyDim[kv.first] = kVarDim[kv.second.type];
  auto trackingRemoveSubtree = [&](const ICliquePtr& subtreeRoot) {
    const std::vector<ICliquePtr> removed = isam2_remove_subtree(S, subtreeRoot);
    for (const ICliquePtr& rc : removed) {
      marginalFactors.erase(rc->keys[0]);
      for (int k = 0; k < rc->nFrontal; k++) {
        const Key frontal = rc->keys[k];
        leafKeysRemoved.insert(frontal);
        const auto& involved = S.variableIndex.at(frontal);
        factorIndicesToRemove.insert(involved.begin(), involved.end());
        if (!leafKeys.count(frontal))
          throw std::runtime_error("Requesting to marginalize variables that are not leaves, the ISAM2 object is now in an inconsistent state so should no longer be used.");
      }
    }
    return removed;
  };
  for (Key j : leafKeys) {
    if (leafKeysRemoved.count(j)) continue;
    ICliquePtr clique = S.nodes.at(j);
    while (ICliquePtr parent = clique->parent.lock()) {  // up to the root of the marginalized subtree
      if (leafKeys.count(parent->keys[0])) clique = parent;
      else break;
    }
    bool marginalizeEntireClique = true;
    for (int k = 0; k < clique->nFrontal; k++)
      if (!leafKeys.count(clique->keys[k])) {
        marginalizeEntireClique = false;
        break;
      }
    if (marginalizeEntireClique) {
      // the whole clique and its subtree go; its cached factor is the marginal on its separator and belongs to its parent from now on
      if (ICliquePtr parent = clique->parent.lock()) marginalFactors[parent->keys[0]].push_back(clique->cached);
      trackingRemoveSubtree(clique);
    } else {
      // re-eliminate the marginalized frontals of this clique with the marginals of its removed children and the factors they pull in
      std::vector<GFactor> graph;
      std::vector<ICliquePtr> subtreesToRemove;
      for (const ICliquePtr& child : clique->children)
        for (size_t k = child->nFrontal; k < child->keys.size(); k++)
          if (leafKeys.count(child->keys[k])) {
            subtreesToRemove.push_back(child);
            graph.push_back(child->cached);
            break;
          }
      std::vector<ICliquePtr> childrenRemoved;
      for (const ICliquePtr& st : subtreesToRemove) {
        const std::vector<ICliquePtr> removed = trackingRemoveSubtree(st);
        childrenRemoved.insert(childrenRemoved.end(), removed.begin(), removed.end());
      }
      std::set<size_t> factorsFromMarginalizedInClique;
      for (int k = 0; k < clique->nFrontal; k++)
        if (leafKeys.count(clique->keys[k])) {
          const auto& involved = S.variableIndex.at(clique->keys[k]);
          factorsFromMarginalizedInClique.insert(involved.begin(), involved.end());
        }
      for (const ICliquePtr& rc : childrenRemoved)
        for (int k = 0; k < rc->nFrontal; k++)
          for (size_t f : S.variableIndex.at(rc->keys[k])) factorsFromMarginalizedInClique.erase(f);
      for (size_t index : factorsFromMarginalizedInClique)  // nonlinearFactors_[index]->linearize(theta_): a container gives itself
        graph.push_back(S.isContainer[index] ? S.linearFactors[index] : linearize_factor(S.nonlinearFactors[index], S.theta));
      std::vector<Key> cliqueFrontalsToEliminate;
      {
        std::set<Key> cf(clique->keys.begin(), clique->keys.begin() + clique->nFrontal);
        for (Key k : cf)
          if (leafKeys.count(k)) cliqueFrontalsToEliminate.push_back(k);  // set_intersection: ascending by key
      }
      std::vector<const GFactor*> gathered;
      for (auto& g : graph) gathered.push_back(&g);
      Clique tmp;
      GFactor marginal;
      eliminate_clique(gathered, cliqueFrontalsToEliminate, keyDim, tmp, marginal);
      marginalFactors[clique->keys[0]].push_back(marginal);  // (keyed by the front key the clique has BEFORE the split)
      // split the clique: its leading leaf keys go, the conditional on the rest stays as it is
      size_t nToRemove = 0;
      while (nToRemove < clique->keys.size() && leafKeys.count(clique->keys[nToRemove])) ++nToRemove;
      int dimToRemove = 0;
      for (size_t k = 0; k < nToRemove; k++) dimToRemove += clique->dims[k];
      {
        const int nfOld = clique->RSd.r, nOld = clique->RSd.c;
        Mat R2(nfOld - dimToRemove, nOld - dimToRemove);
        for (int i = dimToRemove; i < nfOld; i++)
          for (int jj = dimToRemove; jj < nOld; jj++) R2(i - dimToRemove, jj - dimToRemove) = clique->RSd(i, jj);
        clique->RSd = R2;
      }
      for (Key k : cliqueFrontalsToEliminate) S.nodes.erase(k);  // (removeVariables below does it in the reference)
      clique->keys.erase(clique->keys.begin(), clique->keys.begin() + nToRemove);
      clique->dims.erase(clique->dims.begin(), clique->dims.begin() + nToRemove);
      clique->nFrontal -= (int)nToRemove;
      for (Key frontal : cliqueFrontalsToEliminate) {
        const auto& involved = S.variableIndex.at(frontal);
        factorIndicesToRemove.insert(involved.begin(), involved.end());
      }
      leafKeysRemoved.insert(cliqueFrontalsToEliminate.begin(), cliqueFrontalsToEliminate.end());
    }
  }
  // the factors the marginals summarise leave the graph (and the variable index)
  for (size_t index : factorIndicesToRemove) {
    for (Key key : isam2_factor_keys(S, index)) {
      auto& entries = S.variableIndex.at(key);
      entries.erase(std::find(entries.begin(), entries.end(), index));
    }
    S.removedFactor[index] = 1;
    S.isContainer[index] = 0;
    S.linearFactors[index] = GFactor();
  }
  // the marginal factors enter it as LinearContainerFactors (no linearization point); their keys are fixed from now on
  std::vector<GFactor> factorsToAdd;
  for (auto& kf : marginalFactors)
    for (auto& f : kf.second) {  // (a marginal on no keys still takes a slot, as the reference's non-null check lets it)
      factorsToAdd.push_back(f);
      for (Key k : f.keys) S.fixedVariables.insert(k);
    }
  std::vector<size_t> newFactorIndices;
  {
    size_t i = 0;
    for (auto& f : factorsToAdd) {
      if (S.findUnusedFactorSlots) {
        while (i < S.nonlinearFactors.size() && !S.removedFactor[i]) ++i;
      } else {
        i = S.nonlinearFactors.size();
      }
      if (i >= S.nonlinearFactors.size()) {
        S.nonlinearFactors.push_back(Factor());
        S.removedFactor.push_back(0);
        S.isContainer.push_back(1);
        S.linearFactors.push_back(f);
        i = S.nonlinearFactors.size() - 1;
      } else {
        S.removedFactor[i] = 0;
        S.isContainer[i] = 1;
        S.linearFactors[i] = f;
      }
      newFactorIndices.push_back(i);
      for (Key k : f.keys) S.variableIndex[k].push_back(i);
    }
  }
  // removeVariables(leafKeys) ISAM2.cpp:385-398
  for (Key key : leafKeys) {
    S.variableIndex.erase(key);
    S.delta.erase(key);
    S.deltaNewton.erase(key);
    S.RgProd.erase(key);
    S.deltaReplacedMask.erase(key);
    S.nodes.erase(key);
    S.theta.erase(key);
    S.fixedVariables.erase(key);
  }
  if (deletedFactorsIndices) deletedFactorsIndices->assign(factorIndicesToRemove.begin(), factorIndicesToRemove.end());
  if (marginalFactorsIndices) *marginalFactorsIndices = newFactorIndices;
}

static void isam2_collect(const ICliquePtr& c, std::vector<ICliquePtr>* out) {
  out->push_back(c);
  for (auto& ch : c->children) isam2_collect(ch, out);
}

}  // namespace orc
