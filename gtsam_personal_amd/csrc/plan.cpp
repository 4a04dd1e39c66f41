// Host symbolic analysis (see plan.hpp).  Restates, array-based:
//   VariableIndex           gtsam/inference/VariableIndex-inl.h:27-49
//   EliminationTree ctor    gtsam/inference/EliminationTree-inst.h:78-156
//   JunctionTree ctor       gtsam/inference/JunctionTree-inst.h:65-153 (merge rule :100-120)
//   Cluster::mergeChildren  gtsam/inference/ClusterTree-inst.h:58-96
//   Scatter key order       gtsam/linear/Scatter.cpp:39-73
#include "plan.hpp"

#include <algorithm>
#include <numeric>

namespace lmgpu {

// The symbolic multifrontal analysis for factors of ANY arity (variables 0 .. n-1 in elimination order, factors in index order):
// elimination tree, junction tree with the reference's clique-merge rule, fronts in post-order.  Plan::build feeds it the
// unary / binary factors of a nonlinear graph; the incremental path (csrc/isam2.hpp) feeds it the affected factors, the
// cached boundary factors and the orphan subtrees' separators (gtsam/nonlinear/ISAM2.cpp:250-362).
std::string symbolic_multifrontal(int32_t n, const std::vector<int32_t>& keyrank, const std::vector<std::vector<int32_t>>& fvars,
                                  SymbolicFronts* out, const std::vector<std::vector<int32_t>>* var_factors) {
  const int32_t none = -1;
  const int32_t m = (int32_t)fvars.size();
  std::vector<std::vector<int32_t>> vi(n);
  for (int32_t i = 0; i < m; i++)
    for (int32_t v : fvars[i]) {
      if (v < 0 || v >= n) return "factor references unknown variable";
      if (!var_factors) vi[v].push_back(i);
    }
  if (var_factors) {  // the caller's VariableIndex: the elimination tree hooks children in the order a variable's factors are listed
    if ((int32_t)var_factors->size() != n) return "variable index of the wrong size";
    vi = *var_factors;
    for (int32_t s = 0; s < n; s++)
      for (int32_t i : vi[s])
        if (i < 0 || i >= m || std::find(fvars[i].begin(), fvars[i].end(), s) == fvars[i].end()) return "variable index does not match the factors";
  }
  for (int32_t s = 0; s < n; s++)
    if (vi[s].empty()) return "EliminationTree: given ordering contains variables that are not involved in the factor graph";

  // ---- elimination tree (EliminationTree-inst.h:94-134); `anc` only accelerates the root walk ----
  std::vector<int32_t>& etree_parent = out->etree_parent;
  etree_parent.assign(n, none);
  std::vector<int32_t> anc(n, none), prevCol(m, none);
  std::vector<std::vector<int32_t>> echildren(n), efactors(n);
  for (int32_t j = 0; j < n; j++) {
    for (int32_t i : vi[j]) {
      if (prevCol[i] != none) {
        int32_t r = prevCol[i];
        while (anc[r] != none) r = anc[r];
        // path compression
        int32_t x = prevCol[i];
        while (anc[x] != none && anc[x] != r) {
          int32_t nx = anc[x];
          anc[x] = r;
          x = nx;
        }
        if (r != j) {
          etree_parent[r] = j;
          anc[r] = j;
          echildren[j].push_back(r);
        }
      } else {
        efactors[j].push_back(i);
      }
      prevCol[i] = j;
    }
  }

  // ---- junction tree by post-order traversal of the elimination tree ----
  // per ETree node (== JT cluster until merged into its parent)
  std::vector<std::vector<int32_t>> sep(n);        // symbolic separator, sorted by key rank
  std::vector<std::vector<int32_t>> jfront(n);     // orderedFrontalKeys (slots)
  std::vector<std::vector<int32_t>> jfactors(n);   // factor indices
  std::vector<std::vector<int32_t>> jchildren(n);  // remaining JT children (ETree node ids)
  std::vector<char> absorbed(n, 0);
  std::vector<int32_t> mark(n, -1);
  // iterative post-order over the forest (roots in slot order, children in hook order)
  std::vector<int32_t> eroots;
  for (int32_t j = 0; j < n; j++)
    if (etree_parent[j] == none) eroots.push_back(j);
  std::vector<std::pair<int32_t, size_t>> stack;
  auto visit_post = [&](int32_t j) {
    // symbolic elimination of j (JunctionTree-inst.h:79-97)
    std::vector<int32_t>& s = sep[j];
    s.clear();
    for (int32_t f : efactors[j])
      for (int32_t v : fvars[f])
        if (v != j && mark[v] != j) {
          mark[v] = j;
          s.push_back(v);
        }
    for (int32_t c : echildren[j])
      for (int32_t v : sep[c])
        if (v != j && mark[v] != j) {
          mark[v] = j;
          s.push_back(v);
        }
    std::sort(s.begin(), s.end(), [&](int32_t a, int32_t b) { return keyrank[a] < keyrank[b]; });
    const size_t myNrParents = s.size();
    jfront[j].assign(1, j);
    jfactors[j] = efactors[j];
    // merge decision (JunctionTree-inst.h:100-120): children in order, running myNrFrontals
    const std::vector<int32_t>& ch = echildren[j];
    std::vector<char> merge(ch.size(), 0);
    size_t myNrFrontals = 1;
    for (size_t i = 0; i < ch.size(); i++) {
      if (myNrParents + myNrFrontals == sep[ch[i]].size()) {
        myNrFrontals += jfront[ch[i]].size();
        merge[i] = 1;
      }
    }
    // mergeChildren (ClusterTree-inst.h:58-96)
    for (size_t i = 0; i < ch.size(); i++) {
      const int32_t c = ch[i];
      if (merge[i]) {
        jfront[j].insert(jfront[j].end(), jfront[c].rbegin(), jfront[c].rend());
        jfactors[j].insert(jfactors[j].end(), jfactors[c].begin(), jfactors[c].end());
        jchildren[j].insert(jchildren[j].end(), jchildren[c].begin(), jchildren[c].end());
        absorbed[c] = 1;
        std::vector<int32_t>().swap(jfront[c]);
        std::vector<int32_t>().swap(jfactors[c]);
        std::vector<int32_t>().swap(jchildren[c]);
      } else {
        jchildren[j].push_back(c);
      }
    }
    std::reverse(jfront[j].begin(), jfront[j].end());
    // separators of absorbed children are no longer needed (kept ones define their front's Scatter)
    for (int32_t c : ch)
      if (absorbed[c]) std::vector<int32_t>().swap(sep[c]);
  };
  for (int32_t r : eroots) {
    stack.emplace_back(r, 0);
    while (!stack.empty()) {
      auto& top = stack.back();
      const int32_t j = top.first;
      if (top.second < echildren[j].size()) {
        const int32_t c = echildren[j][top.second++];
        stack.emplace_back(c, 0);
      } else {
        visit_post(j);
        stack.pop_back();
      }
    }
  }

  // ---- fronts in post-order over the junction tree ----
  out->fronts.clear();
  out->roots.clear();
  std::vector<int32_t> front_id(n, -1);
  for (int32_t r : eroots) {
    stack.emplace_back(r, 0);
    while (!stack.empty()) {
      auto& top = stack.back();
      const int32_t j = top.first;
      if (top.second < jchildren[j].size()) {
        const int32_t c = jchildren[j][top.second++];
        stack.emplace_back(c, 0);
      } else {
        SymbolicFronts::F fr;
        fr.frontals = jfront[j];
        fr.sep = sep[j];
        fr.factors = jfactors[j];
        int32_t lvl = 0;
        for (int32_t c : jchildren[j]) {
          fr.children.push_back(front_id[c]);
          lvl = std::max(lvl, out->fronts[front_id[c]].level + 1);
        }
        fr.level = lvl;
        const int32_t id = (int32_t)out->fronts.size();
        for (int32_t c : fr.children) out->fronts[c].parent = id;
        front_id[j] = id;
        out->fronts.push_back(std::move(fr));
        stack.pop_back();
      }
    }
    out->roots.push_back(front_id[r]);
  }
  return "";
}

std::string Plan::build(int32_t lds_limit_n) {
  const int32_t n = n_vars;
  if (n <= 0) return "no variables";
  dims.resize(n);
  xoff.assign(n + 1, 0);
  voff.assign(n + 1, 0);
  tidx.resize(n);
  for (int t = 0; t < kNumVarTypes; t++) type_count[t] = 0;
  for (int32_t s = 0; s < n; s++) {
    const int t = types[s];
    if (t < 0 || t >= kNumVarTypes) return "bad variable type";
    dims[s] = kVarDim[t];
    xoff[s + 1] = xoff[s] + dims[s];
    voff[s + 1] = voff[s] + kVarStore[t];
    tidx[s] = type_count[t]++;
  }
  // rank of each slot's Key (separators are sorted by Key, Scatter.cpp:69-72)
  std::vector<int32_t> bykey(n), keyrank(n);
  std::iota(bykey.begin(), bykey.end(), 0);
  std::sort(bykey.begin(), bykey.end(), [&](int32_t a, int32_t b) { return keys[a] < keys[b]; });
  for (int32_t r = 0; r < n; r++) {
    if (r > 0 && keys[bykey[r]] == keys[bykey[r - 1]]) return "duplicate variable key";
    keyrank[bykey[r]] = r;
  }

  // factors sorted by graph index (VariableIndex lists factor indices ascending)
  std::sort(factors.begin(), factors.end(), [](const FactorRef& a, const FactorRef& b) { return a.graph_index < b.graph_index; });
  const int32_t m = (int32_t)factors.size();
  std::vector<std::vector<int32_t>> fvars(m);
  for (int32_t i = 0; i < m; i++) {
    const FactorRef& f = factors[i];
    for (int k = 0; k < kMaxArity; k++) {
      if (f.slots[k] < 0) continue;
      if (f.slots[k] >= n) return "factor references unknown slot";
      for (int q = 0; q < k; q++)
        if (f.slots[q] == f.slots[k]) return "factor with repeated variable";
      fvars[i].push_back(f.slots[k]);
    }
  }
  SymbolicFronts sf;
  const std::string e = symbolic_multifrontal(n, keyrank, fvars, &sf);
  if (!e.empty()) return e;
  etree_parent = sf.etree_parent;

  // ---- fronts in post-order over the junction tree ----
  fronts.clear();
  roots = sf.roots;
  front_of_var.assign(n, -1);
  for (size_t id = 0; id < sf.fronts.size(); id++) {
    SymbolicFronts::F& sfr = sf.fronts[id];
    Front fr;
    fr.n_frontal_vars = (int32_t)sfr.frontals.size();
    fr.vars = sfr.frontals;
    fr.vars.insert(fr.vars.end(), sfr.sep.begin(), sfr.sep.end());
    fr.col_off.resize(fr.vars.size() + 1);
    int32_t off = 0;
    for (size_t k = 0; k < fr.vars.size(); k++) {
      fr.col_off[k] = off;
      off += dims[fr.vars[k]];
      if ((int32_t)k == fr.n_frontal_vars - 1) fr.nf = off;
    }
    fr.col_off[fr.vars.size()] = off;
    fr.n = off + 1;
    fr.factors.swap(sfr.factors);
    fr.children.swap(sfr.children);
    fr.parent = sfr.parent;
    fr.level = sfr.level;
    fr.cls = (fr.n <= lds_limit_n) ? 0 : 1;
    for (int32_t k = 0; k < fr.n_frontal_vars; k++) front_of_var[fr.vars[k]] = (int32_t)id;
    max_front_n = std::max(max_front_n, fr.n);
    n_levels = std::max(n_levels, fr.level + 1);
    fronts.push_back(std::move(fr));
  }
  return "";
}

}  // namespace lmgpu
