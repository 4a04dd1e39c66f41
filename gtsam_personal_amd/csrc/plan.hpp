// Host-side symbolic analysis for the multifrontal solve: variable index, elimination tree,
// junction tree (GTSAM's clique-merge rule) and the per-front assembly plan that the HIP kernels consume.
// Done ONCE per (graph structure, ordering); the reference redoes it inside every solve
// (gtsam/inference/EliminateableFactorGraph-inst.h:123-146).  Array-based, no pointers-to-nodes.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace lmgpu {

static const int kNumVarTypes = 6;
static const int kVarDim[6] = {3, 6, 3, 9, 2, 5};
static const int kVarStore[6] = {3, 12, 3, 15, 2, 5};
static const int kMaxArity = 3;
static const int kFactorArity[12] = {2, 2, 2, 1, 1, 1, 1, 2, 2, 2, 3, 1};
static const int kFactorRows[12] = {2, 3, 6, 3, 6, 3, 9, 2, 2, 2, 2, 5};
static const int kFactorMeas[12] = {2, 3, 12, 3, 12, 3, 15, 7, 19, 2, 2, 5};
// variable types each factor type expects (for validation)
static const int kFactorVar0[12] = {3, 0, 1, 0, 1, 2, 3, 1, 1, 0, 1, 5};
static const int kFactorVar1[12] = {2, 0, 1, -1, -1, -1, -1, 2, 2, 4, 2, -1};
static const int kFactorVar2[12] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 5, -1};
inline int factor_var_type(int ftype, int k) { return k == 0 ? kFactorVar0[ftype] : (k == 1 ? kFactorVar1[ftype] : kFactorVar2[ftype]); }

struct FactorRef {
  int32_t bucket;   // bucket index
  int32_t idx;      // index inside the bucket
  int32_t slots[3]; // variable slots (-1 beyond the factor's arity)
  int32_t graph_index;
};

// One front (= one Bayes-tree clique).  Fronts are stored in post-order (children before parents),
// the order the reference's post-order visitor eliminates them in (gtsam/inference/ClusterTree-inst.h:219-266).
struct Front {
  std::vector<int32_t> vars;     // slots: frontals in GTSAM's orderedFrontalKeys order, then separators sorted by Key
  int32_t n_frontal_vars = 0;
  std::vector<int32_t> col_off;  // scalar column offset of each var inside the front (size vars.size()+1; last = rhs column)
  int32_t nf = 0;                // frontal scalar dim
  int32_t n = 0;                 // nf + ns + 1
  std::vector<int32_t> factors;  // indices into Plan::factors (own factors first, then merged children's, reference order)
  std::vector<int32_t> children; // front indices, in the reference's child order
  int32_t parent = -1;
  int32_t level = 0;             // 0 = leaf; parents are > all their children
  int32_t cls = 0;               // 0: assembled + factored in LDS by one workgroup; 1: lives in HBM, multi-kernel dense path
};

// result of symbolic_multifrontal: fronts in post-order (children before parents)
struct SymbolicFronts {
  struct F {
    std::vector<int32_t> frontals;  // in the reference's orderedFrontalKeys order
    std::vector<int32_t> sep;       // sorted by key rank
    std::vector<int32_t> factors;   // own factors first, then merged children's (reference order)
    std::vector<int32_t> children;  // front indices, reference child order
    int32_t parent = -1, level = 0;
  };
  std::vector<F> fronts;
  std::vector<int32_t> roots, etree_parent;
};
// var_factors (optional): per slot its factor indices in the order of the caller's VariableIndex (ISAM2 keeps one incrementally: with
// findUnusedFactorSlots its lists are not ascending, and EliminationTree-inst.h:94-134 walks them as they are); default: ascending
std::string symbolic_multifrontal(int32_t n, const std::vector<int32_t>& keyrank, const std::vector<std::vector<int32_t>>& fvars,
                                  SymbolicFronts* out, const std::vector<std::vector<int32_t>>* var_factors = nullptr);

struct Plan {
  // inputs
  int32_t n_vars = 0;
  std::vector<uint64_t> keys;   // per slot
  std::vector<int32_t> types;   // per slot
  std::vector<int32_t> dims;    // per slot
  std::vector<int32_t> xoff;    // scalar offset of slot in packed tangent vectors (size n_vars+1)
  std::vector<int32_t> voff;    // double offset of slot in packed values (size n_vars+1)
  std::vector<int32_t> tidx;    // index of the slot within its type's device array
  int32_t type_count[kNumVarTypes] = {};
  std::vector<FactorRef> factors;  // sorted by graph_index
  // outputs
  std::vector<int32_t> etree_parent;  // per slot, -1 for roots
  std::vector<Front> fronts;
  std::vector<int32_t> roots;
  std::vector<int32_t> front_of_var;  // front in which the slot is frontal
  int32_t n_levels = 0;
  int32_t max_front_n = 0;

  // Build everything.  lds_limit_n: fronts with n <= lds_limit_n are class 0.
  // Returns empty string on success, else an error message.
  std::string build(int32_t lds_limit_n);
};

}  // namespace lmgpu
