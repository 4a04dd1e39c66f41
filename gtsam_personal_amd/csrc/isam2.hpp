// ISAM2 on the device (BASELINE config 5): incremental relinearization of the affected factor subset +
// partial re-elimination of the top of a DEVICE-RESIDENT Bayes tree + wildfire back-substitution.  Included by lmgpu.hip.
//
// What runs where (same split as the batch path: symbolic bookkeeping on the host, every number on the device):
//   host    ISAM2::update's bookkeeping (gtsam/nonlinear/ISAM2.cpp:419-480, ISAM2-impl.h:113-508): new variables / factors, marked
//           and relinearization keys, findFluid, BayesTree::removeTop (gtsam/inference/BayesTree-inst.h:440-508), the constrained
//           COLAMD call (a boundary input: lmgpu_ccolamd_fn, bound by the reference side to its own ccolamd), elimination tree and
//           junction tree of the affected part (plan.cpp: symbolic_multifrontal, n-ary factors = relinearized factors, cached
//           boundary factors, orphan subtrees' separators, ISAM2.cpp:250-362)
//   device  theta, delta, the linearization cache ([A b] per factor = linearFactors_), [R S d] and the cached separator factor of
//           every clique (ISAM2Clique::cachedFactor_); kernels: the bucket factor kernels on an index list (relinearizeAffectedFactors
//           :66-114, linearizeNewFactors), retract on an index list (retractMasked :465), lds_front_kernel per level of the new
//           cliques (own Jacobians + cached boundary factors / children as extend-add; cliques of more than 139 scalar columns go
//           through the dense-front kernels instead: hbm_assemble_rows_kernel, diag_potrf_kernel / panel_trsm_kernel, the MFMA
//           trailing update), isam2_wildfire_kernel = the top-down walk as a device-side work list over the dirty part of the tree
//           (optimizeWildfireNonRecursive, ISAM2Clique.cpp:211-268), isam2_tree_patch_kernel = the device copy of the tree, patched.
// Storage: one pool of doubles (offsets, so it can grow): Jacobian regions per factor bucket, [R S d] + update matrix per clique
// (one dense block for a wide clique) with exact-size free lists (clique shapes repeat in SLAM), the children lists of the cliques.
// Everything an update sends to the device goes through a pinned staging arena: no allocation, one wait per update (two with
// relinearization).
// ISAM2UpdateParams (gtsam/nonlinear/ISAM2UpdateParams.h:30-90) honoured: removeFactorIndices (a removed factor leaves an empty slot, a
// variable that lost its last factor leaves the system: pushBackFactors / computeUnusedKeys / removeVariables), constrainedKeys,
// noRelinKeys, extraReelimKeys, force_relinearize, forceFullSolve.
// ISAM2Params: Gauss-Newton or Dogleg optimisation params, relinearization thresholds (double or per Symbol character), partial check,
// evaluateNonlinearError, findUnusedFactorSlots.  ISAM2::marginalizeLeaves: is_marginalize_leaves below.
// Limits (fail loudly): Cholesky, no newAffectedKeys (smart factors).
#pragma once

#include <chrono>
#include <list>
#include <set>
#include <type_traits>

struct lmgpu_isam2 {
  lmgpu_config cfg{};
  lmgpu_isam2_params prm{};
  lmgpu_ccolamd_fn ccolamd = nullptr;
  void* user = nullptr;
  int device = -1;
  std::string err;
  uint64_t failed_key = 0;
  hipStream_t stream = nullptr;

  struct Var {
    uint64_t key;
    int32_t type, tidx, xoff;
    bool dead;  // removed from the system (ISAM2::removeVariables): its id, value slot and delta scalars are not reused
  };
  std::vector<Var> vars;
  std::map<uint64_t, int32_t> vid_of;  // the LIVE variables, ascending by key = the order of the reference's Values / VectorValues / VariableIndex
  int type_count[kNumVarTypes] = {}, type_cap[kNumVarTypes] = {};
  double* theta[kNumVarTypes] = {};
  double* est[kNumVarTypes] = {};
  int32_t* d_type_xoff[kNumVarTypes] = {};
  int ntot = 0, ntot_cap = 0;
  double *delta = nullptr, *ones = nullptr;
  unsigned char *d_replaced = nullptr, *d_changed = nullptr;  // per scalar of delta

  struct Fac {
    int32_t type, bucket, lidx, v[3];
    bool removed;       // an empty slot of nonlinearFactors_ (NonlinearFactorGraph::remove): reused only with findUnusedFactorSlots
    int32_t marg = -1;  // >= 0: the slot holds a LinearContainerFactor left by marginalizeLeaves (type = -1): margs[marg]
  };
  std::vector<Fac> facs;
  // A marginal factor (ISAM2.cpp:684-697: LinearContainerFactor around a HessianFactor, no linearization point): its augmented information
  // matrix is the update matrix a clique or a one-front elimination left in the pool (upper, row-major, m = scalars + 1, stride ld); it
  // enters later eliminations through a column map exactly like a cached boundary factor, is never relinearized and has error 0
  // (LinearContainerFactor.cpp:77-79, 104-108).
  struct Marg {
    std::vector<int32_t> vids;  // in the order of the matrix's blocks
    int64_t u_off = -1;
    int32_t ld = 0, m = 0;
    int64_t blk_off = -1;  // the pool block that holds it (given back when the factor goes)
    size_t blk_n = 0;
  };
  std::vector<Marg> margs;
  std::vector<int32_t> free_margs;
  std::set<uint64_t> fixed;        // fixedVariables_ (ISAM2.h:103): keys of marginal factors, never relinearized (ISAM2-impl.h:385-388)
  bool find_unused_slots = false;  // ISAM2Params::findUnusedFactorSlots
  std::vector<uint64_t> last_marginal_idx, last_deleted_idx;  // marginalizeLeaves' two optional outputs, of the last call
  struct Bkt {
    int type = 0, noise_kind = 0, rows = 0, cols = 0, ml = 0, nl = 0, ar = 0;
    int robust = 0;  // noiseModel::Robust around the Gaussian model of every factor of the bucket (lmgpu_robust_kind), tuning constant
    double rk = 0.0;
    int n = 0, cap = 0;
    int64_t joff = -1;  // pool offset of the bucket's Jacobians (cap x rows x cols)
    int32_t* d_vidx = nullptr;
    int32_t* d_epos = nullptr;  // per row: 1 + its index in the factor list = its place in the error buffer; 0 (a dump slot) once removed
    double *d_meas = nullptr, *d_noise = nullptr;
  };
  std::vector<Bkt> bkts;
  std::vector<std::vector<int32_t>> vindex;  // VariableIndex: per variable the factor indices, ascending

  struct Clq {
    std::vector<int32_t> vars;  // vids: frontals (elimination order), then separators ascending by key (Scatter order)
    int32_t nfv = 0, nf = 0, n = 0;
    int64_t rsd_off = -1, u_off = -1;
    // ld == 0: an LDS clique -- [R S d] nf x n at rsd_off, the update matrix (n - nf)^2 at u_off, both dense.
    // ld > 0: a clique of more than 139 scalar columns, eliminated in place in ONE n x ld block of the pool (f_off) by the dense-front
    //         kernels of the batch path: rsd_off = f_off (rows 0 .. nf-1), u_off = f_off + nf ld + nf (the trailing block), stride ld.
    int32_t ld = 0;
    int64_t f_off = -1;
    int64_t xrow_off = -1;  // ld > 0: pool offset of its delta offsets as int32 [nf frontal | n - nf - 1 separator] (too long for a tree row)
    int64_t kids_off = -1;  // pool offset (doubles) of the children's clique ids as int32 (the wildfire kernel pushes them), -1: none
    int32_t kids_n = 0;
    std::vector<int32_t> children;
    int32_t parent = -1;
    bool alive = false;
  };
  std::vector<Clq> clq;
  int n_alive = 0;  // cliques alive = cliques of the tree once an update has re-attached its orphans (ISAM2Result's clique count)
  std::vector<int32_t> free_clq, roots, node_of;  // node_of: per variable the clique it is frontal in (-1: none yet)
  std::vector<char> replaced;                     // deltaReplacedMask_, per variable
  bool any_replaced = false;

  double* pool = nullptr;
  size_t pool_cap = 0, pool_top = 0;
  std::map<size_t, std::vector<int64_t>> freelist;

  // pending input of the next update
  struct NewVar {
    uint64_t key;
    int32_t type;
    double v[15];
  };
  std::vector<NewVar> new_vars;
  struct NewFac {
    int32_t type, noise_kind;
    uint64_t k[3];
    std::vector<double> meas, noise;
    int32_t robust = 0;  // lmgpu_robust_kind around the Gaussian model (0 = none) and its tuning constant
    double rk = 0.0;
  };
  std::vector<NewFac> new_facs;
  int update_count = 0;
  // ISAM2UpdateParams of one update
  struct UpParams {
    std::vector<uint64_t> remove;
    bool has_constrained = false;
    std::map<uint64_t, int> constrained;
    std::vector<uint64_t> no_relin, extra_reelim;
    bool force_relinearize = false, force_full_solve = false;
  };
  std::vector<uint64_t> last_unused;  // ISAM2Result::unusedKeys of the last update
  // ISAM2Params::relinearizeThreshold as FastMap<char, Vector> (non-empty: in force) and enablePartialRelinearizationCheck
  std::map<unsigned char, std::vector<double>> relin_thresholds;
  bool partial_relin_check = false;
  // ISAM2Params::optimizationParams = ISAM2DoglegParams (ISAM2Params.h:68-110): Powell's dog leg in updateDelta (ISAM2.cpp:739-779)
  bool dogleg = false;
  double dogleg_delta = 1.0, dogleg_wildfire = 1e-5;  // doglegDelta_ (the trust-region radius), ISAM2DoglegParams::wildfireThreshold
  int dogleg_mode = 0;                                // 0 SEARCH_EACH_ITERATION, 1 SEARCH_REDUCE_ONLY, 2 ONE_STEP_PER_ITERATION
  double *delta_newton = nullptr, *rgprod = nullptr, *grad = nullptr, *dx_u = nullptr;  // laid out like delta
  double *d_cerr = nullptr, *d_dlscal = nullptr, *h_dlscal = nullptr;  // per-clique tree errors; eight scalars on the device / pinned
  size_t cerr_cap = 0;
  double* d_gpart = nullptr;  // the gradient's terms, a slot per (clique, column): summed per scalar in a fixed order
  size_t gpart_cap = 0;
  // ISAM2Params::evaluateNonlinearError (ISAM2Params.h:200-203): ISAM2Result::errorBefore / errorAfter of the last update
  bool evaluate_error = false;
  double error_before = 0, error_after = 0;
  double *d_ebuf = nullptr, *d_epart = nullptr, *h_escal = nullptr;  // per-factor errors (slot 0 = dump), partial sums + the total, pinned total
  size_t ebuf_cap = 0;

  // device scratch
  int *d_status = nullptr, *h_status = nullptr;
  // Per-update staging: everything an update sends to the device goes through ONE pinned host arena and (for tables only the
  // update's kernels read) a device arena of the same size -- asynchronous copies from memory that stays valid until the next update,
  // no allocation, no wait.  (One hipMalloc + copy + wait + hipFree per index list and per table, and three blocking copies per new
  // factor, were most of the 0.4-0.7 ms an update cost.)  Requests beyond the arena get a chunk of their own, freed at the next update.
  char *h_stage = nullptr, *d_stage = nullptr;
  size_t stage_cap = 0, stage_used = 0, stage_want = 0;
  std::vector<std::pair<void*, void*>> stage_extra;  // (pinned host, device): overflow slabs
  char *ovf_h = nullptr, *ovf_d = nullptr;           // the current one
  size_t ovf_used = 0, ovf_cap = 0;
  // Deferred uploads: what is_stage / is_push put into the pinned arena since the last flush travels to the device arena in ONE copy,
  // and the pushes (payload in the arena -> a persistent device array) are carried out by ONE scatter kernel (is_flush) -- an update
  // issued 28 copy commands of a few hundred bytes each before (rocprofv3: half of its GPU time, and ~4 us of host time apiece).
  size_t stage_flushed = 0;
  // Zero-copy (round 3): the pinned arena is mapped into the device's address space, and tables a kernel reads ONCE (index lists, the
  // payload of pushes, the tree patch) are read by the kernel straight from it -- no copy command at all.  Only the elimination tables,
  // which the front kernels walk with chains of dependent reads, still travel into the device arena: bytes below stage_copy_mark.
  // (An update of VisualISAM2Example issued ~35 device operations, 13 of them copies of a few hundred bytes, each ~6 us of stream time.)
  size_t stage_copy_mark = 0;
  unsigned char epoch = 1;  // value that means 'set' in d_replaced / d_changed for the NEXT back-substitution (no clears between them)
  int* h_status_dev = nullptr;  // device-side address of the pinned status word: the last kernel of a phase relays the status there
  double *h_val = nullptr, *h_val_dev = nullptr;  // pinned + mapped: calculateEstimate(key) is retracted straight into it (no copy command)
  struct PushRec {
    void* dst;
    const void* src;  // in the device arena
    uint32_t words, pad;
  };
  std::vector<PushRec> pushes;
  unsigned int* d_eticket = nullptr;  // ticket counter of the merged elimination launch
  double* inv16 = nullptr;    // 16 x 256 doubles: the 16 x 16 inverses of the panel being factored (wide cliques)
  double* d_marg = nullptr;   // marginalCovariance: one work vector per column of the block + the block itself
  size_t marg_cap = 0;
  double* h_delta = nullptr;  // pinned copy of delta for CheckRelinearizationFull
  size_t h_delta_cap = 0;
  // (Gauss-Newton mode) the pinned copy is a MIRROR: delta only changes in the walk, which stores what it keeps to both places (no copy
  // command behind the walk).  mirror_ntot = leading scalars of h_delta that equal the device's; 0 = not valid (a full copy brings it back)
  double* h_delta_dev = nullptr;
  size_t mirror_ntot = 0;
  // what a walk needs on the device besides the tree patch (work-list seeds, counters, a fresh status word, the all-ones fill of the
  // re-eliminated top) has been pushed -- by the update, whose elimination flush carries it (one scatter launch less per update + walk)
  bool delta_zero_pending = false;  // a zero-fill record over new scalars of delta waits for its flush (an all-ones fill must not share it)
  bool walk_prepared = false;
  unsigned int seeded = 0;  // queue slots the last seeding wrote
  // An update ends with the walk (behind its elimination, under the same wait) when delta is going to be asked for before the next
  // elimination anyway: the next update checks relinearization, or the caller read an estimate after the previous update.
  int delta_reads = 0;      // readers of delta since the last update returned
  // LMGPU_ISAM2_TRACE=1: wall time per phase of update(), printed when the handle is destroyed (development aid)
  bool trace = false;
  double t_phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::vector<int> elim_cid;  // cliques of the elimination whose status word is still on the device (checked when the update ends)
  bool elim_pending = false;
  bool elim_relay_pending = false;  // the elimination's status word has not been handed to the host yet
  // Device copy of the Bayes tree for the wildfire, PATCHED per update (only the cliques an update created or re-parented are
  // rewritten): one FrontDesc per clique slot (child_begin / child_count = its children as a list of clique ids in the pool), its frontal
  // and separator delta offsets in fixed-stride rows (kTreeRow ints per slot).
  std::vector<int32_t> touched;  // clique slots whose descriptor changed since the last patch
  FrontDesc* d_tree = nullptr;
  int32_t *d_tree_fx = nullptr, *d_tree_sx = nullptr;
  size_t tree_slots = 0;         // capacity of the three arrays, in clique slots
  long long* d_queue = nullptr;  // work list of the wildfire kernel (clique id | parent clique id << 32; -1 = not published; consumers reset what they take)
  unsigned char* d_tree_done = nullptr;  // per clique slot: the epoch of the walk that finished it (a child starts under its parent's solve and waits for this)
  size_t queue_cap = 0;
  unsigned int* d_wl = nullptr;  // [0] tail, [1] next ticket, [2] items published and not finished
  size_t tree_lds = 0;           // largest nf x (n | 1) of a clique: doubles of LDS the wildfire kernel stages
  // taps
  std::vector<int32_t> snap;
};

// the pending pushes of an update: record i = words 4-byte words from the device staging arena to their place in a persistent array
__global__ __launch_bounds__(256) void isam2_scatter_kernel(const lmgpu_isam2::PushRec* __restrict__ recs) {
  const lmgpu_isam2::PushRec r = recs[blockIdx.x];
  uint32_t* dst = (uint32_t*)r.dst;
  const uint32_t* src = (const uint32_t*)r.src;
  if (r.pad == 1) {  // a fill with the all-ones pattern ("not published yet": the update matrices of a merged elimination launch)
    for (uint32_t w = threadIdx.x; w < r.words; w += 256) dst[w] = 0xffffffffu;
    return;
  }
  if (r.pad == 2) {  // `words` BYTES set to the value carried in src (deltaReplacedMask_ |= affected keys: the walk's epoch)
    const unsigned char v = (unsigned char)(uintptr_t)r.src;
    for (uint32_t w = threadIdx.x; w < r.words; w += 256) ((unsigned char*)r.dst)[w] = v;
    return;
  }
  // (the source is pinned host memory read over the link: four loads in flight per thread, one round trip per 4 KB, not one per KB)
  for (uint32_t w = threadIdx.x; w < r.words; w += 1024) {
    uint32_t v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = (w + 256 * u < r.words) ? src[w + 256 * u] : 0u;
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (w + 256 * u < r.words) dst[w + 256 * u] = v[u];
  }
}

namespace {

#define ISCHECK(expr)                                                 \
  do {                                                                \
    hipError_t _e = (expr);                                           \
    if (_e != hipSuccess) {                                           \
      S->err = std::string(#expr) + ": " + hipGetErrorString(_e);     \
      return LMGPU_HIP_ERROR;                                         \
    }                                                                 \
  } while (0)

int is_flush(lmgpu_isam2* S);
// reallocate a device array to exactly `newcap` elements, keeping its `used` leading elements
template <typename T>
int is_realloc(lmgpu_isam2* S, T** p, size_t newcap, size_t used) {
  const int rcf = is_flush(S);  // a pending push may point into the array that goes away
  if (rcf) return rcf;
  T* q = nullptr;
  ISCHECK(hipMalloc((void**)&q, std::max<size_t>(1, newcap) * sizeof(T)));
  if (*p && used) ISCHECK(hipMemcpyAsync(q, *p, used * sizeof(T), hipMemcpyDeviceToDevice, S->stream));
  ISCHECK(hipStreamSynchronize(S->stream));
  if (*p) (void)hipFree(*p);
  *p = q;
  return LMGPU_OK;
}
// (floor 2048: a reallocation is a hipMalloc / hipHostMalloc + copy + wait + free, several milliseconds -- the 16 variables of
// VisualISAM2Example crossed the old floor of 64 scalars in their last frame and paid 7 ms for it)
inline size_t is_next_cap(size_t cap, size_t need) { return std::max<size_t>(need, std::max<size_t>(2048, cap * 2)); }

// start of an update: the stream is idle (every entry point ends with a wait), so the arenas can be reused / regrown
int is_stage_begin(lmgpu_isam2* S) {
  const int rcf = is_flush(S);
  if (rcf) return rcf;
  ISCHECK(hipStreamSynchronize(S->stream));
  for (auto& e : S->stage_extra) {
    (void)hipHostFree(e.first);
    if (e.second) (void)hipFree(e.second);
  }
  S->stage_extra.clear();
  S->ovf_h = S->ovf_d = nullptr;
  S->ovf_used = S->ovf_cap = 0;
  size_t floor_bytes = size_t(1) << 20;
  if (const char* e = dev_switch("LMGPU_ISAM2_STAGE_BYTES")) floor_bytes = (size_t)std::max(64L, atol(e));  // tests: a tiny arena, every request overflows
  const size_t want = dev_switch("LMGPU_ISAM2_STAGE_BYTES") ? floor_bytes : std::max<size_t>(floor_bytes, 2 * S->stage_want);
  if (want > S->stage_cap) {
    if (S->h_stage) (void)hipHostFree(S->h_stage);
    if (S->d_stage) (void)hipFree(S->d_stage);
    S->h_stage = S->d_stage = nullptr;
    S->stage_cap = 0;
    ISCHECK(hipHostMalloc((void**)&S->h_stage, want, hipHostMallocMapped));  // kernels read it in place (zero-copy tables)
    {
      void* dev = nullptr;
      ISCHECK(hipHostGetDevicePointer(&dev, S->h_stage, 0));
      if (dev != (void*)S->h_stage) {
        S->err = "ISAM2: pinned host memory is not mapped at its host address on this device";
        return LMGPU_HIP_ERROR;
      }
    }
    ISCHECK(hipMalloc((void**)&S->d_stage, want));
    S->stage_cap = want;
  }
  S->stage_used = 0;
  S->stage_flushed = 0;
  S->stage_copy_mark = 0;
  S->stage_want = 0;
  return LMGPU_OK;
}
// Read-only entry points stage tables too, but only an update starts a new arena: between updates (marginals or single estimates of
// thousands of variables) the arena would fill up and every request would take the overflow path (a pinned allocation each).  They call
// this first: every entry point ends with a wait, so the stream is idle and a half-full arena can start over.
int is_stage_recycle(lmgpu_isam2* S) { return (S->stage_used > S->stage_cap / 2 || !S->stage_extra.empty()) ? is_stage_begin(S) : LMGPU_OK; }

// pinned bytes for `bytes` of payload (+ where the device arena mirrors them, if asked for)
int is_stage_raw(lmgpu_isam2* S, size_t bytes, char** hp, char** dp) {
  const size_t b = (std::max<size_t>(bytes, 1) + 63) & ~size_t(63);
  S->stage_want += b;
  if (S->stage_used + b <= S->stage_cap) {
    *hp = S->h_stage + S->stage_used;
    if (dp) *dp = S->d_stage + S->stage_used;
    S->stage_used += b;
    return LMGPU_OK;
  }
  // beyond the arena: overflow slabs of the arena's size (freed when the next update starts a bigger arena), requests bump-allocated from
  // the current one -- a pinned allocation per request was 0.1 ms each, and a loop closure that re-eliminates thousands of cliques makes
  // thousands of requests (400 ms in the last update of city10000, and 6 s to free them one by one)
  if (!S->ovf_h || S->ovf_used + b > S->ovf_cap) {
    void *h = nullptr, *d = nullptr;
    const size_t cap = dev_switch("LMGPU_ISAM2_STAGE_BYTES") ? b : std::max(b, std::max<size_t>(S->stage_cap, size_t(1) << 20));
    if (S->trace) std::fprintf(stderr, "isam2 staging: overflow slab of %zu bytes (request %zu, arena %zu of %zu used) in update %d\n", cap, b, S->stage_used, S->stage_cap, S->update_count);
    ISCHECK(hipHostMalloc(&h, cap, hipHostMallocMapped));
    ISCHECK(hipMalloc(&d, cap));
    S->stage_extra.emplace_back(h, d);
    S->ovf_h = (char*)h;
    S->ovf_d = (char*)d;
    S->ovf_used = 0;
    S->ovf_cap = cap;
  }
  *hp = S->ovf_h + S->ovf_used;
  if (dp) *dp = S->ovf_d + S->ovf_used;
  S->ovf_used += b;
  return LMGPU_OK;
}
// a table the update's kernels read: host vector -> device arena (with the next flush; a request beyond the arena is copied at once)
// device_copy = false (the default): the kernel reads the pinned host bytes themselves (read once: an index list, a patch blob)
template <typename T>
int is_stage(lmgpu_isam2* S, const std::vector<T>& src, T** d, bool device_copy = false) {
  char *hp, *dp = nullptr;
  const int rc = is_stage_raw(S, src.size() * sizeof(T), &hp, device_copy ? &dp : nullptr);
  if (rc) return rc;
  if (!src.empty()) std::memcpy(hp, src.data(), src.size() * sizeof(T));
  if (!device_copy) {
    *d = (T*)hp;  // pinned host memory is mapped at the same address on the device
    return LMGPU_OK;
  }
  if (!(S->h_stage && hp >= S->h_stage && hp < S->h_stage + S->stage_cap)) {  // from an overflow slab
    if (!src.empty()) ISCHECK(hipMemcpyAsync(dp, hp, src.size() * sizeof(T), hipMemcpyHostToDevice, S->stream));
  } else {
    S->stage_copy_mark = S->stage_used;
  }
  *d = (T*)dp;
  return LMGPU_OK;
}
// host data into a persistent device array (the source is copied into the pinned arena; the scatter kernel of the next flush moves it)
int is_push(lmgpu_isam2* S, void* dst, const void* src, size_t bytes) {
  if (!bytes) return LMGPU_OK;
  char* hp;
  const int rc = is_stage_raw(S, bytes, &hp, nullptr);
  if (rc) return rc;
  std::memcpy(hp, src, bytes);
  if ((bytes & 3) != 0) {  // not whole words: its own copy, now
    ISCHECK(hipMemcpyAsync(dst, hp, bytes, hipMemcpyHostToDevice, S->stream));
    return LMGPU_OK;
  }
  S->pushes.push_back(lmgpu_isam2::PushRec{dst, hp, (uint32_t)(bytes >> 2), 0u});  // the scatter kernel reads the pinned bytes
  return LMGPU_OK;
}
// everything staged or pushed so far reaches the device: one copy of the arena's new part, one scatter kernel for the pushes.
// Called before anything that consumes staged tables or pushed arrays is launched (and before an array a push points into moves).
int is_flush(lmgpu_isam2* S) {
  const lmgpu_isam2::PushRec* d_recs = nullptr;
  if (S->stage_copy_mark > S->stage_flushed) {
    // device-arena tables staged since the last flush: carried by the scatter kernel too (records of 4 KB, one round trip over the link
    // each, side by side: it reads the pinned arena in place) -- a copy command of its own was one more dependent device operation per update
    const size_t bytes = (S->stage_copy_mark - S->stage_flushed + 3) & ~size_t(3);
    if (bytes <= (size_t(1) << 20)) {
      for (size_t o = 0; o < bytes; o += 4096) {
        const size_t nb = std::min<size_t>(4096, bytes - o);
        S->pushes.push_back(lmgpu_isam2::PushRec{S->d_stage + S->stage_flushed + o, S->h_stage + S->stage_flushed + o, (uint32_t)(nb >> 2), 0u});
      }
    } else {  // (a batch step's tables: the copy engine)
      ISCHECK(hipMemcpyAsync(S->d_stage + S->stage_flushed, S->h_stage + S->stage_flushed, S->stage_copy_mark - S->stage_flushed, hipMemcpyHostToDevice,
                             S->stream));
    }
    S->stage_flushed = S->stage_copy_mark;
  }
  const size_t npush = S->pushes.size();
  S->delta_zero_pending = false;
  bool recs_in_arena = true;
  if (npush) {
    char* hp;
    const int rc = is_stage_raw(S, npush * sizeof(lmgpu_isam2::PushRec), &hp, nullptr);
    if (rc) return rc;
    std::memcpy(hp, S->pushes.data(), npush * sizeof(lmgpu_isam2::PushRec));
    d_recs = (const lmgpu_isam2::PushRec*)hp;
    S->pushes.clear();
  }
  (void)recs_in_arena;
  if (npush) hipLaunchKernelGGL(isam2_scatter_kernel, dim3((unsigned)npush), dim3(256), 0, S->stream, d_recs);
  return LMGPU_OK;
}

int is_pool_alloc(lmgpu_isam2* S, size_t n, int64_t* off) {
  n = std::max<size_t>(n, 1);
  auto it = S->freelist.find(n);
  if (it != S->freelist.end() && !it->second.empty()) {
    *off = it->second.back();
    it->second.pop_back();
    return LMGPU_OK;
  }
  if (S->pool_top + n + 64 > S->pool_cap) {
    const size_t cap = is_next_cap(S->pool_cap, std::max<size_t>(S->pool_top + n + 64, 1 << 20));  // (8 MB to start with: a growth step is a reallocation of several milliseconds)
    const int rc = is_realloc(S, &S->pool, cap, S->pool_top);
    if (rc) return rc;
    ISCHECK(hipMemsetAsync(S->pool + S->pool_top, 0, (cap - S->pool_top) * sizeof(double), S->stream));
    S->pool_cap = cap;
  }
  *off = (int64_t)S->pool_top;
  S->pool_top += n;
  return LMGPU_OK;
}
void is_pool_free(lmgpu_isam2* S, int64_t off, size_t n) {
  if (off >= 0) S->freelist[std::max<size_t>(n, 1)].push_back(off);
}

// the variables of the factor in slot i (a typed factor's first `arity` variables; a marginal factor's own list)
std::vector<int32_t> is_fac_vids(const lmgpu_isam2* S, const lmgpu_isam2::Fac& f) {
  if (f.marg >= 0) return S->margs[f.marg].vids;
  return std::vector<int32_t>(f.v, f.v + kFactorArity[f.type]);
}
void is_free_marg(lmgpu_isam2* S, int32_t mi) {
  lmgpu_isam2::Marg& m = S->margs[mi];
  is_pool_free(S, m.blk_off, m.blk_n);
  m = lmgpu_isam2::Marg();
  S->free_margs.push_back(mi);
}
int32_t is_new_marg(lmgpu_isam2* S, const lmgpu_isam2::Marg& m) {
  if (!S->free_margs.empty()) {
    const int32_t mi = S->free_margs.back();
    S->free_margs.pop_back();
    S->margs[mi] = m;
    return mi;
  }
  S->margs.push_back(m);
  return (int32_t)S->margs.size() - 1;
}

int is_new_clique(lmgpu_isam2* S) {
  int id;
  if (!S->free_clq.empty()) {
    id = S->free_clq.back();
    S->free_clq.pop_back();
    S->clq[id] = lmgpu_isam2::Clq();
  } else {
    id = (int)S->clq.size();
    S->clq.emplace_back();
  }
  S->clq[id].alive = true;
  S->n_alive++;
  return id;
}
// keep_u: the clique's update matrix (its cached factor) lives on as a marginal factor -- `*keep` takes over the block that holds it
void is_release_clique(lmgpu_isam2* S, int id, lmgpu_isam2::Marg* keep = nullptr) {
  lmgpu_isam2::Clq& c = S->clq[id];
  if (keep) {
    keep->vids.assign(c.vars.begin() + c.nfv, c.vars.end());
    keep->u_off = c.u_off;
    keep->m = c.n - c.nf;
    keep->ld = c.ld > 0 ? c.ld : c.n - c.nf;
    keep->blk_off = c.ld > 0 ? c.f_off : c.u_off;
    keep->blk_n = c.ld > 0 ? (size_t)c.n * c.ld : (size_t)(c.n - c.nf) * (c.n - c.nf);
  }
  if (c.ld > 0) {
    if (!keep) is_pool_free(S, c.f_off, (size_t)c.n * c.ld);
  } else {
    is_pool_free(S, c.rsd_off, (size_t)c.nf * c.n);
    if (!keep) is_pool_free(S, c.u_off, (size_t)(c.n - c.nf) * (c.n - c.nf));
  }
  if (c.xrow_off >= 0) is_pool_free(S, c.xrow_off, (size_t)c.n / 2 + 1);
  c.xrow_off = -1;
  c.ld = 0;
  c.f_off = -1;
  if (c.kids_off >= 0) is_pool_free(S, c.kids_off, (size_t)(c.kids_n + 1) / 2);
  c.kids_off = -1;
  c.kids_n = 0;
  c.alive = false;
  S->n_alive--;
  c.children.clear();
  c.vars.clear();
  S->free_clq.push_back(id);
}

// BayesTree::removeClique gtsam/inference/BayesTree-inst.h:440-460 (the clique's storage is released by the caller once its
// conditional has been read for the affected keys)
void is_remove_clique(lmgpu_isam2* S, int id) {
  lmgpu_isam2::Clq& c = S->clq[id];
  if (c.parent < 0) {
    auto it = std::find(S->roots.begin(), S->roots.end(), id);
    if (it != S->roots.end()) S->roots.erase(it);
  } else {
    auto& pc = S->clq[c.parent].children;
    pc.erase(std::find(pc.begin(), pc.end(), id));
  }
  for (int ch : c.children) S->clq[ch].parent = -1;
  for (int k = 0; k < c.nfv; k++) S->node_of[c.vars[k]] = -1;
}
// BayesTree::removePath :464-486
void is_remove_path(lmgpu_isam2* S, int id, std::vector<int>* bn, std::list<int>* orphans) {
  if (id < 0) return;
  orphans->remove(id);
  const int parent = S->clq[id].parent;
  is_remove_clique(S, id);
  is_remove_path(S, parent, bn, orphans);
  orphans->insert(orphans->begin(), S->clq[id].children.begin(), S->clq[id].children.end());
  S->clq[id].children.clear();
  bn->push_back(id);
}

// launch the factor kernels of bucket b on the index list sel (device), count entries: Jacobians into the linearization cache
void is_linearize_sel(lmgpu_isam2* S, const lmgpu_isam2::Bkt& b, const int32_t* d_sel, int count) {
  if (count <= 0) return;
  BucketDev d;
  d.type = b.type;
  d.n = count;
  d.noise_kind = b.noise_kind;
  d.vidx = b.d_vidx;
  d.meas = b.d_meas;
  d.noise = b.d_noise;
  d.J = S->pool + b.joff;
  d.epos = nullptr;
  d.robust = b.robust;
  d.rk = b.rk;
  d.sel = d_sel;
  ValuesDev vals;
  for (int t = 0; t < kNumVarTypes; t++) vals.v[t] = S->theta[t];
  const int g256 = (count + 255) / 256, g128 = (count + 127) / 128;
  hipStream_t s = S->stream;
  double* nob = nullptr;
  switch (b.type) {
    case LMGPU_F_SFM: hipLaunchKernelGGL(sfm_linearize_sel_kernel, dim3(g256), dim3(256), 0, s, d, vals); break;
    case LMGPU_F_BETWEEN_POSE2: hipLaunchKernelGGL((generic_factor_kernel<1, 3, 3, 3, 3, 0, 3, 0, 3, true>), dim3(g128), dim3(128), 0, s, d, vals, nob); break;
    case LMGPU_F_BETWEEN_POSE3: hipLaunchKernelGGL((generic_factor_kernel<2, 6, 6, 6, 12, 1, 12, 1, 12, true>), dim3(g128), dim3(128), 0, s, d, vals, nob); break;
    case LMGPU_F_PRIOR_POSE2: hipLaunchKernelGGL((generic_factor_kernel<3, 3, 3, 0, 3, 0, 3, -1, 0, true>), dim3(g128), dim3(128), 0, s, d, vals, nob); break;
    case LMGPU_F_PRIOR_POSE3: hipLaunchKernelGGL((generic_factor_kernel<4, 6, 6, 0, 12, 1, 12, -1, 0, true>), dim3(g128), dim3(128), 0, s, d, vals, nob); break;
    case LMGPU_F_PRIOR_POINT3: hipLaunchKernelGGL((generic_factor_kernel<5, 3, 3, 0, 3, 2, 3, -1, 0, true>), dim3(g128), dim3(128), 0, s, d, vals, nob); break;
    case LMGPU_F_PRIOR_CAM: hipLaunchKernelGGL((generic_factor_kernel<6, 9, 9, 0, 15, 3, 15, -1, 0, true>), dim3(g128), dim3(128), 0, s, d, vals, nob); break;
    case LMGPU_F_PROJECTION: hipLaunchKernelGGL((generic_factor_kernel<7, 2, 6, 3, 7, 1, 12, 2, 3, true>), dim3(g128), dim3(128), 0, s, d, vals, nob); break;
    case LMGPU_F_PROJECTION_BPS: hipLaunchKernelGGL((generic_factor_kernel<8, 2, 6, 3, 19, 1, 12, 2, 3, true>), dim3(g128), dim3(128), 0, s, d, vals, nob); break;
    case LMGPU_F_BEARING_RANGE_2D: hipLaunchKernelGGL((generic_factor_kernel<9, 2, 3, 2, 2, 0, 3, 4, 2, true>), dim3(g128), dim3(128), 0, s, d, vals, nob); break;
    case LMGPU_F_SFM2: hipLaunchKernelGGL(sfm2_factor_kernel<true>, dim3(g128), dim3(128), 0, s, d, vals, nob); break;
    case LMGPU_F_PRIOR_CAL3_S2: hipLaunchKernelGGL((generic_factor_kernel<11, 5, 5, 0, 5, 5, 5, -1, 0, true>), dim3(g128), dim3(128), 0, s, d, vals, nob); break;
  }
}

// several (bucket, index list) linearizations behind ONE flush and -- for the factor types of the generic kernel -- in ONE launch
int is_linearize_jobs(lmgpu_isam2* S, const std::vector<std::pair<int, std::vector<int32_t>>>& jobs) {
  if (jobs.empty()) return LMGPU_OK;
  std::vector<int32_t*> d(jobs.size(), nullptr);
  int rc;
  for (size_t q = 0; q < jobs.size(); q++)
    if (!jobs[q].second.empty() && (rc = is_stage(S, jobs[q].second, &d[q]))) return rc;
  if ((rc = is_flush(S))) return rc;
  ValuesDev vals;
  for (int t = 0; t < kNumVarTypes; t++) vals.v[t] = S->theta[t];
  MultiLin ml{};
  int blocks = 0;
  auto launch = [&]() {
    if (ml.nb == 0) return;
    ml.first[ml.nb] = blocks;
    hipLaunchKernelGGL(linearize_multi_kernel, dim3(blocks), dim3(128), 0, S->stream, ml, vals);
    ml = MultiLin{};
    blocks = 0;
  };
  for (size_t q = 0; q < jobs.size(); q++) {
    const lmgpu_isam2::Bkt& b = S->bkts[jobs[q].first];
    const int count = (int)jobs[q].second.size();
    if (count == 0) continue;
    if (b.type == LMGPU_F_SFM || b.type == LMGPU_F_SFM2) {  // kernels of their own shape
      is_linearize_sel(S, b, d[q], count);
      continue;
    }
    BucketDev bd;
    bd.type = b.type;
    bd.n = count;
    bd.noise_kind = b.noise_kind;
    bd.vidx = b.d_vidx;
    bd.meas = b.d_meas;
    bd.noise = b.d_noise;
    bd.J = S->pool + b.joff;
    bd.epos = nullptr;
    bd.robust = b.robust;
    bd.rk = b.rk;
    bd.sel = d[q];
    ml.b[ml.nb] = bd;
    ml.first[ml.nb] = blocks;
    blocks += (count + 127) / 128;
    if (++ml.nb == LIN_MULTI_MAX) launch();
  }
  launch();
  return LMGPU_OK;
}

}  // namespace

// deltaReplacedMask_ |= affected keys (ISAM2.cpp:172): marks = (offset, dimension) pairs of the re-eliminated variables
// relay != nullptr: this is the last launch of the update's elimination; one thread hands the status word of the front kernels (all
// finished: same stream) to the host's pinned word, instead of a copy command of its own
__global__ __launch_bounds__(256) void isam2_mark_kernel(const int32_t* __restrict__ marks, int n, unsigned char* __restrict__ replaced, unsigned char epoch,
                                                          const int* __restrict__ status, int* __restrict__ relay) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0 && relay) {
    __hip_atomic_store(relay, __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (i >= n) return;
  const int xo = marks[2 * i], d = marks[2 * i + 1];
  for (int k = 0; k < d; k++) replaced[xo + k] = epoch;
}

#define ISAM2_TREE_ROW 140  // ints per clique slot for its frontal / separator delta offsets (a clique has at most 139 scalar columns)
#define ISAM2_WL_GROUPS 64  // persistent workgroups of the wildfire kernel

// rewrites the descriptors of the cliques an update touched: entry i of the blob -> clique slot ids[i]
__global__ __launch_bounds__(256) void isam2_tree_patch_kernel(const int32_t* __restrict__ ids, const lmgpu::FrontDesc* __restrict__ td,
                                                                const int32_t* __restrict__ fx, const int32_t* __restrict__ sx,
                                                                const int32_t* __restrict__ kids, const int32_t* __restrict__ kids_begin,
                                                                const int32_t* __restrict__ xrow_begin, lmgpu::FrontDesc* __restrict__ tree,
                                                                int32_t* __restrict__ tree_fx, int32_t* __restrict__ tree_sx, double* __restrict__ pool) {
  const int i = blockIdx.x, id = ids[i], tid = threadIdx.x;
  const lmgpu::FrontDesc F = td[i];
  if (tid == 0) tree[id] = F;
  const int nf = F.nf, ns = F.n - F.nf - 1;
  if (F.par_ld == 0) {  // an LDS clique: its delta offsets in the fixed-stride rows
    if (tid < nf) tree_fx[(size_t)id * ISAM2_TREE_ROW + tid] = fx[(size_t)i * ISAM2_TREE_ROW + tid];
    if (tid < ns) tree_sx[(size_t)id * ISAM2_TREE_ROW + tid] = sx[(size_t)i * ISAM2_TREE_ROW + tid];
  } else {  // a wide clique: [frontal | separator] offsets in an array of its own in the pool (par_off)
    int32_t* xr = (int32_t*)(pool + F.par_off);
    for (int k = tid; k < nf + ns; k += 256) xr[k] = kids[xrow_begin[i] + k];
  }
  int32_t* dst = (int32_t*)(pool + F.child_begin);
  for (int k = tid; k < F.child_count; k += 256) dst[k] = kids[kids_begin[i] + k];
}

// ISAM2::updateDelta's top-down walk as a device-side WORK LIST (optimizeWildfireNonRecursive, gtsam/nonlinear/ISAM2-impl.cpp:47-77 ->
// ISAM2Clique::optimizeWildfireNode, ISAM2Clique.cpp:211-234): the queue starts with the roots; a persistent workgroup takes the next
// ticket, waits for that queue entry, processes the clique and -- if the clique was dirty, exactly the reference's rule for descending
// -- appends its children.  Work is O(cliques visited), not O(cliques): the first single-launch form had one workgroup per clique of the
// tree, whatever had changed (0.25 ms per update after 500 poses of city10000, 0.40 ms after 3 000).
//   dirty   = the clique was re-eliminated (replaced flag of its first frontal scalar) or a separator scalar changed (isDirty :56-77)
//   solve   = x_F = R^-1 (d - S x_S)   ([R S d] staged in LDS, the register solve of the batch path's LDS fronts: ldsb_stage / ldsb_solve_core)
//   keep    = replaced or max |x_F_old - x_F_new| >= threshold (valuesChanged :151-158): write x_F, flag the frontal scalars as changed;
//             otherwise the old values stay (restoreFromOriginals).   threshold <= 0: every clique is solved (full back-substitution).
// Progress: tickets and queue slots are both handed out in increasing order, an entry is published by a clique that is being processed
// (counted in wl[2]), and a waiting workgroup leaves when its entry is empty AND wl[2] == 0.  Every spin is bounded.
// wl: [0] tail (next free queue slot), [1] next ticket, [2] published and unfinished items; the host presets them and the roots.
__global__ __launch_bounds__(256) void isam2_wildfire_kernel(long long* __restrict__ queue, unsigned int* __restrict__ wl,
                                                              const lmgpu::FrontDesc* __restrict__ tree, const int32_t* __restrict__ tree_fx,
                                                              const int32_t* __restrict__ tree_sx, const double* __restrict__ pool,
                                                              double* __restrict__ delta, const unsigned char* __restrict__ replaced,
                                                              unsigned char* __restrict__ changed, double threshold, int* __restrict__ status,
                                                              unsigned char epoch, int* __restrict__ relay, unsigned char* __restrict__ done,
                                                              int by_value, double* __restrict__ mirror) {
  // mirror: (or null) the host's pinned copy of delta, kept equal to it: every x the walk keeps is stored there as well
  // by_value: the re-eliminated top of the tree (every clique reached from a root through replaced cliques; the host filled their frontal
  // scalars of delta with the all-ones pattern in the flush in front of this launch) hands x over BY VALUE, as the merged back-substitution
  // of the batch path does: the parent stores x_F with agent-scope stores right after its solve, the child polls its separator scalars.
  // No done flag, no fence and no second round trip between two such cliques (a fixed-lag smoother re-eliminates its whole chain of ~50
  // cliques every update: 5.5 us per clique before).  Bit 63 of a queue entry = "the parent is part of that top".
  extern __shared__ double Ls[];
  __shared__ int s_id, s_par, s_pare, flag;
  __shared__ double red[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (;;) {
    if (tid == 0) {
      const unsigned int my = atomicAdd(&wl[1], 1u);
      int id, par = -1, pare = 0;
      long spins = 0;
      for (;;) {
        const long long ent = __hip_atomic_load(&queue[my], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        id = (int)(ent & 0xffffffffLL);
        par = (int)((ent >> 32) & 0x7fffffffLL);
        pare = (int)((unsigned long long)ent >> 63);
        if (ent != -1LL) break;
        id = -1;
        if (__hip_atomic_load(&wl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          id = -2;  // nothing is being processed and nothing is queued: the walk is over
          break;
        }
        __builtin_amdgcn_s_sleep(2);
        if (++spins > 4000000L) {
          id = -3;
          break;
        }
      }
      if (id >= 0) __hip_atomic_store(&queue[my], -1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the slot is clean for the next launch
      s_id = id;
      s_par = par == 0x7fffffff ? -1 : par;
      s_pare = pare;
    }
    __syncthreads();
    const int id = s_id;
    if (id < 0) {
      if (tid == 0) {
        if (id == -3) atomicMin(status, -1);  // never expected: spin bound hit (reported as a fault by the host)
        // the last workgroup out hands the status word to the host's pinned word (wl[3] counts the leavers; the host resets it per launch)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (relay && atomicAdd(&wl[3], 1u) == gridDim.x - 1) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          __hip_atomic_store(relay, __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      return;
    }
    const lmgpu::FrontDesc F = tree[id];
    const int n = F.n, nf = F.nf, ns = n - nf - 1;
    const bool wide = F.par_ld != 0;  // more than 139 scalar columns: offsets in the pool, [R S d] walked from memory
    const int32_t* fxr = wide ? (const int32_t*)(pool + F.par_off) : tree_fx + (size_t)id * ISAM2_TREE_ROW;
    const int32_t* sxr = wide ? fxr + nf : tree_sx + (size_t)id * ISAM2_TREE_ROW;
    const int so = ns > 0 ? sxr[min(tid, ns - 1)] : 0, fo = fxr[min(tid, nf - 1)];
    const bool is_replaced = replaced[fxr[0]] == epoch;
    // A clique is queued by its parent as soon as the parent knows it is dirty, BEFORE the parent solves: the child's descriptor chain
    // (queue -> clique -> offsets -> flags) and the staging of its [R S d] run under the parent's solve; what depends on the parent -- the
    // changed flags of the separator, x_S -- waits for the parent's done flag (the epoch of this walk).  The hand-off between two levels was
    // ten dependent round trips (7.8 us per level of a chain of small cliques); now the parent's publish and the child's x_S.
    const int par = s_par;
    const bool in_top = by_value && is_replaced && !wide && (par < 0 || s_pare);  // workgroup-uniform
    if (!wide) lmgpu::ldsb_stage(F, pool, Ls, tid);
    if (tid == 0) {
      int ok = 1;
      if (par >= 0 && !in_top) {
        long spins = 0;
        while (__hip_atomic_load(&done[par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > 4000000L) {
            ok = 0;
            break;
          }
        }
      }
      if (!ok) atomicMin(status, -1);  // never expected: spin bound hit (reported as a fault by the host)
      flag = (threshold <= 0.0 || is_replaced) ? 1 : 0;
    }
    __syncthreads();
    if (!in_top) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the parent's delta / changed flags
    const double xs_pre = in_top ? 0.0 : delta[ns > 0 ? so : fo];  // (in flight together with the changed flags below: one round trip, not two)
    if (!(threshold <= 0.0 || is_replaced)) {
      for (int j = tid; j < ns; j += 256)
        if (changed[sxr[j]] == epoch) flag = 1;  // benign race: every writer stores 1
      __syncthreads();
    }
    const bool dirty = flag != 0;  // workgroup-uniform
    // the children (a dirty clique's only) are queued now, with this clique as the parent they wait for
    const int nk = dirty ? F.child_count : 0;
    if (nk > 0) {
      if (tid == 0) {
        atomicAdd(&wl[2], (unsigned int)nk);
        s_id = (int)atomicAdd(&wl[0], (unsigned int)nk);
      }
      __syncthreads();
      const int32_t* kids = (const int32_t*)(pool + F.child_begin);
      for (int k = tid; k < nk; k += 256)
        __hip_atomic_store(&queue[(unsigned int)s_id + k],
                           (long long)((unsigned long long)(unsigned int)kids[k] | (unsigned long long)(unsigned int)id << 32 | (in_top ? 1ull << 63 : 0ull)),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (dirty && wide) {
      // A clique too wide for the LDS staging (rare: loop closures of large graphs): x_S and y in LDS, R / S streamed from memory by this
      // one workgroup -- y = d - S x_S one wave per row, then 64 unknowns at a time (the diagonal block staged in LDS, the readlane
      // chain of the LDS path), the rows above each block folded by all waves.
      double* xs = Ls;               // [ns]
      double* y = Ls + ns;           // [nf]
      double* Db = Ls + ns + nf;     // [64][65]
      const double* A = pool + F.rsd_off;
      const int ld = F.ld_rsd;
      for (int j = tid; j < ns; j += 256) xs[j] = delta[sxr[j]];
      __syncthreads();
      for (int i = w; i < nf; i += 4) {
        const double* row = A + (size_t)i * ld;
        double a = 0.0;
        for (int j = lane; j < ns; j += 64) a += row[nf + j] * xs[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0) y[i] = row[n - 1] - a;
      }
      __syncthreads();
      bool bad = false;
      for (int b = (nf + 63) / 64 - 1; b >= 0; b--) {
        const int r0 = 64 * b, nb = min(64, nf - r0);
        for (int idx = tid; idx < 64 * 64; idx += 256) {
          const int p = idx >> 6, q = idx & 63;
          const double v = A[(size_t)(r0 + min(p, nb - 1)) * ld + r0 + min(q, nb - 1)];
          Db[p * 65 + q] = (p < nb && q < nb && q >= p) ? v : ((p == q) ? 1.0 : 0.0);
        }
        __syncthreads();
        if (w == 0) {
          const double rd = 1.0 / Db[lane * 65 + lane];
          double yi = (lane < nb) ? y[r0 + lane] * rd : 0.0;
          for (int k0 = 63; k0 >= 0; k0 -= 8) {
            double cf[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
              const double c = Db[lane * 65 + k0 - u];
              cf[u] = (lane < k0 - u) ? c * rd : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) yi = fma(-cf[u], lmgpu::readlane_dyn(yi, k0 - u), yi);
          }
          if (lane < nb) {
            y[r0 + lane] = yi;
            if (yi != yi) bad = true;
          }
        }
        __syncthreads();
        for (int i = w; i < r0; i += 4) {
          const double a0 = (lane < nb) ? A[(size_t)i * ld + r0 + lane] * y[r0 + lane] : 0.0;
          double a = a0;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
          if (lane == 0) y[i] -= a;
        }
        __syncthreads();
      }
      double md = 0.0;
      for (int i = tid; i < nf; i += 256) md = fmax(md, fabs(delta[fxr[i]] - y[i]));
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) md = fmax(md, __shfl_xor(md, o));
      if (lane == 0) red[w] = md;
      __syncthreads();
      const double mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
      if (bad && lane == 0) atomicMin(status, F.id);
      const bool keep = threshold <= 0.0 || is_replaced || mx >= threshold;
      if (keep)
        for (int i = tid; i < nf; i += 256) {
          const int xo = fxr[i];
          delta[xo] = y[i];
          changed[xo] = epoch;
          if (mirror) mirror[xo] = y[i];
        }
    } else if (in_top) {
      bool bad, timed_out = false;
      const double* x = lmgpu::ldsb_solve_core<true>(F, Ls, ns > 0 ? so : fo, delta, &bad, &timed_out);
      if (tid < nf) {
        const double xv = x[tid];
        __hip_atomic_store(&delta[fo], (xv != xv) ? __longlong_as_double(0x7ff8000000000000LL) : xv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // never the sentinel
        changed[fo] = epoch;
        if (mirror) mirror[fo] = xv;
      }
      if (timed_out) atomicMin(status, -1);  // never expected: spin bound hit (reported as a fault by the host)
      if (bad && lane == 0) atomicMin(status, F.id);  // NaN: IndeterminantLinearSystemException (ISAM2Clique.cpp:124-126)
    } else if (dirty) {
      bool bad;
      const double* x = lmgpu::ldsb_solve_core(F, Ls, ns > 0 ? so : fo, delta, &bad, nullptr, &xs_pre);
      double md = 0.0;
      if (tid < nf) md = fabs(delta[fo] - x[tid]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) md = fmax(md, __shfl_xor(md, o));
      if (lane == 0) red[w] = md;
      __syncthreads();
      const double mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
      if (bad && lane == 0) atomicMin(status, F.id);  // NaN: IndeterminantLinearSystemException (ISAM2Clique.cpp:124-126)
      const bool keep = threshold <= 0.0 || is_replaced || mx >= threshold;
      if (keep && tid < nf) {
        delta[fo] = x[tid];
        changed[fo] = epoch;
        if (mirror) mirror[fo] = x[tid];
      }
    }
    // done: every wave's stores have been performed, then one release + the flag the children wait for
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&done[id], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) atomicSub(&wl[2], 1u);  // this item is finished (its children, if any, are counted already)
    __syncthreads();
  }
}


// ISAM2::marginalCovariance(key) (gtsam/nonlinear/ISAM2.h:253-257 -> BayesTree::marginalFactor(key)->information().inverse()):
// column c of Sigma = (R^T R)^-1 restricted to the variable is x with R^T y = e_c, R x = y.  In the Bayes tree both solves only touch the
// PATH from the clique of the variable to its root: y is zero below it, and x is only wanted at the variable.  One wave per column, no
// hand-offs: forward (R^T) up the path, back-substitution down again; w = one work vector per column (zeroed by the host), laid out like delta.
// path[0] = the clique the variable is frontal in ... path[npath - 1] = its root.
__global__ __launch_bounds__(64) void isam2_marginal_kernel(const int32_t* __restrict__ path, int npath, const lmgpu::FrontDesc* __restrict__ tree,
                                                             const int32_t* __restrict__ tree_fx, const int32_t* __restrict__ tree_sx,
                                                             const double* __restrict__ pool, double* W, int ntot, int xoff, int dim,
                                                             double* __restrict__ out, int* __restrict__ status) {
  extern __shared__ double v[];  // the clique's slice of the work vector: [nf frontal | ns separator]
  const int lane = threadIdx.x, c = blockIdx.x;
  double* w = W + (size_t)c * ntot;
  auto ld_w = [&](int i) { return __hip_atomic_load(&w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto st_w = [&](int i, double x) { __hip_atomic_store(&w[i], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto wave_sync = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  if (lane == 0) st_w(xoff + c, 1.0);
  wave_sync();
  for (int dir = 0; dir < 2; dir++)
    for (int q = 0; q < npath; q++) {
      const int p = dir == 0 ? q : npath - 1 - q;
      const lmgpu::FrontDesc F = tree[path[p]];
      const int n = F.n, nf = F.nf, ns = n - nf - 1, ld = F.ld_rsd;
      const bool wide = F.par_ld != 0;
      const int32_t* fxr = wide ? (const int32_t*)(pool + F.par_off) : tree_fx + (size_t)path[p] * ISAM2_TREE_ROW;
      const int32_t* sxr = wide ? fxr + nf : tree_sx + (size_t)path[p] * ISAM2_TREE_ROW;
      const double* A = pool + F.rsd_off;
      for (int i = lane; i < nf; i += 64) v[i] = ld_w(fxr[i]);
      if (dir == 1)
        for (int j = lane; j < ns; j += 64) v[nf + j] = ld_w(sxr[j]);
      wave_sync();
      if (dir == 0) {
        // R^T y_F = b_F, then b_S -= S^T y_F
        for (int i = 0; i < nf; i++) {
          double a = 0.0;
          for (int k = lane; k < i; k += 64) a += A[(size_t)k * ld + i] * v[k];
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
          if (lane == 0) v[i] = (v[i] - a) / A[(size_t)i * ld + i];
          wave_sync();
        }
        for (int j = lane; j < ns; j += 64) {
          double a = 0.0;
          for (int k = 0; k < nf; k++) a += A[(size_t)k * ld + nf + j] * v[k];
          st_w(sxr[j], ld_w(sxr[j]) - a);
        }
      } else {
        // x_F = R^-1 (y_F - S x_S)
        for (int i = nf - 1; i >= 0; i--) {
          const double* row = A + (size_t)i * ld;
          double a = 0.0;
          for (int j = i + 1 + lane; j < n - 1; j += 64) a += row[j] * v[j];
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
          if (lane == 0) v[i] = (v[i] - a) / row[i];
          wave_sync();
        }
      }
      for (int i = lane; i < nf; i += 64) st_w(fxr[i], v[i]);
      wave_sync();
    }
  if (lane < dim) {
    const double x = ld_w(xoff + lane);
    out[c * dim + lane] = x;
    if (!(fabs(x) < 1.7e308)) atomicMin(status, 0);  // NaN / inf: a singular clique on the path
  }
}

// ---- the Bayes tree as a Gaussian factor graph of unit-noise factors [R S | d], one per clique (GaussianBayesTree): the three products
//      Powell's dog leg needs (ISAM2.cpp:739-779), on the device copy of the tree.  One wave per clique, four per workgroup.
//      Columns j >= i only in the R part (what lies below the diagonal of a wide clique's rows is not part of R).
__device__ __forceinline__ void isam2_clique_rows(const lmgpu::FrontDesc& F, int id, const int32_t* tree_fx, const int32_t* tree_sx, const double* pool,
                                                  const int32_t** fxr, const int32_t** sxr) {
  const bool wide = F.par_ld != 0;
  *fxr = wide ? (const int32_t*)(pool + F.par_off) : tree_fx + (size_t)id * ISAM2_TREE_ROW;
  *sxr = wide ? *fxr + F.nf : tree_sx + (size_t)id * ISAM2_TREE_ROW;
}
// g -= [R S]^T d  (ISAM2::gradientAtZero, ISAM2.cpp:825-833: the sum of the cliques' gradient contributions, ISAM2Clique.cpp:35-46)
__global__ __launch_bounds__(256) void isam2_tree_gradient_kernel(const int32_t* __restrict__ list, int nlist, const lmgpu::FrontDesc* __restrict__ tree,
                                                                  const int32_t* __restrict__ tree_fx, const int32_t* __restrict__ tree_sx,
                                                                  const double* __restrict__ pool, const int32_t* __restrict__ part_off,
                                                                  double* __restrict__ part) {
  // every clique writes the terms of its columns into slots of their own; isam2_gradient_gather_kernel adds, per scalar, the slots the
  // host listed for it in a fixed order (FP64 atomics before round 3: the dog leg's trust-region decisions could differ in the last bit)
  const int li = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (li >= nlist) return;
  const int id = list[li];
  const lmgpu::FrontDesc F = tree[id];
  const int n = F.n, nf = F.nf, ld = F.ld_rsd;
  const double* A = pool + F.rsd_off;
  double* out = part + part_off[li];
  for (int j = lane; j < n - 1; j += 64) {
    double s = 0.0;
    const int imax = j < nf ? j : nf - 1;
    for (int i = 0; i <= imax; i++) s += A[(size_t)i * ld + j] * A[(size_t)i * ld + n - 1];
    out[j] = -s;
  }
}
__global__ __launch_bounds__(256) void isam2_gradient_gather_kernel(const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx, const double* __restrict__ part,
                                                                     int n, double* __restrict__ g) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= n) return;
  double s = 0;
  for (int e = ptr[x]; e < ptr[x + 1]; e++) s += part[idx[e]];
  g[x] = s;
}
// RgProd_F = R g_F + S g_S  (UpdateRgProd, ISAM2-impl.cpp:82-141; recomputed for every clique: below a clique without a replaced key
// neither its rows nor the gradient on its keys have changed, so the reference's walk would have left the same numbers)
__global__ __launch_bounds__(256) void isam2_tree_rg_kernel(const int32_t* __restrict__ list, int nlist, const lmgpu::FrontDesc* __restrict__ tree,
                                                            const int32_t* __restrict__ tree_fx, const int32_t* __restrict__ tree_sx,
                                                            const double* __restrict__ pool, const double* __restrict__ g, double* __restrict__ rg) {
  const int li = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (li >= nlist) return;
  const int id = list[li];
  const lmgpu::FrontDesc F = tree[id];
  const int n = F.n, nf = F.nf, ld = F.ld_rsd;
  const int32_t *fxr, *sxr;
  isam2_clique_rows(F, id, tree_fx, tree_sx, pool, &fxr, &sxr);
  const double* A = pool + F.rsd_off;
  for (int i = 0; i < nf; i++) {
    double s = 0.0;
    for (int j = i + lane; j < n - 1; j += 64) s += A[(size_t)i * ld + j] * g[j < nf ? fxr[j] : sxr[j - nf]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) rg[fxr[i]] = s;
  }
}
// out[li] = || [R S] x - d ||^2 of clique li (ISAM2::error(x) = GaussianFactorGraph(*this).error(x), ISAM2.cpp:820-823); x == nullptr: x = 0
__global__ __launch_bounds__(256) void isam2_tree_error_kernel(const int32_t* __restrict__ list, int nlist, const lmgpu::FrontDesc* __restrict__ tree,
                                                               const int32_t* __restrict__ tree_fx, const int32_t* __restrict__ tree_sx,
                                                               const double* __restrict__ pool, const double* __restrict__ x, double* __restrict__ out) {
  const int li = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (li >= nlist) return;
  const int id = list[li];
  const lmgpu::FrontDesc F = tree[id];
  const int n = F.n, nf = F.nf, ld = F.ld_rsd;
  const int32_t *fxr, *sxr;
  isam2_clique_rows(F, id, tree_fx, tree_sx, pool, &fxr, &sxr);
  const double* A = pool + F.rsd_off;
  double tot = 0.0;
  for (int i = 0; i < nf; i++) {
    double s = 0.0;
    if (x)
      for (int j = i + lane; j < n - 1; j += 64) s += A[(size_t)i * ld + j] * x[j < nf ? fxr[j] : sxr[j - nf]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const double e = s - A[(size_t)i * ld + n - 1];
    tot += e * e;
  }
  if (lane == 0) out[li] = tot;
}
// out[slot] = sum_i a[i] b[i], one workgroup, fixed order
__global__ __launch_bounds__(1024) void isam2_dot_kernel(const double* __restrict__ a, const double* __restrict__ b, int n, double* __restrict__ out) {
  __shared__ double red[1024];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) s += a[i] * b[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = red[0];
}
// dx_u = -(g.g / Rg.Rg) g  (ComputeGradientSearch, ISAM2-impl.cpp:144-157); scal[0] = g.g, scal[1] = Rg.Rg
__global__ __launch_bounds__(256) void isam2_gradient_search_kernel(const double* __restrict__ g, const double* __restrict__ scal, int n, double* __restrict__ dx_u) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dx_u[i] = -(scal[0] / scal[1]) * g[i];
}
// dx_d = a dx_u + b dx_n  (ComputeDoglegPoint / ComputeBlend, DoglegOptimizerImpl.cpp:26-91)
__global__ __launch_bounds__(256) void isam2_blend_kernel(const double* __restrict__ dx_u, const double* __restrict__ dx_n, double a, double b, int n,
                                                          double* __restrict__ dx_d) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dx_d[i] = a * dx_u[i] + b * dx_n[i];
}

namespace {

// bring the device copy of the tree up to date: only the cliques the updates since the last call created or re-parented
int is_patch_tree(lmgpu_isam2* S) {
  const size_t NC = S->clq.size();
  int rc;
  if (NC > S->tree_slots) {  // more clique slots: the arrays grow, their contents are kept
    const size_t cap = is_next_cap(S->tree_slots, NC);
    if ((rc = is_realloc(S, &S->d_tree, cap, S->tree_slots))) return rc;
    if ((rc = is_realloc(S, &S->d_tree_fx, cap * ISAM2_TREE_ROW, S->tree_slots * ISAM2_TREE_ROW))) return rc;
    if ((rc = is_realloc(S, &S->d_tree_sx, cap * ISAM2_TREE_ROW, S->tree_slots * ISAM2_TREE_ROW))) return rc;
    if ((rc = is_realloc(S, &S->d_tree_done, cap, 0))) return rc;
    ISCHECK(hipMemsetAsync(S->d_tree_done, 0, cap, S->stream));  // (epochs: a fresh array must not hold the current one)
    S->tree_slots = cap;
  }
  if (NC + ISAM2_WL_GROUPS + 1 > S->queue_cap) {  // every clique once + the tickets of the workgroups that find nothing
    const size_t cap = is_next_cap(S->queue_cap, NC + ISAM2_WL_GROUPS + 1);
    if ((rc = is_realloc(S, &S->d_queue, cap, 0))) return rc;
    ISCHECK(hipMemsetAsync(S->d_queue, 0xff, cap * sizeof(long long), S->stream));  // all slots "not published"; consumers keep it that way
    S->queue_cap = cap;
  }
  if (!S->d_wl) ISCHECK(hipMalloc((void**)&S->d_wl, 4 * sizeof(unsigned int)));
  std::sort(S->touched.begin(), S->touched.end());
  S->touched.erase(std::unique(S->touched.begin(), S->touched.end()), S->touched.end());
  // the descriptors of the touched cliques travel as pushes of the next flush (the scatter kernel that also seeds the walk): the descriptor,
  // the delta offsets of its frontal and separator scalars (fixed-stride rows; a wide clique: an array of its own in the pool), its children
  std::vector<int32_t> row;
  for (int32_t id : S->touched) {
    lmgpu_isam2::Clq& c = S->clq[id];
    if (!c.alive) continue;
    if (c.kids_off >= 0 && c.kids_n != (int)c.children.size()) {
      is_pool_free(S, c.kids_off, (size_t)(c.kids_n + 1) / 2);
      c.kids_off = -1;
    }
    c.kids_n = (int)c.children.size();
    if (c.kids_n > 0 && c.kids_off < 0 && (rc = is_pool_alloc(S, (size_t)(c.kids_n + 1) / 2, &c.kids_off))) return rc;
    if (c.ld > 0 && c.xrow_off < 0 && (rc = is_pool_alloc(S, (size_t)c.n / 2 + 1, &c.xrow_off))) return rc;
  }
  for (int32_t id : S->touched) {  // (second pass: the pool may have moved while the first one allocated)
    lmgpu_isam2::Clq& c = S->clq[id];
    if (!c.alive) continue;
    if (c.kids_off > (int64_t)INT32_MAX) {
      S->err = "ISAM2: pool offset beyond the range of a tree descriptor";
      return LMGPU_INVALID;
    }
    FrontDesc F{};
    F.n = c.n;
    F.nf = c.nf;
    F.rsd_off = c.rsd_off;
    F.ld_rsd = c.ld > 0 ? c.ld : c.n;
    F.id = id;
    F.child_begin = c.kids_n > 0 ? (int32_t)c.kids_off : 0;
    F.child_count = c.kids_n;
    if (c.ld > 0)  // walked from memory by one workgroup (isam2_wildfire_kernel): x_S, y and one 64 x 64 diagonal block in LDS
      S->tree_lds = std::max(S->tree_lds, (size_t)c.n + 64 * 65 + 64);
    else
      S->tree_lds = std::max(S->tree_lds, (size_t)c.nf * (size_t)(c.n | 1));
    if (c.ld == 0) {
      row.clear();
      for (int k = 0; k < c.nfv; k++)
        for (int d = 0; d < kVarDim[S->vars[c.vars[k]].type]; d++) row.push_back(S->vars[c.vars[k]].xoff + d);
      if ((rc = is_push(S, S->d_tree_fx + (size_t)id * ISAM2_TREE_ROW, row.data(), row.size() * sizeof(int32_t)))) return rc;
      row.clear();
      for (size_t k = c.nfv; k < c.vars.size(); k++)
        for (int d = 0; d < kVarDim[S->vars[c.vars[k]].type]; d++) row.push_back(S->vars[c.vars[k]].xoff + d);
      if ((rc = is_push(S, S->d_tree_sx + (size_t)id * ISAM2_TREE_ROW, row.data(), row.size() * sizeof(int32_t)))) return rc;
    } else {
      F.par_off = c.xrow_off;
      F.par_ld = 1;
      row.clear();
      for (size_t k = 0; k < c.vars.size(); k++)
        for (int d = 0; d < kVarDim[S->vars[c.vars[k]].type]; d++) row.push_back(S->vars[c.vars[k]].xoff + d);
      if ((rc = is_push(S, S->pool + c.xrow_off, row.data(), row.size() * sizeof(int32_t)))) return rc;
    }
    if (c.kids_n > 0 && (rc = is_push(S, S->pool + c.kids_off, c.children.data(), c.children.size() * sizeof(int32_t)))) return rc;
    if ((rc = is_push(S, S->d_tree + id, &F, sizeof(F)))) return rc;
  }
  S->touched.clear();
  return LMGPU_OK;
}

// ISAM2::updateDelta (gtsam/nonlinear/ISAM2.cpp:701-719) -> DeltaImpl::UpdateGaussNewtonDelta (ISAM2-impl.cpp:47-77)
// In two halves so that a caller can queue its own reads of delta behind the walk and wait ONCE: is_update_delta_enqueue launches,
// is_update_delta_finish waits and looks at the status word.
// host_delta: also bring delta to the pinned host copy (CheckRelinearizationFull reads it; the estimate readers do not need it)
int is_update_delta_dogleg(lmgpu_isam2* S, bool force_full, bool host_delta);
static int is_walk_by_value(const lmgpu_isam2* S) { return (!S->dogleg && !dev_switch("LMGPU_ISAM2_NO_BYVALUE")) ? 1 : 0; }
// the pushes a walk consumes (see walk_prepared): the tree patch, and for a non-empty tree the seeds, the counters, the status word and
// (by_value) the all-ones pattern over the frontal scalars of the re-eliminated top
int is_walk_prepare(lmgpu_isam2* S, int by_value) {
  int rc = is_patch_tree(S);
  if (rc) return rc;
  S->walk_prepared = false;
  if (S->ntot == 0 || S->roots.empty()) return LMGPU_OK;
  // the walk starts at every root (ISAM2-impl.cpp:60-66): queue[0 .. r) = roots, tail = r, next ticket = 0, unfinished = r, leavers = 0
  const unsigned int r = (unsigned int)S->roots.size();
  const unsigned int ctl[4] = {r, 0u, r, 0u};
  const int32_t fresh = 0x7f7f7f7f;
  std::vector<long long> seeds(std::max(r, S->seeded), -1LL);  // (slots of an earlier seeding no walk consumed go back to "empty")
  for (unsigned int q = 0; q < r; q++) seeds[q] = (long long)(unsigned int)S->roots[q] | (long long)0x7fffffff << 32;  // (no parent)
  if (by_value) {
    if (S->delta_zero_pending && (rc = is_flush(S))) return rc;
    std::vector<int32_t> stack;
    std::vector<std::pair<int32_t, int32_t>> runs;  // (xoff, scalars)
    for (int32_t rt : S->roots) stack.push_back(rt);
    while (!stack.empty()) {
      const lmgpu_isam2::Clq& c = S->clq[stack.back()];
      stack.pop_back();
      if (c.ld > 0 || !S->replaced[c.vars[0]]) continue;  // (the kernel's rule: a wide clique and everything below it wait for done flags)
      for (int k = 0; k < c.nfv; k++) runs.emplace_back(S->vars[c.vars[k]].xoff, kVarDim[S->vars[c.vars[k]].type]);
      for (int32_t ch : c.children) stack.push_back(ch);
    }
    std::sort(runs.begin(), runs.end());
    for (size_t a = 0; a < runs.size();) {
      size_t b = a + 1;
      int32_t end = runs[a].first + runs[a].second;
      while (b < runs.size() && runs[b].first == end) end += runs[b++].second;
      S->pushes.push_back(lmgpu_isam2::PushRec{S->delta + runs[a].first, nullptr, (uint32_t)(2 * (end - runs[a].first)), 1u});
      a = b;
    }
  }
  if ((rc = is_push(S, S->d_queue, seeds.data(), seeds.size() * sizeof(long long)))) return rc;
  if ((rc = is_push(S, S->d_wl, ctl, sizeof(ctl)))) return rc;
  if ((rc = is_push(S, S->d_status + 1, &fresh, sizeof(fresh)))) return rc;  // (word 1: the walk's own, an elimination in front of it uses word 0)
  S->seeded = r;
  S->walk_prepared = true;
  return LMGPU_OK;
}
int is_update_delta_enqueue(lmgpu_isam2* S, bool force_full, bool host_delta, double* target = nullptr) {
  if (S->dogleg && !target) return is_update_delta_dogleg(S, force_full, host_delta);  // (waits itself; _finish then finds an idle stream)
  const int by_value = target ? 0 : is_walk_by_value(S);
  int rc = S->walk_prepared ? is_patch_tree(S) : is_walk_prepare(S, by_value);  // (prepared by the update: only what touched the tree since)
  if (rc) return rc;
  S->walk_prepared = false;
  S->h_status[1] = 0x7f7f7f7f;
  if (S->ntot == 0) return LMGPU_OK;
  const double thr = force_full ? 0.0 : (S->dogleg ? S->dogleg_wildfire : S->prm.wildfireThreshold);
  double* const wf_delta = target ? target : S->delta;
  // the host's mirror of delta (see h_delta): the scalars added since the last walk are zero on both sides (delta_.insert(zeroVectors))
  double* mirror = nullptr;
  if (!S->dogleg && !target) {
    if ((size_t)S->ntot > S->h_delta_cap) {
      if (S->h_delta) {
        ISCHECK(hipStreamSynchronize(S->stream));
        (void)hipHostFree(S->h_delta);
      }
      S->h_delta = nullptr;
      S->mirror_ntot = 0;
      S->h_delta_cap = is_next_cap(S->h_delta_cap, (size_t)S->ntot);
      ISCHECK(hipHostMalloc((void**)&S->h_delta, S->h_delta_cap * sizeof(double), hipHostMallocMapped));
      ISCHECK(hipHostGetDevicePointer((void**)&S->h_delta_dev, S->h_delta, 0));
    }
    if (dev_switch("LMGPU_ISAM2_NO_MIRROR")) S->mirror_ntot = 0;
    if (S->mirror_ntot > 0) {
      for (size_t i = S->mirror_ntot; i < (size_t)S->ntot; i++) S->h_delta[i] = 0.0;
      S->mirror_ntot = (size_t)S->ntot;
      mirror = S->h_delta_dev;
    }
  }
  // No clears and no status copy: d_changed / d_replaced hold the EPOCH of the walk that set them (a new epoch per walk; both arrays
  // are cleared only when the 8-bit epoch wraps), the status word is reset by the same scatter kernel that seeds the work list, and the
  // last workgroup to leave the walk stores it into the host's pinned word.  (Each was a device operation of its own: ~6 us of stream time.)
  if (!S->roots.empty()) {
    if ((rc = is_flush(S))) return rc;
    hipLaunchKernelGGL(isam2_wildfire_kernel, dim3(ISAM2_WL_GROUPS), dim3(256), (S->tree_lds + LDSB_TAIL) * sizeof(double), S->stream, S->d_queue, S->d_wl,
                       (const FrontDesc*)S->d_tree, (const int32_t*)S->d_tree_fx, (const int32_t*)S->d_tree_sx, (const double*)S->pool, wf_delta,
                       (const unsigned char*)S->d_replaced, S->d_changed, thr, S->d_status + 1, S->epoch, S->h_status_dev + 1, S->d_tree_done, by_value, mirror);
    ISCHECK(hipGetLastError());
  }
  if (++S->epoch == 0) {  // wrapped: start over from clean arrays
    S->epoch = 1;
    ISCHECK(hipMemsetAsync(S->d_changed, 0, (size_t)S->ntot_cap, S->stream));
    ISCHECK(hipMemsetAsync(S->d_replaced, 0, (size_t)S->ntot_cap, S->stream));
    if (S->d_tree_done) ISCHECK(hipMemsetAsync(S->d_tree_done, 0, S->tree_slots, S->stream));
  }
  if (host_delta && !target && !mirror) {  // CheckRelinearizationFull reads it; from here on the walk keeps it current
    ISCHECK(hipMemcpyAsync(S->h_delta, S->delta, (size_t)S->ntot * sizeof(double), hipMemcpyDeviceToHost, S->stream));
    S->mirror_ntot = (size_t)S->ntot;
  }
  return LMGPU_OK;
}
int is_update_delta_finish(lmgpu_isam2* S) {
  ISCHECK(hipStreamSynchronize(S->stream));
  std::fill(S->replaced.begin(), S->replaced.end(), 0);
  S->any_replaced = false;
  if (S->h_status[1] < 0) {
    S->err = "ISAM2 back-substitution: a parent-to-child hand-off timed out";
    return LMGPU_HIP_ERROR;
  }
  if (S->h_status[1] < (int)S->clq.size()) {
    S->failed_key = S->vars[S->clq[S->h_status[1]].vars[0]].key;
    S->err = "indeterminate linear system in back-substitution";
    return LMGPU_INDETERMINATE;
  }
  return LMGPU_OK;
}
int is_update_delta(lmgpu_isam2* S, bool force_full, bool host_delta = false) {
  const int rc = is_update_delta_enqueue(S, force_full, host_delta);
  return rc ? rc : is_update_delta_finish(S);
}

// Ordering::ColamdConstrained (gtsam/inference/Ordering.cpp:50-125, 193-210) on a VariableIndex given as (variables ascending by
// key, their factor lists); groups: vid -> group.  Returns the elimination order as positions into `vids`.
int is_colamd(lmgpu_isam2* S, const std::vector<int32_t>& vids, const std::vector<std::vector<int32_t>>& cols, int n_factors,
              const std::map<int32_t, int>& groups, std::vector<int32_t>* perm) {
  const int nVars = (int)vids.size();
  perm->assign(nVars, 0);
  if (nVars <= 1) return LMGPU_OK;
  std::vector<int32_t> p(nVars + 1, 0), A, cmember(nVars, 0);
  for (int j = 0; j < nVars; j++) {
    for (int32_t f : cols[j]) A.push_back(f);
    p[j + 1] = (int32_t)A.size();
    auto g = groups.find(vids[j]);
    if (g != groups.end()) cmember[j] = g->second;
  }
  if (!S->ccolamd || S->ccolamd(S->user, n_factors, nVars, p.data(), A.data(), cmember.data(), perm->data()) != 1) {
    S->err = "the ccolamd callback failed";
    return LMGPU_INVALID;
  }
  return LMGPU_OK;
}

// one entry of the linear graph handed to the partial elimination
struct IsGF {
  int kind;  // 0: linear factor of nonlinear factor `id`; 1: cached boundary factor of orphan clique `id`; 2: the orphan subtree itself;
             // 3: the marginal factor in slot `id` of the factor list
  int32_t id;
  std::vector<int32_t> vids;
};

int is_eliminate_fronts(lmgpu_isam2* S, const std::vector<IsGF>& gfs, const std::vector<int32_t>& vid_of_slot, const SymbolicFronts& sf, bool attach,
                        std::vector<int>* cid_out);
// eliminate `gfs` over the variables `vids` (ascending by key) in the order `perm`: new cliques on the device, attached to the tree
// var_cols (optional, batch): per variable of `vids` its factor list as the variable index keeps it (the elimination tree follows that order)
int is_eliminate(lmgpu_isam2* S, const std::vector<IsGF>& gfs, const std::vector<int32_t>& vids, const std::vector<int32_t>& perm,
                 const std::vector<std::vector<int32_t>>* var_cols = nullptr) {
  const int n = (int)vids.size();
  if (n == 0) return LMGPU_OK;
  // slot = position in the elimination order; keyrank = rank by key (vids is ascending by key)
  std::vector<int32_t> keyrank(n), vid_of_slot(n);
  for (int j = 0; j < n; j++) {
    keyrank[j] = perm[j];
    vid_of_slot[j] = vids[perm[j]];
  }
  std::map<int32_t, int32_t> slot_of_vid;
  for (int j = 0; j < n; j++) slot_of_vid[vid_of_slot[j]] = j;
  std::vector<std::vector<int32_t>> fvars(gfs.size());
  for (size_t i = 0; i < gfs.size(); i++)
    for (int32_t v : gfs[i].vids) fvars[i].push_back(slot_of_vid.at(v));
  SymbolicFronts sf;
  std::vector<std::vector<int32_t>> slot_cols;
  if (var_cols) {
    slot_cols.resize(n);
    for (int j = 0; j < n; j++) slot_cols[j] = (*var_cols)[perm[j]];
  }
  const std::string e = symbolic_multifrontal(n, keyrank, fvars, &sf, var_cols ? &slot_cols : nullptr);
  if (!e.empty()) {
    S->err = e;
    return LMGPU_INVALID;
  }
  return is_eliminate_fronts(S, gfs, vid_of_slot, sf, true, nullptr);
}

// the numeric half: the fronts of `sf` (over slots; vid_of_slot names their variables) as new cliques on the device.
// attach = false: the cliques stay outside the tree (marginalizeLeaves eliminates ONE front to get a marginal and throws the conditional
// away, ISAM2.cpp:624-637); their ids come back in *cid_out and the caller releases them.
int is_eliminate_fronts(lmgpu_isam2* S, const std::vector<IsGF>& gfs, const std::vector<int32_t>& vid_of_slot, const SymbolicFronts& sf, bool attach,
                        std::vector<int>* cid_out) {
  const int NF = (int)sf.fronts.size();
  std::vector<int> cid(NF);
  std::vector<FrontDesc> fds(NF);
  std::vector<FrontFac> ffac;
  std::vector<FacDesc> fd;
  std::vector<ChildRef> childs;
  std::vector<int32_t> cmap, fxoff;
  int max_level = 0;
  std::map<int32_t, int32_t> colof;  // vid -> column offset inside the current front
  int rc;
  for (int fi = 0; fi < NF; fi++) {
    const SymbolicFronts::F& fr = sf.fronts[fi];
    const int id = is_new_clique(S);
    cid[fi] = id;
    if (attach) S->touched.push_back(id);  // its descriptor reaches the device copy of the tree with the next patch
    lmgpu_isam2::Clq& c = S->clq[id];
    for (int32_t s : fr.frontals) c.vars.push_back(vid_of_slot[s]);
    c.nfv = (int)fr.frontals.size();
    for (int32_t s : fr.sep) c.vars.push_back(vid_of_slot[s]);
    colof.clear();
    int off = 0;
    for (size_t k = 0; k < c.vars.size(); k++) {
      colof[c.vars[k]] = off;
      off += kVarDim[S->vars[c.vars[k]].type];
      if ((int)k == c.nfv - 1) c.nf = off;
    }
    c.n = off + 1;
    if (c.n > kLdsLimitN) {
      // wider than an LDS front (loop closures of large graphs): one n x ld block, eliminated in place by the dense-front kernels
      c.ld = (c.n + 15) & ~15;
      if ((rc = is_pool_alloc(S, (size_t)c.n * c.ld, &c.f_off))) return rc;
      c.rsd_off = c.f_off;
      c.u_off = c.f_off + (int64_t)c.nf * c.ld + c.nf;
    } else {
      if ((rc = is_pool_alloc(S, (size_t)c.nf * c.n, &c.rsd_off))) return rc;
      if ((rc = is_pool_alloc(S, (size_t)(c.n - c.nf) * (c.n - c.nf), &c.u_off))) return rc;
    }
    if (attach)
      for (int k = 0; k < c.nfv; k++) S->node_of[c.vars[k]] = id;
    FrontDesc& F = fds[fi];
    F = FrontDesc{};
    F.n = c.n;
    F.nf = c.nf;
    F.rsd_off = c.rsd_off;
    F.u_off = c.u_off;
    F.ld_rsd = c.ld > 0 ? c.ld : c.n;
    F.ld_u = c.ld > 0 ? c.ld : c.n - c.nf;
    F.id = fi;
    F.fac_begin = (int)ffac.size();
    F.child_begin = (int)childs.size();
    auto add_child = [&](const lmgpu_isam2::Clq& ch) {  // its cached factor (update matrix) is extend-added through a column map
      ChildRef cr{};
      cr.u_off = ch.u_off;
      cr.ld = ch.ld > 0 ? ch.ld : ch.n - ch.nf;
      cr.m = ch.n - ch.nf;
      cr.map_begin = (int)cmap.size();
      for (size_t k = ch.nfv; k < ch.vars.size(); k++)
        for (int d = 0; d < kVarDim[S->vars[ch.vars[k]].type]; d++) cmap.push_back(colof.at(ch.vars[k]) + d);
      cmap.push_back(c.n - 1);
      childs.push_back(cr);
    };
    for (int32_t g : fr.factors) {  // own factors in the reference's order: Jacobians, cached boundary factors; orphans become children
      const IsGF& gf = gfs[g];
      if (gf.kind == 0) {
        const lmgpu_isam2::Fac& f = S->facs[gf.id];
        const lmgpu_isam2::Bkt& b = S->bkts[f.bucket];
        FacDesc d{};
        d.joff = b.joff + (int64_t)f.lidx * b.rows * b.cols;
        d.rows = (int16_t)b.rows;
        d.d0 = (int16_t)kVarDim[S->vars[f.v[0]].type];
        d.d1 = (int16_t)(f.v[1] >= 0 ? kVarDim[S->vars[f.v[1]].type] : 0);
        d.x0 = S->vars[f.v[0]].xoff;
        d.x1 = f.v[1] >= 0 ? S->vars[f.v[1]].xoff : -1;
        d.d2 = (int16_t)(f.v[2] >= 0 ? kVarDim[S->vars[f.v[2]].type] : 0);
        d.x2 = f.v[2] >= 0 ? S->vars[f.v[2]].xoff : -1;
        FrontFac ff{};
        ff.fac = (int32_t)fd.size();
        ff.c0 = colof.at(f.v[0]);
        ff.c1 = f.v[1] >= 0 ? colof.at(f.v[1]) : 0;
        ff.c2 = f.v[2] >= 0 ? colof.at(f.v[2]) : 0;
        fd.push_back(d);
        ffac.push_back(ff);
      } else if (gf.kind == 1) {
        add_child(S->clq[gf.id]);
      } else if (gf.kind == 3) {  // a marginal factor: its information matrix through a column map, like a cached boundary factor
        const lmgpu_isam2::Marg& mg = S->margs[S->facs[gf.id].marg];
        ChildRef cr{};
        cr.u_off = mg.u_off;
        cr.ld = mg.ld;
        cr.m = mg.m;
        cr.map_begin = (int)cmap.size();
        for (int32_t v : mg.vids)
          for (int d = 0; d < kVarDim[S->vars[v].type]; d++) cmap.push_back(colof.at(v) + d);
        cmap.push_back(c.n - 1);
        childs.push_back(cr);
      }
    }
    F.fac_count = (int)ffac.size() - F.fac_begin;
    for (int32_t chf : fr.children) {  // junction-tree children first (pre-order visitor), in order
      add_child(S->clq[cid[chf]]);
      c.children.push_back(cid[chf]);
      S->clq[cid[chf]].parent = id;
    }
    for (int32_t g : fr.factors)  // then the orphan subtrees whose separator this clique eliminates (ClusterTree-inst.h:228-236)
      if (gfs[g].kind == 2) {
        c.children.push_back(gfs[g].id);
        S->clq[gfs[g].id].parent = id;
      }
    F.child_count = (int)childs.size() - F.child_begin;
    F.fx_begin = (int)fxoff.size();
    for (int k = 0; k < c.nfv; k++)
      for (int d = 0; d < kVarDim[S->vars[c.vars[k]].type]; d++) fxoff.push_back(S->vars[c.vars[k]].xoff + d);
    max_level = std::max(max_level, (int)fr.level);
  }
  if (attach)
    for (int32_t r : sf.roots) S->roots.push_back(cid[r]);
  if (cid_out) *cid_out = cid;
  // ---- device: one lds_front_kernel launch per level of the new cliques
  FrontDesc* d_fds = nullptr;
  FrontFac* d_ffac = nullptr;
  FacDesc* d_fd = nullptr;
  ChildRef* d_childs = nullptr;
  int32_t *d_cmap = nullptr, *d_fxoff = nullptr, *d_list = nullptr;
  std::vector<int32_t> list;
  std::vector<std::pair<int, int>> lv(max_level + 1, {0, 0});
  std::vector<std::vector<int32_t>> wide(max_level + 1);  // per level: the fronts of more than 139 columns
  for (int l = 0; l <= max_level; l++) {
    lv[l].first = (int)list.size();
    for (int fi = 0; fi < NF; fi++) {
      if (sf.fronts[fi].level != l) continue;
      if (S->clq[cid[fi]].ld > 0)
        wide[l].push_back(fi);
      else
        list.push_back(fi);
    }
    lv[l].second = (int)list.size() - lv[l].first;
  }
  // wide fronts are assembled by one wave per row walking that row's sources in a fixed order (hbm_assemble_rows_kernel, as in the batch path)
  std::vector<int32_t> rowptr;
  std::vector<RowSrc> rowsrc;
  std::map<int32_t, int32_t> row_begin;  // front -> start of its n + 1 row pointers
  for (int l = 0; l <= max_level; l++)
    for (int32_t fi : wide[l]) {
      const FrontDesc& F = fds[fi];
      std::vector<std::vector<RowSrc>> rows(F.n);
      for (int k = 0; k < F.child_count; k++) {
        const ChildRef& c = childs[F.child_begin + k];
        for (int i = 0; i < c.m; i++) rows[cmap[c.map_begin + i]].push_back(RowSrc{F.child_begin + k, i});
      }
      for (int k = 0; k < F.fac_count; k++) {
        const FrontFac& ff = ffac[F.fac_begin + k];
        const FacDesc& d = fd[ff.fac];
        const int nc = d.d0 + d.d1 + d.d2 + 1;
        for (int p = 0; p < nc; p++) {
          const int gp = (p < d.d0) ? ff.c0 + p : (p < d.d0 + d.d1 ? ff.c1 + (p - d.d0) : (p < d.d0 + d.d1 + d.d2 ? ff.c2 + (p - d.d0 - d.d1) : F.n - 1));
          rows[gp].push_back(RowSrc{-(F.fac_begin + k) - 1, p});
        }
      }
      row_begin[fi] = (int32_t)rowptr.size();
      for (int r = 0; r < F.n; r++) {
        rowptr.push_back((int32_t)rowsrc.size());
        rowsrc.insert(rowsrc.end(), rows[r].begin(), rows[r].end());
      }
      rowptr.push_back((int32_t)rowsrc.size());
    }
  int32_t* d_rowptr = nullptr;
  RowSrc* d_rowsrc = nullptr;
  if (!rowptr.empty()) {
    if ((rc = is_stage(S, rowptr, &d_rowptr, true)) || (rc = is_stage(S, rowsrc, &d_rowsrc, true))) return rc;
    if (!S->inv16) ISCHECK(hipMalloc((void**)&S->inv16, 16 * 256 * sizeof(double)));
  }
  // (the front kernels walk these tables with chains of dependent reads: they go into the device arena, one copy for all of them)
  if ((rc = is_stage(S, fds, &d_fds, true)) || (rc = is_stage(S, ffac, &d_ffac, true)) || (rc = is_stage(S, fd, &d_fd, true)) ||
      (rc = is_stage(S, childs, &d_childs, true)) || (rc = is_stage(S, cmap, &d_cmap, true)) || (rc = is_stage(S, fxoff, &d_fxoff, true)) ||
      (rc = is_stage(S, list, &d_list, true)))
    return rc;
  {
    const int32_t fresh = 0x7f7f7f7f;  // the status word is reset by the scatter kernel of this flush
    if ((rc = is_push(S, S->d_status, &fresh, sizeof(fresh)))) return rc;
  }
  // Several levels of LDS cliques and no wide one: ONE dataflow launch for all of them (lds_front_merged_kernel: tickets bottom-up, update
  // matrices handed from child to parent by value) instead of a launch per level -- an update of VisualISAM2Example is two levels, one of
  // the city10000 loop two to five, each a dependent launch of a few workgroups.
  bool any_wide = false;
  for (int l = 0; l <= max_level; l++) any_wide = any_wide || !wide[l].empty();
  const bool merged = !any_wide && list.size() <= 4096 && !dev_switch("LMGPU_ISAM2_NO_MERGE");  // (a single level too: the launch relays the status)
  int m_nmax = 1, m_jc = 96;
  if (merged) {
    for (int32_t fi : list) {
      // "not published yet" over the clique's update matrix: a fill record of the flush's scatter kernel (LDS cliques: a dense block of its own)
      const int mm = fds[fi].n - fds[fi].nf;
      S->pushes.push_back(lmgpu_isam2::PushRec{S->pool + fds[fi].u_off, nullptr, (uint32_t)(2 * mm * mm), 1u});
      m_nmax = std::max(m_nmax, fds[fi].n);
      int tot = 0;
      for (int k = 0; k < fds[fi].fac_count; k++) {
        const FacDesc& d = fd[ffac[fds[fi].fac_begin + k].fac];
        tot += d.rows * (d.d0 + d.d1 + d.d2 + 1);
      }
      m_jc = std::max(m_jc, std::min(tot, LDSF_JCAP));
    }
    if (!S->d_eticket) ISCHECK(hipMalloc((void**)&S->d_eticket, 2 * sizeof(unsigned int)));
    const unsigned int zero[2] = {0u, 0u};  // the ticket counter, the count of finished workgroups
    if ((rc = is_push(S, S->d_eticket, zero, sizeof(zero)))) return rc;
  }
  // the tree is final here (new cliques, adopted orphans, roots): what the next walk needs rides this flush (see walk_prepared)
  if (attach && !S->dogleg && !dev_switch("LMGPU_ISAM2_LATE_WALK_PREP") && (rc = is_walk_prepare(S, is_walk_by_value(S)))) return rc;
  *S->h_status = 0x7f7f7f7f;  // (before the launch that may relay the status word)
  if ((rc = is_flush(S))) return rc;  // the tables above, and whatever the update pushed before (new values, factor rows)
  if (merged) {
    const int jcap = (m_jc + 7) & ~7;
    const size_t lds = kLdsFrontExtra - (size_t)(LDSF_JCAP - jcap) * 8 + 64 + (size_t)m_nmax * m_nmax * sizeof(double);
    if (m_nmax > 72)
      hipLaunchKernelGGL(lds_front_merged_kernel<1024>, dim3((unsigned)list.size()), dim3(1024), lds, S->stream, (const int32_t*)d_list, 0, (int)list.size(),
                         (const FrontDesc*)d_fds, (const FrontFac*)d_ffac, (const FacDesc*)d_fd, (const ChildRef*)d_childs, (const int32_t*)d_cmap,
                         (const int32_t*)d_fxoff, S->pool, 0.0, (const double*)nullptr, (const double*)S->ones, S->d_status, m_nmax, jcap,
                         (const double*)nullptr, S->d_eticket, S->h_status_dev);
    else
      // (four waves unless every clique takes the one-wave register path: the blocked Cholesky of the LDS body needs four -- with two, the 30-40
      //  pivots of VisualISAM2Example's root clique were taken one by one: 33 of the 100 us its launch took)
      hipLaunchKernelGGL(lds_front_merged_kernel<256>, dim3((unsigned)list.size()), dim3(m_nmax <= 16 ? 64 : 256), lds, S->stream,
                         (const int32_t*)d_list, 0, (int)list.size(), (const FrontDesc*)d_fds, (const FrontFac*)d_ffac, (const FacDesc*)d_fd,
                         (const ChildRef*)d_childs, (const int32_t*)d_cmap, (const int32_t*)d_fxoff, S->pool, 0.0, (const double*)nullptr,
                         (const double*)S->ones, S->d_status, m_nmax, jcap, (const double*)nullptr, S->d_eticket, S->h_status_dev);
  }
  for (int l = 0; l <= max_level && !merged; l++) {
    for (int32_t fi : wide[l]) {  // (the level's LDS fronts and these only depend on the levels below)
      const FrontDesc& F = fds[fi];
      const lmgpu_isam2::Clq& c = S->clq[cid[fi]];
      const int ld = c.ld, n = c.n;
      double* A = S->pool + c.f_off;
      ISCHECK(hipMemsetAsync(A, 0, (size_t)n * ld * sizeof(double), S->stream));
      hipLaunchKernelGGL(hbm_assemble_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, S->stream, F, c.f_off, ld, (const int32_t*)(d_rowptr + row_begin[fi]),
                         (const RowSrc*)d_rowsrc, (const ChildRef*)d_childs, (const int32_t*)d_cmap, (const FrontFac*)d_ffac, (const FacDesc*)d_fd, S->pool, 1);
      // right-looking over 256-row outer panels, the two-launch panel form + the trailing update (lmgpu.hip: panel_alone / the unfused step)
      const int np = (F.nf + NBO - 1) / NBO;
      for (int i = 0; i < np; i++) {
        const int k0 = i * NBO, kb = std::min(F.nf, (i + 1) * NBO) - k0, r0 = k0 + kb, cols = n - r0, m = n - r0;
        hipLaunchKernelGGL(diag_potrf_kernel, dim3(1), dim3(256), DIAG_LDS_BYTES, S->stream, A, ld, F.nf, k0, kb, F.id, S->d_status, S->inv16);
        if (cols > 0) hipLaunchKernelGGL(panel_trsm_kernel, dim3((cols + 63) / 64), dim3(256), 0, S->stream, A, ld, n, k0, kb, (const double*)S->inv16);
        if (m > 0) {
          if (m <= 1024) {
            const int Sx = (m + 63) / 64;
            hipLaunchKernelGGL(syrk_quadrants_kernel, dim3(4 * Sx, Sx), dim3(256), 0, S->stream, A, ld, n, k0, kb, r0);
          } else {
            const int T = (m + 127) / 128;
            hipLaunchKernelGGL(syrk_mfma_kernel, dim3(T, T), dim3(256), kSyrkLds, S->stream, A, ld, n, k0, kb, r0, n);
          }
        }
      }
    }
    if (lv[l].second == 0) continue;
    int nmax = 1, jc = 96;
    for (int q = 0; q < lv[l].second; q++) {
      const int fi = list[lv[l].first + q];
      nmax = std::max(nmax, fds[fi].n);
      int tot = 0;
      for (int k = 0; k < fds[fi].fac_count; k++) {
        const FacDesc& d = fd[ffac[fds[fi].fac_begin + k].fac];
        tot += d.rows * (d.d0 + d.d1 + d.d2 + 1);
      }
      jc = std::max(jc, std::min(tot, LDSF_JCAP));
    }
    const int jcap = (jc + 7) & ~7;
    const int threads = nmax <= 24 ? 64 : 256;
    const size_t lds = kLdsFrontExtra - (size_t)(LDSF_JCAP - jcap) * 8 + 64 + (size_t)nmax * nmax * sizeof(double);
    if (nmax > 72 && lv[l].second <= 256)  // a handful of wide cliques: sixteen waves each (the batch path's rule)
      hipLaunchKernelGGL((lds_front_kernel<false, 1024>), dim3(lv[l].second), dim3(1024), lds, S->stream, (const int32_t*)(d_list + lv[l].first),
                         (const FrontDesc*)d_fds, (const FrontFac*)d_ffac, (const FacDesc*)d_fd, (const ChildRef*)d_childs, (const int32_t*)d_cmap,
                         (const int32_t*)d_fxoff, S->pool, 0.0, (const double*)nullptr, (const double*)S->ones, S->d_status, nmax, nmax, (double*)nullptr, jcap,
                         (const double*)nullptr, (const char*)nullptr, 0);
    else
      hipLaunchKernelGGL(lds_front_kernel<false>, dim3(lv[l].second), dim3(threads), lds, S->stream, (const int32_t*)(d_list + lv[l].first),
                         (const FrontDesc*)d_fds, (const FrontFac*)d_ffac, (const FacDesc*)d_fd, (const ChildRef*)d_childs, (const int32_t*)d_cmap,
                         (const int32_t*)d_fxoff, S->pool, 0.0, (const double*)nullptr, (const double*)S->ones, S->d_status, nmax, nmax, (double*)nullptr, jcap,
                         (const double*)nullptr, (const char*)nullptr, 0);
  }
  // the status word comes back with the one wait that ends the update (is_finish_elimination): the mark kernel, the last launch of an
  // update that eliminated anything, relays it into the host's pinned word
  S->elim_relay_pending = !merged;  // (the merged launch hands the status to the host itself)
  S->elim_cid = cid;
  S->elim_pending = true;
  return LMGPU_OK;
}

// the wait that ends an update: EliminateCholesky failed -> IndeterminantLinearSystemException(first frontal key), HessianFactor.cpp:475-482
int is_finish_elimination(lmgpu_isam2* S) {
  const int rcf = is_flush(S);  // nothing staged or pushed outlives the entry point
  if (rcf) return rcf;
  if (S->elim_relay_pending) {  // (an elimination that marked nothing: no kernel relayed the status)
    ISCHECK(hipMemcpyAsync(S->h_status, S->d_status, sizeof(int), hipMemcpyDeviceToHost, S->stream));
    S->elim_relay_pending = false;
  }
  ISCHECK(hipStreamSynchronize(S->stream));
  ISCHECK(hipGetLastError());
  if (!S->elim_pending) return LMGPU_OK;
  S->elim_pending = false;
  if (*S->h_status < (int)S->elim_cid.size()) {
    S->failed_key = S->vars[S->clq[S->elim_cid[*S->h_status]].vars[0]].key;
    S->err = "indeterminate linear system";
    S->walk_prepared = false;  // (the status word a walk starts from has to be pushed again)
    return LMGPU_INDETERMINATE;
  }
  return LMGPU_OK;
}


// upload an index list and run `fn(device list, count)`
template <typename Fn>
int is_with_list(lmgpu_isam2* S, const std::vector<int32_t>& v, Fn fn) {
  if (v.empty()) return LMGPU_OK;
  int32_t* d = nullptr;
  int rc = is_stage(S, v, &d);
  if (rc) return rc;
  if ((rc = is_flush(S))) return rc;
  fn((const int32_t*)d, (int)v.size());
  return LMGPU_OK;
}

// nonlinearFactors_.error(values) (gtsam/nonlinear/NonlinearFactorGraph.cpp:170-179) over the factors still in the graph: the error kernels
// of the batch path on every bucket, one error per factor into its place (removed rows into the dump slot), fixed-order reduction.
// at_estimate: the values are theta retracted by the current delta (the caller has brought delta up to date), else theta itself.
int is_graph_error(lmgpu_isam2* S, bool at_estimate, double* out) {
  const size_t nfac = S->facs.size();
  *out = 0.0;
  if (nfac == 0) return LMGPU_OK;
  int rc;
  if (nfac + 1 > S->ebuf_cap) {
    const size_t cap = is_next_cap(S->ebuf_cap, nfac + 1);
    if ((rc = is_realloc(S, &S->d_ebuf, cap, 0))) return rc;
    S->ebuf_cap = cap;
  }
  if (!S->d_epart) ISCHECK(hipMalloc((void**)&S->d_epart, 264 * sizeof(double)));
  if (!S->h_escal) ISCHECK(hipHostMalloc((void**)&S->h_escal, sizeof(double), hipHostMallocDefault));
  hipStream_t s = S->stream;
  if ((rc = is_flush(S))) return rc;
  ISCHECK(hipMemsetAsync(S->d_ebuf, 0, (nfac + 1) * sizeof(double), s));
  ValuesDev vals;
  for (int t = 0; t < kNumVarTypes; t++) {
    vals.v[t] = S->theta[t];
    if (at_estimate && S->type_count[t] > 0) {
      hipLaunchKernelGGL(retract_kernel, dim3((S->type_count[t] + 255) / 256), dim3(256), 0, s, t, S->type_count[t], (const double*)S->theta[t], S->est[t],
                         (const int32_t*)S->d_type_xoff[t], (const double*)S->delta, (const int32_t*)nullptr);
      vals.v[t] = S->est[t];
    }
  }
  for (const lmgpu_isam2::Bkt& b : S->bkts) {
    if (b.n == 0) continue;
    BucketDev d;
    d.type = b.type;
    d.n = b.n;
    d.noise_kind = b.noise_kind;
    d.vidx = b.d_vidx;
    d.meas = b.d_meas;
    d.noise = b.d_noise;
    d.J = nullptr;
    d.epos = b.d_epos;
    d.robust = b.robust;
    d.rk = b.rk;
    d.sel = nullptr;
    const int g256 = (b.n + 255) / 256, g128 = (b.n + 127) / 128;
    double* eb = S->d_ebuf;
    switch (b.type) {
      case LMGPU_F_SFM: hipLaunchKernelGGL(sfm_error_kernel, dim3(g256), dim3(256), 0, s, d, vals, eb); break;
      case LMGPU_F_BETWEEN_POSE2: hipLaunchKernelGGL((generic_factor_kernel<1, 3, 3, 3, 3, 0, 3, 0, 3, false>), dim3(g128), dim3(128), 0, s, d, vals, eb); break;
      case LMGPU_F_BETWEEN_POSE3: hipLaunchKernelGGL((generic_factor_kernel<2, 6, 6, 6, 12, 1, 12, 1, 12, false>), dim3(g128), dim3(128), 0, s, d, vals, eb); break;
      case LMGPU_F_PRIOR_POSE2: hipLaunchKernelGGL((generic_factor_kernel<3, 3, 3, 0, 3, 0, 3, -1, 0, false>), dim3(g128), dim3(128), 0, s, d, vals, eb); break;
      case LMGPU_F_PRIOR_POSE3: hipLaunchKernelGGL((generic_factor_kernel<4, 6, 6, 0, 12, 1, 12, -1, 0, false>), dim3(g128), dim3(128), 0, s, d, vals, eb); break;
      case LMGPU_F_PRIOR_POINT3: hipLaunchKernelGGL((generic_factor_kernel<5, 3, 3, 0, 3, 2, 3, -1, 0, false>), dim3(g128), dim3(128), 0, s, d, vals, eb); break;
      case LMGPU_F_PRIOR_CAM: hipLaunchKernelGGL((generic_factor_kernel<6, 9, 9, 0, 15, 3, 15, -1, 0, false>), dim3(g128), dim3(128), 0, s, d, vals, eb); break;
      case LMGPU_F_PROJECTION: hipLaunchKernelGGL((generic_factor_kernel<7, 2, 6, 3, 7, 1, 12, 2, 3, false>), dim3(g128), dim3(128), 0, s, d, vals, eb); break;
      case LMGPU_F_PROJECTION_BPS: hipLaunchKernelGGL((generic_factor_kernel<8, 2, 6, 3, 19, 1, 12, 2, 3, false>), dim3(g128), dim3(128), 0, s, d, vals, eb); break;
      case LMGPU_F_BEARING_RANGE_2D: hipLaunchKernelGGL((generic_factor_kernel<9, 2, 3, 2, 2, 0, 3, 4, 2, false>), dim3(g128), dim3(128), 0, s, d, vals, eb); break;
      case LMGPU_F_SFM2: hipLaunchKernelGGL(sfm2_factor_kernel<false>, dim3(g128), dim3(128), 0, s, d, vals, eb); break;
      case LMGPU_F_PRIOR_CAL3_S2: hipLaunchKernelGGL((generic_factor_kernel<11, 5, 5, 0, 5, 5, 5, -1, 0, false>), dim3(g128), dim3(128), 0, s, d, vals, eb); break;
    }
  }
  const int g = std::min(256, std::max(1, ((int)nfac + 255) / 256));
  hipLaunchKernelGGL(reduce_stage1, dim3(g), dim3(256), 0, s, (const double*)(S->d_ebuf + 1), (int)nfac, S->d_epart);
  hipLaunchKernelGGL(reduce_stage2, dim3(1), dim3(256), 0, s, (const double*)S->d_epart, g, S->d_epart + 256);
  ISCHECK(hipMemcpyAsync(S->h_escal, S->d_epart + 256, sizeof(double), hipMemcpyDeviceToHost, s));
  ISCHECK(hipStreamSynchronize(s));
  *out = *S->h_escal;
  return LMGPU_OK;
}

// ISAM2::updateDelta with ISAM2DoglegParams (gtsam/nonlinear/ISAM2.cpp:739-779): the Newton point by the wildfire walk (into deltaNewton_),
// the steepest-descent point from the tree's gradient (gradientAtZero :825-833, UpdateRgProd / ComputeGradientSearch ISAM2-impl.cpp:82-157),
// then DoglegOptimizerImpl::Iterate (DoglegOptimizerImpl.h:139-254) with Rd = the tree, f = the nonlinear graph, x0 = theta.  The vectors
// stay on the device; per trial point the host sees three scalars (the graph error at theta (+) dx_d, the tree's error at dx_d) and
// decides about the radius like the reference.  delta_ = dx_d, doglegDelta_ = the new radius.
int is_update_delta_dogleg(lmgpu_isam2* S, bool force_full, bool host_delta) {
  int rc = is_update_delta_enqueue(S, force_full, false, S->delta_newton);  // patches the tree, walks, clears the replaced flags
  if (rc) return rc;
  if ((rc = is_update_delta_finish(S))) return rc;
  if (S->ntot == 0) return LMGPU_OK;
  hipStream_t s = S->stream;
  const int n = S->ntot;
  auto to_host = [&]() -> int {  // the pinned copy of delta for CheckRelinearizationFull, and the wait that ends the call
    if (host_delta) {
      if ((size_t)S->ntot > S->h_delta_cap) {
        if (S->h_delta) (void)hipHostFree(S->h_delta);
        S->h_delta = nullptr;
        S->h_delta_cap = is_next_cap(S->h_delta_cap, (size_t)S->ntot);
        ISCHECK(hipHostMalloc((void**)&S->h_delta, S->h_delta_cap * sizeof(double), hipHostMallocDefault));
      }
      ISCHECK(hipMemcpyAsync(S->h_delta, S->delta, (size_t)S->ntot * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    ISCHECK(hipStreamSynchronize(s));
    return LMGPU_OK;
  };
  std::vector<int32_t> alive;
  for (int id = 0; id < (int)S->clq.size(); id++)
    if (S->clq[id].alive) alive.push_back(id);
  const int nc = (int)alive.size();
  if (nc == 0) return to_host();  // no tree yet (the first update): delta stays zero
  int32_t* d_alive = nullptr;
  if ((rc = is_stage(S, alive, &d_alive))) return rc;
  if ((size_t)nc > S->cerr_cap) {
    const size_t cap = is_next_cap(S->cerr_cap, (size_t)nc);
    if ((rc = is_realloc(S, &S->d_cerr, cap, 0))) return rc;
    S->cerr_cap = cap;
  }
  if (!S->d_dlscal) ISCHECK(hipMalloc((void**)&S->d_dlscal, 264 * sizeof(double)));
  if (!S->h_dlscal) ISCHECK(hipHostMalloc((void**)&S->h_dlscal, 8 * sizeof(double), hipHostMallocDefault));
  if ((rc = is_flush(S))) return rc;
  const FrontDesc* T = (const FrontDesc*)S->d_tree;
  const int32_t *FX = (const int32_t*)S->d_tree_fx, *SX = (const int32_t*)S->d_tree_sx;
  const dim3 gq((nc + 3) / 4), gv((n + 255) / 256);
  auto tree_error = [&](const double* x, double* dst) {  // dst = sum over the cliques of ||[R S] x - d||^2
    hipLaunchKernelGGL(isam2_tree_error_kernel, gq, dim3(256), 0, s, (const int32_t*)d_alive, nc, T, FX, SX, (const double*)S->pool, x, S->d_cerr);
    const int g = std::min(256, std::max(1, (nc + 255) / 256));
    hipLaunchKernelGGL(reduce_stage1, dim3(g), dim3(256), 0, s, (const double*)S->d_cerr, nc, S->d_dlscal + 8);
    hipLaunchKernelGGL(reduce_stage2, dim3(1), dim3(256), 0, s, (const double*)(S->d_dlscal + 8), g, dst);
  };
  {
    // the gradient as a fixed-order sum: slot table of this tree (clique by clique in `alive` order), per scalar the slots that are its
    std::vector<int32_t> part_off((size_t)nc), ptr((size_t)n + 1, 0), idx;
    int32_t off = 0;
    for (int li = 0; li < nc; li++) {
      part_off[(size_t)li] = off;
      off += S->clq[alive[(size_t)li]].n - 1;
    }
    for (int li = 0; li < nc; li++)  // count
      for (int32_t v : S->clq[alive[(size_t)li]].vars) {
        const int xo = S->vars[v].xoff, d = kVarDim[S->vars[v].type];
        for (int q = 0; q < d; q++) ptr[(size_t)(xo + q) + 1]++;
      }
    for (int x = 0; x < n; x++) ptr[(size_t)x + 1] += ptr[(size_t)x];
    idx.resize((size_t)ptr[(size_t)n]);
    std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
    for (int li = 0; li < nc; li++) {
      int col = 0;
      for (int32_t v : S->clq[alive[(size_t)li]].vars) {
        const int xo = S->vars[v].xoff, d = kVarDim[S->vars[v].type];
        for (int q = 0; q < d; q++) idx[(size_t)fill[(size_t)(xo + q)]++] = part_off[(size_t)li] + col + q;
        col += d;
      }
    }
    if ((size_t)off > S->gpart_cap) {
      const size_t cap = is_next_cap(S->gpart_cap, (size_t)off);
      if ((rc = is_realloc(S, &S->d_gpart, cap, 0))) return rc;
      S->gpart_cap = cap;
    }
    int32_t *d_poff = nullptr, *d_ptr = nullptr, *d_idx = nullptr;
    if ((rc = is_stage(S, part_off, &d_poff)) || (rc = is_stage(S, ptr, &d_ptr)) || (rc = is_stage(S, idx, &d_idx))) return rc;
    if ((rc = is_flush(S))) return rc;
    hipLaunchKernelGGL(isam2_tree_gradient_kernel, gq, dim3(256), 0, s, (const int32_t*)d_alive, nc, T, FX, SX, (const double*)S->pool, (const int32_t*)d_poff,
                       S->d_gpart);
    hipLaunchKernelGGL(isam2_gradient_gather_kernel, gv, dim3(256), 0, s, (const int32_t*)d_ptr, (const int32_t*)d_idx, (const double*)S->d_gpart, n, S->grad);
  }
  hipLaunchKernelGGL(isam2_tree_rg_kernel, gq, dim3(256), 0, s, (const int32_t*)d_alive, nc, T, FX, SX, (const double*)S->pool, (const double*)S->grad, S->rgprod);
  double* sc = S->d_dlscal;  // [0] g.g  [1] Rg.Rg  [2] u.u  [3] n.n  [4] u.n  [5] M(0) x 2  [6] M(dx_d) x 2
  hipLaunchKernelGGL(isam2_dot_kernel, dim3(1), dim3(1024), 0, s, (const double*)S->grad, (const double*)S->grad, n, sc + 0);
  hipLaunchKernelGGL(isam2_dot_kernel, dim3(1), dim3(1024), 0, s, (const double*)S->rgprod, (const double*)S->rgprod, n, sc + 1);
  hipLaunchKernelGGL(isam2_gradient_search_kernel, gv, dim3(256), 0, s, (const double*)S->grad, (const double*)sc, n, S->dx_u);
  hipLaunchKernelGGL(isam2_dot_kernel, dim3(1), dim3(1024), 0, s, (const double*)S->dx_u, (const double*)S->dx_u, n, sc + 2);
  hipLaunchKernelGGL(isam2_dot_kernel, dim3(1), dim3(1024), 0, s, (const double*)S->delta_newton, (const double*)S->delta_newton, n, sc + 3);
  hipLaunchKernelGGL(isam2_dot_kernel, dim3(1), dim3(1024), 0, s, (const double*)S->dx_u, (const double*)S->delta_newton, n, sc + 4);
  tree_error(nullptr, sc + 5);
  ISCHECK(hipMemcpyAsync(S->h_dlscal, sc, 8 * sizeof(double), hipMemcpyDeviceToHost, s));
  double f_error = 0.0;
  if ((rc = is_graph_error(S, false, &f_error))) return rc;  // nonlinearFactors_.error(theta_); waits
  const double uu = S->h_dlscal[2], nn = S->h_dlscal[3], un = S->h_dlscal[4], M_error = 0.5 * S->h_dlscal[5];
  double delta = S->dogleg_delta;
  enum { NONE, INCREASED_DELTA, DECREASED_DELTA } lastAction = NONE;
  const int mode = S->dogleg_mode;
  bool stay = true, zero_step = false;
  while (stay) {
    // ComputeDoglegPoint (DoglegOptimizerImpl.cpp:26-64): dx_d = a dx_u + b dx_n
    double a = 0.0, b = 1.0;
    const double deltaSq = delta * delta;
    if (deltaSq < uu) {
      a = std::sqrt(deltaSq / uu);
      b = 0.0;
    } else if (deltaSq < nn) {  // ComputeBlend :67-91
      const double qa = uu - 2. * un + nn, qb = 2. * (un - uu), qc = uu - deltaSq;
      const double sq = std::sqrt(qb * qb - 4 * qa * qc);
      const double tau1 = (-qb + sq) / (2. * qa), tau2 = (-qb - sq) / (2. * qa);
      const double eps = std::numeric_limits<double>::epsilon();
      const double tau = (-eps <= tau1 && tau1 <= 1.0 + eps) ? tau1 : tau2;
      a = 1. - tau;
      b = tau;
    }
    hipLaunchKernelGGL(isam2_blend_kernel, gv, dim3(256), 0, s, (const double*)S->dx_u, (const double*)S->delta_newton, a, b, n, S->delta);
    tree_error((const double*)S->delta, sc + 6);
    ISCHECK(hipMemcpyAsync(S->h_dlscal + 6, sc + 6, sizeof(double), hipMemcpyDeviceToHost, s));
    double new_f = 0.0;
    if ((rc = is_graph_error(S, true, &new_f))) return rc;  // f.error(x0.retract(dx_d)); waits
    const double new_M = 0.5 * S->h_dlscal[6];
    const double rho = (std::fabs(f_error - new_f) < 1e-15 || std::fabs(M_error - new_M) < 1e-15) ? 0.5 : (f_error - new_f) / (M_error - new_M);
    if (rho >= 0.75) {
      const double dnorm = std::sqrt(std::max(0.0, a * a * uu + 2. * a * b * un + b * b * nn));
      const double newDelta = std::max(delta, 3.0 * dnorm);
      if (mode == 2 || mode == 1) {
        stay = false;
      } else if (std::fabs(newDelta - delta) < 1e-15 || lastAction == DECREASED_DELTA) {
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
    } else {
      if (delta > 1e-5) {
        delta *= 0.5;
        stay = true;
        lastAction = DECREASED_DELTA;
      } else {
        zero_step = true;  // "don't allow error to increase"
        stay = false;
      }
    }
  }
  if (zero_step) ISCHECK(hipMemsetAsync(S->delta, 0, (size_t)n * sizeof(double), s));
  S->dogleg_delta = delta;
  return to_host();
}

// NonlinearFactorGraph::remove(i) + linearFactors_.remove(i): the slot empties (the caller has taken it out of the variable index)
int is_empty_slot(lmgpu_isam2* S, int32_t idx) {
  lmgpu_isam2::Fac& f = S->facs[idx];
  f.removed = true;
  if (f.marg >= 0) {
    is_free_marg(S, f.marg);
    f.marg = -1;
    return LMGPU_OK;
  }
  const int32_t dump = 0;  // a typed factor's row stays in its bucket; its error lands in the dump slot from now on
  return is_push(S, S->bkts[f.bucket].d_epos + f.lidx, &dump, sizeof(dump));
}

// ISAM2::update (gtsam/nonlinear/ISAM2.cpp:419-480)
int is_update(lmgpu_isam2* S, const lmgpu_isam2::UpParams& up, lmgpu_isam2_result* result) {
  const bool force_relinearize = up.force_relinearize;
  // what can be refused is refused before anything changes
  for (uint64_t idx : up.remove)
    if (idx >= S->facs.size()) {
      S->err = "ISAM2: removeFactorIndices names a factor that does not exist (factors added by the same update cannot be removed by it)";
      return LMGPU_INVALID;
    }
  for (uint64_t k : up.extra_reelim) {
    bool known = S->vid_of.count(k) > 0;
    for (const lmgpu_isam2::NewVar& nv : S->new_vars) known = known || nv.key == k;
    if (!known) {
      S->err = "ISAM2: extraReelimKeys names an unknown variable";
      return LMGPU_INVALID;
    }
  }
  {
    // the pending input is checked as a whole BEFORE anything changes; a refused update drops it, so that the handle stays as it was and the
    // next update starts clean (a refusal in the middle used to leave variables committed, factors half entered and the queues still full)
    auto refuse = [&](const char* why) {
      S->new_vars.clear();
      S->new_facs.clear();
      S->err = why;
      return LMGPU_INVALID;
    };
    std::map<uint64_t, int32_t> pending_type;
    for (const lmgpu_isam2::NewVar& nv : S->new_vars) {
      if (S->vid_of.count(nv.key) || !pending_type.emplace(nv.key, nv.type).second) return refuse("ISAM2: variable already exists");
      if (!S->relin_thresholds.empty()) {  // the reference throws when a variable's Symbol character has no vector of its dimension (ISAM2-impl.h:258-262)
        auto it = S->relin_thresholds.find((unsigned char)(nv.key >> 56));
        if (it == S->relin_thresholds.end() || (int)it->second.size() != kVarDim[nv.type])
          return refuse("ISAM2: the relinearization threshold (FastMap<char, Vector>) has no vector of the right dimension for a new variable's Symbol character");
      }
    }
    for (const lmgpu_isam2::NewFac& nf : S->new_facs)
      for (int k = 0; k < kFactorArity[nf.type]; k++) {
        int32_t t = -1;
        auto it = S->vid_of.find(nf.k[k]);
        if (it != S->vid_of.end()) {
          t = S->vars[it->second].type;
        } else {
          auto pt = pending_type.find(nf.k[k]);
          if (pt == pending_type.end()) return refuse("ISAM2: a new factor references a variable that has no value");
          t = pt->second;
        }
        if (t != factor_var_type(nf.type, k)) return refuse("factor/variable type mismatch");
      }
    if (up.has_constrained)
      for (auto& kg : up.constrained)
        if (!S->vid_of.count(kg.first) && !pending_type.count(kg.first)) return refuse("ISAM2: constrainedKeys names a variable that is not in the system");
  }
  S->update_count += 1;
  lmgpu_isam2_result res{};
  int rc;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_last = S->trace ? now() : 0.0;
  auto lap = [&](int i) {
    if (!S->trace) return;
    const double t = now();
    S->t_phase[i] += t - t_last;
    if (t - t_last > 0.02) std::fprintf(stderr, "isam2 update %d: phase %d took %.1f ms\n", S->update_count, i, 1e3 * (t - t_last));
    t_last = t;
  };
  if ((rc = is_stage_begin(S))) return rc;
  S->elim_pending = false;
  // ---- addVariables :365-384
  for (const lmgpu_isam2::NewVar& nv : S->new_vars)
    if (S->vid_of.count(nv.key)) {
      S->err = "ISAM2: variable already exists";
      return LMGPU_INVALID;
    }
  std::vector<lmgpu_isam2::NewVar> new_vars;
  new_vars.swap(S->new_vars);
  std::vector<lmgpu_isam2::NewFac> new_facs;
  new_facs.swap(S->new_facs);
  if (!new_vars.empty()) {
    int add[kNumVarTypes] = {}, addtot = 0;
    for (auto& nv : new_vars) {
      add[nv.type]++;
      addtot += kVarDim[nv.type];
    }
    for (int t = 0; t < kNumVarTypes; t++) {
      const size_t need = (size_t)(S->type_count[t] + add[t]);
      if (!add[t] || need <= (size_t)S->type_cap[t]) continue;
      const size_t ncap = is_next_cap((size_t)S->type_cap[t], need);
      if ((rc = is_realloc(S, &S->theta[t], ncap * kVarStore[t], (size_t)S->type_count[t] * kVarStore[t]))) return rc;
      if ((rc = is_realloc(S, &S->est[t], ncap * kVarStore[t], 0))) return rc;
      if ((rc = is_realloc(S, &S->d_type_xoff[t], ncap, (size_t)S->type_count[t]))) return rc;
      S->type_cap[t] = (int)ncap;
    }
    if (S->ntot + addtot > S->ntot_cap) {
      const size_t ncap = is_next_cap((size_t)S->ntot_cap, (size_t)(S->ntot + addtot));
      if ((rc = is_realloc(S, &S->delta, ncap, (size_t)S->ntot))) return rc;
      if (S->dogleg) {
        if ((rc = is_realloc(S, &S->delta_newton, ncap, (size_t)S->ntot)) || (rc = is_realloc(S, &S->rgprod, ncap, (size_t)S->ntot)) ||
            (rc = is_realloc(S, &S->grad, ncap, 0)) || (rc = is_realloc(S, &S->dx_u, ncap, 0)))
          return rc;
      }
      if ((rc = is_realloc(S, &S->ones, ncap, 0))) return rc;
      if ((rc = is_realloc(S, &S->d_replaced, ncap, (size_t)S->ntot))) return rc;
      if ((rc = is_realloc(S, &S->d_changed, ncap, 0))) return rc;
      ISCHECK(hipMemsetAsync(S->d_changed, 0, ncap, S->stream));  // (epochs: a fresh array must not hold the current one)
      std::vector<double> one(ncap, 1.0);
      ISCHECK(hipMemcpy(S->ones, one.data(), ncap * sizeof(double), hipMemcpyHostToDevice));
      ISCHECK(hipMemsetAsync(S->d_replaced + S->ntot, 0, ncap - S->ntot, S->stream));
      S->ntot_cap = (int)ncap;
    }
    // the new variables of a type take consecutive places: one copy per type for the values, one for the delta offsets
    const int ntot0 = S->ntot;
    int first_tidx[kNumVarTypes];
    std::vector<double> vals[kNumVarTypes];
    std::vector<int32_t> xoffs[kNumVarTypes];
    for (int t = 0; t < kNumVarTypes; t++) first_tidx[t] = S->type_count[t];
    for (auto& nv : new_vars) {
      lmgpu_isam2::Var v{nv.key, nv.type, S->type_count[nv.type]++, S->ntot, false};
      const int vid = (int)S->vars.size();
      S->vars.push_back(v);
      S->vid_of[nv.key] = vid;
      S->vindex.emplace_back();
      S->node_of.push_back(-1);
      S->replaced.push_back(0);
      vals[nv.type].insert(vals[nv.type].end(), nv.v, nv.v + kVarStore[nv.type]);
      xoffs[nv.type].push_back(v.xoff);
      S->ntot += kVarDim[nv.type];
    }
    for (int t = 0; t < kNumVarTypes; t++) {
      if (xoffs[t].empty()) continue;
      if ((rc = is_push(S, S->theta[t] + (size_t)first_tidx[t] * kVarStore[t], vals[t].data(), vals[t].size() * sizeof(double)))) return rc;
      if ((rc = is_push(S, S->d_type_xoff[t] + first_tidx[t], xoffs[t].data(), xoffs[t].size() * sizeof(int32_t)))) return rc;
    }
    // delta_.insert(zeroVectors): a byte-fill record of the next flush (a fill command of its own was one more device operation per update).
    // (The all-ones fill a walk pushes over the same scalars must not ride the same flush: two records of one scatter launch are not
    // ordered.  The linearization of the new factors flushes in between; is_walk_prepare flushes itself if it did not.)
    S->delta_zero_pending = true;
    S->pushes.push_back(lmgpu_isam2::PushRec{S->delta + ntot0, (const void*)(uintptr_t)0, (uint32_t)((size_t)(S->ntot - ntot0) * sizeof(double)), 2u});
    if (S->dogleg) {  // deltaNewton_ / RgProd_.insert(zeroVectors) (ISAM2.cpp:373-374)
      ISCHECK(hipMemsetAsync(S->delta_newton + ntot0, 0, (size_t)(S->ntot - ntot0) * sizeof(double), S->stream));
      ISCHECK(hipMemsetAsync(S->rgprod + ntot0, 0, (size_t)(S->ntot - ntot0) * sizeof(double), S->stream));
    }
  }
  const bool relinNeeded = force_relinearize || (S->prm.enableRelinearization && S->prm.relinearizeSkip > 0 && S->update_count % S->prm.relinearizeSkip == 0);
  const bool reads_before = S->delta_reads > 0;
  S->delta_reads = 0;
  if (relinNeeded) {
    // delta and the host's mirror of it are current when nothing was re-eliminated since the last walk (the previous update ended with
    // it): updateDelta would visit the roots and find nothing to do
    const bool current = !S->dogleg && !S->any_replaced && !up.force_full_solve && S->mirror_ntot > 0 && (size_t)S->ntot <= S->h_delta_cap;
    if (current) {
      for (size_t i = S->mirror_ntot; i < (size_t)S->ntot; i++) S->h_delta[i] = 0.0;  // delta_.insert(zeroVectors)
      S->mirror_ntot = (size_t)S->ntot;
    } else if ((rc = is_update_delta(S, up.force_full_solve, true))) {
      return rc;
    }
  }
  const int relin_ntot = S->ntot;  // scalars of delta the pinned copy holds
  lap(0);  // new variables + updateDelta (wildfire, one wait)
  // ---- 1. pushBackFactors (ISAM2-impl.h:145-175): FactorGraph::add_factors (FactorGraph-inst.h:109-137) -- the indices continue the list,
  //         or (findUnusedFactorSlots) the new factors fill the empty slots from the front
  std::vector<int32_t> new_idx;
  size_t slot_scan = 0;
  std::set<uint64_t> markedKeys;
  std::map<int, std::vector<int32_t>> new_by_bucket;  // bucket -> new local indices
  struct NewRows {
    int first = -1;
    std::vector<int32_t> vidx, epos;
    std::vector<double> meas, noise;
  };
  std::map<int, NewRows> new_rows;  // bucket -> the descriptor rows of its new factors (consecutive local indices), uploaded after the loop
  for (const lmgpu_isam2::NewFac& nf : new_facs) {
    const int ar = kFactorArity[nf.type];
    lmgpu_isam2::Fac f{nf.type, -1, -1, {-1, -1, -1}, false, -1};
    size_t slot = S->facs.size();
    if (S->find_unused_slots) {
      while (slot_scan < S->facs.size() && !S->facs[slot_scan].removed) ++slot_scan;
      slot = slot_scan;
    }
    for (int k = 0; k < ar; k++) {
      auto it = S->vid_of.find(nf.k[k]);
      if (it == S->vid_of.end()) {
        S->err = "ISAM2: a new factor references a variable that has no value";
        return LMGPU_INVALID;
      }
      const int want = factor_var_type(nf.type, k);
      if (S->vars[it->second].type != want) {
        S->err = "factor/variable type mismatch";
        return LMGPU_INVALID;
      }
      f.v[k] = it->second;
      markedKeys.insert(nf.k[k]);  // 3. markedKeys = keys of the new factors (:199-228)
    }
    int bi = -1;
    for (size_t b = 0; b < S->bkts.size(); b++)
      if (S->bkts[b].type == nf.type && S->bkts[b].noise_kind == nf.noise_kind && S->bkts[b].robust == nf.robust && S->bkts[b].rk == nf.rk) bi = (int)b;
    if (bi < 0) {
      lmgpu_isam2::Bkt b;
      b.type = nf.type;
      b.noise_kind = nf.noise_kind;
      b.robust = nf.robust;
      b.rk = nf.rk;
      b.rows = kFactorRows[nf.type];
      b.ar = ar;
      b.ml = kFactorMeas[nf.type];
      b.nl = nf.noise_kind == LMGPU_N_DIAG ? b.rows : (nf.noise_kind == LMGPU_N_GAUSS ? b.rows * b.rows : 0);
      b.cols = 1;
      for (int k = 0; k < ar; k++) b.cols += kVarDim[factor_var_type(nf.type, k)];
      bi = (int)S->bkts.size();
      S->bkts.push_back(b);
    }
    lmgpu_isam2::Bkt& b = S->bkts[bi];
    if (b.n + 1 > b.cap) {  // grow the bucket: descriptor arrays and its Jacobian region in the pool
      const size_t ncap = std::max<size_t>(64, (size_t)b.cap * 2);
      if ((rc = is_realloc(S, &b.d_vidx, ncap * b.ar, (size_t)b.n * b.ar))) return rc;
      if ((rc = is_realloc(S, &b.d_epos, ncap, (size_t)b.n))) return rc;
      if ((rc = is_realloc(S, &b.d_meas, ncap * b.ml, (size_t)b.n * b.ml))) return rc;
      if ((rc = is_realloc(S, &b.d_noise, ncap * std::max(1, b.nl), (size_t)b.n * b.nl))) return rc;
      int64_t noff;
      if ((rc = is_pool_alloc(S, ncap * b.rows * b.cols, &noff))) return rc;
      if (b.joff >= 0) {
        ISCHECK(hipMemcpyAsync(S->pool + noff, S->pool + b.joff, (size_t)b.n * b.rows * b.cols * sizeof(double), hipMemcpyDeviceToDevice, S->stream));
        ISCHECK(hipStreamSynchronize(S->stream));
        is_pool_free(S, b.joff, (size_t)b.cap * b.rows * b.cols);
      }
      b.joff = noff;
      b.cap = (int)ncap;
    }
    NewRows& nr = new_rows[bi];
    if (nr.first < 0) nr.first = b.n;
    for (int k = 0; k < ar; k++) nr.vidx.push_back(S->vars[f.v[k]].tidx);
    nr.epos.push_back(1 + (int32_t)slot);
    nr.meas.insert(nr.meas.end(), nf.meas.begin(), nf.meas.begin() + b.ml);
    if (b.nl) nr.noise.insert(nr.noise.end(), nf.noise.begin(), nf.noise.begin() + b.nl);
    f.bucket = bi;
    f.lidx = b.n++;
    new_by_bucket[bi].push_back(f.lidx);
    if (slot == S->facs.size())
      S->facs.push_back(f);
    else
      S->facs[slot] = f;
    new_idx.push_back((int32_t)slot);
  }
  for (auto& kv : new_rows) {  // (a bucket that grew in the loop moved its old rows only; the new ones arrive here)
    lmgpu_isam2::Bkt& b = S->bkts[kv.first];
    const NewRows& nr = kv.second;
    if ((rc = is_push(S, b.d_vidx + (size_t)nr.first * b.ar, nr.vidx.data(), nr.vidx.size() * sizeof(int32_t)))) return rc;
    if ((rc = is_push(S, b.d_epos + nr.first, nr.epos.data(), nr.epos.size() * sizeof(int32_t)))) return rc;
    if ((rc = is_push(S, b.d_meas + (size_t)nr.first * b.ml, nr.meas.data(), nr.meas.size() * sizeof(double)))) return rc;
    if (b.nl && (rc = is_push(S, b.d_noise + (size_t)nr.first * b.nl, nr.noise.data(), nr.noise.size() * sizeof(double)))) return rc;
  }
  // the removals of pushBackFactors (ISAM2-impl.h:157-172): the slot empties, the variable index forgets the factor
  std::set<uint64_t> keysWithRemoved, newFactorKeys = markedKeys, unusedKeys;
  for (uint64_t idx : up.remove) {
    lmgpu_isam2::Fac& f = S->facs[idx];
    if (f.removed) continue;
    for (int32_t v : is_fac_vids(S, f)) {
      keysWithRemoved.insert(S->vars[v].key);
      std::vector<int32_t>& entries = S->vindex[v];
      entries.erase(std::find(entries.begin(), entries.end(), (int32_t)idx));
    }
    if ((rc = is_empty_slot(S, (int32_t)idx))) return rc;
  }
  // computeUnusedKeys (:175-190): keys whose last factor went and which no new factor mentions
  for (uint64_t k : keysWithRemoved)
    if (S->vindex[S->vid_of.at(k)].empty() && !newFactorKeys.count(k)) unusedKeys.insert(k);
  // 2. errorBefore (ISAM2.cpp:444-446): the graph with the new factors at calculateEstimate(), which brings delta up to date first
  if (S->evaluate_error) {
    if (S->any_replaced && (rc = is_update_delta(S, false))) return rc;
    if ((rc = is_graph_error(S, true, &S->error_before))) return rc;
  }
  // gatherInvolvedKeys (:199-226) + updateKeys (:228-244)
  markedKeys.insert(keysWithRemoved.begin(), keysWithRemoved.end());
  markedKeys.insert(up.extra_reelim.begin(), up.extra_reelim.end());
  std::vector<int32_t> observed;  // observedKeys = the marked keys that stay in the system, ascending by key
  for (uint64_t k : markedKeys)
    if (!unusedKeys.count(k)) observed.push_back(S->vid_of.at(k));
  std::set<int32_t> relin;  // relinKeys as vids
  if (relinNeeded) {
    // ---- 4. CheckRelinearizationFull (:353-383) on the delta just updated
    // (is_update_delta above brought delta to the host with its own wait; variables added by this update are not in it: delta = 0)
    const double* hdelta = S->h_delta;
    const int ntot_checked = relin_ntot;
    // gatherRelinearizeKeys :367-399.  above(): double -> infinity norm >= threshold (:287, 361); FastMap<char, Vector> -> any
    // |delta_i| > threshold_i of the vector registered for the key's Symbol character (:252-268, 365-377)
    std::set<uint64_t> noRelin(up.no_relin.begin(), up.no_relin.end());
    bool bad_threshold = false;
    auto above = [&](int32_t v) {
      const lmgpu_isam2::Var& var = S->vars[v];
      const int dim = kVarDim[var.type];
      if (var.xoff >= ntot_checked) return false;  // added by this update: delta = 0 (never above a positive threshold; see below for 0)
      if (up.force_full_solve) return true;
      if (S->relin_thresholds.empty()) {
        double m = 0;
        for (int d = 0; d < dim; d++) m = std::max(m, std::fabs(hdelta[var.xoff + d]));
        return m >= S->prm.relinearizeThreshold;
      }
      auto it = S->relin_thresholds.find((unsigned char)(var.key >> 56));
      if (it == S->relin_thresholds.end() || (int)it->second.size() != dim) {
        bad_threshold = true;
        return false;
      }
      for (int d = 0; d < dim; d++)
        if (std::fabs(hdelta[var.xoff + d]) > it->second[d]) return true;
      return false;
    };
    std::set<int32_t> cand;
    if (S->partial_relin_check && !up.force_full_solve) {
      // CheckRelinearizationPartial :246-331: from the roots down, every key of a clique's conditional (frontals and parents) is
      // checked, the children only when one of them was above its threshold
      std::vector<int32_t> stack(S->roots.rbegin(), S->roots.rend());
      while (!stack.empty()) {
        const lmgpu_isam2::Clq& c = S->clq[stack.back()];
        stack.pop_back();
        bool any = false;
        for (int32_t v : c.vars)
          if (above(v)) {
            cand.insert(v);
            any = true;
          }
        if (any)
          for (auto ch = c.children.rbegin(); ch != c.children.rend(); ++ch) stack.push_back(*ch);
      }
    } else {
      for (size_t v = 0; v < S->vars.size(); v++)
        if (!S->vars[v].dead && above((int32_t)v)) cand.insert((int32_t)v);
    }
    if (bad_threshold) {
      S->err = "ISAM2: relinearization threshold vector missing for a Symbol character or of the wrong dimension (ISAM2-impl.h:258-262)";
      return LMGPU_INVALID;
    }
    for (int32_t v : cand)  // minus the keys whose linearization point is fixed (marginal factors) and the caller's noRelinKeys (ISAM2-impl.h:385-392)
      if (!noRelin.count(S->vars[v].key) && !S->fixed.count(S->vars[v].key)) {
        relin.insert(v);
        markedKeys.insert(S->vars[v].key);
      }
    if (!relin.empty()) {
      // ---- 5. findFluid (:431-451): cliques whose separator holds a relinearized variable
      for (const lmgpu_isam2::Clq& c : S->clq) {
        if (!c.alive) continue;
        bool found = false;
        for (size_t k = c.nfv; k < c.vars.size() && !found; k++) found = relin.count(c.vars[k]) > 0;
        if (found)
          for (int k = 0; k < c.nfv; k++) markedKeys.insert(S->vars[c.vars[k]].key);
      }
      // ---- 6. theta_.retractMasked(delta_, relinKeys) on the device
      std::vector<int32_t> sel[kNumVarTypes];
      for (int32_t v : relin) sel[S->vars[v].type].push_back(S->vars[v].tidx);
      {  // one launch for all the types
        RetractMulti rm{};
        int ne = 0, maxn = 0;
        for (int t = 0; t < kNumVarTypes; t++) {
          if (sel[t].empty()) continue;
          int32_t* d = nullptr;
          if ((rc = is_stage(S, sel[t], &d))) return rc;
          rm.type[ne] = t;
          rm.n[ne] = (int)sel[t].size();
          rm.cur[ne] = S->theta[t];
          rm.out[ne] = S->theta[t];
          rm.xoff[ne] = S->d_type_xoff[t];
          rm.sel[ne] = d;
          maxn = std::max(maxn, (int)sel[t].size());
          ne++;
        }
        if (ne > 0) {
          if ((rc = is_flush(S))) return rc;
          hipLaunchKernelGGL(retract_multi_kernel, dim3((maxn + 255) / 256, ne), dim3(256), 0, S->stream, rm, (const double*)S->delta);
        }
      }
    }
    res.variablesRelinearized = (int32_t)markedKeys.size();
  }
  // ---- 7. linearizeNewFactors (:454-468) + augmentVariableIndex
  {
    std::vector<std::pair<int, std::vector<int32_t>>> jobs(new_by_bucket.begin(), new_by_bucket.end());
    if ((rc = is_linearize_jobs(S, jobs))) return rc;
  }
  for (int32_t i : new_idx)
    for (int k = 0; k < kFactorArity[S->facs[i].type]; k++) S->vindex[S->facs[i].v[k]].push_back(i);
  lap(1);  // new factors, relinearization check, retract, linearize
  // ---- 8. recalculate (ISAM2.cpp:117-175)
  if (!markedKeys.empty()) {
    std::vector<int> bn;
    std::list<int> orphans;
    for (uint64_t k : markedKeys) {  // removeTop
      const int node = S->node_of[S->vid_of.at(k)];
      if (node >= 0) is_remove_path(S, node, &bn, &orphans);
    }
    std::vector<int32_t> affected;  // affectedKeys: frontals of the removed conditionals
    for (int id : bn)
      for (int k = 0; k < S->clq[id].nfv; k++) affected.push_back(S->clq[id].vars[k]);
    for (int id : bn) is_release_clique(S, id);

    std::set<int32_t> affectedSet;
    // deltaReplacedMask_ |= affectedKeysSet: byte fills carried by the scatter kernel of the flush in front of the elimination
    auto push_marks = [&]() {
      for (int32_t v : affectedSet) {
        if (unusedKeys.count(S->vars[v].key)) continue;  // leaves the system below
        S->replaced[v] = 1;
        S->pushes.push_back(lmgpu_isam2::PushRec{S->d_replaced + S->vars[v].xoff, (const void*)(uintptr_t)S->epoch, (uint32_t)kVarDim[S->vars[v].type], 2u});
      }
    };
    if ((double)affected.size() >= (double)S->vid_of.size() * 0.65) {
      // ---- recalculateBatch :178-247: reorder, relinearize and re-eliminate everything
      res.batch = 1;
      for (int id = 0; id < (int)S->clq.size(); id++)
        if (S->clq[id].alive) is_release_clique(S, id);
      S->roots.clear();
      std::fill(S->node_of.begin(), S->node_of.end(), -1);
      std::vector<int32_t> vids;
      for (auto& kv : S->vid_of)  // the keys of variableIndex_ (ISAM2.cpp:186-190): a variable that has a value but no factor yet stays out of the tree
        if (!unusedKeys.count(kv.first) && !S->vindex[kv.second].empty()) vids.push_back(kv.second);
      std::vector<std::vector<int32_t>> cols;
      for (int32_t v : vids) cols.push_back(S->vindex[v]);
      std::map<int32_t, int> groups;
      if (up.has_constrained) {
        for (auto& kg : up.constrained) {
          auto it = S->vid_of.find(kg.first);
          if (it == S->vid_of.end() || unusedKeys.count(kg.first)) {
            S->err = "ISAM2: constrainedKeys names a variable that is not in the system";
            return LMGPU_INVALID;
          }
          groups[it->second] = kg.second;
        }
      } else if (S->vid_of.size() > observed.size()) {
        for (int32_t v : observed) groups[v] = 1;
      }
      std::vector<int32_t> perm;
      lap(2);
      if ((rc = is_colamd(S, vids, cols, (int)S->facs.size(), groups, &perm))) return rc;
      lap(3);
      {
        std::vector<std::pair<int, std::vector<int32_t>>> jobs;
        for (size_t b = 0; b < S->bkts.size(); b++) {
          std::vector<int32_t> all(S->bkts[b].n);
          for (int i = 0; i < S->bkts[b].n; i++) all[i] = i;
          jobs.emplace_back((int)b, std::move(all));
        }
        if ((rc = is_linearize_jobs(S, jobs))) return rc;
      }
      std::vector<IsGF> gfs(S->facs.size());
      for (size_t i = 0; i < S->facs.size(); i++) {
        gfs[i].kind = 0;
        gfs[i].id = (int32_t)i;
        if (S->facs[i].removed) continue;  // an empty slot: an entry without variables, so that positions stay factor indices
        if (S->facs[i].marg >= 0) gfs[i].kind = 3;
        gfs[i].vids = is_fac_vids(S, S->facs[i]);
      }
      lap(1);  // (batch: relinearization of everything counts with the linearize phase)
      for (int32_t v : vids) affectedSet.insert(v);
      push_marks();
      if ((rc = is_eliminate(S, gfs, vids, perm, &cols))) return rc;  // (variableIndex_ as it stands: GaussianEliminationTree(*linearized, affectedFactorsVarIndex, order))
      lap(4);
      res.variablesReeliminated = (int32_t)vids.size();
      res.factorsRecalculated = (int32_t)S->facs.size();
    } else {
      // ---- recalculateIncremental :250-362
      std::vector<int32_t> affectedAndNew = affected;
      affectedAndNew.insert(affectedAndNew.end(), observed.begin(), observed.end());
      const std::set<int32_t> inSet(affectedAndNew.begin(), affectedAndNew.end());
      std::set<int32_t> candidates;  // relinearizeAffectedFactors :66-114
      for (int32_t v : affectedAndNew)
        for (int32_t f : S->vindex[v]) candidates.insert(f);
      std::vector<IsGF> gfs;
      std::map<int, std::vector<int32_t>> relin_by_bucket;
      for (int32_t idx : candidates) {
        const lmgpu_isam2::Fac& f = S->facs[idx];
        bool inside = true, useCached = true;
        IsGF g;
        g.vids = is_fac_vids(S, f);
        for (int32_t v : g.vids) {
          if (!inSet.count(v)) {
            inside = false;
            break;
          }
          if (relin.count(v)) useCached = false;
        }
        if (!inside) continue;
        if (!useCached && f.marg < 0) relin_by_bucket[f.bucket].push_back(f.lidx);  // (a marginal factor linearizes to itself)
        g.kind = f.marg >= 0 ? 3 : 0;
        g.id = idx;
        gfs.push_back(std::move(g));
      }
      {
        std::vector<std::pair<int, std::vector<int32_t>>> jobs(relin_by_bucket.begin(), relin_by_bucket.end());
        if ((rc = is_linearize_jobs(S, jobs))) return rc;
      }
      res.variablesReeliminated = (int32_t)affectedAndNew.size();
      res.factorsRecalculated = (int32_t)gfs.size();
      for (int kind = 1; kind <= 2; kind++)  // GetCachedBoundaryFactors (ISAM2-impl.h:499-509), then the orphan wrappers
        for (int o : orphans) {
          IsGF g;
          g.kind = kind;
          g.id = o;
          g.vids.assign(S->clq[o].vars.begin() + S->clq[o].nfv, S->clq[o].vars.end());
          gfs.push_back(std::move(g));
        }
      for (uint64_t k : markedKeys) affectedSet.insert(S->vid_of.at(k));
      for (int32_t v : affected) affectedSet.insert(v);
      // VariableIndex of `gfs`, ascending by key
      std::map<uint64_t, std::vector<int32_t>> vi;
      for (size_t i = 0; i < gfs.size(); i++)
        for (int32_t v : gfs[i].vids) vi[S->vars[v].key].push_back((int32_t)i);
      std::vector<int32_t> vids;
      std::vector<std::vector<int32_t>> cols;
      for (auto& kv : vi) {
        vids.push_back(S->vid_of.at(kv.first));
        cols.push_back(kv.second);
      }
      std::map<int32_t, int> groups;  // constraint groups, minus unused / unaffected keys (ISAM2.cpp:318-340)
      if (up.has_constrained) {
        for (auto& kg : up.constrained) {
          auto it = S->vid_of.find(kg.first);
          if (it != S->vid_of.end() && !unusedKeys.count(kg.first) && affectedSet.count(it->second)) groups.emplace(it->second, kg.second);
        }
      } else {
        const int group = observed.size() < vids.size() ? 1 : 0;
        for (int32_t v : observed)
          if (affectedSet.count(v)) groups.emplace(v, group);
      }
      std::vector<int32_t> perm;
      lap(2);  // removeTop, affected factors, variable index
      if ((rc = is_colamd(S, vids, cols, (int)gfs.size(), groups, &perm))) return rc;
      lap(3);  // constrained COLAMD callback
      push_marks();
      if ((rc = is_eliminate(S, gfs, vids, perm))) return rc;
      lap(4);  // symbolic elimination, tables, launches
    }
    S->any_replaced = S->any_replaced || !affectedSet.empty();
  }
  // ---- removeVariables (ISAM2.cpp:385-398): the variable leaves theta / delta / the variable index; its storage is not reused
  for (uint64_t k : unusedKeys) {
    const int32_t v = S->vid_of.at(k);
    S->vars[v].dead = true;
    if (S->dogleg) {  // the vectors of the dog leg are summed over all scalars: a retired variable must not count
      const size_t o = (size_t)S->vars[v].xoff, nb = (size_t)kVarDim[S->vars[v].type] * sizeof(double);
      ISCHECK(hipMemsetAsync(S->delta + o, 0, nb, S->stream));
      ISCHECK(hipMemsetAsync(S->delta_newton + o, 0, nb, S->stream));
      ISCHECK(hipMemsetAsync(S->rgprod + o, 0, nb, S->stream));
    }
    S->replaced[v] = 0;
    S->node_of[v] = -1;
    S->vid_of.erase(k);
    S->fixed.erase(k);
  }
  S->last_unused.assign(unusedKeys.begin(), unusedKeys.end());
  res.cliques = S->n_alive;  // (every alive clique hangs in the tree again by now; counting them by a walk was O(cliques) per update)
  if (result) *result = res;
  lap(5);
  bool walked = false;
  if (!S->dogleg && S->any_replaced && !dev_switch("LMGPU_ISAM2_NO_PREWALK")) {
    const bool next_relin = S->prm.enableRelinearization && S->prm.relinearizeSkip > 0 && (S->update_count + 1) % S->prm.relinearizeSkip == 0;
    if (next_relin || reads_before) {
      if ((rc = is_update_delta_enqueue(S, false, next_relin))) return rc;
      walked = true;
    }
  }
  rc = is_finish_elimination(S);  // the one wait of an update
  if (rc == LMGPU_OK && walked) rc = is_update_delta_finish(S);  // (an indeterminate back-substitution is reported one call early)
  lap(6);
  if (rc == LMGPU_OK && S->evaluate_error) {  // errorAfter (ISAM2.cpp:481-483), again through calculateEstimate()
    if (S->any_replaced && (rc = is_update_delta(S, false))) return rc;
    rc = is_graph_error(S, true, &S->error_after);
  }
  return rc;
}

// ISAM2::marginalizeLeaves (gtsam/nonlinear/ISAM2.cpp:487-720).  The host walks the tree and the variable index exactly as the reference
// does; the numbers stay on the device:
//   * a clique that goes entirely hands its cached factor (its update matrix, already in the pool) to its parent as a marginal factor --
//     the block changes owner, nothing is copied (:556-571);
//   * a clique that loses its leading frontals gets ONE front eliminated over them -- the cached factors of the removed children, the
//     linear factors the leaving variables pull in (the linearization cache: it is at theta, :600-622) and marginal factors of earlier
//     calls -- by the same front kernels an update uses; the conditional is thrown away, the update matrix is the marginal (:624-637);
//     its [R S d] loses the leading rows and columns (the reference re-points its block matrix, :639-653; here the remaining block is
//     copied into storage of its own shape, so that every kernel keeps seeing dense cliques);
//   * the factors the marginals summarise leave the graph, the marginals enter it as LinearContainerFactors (slots as add_factors gives
//     them, findUnusedFactorSlots honoured), their keys become fixedVariables_, the leaves leave theta / delta (:669-712).
// What the reference only checks in debug builds (a key that is not a leaf leaves the object "in an inconsistent state", :516-523) is
// checked here BEFORE anything changes, and refused.
int is_marginalize_leaves(lmgpu_isam2* S, const std::vector<uint64_t>& leafList) {
  S->last_marginal_idx.clear();
  S->last_deleted_idx.clear();
  if (!S->new_vars.empty() || !S->new_facs.empty()) {
    S->err = "ISAM2::marginalizeLeaves: variables or factors are waiting for an update";
    return LMGPU_INVALID;
  }
  S->walk_prepared = false;  // (cliques and roots change below: the next walk seeds itself)
  const std::set<uint64_t> leafKeys(leafList.begin(), leafList.end());
  std::set<int32_t> leafV;
  for (uint64_t k : leafKeys) {
    auto it = S->vid_of.find(k);
    if (it == S->vid_of.end() || S->node_of[it->second] < 0) {
      S->failed_key = k;
      S->err = "ISAM2::marginalizeLeaves: a key is not a variable of the Bayes tree";
      return LMGPU_INVALID;
    }
    leafV.insert(it->second);
  }
  if (leafV.empty()) return LMGPU_OK;
  // ---- the plan: the reference's loop (:530-667) without touching anything
  struct Act {
    bool whole;
    int clique;
    std::vector<int> subtrees;  // partial: the children that hang on a leaving variable
  };
  std::vector<Act> plan;
  std::set<int32_t> goneV;
  auto not_leaf = [&](int32_t v) {
    S->failed_key = S->vars[v].key;
    S->err = "ISAM2::marginalizeLeaves: requesting to marginalize variables that are not leaves (a variable that stays is eliminated before this one)";
    return LMGPU_INVALID;
  };
  auto plan_subtree = [&](int root) -> int {  // every frontal below must leave too
    std::vector<int> q{root};
    for (size_t i = 0; i < q.size(); i++) {
      const lmgpu_isam2::Clq& c = S->clq[q[i]];
      for (int k = 0; k < c.nfv; k++) {
        if (!leafV.count(c.vars[k])) return not_leaf(c.vars[k]);
        goneV.insert(c.vars[k]);
      }
      q.insert(q.end(), c.children.begin(), c.children.end());
    }
    return LMGPU_OK;
  };
  int rc;
  for (uint64_t key : leafKeys) {
    const int32_t v = S->vid_of.at(key);
    if (goneV.count(v)) continue;
    int id = S->node_of[v];
    while (S->clq[id].parent >= 0) {  // up to the root of the marginalized subtree: only the first variable of the parent needs a look
      const int32_t pf = S->clq[S->clq[id].parent].vars[0];
      if (leafV.count(pf) && !goneV.count(pf))
        id = S->clq[id].parent;
      else
        break;
    }
    const lmgpu_isam2::Clq& c = S->clq[id];
    int nleaf = 0;
    for (int k = 0; k < c.nfv; k++) nleaf += leafV.count(c.vars[k]) && !goneV.count(c.vars[k]);
    Act a{nleaf == c.nfv, id, {}};
    if (a.whole) {
      if ((rc = plan_subtree(id))) return rc;
    } else {
      for (int k = 0; k < c.nfv; k++)  // the leaving frontals lead the clique (:639-642 counts them from the front)
        if ((k < nleaf) != (leafV.count(c.vars[k]) > 0)) return not_leaf(c.vars[k < nleaf ? k : nleaf]);
      for (int ch : c.children) {
        bool hangs = false;
        for (size_t k = S->clq[ch].nfv; k < S->clq[ch].vars.size() && !hangs; k++) hangs = leafV.count(S->clq[ch].vars[k]) > 0;
        if (!hangs) continue;
        a.subtrees.push_back(ch);
        if ((rc = plan_subtree(ch))) return rc;
      }
      for (int k = 0; k < nleaf; k++) goneV.insert(c.vars[k]);
    }
    plan.push_back(std::move(a));
  }
  S->failed_key = 0;
  // ---- carry it out
  if ((rc = is_stage_begin(S))) return rc;
  S->elim_pending = false;
  std::map<uint64_t, std::vector<lmgpu_isam2::Marg>> marginalFactors;  // front key of a clique -> the marginals passed up to it
  std::set<int32_t> factorIndicesToRemove;
  std::vector<int> deferred;  // removed cliques whose update matrices a launch of this call still reads
  auto drop_marginals = [&](uint64_t frontKey) {
    auto it = marginalFactors.find(frontKey);
    if (it == marginalFactors.end()) return;
    for (lmgpu_isam2::Marg& m : it->second) is_pool_free(S, m.blk_off, m.blk_n);
    marginalFactors.erase(it);
  };
  // BayesTree::removeSubtree (BayesTree-inst.h:512-547) + the bookkeeping of trackingRemoveSubtree (:506-527); returns the cliques, root first
  auto remove_subtree = [&](int root) {
    lmgpu_isam2::Clq& r = S->clq[root];
    if (r.parent >= 0) {
      auto& pc = S->clq[r.parent].children;
      pc.erase(std::find(pc.begin(), pc.end(), root));
      S->touched.push_back(r.parent);
    } else {
      S->roots.erase(std::find(S->roots.begin(), S->roots.end(), root));
    }
    r.parent = -1;
    std::vector<int> q{root};
    for (size_t i = 0; i < q.size(); i++) {
      const lmgpu_isam2::Clq& c = S->clq[q[i]];
      q.insert(q.end(), c.children.begin(), c.children.end());
      drop_marginals(S->vars[c.vars[0]].key);
      for (int k = 0; k < c.nfv; k++) {
        S->node_of[c.vars[k]] = -1;
        factorIndicesToRemove.insert(S->vindex[c.vars[k]].begin(), S->vindex[c.vars[k]].end());
      }
    }
    return q;
  };
  for (const Act& a : plan) {
    if (a.whole) {
      const int parent = S->clq[a.clique].parent;
      const uint64_t parentFront = parent >= 0 ? S->vars[S->clq[parent].vars[0]].key : 0;
      const std::vector<int> gone = remove_subtree(a.clique);
      for (int id : gone) {
        if (id == a.clique && parent >= 0) {  // its cached factor belongs to the parent from now on; a root's is dropped (:559-568)
          lmgpu_isam2::Marg m;
          is_release_clique(S, id, &m);
          marginalFactors[parentFront].push_back(std::move(m));
        } else {
          is_release_clique(S, id);
        }
      }
      continue;
    }
    // ---- part of a clique: one front over its leaving frontals
    const int id = a.clique;
    std::vector<IsGF> gfs;
    for (int ch : a.subtrees) {  // the child marginals (:583-592)
      IsGF g;
      g.kind = 1;
      g.id = ch;
      g.vids.assign(S->clq[ch].vars.begin() + S->clq[ch].nfv, S->clq[ch].vars.end());
      gfs.push_back(std::move(g));
    }
    std::set<int32_t> pulled;  // factors the leaving frontals pull in, minus those of the subtrees removed at this step (:600-622)
    std::vector<int32_t> leaving;
    for (int k = 0; k < S->clq[id].nfv && leafV.count(S->clq[id].vars[k]); k++) {
      leaving.push_back(S->clq[id].vars[k]);
      pulled.insert(S->vindex[S->clq[id].vars[k]].begin(), S->vindex[S->clq[id].vars[k]].end());
    }
    for (int ch : a.subtrees)
      for (int gid : remove_subtree(ch)) {
        for (int k = 0; k < S->clq[gid].nfv; k++)
          for (int32_t f : S->vindex[S->clq[gid].vars[k]]) pulled.erase(f);
        deferred.push_back(gid);
      }
    for (int32_t f : pulled) {
      IsGF g;
      g.kind = S->facs[f].marg >= 0 ? 3 : 0;
      g.id = f;
      g.vids = is_fac_vids(S, S->facs[f]);
      gfs.push_back(std::move(g));
    }
    // the front: the leaving frontals ascending by key (Ordering(cliqueFrontalsToEliminate), :627-633), then everything else they touch
    auto by_key = [&](int32_t x, int32_t y) { return S->vars[x].key < S->vars[y].key; };
    std::sort(leaving.begin(), leaving.end(), by_key);
    std::set<int32_t> others;
    for (const IsGF& g : gfs)
      for (int32_t v : g.vids)
        if (!std::count(leaving.begin(), leaving.end(), v)) others.insert(v);
    std::vector<int32_t> sep(others.begin(), others.end());
    std::sort(sep.begin(), sep.end(), by_key);
    std::vector<int32_t> vid_of_slot = leaving;
    vid_of_slot.insert(vid_of_slot.end(), sep.begin(), sep.end());
    SymbolicFronts sf;
    sf.fronts.resize(1);
    for (size_t i = 0; i < leaving.size(); i++) sf.fronts[0].frontals.push_back((int32_t)i);
    for (size_t i = 0; i < sep.size(); i++) sf.fronts[0].sep.push_back((int32_t)(leaving.size() + i));
    for (size_t i = 0; i < gfs.size(); i++) sf.fronts[0].factors.push_back((int32_t)i);
    sf.roots.push_back(0);
    const uint64_t frontKey = S->vars[S->clq[id].vars[0]].key;  // (the marginal is filed under the front key the clique has BEFORE the split, :637)
    std::vector<int> tmp;
    if ((rc = is_eliminate_fronts(S, gfs, vid_of_slot, sf, false, &tmp))) return rc;
    if ((rc = is_finish_elimination(S))) return rc;
    {
      lmgpu_isam2::Marg m;
      is_release_clique(S, tmp[0], &m);
      marginalFactors[frontKey].push_back(std::move(m));
    }
    // ---- split the clique (:639-653): the leading `leaving` variables go, the conditional on the rest stays as it is
    lmgpu_isam2::Clq& c = S->clq[id];
    int dimToRemove = 0;
    for (int32_t v : leaving) dimToRemove += kVarDim[S->vars[v].type];
    const int n0 = c.n, nf0 = c.nf, n1 = n0 - dimToRemove, nf1 = nf0 - dimToRemove, m1 = n1 - nf1;
    const int ld0 = c.ld > 0 ? c.ld : n0;
    const int64_t src = c.rsd_off + (int64_t)dimToRemove * ld0 + dimToRemove;
    if (n1 > kLdsLimitN) {  // stays a dense-front clique: one n1 x ld1 block
      const int ld1 = (n1 + 15) & ~15;
      int64_t f1;
      if ((rc = is_pool_alloc(S, (size_t)n1 * ld1, &f1))) return rc;
      ISCHECK(hipMemsetAsync(S->pool + f1, 0, (size_t)n1 * ld1 * sizeof(double), S->stream));
      ISCHECK(hipMemcpy2DAsync(S->pool + f1, (size_t)ld1 * sizeof(double), S->pool + src, (size_t)ld0 * sizeof(double), (size_t)n1 * sizeof(double), (size_t)n1,
                               hipMemcpyDeviceToDevice, S->stream));
      is_pool_free(S, c.f_off, (size_t)n0 * c.ld);
      c.f_off = f1;
      c.ld = ld1;
      c.rsd_off = f1;
      c.u_off = f1 + (int64_t)nf1 * ld1 + nf1;
    } else {
      int64_t r1;
      if ((rc = is_pool_alloc(S, (size_t)nf1 * n1, &r1))) return rc;
      ISCHECK(hipMemcpy2DAsync(S->pool + r1, (size_t)n1 * sizeof(double), S->pool + src, (size_t)ld0 * sizeof(double), (size_t)n1 * sizeof(double), (size_t)nf1,
                               hipMemcpyDeviceToDevice, S->stream));
      if (c.ld > 0) {  // it now fits an LDS front: [R S d] and the update matrix in dense blocks of their own
        int64_t u1;
        if ((rc = is_pool_alloc(S, (size_t)m1 * m1, &u1))) return rc;
        ISCHECK(hipMemcpy2DAsync(S->pool + u1, (size_t)m1 * sizeof(double), S->pool + c.u_off, (size_t)c.ld * sizeof(double), (size_t)m1 * sizeof(double), (size_t)m1,
                                 hipMemcpyDeviceToDevice, S->stream));
        is_pool_free(S, c.f_off, (size_t)n0 * c.ld);
        c.f_off = -1;
        c.ld = 0;
        c.u_off = u1;
      } else {
        is_pool_free(S, c.rsd_off, (size_t)nf0 * n0);
      }
      c.rsd_off = r1;
    }
    if (c.xrow_off >= 0) is_pool_free(S, c.xrow_off, (size_t)n0 / 2 + 1);
    c.xrow_off = -1;
    for (int32_t v : leaving) {
      S->node_of[v] = -1;
      factorIndicesToRemove.insert(S->vindex[v].begin(), S->vindex[v].end());
    }
    c.vars.erase(c.vars.begin(), c.vars.begin() + leaving.size());
    c.nfv -= (int)leaving.size();
    c.nf = nf1;
    c.n = n1;
    S->touched.push_back(id);
  }
  for (int id : deferred) is_release_clique(S, id);
  // ---- the factors the marginals summarise leave the graph and the variable index (:672-682)
  for (int32_t idx : factorIndicesToRemove) {
    for (int32_t v : is_fac_vids(S, S->facs[idx])) {
      std::vector<int32_t>& entries = S->vindex[v];
      entries.erase(std::find(entries.begin(), entries.end(), idx));
    }
    if ((rc = is_empty_slot(S, idx))) return rc;
  }
  // ---- the marginals enter it (:684-709), in the order of the map (ascending front key), slots as FactorGraph::add_factors gives them
  size_t slot_scan = 0;
  for (auto& kf : marginalFactors)
    for (lmgpu_isam2::Marg& m : kf.second) {
      size_t slot = S->facs.size();
      if (S->find_unused_slots) {
        while (slot_scan < S->facs.size() && !S->facs[slot_scan].removed) ++slot_scan;
        slot = slot_scan;
      }
      lmgpu_isam2::Fac f{-1, -1, -1, {-1, -1, -1}, false, -1};
      for (int32_t v : m.vids) {
        S->fixed.insert(S->vars[v].key);
        S->vindex[v].push_back((int32_t)slot);
      }
      f.marg = is_new_marg(S, m);
      if (slot == S->facs.size())
        S->facs.push_back(f);
      else
        S->facs[slot] = f;
      S->last_marginal_idx.push_back((uint64_t)slot);
    }
  // ---- removeVariables(leafKeys) (:385-398, 712)
  for (int32_t v : leafV) {
    S->vars[v].dead = true;
    S->vindex[v].clear();
    if (S->dogleg) {
      const size_t o = (size_t)S->vars[v].xoff, nb = (size_t)kVarDim[S->vars[v].type] * sizeof(double);
      ISCHECK(hipMemsetAsync(S->delta + o, 0, nb, S->stream));
      ISCHECK(hipMemsetAsync(S->delta_newton + o, 0, nb, S->stream));
      ISCHECK(hipMemsetAsync(S->rgprod + o, 0, nb, S->stream));
    }
    S->replaced[v] = 0;
    S->node_of[v] = -1;
    S->fixed.erase(S->vars[v].key);
    S->vid_of.erase(S->vars[v].key);
  }
  S->last_deleted_idx.assign(factorIndicesToRemove.begin(), factorIndicesToRemove.end());
  return is_finish_elimination(S);
}

}  // namespace

// =============================================================================================== C ABI (ISAM2)
extern "C" {

int lmgpu_isam2_create(const lmgpu_config* cfg, const lmgpu_isam2_params* prm, lmgpu_ccolamd_fn ccolamd, void* user, lmgpu_isam2** out) {
  if (!cfg || !prm || !out || !ccolamd) return LMGPU_INVALID;
  lmgpu_isam2* S = new lmgpu_isam2();
  S->cfg = *cfg;
  S->prm = *prm;
  S->ccolamd = ccolamd;
  S->user = user;
  S->device = cfg->device;
  S->trace = getenv("LMGPU_ISAM2_TRACE") != nullptr;
  *out = S;
  if (S->device < 0) {
    S->err = "no HIP device bound to this handle; the incremental path has no CPU fallback";
    return LMGPU_HIP_ERROR;
  }
  ISCHECK(hipSetDevice(S->device));
  ISCHECK(hipStreamCreate(&S->stream));
  ISCHECK(hipMalloc((void**)&S->d_status, 2 * sizeof(int)));  // [0] eliminations (and marginalCovariance), [1] the walks
  ISCHECK(hipHostMalloc((void**)&S->h_status, 2 * sizeof(int), hipHostMallocMapped));
  ISCHECK(hipHostGetDevicePointer((void**)&S->h_status_dev, S->h_status, 0));
  ISCHECK(hipHostMalloc((void**)&S->h_val, 32 * sizeof(double), hipHostMallocMapped));
  ISCHECK(hipHostGetDevicePointer((void**)&S->h_val_dev, S->h_val, 0));
  ISCHECK(hipFuncSetAttribute((const void*)lds_front_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitN * kLdsLimitN * 8 + kLdsFrontExtra + 64));
  ISCHECK(hipFuncSetAttribute((const void*)lds_front_kernel<false, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitN * kLdsLimitN * 8 + kLdsFrontExtra + 64));
  ISCHECK(hipFuncSetAttribute((const void*)lds_front_merged_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitN * kLdsLimitN * 8 + kLdsFrontExtra + 64));
  ISCHECK(hipFuncSetAttribute((const void*)lds_front_merged_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitN * kLdsLimitN * 8 + kLdsFrontExtra + 64));
  ISCHECK(hipFuncSetAttribute((const void*)isam2_wildfire_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (kLdsLimitN * kLdsLimitN + LDSB_TAIL) * 8));
  ISCHECK(hipFuncSetAttribute((const void*)diag_potrf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DIAG_LDS_BYTES));
  ISCHECK(hipFuncSetAttribute((const void*)syrk_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kSyrkLds));
  return LMGPU_OK;
}

int lmgpu_isam2_destroy(lmgpu_isam2* S) {
  if (!S) return LMGPU_INVALID;
  if (S->trace && S->update_count > 0) {
    static const char* nm[7] = {"new variables + updateDelta", "new factors / relinearization check / linearize", "removeTop + affected factors + variable index",
                                "constrained COLAMD callback", "symbolic elimination + tables + launches", "marks + clique count", "final wait"};
    for (int i = 0; i < 7; i++) std::fprintf(stderr, "isam2 update: %-50s %8.1f us per update\n", nm[i], 1e6 * S->t_phase[i] / S->update_count);
  }
  if (S->device >= 0) {
    (void)hipSetDevice(S->device);
    for (int t = 0; t < kNumVarTypes; t++) {
      if (S->theta[t]) (void)hipFree(S->theta[t]);
      if (S->est[t]) (void)hipFree(S->est[t]);
      if (S->d_type_xoff[t]) (void)hipFree(S->d_type_xoff[t]);
    }
    for (auto& b : S->bkts) {
      if (b.d_vidx) (void)hipFree(b.d_vidx);
      if (b.d_epos) (void)hipFree(b.d_epos);
      if (b.d_meas) (void)hipFree(b.d_meas);
      if (b.d_noise) (void)hipFree(b.d_noise);
    }
    for (void* p : {(void*)S->delta, (void*)S->ones, (void*)S->d_replaced, (void*)S->d_changed, (void*)S->pool, (void*)S->d_status, (void*)S->d_gpart, (void*)S->d_tree,
                    (void*)S->d_tree_fx, (void*)S->d_tree_sx, (void*)S->d_tree_done, (void*)S->d_queue, (void*)S->d_wl, (void*)S->inv16, (void*)S->d_eticket, (void*)S->d_marg, (void*)S->d_ebuf,
                    (void*)S->d_epart, (void*)S->delta_newton, (void*)S->rgprod, (void*)S->grad, (void*)S->dx_u, (void*)S->d_cerr, (void*)S->d_dlscal})
      if (p) (void)hipFree(p);
    if (S->h_status) (void)hipHostFree(S->h_status);
    if (S->h_val) (void)hipHostFree(S->h_val);
    if (S->h_delta) (void)hipHostFree(S->h_delta);
    if (S->h_escal) (void)hipHostFree(S->h_escal);
    if (S->h_dlscal) (void)hipHostFree(S->h_dlscal);
    if (S->h_stage) (void)hipHostFree(S->h_stage);
    if (S->d_stage) (void)hipFree(S->d_stage);
    for (auto& e : S->stage_extra) {
      (void)hipHostFree(e.first);
      if (e.second) (void)hipFree(e.second);
    }
    if (S->stream) (void)hipStreamDestroy(S->stream);
  }
  delete S;
  return LMGPU_OK;
}

const char* lmgpu_isam2_last_error(const lmgpu_isam2* S) { return S ? S->err.c_str() : "null handle"; }
uint64_t lmgpu_isam2_last_failed_key(const lmgpu_isam2* S) { return S ? S->failed_key : 0; }

int lmgpu_isam2_add_variables(lmgpu_isam2* S, int32_t n, const uint64_t* keys, const int32_t* types, const double* packed_values) {
  if (!S || n < 0 || (n && (!keys || !types || !packed_values))) return LMGPU_INVALID;
  const double* p = packed_values;
  for (int i = 0; i < n; i++) {
    if (types[i] < 0 || types[i] >= LMGPU_NUM_VAR_TYPES) {
      S->err = "bad variable type";
      return LMGPU_INVALID;
    }
    lmgpu_isam2::NewVar nv{};
    nv.key = keys[i];
    nv.type = types[i];
    std::memcpy(nv.v, p, kVarStore[types[i]] * sizeof(double));
    p += kVarStore[types[i]];
    S->new_vars.push_back(nv);
  }
  return LMGPU_OK;
}

int lmgpu_isam2_add_factors(lmgpu_isam2* S, int32_t factor_type, int32_t n, const uint64_t* keys, const double* meas, int32_t noise_kind,
                            const double* noise) {
  if (!S || factor_type < 0 || factor_type >= LMGPU_NUM_FACTOR_TYPES || n < 0) return LMGPU_INVALID;
  if (noise_kind != LMGPU_N_UNIT && noise_kind != LMGPU_N_DIAG && noise_kind != LMGPU_N_GAUSS) return LMGPU_INVALID;
  if (n == 0) return LMGPU_OK;
  if (!keys || !meas || (noise_kind != LMGPU_N_UNIT && !noise)) return LMGPU_INVALID;
  const int ar = kFactorArity[factor_type], rows = kFactorRows[factor_type], ml = kFactorMeas[factor_type];
  const int nl = noise_kind == LMGPU_N_DIAG ? rows : (noise_kind == LMGPU_N_GAUSS ? rows * rows : 0);
  for (int i = 0; i < n; i++) {
    lmgpu_isam2::NewFac f;
    f.type = factor_type;
    f.noise_kind = noise_kind;
    for (int k = 0; k < 3; k++) f.k[k] = k < ar ? keys[(size_t)i * ar + k] : 0;
    f.meas.assign(meas + (size_t)i * ml, meas + (size_t)(i + 1) * ml);
    if (nl) f.noise.assign(noise + (size_t)i * nl, noise + (size_t)(i + 1) * nl);
    S->new_facs.push_back(std::move(f));
  }
  return LMGPU_OK;
}

int lmgpu_isam2_add_factors_robust(lmgpu_isam2* S, int32_t factor_type, int32_t n, const uint64_t* keys, const double* meas, int32_t noise_kind,
                                   const double* noise, int32_t robust_kind, double robust_k) {
  if (!S || robust_kind < LMGPU_ROBUST_NONE || robust_kind > LMGPU_ROBUST_L2_WITH_DEAD_ZONE) return LMGPU_INVALID;
  const size_t first = S->new_facs.size();
  const int rc = lmgpu_isam2_add_factors(S, factor_type, n, keys, meas, noise_kind, noise);
  if (rc) return rc;
  for (size_t i = first; i < S->new_facs.size(); i++) {
    S->new_facs[i].robust = robust_kind;
    S->new_facs[i].rk = robust_k;
  }
  return LMGPU_OK;
}

int lmgpu_isam2_update(lmgpu_isam2* S, int32_t force_relinearize, lmgpu_isam2_result* out) {
  if (!S) return LMGPU_INVALID;
  if (S->device < 0) {
    S->err = "no HIP device bound to this handle; the incremental path has no CPU fallback";
    return LMGPU_HIP_ERROR;
  }
  ISCHECK(hipSetDevice(S->device));
  lmgpu_isam2::UpParams up;
  up.force_relinearize = force_relinearize != 0;
  return is_update(S, up, out);
}

int lmgpu_isam2_update_with(lmgpu_isam2* S, const lmgpu_isam2_update_params* p, lmgpu_isam2_result* out) {
  if (!S || !p) return LMGPU_INVALID;
  if (p->n_remove < 0 || p->n_constrained < 0 || p->n_no_relin < 0 || p->n_extra_reelim < 0 || (p->n_remove && !p->removeFactorIndices) ||
      (p->n_constrained && (!p->constrainedKeys || !p->constrainedGroups)) || (p->n_no_relin && !p->noRelinKeys) ||
      (p->n_extra_reelim && !p->extraReelimKeys)) {
    S->err = "bad ISAM2 update parameters";
    return LMGPU_INVALID;
  }
  if (S->device < 0) {
    S->err = "no HIP device bound to this handle; the incremental path has no CPU fallback";
    return LMGPU_HIP_ERROR;
  }
  ISCHECK(hipSetDevice(S->device));
  lmgpu_isam2::UpParams up;
  up.remove.assign(p->removeFactorIndices, p->removeFactorIndices + p->n_remove);
  up.has_constrained = p->has_constrained != 0;
  for (int i = 0; i < p->n_constrained; i++) up.constrained[p->constrainedKeys[i]] = p->constrainedGroups[i];
  up.no_relin.assign(p->noRelinKeys, p->noRelinKeys + p->n_no_relin);
  up.extra_reelim.assign(p->extraReelimKeys, p->extraReelimKeys + p->n_extra_reelim);
  up.force_relinearize = p->force_relinearize != 0;
  up.force_full_solve = p->forceFullSolve != 0;
  return is_update(S, up, out);
}

int lmgpu_isam2_set_relinearize_thresholds(lmgpu_isam2* S, int32_t n, const char* chrs, const int32_t* dims, const double* values) {
  if (!S || n < 0 || (n && (!chrs || !dims || !values))) return LMGPU_INVALID;
  S->relin_thresholds.clear();
  for (int i = 0; i < n; i++) {
    if (dims[i] <= 0) {
      S->err = "lmgpu_isam2_set_relinearize_thresholds: bad dimension";
      S->relin_thresholds.clear();
      return LMGPU_INVALID;
    }
    S->relin_thresholds[(unsigned char)chrs[i]] = std::vector<double>(values, values + dims[i]);
    values += dims[i];
  }
  return LMGPU_OK;
}
int lmgpu_isam2_set_dogleg(lmgpu_isam2* S, double initialDelta, double wildfireThreshold, int32_t adaptationMode) {
  if (!S) return LMGPU_INVALID;
  if (S->ntot > 0 || !S->new_vars.empty() || adaptationMode < 0 || adaptationMode > 2 || !(initialDelta > 0.0)) {
    S->err = "lmgpu_isam2_set_dogleg: before the first variable, with a positive radius and an adaptation mode 0..2";
    return LMGPU_INVALID;
  }
  S->dogleg = true;
  S->dogleg_delta = initialDelta;
  S->dogleg_wildfire = wildfireThreshold;
  S->dogleg_mode = adaptationMode;
  return LMGPU_OK;
}
double lmgpu_isam2_get_dogleg_delta(const lmgpu_isam2* S) { return S ? S->dogleg_delta : 0.0; }
int lmgpu_isam2_set_evaluate_nonlinear_error(lmgpu_isam2* S, int32_t enable) {
  if (!S) return LMGPU_INVALID;
  S->evaluate_error = enable != 0;
  return LMGPU_OK;
}
int lmgpu_isam2_get_errors(const lmgpu_isam2* S, double* error_before, double* error_after) {
  if (!S) return LMGPU_INVALID;
  if (error_before) *error_before = S->error_before;
  if (error_after) *error_after = S->error_after;
  return LMGPU_OK;
}
// getFactorsUnsafe().error(calculateEstimate()) (which = 0) / .error(getLinearizationPoint()) (which = 2)
int lmgpu_isam2_error(lmgpu_isam2* S, int32_t which, double* out) {
  if (!S || !out || (which != 0 && which != 2)) return LMGPU_INVALID;
  if (S->device < 0) return LMGPU_HIP_ERROR;
  ISCHECK(hipSetDevice(S->device));
  int rc;
  if ((rc = is_stage_recycle(S))) return rc;
  if (which == 0) S->delta_reads++;
  if (which == 0 && S->any_replaced && (rc = is_update_delta(S, false))) return rc;
  return is_graph_error(S, which == 0, out);
}
int lmgpu_isam2_set_partial_relinearization_check(lmgpu_isam2* S, int32_t enable) {
  if (!S) return LMGPU_INVALID;
  S->partial_relin_check = enable != 0;
  return LMGPU_OK;
}

int lmgpu_isam2_get_unused_keys(const lmgpu_isam2* S, uint64_t* keys_out) {
  if (!S) return -1;
  if (keys_out) std::copy(S->last_unused.begin(), S->last_unused.end(), keys_out);
  return (int)S->last_unused.size();
}
int lmgpu_isam2_factor_exists(const lmgpu_isam2* S, int32_t i) { return S && i >= 0 && i < (int)S->facs.size() && !S->facs[i].removed; }

int lmgpu_isam2_set_find_unused_factor_slots(lmgpu_isam2* S, int32_t enable) {
  if (!S) return LMGPU_INVALID;
  S->find_unused_slots = enable != 0;
  return LMGPU_OK;
}
int lmgpu_isam2_marginalize_leaves(lmgpu_isam2* S, int32_t n, const uint64_t* leaf_keys, int32_t* n_marginal_out, int32_t* n_deleted_out) {
  if (!S || n < 0 || (n && !leaf_keys)) return LMGPU_INVALID;
  if (S->device < 0) return LMGPU_HIP_ERROR;
  ISCHECK(hipSetDevice(S->device));
  const int rc = is_marginalize_leaves(S, std::vector<uint64_t>(leaf_keys, leaf_keys + n));
  if (n_marginal_out) *n_marginal_out = (int32_t)S->last_marginal_idx.size();
  if (n_deleted_out) *n_deleted_out = (int32_t)S->last_deleted_idx.size();
  return rc;
}
int lmgpu_isam2_get_marginalize_result(const lmgpu_isam2* S, uint64_t* marginal_idx_out, uint64_t* deleted_idx_out) {
  if (!S) return LMGPU_INVALID;
  if (marginal_idx_out) std::copy(S->last_marginal_idx.begin(), S->last_marginal_idx.end(), marginal_idx_out);
  if (deleted_idx_out) std::copy(S->last_deleted_idx.begin(), S->last_deleted_idx.end(), deleted_idx_out);
  return LMGPU_OK;
}
int lmgpu_isam2_get_fixed_variables(const lmgpu_isam2* S, uint64_t* keys_out) {
  if (!S) return -1;
  if (keys_out) std::copy(S->fixed.begin(), S->fixed.end(), keys_out);
  return (int)S->fixed.size();
}
int lmgpu_isam2_get_marginal_factor(lmgpu_isam2* S, int32_t i, uint64_t* keys_out, int32_t* dims_out, double* info_colmajor) {
  if (!S || i < 0 || i >= (int)S->facs.size() || S->facs[i].removed || S->facs[i].marg < 0) return -1;
  const lmgpu_isam2::Marg& m = S->margs[S->facs[i].marg];
  for (size_t k = 0; k < m.vids.size(); k++) {
    if (keys_out) keys_out[k] = S->vars[m.vids[k]].key;
    if (dims_out) dims_out[k] = kVarDim[S->vars[m.vids[k]].type];
  }
  if (info_colmajor) {
    if (S->device < 0) return -1;
    if (hipSetDevice(S->device) != hipSuccess) return -1;
    std::vector<double> rm((size_t)m.m * m.ld);
    if (hipMemcpy(rm.data(), S->pool + m.u_off, ((size_t)(m.m - 1) * m.ld + m.m) * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    for (int r = 0; r < m.m; r++)
      for (int c = r; c < m.m; c++) info_colmajor[(size_t)c * m.m + r] = info_colmajor[(size_t)r * m.m + c] = rm[(size_t)r * m.ld + c];
  }
  return (int)m.vids.size();
}

int lmgpu_isam2_num_variables(const lmgpu_isam2* S) { return S ? (int)S->vid_of.size() : -1; }
int lmgpu_isam2_num_factors(const lmgpu_isam2* S) { return S ? (int)S->facs.size() : -1; }

int lmgpu_isam2_get_values(lmgpu_isam2* S, int32_t which, uint64_t* keys_out, int32_t* types_out, double* packed_out) {
  if (!S || which < 0 || which > 2) return LMGPU_INVALID;
  if (S->device < 0) return LMGPU_HIP_ERROR;
  ISCHECK(hipSetDevice(S->device));
  int rc;
  if ((rc = is_stage_recycle(S))) return rc;
  if ((rc = is_flush(S))) return rc;
  if (which == 0) S->delta_reads++;
  auto tnow = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tg0 = tnow();
  if (which == 1) {  // calculateBestEstimate: full back-substitution (ISAM2.cpp:763-766)
    if ((rc = is_update_delta(S, true))) return rc;
  } else if (which == 0 && S->any_replaced) {  // getDelta (:776-779)
    if ((rc = is_update_delta(S, false))) return rc;
  }
  if (S->trace) std::fprintf(stderr, "isam2 get_values(%d): updateDelta %.1f ms\n", which, 1e3 * (tnow() - tg0));
  std::vector<std::vector<double>> host(kNumVarTypes);
  for (int t = 0; t < kNumVarTypes; t++) {
    const int n = S->type_count[t];
    if (n == 0) continue;
    const double* src = S->theta[t];
    if (which != 2) {  // theta_.retract(delta_)
      hipLaunchKernelGGL(retract_kernel, dim3((n + 255) / 256), dim3(256), 0, S->stream, t, n, (const double*)S->theta[t], S->est[t],
                         (const int32_t*)S->d_type_xoff[t], (const double*)S->delta, (const int32_t*)nullptr);
      src = S->est[t];
    }
    host[t].resize((size_t)n * kVarStore[t]);
    ISCHECK(hipMemcpyAsync(host[t].data(), src, host[t].size() * sizeof(double), hipMemcpyDeviceToHost, S->stream));
  }
  ISCHECK(hipStreamSynchronize(S->stream));
  if (S->trace) std::fprintf(stderr, "isam2 get_values(%d): retract + copies %.1f ms\n", which, 1e3 * (tnow() - tg0));
  for (auto& kv : S->vid_of) {  // ascending by key
    const lmgpu_isam2::Var& v = S->vars[kv.second];
    if (keys_out) *keys_out++ = v.key;
    if (types_out) *types_out++ = v.type;
    if (packed_out) {
      std::memcpy(packed_out, &host[v.type][(size_t)v.tidx * kVarStore[v.type]], kVarStore[v.type] * sizeof(double));
      packed_out += kVarStore[v.type];
    }
  }
  return LMGPU_OK;
}

// ISAM2::marginalCovariance(key) (gtsam/nonlinear/ISAM2.h:253-257; tests/testGaussianISAM2.cpp:977-986): dim x dim, symmetric
int lmgpu_isam2_marginal_covariance(lmgpu_isam2* S, uint64_t key, double* cov) {
  if (!S || !cov) return LMGPU_INVALID;
  if (S->device < 0) return LMGPU_HIP_ERROR;
  auto it = S->vid_of.find(key);
  if (it == S->vid_of.end() || S->node_of[it->second] < 0) {
    S->err = "ISAM2::marginalCovariance: the variable is not in the Bayes tree";
    return LMGPU_INVALID;
  }
  ISCHECK(hipSetDevice(S->device));
  {
    const int rcr = is_stage_recycle(S);
    if (rcr) return rcr;
  }
  int rc = is_patch_tree(S);
  if (rc) return rc;
  const lmgpu_isam2::Var& var = S->vars[it->second];
  const int dim = kVarDim[var.type];
  std::vector<int32_t> path;
  size_t max_n = 1;
  for (int id = S->node_of[it->second]; id >= 0; id = S->clq[id].parent) {
    path.push_back(id);
    max_n = std::max(max_n, (size_t)S->clq[id].n);
  }
  const size_t need = (size_t)dim * S->ntot + 128;
  if (need > S->marg_cap) {
    if (S->d_marg) (void)hipFree(S->d_marg);
    S->d_marg = nullptr;
    S->marg_cap = is_next_cap(S->marg_cap, need);
    ISCHECK(hipMalloc((void**)&S->d_marg, S->marg_cap * sizeof(double)));
  }
  double* d_out = S->d_marg + (size_t)dim * S->ntot;
  int32_t* d_path;
  if ((rc = is_stage(S, path, &d_path))) return rc;
  if ((rc = is_flush(S))) return rc;
  ISCHECK(hipMemsetAsync(S->d_marg, 0, (size_t)dim * S->ntot * sizeof(double), S->stream));
  ISCHECK(hipMemsetAsync(S->d_status, 0x7f, sizeof(int), S->stream));
  if (max_n * sizeof(double) > 64 * 1024) {  // (a clique of more than 8 192 scalar columns: beyond the kernel's LDS row; refused, not mis-launched)
    S->err = "ISAM2::marginalCovariance: a clique on the variable's path is wider than this kernel handles";
    return LMGPU_INVALID;
  }
  hipLaunchKernelGGL(isam2_marginal_kernel, dim3(dim), dim3(64), max_n * sizeof(double), S->stream, (const int32_t*)d_path, (int)path.size(),
                     (const FrontDesc*)S->d_tree, (const int32_t*)S->d_tree_fx, (const int32_t*)S->d_tree_sx, (const double*)S->pool, S->d_marg, S->ntot,
                     var.xoff, dim, d_out, S->d_status);
  ISCHECK(hipGetLastError());  // a rejected launch would leave the status word untouched and a stale block in d_out
  ISCHECK(hipMemcpyAsync(cov, d_out, (size_t)dim * dim * sizeof(double), hipMemcpyDeviceToHost, S->stream));
  ISCHECK(hipMemcpyAsync(S->h_status, S->d_status, sizeof(int), hipMemcpyDeviceToHost, S->stream));
  ISCHECK(hipStreamSynchronize(S->stream));
  if (*S->h_status == 0) {
    S->failed_key = key;
    S->err = "indeterminate linear system in marginalCovariance";
    return LMGPU_INDETERMINATE;
  }
  return LMGPU_OK;
}

// ISAM2::calculateEstimate(Key) (gtsam/nonlinear/ISAM2.cpp:757-760: theta_.at(key) retracted by getDelta()[key]) / the linearization
// point of one variable: the back-substitution is brought up to date like for the whole estimate, but only this variable is retracted
// and downloaded (the whole estimate of a 10 000-pose graph is an 8.6 ms download; timing/timeIncremental.cpp asks for ONE pose per step)
int lmgpu_isam2_get_value(lmgpu_isam2* S, int32_t which, uint64_t key, int32_t* type_out, double* packed_out) {
  if (!S || (which != 0 && which != 2) || !packed_out) return LMGPU_INVALID;
  if (S->device < 0) return LMGPU_HIP_ERROR;
  auto it = S->vid_of.find(key);
  if (it == S->vid_of.end()) {
    S->err = "ISAM2: the key is not in the system (ValuesKeyDoesNotExist)";
    return LMGPU_INVALID;
  }
  ISCHECK(hipSetDevice(S->device));
  int rc;
  if ((rc = is_stage_recycle(S))) return rc;
  if (which == 0) S->delta_reads++;
  const bool walk = which == 0 && S->any_replaced;
  if (walk && (rc = is_update_delta_enqueue(S, false, false))) return rc;
  const lmgpu_isam2::Var& v = S->vars[it->second];
  const int t = v.type;
  if (which == 0) {
    // retracted by one thread straight into pinned host memory (the kernel's output pointer is placed so that THIS variable's slot is the
    // buffer): no copy command behind it
    const std::vector<int32_t> one{v.tidx};
    if ((rc = is_with_list(S, one, [&](const int32_t* d, int cnt) {
           hipLaunchKernelGGL(retract_kernel, dim3(1), dim3(256), 0, S->stream, t, cnt, (const double*)S->theta[t],
                              S->h_val_dev - (size_t)v.tidx * kVarStore[t], (const int32_t*)S->d_type_xoff[t], (const double*)S->delta, d);
         })))
      return rc;
  } else {
    ISCHECK(hipMemcpyAsync(S->h_val, S->theta[t] + (size_t)v.tidx * kVarStore[t], kVarStore[t] * sizeof(double), hipMemcpyDeviceToHost, S->stream));
  }
  if (walk) {  // the one wait of this call, and the status of the back-substitution
    if ((rc = is_update_delta_finish(S))) return rc;
  } else {
    ISCHECK(hipStreamSynchronize(S->stream));
  }
  std::memcpy(packed_out, S->h_val, kVarStore[t] * sizeof(double));
  if (type_out) *type_out = t;
  return LMGPU_OK;
}

int lmgpu_isam2_get_delta(lmgpu_isam2* S, double* packed) {
  if (!S || !packed) return LMGPU_INVALID;
  if (S->device < 0) return LMGPU_HIP_ERROR;
  ISCHECK(hipSetDevice(S->device));
  int rc;
  if ((rc = is_stage_recycle(S))) return rc;
  S->delta_reads++;
  if (S->any_replaced && (rc = is_update_delta(S, false))) return rc;
  std::vector<double> h((size_t)S->ntot);
  if (S->ntot) ISCHECK(hipMemcpy(h.data(), S->delta, h.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (auto& kv : S->vid_of) {
    const lmgpu_isam2::Var& v = S->vars[kv.second];
    for (int d = 0; d < kVarDim[v.type]; d++) *packed++ = h[v.xoff + d];
  }
  return LMGPU_OK;
}

static void is_collect(const lmgpu_isam2* S, int id, std::vector<int32_t>* out) {
  out->push_back(id);
  for (int ch : S->clq[id].children) is_collect(S, ch, out);
}

int lmgpu_isam2_num_cliques(lmgpu_isam2* S) {
  if (!S) return -1;
  S->snap.clear();
  for (int r : S->roots) is_collect(S, r, &S->snap);
  return (int)S->snap.size();
}

int lmgpu_isam2_clique_info(const lmgpu_isam2* S, int32_t i, int32_t* info5) {
  if (!S || !info5 || i < 0 || i >= (int)S->snap.size()) return LMGPU_INVALID;
  const lmgpu_isam2::Clq& c = S->clq[S->snap[i]];
  info5[0] = (int32_t)c.vars.size();
  info5[1] = c.nfv;
  info5[2] = c.nf;
  info5[3] = c.n;
  info5[4] = -1;
  if (c.parent >= 0) info5[4] = (int32_t)(std::find(S->snap.begin(), S->snap.end(), c.parent) - S->snap.begin());
  return LMGPU_OK;
}

int lmgpu_isam2_get_clique(lmgpu_isam2* S, int32_t i, uint64_t* keys, double* RSd_colmajor) {
  if (!S || i < 0 || i >= (int)S->snap.size()) return LMGPU_INVALID;
  const lmgpu_isam2::Clq& c = S->clq[S->snap[i]];
  if (keys)
    for (size_t k = 0; k < c.vars.size(); k++) keys[k] = S->vars[c.vars[k]].key;
  if (RSd_colmajor) {
    if (S->device < 0) return LMGPU_HIP_ERROR;
    std::vector<double> rm((size_t)c.nf * c.n);
    const int ldr = c.ld > 0 ? c.ld : c.n;
    rm.resize((size_t)c.nf * ldr);
    ISCHECK(hipMemcpy(rm.data(), S->pool + c.rsd_off, rm.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int r = 0; r < c.nf; r++)
      for (int j = 0; j < c.n; j++) RSd_colmajor[(size_t)j * c.nf + r] = (j >= r) ? rm[(size_t)r * ldr + j] : 0.0;
  }
  return LMGPU_OK;
}

}  // extern "C"
