// liblmgpu — C ABI (include/lmgpu.h) over the HIP engine.  gfx950 only; no CPU fallback: every compute entry
// point fails with LMGPU_HIP_ERROR when no device is bound.
//
// Host-side pieces in this file:
//   * graph intake + symbolic plan (plan.cpp) -> device descriptors
//   * the level-scheduled multifrontal solve (LDS fronts batched per level, HBM fronts blocked on MFMA)
//   * the LM control loop: a restatement of LevenbergMarquardtOptimizer::iterate / tryLambda
//     (gtsam/nonlinear/LevenbergMarquardtOptimizer.cpp:121-308) and NonlinearOptimizer::defaultOptimize
//     (gtsam/nonlinear/NonlinearOptimizer.cpp:62-117) that only reads three scalars back per inner iteration.
//   * multi-GPU: the subtrees hanging below the replicated top (HBM) fronts are dealt to the ranks; each rank
//     linearizes / eliminates / back-substitutes its own subtrees; the partial assembly of a replicated top front is summed
//     over the ranks in 256-row chunks (ncclAllReduce on a communication stream, RCCL over xGMI) and folded into the
//     working matrix just before each panel is factored; error scalars with a 3-double all-reduce.
//   * Gauss-Newton and Dogleg on the same device-resident graph / Bayes tree (gn_iterate, dl_iterate).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <limits>
#include <mutex>
#include <string>
#include <map>
#include <vector>

#include "../../include/lmgpu.h"
#include "kernels_dense.hpp"
#include "kernels_potrf.hpp"
#include "kernels_step.hpp"
#include "kernels_batched.hpp"
#include "kernels_level.hpp"
#include "kernels_bayes.hpp"
#include "kernels_schur.hpp"
#include "plan.hpp"

// Development switches (A/B forms of the same arithmetic, cross-checked by tests/test_gpu_lookahead.py): read from the environment by the
// TEST library only (liblmgpu_test.so, -DLMGPU_TEST_HOOKS).  The product library has two: LMGPU_GRAPH (graph replay on / off) and
// LMGPU_ISAM2_TRACE (phase times of the incremental path on stderr).
static const char* dev_switch(const char* name) {
#ifdef LMGPU_TEST_HOOKS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}


using namespace lmgpu;

#define HIPCHECK(expr)                                                                         \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      h->err = std::string(#expr) + ": " + hipGetErrorString(_e);                              \
      return LMGPU_HIP_ERROR;                                                                  \
    }                                                                                          \
  } while (0)
#define NCCLCHECK(expr)                                                                        \
  do {                                                                                         \
    ncclResult_t _r = (expr);                                                                  \
    if (_r != ncclSuccess) {                                                                   \
      h->err = std::string(#expr) + ": " + ncclGetErrorString(_r);                             \
      return LMGPU_HIP_ERROR;                                                                  \
    }                                                                                          \
  } while (0)

namespace {

struct Bucket {
  int type = 0, n = 0, noise_kind = 0;  // n = factors given by the caller
  int robust = 0;                       // lmgpu_robust_kind of the whole bucket (noiseModel::Robust around the Gaussian model)
  double robust_k = 0.0;
  std::vector<int32_t> graph_index, slots;
  std::vector<double> meas, noise;
  int rows = 0, cols = 0;  // Jacobian shape (cols includes b)
  // local (this rank's) part
  int n_loc = 0;
  std::vector<int32_t> loc_of;  // caller idx -> local idx inside the bucket, -1 if not evaluated on this rank
  int64_t joff = 0;             // pool offset of the bucket's local Jacobians
  int32_t* d_vidx = nullptr;
  double* d_meas = nullptr;
  double* d_noise = nullptr;
  int32_t* d_epos = nullptr;
};

struct LevelWork {
  // LDS fronts of this level, grouped by LDS-size bin: [bin_begin[b], bin_begin[b+1]) inside the level's list
  int list_begin = 0, list_count = 0;
  int lds_nf_max = 0;  // largest frontal dimension among the level's LDS fronts
  int lds_rsd_max = 0; // largest nf x (n | 1) among them: doubles of LDS the workgroup-per-front back-substitution stages
  int bin_begin[16] = {0};
  int bin_srows[16] = {0};  // rows of the front kept in LDS for the bin's launch
  int bin_jcap[16] = {0};   // doubles of Jacobian staging per workgroup (largest front of the bin, 96 .. LDSF_JCAP)
  int64_t pack_off[16] = {0};  // byte offset of the launch's packed leaf records in d_leafpack, and their stride (0: none)
  int pack_stride[16] = {0};
  std::vector<int> hbm;  // HBM fronts of this level
  int small_begin = 0, small_count = 0;  // those with nf <= BSS_MAX_NF, in d_hbm_small: back-substituted in one launch per level
  int bsd_begin = 0, bsd_count = 0;      // their 64-row blocks in d_bsd_table (hbm_backsolve_blocks_kernel: one workgroup per block)
  // "medium" HBM fronts (one outer panel, no gather leaves, not replicated): eliminated with batched launches (kernels_batched.hpp)
  int med_begin = 0, med_count = 0, med_max_fac = 0, med_max_child = 0, med_max_nf = 0, med_max_cols = 0, med_max_n = 0;
  // LDS fronts + medium fronts of the level as ONE launch (level_fused_kernel): its slice of d_level_tasks, the LDS size class of the launch
  int fuse_task_begin = 0, fuse_task_count = 0, fuse_nmax = 0, fuse_jcap = 0, fuse_threads = 0;
};

struct KTimer {
  bool on = false;
  bool dominant_only = false;  // lmgpu_set_kernel_timing(h, 2): only the roofline kernels (LINEARIZE; CHAIN, or SYRK = the per-step form of the same work) carry events
  std::vector<hipEvent_t> pool;
  struct Rec {
    int cat, e0, e1, launches;
  };
  std::vector<Rec> recs;
  int used = 0;
  double ms[LMGPU_KT_NUM] = {0}, work[LMGPU_KT_NUM] = {0};
  long long cnt[LMGPU_KT_NUM] = {0};
  int grab() {
    if (used == (int)pool.size()) {
      hipEvent_t e;
      (void)hipEventCreate(&e);
      pool.push_back(e);
    }
    return used++;
  }
  int begin(int cat, hipStream_t s) {
    if (!on || (dominant_only && cat != LMGPU_KT_LINEARIZE && cat != LMGPU_KT_CHAIN && cat != LMGPU_KT_SYRK)) return -1;
    const int e = grab();
    (void)hipEventRecord(pool[e], s);
    recs.push_back({cat, e, -1, 1});
    return (int)recs.size() - 1;
  }
  // `launches` back-to-back launches of one kernel may share one begin / end pair (an event record between two
  // launches costs ~8 us of idle device time: measured 35 x 11 us on the root's step kernels)
  void end(int r, hipStream_t s, double w = 0, int launches = 1) {
    if (!on || r < 0) return;
    const int e = grab();
    (void)hipEventRecord(pool[e], s);
    recs[r].e1 = e;
    recs[r].launches = launches;
    work[recs[r].cat] += w;
  }
  void resolve() {  // call after the stream is synchronised
    for (const Rec& r : recs) {
      float t = 0;
      if (r.e1 >= 0 && hipEventElapsedTime(&t, pool[r.e0], pool[r.e1]) == hipSuccess) {
        ms[r.cat] += t;
        cnt[r.cat] += r.launches;
      }
    }
    recs.clear();
    used = 0;
  }
  void reset() {
    for (int i = 0; i < LMGPU_KT_NUM; i++) ms[i] = work[i] = 0, cnt[i] = 0;
  }
};

const int kBinN[6] = {24, 48, 72, 96, 120, 139};  // 139^2 * 8 + staging = 162.5 KB <= 160 KiB of LDS per workgroup
const int kNumBins = 12;  // 0..5: LDS fronts by size; 6..11: the same sizes for GATHER leaves (only nf rows in LDS)
const int kLdsLimitN = 139;
const int kLdsFrontExtra = LDSF_EXTRA_BYTES;
const int NB = 64;    // potrf / trsm step
const int NBO = 256;  // outer panel: rows eliminated per trailing update of the HBM front
const int kSyrkLds = 2 * 2 * SYRK_KC * SYRK_LDW * 8;

}  // namespace

// In-process stand-in for the RCCL communicator: W handles of ONE process (one thread each) sum their buffers through a
// host rendezvous.  Same call sites and semantics as the ncclAllReduce path; exists so that the sharded LM loop can be
// exercised end to end on a single GPU (two RCCL ranks may not share a device).  Not a performance path.
struct lmgpu_local_group {
  int world = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  long generation = 0;
  std::vector<void*> ptr;
  std::vector<hipStream_t> stream;
  // rendezvous: returns once all ranks are in
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const long g = generation;
    if (++arrived == world) {
      arrived = 0;
      generation++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != g; });
    }
  }
};

__global__ void local_sum_kernel(double* __restrict__ dst, const double* __restrict__ src, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] += src[i];
}
__global__ void local_min_kernel(int* __restrict__ dst, const int* __restrict__ src) { *dst = min(*dst, *src); }

// damping weights of diagonal damping: w = sqrt(clamp(diag, min, max))^2 -- the square of what the reference's prior carries
// (value.cwiseMax(minDiagonal).cwiseMin(maxDiagonal).cwiseSqrt(), LevenbergMarquardtOptimizer.cpp:294-298; the prior squares it again)
__global__ void clamp_diag_kernel(int n, const double* __restrict__ diag, double* __restrict__ w, double lo, double hi) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double s = sqrt(fmin(fmax(diag[i], lo), hi));
  w[i] = s * s;
}

struct lmgpu_handle {
  lmgpu_config cfg;
  int device = -1;
  std::string err;
  int failed_slot = -1;
  Plan plan;
  std::vector<Bucket> buckets;
  bool finalized = false, have_values = false, linearized = false, solved = false;
  bool have_jacobians = false;  // the pool holds the [A b] of the last linearization (they outlive `linearized`: an accepted step moves the values, not the Jacobians)
  std::vector<int32_t> graph_index_sorted;  // parallel to plan.factors

  // ---- ownership (multi-GPU).  Everything is "active" and "counted" when world_size == 1.
  std::vector<int> front_owner;      // -1 = replicated on every rank
  std::vector<char> front_active;    // this rank assembles / factors / back-substitutes the front
  std::vector<int32_t> fac_local;    // plan.factors index -> local factor index (-1: not on this rank)
  int nfac = 0;                      // local factors
  int n_counted = 0;                 // local factors [0, n_counted) enter this rank's error sums (each factor counted on one rank)

  // ---- device state
  hipStream_t stream = nullptr;
  double *bs_inv = nullptr, *bs_x = nullptr;  // dataflow back-substitution scratch
  double* inv16 = nullptr;                     // 16 x (16x16) inverses of the current outer panel's diagonal tiles
  bool two_launch_panel = false;               // LMGPU_PANEL_2L=1: diag_potrf + panel_trsm for every outer panel (A/B)
  bool no_fuse = false;                        // LMGPU_NO_FUSE=1: trailing update and next panel as separate launches (A/B)
  struct ChainPlan { int i0 = -1, nsteps = 0, ntasks = 0; int2* d_tasks = nullptr; double flop = 0; };
  std::map<int, std::vector<ChainPlan>> chain_plans;  // per HBM front: ticket order of its chained launch(es) (built at first use)
  int chain_split_pct = 60;                    // LMGPU_CHAIN_SPLIT (development): share of a block's update tasks listed in front of its row-panel workgroups
  bool chain_merge = true;                     // LMGPU_NO_MERGE: update tiles one step per pass instead of pairs of steps (chain_schedule)
  int chain_far_pct = 50;                      // LMGPU_CHAIN_FAR: tile rows beyond this percentage of the front are scheduled late (chain_schedule)
  double* d_lambda = nullptr;   // damping parameter of the solve being queued (device memory: see do_solve_enqueue)
  bool merge_backsub = false;   // LDS fronts of consecutive levels in one dataflow launch (deep trees; LMGPU_MERGE_BACKSUB=0/1)
  // the same on the way up: runs of consecutive levels without dense fronts in between whose LDS fronts go as ONE launch
  // (lds_front_merged_kernel); elim_seg_of[level] = index into elim_segs when the level starts a segment, -2 when it lies inside one
  struct ElimSeg {
    int lvl_lo, lvl_hi, nmax, jcap, threads;
  };
  std::vector<ElimSeg> elim_segs;
  std::vector<int> elim_seg_of;
  bool merge_elim = false;
  FillUpper* d_fill_upper = nullptr;  // the update matrices of the merged launches' fronts (fill_upper_kernel)
  int n_fill_upper = 0;
  bool fuse_levels = false;            // levels with LDS fronts and medium fronts: one launch for both (LMGPU_FUSE_LEVELS=0/1)
  LevelTask* d_level_tasks = nullptr;
  LevelSync* d_level_sync = nullptr;   // one record per medium front (all levels)
  int n_level_sync = 0;
  int32_t *d_bs_parent = nullptr, *d_bs_pos = nullptr;  // per front: parent front if it is an LDS front (else -1); position in d_lists (-1: HBM)
  char* d_leafpack = nullptr;                           // packed descriptors of the LDS fronts (kernels_front.hpp, LEAFPACK_*)
  unsigned int* d_bs_done = nullptr;                    // per front flag + one ticket counter per level
  bool use_graph = false;       // replay the solve's launch sequence as a hipGraph (deep trees; LMGPU_GRAPH=0/1 overrides)
  int eager_solves = 0;
  hipGraphExec_t solve_graph[2] = {nullptr, nullptr};  // [1]: with the extra gradient vector of the marginal solves
  bool no_wide16 = false;                      // LMGPU_NO_WIDE16=1: LDS fronts always with four waves (A/B)
  int wide16_max = 256;                        // launches of at most this many wide LDS fronts take sixteen waves per front (LMGPU_WIDE16_MAX)
  bool bsd_ticket = false;                     // LMGPU_BSD_TICKET=1: the block back-substitution always draws tickets (tests: the path of levels with more blocks than CUs)
  bool no_tail = false;                        // LMGPU_NO_TAIL=1: the end of a front as separate update / panel launches (A/B)
  bool no_chain = false;                       // LMGPU_NO_CHAIN=1: one launch per fused step instead of one per run of steps (A/B)
  unsigned int* d_pflags = nullptr;            // hand-off flags of panel_dataflow_kernel, PDF_FLAG_WORDS per outer panel
  int pflags_panels = 0;
  unsigned int* bs_flags = nullptr;
  double* pool = nullptr;
  size_t pool_doubles = 0;
  double* vals[2][kNumVarTypes] = {};
  double* saved[kNumVarTypes] = {};
  int cur = 0;
  int32_t* type_xoff[kNumVarTypes] = {};
  double *delta = nullptr, *dampw = nullptr, *hdiag = nullptr, *ebuf0 = nullptr, *ebuf1 = nullptr, *partial = nullptr, *dscal = nullptr,
         *ywork = nullptr;
  int* d_status = nullptr;
  double* h_scal = nullptr;  // pinned: [0]=err, [1]=lin0, [2]=lin1
  int* h_status = nullptr;   // pinned
  FacDesc* d_fd = nullptr;
  FrontDesc* d_fronts = nullptr;
  FrontFac* d_ffac = nullptr;
  ChildRef* d_childs = nullptr;
  int32_t *d_cmap = nullptr, *d_fxoff = nullptr, *d_sxoff = nullptr, *d_lists = nullptr;
  int32_t *d_hbm_small = nullptr, *d_f_ld = nullptr, *d_med_list = nullptr;
  MedFront* d_med_fronts = nullptr;     // packed records of the medium fronts, level by level (same order as d_med_list)
  BsdBlock* d_bsd_table = nullptr;      // 64-row blocks of the smaller HBM fronts, per level, each front from its last block to its first
  // runs of consecutive levels that hold nothing but such fronts: their blocks as ONE launch, levels top-down (merged back-substitution only)
  struct BsdRun {
    int lvl_hi, lvl_lo, begin, count;
  };
  std::vector<BsdRun> bsd_runs;
  std::vector<int> bsd_run_of;          // per level: index of the run that starts (top) here, -2 inside a run, -1 none
  BsdBlock* d_bsd_run_table = nullptr;
  // the clears and sentinel fills at the head of an elimination / a back-substitution, one launch each (fill_chunks_kernel)
  FillChunk* d_fill_elim = nullptr;
  FillChunk* d_fill_backsub = nullptr;
  int n_fill_elim = -1, n_fill_backsub = -1, fill_backsub_merge = -1;
  double* d_bsd_x = nullptr;            // their published x (64 per block), preset to the all-ones sentinel at the start of a back-substitution
  unsigned int* d_bsd_ticket = nullptr; // one ticket counter per level
  size_t bsd_x_count = 0;
  ZeroRange* d_zero_ranges = nullptr;   // the HBM fronts a solve clears first, when they are many and scattered (zero_ranges_kernel)
  int n_zero_ranges = 0;
  int num_cus = 0;                      // compute units of the device (a grid of at most this many small workgroups is resident at once)
  // deterministic assembly of HBM fronts (kernels_dense.hpp: hbm_assemble_rows_kernel): per front the start of its n + 1 row
  // pointers in d_rowptr (-1: none), the pointers (offsets into d_rowsrc) and the sources
  std::vector<int32_t> row_begin;
  int32_t *d_row_begin = nullptr, *d_rowptr = nullptr;
  RowSrc* d_rowsrc = nullptr;
  int n_lds_fronts = 0;                         // all levels' LDS-class fronts are contiguous in d_lists
  double *bt_ebuf = nullptr, *bt_vec = nullptr;  // Dogleg: per-clique / per-row squared residuals; gradient / zero vector
  int bt_ebuf_len = 0;
  // the gradient as a fixed-order sum: a slot per (clique, column) [per 64-row chunk for HBM fronts], gathered per scalar (built on first use)
  double* bt_part = nullptr;
  int64_t* d_bt_part_off = nullptr;  // per entry of d_lists (LDS-class fronts)
  std::vector<int64_t> bt_hbm_part_off;  // per front id (HBM fronts)
  int32_t *d_bt_ptr = nullptr, *d_bt_idx = nullptr;
  bool bt_gather_built = false;
  double* inv16_med = nullptr;
  std::vector<char> is_med;
  int64_t* d_f_off = nullptr;
  int32_t *d_scalar_var = nullptr, *d_scalar_col = nullptr, *d_vi_ptr = nullptr, *d_vi_fac = nullptr;
  int8_t* d_vi_pos = nullptr;
  // gather-mode (Schur form) assembly of HBM fronts from their leaf children
  struct GatherRange {
    int pblk_begin = 0, pblk_short = 0, pblk_long = 0, vblk_begin = 0, vblk_count = 0, leaf_begin = 0, leaf_count = 0;  // short blocks first
    // blocks are sorted by destination row: [cs[c], cs[c+1]) = short pair blocks whose rows lie in row chunk c (256 rows),
    // likewise cl (long pair blocks) and cv (variable blocks); offsets relative to the three list starts
    std::vector<int> cs, cl, cv;
    // write mode: every contribution to the front's upper triangle before the gather comes from the gather itself (all children are
    // gather leaves), so the gather writes its blocks and only the blocks no list covers are cleared: [zero_begin, +zero_count)
    bool write_ok = false;
    int zero_begin = 0, zero_count = 0;
  };
  std::vector<GatherRange> gather;  // per front (only HBM fronts have non-empty ranges)
  GPairBlock* d_gpblk = nullptr;
  GZeroBlock* d_gzero = nullptr;
  int n_hbm_fronts = 0;          // number of HBM-class fronts on this rank (selects the clearing regime in do_eliminate)
  bool no_gather_write = false;  // LMGPU_NO_GATHER_WRITE=1: clear the front and add (A/B)
  GPairEntry* d_gpent = nullptr;
  GVarBlock* d_gvblk = nullptr;
  GVarEntry* d_gvent = nullptr;
  double* d_gcorner = nullptr;
  std::vector<FrontDesc> h_fronts;
  std::vector<int64_t> f_off;  // HBM fronts: pool offset of the dense front (else -1)
  std::vector<int> f_ld;
  std::vector<int64_t> s_off;  // multi-rank, replicated HBM front: pool offset of this rank's PARTIAL assembly (else -1)
  hipStream_t comm_stream = nullptr;  // chunked all-reduce of the partial assembly, beside the factorisation
  hipEvent_t asm_ev = nullptr;
  std::vector<hipEvent_t> chunk_ev;
  std::vector<LevelWork> levels;
  int ntot = 0, nstore = 0;
  bool dampw_is_ones = false;
  int inv16_owner = -1;  // HBM front whose panels' 16x16 inverses inv16 currently holds
  double* gex = nullptr;         // ntot doubles: extra gradient vector for solves with a prescribed right-hand side (marginals)
  double* gex_active = nullptr;  // == gex while such a solve is being assembled, else nullptr

  // ---- LM
  lmgpu_lm_state lm{};
  lmgpu_timings tim{};
  hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  KTimer kt;

  ncclComm_t comm = nullptr;       // scalars / status / Hessian diagonal: issued on the compute stream only
  ncclComm_t comm_data = nullptr;  // the chunked all-reduce of a replicated front's partial assembly: issued on the communication stream only
                                   // (a communicator of its own, so that no communicator is ever driven from two streams at once)
  lmgpu_local_group* lgroup = nullptr;  // test-only in-process communicator (lmgpu_comm_init_local)
};

namespace {

template <typename T>
int upload(lmgpu_handle* h, T** dst, const std::vector<T>& src) {
  *dst = nullptr;
  if (src.empty()) {
    HIPCHECK(hipMalloc((void**)dst, sizeof(T)));
    return LMGPU_OK;
  }
  HIPCHECK(hipMalloc((void**)dst, src.size() * sizeof(T)));
  HIPCHECK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return LMGPU_OK;
}

int need_device(lmgpu_handle* h) {
  if (h->device < 0) {
    h->err = "no HIP device bound to this handle (structure-only handle); the hot path has no CPU fallback";
    return LMGPU_HIP_ERROR;
  }
  return LMGPU_OK;
}

int need_comm(lmgpu_handle* h) {
  if ((h->cfg.world_size > 1 || (h->cfg.flags & LMGPU_FLAG_SPLIT_ROOT)) && !h->comm && !h->lgroup) {
    h->err = "world_size > 1 but lmgpu_comm_init has not been called";
    return LMGPU_INVALID;
  }
  return LMGPU_OK;
}

ValuesDev values_dev(lmgpu_handle* h, int which) {
  ValuesDev v;
  for (int t = 0; t < kNumVarTypes; t++) v.v[t] = h->vals[which][t];
  return v;
}

BucketDev bucket_dev(lmgpu_handle* h, const Bucket& b) {
  BucketDev d;
  d.type = b.type;
  d.n = b.n_loc;
  d.noise_kind = b.noise_kind;
  d.robust = b.robust;
  d.rk = b.robust_k;
  d.vidx = b.d_vidx;
  d.meas = b.d_meas;
  d.noise = b.d_noise;
  d.J = h->pool + b.joff;
  d.epos = b.d_epos;
  d.sel = nullptr;
  return d;
}

// launch the factor kernels of every bucket: JAC -> Jacobians, else per-factor errors into ebuf0
template <bool JAC>
void launch_factors(lmgpu_handle* h, int which) {
  const ValuesDev vals = values_dev(h, which);
  for (const Bucket& b : h->buckets) {
    if (b.n_loc == 0) continue;
    const BucketDev d = bucket_dev(h, b);
    const int g256 = (b.n_loc + 255) / 256, g128 = (b.n_loc + 127) / 128;
    hipStream_t s = h->stream;
    switch (b.type) {
      case LMGPU_F_SFM:
        if (JAC) {
          // algorithmic bytes: 232 B per factor + each variable once (120 B camera, 24 B point)
          const int kt = h->kt.begin(LMGPU_KT_LINEARIZE, s);
          hipLaunchKernelGGL(sfm_linearize_kernel, dim3(g256), dim3(256), 0, s, d, vals);
          h->kt.end(kt, s, 232.0 * b.n_loc + 120.0 * h->plan.type_count[3] + 24.0 * h->plan.type_count[2]);
        } else {
          hipLaunchKernelGGL(sfm_error_kernel, dim3(g256), dim3(256), 0, s, d, vals, h->ebuf0);
        }
        break;
      case LMGPU_F_BETWEEN_POSE2:
        hipLaunchKernelGGL((generic_factor_kernel<1, 3, 3, 3, 3, 0, 3, 0, 3, JAC>), dim3(g128), dim3(128), 0, s, d, vals, h->ebuf0);
        break;
      case LMGPU_F_BETWEEN_POSE3:
        hipLaunchKernelGGL((generic_factor_kernel<2, 6, 6, 6, 12, 1, 12, 1, 12, JAC>), dim3(g128), dim3(128), 0, s, d, vals, h->ebuf0);
        break;
      case LMGPU_F_PRIOR_POSE2:
        hipLaunchKernelGGL((generic_factor_kernel<3, 3, 3, 0, 3, 0, 3, -1, 0, JAC>), dim3(g128), dim3(128), 0, s, d, vals, h->ebuf0);
        break;
      case LMGPU_F_PRIOR_POSE3:
        hipLaunchKernelGGL((generic_factor_kernel<4, 6, 6, 0, 12, 1, 12, -1, 0, JAC>), dim3(g128), dim3(128), 0, s, d, vals, h->ebuf0);
        break;
      case LMGPU_F_PRIOR_POINT3:
        hipLaunchKernelGGL((generic_factor_kernel<5, 3, 3, 0, 3, 2, 3, -1, 0, JAC>), dim3(g128), dim3(128), 0, s, d, vals, h->ebuf0);
        break;
      case LMGPU_F_PRIOR_CAM:
        hipLaunchKernelGGL((generic_factor_kernel<6, 9, 9, 0, 15, 3, 15, -1, 0, JAC>), dim3(g128), dim3(128), 0, s, d, vals, h->ebuf0);
        break;
      case LMGPU_F_PROJECTION:
        hipLaunchKernelGGL((generic_factor_kernel<7, 2, 6, 3, 7, 1, 12, 2, 3, JAC>), dim3(g128), dim3(128), 0, s, d, vals, h->ebuf0);
        break;
      case LMGPU_F_PROJECTION_BPS:
        hipLaunchKernelGGL((generic_factor_kernel<8, 2, 6, 3, 19, 1, 12, 2, 3, JAC>), dim3(g128), dim3(128), 0, s, d, vals, h->ebuf0);
        break;
      case LMGPU_F_BEARING_RANGE_2D:
        hipLaunchKernelGGL((generic_factor_kernel<9, 2, 3, 2, 2, 0, 3, 4, 2, JAC>), dim3(g128), dim3(128), 0, s, d, vals, h->ebuf0);
        break;
      case LMGPU_F_SFM2:
        hipLaunchKernelGGL(sfm2_factor_kernel<JAC>, dim3(g128), dim3(128), 0, s, d, vals, h->ebuf0);
        break;
      case LMGPU_F_PRIOR_CAL3_S2:
        hipLaunchKernelGGL((generic_factor_kernel<11, 5, 5, 0, 5, 5, 5, -1, 0, JAC>), dim3(g128), dim3(128), 0, s, d, vals, h->ebuf0);
        break;
    }
  }
}

void reduce_to(lmgpu_handle* h, const double* buf, int n, double* dst, hipStream_t st = nullptr, double* scratch = nullptr) {
  const int g = std::min(256, std::max(1, (n + 255) / 256));
  if (!st) st = h->stream;
  if (!scratch) scratch = h->partial;
  hipLaunchKernelGGL(reduce_stage1, dim3(g), dim3(256), 0, st, buf, n, scratch);
  hipLaunchKernelGGL(reduce_stage2, dim3(1), dim3(256), 0, st, (const double*)scratch, g, dst);
}

// all-reduce over the ranks: RCCL, or the in-process local group
int allreduce_sum(lmgpu_handle* h, double* buf, size_t count, hipStream_t s) {
  if (h->comm) {
    NCCLCHECK(ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, h->comm, s));
  } else if (h->lgroup) {
    lmgpu_local_group* g = h->lgroup;
    HIPCHECK(hipStreamSynchronize(s));
    g->ptr[h->cfg.rank] = buf;
    g->barrier();
    if (h->cfg.rank == 0) {
      for (int r = 1; r < g->world; r++)
        hipLaunchKernelGGL(local_sum_kernel, dim3(1024), dim3(256), 0, s, buf, (const double*)g->ptr[r], count);
      HIPCHECK(hipStreamSynchronize(s));
      for (int r = 1; r < g->world; r++) HIPCHECK(hipMemcpy(g->ptr[r], buf, count * sizeof(double), hipMemcpyDeviceToDevice));
    }
    g->barrier();
  }
  return LMGPU_OK;
}
int allreduce_min_int(lmgpu_handle* h, int* buf, hipStream_t s) {
  if (h->comm) {
    NCCLCHECK(ncclAllReduce(buf, buf, 1, ncclInt, ncclMin, h->comm, s));
  } else if (h->lgroup) {
    lmgpu_local_group* g = h->lgroup;
    HIPCHECK(hipStreamSynchronize(s));
    g->ptr[h->cfg.rank] = buf;
    g->barrier();
    if (h->cfg.rank == 0) {
      for (int r = 1; r < g->world; r++) hipLaunchKernelGGL(local_min_kernel, dim3(1), dim3(1), 0, s, buf, (const int*)g->ptr[r]);
      HIPCHECK(hipStreamSynchronize(s));
      for (int r = 1; r < g->world; r++) HIPCHECK(hipMemcpy(g->ptr[r], buf, sizeof(int), hipMemcpyDeviceToDevice));
    }
    g->barrier();
  }
  return LMGPU_OK;
}

// sum `count` doubles at dscal+first over the ranks (each factor is counted on exactly one rank)
int allreduce_scalars(lmgpu_handle* h, int first, int count) {
  return allreduce_sum(h, h->dscal + first, count, h->stream);
}

int compute_error(lmgpu_handle* h, int which, double* out) {
  launch_factors<false>(h, which);
  reduce_to(h, h->ebuf0, h->n_counted, h->dscal);
  int rc = allreduce_scalars(h, 0, 1);
  if (rc) return rc;
  HIPCHECK(hipMemcpyAsync(h->h_scal, h->dscal, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  *out = h->h_scal[0];
  return LMGPU_OK;
}

int do_linearize(lmgpu_handle* h) {
  launch_factors<true>(h, h->cur);
  HIPCHECK(hipGetLastError());
  h->linearized = true;
  h->have_jacobians = true;
  return LMGPU_OK;
}

int launch_hessian_diag(lmgpu_handle* h) {
  hipLaunchKernelGGL(hessian_diag_kernel, dim3((h->ntot + 255) / 256), dim3(256), 0, h->stream, h->ntot, h->d_scalar_var, h->d_scalar_col,
                     h->d_vi_ptr, h->d_vi_fac, h->d_vi_pos, h->d_fd, (const double*)h->pool, h->hdiag);
  return allreduce_sum(h, h->hdiag, h->ntot, h->stream);
}

int fill_dampw(lmgpu_handle* h, int diagonal, double min_diag, double max_diag) {
  if (!diagonal) {
    if (!h->dampw_is_ones) {
      std::vector<double> ones(h->ntot, 1.0);
      HIPCHECK(hipMemcpyAsync(h->dampw, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
      HIPCHECK(hipStreamSynchronize(h->stream));
      h->dampw_is_ones = true;
    }
    return LMGPU_OK;
  }
  // hessianDiagonal, clamped (LevenbergMarquardtOptimizer.cpp:291-298: sqrt then squared by the prior = clamp(diag)); an
  // element-wise kernel on the stream, nothing crosses PCIe
  int rc = launch_hessian_diag(h);
  if (rc) return rc;
  hipLaunchKernelGGL(clamp_diag_kernel, dim3((h->ntot + 255) / 256), dim3(256), 0, h->stream, h->ntot, (const double*)h->hdiag, h->dampw, min_diag, max_diag);
  h->dampw_is_ones = false;
  return LMGPU_OK;
}

// ---- numeric elimination of all (active) fronts, level by level (a10-a13)
// The front's first contribution is its own Schur gather (all children are gather leaves, single rank): no clear, the gather writes.
// Only in the per-front clearing regime (few large HBM fronts); the span memset of general sparse graphs covers everything anyway.
// With several ranks the same holds for the rank's partial-assembly buffer (the working matrix still starts from zero): after the
// all-reduce it holds the sums of the previous solve, which this rank's gather overwrites where it has entries and the list of
// blocks it does not cover clears.
static bool gather_writes(const lmgpu_handle* h, int fi) {
  const lmgpu_handle::GatherRange& G = h->gather[fi];
  if (!(G.write_ok && G.leaf_count > 0 && !h->no_gather_write && h->n_hbm_fronts <= 4)) return false;
  if (h->s_off[fi] < 0) return true;
  return (h->h_fronts[fi].pad & 1) != 0 && (h->comm || h->lgroup);  // the partial-assembly buffer is in use (`split` below)
}

static void add_fill(std::vector<FillChunk>& t, const void* p, size_t bytes, uint32_t value) {
  for (size_t o = 0; o < bytes; o += FILL_CHUNK_BYTES)
    t.push_back(FillChunk{(unsigned long long)(uintptr_t)p + o, (uint32_t)std::min<size_t>(FILL_CHUNK_BYTES, bytes - o), value});
}

int do_eliminate(lmgpu_handle* h, double lambda_v, const double* lambda_p) {  // lambda by value (eager) or in device memory (graph replay)
  hipStream_t s = h->stream;
  const bool merge_el = h->merge_elim && !h->elim_segs.empty();
  std::vector<FillChunk> fills;
  const bool build_fills = h->n_fill_elim < 0;
  if (build_fills) {
    add_fill(fills, h->d_status, sizeof(int), 0x7f7f7f7fu);
    add_fill(fills, h->d_status + 1, sizeof(int), 0u);  // [1]: a dataflow hand-off timed out (1 + front id)
    if (merge_el || h->fuse_levels)  // one ticket counter per level (shared with the back-substitution, which clears them again)
      add_fill(fills, h->d_bs_done, (size_t)(h->h_fronts.size() + h->levels.size() + 1) * sizeof(unsigned int), 0u);
    if (h->fuse_levels) add_fill(fills, h->d_level_sync, (size_t)h->n_level_sync * sizeof(LevelSync), 0u);
  }
  {  // HBM fronts are accumulated into by their children (atomics) before their own level runs: clear them all first
    const int kt0 = h->kt.begin(LMGPU_KT_HBM_ASSEMBLE, s);
    int n_hbm = 0;
    int64_t lo = INT64_MAX, hi = 0, sum = 0;
    for (const LevelWork& L : h->levels)
      for (int fi : L.hbm) {
        const int64_t cnt = (int64_t)h->h_fronts[fi].n * h->f_ld[fi] * (h->s_off[fi] >= 0 ? 2 : 1);
        n_hbm++;
        sum += cnt;
        lo = std::min(lo, h->f_off[fi]);
        hi = std::max(hi, h->f_off[fi] + cnt);
      }
    if (n_hbm > 4 && hi - lo <= 2 * sum) {
      // many mid-size fronts (general sparse graphs): one memset over their span; what lies between them ([R S d] / update
      // storage of LDS fronts, laid out in the same post-order) is rewritten by this elimination before it is read
      if (build_fills) add_fill(fills, h->pool + lo, (size_t)(hi - lo) * sizeof(double), 0u);
    } else if (n_hbm > 4) {
      // scattered through the pool: one launch over the list of their ranges (built once; f_ld and the pool offsets are multiples of 16)
      if (!h->d_zero_ranges) {
        std::vector<ZeroRange> zr;
        for (const LevelWork& L : h->levels)
          for (int fi : L.hbm) {
            const bool gw = gather_writes(h, fi);
            const int64_t cnt = (int64_t)h->h_fronts[fi].n * h->f_ld[fi];
            if (!gw || h->s_off[fi] >= 0) zr.push_back(ZeroRange{h->f_off[fi], cnt});
            if (!gw && h->s_off[fi] >= 0) zr.push_back(ZeroRange{h->s_off[fi], cnt});
          }
        h->n_zero_ranges = (int)zr.size();
        const int rcu = upload(h, &h->d_zero_ranges, zr);
        if (rcu) return rcu;
      }
      if (h->n_zero_ranges > 0)
        hipLaunchKernelGGL(zero_ranges_kernel, dim3(16, h->n_zero_ranges), dim3(256), 0, s, (const ZeroRange*)h->d_zero_ranges, h->pool);
    } else {
      for (const LevelWork& L : h->levels)
        for (int fi : L.hbm) {
          // gather_writes: the upper triangle of the buffer the front is assembled into is written by the gather and a list of
          // uncovered blocks -- the front itself on one rank, the partial-assembly buffer with several
          const bool gw = gather_writes(h, fi);
          const size_t bytes = (size_t)h->h_fronts[fi].n * h->f_ld[fi] * sizeof(double);
          if (!gw || h->s_off[fi] >= 0) HIPCHECK(hipMemsetAsync(h->pool + h->f_off[fi], 0, bytes, s));
          if (!gw && h->s_off[fi] >= 0) HIPCHECK(hipMemsetAsync(h->pool + h->s_off[fi], 0, bytes, s));
        }
    }
    if (build_fills) {
      h->n_fill_elim = (int)fills.size();
      const int rcu = upload(h, &h->d_fill_elim, fills);
      if (rcu) return rcu;
    }
    hipLaunchKernelGGL(fill_chunks_kernel, dim3(h->n_fill_elim), dim3(256), 0, s, (const FillChunk*)h->d_fill_elim);
    h->kt.end(kt0, s);
  }
  // (Running a level's LDS fronts on a second stream beside its dense fronts -- they only depend on the levels below -- was measured
  //  with the replayed graph: 4.9 vs 2.5 ms per LM iteration on sphere2500; every cross-stream edge of the graph costs more than the
  //  45 us of LDS-front latency it hides.  Not kept.)
  if (merge_el)  // "not published yet" over the update matrices of the merged launches' fronts
    hipLaunchKernelGGL(fill_upper_kernel, dim3(h->n_fill_upper), dim3(256), 0, s, (const FillUpper*)h->d_fill_upper, h->pool);
  for (size_t li = 0; li < h->levels.size(); li++) {
    const LevelWork& L = h->levels[li];
    const int seg = merge_el ? h->elim_seg_of[li] : -1;
    if (seg >= 0) {
      const lmgpu_handle::ElimSeg& E = h->elim_segs[seg];
      const int b0 = h->levels[E.lvl_lo].list_begin, e0 = h->levels[E.lvl_hi].list_begin + h->levels[E.lvl_hi].list_count;
      const size_t lds = kLdsFrontExtra - (size_t)(LDSF_JCAP - E.jcap) * 8 + 64 + (size_t)E.nmax * E.nmax * sizeof(double);
      const int kt = h->kt.begin(LMGPU_KT_LDS_FRONT, s);
      unsigned int* ticket = h->d_bs_done + h->h_fronts.size() + E.lvl_hi;
      if (E.threads == 1024)
        hipLaunchKernelGGL(lds_front_merged_kernel<1024>, dim3(e0 - b0), dim3(1024), lds, s, (const int32_t*)h->d_lists, b0, e0, (const FrontDesc*)h->d_fronts,
                           (const FrontFac*)h->d_ffac, (const FacDesc*)h->d_fd, (const ChildRef*)h->d_childs, (const int32_t*)h->d_cmap,
                           (const int32_t*)h->d_fxoff, h->pool, lambda_v, lambda_p, (const double*)h->dampw, h->d_status, E.nmax, E.jcap,
                           (const double*)h->gex_active, ticket);
      else
        hipLaunchKernelGGL(lds_front_merged_kernel<256>, dim3(e0 - b0), dim3(E.threads), lds, s, (const int32_t*)h->d_lists, b0, e0, (const FrontDesc*)h->d_fronts,
                           (const FrontFac*)h->d_ffac, (const FacDesc*)h->d_fd, (const ChildRef*)h->d_childs, (const int32_t*)h->d_cmap,
                           (const int32_t*)h->d_fxoff, h->pool, lambda_v, lambda_p, (const double*)h->dampw, h->d_status, E.nmax, E.jcap,
                           (const double*)h->gex_active, ticket);
      h->kt.end(kt, s);
    }
    const bool fused = h->fuse_levels && L.fuse_task_count > 0;
    if (fused) {  // the level's LDS fronts and medium fronts in one grid
      const MedLevel ML{(const MedFront*)(h->d_med_fronts + L.med_begin)};
      const size_t lds = std::max<size_t>(kLdsFrontExtra - (size_t)(LDSF_JCAP - L.fuse_jcap) * 8 + 64 + (size_t)L.fuse_nmax * L.fuse_nmax * sizeof(double),
                                          (size_t)DIAG_LDS_BYTES);
      unsigned int* ticket = h->d_bs_done + h->h_fronts.size() + li;
      const int kt = h->kt.begin(LMGPU_KT_PANEL, s);
      if (L.fuse_threads == 1024)
        hipLaunchKernelGGL(level_fused_kernel<1024>, dim3(L.fuse_task_count), dim3(1024), lds, s, (const LevelTask*)(h->d_level_tasks + L.fuse_task_begin), ticket,
                           (const int32_t*)h->d_lists, (const FrontDesc*)h->d_fronts, (const FrontFac*)h->d_ffac, (const FacDesc*)h->d_fd,
                           (const ChildRef*)h->d_childs, (const int32_t*)h->d_cmap, (const int32_t*)h->d_fxoff, h->pool, lambda_v, lambda_p,
                           (const double*)h->dampw, h->d_status, L.fuse_nmax, L.fuse_jcap, (const double*)h->gex_active, ML, (const int32_t*)h->d_rowptr,
                           (const RowSrc*)h->d_rowsrc, h->inv16_med, h->d_level_sync + L.med_begin);
      else
        hipLaunchKernelGGL(level_fused_kernel<256>, dim3(L.fuse_task_count), dim3(256), lds, s, (const LevelTask*)(h->d_level_tasks + L.fuse_task_begin), ticket,
                           (const int32_t*)h->d_lists, (const FrontDesc*)h->d_fronts, (const FrontFac*)h->d_ffac, (const FacDesc*)h->d_fd,
                           (const ChildRef*)h->d_childs, (const int32_t*)h->d_cmap, (const int32_t*)h->d_fxoff, h->pool, lambda_v, lambda_p,
                           (const double*)h->dampw, h->d_status, L.fuse_nmax, L.fuse_jcap, (const double*)h->gex_active, ML, (const int32_t*)h->d_rowptr,
                           (const RowSrc*)h->d_rowsrc, h->inv16_med, h->d_level_sync + L.med_begin);
      h->kt.end(kt, s);
    }
    for (int b = 0; b < kNumBins && seg == -1 && !fused; b++) {
      const int cnt = L.bin_begin[b + 1] - L.bin_begin[b];
      if (cnt == 0) continue;
      const int nmax = kBinN[b % 6], srows = L.bin_srows[b];
      // gather leaves keep only nf rows: one wave per front (barriers become free, ~2.5x more fronts resident per CU)
      const int threads = (b >= 6 || b % 6 == 0) ? 64 : (b % 6 == 1 ? 128 : 256);
      const int jcap = L.bin_jcap[b];
      const size_t lds = kLdsFrontExtra - (size_t)(LDSF_JCAP - jcap) * 8 + 64 + (size_t)srows * nmax * sizeof(double);
      const int kt = h->kt.begin(LMGPU_KT_LDS_FRONT, s);
      // a handful of wide fronts (the upper levels of a general sparse tree, where a launch lasts as long as its slowest front and the
      // device is idle): sixteen waves per front -- the rank-4 trailing updates, the extend-add and the emission all scale with them
      const bool wide16 = b >= 3 && b < 6 && cnt <= h->wide16_max && !h->no_wide16;
      if (wide16)
        hipLaunchKernelGGL((lds_front_kernel<false, 1024>), dim3(cnt), dim3(1024), lds, s,
                           (const int32_t*)(h->d_lists + L.list_begin + L.bin_begin[b]), (const FrontDesc*)h->d_fronts,
                           (const FrontFac*)h->d_ffac, (const FacDesc*)h->d_fd, (const ChildRef*)h->d_childs, (const int32_t*)h->d_cmap,
                           (const int32_t*)h->d_fxoff, h->pool, lambda_v, lambda_p, (const double*)h->dampw, h->d_status, nmax, srows, h->d_gcorner, jcap,
                           (const double*)h->gex_active, (const char*)(h->d_leafpack && L.pack_stride[b] ? h->d_leafpack + L.pack_off[b] : nullptr),
                           L.pack_stride[b]);
      else if (b < 6)
        hipLaunchKernelGGL(lds_front_kernel<false>, dim3(cnt), dim3(threads), lds, s,
                           (const int32_t*)(h->d_lists + L.list_begin + L.bin_begin[b]), (const FrontDesc*)h->d_fronts,
                           (const FrontFac*)h->d_ffac, (const FacDesc*)h->d_fd, (const ChildRef*)h->d_childs, (const int32_t*)h->d_cmap,
                           (const int32_t*)h->d_fxoff, h->pool, lambda_v, lambda_p, (const double*)h->dampw, h->d_status, nmax, srows, h->d_gcorner, jcap,
                           (const double*)h->gex_active, (const char*)(h->d_leafpack && L.pack_stride[b] ? h->d_leafpack + L.pack_off[b] : nullptr),
                           L.pack_stride[b]);
      else
        hipLaunchKernelGGL(lds_front_kernel<true>, dim3(cnt), dim3(threads), lds, s,
                           (const int32_t*)(h->d_lists + L.list_begin + L.bin_begin[b]), (const FrontDesc*)h->d_fronts,
                           (const FrontFac*)h->d_ffac, (const FacDesc*)h->d_fd, (const ChildRef*)h->d_childs, (const int32_t*)h->d_cmap,
                           (const int32_t*)h->d_fxoff, h->pool, lambda_v, lambda_p, (const double*)h->dampw, h->d_status, nmax, srows, h->d_gcorner, jcap,
                           (const double*)h->gex_active, (const char*)(h->d_leafpack && L.pack_stride[b] ? h->d_leafpack + L.pack_off[b] : nullptr),
                           L.pack_stride[b]);
      h->kt.end(kt, s);
    }
    if (L.med_count > 0 && !fused) {  // medium fronts of this level: six launches for all of them
      const MedLevel ML{(const MedFront*)(h->d_med_fronts + L.med_begin)};
      const unsigned cnt = (unsigned)L.med_count;
      int ktm = h->kt.begin(LMGPU_KT_HBM_ASSEMBLE, s);
      if (L.med_max_fac > 0 || L.med_max_child > 0)
        hipLaunchKernelGGL(med_assemble_rows_kernel, dim3((L.med_max_n + 3) / 4, cnt), dim3(256), 0, s, ML, (const int32_t*)h->d_rowptr, (const RowSrc*)h->d_rowsrc, (const ChildRef*)h->d_childs, (const int32_t*)h->d_cmap,
                           (const FrontFac*)h->d_ffac, (const FacDesc*)h->d_fd, h->pool, (const int32_t*)h->d_fxoff, lambda_v, lambda_p,
                           (const double*)h->dampw, (const double*)h->gex_active);
      else  // (the row-owner assembly damps its own rows)
        hipLaunchKernelGGL(med_damp_kernel, dim3((L.med_max_nf + 255) / 256, cnt), dim3(256), 0, s, ML, (const int32_t*)h->d_fxoff, h->pool, lambda_v,
                           lambda_p, (const double*)h->dampw, (const double*)h->gex_active);
      h->kt.end(ktm, s);
      ktm = h->kt.begin(LMGPU_KT_PANEL, s);
      hipLaunchKernelGGL(med_diag_potrf_kernel, dim3(cnt), dim3(256), DIAG_LDS_BYTES, s, ML, h->pool, h->d_status, h->inv16_med);
      if (L.med_max_cols > 0)
        hipLaunchKernelGGL(med_panel_trsm_kernel, dim3((L.med_max_cols + 63) / 64, cnt), dim3(256), 0, s, ML, h->pool, (const double*)h->inv16_med);
      h->kt.end(ktm, s);
      if (L.med_max_cols > 0) {
        const int S = (L.med_max_cols + 63) / 64;
        ktm = h->kt.begin(LMGPU_KT_SYRK, s);
        hipLaunchKernelGGL(med_syrk_kernel, dim3(4 * S, S, cnt), dim3(256), 0, s, ML, h->pool);
        h->kt.end(ktm, s);
      }
    }
    for (int fi : L.hbm) {
      if (h->is_med[fi]) continue;
      const FrontDesc& F = h->h_fronts[fi];
      const int64_t off = h->f_off[fi];
      const int ld = h->f_ld[fi];
      double* A = h->pool + off;
      const bool replicated = (F.pad & 1) != 0, own_terms = (F.pad & 2) == 0;
      const lmgpu_handle::GatherRange& G = h->gather[fi];
      const int np = (F.nf + NBO - 1) / NBO;
      const int nchunks = (F.n + NBO - 1) / NBO;
      // `split`: the assembled contributions go to a buffer of their own (aoff) and reach the working matrix A (which starts
      // from zero and collects the trailing updates) in 256-row chunks, each just before its panel is factored.
      //   multi-rank   : the chunks are summed over the ranks (RCCL on the communication stream / the in-process group)
      // (gathering the chunks on a second stream beside the factorisation was measured in round 1 -- 13.5 vs 10.4 ms per step: two
      //  resident workgroups of the update kernel hold every VGPR of a SIMD, the gather waves only get in between -- and removed)
      const bool multi = replicated && (h->comm || h->lgroup);
      const bool split = h->s_off[fi] >= 0 && multi;
      const int64_t aoff = split ? h->s_off[fi] : off;
      double* Asm = h->pool + aoff;
      hipStream_t sa = s;
      if (split)
        while ((int)h->chunk_ev.size() < 2 * nchunks) {
          hipEvent_t e;
          HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
          h->chunk_ev.push_back(e);
        }
      // chunk c = rows [256 c, 256 (c+1)) from the first column of its diagonal block to the end of its last row
      auto chunk_range = [&](int c, size_t& begin, size_t& count) {
        const int r0c = c * NBO, rows = std::min(F.n, r0c + NBO) - r0c;
        begin = (size_t)r0c * ld + r0c;  // r0c is a multiple of 256: 16-aligned
        count = (size_t)rows * ld - r0c;
      };
      int kt = h->kt.begin(LMGPU_KT_HBM_ASSEMBLE, sa);
      const bool gwrite = gather_writes(h, fi);
      if (gwrite && G.zero_count > 0)
        hipLaunchKernelGGL(zero_blocks_kernel, dim3(G.zero_count), dim3(128), 0, sa, (const GZeroBlock*)(h->d_gzero + G.zero_begin), h->pool, aoff, ld);
      // own factors and children's update matrices: one wave per row of the front walks that row's sources in a fixed order (no
      // atomics, bitwise reproducible); with_factors = 0 on the ranks that leave the own terms to rank 0
      auto assemble_rows = [&](bool with_factors) {
        if (h->row_begin[fi] >= 0 && (F.child_count > 0 || (with_factors && F.fac_count > 0)))
          hipLaunchKernelGGL(hbm_assemble_rows_kernel, dim3((F.n + 3) / 4), dim3(256), 0, sa, F, aoff, ld, (const int32_t*)(h->d_rowptr + h->row_begin[fi]),
                             (const RowSrc*)h->d_rowsrc, (const ChildRef*)h->d_childs, (const int32_t*)h->d_cmap, (const FrontFac*)h->d_ffac,
                             (const FacDesc*)h->d_fd, h->pool, with_factors ? 1 : 0);
      };
      assemble_rows(own_terms && !gwrite);
      auto own_additive_terms = [&]() {  // the front's own factors and the damping (added: after whatever initialises the entries)
        if (F.fac_count > 0 && own_terms) assemble_rows(true);  // (a gather-write front has no update-matrix children: only its factors are added here)
        if (own_terms)
          hipLaunchKernelGGL(hbm_damp_kernel, dim3((F.nf + 255) / 256), dim3(256), 0, sa, F, aoff, ld, (const int32_t*)h->d_fxoff, h->pool, lambda_v,
                             lambda_p, (const double*)h->dampw, (const double*)h->gex_active);
      };
      if (!gwrite && own_terms)
        hipLaunchKernelGGL(hbm_damp_kernel, dim3((F.nf + 255) / 256), dim3(256), 0, sa, F, aoff, ld, (const int32_t*)h->d_fxoff, h->pool, lambda_v,
                           lambda_p, (const double*)h->dampw, (const double*)h->gex_active);
      // leaf children in Schur form: deterministic gather instead of atomics; rows [c0, c1) of the chunk table
      const int schur_masked = dev_switch("LMGPU_SCHUR_UNMASKED") ? 0 : 2;  // (operand loads that only the lanes holding an element take part in)
      auto gather_chunks = [&](int c0, int c1, bool whole) {
        const int s0 = whole ? 0 : G.cs[c0], s1 = whole ? G.pblk_short : G.cs[c1];
        const int l0 = whole ? 0 : G.cl[c0], l1 = whole ? G.pblk_long : G.cl[c1];
        const int v0 = whole ? 0 : G.cv[c0], v1 = whole ? G.vblk_count : G.cv[c1];
        if (s1 > s0)
          hipLaunchKernelGGL((schur_pairs_kernel<1>), dim3((s1 - s0 + 7) & ~7), dim3(64), 0, sa, (const GPairBlock*)(h->d_gpblk + G.pblk_begin + s0),
                             (const GPairEntry*)h->d_gpent, h->pool, aoff, ld, s1 - s0, (gwrite ? 1 : 0) | schur_masked);
        if (l1 > l0)
          hipLaunchKernelGGL((schur_pairs_kernel<4>), dim3((l1 - l0 + 7) & ~7), dim3(256), 0, sa,
                             (const GPairBlock*)(h->d_gpblk + G.pblk_begin + G.pblk_short + l0), (const GPairEntry*)h->d_gpent, h->pool, aoff, ld,
                             l1 - l0, (gwrite ? 1 : 0) | schur_masked);
        if (v1 > v0)
          hipLaunchKernelGGL(schur_factor_kernel, dim3(v1 - v0), dim3(64 * SCHUR_FW), 0, sa, (const GVarBlock*)(h->d_gvblk + G.vblk_begin + v0),
                             (const GVarEntry*)h->d_gvent, h->pool, aoff, ld, F.n, schur_masked);
        if (G.leaf_count > 0 && (whole || c1 == nchunks)) {  // the (rhs, rhs) corner lives in the last chunk
          double* scal = h->dscal + 4;
          reduce_to(h, h->d_gcorner + G.leaf_begin, G.leaf_count, scal, sa, nullptr);
          hipLaunchKernelGGL(add_scalar_kernel, dim3(1), dim3(1), 0, sa, Asm + (size_t)(F.n - 1) * ld + F.n - 1, (const double*)scal);
        }
      };
      gather_chunks(0, nchunks, true);
      if (gwrite) own_additive_terms();
      h->kt.end(kt, sa);
      if (multi && h->comm && split) {  // RCCL: all chunks queued on the communication stream, one event each
        HIPCHECK(hipEventRecord(h->asm_ev, s));
        HIPCHECK(hipStreamWaitEvent(h->comm_stream, h->asm_ev, 0));
        for (int c = 0; c < nchunks; c++) {
          size_t cb, cn;
          chunk_range(c, cb, cn);
          NCCLCHECK(ncclAllReduce(Asm + cb, Asm + cb, cn, ncclDouble, ncclSum, h->comm_data ? h->comm_data : h->comm, h->comm_stream));
          HIPCHECK(hipEventRecord(h->chunk_ev[c], h->comm_stream));
        }
      }
      // chunk c is complete (gathered / reduced) for everything queued on the main stream after this call
      auto wait_chunk = [&](int c) -> int {
        if (!split || c >= nchunks) return LMGPU_OK;
        if (h->comm) {
          HIPCHECK(hipStreamWaitEvent(s, h->chunk_ev[c], 0));
        } else {  // in-process group: synchronous
          size_t cb, cn;
          chunk_range(c, cb, cn);
          const int rca = allreduce_sum(h, Asm + cb, cn, s);
          if (rca) return rca;
        }
        return LMGPU_OK;
      };
      // fold chunk c into the working matrix (which already carries the trailing updates of earlier panels)
      auto add_chunk = [&](int c) -> int {
        if (!split || c >= nchunks) return LMGPU_OK;
        const int ktc = h->kt.begin(LMGPU_KT_ALLREDUCE, s);
        const int rcw = wait_chunk(c);
        if (rcw) return rcw;
        size_t cb, cn;
        chunk_range(c, cb, cn);
        hipLaunchKernelGGL(local_sum_kernel, dim3(std::min<size_t>(2048, (cn + 255) / 256)), dim3(256), 0, s, A + cb, (const double*)(Asm + cb), cn);
        h->kt.end(ktc, s, (double)cn * 8.0);
        return LMGPU_OK;
      };
      { const int rc0 = add_chunk(0); if (rc0) return rc0; }
      // Blocked right-looking partial Cholesky, outer panels of NBO = 256 rows.  Panel 0 is one dataflow launch
      // (panel_dataflow_kernel); after that ONE launch per outer panel i (step_kernel): trailing update with panel i
      // + factorisation of panel i+1 beside/behind it (look-ahead inside the launch, kernels_step.hpp).  A panel whose row
      // count is not a multiple of 64 (the last one) takes the two-launch form diag_potrf_kernel + panel_trsm_kernel.
      if (np + 1 > h->pflags_panels) {
        h->err = "panel flag buffer too small";
        return LMGPU_INVALID;
      }
      HIPCHECK(hipMemsetAsync(h->d_pflags, 0, (size_t)(np + 1) * PDF_FLAG_WORDS * sizeof(unsigned int), s));
      auto rows_of = [&](int i) { return std::min(F.nf, (i + 1) * NBO) - i * NBO; };
      auto dataflow_ok = [&](int i) {  // panel i can run as block-column workgroups with flag hand-offs
        return rows_of(i) % 64 == 0 && !h->two_launch_panel && (F.n - i * NBO + 63) / 64 <= PDF_MAX_COLTILES;
      };
      auto panel_flop = [&](int i) {
        const double kb = rows_of(i), cols = F.n - i * NBO - kb;
        return kb * kb * kb / 3.0 + kb * kb * cols;
      };
      auto panel_alone = [&](int i) {
        const int k0 = i * NBO, kb = rows_of(i), cols = F.n - k0 - kb;
        const int ktp = h->kt.begin(LMGPU_KT_PANEL, s);
        if (dataflow_ok(i)) {
          hipLaunchKernelGGL(panel_dataflow_kernel, dim3(kb / 64 + (cols + 63) / 64), dim3(256), PDF_LDS_BYTES, s, A, ld, F.n, F.nf, k0, kb, F.id,
                             h->d_status, h->inv16 + (size_t)i * 4096, h->d_pflags + (size_t)i * PDF_FLAG_WORDS);
        } else {
          hipLaunchKernelGGL(diag_potrf_kernel, dim3(1), dim3(256), DIAG_LDS_BYTES, s, A, ld, F.nf, k0, kb, F.id, h->d_status, h->inv16 + (size_t)i * 4096);
          if (cols > 0)
            hipLaunchKernelGGL(panel_trsm_kernel, dim3((cols + 63) / 64), dim3(256), 0, s, A, ld, F.n, k0, kb, (const double*)(h->inv16 + (size_t)i * 4096));
        }
        h->kt.end(ktp, s, panel_flop(i));
      };
      panel_alone(0);
      h->inv16_owner = fi;  // the per-panel 16x16 inverses now belong to this front (read again by its back-substitution)
      int kt_run = -1, run_launches = 0;  // one event pair around a run of consecutive step launches
      double run_flop = 0;
      auto close_run = [&]() {
        if (run_launches > 0) h->kt.end(kt_run, s, run_flop, run_launches);
        kt_run = -1;
        run_launches = 0;
        run_flop = 0;
      };
      // a step that can be fused with the factorisation of the next panel; consecutive ones with full panels go as ONE launch
      auto fusable = [&](int i) { return (i + 1 < np) && dataflow_ok(i + 1) && !h->no_fuse && F.n - i * NBO - rows_of(i) > 0; };
      // multi-rank (RCCL): the steps still go as chained launches -- their head tiles fold the all-reduced row chunks in -- but in
      // SEGMENTS of 1, 1, 2, 4, 8, ... steps, each launched behind a stream wait for the event of the last chunk it touches: no
      // workgroup ever waits for the network inside a launch (nothing to deadlock on), the first panels start after two chunks, and
      // the communication stream gets further ahead with every segment (one launch per step cost 6.9 vs 6.2 ms for the C4 root in
      // round 1).  The in-process test communicator sums on the host between the launches and keeps the per-step form.
      const bool chain_split = split && h->comm != nullptr;
      auto chainable = [&](int i) {
        return fusable(i) && (!split || chain_split) && !h->no_chain && rows_of(i) == NBO && (F.n - (i + 1) * NBO + 127) / 128 <= PDF_MAX_CHAIN_T;
      };
      for (int i = 0; i < np; i++) {
        const int k0 = i * NBO, kb = rows_of(i), r0 = k0 + kb, m = F.n - r0;
        if (m <= 0) break;
        if (chainable(i) && chainable(i + 1)) {
          // the run of chainable steps starting here: one launch (single rank) or a few segment launches (multi-rank), ticket order
          // built once per front and segment (chain_schedule)
          int run = 0;
          while (chainable(i + run)) run++;
          std::vector<lmgpu_handle::ChainPlan>& plans = h->chain_plans[fi];
          if (plans.empty() || plans[0].i0 != i) {
            for (auto& p : plans)
              if (p.d_tasks) HIPCHECK(hipFree(p.d_tasks));
            plans.clear();
            int at = i, seg = 1, nseg = 0;
            while (at < i + run) {
              lmgpu_handle::ChainPlan cp;
              cp.i0 = at;
              cp.nsteps = chain_split ? std::min(seg, i + run - at) : run;
              if (chain_split && i + run - (at + cp.nsteps) == 1) cp.nsteps++;  // no one-step remainder
              for (int q = 0; q < cp.nsteps; q++) {
                const int is = at + q, ms = F.n - (is + 1) * NBO;
                cp.flop += 2.0 * NBO * ((double)ms * (ms + 1) / 2.0) + panel_flop(is + 1);
              }
              const std::vector<int2> tasks = chain_schedule(F.n, F.nf, cp.i0, cp.nsteps, h->chain_far_pct, h->chain_merge, h->chain_split_pct);
              cp.ntasks = (int)tasks.size();
              HIPCHECK(hipMalloc((void**)&cp.d_tasks, tasks.size() * sizeof(int2)));
              HIPCHECK(hipMemcpyAsync(cp.d_tasks, tasks.data(), tasks.size() * sizeof(int2), hipMemcpyHostToDevice, s));
              HIPCHECK(hipStreamSynchronize(s));  // `tasks` is a local
              at += cp.nsteps;
              if (++nseg >= 2) seg *= 2;
              plans.push_back(cp);
            }
          }
          close_run();
          for (const lmgpu_handle::ChainPlan& cp : plans) {
            if (split) {  // every row chunk the segment folds in (cp.i0 + 1 .. cp.i0 + cp.nsteps) is summed over the ranks
              const int rcw = wait_chunk(cp.i0 + cp.nsteps);
              if (rcw) return rcw;
            }
            ChainArgs ca{A, ld, F.n, F.nf, cp.i0, cp.nsteps, F.id, h->d_status, h->inv16, h->d_pflags, cp.d_tasks, split ? (const double*)Asm : nullptr};
            const int ktc = h->kt.begin(LMGPU_KT_CHAIN, s);
            hipLaunchKernelGGL(chain_kernel, dim3(cp.ntasks), dim3(256), STEP_LDS_BYTES, s, ca);
            h->kt.end(ktc, s, cp.flop, 1);
          }
          i += run - 1;
          continue;
        }
        // the end of the front as one small launch: update with panel i, factor the last (partial) panel, update what follows
        if (!split && !h->no_tail && i + 2 == np && kb <= TAIL_MAX_KP && m <= TAIL_MAX_M && rows_of(i + 1) < 64) {
          close_run();
          const int ktt = h->kt.begin(LMGPU_KT_PANEL, s);
          hipLaunchKernelGGL(front_tail_kernel, dim3(1), dim3(256), TAIL_LDS_BYTES, s, A, ld, F.n, F.nf, k0, kb, F.id, h->d_status,
                             h->inv16 + (size_t)(i + 1) * 4096);
          h->kt.end(ktt, s, 2.0 * kb * ((double)m * (m + 1) / 2.0) + panel_flop(i + 1));
          break;
        }
        const bool fuse = fusable(i);
        // rows of panel i+1 (and, for i = np-1, of the separator part): a fused step folds them in itself (its 64x64 head tiles
        // cover exactly those rows), otherwise an add kernel does
        const bool fold_in_step = split && fuse && r0 == (i + 1) * NBO;
        if (fold_in_step) {
          const int rcw = wait_chunk(i + 1);
          if (rcw) return rcw;
        } else {
          if (split && i + 1 < nchunks) close_run();  // the fold-in kernel is timed separately
          const int rcc = add_chunk(i + 1);
          if (rcc) return rcc;
        }
        const int T = (m + 127) / 128;
        // algorithmic flop of the update: 2 x kb x (upper-triangle entries of the m x m trailing matrix)
        const double upd_flop = 2.0 * kb * ((double)m * (m + 1) / 2.0);
        if (fuse) {
          const int kbn = rows_of(i + 1);
          StepArgs a{A, ld, F.n, F.nf, k0, kb, kbn, F.id, h->d_status, h->inv16 + (size_t)(i + 1) * 4096, h->d_pflags + (size_t)(i + 1) * PDF_FLAG_WORDS,
                     fold_in_step ? (const double*)Asm : nullptr};
          const int grid = step_grid(m, kbn);
          if (run_launches == 0) kt_run = h->kt.begin(LMGPU_KT_SYRK, s);
          hipLaunchKernelGGL(step_kernel, dim3(grid), dim3(256), STEP_LDS_BYTES, s, a);
          run_launches++;
          run_flop += upd_flop + panel_flop(i + 1);
        } else {
          close_run();
          const int kts = h->kt.begin(LMGPU_KT_SYRK, s);
          if (m <= 1024) {  // a few tiles: one workgroup per 32 x 32 quadrant instead (each 128-tile is K / 4 x 16 dependent MFMAs on one CU)
            const int S = (m + 63) / 64;
            hipLaunchKernelGGL(syrk_quadrants_kernel, dim3(4 * S, S), dim3(256), 0, s, A, ld, F.n, k0, kb, r0);
          } else {
            hipLaunchKernelGGL(syrk_mfma_kernel, dim3(T, T), dim3(256), kSyrkLds, s, A, ld, F.n, k0, kb, r0, F.n);
          }
          h->kt.end(kts, s, upd_flop);
          if (i + 1 < np) panel_alone(i + 1);
        }
      }
      close_run();
      for (int c = np + 1; c < nchunks; c++) {  // separator rows beyond the chunk after the last panel
        const int rcc = add_chunk(c);
        if (rcc) return rcc;
      }
    }
  }
  HIPCHECK(hipGetLastError());
  return LMGPU_OK;
}

// the LDS fronts of level l as one launch sized for the level's largest front (merged and fused launches): false when the level has none, too
// many of them (a launch sized for the largest front would cost the small ones their occupancy), or gather leaves
static bool level_lds_class(const lmgpu_handle* h, int l, int* nmax, int* jcap, int* threads) {
  const int kMergeCap = 160;
  const LevelWork& L = h->levels[l];
  if (L.list_count == 0 || L.list_count > kMergeCap || L.bin_begin[kNumBins] != L.bin_begin[6]) return false;
  int top = -1, jc = 96, cnt_top = 0;
  for (int b = 0; b < 6; b++)
    if (L.bin_begin[b + 1] > L.bin_begin[b]) {
      top = b;
      cnt_top = L.bin_begin[b + 1] - L.bin_begin[b];
      jc = std::max(jc, L.bin_jcap[b]);
    }
  *nmax = kBinN[top];
  *jcap = jc;
  const bool wide16 = top >= 3 && cnt_top <= h->wide16_max && !h->no_wide16;
  *threads = wide16 ? 1024 : (top == 0 ? 64 : 256);  // (four waves from 25 columns on: the blocked Cholesky of the LDS body needs them)
  return true;
}

// ---- back-substitution, top-down (a14)
int do_backsub(lmgpu_handle* h) {
  hipStream_t s = h->stream;
  if (h->cfg.world_size > 1) HIPCHECK(hipMemsetAsync(h->delta, 0, h->ntot * sizeof(double), s));
  // deep trees: the LDS fronts of consecutive levels without HBM fronts in between go as ONE dataflow launch
  const bool merge = h->merge_backsub && h->cfg.world_size == 1 && !(h->cfg.flags & LMGPU_FLAG_SPLIT_ROOT);
  const bool no_bsd_runs = dev_switch("LMGPU_NO_BSD_RUNS") != nullptr;
  const int NFR = (int)h->h_fronts.size();
  if (h->n_fill_backsub < 0 || h->fill_backsub_merge != (int)merge) {
    std::vector<FillChunk> fills;
    if (merge) {
      add_fill(fills, h->d_bs_done, (size_t)(NFR + h->levels.size() + 1) * sizeof(unsigned int), 0u);  // (the ticket counters)
      add_fill(fills, h->delta, h->ntot * sizeof(double), 0xffffffffu);  // "not published yet": merged launches hand x over through delta itself
    }
    if (h->bsd_x_count > 0) {
      add_fill(fills, h->d_bsd_x, h->bsd_x_count * sizeof(double), 0xffffffffu);  // "not published yet"
      add_fill(fills, h->d_bsd_ticket, h->levels.size() * sizeof(unsigned int), 0u);
    }
    if (h->d_fill_backsub) {
      HIPCHECK(hipStreamSynchronize(s));
      (void)hipFree(h->d_fill_backsub);
      h->d_fill_backsub = nullptr;
    }
    h->n_fill_backsub = (int)fills.size();
    h->fill_backsub_merge = (int)merge;
    if (!fills.empty()) {
      const int rcu = upload(h, &h->d_fill_backsub, fills);
      if (rcu) return rcu;
    }
  }
  if (h->n_fill_backsub > 0) hipLaunchKernelGGL(fill_chunks_kernel, dim3(h->n_fill_backsub), dim3(256), 0, s, (const FillChunk*)h->d_fill_backsub);
  int seg_hi = -1, seg_lo = -1;  // levels of the pending segment (top, bottom)
  auto run_level = [&](const LevelWork& L) {  // one level as a launch of its own
    if (L.lds_nf_max > LDSB_SMALL_NF) {
      hipLaunchKernelGGL(lds_backsub_wide_kernel, dim3(L.list_count), dim3(256), (size_t)(L.lds_rsd_max + LDSB_TAIL) * sizeof(double), s,
                         (const int32_t*)(h->d_lists + L.list_begin), L.list_count, (const FrontDesc*)h->d_fronts, (const int32_t*)h->d_fxoff,
                         (const int32_t*)h->d_sxoff, (const double*)h->pool, h->delta, h->d_status);
    } else {
      hipLaunchKernelGGL(lds_backsub_kernel, dim3((L.list_count + 3) / 4), dim3(256), 0, s, (const int32_t*)(h->d_lists + L.list_begin),
                         L.list_count, (const FrontDesc*)h->d_fronts, (const int32_t*)h->d_fxoff, (const int32_t*)h->d_sxoff,
                         (const double*)h->pool, h->delta, h->d_status);
    }
  };
  auto flush_segment = [&]() -> int {
    if (seg_hi < 0) return LMGPU_OK;
    const int kt = h->kt.begin(LMGPU_KT_BACKSUB_LDS, s);
    if (seg_hi == seg_lo) {
      run_level(h->levels[seg_hi]);
    } else {
      const int b = h->levels[seg_lo].list_begin, e = h->levels[seg_hi].list_begin + h->levels[seg_hi].list_count;
      int rsdmax = 0;
      for (int l = seg_lo; l <= seg_hi; l++) rsdmax = std::max(rsdmax, h->levels[l].lds_rsd_max);
      hipLaunchKernelGGL(lds_backsub_merged_kernel, dim3(e - b), dim3(256), (size_t)(rsdmax + LDSB_TAIL) * sizeof(double), s, (const int32_t*)h->d_lists, b, e,
                         (const FrontDesc*)h->d_fronts, (const int32_t*)h->d_fxoff, (const int32_t*)h->d_sxoff, (const double*)h->pool, h->delta,
                         h->d_status, (const int32_t*)h->d_bs_parent, (const int32_t*)h->d_bs_pos, h->d_bs_done, h->d_bs_done + NFR + seg_hi);
    }
    h->kt.end(kt, s);
    seg_hi = seg_lo = -1;
    return LMGPU_OK;
  };
  for (int li = (int)h->levels.size() - 1; li >= 0; li--) {
    const LevelWork& L = h->levels[li];
    if (merge && !L.hbm.empty()) {  // this level's HBM fronts need every ancestor: finish the pending LDS levels first
      const int rcf = flush_segment();
      if (rcf) return rcf;
    }
    const int brun = (merge && !h->bsd_run_of.empty() && !no_bsd_runs) ? h->bsd_run_of[li] : -1;
    if (brun >= 0) {  // this level and the ones below it that hold nothing but block-solved fronts: one launch, separator values awaited by value
      const lmgpu_handle::BsdRun& R = h->bsd_runs[brun];
      const int kt = h->kt.begin(LMGPU_KT_BACKSUB_HBM, s);
      hipLaunchKernelGGL(hbm_backsolve_blocks_kernel, dim3(R.count), dim3(256), 0, s, (const BsdBlock*)(h->d_bsd_run_table + R.begin), h->d_bsd_ticket + li,
                         (const int32_t*)h->d_fxoff, (const int32_t*)h->d_sxoff, (const double*)h->pool, h->delta, h->d_bsd_x, h->d_status, 1);
      h->kt.end(kt, s);
    }
    if (L.bsd_count > 0 && brun == -1) {  // the smaller HBM fronts of the level: one workgroup per 64-row block, one launch
      const int kt = h->kt.begin(LMGPU_KT_BACKSUB_HBM, s);
      hipLaunchKernelGGL(hbm_backsolve_blocks_kernel, dim3(L.bsd_count), dim3(256), 0, s, (const BsdBlock*)(h->d_bsd_table + L.bsd_begin),
                         (L.bsd_count <= h->num_cus && !h->bsd_ticket) ? (unsigned int*)nullptr : h->d_bsd_ticket + li, (const int32_t*)h->d_fxoff,
                         (const int32_t*)h->d_sxoff, (const double*)h->pool, h->delta, h->d_bsd_x, h->d_status);
      h->kt.end(kt, s);
    }
    for (int fi : L.hbm) {
      const FrontDesc& F = h->h_fronts[fi];
      if (F.nf <= BSS_MAX_NF) continue;  // done above
      const int64_t off = h->f_off[fi];
      const int ld = h->f_ld[fi];
      const int kt = h->kt.begin(LMGPU_KT_BACKSUB_HBM, s);
      const int nblk = (F.nf + NB - 1) / NB;
      const bool has_sep = F.n - F.nf - 1 > 0;
      if (has_sep)  // y = d - S x_S (a root reads d in place)
        hipLaunchKernelGGL(hbm_rhs_init_kernel, dim3(F.nf), dim3(64), 0, s, F, off, ld, (const int32_t*)h->d_sxoff, (const double*)h->pool,
                           (const double*)h->delta, h->ywork);
      // ONE dataflow launch: workgroup b waits for x_j (j > b); the inverses of the diagonal blocks are built inside it when the 16 x 16
      // inverses of this front's factorisation are still there (else by a launch of their own in front of it)
      HIPCHECK(hipMemsetAsync(h->bs_flags, 0, (nblk + 1) * sizeof(unsigned int), s));  // flags + ticket
      HIPCHECK(hipMemsetAsync(h->bs_x, 0xff, (size_t)nblk * NB * sizeof(double), s));  // sentinel: "not published yet"
      if (h->inv16_owner == fi && !dev_switch("LMGPU_NO_INV16_REUSE")) {
        hipLaunchKernelGGL(hbm_backsolve_dataflow2_kernel, dim3(nblk), dim3(256), 0, s, F, off, ld, (const int32_t*)h->d_fxoff, (const double*)h->pool,
                           (const double*)h->inv16, has_sep ? (const double*)h->ywork : (const double*)nullptr, h->bs_x, h->bs_flags, h->delta,
                           h->d_status);
      } else {
        if (!has_sep)
          hipLaunchKernelGGL(hbm_rhs_init_kernel, dim3(F.nf), dim3(64), 0, s, F, off, ld, (const int32_t*)h->d_sxoff, (const double*)h->pool,
                             (const double*)h->delta, h->ywork);
        hipLaunchKernelGGL((hbm_invert_diag_kernel<NB>), dim3(nblk), dim3(NB), 0, s, (const double*)(h->pool + off), ld, F.nf, h->bs_inv);
        hipLaunchKernelGGL((hbm_backsolve_dataflow_kernel<NB>), dim3(nblk), dim3(256), 0, s, F, off, ld, (const int32_t*)h->d_fxoff,
                           (const double*)h->pool, (const double*)h->bs_inv, (const double*)h->ywork, h->bs_x, h->bs_flags, h->delta,
                           h->d_status);
      }
      h->kt.end(kt, s);
    }
    if (merge) {  // accumulated; flushed before the next HBM front or at the end
      if (L.list_count > 0) {
        if (seg_hi < 0) seg_hi = li;
        seg_lo = li;
      }
      if (li == 0 || !h->levels[li - 1].hbm.empty()) {
        const int rcf = flush_segment();
        if (rcf) return rcf;
      }
    } else if (L.list_count > 0) {
      const int kt = h->kt.begin(LMGPU_KT_BACKSUB_LDS, s);
      run_level(L);
      h->kt.end(kt, s);
    }
  }
  HIPCHECK(hipGetLastError());
  return LMGPU_OK;
}

// Solve the damped system.  do_solve_enqueue queues everything (eliminate, back-substitute, the two linear errors and the
// copies of the scalars / the status word to pinned host memory) without waiting; do_solve_finish waits for the stream and
// returns LMGPU_OK / LMGPU_INDETERMINATE with the linear errors in h_scal[1], h_scal[2].  tryLambda queues the retraction and
// the new error behind the solve BEFORE it waits (one host round trip per inner iteration instead of two); their result is
// simply not used when the solve failed or the linearised cost went up.
__global__ void set_scalar_kernel(double* p, double v) { *p = v; }


// everything of one damped solve that is queued on the stream
int solve_sequence(lmgpu_handle* h, bool phase_events, double lambda_v, const double* lambda_p) {
  hipStream_t s = h->stream;
  int rc = do_eliminate(h, lambda_v, lambda_p);
  if (rc) return rc;
  if (phase_events) (void)hipEventRecord(h->ev[2], s);
  rc = do_backsub(h);
  if (rc) return rc;
  if (phase_events) (void)hipEventRecord(h->ev[3], s);
  const int kt = h->kt.begin(LMGPU_KT_LINEAR_ERROR, s);
  if (h->nfac > 0)
    hipLaunchKernelGGL(linear_error_kernel, dim3((h->nfac + 255) / 256), dim3(256), 0, s, (const FacDesc*)h->d_fd, h->nfac,
                       (const double*)h->pool, (const double*)h->delta, h->ebuf0, h->ebuf1);
  reduce_to(h, h->ebuf0, h->n_counted, h->dscal + 1);
  reduce_to(h, h->ebuf1, h->n_counted, h->dscal + 2);
  h->kt.end(kt, s);
  rc = allreduce_scalars(h, 1, 2);
  if (rc) return rc;
  { const int rcs = allreduce_min_int(h, h->d_status, s); if (rcs) return rcs; }
  return LMGPU_OK;
}

// Deep clique trees (general SLAM graphs: 20-130 levels, several launches per level) are bound by launch latency: 265 launches
// of 3.5 ms total kernel time took 6.8 ms per solve on victoria_park / COLAMD.  The launch sequence of a solve depends only on
// the structure, so after the first (eager) solve it is captured into a hipGraph once and replayed; the only per-solve
// input, lambda, lives in device memory.  Not with kernel timers on (their events sit between the launches), with several
// ranks (collectives and host-side rendezvous inside the sequence), or for shallow trees (nothing to gain; the per-phase
// events of the eager path stay available there).
static bool graph_eligible(const lmgpu_handle* h) {
  return h->use_graph && !h->kt.on && h->cfg.world_size == 1 && !h->comm && !h->lgroup && !(h->cfg.flags & LMGPU_FLAG_SPLIT_ROOT) &&
         h->eager_solves >= 1;
}

int do_solve_enqueue(lmgpu_handle* h, double lambda) {
  hipStream_t s = h->stream;
  (void)hipEventRecord(h->ev[1], s);
  int rc;
  if (graph_eligible(h)) {
    hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, s, h->d_lambda, lambda);  // the replay reads lambda from device memory
    hipGraphExec_t& exec = h->solve_graph[h->gex_active ? 1 : 0];
    if (!exec) {
      hipGraph_t graph = nullptr;
      HIPCHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      rc = solve_sequence(h, false, 0.0, h->d_lambda);
      const hipError_t ec = hipStreamEndCapture(s, &graph);
      if (rc) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
      }
      HIPCHECK(ec);
      HIPCHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      (void)hipGraphDestroy(graph);
    }
    HIPCHECK(hipGraphLaunch(exec, s));
    // no phase boundaries inside a replay: the whole sequence is booked as elimination time
    (void)hipEventRecord(h->ev[2], s);
    (void)hipEventRecord(h->ev[3], s);
  } else {
    rc = solve_sequence(h, true, lambda, nullptr);
    if (rc) return rc;
    h->eager_solves++;
  }
  (void)hipEventRecord(h->ev[4], s);
  HIPCHECK(hipMemcpyAsync(h->h_scal + 1, h->dscal + 1, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHECK(hipMemcpyAsync(h->h_status, h->d_status, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
  return LMGPU_OK;
}

int do_solve_finish(lmgpu_handle* h) {
  HIPCHECK(hipStreamSynchronize(h->stream));
  h->kt.resolve();
  h->solved = true;
  if (h->h_status[1] != 0) {  // a bounded spin of an in-launch hand-off ran out: a scheduling / protocol fault, NOT an ill-conditioned system
    h->err = "in-launch dataflow hand-off timed out in front " + std::to_string(h->h_status[1] - 1) + " (spin bound hit)";
    return LMGPU_HIP_ERROR;
  }
  if (*h->h_status < (int)h->h_fronts.size()) {
    const Front& fr = h->plan.fronts[*h->h_status];
    h->failed_slot = fr.vars[0];
    return LMGPU_INDETERMINATE;
  }
  return LMGPU_OK;
}

int do_solve(lmgpu_handle* h, double lambda) {
  const int rc = do_solve_enqueue(h, lambda);
  if (rc) return rc;
  return do_solve_finish(h);
}

int do_retract(lmgpu_handle* h, int from, int to) {
  for (int t = 0; t < kNumVarTypes; t++) {
    const int n = h->plan.type_count[t];
    if (n == 0) continue;
    hipLaunchKernelGGL(retract_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, t, n, (const double*)h->vals[from][t], h->vals[to][t],
                       (const int32_t*)h->type_xoff[t], (const double*)h->delta, (const int32_t*)nullptr);
  }
  HIPCHECK(hipGetLastError());
  return LMGPU_OK;
}

void accumulate_times(lmgpu_handle* h, bool with_retract) {
  float ms = 0;
  if (hipEventElapsedTime(&ms, h->ev[1], h->ev[2]) == hipSuccess) h->tim.eliminate_ms += ms;
  if (hipEventElapsedTime(&ms, h->ev[2], h->ev[3]) == hipSuccess) h->tim.backsub_ms += ms;
  if (hipEventElapsedTime(&ms, h->ev[3], h->ev[4]) == hipSuccess) h->tim.linear_error_ms += ms;
  if (with_retract && hipEventElapsedTime(&ms, h->ev[5], h->ev[6]) == hipSuccess) h->tim.retract_error_ms += ms;
}

// values[cur ^ 1] = retract(values[cur], delta); its nonlinear error into h_scal[0] (queued, no wait)
int enqueue_retract_and_error(lmgpu_handle* h) {
  (void)hipEventRecord(h->ev[5], h->stream);
  const int kt = h->kt.begin(LMGPU_KT_RETRACT_ERROR, h->stream);
  int rc = do_retract(h, h->cur, h->cur ^ 1);
  if (rc) return rc;
  launch_factors<false>(h, h->cur ^ 1);
  reduce_to(h, h->ebuf0, h->n_counted, h->dscal);
  h->kt.end(kt, h->stream);
  rc = allreduce_scalars(h, 0, 1);
  if (rc) return rc;
  (void)hipEventRecord(h->ev[6], h->stream);
  HIPCHECK(hipMemcpyAsync(h->h_scal, h->dscal, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  return LMGPU_OK;
}

// LevenbergMarquardtOptimizer::tryLambda  (gtsam/nonlinear/LevenbergMarquardtOptimizer.cpp:121-270)
int try_lambda(lmgpu_handle* h, const lmgpu_lm_params* p, bool* done) {
  lmgpu_lm_state& st = h->lm;
  double modelFidelity = 0.0;
  bool step_is_successful = false, stopSearchingLambda = false;
  double newError = std::numeric_limits<double>::infinity(), costChange = 0.0;
  int rc = do_solve_enqueue(h, st.lambda);
  if (rc) return rc;
  rc = enqueue_retract_and_error(h);  // speculative: used only if the solve succeeded and the linearised cost did not go up
  if (rc) return rc;
  rc = do_solve_finish(h);
  if (rc != LMGPU_OK && rc != LMGPU_INDETERMINATE) return rc;
  const bool solved = (rc == LMGPU_OK);
  bool retracted = false;
  if (solved) {
    const double oldLin = h->h_scal[1], newLin = h->h_scal[2];
    const double linearizedCostChange = oldLin - newLin;
    if (linearizedCostChange >= 0) {
      newError = h->h_scal[0];
      retracted = true;
      costChange = st.error - newError;
      if (linearizedCostChange > std::numeric_limits<double>::epsilon() * oldLin) {
        modelFidelity = costChange / linearizedCostChange;
        step_is_successful = modelFidelity > p->minModelFidelity;
      }
      const double minAbsoluteTolerance = p->relativeErrorTol * st.error;
      if (std::abs(costChange) < minAbsoluteTolerance) stopSearchingLambda = true;
    }
  }
  accumulate_times(h, retracted);
  h->tim.inner_iterations += 1;
  if (step_is_successful) {
    // decreaseLambda (LevenbergMarquardtState.h:80-93)
    double newLambda = st.lambda, newFactor = st.currentFactor;
    if (p->useFixedLambdaFactor) {
      newLambda /= st.currentFactor;
    } else {
      newLambda *= std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * modelFidelity - 1.0, 3));
      newFactor = 2.0 * st.currentFactor;
    }
    newLambda = std::max(p->lambdaLowerBound, newLambda);
    h->cur ^= 1;
    h->linearized = false;
    st.error = newError;
    st.lambda = newLambda;
    st.currentFactor = newFactor;
    st.iterations += 1;
    st.totalNumberInnerIterations += 1;
    *done = true;
  } else if (!stopSearchingLambda) {
    // increaseLambda (:70-76)
    st.lambda *= st.currentFactor;
    st.totalNumberInnerIterations += 1;
    if (!p->useFixedLambdaFactor) st.currentFactor *= 2.0;
    *done = (st.lambda >= p->lambdaUpperBound);
  } else {
    *done = true;
  }
  return LMGPU_OK;
}

int lm_iterate(lmgpu_handle* h, const lmgpu_lm_params* p) {
  std::memset(&h->tim, 0, sizeof(h->tim));
  (void)hipEventRecord(h->ev[0], h->stream);
  int rc = do_linearize(h);
  if (rc) return rc;
  (void)hipEventRecord(h->ev[7], h->stream);
  rc = fill_dampw(h, p->diagonalDamping, p->minDiagonal, p->maxDiagonal);
  if (rc) return rc;
  bool done = false;
  while (!done) {
    rc = try_lambda(h, p, &done);
    if (rc) return rc;
  }
  HIPCHECK(hipStreamSynchronize(h->stream));
  float ms = 0;
  if (hipEventElapsedTime(&ms, h->ev[0], h->ev[7]) == hipSuccess) h->tim.linearize_ms = ms;
  h->tim.total_ms = h->tim.linearize_ms + h->tim.eliminate_ms + h->tim.backsub_ms + h->tim.linear_error_ms + h->tim.retract_error_ms;
  return LMGPU_OK;
}

// GaussNewtonOptimizer::iterate (gtsam/nonlinear/GaussNewtonOptimizer.cpp:44-66): linearize, solve the UNDAMPED system
// (lambda = 0: nothing is added to the frontal diagonals), retract, error of the new values, iterations + 1.  An
// indeterminate system is returned as LMGPU_INDETERMINATE (the reference lets IndeterminantLinearSystemException escape).
int gn_iterate(lmgpu_handle* h) {
  std::memset(&h->tim, 0, sizeof(h->tim));
  (void)hipEventRecord(h->ev[0], h->stream);
  int rc = do_linearize(h);
  if (rc) return rc;
  (void)hipEventRecord(h->ev[7], h->stream);
  rc = fill_dampw(h, 0, 0.0, 0.0);
  if (rc) return rc;
  rc = do_solve(h, 0.0);
  if (rc) return rc;
  (void)hipEventRecord(h->ev[5], h->stream);
  const int kt = h->kt.begin(LMGPU_KT_RETRACT_ERROR, h->stream);
  rc = do_retract(h, h->cur, h->cur ^ 1);
  if (rc) return rc;
  launch_factors<false>(h, h->cur ^ 1);
  reduce_to(h, h->ebuf0, h->n_counted, h->dscal);
  h->kt.end(kt, h->stream);
  rc = allreduce_scalars(h, 0, 1);
  if (rc) return rc;
  (void)hipEventRecord(h->ev[6], h->stream);
  HIPCHECK(hipMemcpyAsync(h->h_scal, h->dscal, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  h->kt.resolve();
  accumulate_times(h, true);
  h->tim.inner_iterations += 1;
  h->cur ^= 1;
  h->linearized = false;
  h->lm.error = h->h_scal[0];
  h->lm.iterations += 1;
  float ms = 0;
  if (hipEventElapsedTime(&ms, h->ev[0], h->ev[7]) == hipSuccess) h->tim.linearize_ms = ms;
  h->tim.total_ms = h->tim.linearize_ms + h->tim.eliminate_ms + h->tim.backsub_ms + h->tim.linear_error_ms + h->tim.retract_error_ms;
  return LMGPU_OK;
}

// ---- Dogleg (gtsam/nonlinear/DoglegOptimizer.cpp:84-126; DoglegOptimizerImpl.h:139-254 with ONE_STEP_PER_ITERATION;
//      DoglegOptimizerImpl.cpp:26-91).  The Bayes tree stays on the device ([R S d] of every front); the three vectors of the
//      dogleg construction (steepest-descent point, Newton point, dogleg point) are blended on the host.
// sum over the cliques of ||[R S] x - alpha d||^2
int bt_forward(lmgpu_handle* h, const double* x, double alpha, double* out) {
  hipStream_t s = h->stream;
  int len = h->n_lds_fronts;
  for (const LevelWork& L : h->levels)
    for (int fi : L.hbm) len += h->h_fronts[fi].nf;
  if (len > h->bt_ebuf_len) {
    if (h->bt_ebuf) (void)hipFree(h->bt_ebuf);
    HIPCHECK(hipMalloc((void**)&h->bt_ebuf, (size_t)len * sizeof(double)));
    h->bt_ebuf_len = len;
  }
  if (h->n_lds_fronts > 0)
    hipLaunchKernelGGL(bt_lds_forward_kernel, dim3((h->n_lds_fronts + 3) / 4), dim3(256), 0, s, (const int32_t*)h->d_lists, h->n_lds_fronts,
                       (const FrontDesc*)h->d_fronts, (const int32_t*)h->d_fxoff, (const int32_t*)h->d_sxoff, (const double*)h->pool, x, alpha,
                       h->bt_ebuf);
  int o = h->n_lds_fronts;
  for (const LevelWork& L : h->levels)
    for (int fi : L.hbm) {
      const FrontDesc& F = h->h_fronts[fi];
      hipLaunchKernelGGL(bt_hbm_forward_kernel, dim3(F.nf), dim3(64), 0, s, F, h->f_off[fi], h->f_ld[fi], (const int32_t*)h->d_fxoff,
                         (const int32_t*)h->d_sxoff, (const double*)h->pool, x, alpha, h->bt_ebuf + o);
      o += F.nf;
    }
  reduce_to(h, h->bt_ebuf, len, h->dscal + 5);
  HIPCHECK(hipMemcpyAsync(h->h_scal + 5, h->dscal + 5, sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));
  *out = h->h_scal[5];
  return LMGPU_OK;
}

// the slots of the gradient's terms and, per scalar, the list of slots that belong to it (front order: LDS-class fronts in the order of
// d_lists, then the HBM fronts level by level, chunk by chunk): built once per structure, on the first Dogleg iteration
int bt_build_gather(lmgpu_handle* h) {
  if (h->bt_gather_built) return LMGPU_OK;
  const Plan& P = h->plan;
  std::vector<int32_t> lists((size_t)h->n_lds_fronts);
  HIPCHECK(hipMemcpy(lists.data(), h->d_lists, lists.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  std::vector<int64_t> lds_off(lists.size());
  std::vector<std::vector<int32_t>> slots((size_t)h->ntot);
  int64_t off = 0;
  auto scalar_of = [&](const Front& fr, int j) {  // column j of the front -> index in delta
    size_t k = 0;
    while (k + 1 < fr.vars.size() && fr.col_off[k + 1] <= j) k++;
    return P.xoff[fr.vars[k]] + (j - fr.col_off[k]);
  };
  for (size_t li = 0; li < lists.size(); li++) {
    const Front& fr = P.fronts[lists[li]];
    lds_off[li] = off;
    for (int j = 0; j < fr.n - 1; j++) slots[(size_t)scalar_of(fr, j)].push_back((int32_t)(off + j));
    off += fr.n - 1;
  }
  h->bt_hbm_part_off.assign(P.fronts.size(), -1);
  for (const LevelWork& L : h->levels)
    for (int fi : L.hbm) {
      const Front& fr = P.fronts[fi];
      h->bt_hbm_part_off[fi] = off;
      const int nchunk = (fr.nf + 63) / 64;
      for (int c = 0; c < nchunk; c++)
        for (int j = 64 * c; j < fr.n - 1; j++) slots[(size_t)scalar_of(fr, j)].push_back((int32_t)(off + (int64_t)c * (fr.n - 1) + j));
      off += (int64_t)nchunk * (fr.n - 1);
    }
  if (off >= (int64_t)1 << 31) {
    h->err = "Dogleg: the gradient's slot table exceeds 2^31 entries";
    return LMGPU_INVALID;
  }
  std::vector<int32_t> ptr((size_t)h->ntot + 1, 0), idx;
  for (int x = 0; x < h->ntot; x++) {
    ptr[(size_t)x] = (int32_t)idx.size();
    idx.insert(idx.end(), slots[(size_t)x].begin(), slots[(size_t)x].end());
  }
  ptr[(size_t)h->ntot] = (int32_t)idx.size();
  HIPCHECK(hipMalloc((void**)&h->bt_part, std::max<int64_t>(1, off) * sizeof(double)));
  HIPCHECK(hipMalloc((void**)&h->d_bt_part_off, std::max<size_t>(1, lds_off.size()) * sizeof(int64_t)));
  HIPCHECK(hipMalloc((void**)&h->d_bt_ptr, ptr.size() * sizeof(int32_t)));
  HIPCHECK(hipMalloc((void**)&h->d_bt_idx, std::max<size_t>(1, idx.size()) * sizeof(int32_t)));
  if (!lds_off.empty()) HIPCHECK(hipMemcpy(h->d_bt_part_off, lds_off.data(), lds_off.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(h->d_bt_ptr, ptr.data(), ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  if (!idx.empty()) HIPCHECK(hipMemcpy(h->d_bt_idx, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  h->bt_gather_built = true;
  return LMGPU_OK;
}

// g = - sum over the cliques of [R S]^T d   (into h->bt_vec), every entry a sum in a fixed order
int bt_gradient(lmgpu_handle* h) {
  hipStream_t s = h->stream;
  const int rcb = bt_build_gather(h);
  if (rcb) return rcb;
  if (h->n_lds_fronts > 0)
    hipLaunchKernelGGL(bt_lds_transpose_kernel, dim3((h->n_lds_fronts + 3) / 4), dim3(256), 0, s, (const int32_t*)h->d_lists, h->n_lds_fronts,
                       (const FrontDesc*)h->d_fronts, (const int32_t*)h->d_fxoff, (const int32_t*)h->d_sxoff, (const double*)h->pool,
                       (const int64_t*)h->d_bt_part_off, h->bt_part);
  for (const LevelWork& L : h->levels)
    for (int fi : L.hbm) {
      const FrontDesc& F = h->h_fronts[fi];
      hipLaunchKernelGGL(bt_hbm_transpose_kernel, dim3((F.n - 1 + 255) / 256, (F.nf + 63) / 64), dim3(256), 0, s, F, h->f_off[fi], h->f_ld[fi],
                         (const double*)h->pool, h->bt_part + h->bt_hbm_part_off[fi]);
    }
  hipLaunchKernelGGL(bt_gather_kernel, dim3((h->ntot + 255) / 256), dim3(256), 0, s, (const int32_t*)h->d_bt_ptr, (const int32_t*)h->d_bt_idx,
                     (const double*)h->bt_part, h->ntot, h->bt_vec);
  HIPCHECK(hipGetLastError());
  return LMGPU_OK;
}

int dl_iterate(lmgpu_handle* h) {
  if (h->cfg.world_size > 1) {
    h->err = "Dogleg is single-rank in this round";
    return LMGPU_INVALID;
  }
  hipStream_t s = h->stream;
  std::memset(&h->tim, 0, sizeof(h->tim));
  if (!h->bt_vec) HIPCHECK(hipMalloc((void**)&h->bt_vec, std::max(1, h->ntot) * sizeof(double)));
  int rc = do_linearize(h);
  if (rc) return rc;
  rc = fill_dampw(h, 0, 0.0, 0.0);
  if (rc) return rc;
  rc = do_solve(h, 0.0);  // undamped Bayes tree; h->delta = Newton point dx_n
  if (rc) return rc;
  const int n = h->ntot;
  std::vector<double> dx_n(n), dx_u(n), dx_d(n);
  HIPCHECK(hipMemcpyAsync(dx_n.data(), h->delta, n * sizeof(double), hipMemcpyDeviceToHost, s));
  rc = bt_gradient(h);
  if (rc) return rc;
  HIPCHECK(hipMemcpyAsync(dx_u.data(), h->bt_vec, n * sizeof(double), hipMemcpyDeviceToHost, s));
  double RgSq = 0, M_error = 0;
  rc = bt_forward(h, h->bt_vec, 0.0, &RgSq);  // synchronises: dx_n / gradient are on the host now
  if (rc) return rc;
  double gg = 0;
  for (int i = 0; i < n; i++) gg += dx_u[i] * dx_u[i];
  const double step = -gg / RgSq;  // GaussianFactorGraph::optimizeGradientSearch, GaussianFactorGraph.cpp:381-406
  for (int i = 0; i < n; i++) dx_u[i] *= step;
  HIPCHECK(hipMemsetAsync(h->bt_vec, 0, n * sizeof(double), s));
  rc = bt_forward(h, h->bt_vec, 1.0, &M_error);
  if (rc) return rc;
  M_error *= 0.5;
  double uu = 0, nn = 0, un = 0;
  for (int i = 0; i < n; i++) {
    uu += dx_u[i] * dx_u[i];
    nn += dx_n[i] * dx_n[i];
    un += dx_u[i] * dx_n[i];
  }
  double delta = h->lm.lambda;  // trust radius
  const double f_error = h->lm.error;
  double new_f = f_error;
  bool stay = true, moved = true;
  while (stay) {
    // ComputeDoglegPoint / ComputeBlend
    const double deltaSq = delta * delta;
    if (deltaSq < uu) {
      const double f = std::sqrt(deltaSq / uu);
      for (int i = 0; i < n; i++) dx_d[i] = f * dx_u[i];
    } else if (deltaSq < nn) {
      const double a = uu - 2. * un + nn, b = 2. * (un - uu), c = uu - delta * delta;
      const double sq = std::sqrt(b * b - 4 * a * c);
      const double tau1 = (-b + sq) / (2. * a), tau2 = (-b - sq) / (2. * a);
      const double eps = std::numeric_limits<double>::epsilon();
      const double tau = (-eps <= tau1 && tau1 <= 1.0 + eps) ? tau1 : tau2;
      for (int i = 0; i < n; i++) dx_d[i] = (1. - tau) * dx_u[i] + tau * dx_n[i];
    } else {
      dx_d = dx_n;
    }
    HIPCHECK(hipMemcpyAsync(h->delta, dx_d.data(), n * sizeof(double), hipMemcpyHostToDevice, s));
    rc = enqueue_retract_and_error(h);  // values[cur ^ 1] = retract(values[cur], dx_d), f(x_d) -> h_scal[0]
    if (rc) return rc;
    double new_M = 0;
    rc = bt_forward(h, h->delta, 1.0, &new_M);  // synchronises
    if (rc) return rc;
    h->kt.resolve();
    new_M *= 0.5;
    new_f = h->h_scal[0];
    h->tim.inner_iterations += 1;
    const double rho = (std::abs(f_error - new_f) < 1e-15 || std::abs(M_error - new_M) < 1e-15) ? 0.5 : (f_error - new_f) / (M_error - new_M);
    if (rho >= 0.75) {
      double nd = 0;
      for (int i = 0; i < n; i++) nd += dx_d[i] * dx_d[i];
      delta = std::max(delta, 3.0 * std::sqrt(nd));
      stay = false;
    } else if (rho >= 0.25) {
      stay = false;
    } else if (rho >= 0.0) {
      if (delta > 1e-5) delta *= 0.5;
      stay = false;  // ONE_STEP_PER_ITERATION
    } else {  // f increased (NaN lands here too): halve the radius until it does not
      if (delta > 1e-5) {
        delta *= 0.5;
        stay = true;
      } else {
        moved = false;  // dx_d = 0: keep the values, keep the error
        new_f = f_error;
        stay = false;
      }
    }
  }
  if (moved) {
    h->cur ^= 1;
  } else {
    HIPCHECK(hipMemsetAsync(h->delta, 0, n * sizeof(double), s));
  }
  h->linearized = false;
  h->lm.error = new_f;
  h->lm.lambda = delta;
  h->lm.iterations += 1;
  h->lm.totalNumberInnerIterations += h->tim.inner_iterations;
  HIPCHECK(hipStreamSynchronize(s));
  return LMGPU_OK;
}

// checkConvergence gtsam/nonlinear/NonlinearOptimizer.cpp:182-231
bool check_convergence(double relTol, double absTol, double errTol, double currentError, double newError) {
  if (newError <= errTol) return true;
  const double absoluteDecrease = currentError - newError;
  const double relativeDecrease = absoluteDecrease / currentError;
  return (relTol && (relativeDecrease <= relTol)) || (absoluteDecrease <= absTol);
}

// Deal the subtrees below the replicated top fronts to the ranks.  A front is replicated iff it is an HBM front
// whose ancestors are all replicated (for BAL: the camera root); with no such front every rank does everything.
// Which fronts are REPLICATED (eliminated by every rank from the all-reduced sum of the ranks' partial assemblies) and which rank
// OWNS each of the others.  A subtree goes to one rank as a whole -- its fronts of either class: with a nested-dissection ordering
// (Ordering::Metis, gtsam/inference/Ordering.cpp:211-256) of a graph whose camera blocks are sparse these are the camera
// subtrees below the top separators, dense HBM fronts included -- unless it is too heavy for one rank (more than 1 / world_size
// of the whole elimination: nf n^2 per front as the measure); then its root front is replicated and its children are looked at
// in turn.  The synthetic C4 graph (every camera pair co-visible) has ONE dense root that is all of the work above the point
// leaves: it is replicated and the leaves are dealt out, as in rounds 1-2.
void assign_owners(lmgpu_handle* h) {
  const Plan& P = h->plan;
  const int NF = (int)P.fronts.size(), W = std::max(1, h->cfg.world_size), R = h->cfg.rank;
  h->front_owner.assign(NF, -1);
  h->front_active.assign(NF, 1);
  if (W == 1) return;
  // cost of the subtree below (and including) every front (post-order: children come first)
  std::vector<double> wsub(NF, 0.0);
  double total = 0;
  for (int fi = 0; fi < NF; fi++) {
    const Front& fr = P.fronts[fi];
    wsub[fi] += (double)fr.nf * fr.n * fr.n + 40.0 * (double)fr.factors.size() + 100.0;
    if (fr.parent >= 0) wsub[fr.parent] += wsub[fi]; else total += wsub[fi];
  }
  std::vector<char> rep(NF, 0);
  bool any = false;
  for (int fi = NF - 1; fi >= 0; fi--) {  // parents before children
    const Front& fr = P.fronts[fi];
    rep[fi] = (fr.cls == 1) && (fr.parent < 0 || rep[fr.parent]) && wsub[fi] * W > total;
    any = any || rep[fi];
  }
  if (!any) return;  // nothing to shard: replicate the whole (small) tree
  // the subtrees to deal out: roots = fronts that are not replicated under a replicated parent (or tree roots)
  std::vector<int> tops;
  double top_total = 0, top_max = 0;
  for (int fi = NF - 1; fi >= 0; fi--) {
    const Front& fr = P.fronts[fi];
    if (!rep[fi] && (fr.parent < 0 || rep[fr.parent])) {
      tops.push_back(fi);
      top_total += wsub[fi];
      top_max = std::max(top_max, wsub[fi]);
    }
  }
  std::vector<int> top_owner(NF, -1);
  if (top_max * 4.0 * W > top_total) {
    // a few heavy subtrees: largest first to the rank with the least work so far (ties: the lower rank; stable order of equal weights)
    std::vector<int> order(tops);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return wsub[a] > wsub[b]; });
    std::vector<double> load(W, 0.0);
    for (int fi : order) {
      int best = 0;
      for (int r = 1; r < W; r++)
        if (load[r] < load[best]) best = r;
      top_owner[fi] = best;
      load[best] += wsub[fi];
    }
  } else {
    // many light subtrees (BAL's point leaves): contiguous chunks of about equal cost, in front order
    double cum = 0;
    for (int fi : tops) {
      top_owner[fi] = std::min(W - 1, (int)(cum / top_total * W));
      cum += wsub[fi];
    }
  }
  for (int fi = NF - 1; fi >= 0; fi--) {
    const Front& fr = P.fronts[fi];
    if (rep[fi]) continue;
    h->front_owner[fi] = (fr.parent < 0 || rep[fr.parent]) ? top_owner[fi] : h->front_owner[fr.parent];
  }
  for (int fi = 0; fi < NF; fi++) h->front_active[fi] = rep[fi] || h->front_owner[fi] == R;
}

}  // namespace

// =============================================================================================== C ABI
extern "C" {

int lmgpu_create(const lmgpu_config* cfg, lmgpu_handle** out) {
  if (!cfg || !out) return LMGPU_INVALID;
  lmgpu_handle* h = new lmgpu_handle();
  h->cfg = *cfg;
  if (h->cfg.world_size < 1) h->cfg.world_size = 1;
  h->device = cfg->device;
  h->two_launch_panel = dev_switch("LMGPU_PANEL_2L") != nullptr;
  h->no_fuse = dev_switch("LMGPU_NO_FUSE") != nullptr;
  h->no_chain = dev_switch("LMGPU_NO_CHAIN") != nullptr;
  h->no_tail = dev_switch("LMGPU_NO_TAIL") != nullptr;
  h->no_wide16 = dev_switch("LMGPU_NO_WIDE16") != nullptr;
  if (const char* e = dev_switch("LMGPU_WIDE16_MAX")) h->wide16_max = atoi(e);
  h->bsd_ticket = dev_switch("LMGPU_BSD_TICKET") != nullptr;
  h->no_gather_write = dev_switch("LMGPU_NO_GATHER_WRITE") != nullptr;
  if (dev_switch("LMGPU_NO_MERGE")) h->chain_merge = false;
  if (const char* e = dev_switch("LMGPU_CHAIN_SPLIT")) h->chain_split_pct = std::max(0, std::min(100, atoi(e)));
  if (const char* e = dev_switch("LMGPU_CHAIN_FAR")) h->chain_far_pct = std::max(10, std::min(100, atoi(e)));
  *out = h;
  if (h->device >= 0) {
    HIPCHECK(hipSetDevice(h->device));
    {
      hipDeviceProp_t prop;
      HIPCHECK(hipGetDeviceProperties(&prop, h->device));
      h->num_cus = prop.multiProcessorCount;
    }
    HIPCHECK(hipStreamCreate(&h->stream));
    // the chunk all-reduces gate the panels of the factorisation that runs beside them: highest dispatch priority
    {
      int prio_lo = 0, prio_hi = 0;
      HIPCHECK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
      HIPCHECK(hipStreamCreateWithPriority(&h->comm_stream, hipStreamNonBlocking, prio_hi));
    }
    HIPCHECK(hipEventCreateWithFlags(&h->asm_ev, hipEventDisableTiming));
    for (int i = 0; i < 8; i++) HIPCHECK(hipEventCreate(&h->ev[i]));
    HIPCHECK(hipHostMalloc((void**)&h->h_scal, 8 * sizeof(double), hipHostMallocDefault));
    HIPCHECK(hipHostMalloc((void**)&h->h_status, 2 * sizeof(int), hipHostMallocDefault));
    HIPCHECK(hipMalloc((void**)&h->d_lambda, sizeof(double)));
    HIPCHECK(hipMemset(h->d_lambda, 0, sizeof(double)));
    HIPCHECK(hipFuncSetAttribute((const void*)lds_front_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitN * kLdsLimitN * 8 + kLdsFrontExtra + 64));
    HIPCHECK(hipFuncSetAttribute((const void*)lds_front_kernel<false, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitN * kLdsLimitN * 8 + kLdsFrontExtra + 64));
    HIPCHECK(hipFuncSetAttribute((const void*)lds_front_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitN * kLdsLimitN * 8 + kLdsFrontExtra + 64));
    HIPCHECK(hipFuncSetAttribute((const void*)lds_front_merged_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitN * kLdsLimitN * 8 + kLdsFrontExtra + 64));
    HIPCHECK(hipFuncSetAttribute((const void*)lds_front_merged_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitN * kLdsLimitN * 8 + kLdsFrontExtra + 64));
    HIPCHECK(hipFuncSetAttribute((const void*)level_fused_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitN * kLdsLimitN * 8 + kLdsFrontExtra + 64));
    HIPCHECK(hipFuncSetAttribute((const void*)level_fused_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitN * kLdsLimitN * 8 + kLdsFrontExtra + 64));
    HIPCHECK(hipFuncSetAttribute((const void*)syrk_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kSyrkLds));
    HIPCHECK(hipFuncSetAttribute((const void*)lds_backsub_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (kLdsLimitN * kLdsLimitN + LDSB_TAIL) * 8));
    HIPCHECK(hipFuncSetAttribute((const void*)lds_backsub_merged_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (kLdsLimitN * kLdsLimitN + LDSB_TAIL) * 8));
    HIPCHECK(hipFuncSetAttribute((const void*)diag_potrf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DIAG_LDS_BYTES));
    HIPCHECK(hipFuncSetAttribute((const void*)panel_dataflow_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PDF_LDS_BYTES));
    HIPCHECK(hipFuncSetAttribute((const void*)step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, STEP_LDS_BYTES));
    HIPCHECK(hipFuncSetAttribute((const void*)chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, STEP_LDS_BYTES));
    HIPCHECK(hipFuncSetAttribute((const void*)front_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS_BYTES));
    HIPCHECK(hipFuncSetAttribute((const void*)med_diag_potrf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DIAG_LDS_BYTES));
  }
  return LMGPU_OK;
}

int lmgpu_destroy(lmgpu_handle* h) {
  if (!h) return LMGPU_INVALID;
  if (h->device >= 0) {
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->comm_data) ncclCommDestroy(h->comm_data);
    if (h->comm) ncclCommDestroy(h->comm);
    auto fr = [](void* p) {
      if (p) (void)hipFree(p);
    };
    fr(h->pool);
    for (int w = 0; w < 2; w++)
      for (int t = 0; t < kNumVarTypes; t++) fr(h->vals[w][t]);
    for (int t = 0; t < kNumVarTypes; t++) fr(h->type_xoff[t]);
    for (int t = 0; t < kNumVarTypes; t++) fr(h->saved[t]);
    fr(h->gex); fr(h->delta); fr(h->dampw); fr(h->hdiag); fr(h->ebuf0); fr(h->ebuf1); fr(h->partial); fr(h->dscal); fr(h->ywork); fr(h->d_status);
    fr(h->d_fd); fr(h->d_fronts); fr(h->d_ffac); fr(h->d_childs); fr(h->d_cmap); fr(h->d_fxoff); fr(h->d_sxoff); fr(h->d_lists); fr(h->d_hbm_small); fr(h->d_med_list); fr(h->d_med_fronts); fr(h->inv16_med); fr(h->bt_ebuf); fr(h->bt_vec); fr(h->bt_part); fr(h->d_bt_part_off); fr(h->d_bt_ptr); fr(h->d_bt_idx); fr(h->d_f_ld); fr(h->d_f_off);
    fr(h->d_scalar_var); fr(h->d_scalar_col); fr(h->d_vi_ptr); fr(h->d_vi_fac); fr(h->d_vi_pos);
    for (Bucket& b : h->buckets) {
      fr(b.d_vidx); fr(b.d_meas); fr(b.d_noise); fr(b.d_epos);
    }
    if (h->h_scal) (void)hipHostFree(h->h_scal);
    if (h->h_status) (void)hipHostFree(h->h_status);
    for (int i = 0; i < 8; i++)
      if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    for (hipEvent_t e : h->kt.pool) (void)hipEventDestroy(e);
    fr(h->bs_inv); fr(h->bs_x); fr(h->bs_flags); fr(h->inv16); fr(h->d_pflags); fr(h->d_bs_parent); fr(h->d_bs_pos); fr(h->d_bs_done); fr(h->d_fill_upper); fr(h->d_level_tasks); fr(h->d_level_sync); fr(h->d_bsd_table); fr(h->d_bsd_run_table); fr(h->d_fill_elim); fr(h->d_fill_backsub); fr(h->d_zero_ranges); fr(h->d_bsd_x); fr(h->d_bsd_ticket); fr(h->d_leafpack); fr(h->d_gzero);
    for (auto& kv : h->chain_plans)
      for (auto& cp : kv.second) fr(cp.d_tasks);
    fr(h->d_gpblk); fr(h->d_gpent); fr(h->d_gvblk); fr(h->d_gvent); fr(h->d_gcorner); fr(h->d_row_begin); fr(h->d_rowptr); fr(h->d_rowsrc);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
    for (int g = 0; g < 2; g++)
      if (h->solve_graph[g]) (void)hipGraphExecDestroy(h->solve_graph[g]);
    if (h->d_lambda) (void)hipFree(h->d_lambda);
    if (h->asm_ev) (void)hipEventDestroy(h->asm_ev);
    for (hipEvent_t e : h->chunk_ev) (void)hipEventDestroy(e);
  }
  delete h;
  return LMGPU_OK;
}

const char* lmgpu_last_error(const lmgpu_handle* h) { return h ? h->err.c_str() : "null handle"; }
int lmgpu_last_failed_slot(const lmgpu_handle* h) { return h ? h->failed_slot : -1; }

int lmgpu_set_variables(lmgpu_handle* h, int32_t n_vars, const uint64_t* keys, const int32_t* types) {
  if (!h) return LMGPU_INVALID;
  if (n_vars <= 0 || !keys || !types || h->finalized) {
    h->err = "lmgpu_set_variables: refused (n_vars <= 0 || !keys || !types || h->finalized)";
    return LMGPU_INVALID;
  }
  h->plan.n_vars = n_vars;
  h->plan.keys.assign(keys, keys + n_vars);
  h->plan.types.assign(types, types + n_vars);
  for (int i = 0; i < n_vars; i++)
    if (types[i] < 0 || types[i] >= LMGPU_NUM_VAR_TYPES) {
      h->err = "bad variable type";
      return LMGPU_INVALID;
    }
  return LMGPU_OK;
}

int lmgpu_add_factor_bucket(lmgpu_handle* h, int32_t type, int32_t n, const int32_t* graph_index, const int32_t* var_slots, const double* meas,
                            int32_t noise_kind, const double* noise) {
  return lmgpu_add_factor_bucket_robust(h, type, n, graph_index, var_slots, meas, noise_kind, noise, LMGPU_ROBUST_NONE, 0.0);
}

int lmgpu_add_factor_bucket_robust(lmgpu_handle* h, int32_t type, int32_t n, const int32_t* graph_index, const int32_t* var_slots,
                                   const double* meas, int32_t noise_kind, const double* noise, int32_t robust_kind, double robust_k) {
  if (!h) return LMGPU_INVALID;
  if (h->finalized || type < 0 || type >= LMGPU_NUM_FACTOR_TYPES || n < 0 || h->plan.n_vars == 0) {
    h->err = "lmgpu_add_factor_bucket_robust: refused (h->finalized || type < 0 || type >= LMGPU_NUM_FACTOR_TYPES || n < 0 || h->plan.n_vars == 0)";
    return LMGPU_INVALID;
  }
  if (robust_kind < LMGPU_ROBUST_NONE || robust_kind > LMGPU_ROBUST_L2_WITH_DEAD_ZONE || (robust_kind != LMGPU_ROBUST_NONE && !(robust_k > 0.0)))
    return LMGPU_INVALID;
  if (noise_kind != LMGPU_N_UNIT && noise_kind != LMGPU_N_DIAG && noise_kind != LMGPU_N_GAUSS) return LMGPU_INVALID;
  if (n == 0) return LMGPU_OK;
  if (!graph_index || !var_slots || !meas || (noise_kind != LMGPU_N_UNIT && !noise)) return LMGPU_INVALID;
  const int ar = kFactorArity[type], rows = kFactorRows[type], ml = kFactorMeas[type];
  for (int i = 0; i < n; i++)
    for (int k = 0; k < ar; k++) {
      const int s = var_slots[i * ar + k];
      if (s < 0 || s >= h->plan.n_vars) {
        h->err = "factor references unknown slot";
        return LMGPU_INVALID;
      }
      const int want = factor_var_type(type, k);
      if (h->plan.types[s] != want) {
        h->err = "factor/variable type mismatch";
        return LMGPU_INVALID;
      }
    }
  Bucket b;
  b.type = type;
  b.n = n;
  b.noise_kind = noise_kind;
  b.robust = robust_kind;
  b.robust_k = robust_k;
  b.graph_index.assign(graph_index, graph_index + n);
  b.slots.assign(var_slots, var_slots + (size_t)n * ar);
  b.meas.assign(meas, meas + (size_t)n * ml);
  const int nl = noise_kind == LMGPU_N_DIAG ? rows : (noise_kind == LMGPU_N_GAUSS ? rows * rows : 0);
  if (nl) b.noise.assign(noise, noise + (size_t)n * nl);
  int cols = 1;
  for (int k = 0; k < ar; k++) cols += kVarDim[factor_var_type(type, k)];
  b.rows = rows;
  b.cols = cols;
  const int bi = (int)h->buckets.size();
  for (int i = 0; i < n; i++) {
    FactorRef f;
    f.bucket = bi;
    f.idx = i;
    for (int k = 0; k < kMaxArity; k++) f.slots[k] = k < ar ? var_slots[i * ar + k] : -1;
    f.graph_index = graph_index[i];
    h->plan.factors.push_back(f);
  }
  h->buckets.push_back(std::move(b));
  return LMGPU_OK;
}

int lmgpu_finalize_structure(lmgpu_handle* h) {
  if (!h) return LMGPU_INVALID;
  if (h->finalized) {
    h->err = "lmgpu_finalize_structure: refused (h->finalized)";
    return LMGPU_INVALID;
  }
  std::string e = h->plan.build(kLdsLimitN);
  if (!e.empty()) {
    h->err = e;
    return LMGPU_INVALID;
  }
  Plan& P = h->plan;
  const int NFAC = (int)P.factors.size();
  h->ntot = P.xoff[P.n_vars];
  h->nstore = P.voff[P.n_vars];
  for (int i = 1; i < NFAC; i++)
    if (P.factors[i].graph_index == P.factors[i - 1].graph_index) {
      h->err = "duplicate graph_index";
      return LMGPU_INVALID;
    }
  h->graph_index_sorted.resize(NFAC);
  for (int i = 0; i < NFAC; i++) h->graph_index_sorted[i] = P.factors[i].graph_index;

  // ---- ownership: which fronts / factors live on this rank
  assign_owners(h);
  const int NF = (int)P.fronts.size(), R = h->cfg.rank;
  std::vector<char> fac_active(NFAC, 0), fac_counted(NFAC, 0);
  for (int fi = 0; fi < NF; fi++) {
    if (!h->front_active[fi]) continue;
    const bool counted = (h->front_owner[fi] == R) || (h->front_owner[fi] < 0 && R == 0);
    for (int32_t f : P.fronts[fi].factors) {
      fac_active[f] = 1;
      fac_counted[f] = counted;
    }
  }
  h->fac_local.assign(NFAC, -1);
  int nloc = 0;
  for (int i = 0; i < NFAC; i++)
    if (fac_active[i] && fac_counted[i]) h->fac_local[i] = nloc++;
  h->n_counted = nloc;
  for (int i = 0; i < NFAC; i++)
    if (fac_active[i] && !fac_counted[i]) h->fac_local[i] = nloc++;
  h->nfac = nloc;
  for (Bucket& b : h->buckets) {
    b.loc_of.assign(b.n, -1);
    b.n_loc = 0;
  }
  for (int i = 0; i < NFAC; i++)  // factors are sorted by graph index; bucket-local order follows it
    if (fac_active[i]) {
      Bucket& b = h->buckets[P.factors[i].bucket];
      b.loc_of[P.factors[i].idx] = 0;  // mark
    }
  for (Bucket& b : h->buckets)
    for (int i = 0; i < b.n; i++)
      if (b.loc_of[i] == 0) b.loc_of[i] = b.n_loc++;

  // ---- pool layout: Jacobians | [R S d] + updates of LDS fronts | dense HBM fronts
  int64_t off = 0;
  for (Bucket& b : h->buckets) {
    b.joff = off;
    off += (int64_t)b.n_loc * b.rows * b.cols;
    off = (off + 15) & ~int64_t(15);
  }
  h->h_fronts.assign(NF, FrontDesc{});
  h->f_off.assign(NF, -1);
  h->f_ld.assign(NF, 0);
  h->s_off.assign(NF, -1);
  std::vector<FacDesc> fd(h->nfac);
  for (int i = 0; i < NFAC; i++) {
    if (h->fac_local[i] < 0) continue;
    const FactorRef& f = P.factors[i];
    const Bucket& b = h->buckets[f.bucket];
    FacDesc d{};
    d.joff = b.joff + (int64_t)b.loc_of[f.idx] * b.rows * b.cols;
    d.rows = (int16_t)b.rows;
    d.d0 = (int16_t)P.dims[f.slots[0]];
    d.d1 = (int16_t)(f.slots[1] >= 0 ? P.dims[f.slots[1]] : 0);
    d.x0 = P.xoff[f.slots[0]];
    d.x1 = f.slots[1] >= 0 ? P.xoff[f.slots[1]] : -1;
    d.d2 = (int16_t)(f.slots[2] >= 0 ? P.dims[f.slots[2]] : 0);
    d.x2 = f.slots[2] >= 0 ? P.xoff[f.slots[2]] : -1;
    fd[h->fac_local[i]] = d;
  }
  struct GPairTmp {
    int64_t key;
    int16_t da, db;
    GPairEntry ent;
  };
  struct GVarTmp {
    int32_t pv;
    int16_t dv;
    GVarEntry ent;
  };
  int n_gleaf = 0;
  std::vector<GPairTmp> gp_tmp;
  std::vector<GZeroBlock> gzero;
  std::vector<GVarTmp> gv_tmp;
  std::vector<GPairBlock> gpblk;
  std::vector<GPairEntry> gpent;
  std::vector<GVarBlock> gvblk;
  std::vector<GVarEntry> gvent;
  h->gather.assign(NF, lmgpu_handle::GatherRange());
  std::vector<FrontFac> ffac;
  std::vector<int32_t> rowptr;
  std::vector<RowSrc> rowsrc;
  h->row_begin.assign(NF, -1);
  std::vector<ChildRef> childs;
  std::vector<int32_t> cmap, fxoff, sxoff;
  std::vector<int32_t> colof(P.n_vars, -1);
  for (int fi = 0; fi < NF; fi++) {
    const Front& fr = P.fronts[fi];
    FrontDesc& F = h->h_fronts[fi];
    F.n = fr.n;
    F.nf = fr.nf;
    F.id = fi;
    F.pad = 0;
    if (!h->front_active[fi]) continue;
    if ((h->cfg.world_size > 1 || (h->cfg.flags & LMGPU_FLAG_SPLIT_ROOT)) && h->front_owner[fi] < 0 && fr.cls == 1)
      F.pad = (R == 0) ? 1 : 3;  // replicated; own terms on rank 0 only
    if (fr.cls == 0) {
      F.ld_rsd = fr.n;
      F.rsd_off = off;
      off += (int64_t)fr.nf * fr.n;
      F.ld_u = fr.n - fr.nf;
      // an LDS front whose parent lives in HBM hands its update over without an update matrix in HBM:
      //   leaves (no children): the parent GATHERS -[S d]^T [S d] and the factor terms itself (kernels_schur.hpp), par_ld = -1
      //   others: write their update matrix like any child; the parent's rows collect it in a fixed order
      const bool direct = fr.parent >= 0 && P.fronts[fr.parent].cls == 1;
      if (direct && fr.children.empty()) {
        bool binary_only = true;  // the Schur gather reads factors of at most two variables
        for (int32_t f : fr.factors) binary_only = binary_only && P.factors[f].slots[2] < 0;
        if (!binary_only) {  // such a leaf writes its update matrix like any other child
          F.u_off = off;
          off += (int64_t)F.ld_u * F.ld_u;
        } else {
          F.par_ld = -1;
          F.u_off = off;  // gather leaves also keep [S d] transposed ((n - nf) x nf: every variable's block contiguous)
          off += (int64_t)F.ld_u * fr.nf;
        }
      } else {  // (a non-leaf LDS child of an HBM front writes its update matrix too: the parent's rows collect it in a fixed order)
        F.u_off = off;
        off += (int64_t)F.ld_u * F.ld_u;
      }
    } else {
      const int ld = (fr.n + 15) & ~15;
      off = (off + 15) & ~int64_t(15);
      h->f_off[fi] = off;
      h->f_ld[fi] = ld;
      F.ld_rsd = ld;
      F.rsd_off = off;
      F.ld_u = ld;
      F.u_off = off + (int64_t)fr.nf * ld + fr.nf;
      off += (int64_t)fr.n * ld;
      // the assembled contributions get a buffer of their own beside the working matrix when they arrive in row chunks while
      // the factorisation is already running: replicated fronts (chunks all-reduced over the ranks)
      if (F.pad & 1) {
        h->s_off[fi] = off;
        off += (int64_t)fr.n * ld;
      }
    }
    for (size_t k = 0; k < fr.vars.size(); k++) colof[fr.vars[k]] = fr.col_off[k];
    // factors
    F.fac_begin = (int)ffac.size();
    for (int32_t f : fr.factors) {
      FrontFac ff;
      ff.fac = h->fac_local[f];
      ff.c0 = colof[P.factors[f].slots[0]];
      ff.c1 = P.factors[f].slots[1] >= 0 ? colof[P.factors[f].slots[1]] : 0;
      ff.c2 = P.factors[f].slots[2] >= 0 ? colof[P.factors[f].slots[2]] : 0;
      ffac.push_back(ff);
    }
    F.fac_count = (int)ffac.size() - F.fac_begin;
    // children present on this rank: map child's separator scalars (+ rhs) to this front's columns
    F.child_begin = (int)childs.size();
    for (int32_t c : fr.children) {
      if (!h->front_active[c]) continue;
      // a replicated child of a replicated front is identical on every rank: its update matrix enters the parent's sum over the ranks once
      if ((F.pad & 1) && (h->h_fronts[c].pad & 1) && R != 0) continue;
      const Front& ch = P.fronts[c];
      FrontDesc& CF = h->h_fronts[c];
      const int map_begin = (int)cmap.size();
      for (size_t k = ch.n_frontal_vars; k < ch.vars.size(); k++)
        for (int d = 0; d < P.dims[ch.vars[k]]; d++) cmap.push_back(colof[ch.vars[k]] + d);
      cmap.push_back(fr.n - 1);
      if (CF.par_ld < 0) {  // gather-mode leaf: register its S blocks and factors with this (HBM) front
        cmap.resize(map_begin);
        lmgpu_handle::GatherRange& G = h->gather[fi];
        const int li = n_gleaf++;
        if (G.leaf_count == 0) G.leaf_begin = li;
        G.leaf_count++;
        CF.par_map = li;  // index of its corner scalar
        struct Ent {
          int pc, lc, d;
        };
        std::vector<Ent> ents;
        for (size_t k = ch.n_frontal_vars; k < ch.vars.size(); k++) ents.push_back({colof[ch.vars[k]], ch.col_off[k], P.dims[ch.vars[k]]});
        ents.push_back({fr.n - 1, ch.n - 1, 1});
        for (size_t x = 0; x < ents.size(); x++)
          for (size_t y = x; y < ents.size(); y++) {
            if (x + 1 == ents.size()) continue;  // (rhs, rhs): the corner goes through the scalar reduction
            Ent a = ents[x], b = ents[y];
            if (a.pc > b.pc) std::swap(a, b);
            gp_tmp.push_back(GPairTmp{((int64_t)a.pc << 32) | (uint32_t)b.pc, (int16_t)a.d, (int16_t)b.d,
                                      GPairEntry{CF.u_off, (int16_t)((a.lc - ch.nf) * ch.nf), (int16_t)((b.lc - ch.nf) * ch.nf), (int16_t)ch.nf, 0}});
          }
        for (int32_t f : ch.factors)
          for (int pos = 0; pos < 2; pos++) {
            const int v = P.factors[f].slots[pos];
            if (v < 0 || P.front_of_var[v] == c) continue;  // frontal in the leaf: no separator contribution
            const FacDesc& fdd = fd[h->fac_local[f]];
            gv_tmp.push_back(GVarTmp{colof[v], (int16_t)P.dims[v],
                                     GVarEntry{fdd.joff, fdd.rows, (int16_t)(pos == 0 ? 0 : fdd.d0), (int16_t)(fdd.d0 + fdd.d1), 0}});
          }
        continue;
      }
      ChildRef cr{};
      cr.u_off = CF.u_off;
      cr.ld = CF.ld_u;
      cr.m = ch.n - ch.nf;
      cr.map_begin = map_begin;
      cr.pad = c + 1;  // the child front (merged launches wait for children of their own launch)
      childs.push_back(cr);
    }
    F.child_count = (int)childs.size() - F.child_begin;
    if (fr.cls == 1 && (F.child_count > 0 || F.fac_count > 0)) {  // row -> sources, for the deterministic assembly
      std::vector<std::vector<RowSrc>> rows(fr.n);
      for (int k = 0; k < F.child_count; k++) {
        const ChildRef& c = childs[F.child_begin + k];
        for (int i = 0; i < c.m; i++) rows[cmap[c.map_begin + i]].push_back(RowSrc{F.child_begin + k, i});
      }
      for (int k = 0; k < F.fac_count; k++) {
        const FrontFac& ff = ffac[F.fac_begin + k];
        const FacDesc& d = fd[ff.fac];
        const int nc = d.d0 + d.d1 + d.d2 + 1;
        for (int p = 0; p < nc; p++) {
          const int gp = (p < d.d0) ? ff.c0 + p : (p < d.d0 + d.d1 ? ff.c1 + (p - d.d0) : (p < d.d0 + d.d1 + d.d2 ? ff.c2 + (p - d.d0 - d.d1) : fr.n - 1));
          rows[gp].push_back(RowSrc{-(F.fac_begin + k) - 1, p});
        }
      }
      h->row_begin[fi] = (int32_t)rowptr.size();
      for (int R2 = 0; R2 < fr.n; R2++) {
        rowptr.push_back((int32_t)rowsrc.size());
        rowsrc.insert(rowsrc.end(), rows[R2].begin(), rows[R2].end());
      }
      rowptr.push_back((int32_t)rowsrc.size());
    }
    if (!gp_tmp.empty() || !gv_tmp.empty()) {
      lmgpu_handle::GatherRange& G = h->gather[fi];
      std::stable_sort(gp_tmp.begin(), gp_tmp.end(), [](const GPairTmp& a, const GPairTmp& b) { return a.key < b.key; });
      G.pblk_begin = (int)gpblk.size();
      std::vector<GPairBlock> longs;
      for (size_t i = 0; i < gp_tmp.size();) {
        size_t j = i;
        while (j < gp_tmp.size() && gp_tmp[j].key == gp_tmp[i].key) j++;
        const GPairBlock blk{(int32_t)(gp_tmp[i].key >> 32), (int32_t)(gp_tmp[i].key & 0xffffffff), gp_tmp[i].da, gp_tmp[i].db,
                             (int32_t)gpent.size(), (int32_t)(j - i)};
        if (j - i <= 32) gpblk.push_back(blk); else longs.push_back(blk);
        for (size_t e = i; e < j; e++) gpent.push_back(gp_tmp[e].ent);
        i = j;
      }
      G.pblk_short = (int)gpblk.size() - G.pblk_begin;
      G.pblk_long = (int)longs.size();
      gpblk.insert(gpblk.end(), longs.begin(), longs.end());
      std::stable_sort(gv_tmp.begin(), gv_tmp.end(), [](const GVarTmp& a, const GVarTmp& b) { return a.pv < b.pv; });
      G.vblk_begin = (int)gvblk.size();
      for (size_t i = 0; i < gv_tmp.size();) {
        size_t j = i;
        while (j < gv_tmp.size() && gv_tmp[j].pv == gv_tmp[i].pv) j++;
        gvblk.push_back(GVarBlock{gv_tmp[i].pv, gv_tmp[i].dv, 0, (int32_t)gvent.size(), (int32_t)(j - i)});
        for (size_t e = i; e < j; e++) gvent.push_back(gv_tmp[e].ent);
        i = j;
      }
      G.vblk_count = (int)gvblk.size() - G.vblk_begin;
      {  // chunk starts (lists are sorted by destination row)
        const int nchunks = (fr.n + NBO - 1) / NBO;
        auto starts = [&](std::vector<int>& out, int count, auto row_of) {
          out.assign(nchunks + 1, count);
          int c = 0;
          out[0] = 0;
          for (int i = 0; i < count; i++) {
            const int rc = row_of(i) / NBO;
            while (c < rc) out[++c] = i;
          }
          while (c < nchunks) out[++c] = count;
        };
        starts(G.cs, G.pblk_short, [&](int i) { return gpblk[G.pblk_begin + i].pa; });
        starts(G.cl, G.pblk_long, [&](int i) { return gpblk[G.pblk_begin + G.pblk_short + i].pa; });
        starts(G.cv, G.vblk_count, [&](int i) { return gvblk[G.vblk_begin + i].pv; });
      }
      if (F.child_count == 0) {  // no child adds to this front before its own level: the gather can be the first writer
        G.write_ok = true;
        G.zero_begin = (int)gzero.size();
        struct VB { int c, d; };
        std::vector<VB> vb;
        for (size_t k = 0; k < fr.vars.size(); k++) vb.push_back({fr.col_off[k], P.dims[fr.vars[k]]});
        vb.push_back({fr.n - 1, 1});
        size_t gi = 0;  // gpblk of this front is sorted by key among shorts, longs appended: collect the covered keys
        std::vector<int64_t> covered;
        covered.reserve(G.pblk_short + G.pblk_long);
        for (int q = 0; q < G.pblk_short + G.pblk_long; q++)
          covered.push_back(((int64_t)gpblk[G.pblk_begin + q].pa << 32) | (uint32_t)gpblk[G.pblk_begin + q].pb);
        std::sort(covered.begin(), covered.end());
        (void)gi;
        for (size_t x = 0; x < vb.size(); x++)
          for (size_t y = x; y < vb.size(); y++) {
            const int64_t key = ((int64_t)vb[x].c << 32) | (uint32_t)vb[y].c;
            if (!std::binary_search(covered.begin(), covered.end(), key))
              gzero.push_back(GZeroBlock{vb[x].c, vb[y].c, (int16_t)vb[x].d, (int16_t)vb[y].d});
          }
        G.zero_count = (int)gzero.size() - G.zero_begin;
      }
      std::vector<GPairTmp>().swap(gp_tmp);
      std::vector<GVarTmp>().swap(gv_tmp);
    }
    F.fx_begin = (int)fxoff.size();
    for (int k = 0; k < fr.n_frontal_vars; k++)
      for (int d = 0; d < P.dims[fr.vars[k]]; d++) fxoff.push_back(P.xoff[fr.vars[k]] + d);
    F.sx_begin = (int)sxoff.size();
    for (size_t k = fr.n_frontal_vars; k < fr.vars.size(); k++)
      for (int d = 0; d < P.dims[fr.vars[k]]; d++) sxoff.push_back(P.xoff[fr.vars[k]] + d);
  }
  h->pool_doubles = (size_t)off + 1024;  // slack: the syrk operand DMA may read up to 127 columns past a row end
  // ---- level work lists (active fronts only)
  h->levels.assign(P.n_levels, LevelWork());
  std::vector<std::vector<std::vector<int>>> byLevelBin(P.n_levels, std::vector<std::vector<int>>(kNumBins));
  for (int fi = 0; fi < NF; fi++) {
    if (!h->front_active[fi]) continue;
    const Front& fr = P.fronts[fi];
    if (fr.cls == 1) {
      h->levels[fr.level].hbm.push_back(fi);
    } else {
      int b = 0;
      while (fr.n > kBinN[b]) b++;
      if (h->h_fronts[fi].par_ld < 0) b += 6;  // gather leaf
      byLevelBin[fr.level][b].push_back(fi);
    }
  }
  // narrow levels (the upper part of a general clique tree): every launch is bound by one front's latency, so fronts of
  // different LDS sizes share ONE launch (the largest occupied bin of the group) instead of up to six
  for (int l = 0; l < P.n_levels; l++)
    for (int g0 = 0; g0 < kNumBins; g0 += 6) {
      size_t total = 0;
      int top = -1;
      for (int b = g0; b < g0 + 6; b++) {
        total += byLevelBin[l][b].size();
        if (!byLevelBin[l][b].empty()) top = b;
      }
      if (top < 0 || total > 512) continue;
      for (int b = g0; b < top; b++) {
        byLevelBin[l][top].insert(byLevelBin[l][top].end(), byLevelBin[l][b].begin(), byLevelBin[l][b].end());
        byLevelBin[l][b].clear();
      }
    }
  std::vector<int32_t> lists;
  for (int l = 0; l < P.n_levels; l++) {
    LevelWork& L = h->levels[l];
    L.list_begin = (int)lists.size();
    int c = 0;
    for (int b = 0; b < kNumBins; b++) {
      L.bin_begin[b] = c;
      L.bin_srows[b] = kBinN[b % 6];
      if (b >= 6) {
        int mx = 1;
        for (int fi : byLevelBin[l][b]) mx = std::max(mx, P.fronts[fi].nf);
        L.bin_srows[b] = mx;
      }
      int jc = 96;
      for (int fi : byLevelBin[l][b]) {
        int tot = 0;
        for (int32_t f : P.fronts[fi].factors) tot += fd[h->fac_local[f]].rows * (fd[h->fac_local[f]].d0 + fd[h->fac_local[f]].d1 + fd[h->fac_local[f]].d2 + 1);
        jc = std::max(jc, std::min(tot, LDSF_JCAP));
      }
      L.bin_jcap[b] = (jc + 7) & ~7;
      for (int fi : byLevelBin[l][b]) {
        lists.push_back(fi);
        L.lds_nf_max = std::max(L.lds_nf_max, (int)P.fronts[fi].nf);
        L.lds_rsd_max = std::max(L.lds_rsd_max, (int)P.fronts[fi].nf * ((int)P.fronts[fi].n | 1));
      }
      c += (int)byLevelBin[l][b].size();
    }
    L.bin_begin[kNumBins] = c;
    L.list_count = c;
  }
  // ---- merged elimination launches: runs of consecutive levels, each a handful of LDS fronts, with no dense front between them (only
  //      the top level of a run may hold dense fronts: they are launched after it and nothing in the run depends on them)
  {
    h->elim_segs.clear();
    h->elim_seg_of.assign(P.n_levels, -1);
    auto level_class = [&](int l, int* nmax, int* jcap, int* threads) { return level_lds_class(h, l, nmax, jcap, threads); };
    int l = 0;
    while (l < P.n_levels) {
      int nmax, jcap, threads;
      if (!level_class(l, &nmax, &jcap, &threads)) {
        l++;
        continue;
      }
      lmgpu_handle::ElimSeg E{l, l, nmax, jcap, threads};
      while (E.lvl_hi + 1 < P.n_levels && h->levels[E.lvl_hi].hbm.empty()) {
        int n2, j2, t2;
        if (!level_class(E.lvl_hi + 1, &n2, &j2, &t2) || (t2 == 1024) != (E.threads == 1024)) break;
        E.lvl_hi++;
        E.nmax = std::max(E.nmax, n2);
        E.jcap = std::max(E.jcap, j2);
        E.threads = std::max(E.threads, t2);
      }
      if (E.lvl_hi > E.lvl_lo) {
        h->elim_seg_of[E.lvl_lo] = (int)h->elim_segs.size();
        for (int q = E.lvl_lo + 1; q <= E.lvl_hi; q++) h->elim_seg_of[q] = -2;
        h->elim_segs.push_back(E);
      }
      l = E.lvl_hi + 1;
    }
  }
  h->finalized = true;
  if (h->device < 0) return LMGPU_OK;  // structure-only handle: symbolic analysis available, no compute

  h->n_hbm_fronts = 0;
  for (const LevelWork& L : h->levels) h->n_hbm_fronts += (int)L.hbm.size();
  // deep trees are launch-bound: replay the solve as a graph (LMGPU_GRAPH=0 / 1 overrides the depth rule)
  h->use_graph = (int)h->levels.size() >= 12;
  h->merge_backsub = (int)h->levels.size() >= 12;
  if (const char* e = dev_switch("LMGPU_MERGE_BACKSUB")) h->merge_backsub = atoi(e) != 0;
  h->merge_elim = (int)h->levels.size() >= 12 && h->cfg.world_size == 1;
  if (const char* e = dev_switch("LMGPU_MERGE_ELIM")) h->merge_elim = atoi(e) != 0 && h->cfg.world_size == 1;
  if (h->elim_segs.empty()) h->merge_elim = false;
  if (const char* e = getenv("LMGPU_GRAPH")) h->use_graph = atoi(e) != 0;
  // ---- device upload
  HIPCHECK(hipSetDevice(h->device));
  HIPCHECK(hipMalloc((void**)&h->pool, h->pool_doubles * sizeof(double)));
  // once: entries the kernels read but never write (lower triangles inside diagonal tiles, padding columns) must be finite
  HIPCHECK(hipMemset(h->pool, 0, h->pool_doubles * sizeof(double)));
  int rc;
  if (!dev_switch("LMGPU_NO_LEAFPACK")) {  // packed records of the LDS fronts, launch by launch
    std::vector<char> packs;
    for (LevelWork& L : h->levels)
      for (int b = 0; b < kNumBins; b++) {
        const int cnt = L.bin_begin[b + 1] - L.bin_begin[b];
        if (cnt == 0) continue;
        int maxfac = 0;
        for (int q = 0; q < cnt; q++) maxfac = std::max(maxfac, std::min((int)LDSF_MAXB, h->h_fronts[lists[L.list_begin + L.bin_begin[b] + q]].fac_count));
        const int stride = LEAFPACK_FAC + 32 * std::max(1, maxfac);
        packs.resize((packs.size() + 127) & ~size_t(127));
        L.pack_off[b] = (int64_t)packs.size();
        L.pack_stride[b] = stride;
        packs.resize(packs.size() + (size_t)cnt * stride, 0);
        for (int q = 0; q < cnt; q++) {
          const FrontDesc& F = h->h_fronts[lists[L.list_begin + L.bin_begin[b] + q]];
          char* rec = packs.data() + L.pack_off[b] + (size_t)q * stride;
          std::memcpy(rec, &F, sizeof(FrontDesc));
          int32_t* hdr = (int32_t*)(rec + LEAFPACK_HDR);
          hdr[0] = hdr[1] = hdr[2] = hdr[3] = 0;
          if (F.fac_count < 1 || F.fac_count > LDSF_MAXB) continue;
          LFac* lf = (LFac*)(rec + LEAFPACK_FAC);
          int o = 0;
          bool contig = true;
          for (int k = 0; k < F.fac_count; k++) {
            const FrontFac& ff = ffac[F.fac_begin + k];
            const FacDesc& d = fd[ff.fac];
            lf[k].joff = d.joff;
            lf[k].c0 = ff.c0;
            lf[k].c1 = ff.c1;
            lf[k].c2 = ff.c2;
            lf[k].rows = d.rows;
            lf[k].d0 = d.d0;
            lf[k].d1 = d.d1;
            lf[k].d2 = d.d2;
            lf[k].off = (int16_t)o;
            lf[k].sz = (int16_t)(d.rows * (d.d0 + d.d1 + d.d2 + 1));
            if (k > 0 && lf[k].joff != lf[0].joff + o) contig = false;
            o += lf[k].sz;
          }
          if (o > L.bin_jcap[b]) continue;  // more than one staging batch: the general path
          hdr[0] = F.fac_count;
          hdr[1] = o;
          hdr[2] = contig ? 1 : 0;
          if (F.nf <= LEAFPACK_MAXNF) {
            hdr[3] = 1;
            int32_t* xo = (int32_t*)(rec + LEAFPACK_XO);
            for (int i = 0; i < F.nf; i++) xo[i] = fxoff[F.fx_begin + i];
          }
        }
      }
    if (!packs.empty() && h->device >= 0) {
      HIPCHECK(hipSetDevice(h->device));
      HIPCHECK(hipMalloc((void**)&h->d_leafpack, packs.size()));
      HIPCHECK(hipMemcpy(h->d_leafpack, packs.data(), packs.size(), hipMemcpyHostToDevice));
    }
  }
  if ((rc = upload(h, &h->d_fd, fd))) return rc;
  if ((rc = upload(h, &h->d_fronts, h->h_fronts))) return rc;
  if ((rc = upload(h, &h->d_ffac, ffac))) return rc;
  if ((rc = upload(h, &h->d_childs, childs))) return rc;
  if ((rc = upload(h, &h->d_cmap, cmap))) return rc;
  if ((rc = upload(h, &h->d_fxoff, fxoff))) return rc;
  if ((rc = upload(h, &h->d_sxoff, sxoff))) return rc;
  if ((rc = upload(h, &h->d_lists, lists))) return rc;
  {
    std::vector<int32_t> bs_parent(NF, -1), bs_pos(NF, -1);
    for (size_t q = 0; q < lists.size(); q++) bs_pos[lists[q]] = (int32_t)q;
    for (int fi = 0; fi < NF; fi++) {
      const int par = P.fronts[fi].parent;
      if (par >= 0 && P.fronts[par].cls == 0) bs_parent[fi] = par;
    }
    if ((rc = upload(h, &h->d_bs_parent, bs_parent))) return rc;
    if ((rc = upload(h, &h->d_bs_pos, bs_pos))) return rc;
    HIPCHECK(hipMalloc((void**)&h->d_bs_done, (size_t)(NF + h->levels.size() + 1) * sizeof(unsigned int)));
  }
  h->n_lds_fronts = (int)lists.size();
  if (h->merge_elim) {
    std::vector<FillUpper> fu;
    for (const lmgpu_handle::ElimSeg& E : h->elim_segs)
      for (int q = h->levels[E.lvl_lo].list_begin; q < h->levels[E.lvl_hi].list_begin + h->levels[E.lvl_hi].list_count; q++) {
        const FrontDesc& F = h->h_fronts[lists[q]];
        fu.push_back(FillUpper{F.u_off, F.n - F.nf, F.ld_u});
      }
    h->n_fill_upper = (int)fu.size();
    if ((rc = upload(h, &h->d_fill_upper, fu))) return rc;
  }
  {
    std::vector<int32_t> small;
    for (LevelWork& L : h->levels) {
      L.small_begin = (int)small.size();
      for (int fi : L.hbm)
        if (P.fronts[fi].nf <= BSS_MAX_NF) small.push_back(fi);
      L.small_count = (int)small.size() - L.small_begin;
    }
    if ((rc = upload(h, &h->d_hbm_small, small))) return rc;
    std::vector<BsdBlock> bsd;
    int32_t xoff = 0;
    for (LevelWork& L : h->levels) {
      L.bsd_begin = (int)bsd.size();
      for (int q = 0; q < L.small_count; q++) {
        const int fi = small[L.small_begin + q], nblk = (P.fronts[fi].nf + 63) / 64;
        const FrontDesc& F = h->h_fronts[fi];
        for (int b = nblk - 1; b >= 0; b--) bsd.push_back(BsdBlock{h->f_off[fi], h->f_ld[fi], F.n, F.nf, F.fx_begin, F.sx_begin, F.id, b, xoff});
        xoff += 64 * nblk;
      }
      L.bsd_count = (int)bsd.size() - L.bsd_begin;
    }
    h->bsd_x_count = (size_t)xoff;
    if ((rc = upload(h, &h->d_bsd_table, bsd))) return rc;
    {
      h->bsd_runs.clear();
      h->bsd_run_of.assign(h->levels.size(), -1);
      std::vector<BsdBlock> runs;
      // a level can open a run when every HBM front on it is block-solved; the next level down joins when, in addition, none of its
      // fronts hangs below an LDS front of the run's levels (those are solved by the merged LDS launch BEHIND the run's launch)
      auto opens = [&](int l) {
        const LevelWork& L = h->levels[l];
        return L.bsd_count > 0 && L.small_count == (int)L.hbm.size();
      };
      auto joins = [&](int l, int hi) {
        if (!opens(l)) return false;
        for (int fi : h->levels[l].hbm) {
          const int par = P.fronts[fi].parent;
          if (par >= 0 && P.fronts[par].cls != 1 && P.fronts[par].level <= hi) return false;
        }
        return true;
      };
      int l = (int)h->levels.size() - 1;
      while (l >= 0) {
        if (!opens(l)) {
          l--;
          continue;
        }
        int lo = l;
        while (lo - 1 >= 0 && joins(lo - 1, l)) lo--;
        if (lo < l) {
          lmgpu_handle::BsdRun R{l, lo, (int)runs.size(), 0};
          for (int q = l; q >= lo; q--) runs.insert(runs.end(), bsd.begin() + h->levels[q].bsd_begin, bsd.begin() + h->levels[q].bsd_begin + h->levels[q].bsd_count);
          R.count = (int)runs.size() - R.begin;
          h->bsd_run_of[l] = (int)h->bsd_runs.size();
          for (int q = l - 1; q >= lo; q--) h->bsd_run_of[q] = -2;
          h->bsd_runs.push_back(R);
        }
        l = lo - 1;
      }
      if (!runs.empty() && (rc = upload(h, &h->d_bsd_run_table, runs))) return rc;
    }
    if (xoff > 0) {
      HIPCHECK(hipMalloc((void**)&h->d_bsd_x, (size_t)xoff * sizeof(double)));
      HIPCHECK(hipMalloc((void**)&h->d_bsd_ticket, h->levels.size() * sizeof(unsigned int)));
    }
    std::vector<int32_t> med;
    h->is_med.assign(NF, 0);
    int max_med = 0;
    for (LevelWork& L : h->levels) {
      L.med_begin = (int)med.size();
      for (int fi : L.hbm) {
        const Front& fr = P.fronts[fi];
        const FrontDesc& F = h->h_fronts[fi];
        const lmgpu_handle::GatherRange& G = h->gather[fi];
        if (fr.nf > NBO || (F.pad & 1) || G.leaf_count > 0 || G.pblk_short + G.pblk_long + G.vblk_count > 0 || dev_switch("LMGPU_NO_MED")) continue;
        med.push_back(fi);
        h->is_med[fi] = 1;
        L.med_max_fac = std::max(L.med_max_fac, F.fac_count);
        L.med_max_child = std::max(L.med_max_child, F.child_count);
        L.med_max_nf = std::max(L.med_max_nf, fr.nf);
        L.med_max_cols = std::max(L.med_max_cols, fr.n - fr.nf);
        L.med_max_n = std::max(L.med_max_n, (int)fr.n);
      }
      L.med_count = (int)med.size() - L.med_begin;
      max_med = std::max(max_med, L.med_count);
    }
    if ((rc = upload(h, &h->d_med_list, med))) return rc;
    if (max_med > 0) HIPCHECK(hipMalloc((void**)&h->inv16_med, (size_t)max_med * 16 * 256 * sizeof(double)));
    if ((rc = upload(h, &h->d_f_off, h->f_off))) return rc;
    if ((rc = upload(h, &h->d_f_ld, h->f_ld))) return rc;
  }
  if ((rc = upload(h, &h->d_row_begin, h->row_begin))) return rc;
  {
    std::vector<MedFront> mfs;
    for (const LevelWork& L : h->levels)
      for (int fi : L.hbm)
        if (h->is_med[fi]) mfs.push_back(MedFront{h->h_fronts[fi], h->f_off[fi], h->f_ld[fi], h->row_begin[fi]});
    if ((rc = upload(h, &h->d_med_fronts, mfs))) return rc;
    // ---- fused level launches: task lists [LDS fronts | assembly rows | diagonal blocks | panel strips | update quadrants] per level
    h->fuse_levels = (int)h->levels.size() >= 12 && h->cfg.world_size == 1;
    if (const char* e = dev_switch("LMGPU_FUSE_LEVELS")) h->fuse_levels = atoi(e) != 0 && h->cfg.world_size == 1;
    std::vector<LevelTask> tasks;
    int li = 0;
    for (LevelWork& L : h->levels) {
      const int l = li++;
      L.fuse_task_count = 0;
      int nmax, jcap, threads;
      if (!h->fuse_levels || L.med_count == 0 || !(L.med_max_fac > 0 || L.med_max_child > 0)) continue;
      if (h->merge_elim && h->elim_seg_of[l] != -1) continue;  // its LDS fronts belong to a merged launch
      if (!level_lds_class(h, l, &nmax, &jcap, &threads)) continue;
      L.fuse_task_begin = (int)tasks.size();
      L.fuse_nmax = nmax;
      L.fuse_jcap = jcap;
      // four waves per workgroup for every role: with sixteen (what a launch of wide LDS fronts alone takes) the kernel is held to 128 VGPRs and the
      // register Cholesky of the dense fronts' diagonal blocks spills (sphere2500: 1.93 vs 1.87 ms)
      L.fuse_threads = 256;
      if (const char* e = dev_switch("LMGPU_FUSE_THREADS")) L.fuse_threads = atoi(e) == 1024 && threads == 1024 ? 1024 : 256;
      for (int q = 0; q < L.list_count; q++) tasks.push_back(LevelTask{0, L.list_begin + q, 0, 0});
      for (int m = 0; m < L.med_count; m++) {
        const FrontDesc& F = mfs[L.med_begin + m].F;
        for (int r = 0; r < F.n; r += 4) tasks.push_back(LevelTask{1, m, r, 0});
      }
      for (int m = 0; m < L.med_count; m++) tasks.push_back(LevelTask{2, m, 0, 0});
      for (int m = 0; m < L.med_count; m++) {
        const FrontDesc& F = mfs[L.med_begin + m].F;
        for (int st = 0; st * 64 < F.n - F.nf; st++) tasks.push_back(LevelTask{3, m, st, 0});
      }
      for (int m = 0; m < L.med_count; m++) {
        const FrontDesc& F = mfs[L.med_begin + m].F;
        const int S = (F.n - F.nf + 63) / 64;
        for (int sj = 0; sj < S; sj++)
          for (int si = 0; si <= sj; si++)
            for (int qd = 0; qd < 4; qd++) tasks.push_back(LevelTask{4, m, si | (sj << 8), qd});
      }
      L.fuse_task_count = (int)tasks.size() - L.fuse_task_begin;
    }
    if (!tasks.empty()) {
      if ((rc = upload(h, &h->d_level_tasks, tasks))) return rc;
      h->n_level_sync = (int)mfs.size();
      HIPCHECK(hipMalloc((void**)&h->d_level_sync, std::max<size_t>(1, mfs.size()) * sizeof(LevelSync)));
    } else {
      h->fuse_levels = false;
    }
  }
  if ((rc = upload(h, &h->d_rowptr, rowptr))) return rc;
  if ((rc = upload(h, &h->d_rowsrc, rowsrc))) return rc;
  if ((rc = upload(h, &h->d_gpblk, gpblk))) return rc;
  if ((rc = upload(h, &h->d_gzero, gzero))) return rc;
  if ((rc = upload(h, &h->d_gpent, gpent))) return rc;
  if ((rc = upload(h, &h->d_gvblk, gvblk))) return rc;
  if ((rc = upload(h, &h->d_gvent, gvent))) return rc;
  HIPCHECK(hipMalloc((void**)&h->d_gcorner, std::max<size_t>(1, (size_t)n_gleaf) * sizeof(double)));
  // per-bucket arrays, compacted to this rank's factors
  {
    std::vector<std::vector<int32_t>> epos(h->buckets.size());
    for (size_t bi = 0; bi < h->buckets.size(); bi++) epos[bi].resize(h->buckets[bi].n_loc);
    for (int i = 0; i < NFAC; i++)
      if (h->fac_local[i] >= 0) epos[P.factors[i].bucket][h->buckets[P.factors[i].bucket].loc_of[P.factors[i].idx]] = h->fac_local[i];
    for (size_t bi = 0; bi < h->buckets.size(); bi++) {
      Bucket& b = h->buckets[bi];
      const int ar = kFactorArity[b.type], ml = kFactorMeas[b.type];
      const int nl = b.noise_kind == LMGPU_N_DIAG ? b.rows : (b.noise_kind == LMGPU_N_GAUSS ? b.rows * b.rows : 0);
      std::vector<int32_t> vidx((size_t)b.n_loc * ar);
      std::vector<double> meas((size_t)b.n_loc * ml), noise((size_t)b.n_loc * nl);
      for (int i = 0; i < b.n; i++) {
        const int l = b.loc_of[i];
        if (l < 0) continue;
        for (int k = 0; k < ar; k++) vidx[(size_t)l * ar + k] = P.tidx[b.slots[(size_t)i * ar + k]];
        std::memcpy(&meas[(size_t)l * ml], &b.meas[(size_t)i * ml], ml * sizeof(double));
        if (nl) std::memcpy(&noise[(size_t)l * nl], &b.noise[(size_t)i * nl], nl * sizeof(double));
      }
      if ((rc = upload(h, &b.d_vidx, vidx))) return rc;
      if ((rc = upload(h, &b.d_meas, meas))) return rc;
      if (nl && (rc = upload(h, &b.d_noise, noise))) return rc;
      if ((rc = upload(h, &b.d_epos, epos[bi]))) return rc;
    }
  }
  // values, per type
  for (int t = 0; t < kNumVarTypes; t++) {
    const size_t bytes = std::max<size_t>(1, (size_t)P.type_count[t] * kVarStore[t]) * sizeof(double);
    HIPCHECK(hipMalloc((void**)&h->vals[0][t], bytes));
    HIPCHECK(hipMalloc((void**)&h->vals[1][t], bytes));
    std::vector<int32_t> xo(P.type_count[t]);
    for (int s = 0; s < P.n_vars; s++)
      if (P.types[s] == t) xo[P.tidx[s]] = P.xoff[s];
    if ((rc = upload(h, &h->type_xoff[t], xo))) return rc;
  }
  // hessian-diagonal CSR over the factors COUNTED on this rank (summed over ranks by all-reduce)
  {
    std::vector<int32_t> scalar_var(h->ntot), scalar_col(h->ntot), vi_ptr(P.n_vars + 1, 0), vi_fac;
    std::vector<int8_t> vi_pos;
    for (int s = 0; s < P.n_vars; s++)
      for (int d = 0; d < P.dims[s]; d++) {
        scalar_var[P.xoff[s] + d] = s;
        scalar_col[P.xoff[s] + d] = d;
      }
    std::vector<std::vector<std::pair<int32_t, int8_t>>> vi(P.n_vars);
    for (int i = 0; i < NFAC; i++) {
      const int l = h->fac_local[i];
      if (l < 0 || l >= h->n_counted) continue;
      for (int k = 0; k < kMaxArity; k++)
        if (P.factors[i].slots[k] >= 0) vi[P.factors[i].slots[k]].push_back({l, (int8_t)k});
    }
    for (int s = 0; s < P.n_vars; s++) {
      vi_ptr[s] = (int32_t)vi_fac.size();
      for (auto& pr : vi[s]) {
        vi_fac.push_back(pr.first);
        vi_pos.push_back(pr.second);
      }
    }
    vi_ptr[P.n_vars] = (int32_t)vi_fac.size();
    if ((rc = upload(h, &h->d_scalar_var, scalar_var))) return rc;
    if ((rc = upload(h, &h->d_scalar_col, scalar_col))) return rc;
    if ((rc = upload(h, &h->d_vi_ptr, vi_ptr))) return rc;
    if ((rc = upload(h, &h->d_vi_fac, vi_fac))) return rc;
    if ((rc = upload(h, &h->d_vi_pos, vi_pos))) return rc;
  }
  HIPCHECK(hipMalloc((void**)&h->delta, h->ntot * sizeof(double)));
  HIPCHECK(hipMemset(h->delta, 0, h->ntot * sizeof(double)));
  HIPCHECK(hipMalloc((void**)&h->dampw, h->ntot * sizeof(double)));
  HIPCHECK(hipMalloc((void**)&h->hdiag, h->ntot * sizeof(double)));
  HIPCHECK(hipMalloc((void**)&h->ebuf0, std::max(1, h->nfac) * sizeof(double)));
  HIPCHECK(hipMalloc((void**)&h->ebuf1, std::max(1, h->nfac) * sizeof(double)));
  HIPCHECK(hipMalloc((void**)&h->partial, 256 * sizeof(double)));
  HIPCHECK(hipMalloc((void**)&h->dscal, 8 * sizeof(double)));
  HIPCHECK(hipMalloc((void**)&h->ywork, std::max(1, P.max_front_n) * sizeof(double)));
  HIPCHECK(hipMalloc((void**)&h->d_status, 2 * sizeof(int)));
  {
    int max_nf = 1;
    for (int fi = 0; fi < NF; fi++)
      if (h->front_active[fi] && P.fronts[fi].cls == 1) max_nf = std::max(max_nf, P.fronts[fi].nf);
    const int max_blk = (max_nf + NB - 1) / NB;
    HIPCHECK(hipMalloc((void**)&h->bs_inv, (size_t)max_blk * NB * NB * sizeof(double)));
    HIPCHECK(hipMalloc((void**)&h->inv16, (size_t)((P.max_front_n + NBO - 1) / NBO + 2) * 16 * 256 * sizeof(double)));  // 16 blocks per outer panel
    h->pflags_panels = (P.max_front_n + NBO - 1) / NBO + 1;
    HIPCHECK(hipMalloc((void**)&h->d_pflags, (size_t)h->pflags_panels * PDF_FLAG_WORDS * sizeof(unsigned int)));

    HIPCHECK(hipMalloc((void**)&h->bs_x, (size_t)max_blk * NB * sizeof(double)));
    HIPCHECK(hipMalloc((void**)&h->bs_flags, (size_t)(max_blk + 1) * sizeof(unsigned int)));
  }
  return LMGPU_OK;
}

int lmgpu_total_dim(const lmgpu_handle* h) { return (h && h->finalized) ? h->ntot : -1; }
int lmgpu_total_store(const lmgpu_handle* h) { return (h && h->finalized) ? h->nstore : -1; }

int lmgpu_set_values(lmgpu_handle* h, const double* packed) {
  if (!h) return LMGPU_INVALID;
  if (!h->finalized || !packed) {
    h->err = "lmgpu_set_values: refused (!h->finalized || !packed)";
    return LMGPU_INVALID;
  }
  int rc = need_device(h);
  if (rc) return rc;
  const Plan& P = h->plan;
  for (int t = 0; t < kNumVarTypes; t++) {
    if (P.type_count[t] == 0) continue;
    std::vector<double> buf((size_t)P.type_count[t] * kVarStore[t]);
    for (int s = 0; s < P.n_vars; s++)
      if (P.types[s] == t) std::memcpy(&buf[(size_t)P.tidx[s] * kVarStore[t]], packed + P.voff[s], kVarStore[t] * sizeof(double));
    HIPCHECK(hipMemcpy(h->vals[h->cur][t], buf.data(), buf.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  h->have_values = true;
  return LMGPU_OK;
}

int lmgpu_get_values(lmgpu_handle* h, double* packed) {
  if (!h) return LMGPU_INVALID;
  if (!h->finalized || !packed || !h->have_values) {
    h->err = "lmgpu_get_values: refused (!h->finalized || !packed || !h->have_values)";
    return LMGPU_INVALID;
  }
  int rc = need_device(h);
  if (rc) return rc;
  const Plan& P = h->plan;
  HIPCHECK(hipStreamSynchronize(h->stream));
  for (int t = 0; t < kNumVarTypes; t++) {
    if (P.type_count[t] == 0) continue;
    std::vector<double> buf((size_t)P.type_count[t] * kVarStore[t]);
    HIPCHECK(hipMemcpy(buf.data(), h->vals[h->cur][t], buf.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int s = 0; s < P.n_vars; s++)
      if (P.types[s] == t) std::memcpy(packed + P.voff[s], &buf[(size_t)P.tidx[s] * kVarStore[t]], kVarStore[t] * sizeof(double));
  }
  return LMGPU_OK;
}

// device-side snapshot of the current values (e.g. the initial estimate) and its restore: device-to-device copies
int lmgpu_save_values(lmgpu_handle* h) {
  if (!h) return LMGPU_INVALID;
  if (!h->finalized || !h->have_values) {
    h->err = "lmgpu_save_values: refused (!h->finalized || !h->have_values)";
    return LMGPU_INVALID;
  }
  int rc = need_device(h);
  if (rc) return rc;
  for (int t = 0; t < kNumVarTypes; t++) {
    const size_t bytes = std::max<size_t>(1, (size_t)h->plan.type_count[t] * kVarStore[t]) * sizeof(double);
    if (!h->saved[t]) HIPCHECK(hipMalloc((void**)&h->saved[t], bytes));
    HIPCHECK(hipMemcpyAsync(h->saved[t], h->vals[h->cur][t], bytes, hipMemcpyDeviceToDevice, h->stream));
  }
  HIPCHECK(hipStreamSynchronize(h->stream));
  return LMGPU_OK;
}

int lmgpu_restore_values(lmgpu_handle* h) {
  if (!h) return LMGPU_INVALID;
  if (!h->finalized || !h->saved[0]) {
    h->err = "lmgpu_restore_values: refused (!h->finalized || !h->saved[0])";
    return LMGPU_INVALID;
  }
  int rc = need_device(h);
  if (rc) return rc;
  for (int t = 0; t < kNumVarTypes; t++) {
    const size_t bytes = std::max<size_t>(1, (size_t)h->plan.type_count[t] * kVarStore[t]) * sizeof(double);
    HIPCHECK(hipMemcpyAsync(h->vals[h->cur][t], h->saved[t], bytes, hipMemcpyDeviceToDevice, h->stream));
  }
  return LMGPU_OK;
}

int lmgpu_error(lmgpu_handle* h, double* total) {
  if (!h) return LMGPU_INVALID;
  if (!h->finalized || !total || !h->have_values) {
    h->err = "lmgpu_error: refused (!h->finalized || !total || !h->have_values)";
    return LMGPU_INVALID;
  }
  int rc = need_device(h);
  if (rc) return rc;
  if ((rc = need_comm(h))) return rc;
  return compute_error(h, h->cur, total);
}

int lmgpu_linearize(lmgpu_handle* h) {
  if (!h) return LMGPU_INVALID;
  if (!h->finalized || !h->have_values) {
    h->err = "lmgpu_linearize: refused (!h->finalized || !h->have_values)";
    return LMGPU_INVALID;
  }
  int rc = need_device(h);
  if (rc) return rc;
  rc = do_linearize(h);
  if (rc) return rc;
  HIPCHECK(hipStreamSynchronize(h->stream));
  h->kt.resolve();
  return LMGPU_OK;
}

int lmgpu_solve(lmgpu_handle* h, double lambda, int32_t diagonal_damping, double min_diag, double max_diag, double* delta_packed,
                double* lin_err0, double* lin_err1) {
  if (!h) return LMGPU_INVALID;
  if (!h->finalized || !h->linearized) {
    h->err = "lmgpu_solve: no linearization to solve (call lmgpu_linearize after lmgpu_finalize_structure / lmgpu_set_values)";
    return LMGPU_INVALID;
  }
  int rc = need_device(h);
  if (rc) return rc;
  if ((rc = need_comm(h))) return rc;
  rc = fill_dampw(h, diagonal_damping, min_diag, max_diag);
  if (rc) return rc;
  rc = do_solve(h, lambda);
  if (rc != LMGPU_OK && rc != LMGPU_INDETERMINATE) return rc;
  if (delta_packed) HIPCHECK(hipMemcpy(delta_packed, h->delta, h->ntot * sizeof(double), hipMemcpyDeviceToHost));
  if (lin_err0) *lin_err0 = h->h_scal[1];
  if (lin_err1) *lin_err1 = h->h_scal[2];
  return rc;
}

// b := 0 in every stored [A b]: the linear system keeps its matrix A^T A and loses its gradient
__global__ __launch_bounds__(256) void zero_rhs_kernel(const FacDesc* __restrict__ fd, int nfac, double* __restrict__ pool) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= nfac) return;
  const FacDesc d = fd[f];
  double* b = pool + d.joff + (size_t)(d.d0 + d.d1) * d.rows;
  for (int r = 0; r < d.rows; r++) b[r] = 0.0;
}
__global__ void set_one_kernel(double* p) { *p = 1.0; }

int lmgpu_joint_marginal_covariance(lmgpu_handle* h, int32_t nslots, const int32_t* slots, double* cov) {
  if (!h) return LMGPU_INVALID;
  if (!cov || !slots || nslots < 1 || !h->finalized || !h->have_values) {
    h->err = "lmgpu_joint_marginal_covariance: refused (!cov || !slots || nslots < 1 || !h->finalized || !h->have_values)";
    return LMGPU_INVALID;
  }
  for (int a = 0; a < nslots; a++) {
    if (slots[a] < 0 || slots[a] >= h->plan.n_vars) return LMGPU_INVALID;
    for (int b = 0; b < a; b++)
      if (slots[b] == slots[a]) return LMGPU_INVALID;
  }
  if (h->cfg.world_size > 1) {
    h->err = "marginal covariances are computed on one GPU";
    return LMGPU_INVALID;
  }
  int rc = need_device(h);
  if (rc) return rc;
  // Marginals::Marginals (gtsam/nonlinear/Marginals.cpp:28-43): linearize at the solution, eliminate into a Bayes tree
  // (here: the same two kernels as every LM step, lambda = 0);  marginalCovariance (:124-127) = inverse of the marginal
  // information (:109-121) = the variable's diagonal block of (A^T A)^-1;  jointMarginalCovariance (:130-137) = the blocks of
  // (A^T A)^-1 for several variables.  Column k of those blocks is the variables' part of the solution of  A^T A x = e_k,
  // i.e. one elimination + back-substitution with the gradient replaced by a unit vector.
  if ((rc = do_linearize(h))) return rc;
  hipStream_t s = h->stream;
  if (h->nfac > 0) hipLaunchKernelGGL(zero_rhs_kernel, dim3((h->nfac + 255) / 256), dim3(256), 0, s, (const FacDesc*)h->d_fd, h->nfac, h->pool);
  if (!h->gex) HIPCHECK(hipMalloc((void**)&h->gex, h->ntot * sizeof(double)));
  if ((rc = fill_dampw(h, 0, 0.0, 0.0))) return rc;
  std::vector<int> boff(nslots + 1, 0);  // block offsets inside the joint matrix, in the order of `slots`
  for (int a = 0; a < nslots; a++) boff[a + 1] = boff[a] + h->plan.dims[slots[a]];
  const int D = boff[nslots];
  std::vector<double> col(h->ntot);
  h->gex_active = h->gex;
  for (int a = 0; a < nslots && rc == LMGPU_OK; a++) {
    for (int k = 0; k < h->plan.dims[slots[a]] && rc == LMGPU_OK; k++) {
      HIPCHECK(hipMemsetAsync(h->gex, 0, h->ntot * sizeof(double), s));
      hipLaunchKernelGGL(set_one_kernel, dim3(1), dim3(1), 0, s, h->gex + h->plan.xoff[slots[a]] + k);
      rc = do_solve(h, 0.0);
      if (rc != LMGPU_OK) break;
      for (int b = 0; b < nslots; b++) {
        const int db = h->plan.dims[slots[b]];
        HIPCHECK(hipMemcpy(col.data(), h->delta + h->plan.xoff[slots[b]], db * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < db; i++) cov[(size_t)(boff[b] + i) * D + boff[a] + k] = col[i];
      }
    }
  }
  h->gex_active = nullptr;
  h->linearized = false;  // the stored right-hand sides are gone: the next solve linearizes again
  h->solved = false;
  if (rc != LMGPU_OK) return rc;  // LMGPU_INDETERMINATE like Marginals' IndeterminantLinearSystemException
  for (int i = 0; i < D; i++)
    for (int j = i + 1; j < D; j++) cov[(size_t)i * D + j] = cov[(size_t)j * D + i] = 0.5 * (cov[(size_t)i * D + j] + cov[(size_t)j * D + i]);
  return LMGPU_OK;
}

int lmgpu_marginal_covariance(lmgpu_handle* h, int32_t slot, double* cov) { return lmgpu_joint_marginal_covariance(h, 1, &slot, cov); }

// The stored linearization ([A b] per factor) stays valid across lmgpu_set_values / lmgpu_retract / lmgpu_restore_values, like the
// GaussianFactorGraph the reference's linearize() returned: tryLambda solves it again with a larger lambda after a rejected step
// (LevenbergMarquardtOptimizer.cpp:302-305) whatever happened to the Values in between.
int lmgpu_retract(lmgpu_handle* h, const double* delta_packed) {
  if (!h) return LMGPU_INVALID;
  if (!h->finalized || !h->have_values) {
    h->err = "lmgpu_retract: refused (!h->finalized || !h->have_values)";
    return LMGPU_INVALID;
  }
  int rc = need_device(h);
  if (rc) return rc;
  if (delta_packed) HIPCHECK(hipMemcpy(h->delta, delta_packed, h->ntot * sizeof(double), hipMemcpyHostToDevice));
  rc = do_retract(h, h->cur, h->cur ^ 1);
  if (rc) return rc;
  HIPCHECK(hipStreamSynchronize(h->stream));
  h->cur ^= 1;
  return LMGPU_OK;
}

int lmgpu_hessian_diagonal(lmgpu_handle* h, double* diag) {
  if (!h) return LMGPU_INVALID;
  if (!h->finalized || !h->linearized || !diag) {
    h->err = "lmgpu_hessian_diagonal: refused (!h->finalized || !h->linearized || !diag)";
    return LMGPU_INVALID;
  }
  int rc = need_device(h);
  if (rc) return rc;
  if ((rc = need_comm(h))) return rc;
  if ((rc = launch_hessian_diag(h))) return rc;
  HIPCHECK(hipMemcpyAsync(diag, h->hdiag, h->ntot * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return LMGPU_OK;
}

int lmgpu_lm_init(lmgpu_handle* h, const lmgpu_lm_params* p, lmgpu_lm_state* out) {
  if (!h) return LMGPU_INVALID;
  if (!p || !h->finalized || !h->have_values) {
    h->err = "lmgpu_lm_init: refused (!p || !h->finalized || !h->have_values)";
    return LMGPU_INVALID;
  }
  int rc = need_device(h);
  if (rc) return rc;
  if ((rc = need_comm(h))) return rc;
  // LevenbergMarquardtOptimizer ctor (LevenbergMarquardtOptimizer.cpp:47-63): state(values, graph.error(values), lambdaInitial, lambdaFactor)
  double e = 0;
  rc = compute_error(h, h->cur, &e);
  if (rc) return rc;
  h->lm.error = e;
  h->lm.lambda = p->lambdaInitial;
  h->lm.currentFactor = p->lambdaFactor;
  h->lm.iterations = 0;
  h->lm.totalNumberInnerIterations = 0;
  if (out) *out = h->lm;
  return LMGPU_OK;
}

int lmgpu_iterate(lmgpu_handle* h, const lmgpu_lm_params* p, lmgpu_lm_state* inout) {
  if (!h) return LMGPU_INVALID;
  if (!p || !h->finalized || !h->have_values) {
    h->err = "lmgpu_iterate: refused (!p || !h->finalized || !h->have_values)";
    return LMGPU_INVALID;
  }
  int rc = need_device(h);
  if (rc) return rc;
  if ((rc = need_comm(h))) return rc;
  if (inout) h->lm = *inout;  // the caller owns the LevenbergMarquardtState (error, lambda, factor, counters)
  rc = lm_iterate(h, p);
  if (inout) *inout = h->lm;
  return rc;
}

int lmgpu_gn_iterate(lmgpu_handle* h, lmgpu_lm_state* inout) {
  if (!h) return LMGPU_INVALID;
  if (!h->finalized || !h->have_values) {
    h->err = "lmgpu_gn_iterate: refused (!h->finalized || !h->have_values)";
    return LMGPU_INVALID;
  }
  int rc = need_device(h);
  if (rc) return rc;
  if ((rc = need_comm(h))) return rc;
  if (inout) h->lm = *inout;
  rc = gn_iterate(h);
  if (inout) *inout = h->lm;
  return rc;
}
int lmgpu_gn_optimize(lmgpu_handle* h, const lmgpu_lm_params* p, lmgpu_lm_state* inout) {
  if (!h) return LMGPU_INVALID;
  if (!p || !h->finalized || !h->have_values) {
    h->err = "lmgpu_gn_optimize: refused (!p || !h->finalized || !h->have_values)";
    return LMGPU_INVALID;
  }
  int rc = need_device(h);
  if (rc) return rc;
  if ((rc = need_comm(h))) return rc;
  if (inout) h->lm = *inout;
  // NonlinearOptimizer::defaultOptimize (gtsam/nonlinear/NonlinearOptimizer.cpp:62-117) around GaussNewtonOptimizer::iterate
  double currentError = h->lm.error;
  if (!(currentError <= p->errorTol || h->lm.iterations >= p->maxIterations)) {
    double newError = currentError;
    do {
      currentError = newError;
      rc = gn_iterate(h);
      if (rc) break;
      newError = h->lm.error;
    } while (h->lm.iterations < p->maxIterations &&
             !check_convergence(p->relativeErrorTol, p->absoluteErrorTol, p->errorTol, currentError, newError) && std::isfinite(currentError));
  }
  if (inout) *inout = h->lm;
  return rc;
}
int lmgpu_dl_iterate(lmgpu_handle* h, lmgpu_lm_state* inout) {
  if (!h) return LMGPU_INVALID;
  if (!h->finalized || !h->have_values) {
    h->err = "lmgpu_dl_iterate: refused (!h->finalized || !h->have_values)";
    return LMGPU_INVALID;
  }
  int rc = need_device(h);
  if (rc) return rc;
  if (inout) h->lm = *inout;
  rc = dl_iterate(h);
  if (inout) *inout = h->lm;
  return rc;
}
int lmgpu_dl_optimize(lmgpu_handle* h, const lmgpu_lm_params* p, lmgpu_lm_state* inout) {
  if (!h) return LMGPU_INVALID;
  if (!p || !h->finalized || !h->have_values) {
    h->err = "lmgpu_dl_optimize: refused (!p || !h->finalized || !h->have_values)";
    return LMGPU_INVALID;
  }
  int rc = need_device(h);
  if (rc) return rc;
  if (inout) h->lm = *inout;
  double currentError = h->lm.error;
  if (!(currentError <= p->errorTol || h->lm.iterations >= p->maxIterations)) {
    double newError = currentError;
    do {
      currentError = newError;
      rc = dl_iterate(h);
      if (rc) break;
      newError = h->lm.error;
    } while (h->lm.iterations < p->maxIterations &&
             !check_convergence(p->relativeErrorTol, p->absoluteErrorTol, p->errorTol, currentError, newError) && std::isfinite(currentError));
  }
  if (inout) *inout = h->lm;
  return rc;
}
int lmgpu_optimize(lmgpu_handle* h, const lmgpu_lm_params* p, lmgpu_lm_state* inout) {
  if (!h) return LMGPU_INVALID;
  if (!p || !h->finalized || !h->have_values) {
    h->err = "lmgpu_optimize: refused (!p || !h->finalized || !h->have_values)";
    return LMGPU_INVALID;
  }
  int rc = need_device(h);
  if (rc) return rc;
  if ((rc = need_comm(h))) return rc;
  // NonlinearOptimizer::defaultOptimize (gtsam/nonlinear/NonlinearOptimizer.cpp:62-117)
  if (inout) h->lm = *inout;  // the caller owns the LevenbergMarquardtState, as in lmgpu_iterate
  double currentError = h->lm.error;
  if (currentError <= p->errorTol || h->lm.iterations >= p->maxIterations) {
    if (inout) *inout = h->lm;
    return LMGPU_OK;
  }
  double newError = currentError;
  do {
    currentError = newError;
    rc = lm_iterate(h, p);
    if (rc) {
      if (inout) *inout = h->lm;  // the state reached so far, also on an error return
      return rc;
    }
    newError = h->lm.error;
  } while (h->lm.iterations < p->maxIterations &&
           !check_convergence(p->relativeErrorTol, p->absoluteErrorTol, p->errorTol, currentError, newError) && std::isfinite(currentError));
  if (inout) *inout = h->lm;
  return LMGPU_OK;
}

int lmgpu_get_timings(const lmgpu_handle* h, lmgpu_timings* out) {
  if (!h || !out) return LMGPU_INVALID;
  *out = h->tim;
  return LMGPU_OK;
}

int lmgpu_set_kernel_timing(lmgpu_handle* h, int32_t on) {
  if (!h) return LMGPU_INVALID;
  h->kt.on = on != 0;
  h->kt.dominant_only = on == 2;
  h->kt.reset();
  return LMGPU_OK;
}

int lmgpu_get_kernel_times(const lmgpu_handle* h, double* ms, double* work, int64_t* launches) {
  if (!h) return LMGPU_INVALID;
  for (int i = 0; i < LMGPU_KT_NUM; i++) {
    if (ms) ms[i] = h->kt.ms[i];
    if (work) work[i] = h->kt.work[i];
    if (launches) launches[i] = h->kt.cnt[i];
  }
  return LMGPU_OK;
}

int lmgpu_get_jacobian(lmgpu_handle* h, int32_t graph_index, double* out, int32_t* rows, int32_t* cols) {
  if (!h) return LMGPU_INVALID;
  if (!h->finalized) {
    h->err = "lmgpu_get_jacobian: refused (!h->finalized)";
    return LMGPU_INVALID;
  }
  auto it = std::lower_bound(h->graph_index_sorted.begin(), h->graph_index_sorted.end(), graph_index);
  if (it == h->graph_index_sorted.end() || *it != graph_index) return LMGPU_INVALID;
  const FactorRef& f = h->plan.factors[it - h->graph_index_sorted.begin()];
  const Bucket& b = h->buckets[f.bucket];
  if (rows) *rows = b.rows;
  if (cols) *cols = b.cols;
  if (out) {
    int rc = need_device(h);
    if (rc) return rc;
    if (!h->have_jacobians || b.loc_of[f.idx] < 0) return LMGPU_INVALID;  // never linearized, or the factor lives on another rank
    HIPCHECK(hipMemcpy(out, h->pool + b.joff + (int64_t)b.loc_of[f.idx] * b.rows * b.cols, (size_t)b.rows * b.cols * sizeof(double),
                       hipMemcpyDeviceToHost));
  }
  return LMGPU_OK;
}

int lmgpu_get_jacobians(lmgpu_handle* h, int32_t* n_out, int32_t* graph_index, int32_t* rows, int32_t* cols, int64_t* offsets, double* out) {
  if (!h) return LMGPU_INVALID;
  if (!h->finalized) {
    h->err = "lmgpu_get_jacobians: refused (!h->finalized)";
    return LMGPU_INVALID;
  }
  const int NFAC = (int)h->plan.factors.size();
  if (n_out) *n_out = NFAC;
  int64_t off = 0;
  for (int i = 0; i < NFAC; i++) {  // plan.factors is ascending by graph index
    const FactorRef& f = h->plan.factors[i];
    const Bucket& b = h->buckets[f.bucket];
    if (graph_index) graph_index[i] = f.graph_index;
    if (rows) rows[i] = b.rows;
    if (cols) cols[i] = b.cols;
    if (offsets) offsets[i] = off;
    off += (int64_t)b.rows * b.cols;
  }
  if (offsets) offsets[NFAC] = off;
  if (!out) return LMGPU_OK;
  int rc = need_device(h);
  if (rc) return rc;
  if (!h->have_jacobians) {
    h->err = "lmgpu_get_jacobians: refused (no linearization on the device)";
    return LMGPU_INVALID;
  }
  // one copy per bucket, then the host puts every factor at its place in graph order
  std::vector<std::vector<double>> jb(h->buckets.size());
  for (size_t k = 0; k < h->buckets.size(); k++) {
    const Bucket& b = h->buckets[k];
    jb[k].resize((size_t)b.n_loc * b.rows * b.cols);
    if (b.n_loc) HIPCHECK(hipMemcpy(jb[k].data(), h->pool + b.joff, jb[k].size() * sizeof(double), hipMemcpyDeviceToHost));
  }
  off = 0;
  for (int i = 0; i < NFAC; i++) {
    const FactorRef& f = h->plan.factors[i];
    const Bucket& b = h->buckets[f.bucket];
    const size_t sz = (size_t)b.rows * b.cols;
    if (b.loc_of[f.idx] < 0) {
      h->err = "lmgpu_get_jacobians: a factor lives on another rank";
      return LMGPU_INVALID;
    }
    std::memcpy(out + off, jb[f.bucket].data() + (size_t)b.loc_of[f.idx] * sz, sz * sizeof(double));
    off += (int64_t)sz;
  }
  return LMGPU_OK;
}

int lmgpu_num_fronts(const lmgpu_handle* h) { return (h && h->finalized) ? (int)h->plan.fronts.size() : -1; }

int lmgpu_front_info(const lmgpu_handle* h, int32_t front, int32_t* info) {
  if (!h || !h->finalized || front < 0 || front >= (int)h->plan.fronts.size() || !info) return LMGPU_INVALID;
  const Front& fr = h->plan.fronts[front];
  info[0] = (int)fr.vars.size();
  info[1] = fr.n_frontal_vars;
  info[2] = fr.nf;
  info[3] = fr.n;
  info[4] = fr.parent;
  info[5] = fr.cls;
  info[6] = h->front_owner[front];
  info[7] = fr.level;
  return LMGPU_OK;
}

int lmgpu_get_front(lmgpu_handle* h, int32_t front, int32_t* slots, double* RSd) {
  if (!h) return LMGPU_INVALID;
  if (!h->finalized || front < 0 || front >= (int)h->plan.fronts.size()) {
    h->err = "lmgpu_get_front: refused (!h->finalized || front < 0 || front >= (int)h->plan.fronts.size())";
    return LMGPU_INVALID;
  }
  const Front& fr = h->plan.fronts[front];
  if (slots) std::memcpy(slots, fr.vars.data(), fr.vars.size() * sizeof(int32_t));
  if (RSd) {
    int rc = need_device(h);
    if (rc) return rc;
    if (!h->solved || !h->front_active[front]) return LMGPU_INVALID;
    const FrontDesc& F = h->h_fronts[front];
    std::vector<double> rows((size_t)F.nf * F.ld_rsd);
    HIPCHECK(hipMemcpy(rows.data(), h->pool + F.rsd_off, rows.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < F.nf; i++)
      for (int j = 0; j < F.n; j++) RSd[(size_t)j * F.nf + i] = (j >= i) ? rows[(size_t)i * F.ld_rsd + j] : 0.0;
  }
  return LMGPU_OK;
}

int lmgpu_comm_unique_id(char id128[128]) {
  ncclUniqueId id;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  if (ncclGetUniqueId(&id) != ncclSuccess) return LMGPU_HIP_ERROR;
  std::memcpy(id128, &id, 128);
  return LMGPU_OK;
}

int lmgpu_comm_init(lmgpu_handle* h, const char id128[128]) {
  if (!h) return LMGPU_INVALID;
  if (!id128) {
    h->err = "lmgpu_comm_init: refused (!id128)";
    return LMGPU_INVALID;
  }
  int rc = need_device(h);
  if (rc) return rc;
  ncclUniqueId id;
  std::memcpy(&id, id128, 128);
  HIPCHECK(hipSetDevice(h->device));
  NCCLCHECK(ncclCommInitRank(&h->comm, h->cfg.world_size, id, h->cfg.rank));
  NCCLCHECK(ncclCommSplit(h->comm, 0, h->cfg.rank, &h->comm_data, nullptr));  // collective: every rank calls it right after the init
  return LMGPU_OK;
}

#ifdef LMGPU_TEST_HOOKS  // in-process stand-in for the communicator: exported by liblmgpu_test.so only
int lmgpu_local_group_create(int32_t world_size, lmgpu_local_group** out) {
  if (!out || world_size < 1) return LMGPU_INVALID;
  lmgpu_local_group* g = new lmgpu_local_group();
  g->world = world_size;
  g->ptr.assign(world_size, nullptr);
  g->stream.assign(world_size, nullptr);
  *out = g;
  return LMGPU_OK;
}
#endif
// Host-only check of the ticket order of a chained launch (kernels_step.hpp: chain_schedule): every logical workgroup of every
// step exactly once, and every dependency step_body waits for at an earlier ticket.  0 = valid, else the 1-based ticket at fault.
int lmgpu_selftest_chain_schedule(int n, int nf, int i0, int nsteps, int far_pct) {
  if (nsteps < 1 || n <= nf || nf < (i0 + nsteps + 1) * 256 - 255) return -1;
  int split_pct = 60;
  if (far_pct >= 100000) {  // + 100000 x (split_pct + 1): another share of the update tasks in front of the row-panel workgroups
    split_pct = far_pct / 100000 - 1;
    far_pct %= 100000;
  }
  const bool merge2 = far_pct >= 1000;
  if (merge2) far_pct -= 1000;
  const std::vector<int2> tasks = chain_schedule(n, nf, i0, nsteps, far_pct, merge2, split_pct);
  struct Geo { int T, S, nHead, nTA, nd, nTB, ntrsm, grid; };
  std::vector<Geo> g(nsteps);
  std::vector<std::vector<int>> pos(nsteps);
  std::vector<std::vector<char>> pair_tail(nsteps);  // the update tile's task of this step also applied the step before
  long total = 0;
  for (int s = 0; s < nsteps; s++) {
    const int i = i0 + s, m = n - (i + 1) * 256, kbn = std::min(nf, (i + 2) * 256) - (i + 1) * 256;
    if (m <= 0 || kbn <= 0 || kbn % 64) return -1;
    Geo& G = g[s];
    G.T = (m + 127) / 128;
    G.S = (m + 63) / 64;
    G.nHead = 4 * step_head_units(G.S);
    G.nTA = step_ta_workgroups(G.S);
    G.nd = kbn / 64;
    G.nTB = G.T * (G.T + 1) / 2 - (G.T >= 2 ? 2 * G.T - 1 : G.T);
    G.ntrsm = (m - kbn + 63) / 64;
    G.grid = step_grid(m, kbn);
    if (G.grid != G.nTA + G.nd + G.nTB + G.ntrsm) return -1;
    pos[s].assign(G.grid, -1);
    pair_tail[s].assign(G.grid, 0);
    total += G.grid;
  }
  auto tb_index = [&](int s, int ti, int tj) {  // logical workgroup of update tile (ti, tj), ti >= 2
    int off = g[s].nTA + g[s].nd;
    for (int r = 2; r < ti; r++) off += g[s].T - r;
    return off + (tj - ti);
  };
  auto tb_tile = [&](int s, int t, int* ti, int* tj) {
    int rem = t - g[s].nTA - g[s].nd;
    *ti = 2;
    while (rem >= g[s].T - *ti) {
      rem -= g[s].T - *ti;
      (*ti)++;
    }
    *tj = *ti + rem;
  };
  long covered = 0;
  for (size_t k = 0; k < tasks.size(); k++) {
    const bool mg = (tasks[k].x & CHAIN_TASK_MERGED) != 0;
    const int s = tasks[k].x & ~CHAIN_TASK_MERGED, t = tasks[k].y;
    if (s < 0 || s >= nsteps || t < 0 || t >= g[s].grid || pos[s][t] >= 0) return (int)k + 1;
    pos[s][t] = (int)k;
    covered++;
    if (mg) {  // only update tiles pair up, and the pair's first half is the same tile one step earlier
      if (!merge2 || s < 1 || t < g[s].nTA + g[s].nd || t >= g[s].nTA + g[s].nd + g[s].nTB) return (int)k + 1;
      int ti, tj;
      tb_tile(s, t, &ti, &tj);
      const int tp = tb_index(s - 1, ti + 2, tj + 2);
      if (pos[s - 1][tp] >= 0) return (int)k + 1;
      pos[s - 1][tp] = (int)k;
      pair_tail[s][t] = 1;
      covered++;
    }
  }
  if (covered != total) return -2;
  for (int s = 0; s < nsteps; s++) {
    const Geo& G = g[s];
    int last_prev_trsm = -1, last_ta = -1, last_diag = -1;
    if (s > 0)
      for (int t = 0; t < g[s - 1].ntrsm; t++) last_prev_trsm = std::max(last_prev_trsm, pos[s - 1][g[s - 1].nTA + g[s - 1].nd + g[s - 1].nTB + t]);
    for (int t = 0; t < G.nTA; t++) last_ta = std::max(last_ta, pos[s][t]);
    for (int t = 0; t < G.nd; t++) last_diag = std::max(last_diag, pos[s][G.nTA + t]);
    for (int t = 0; t < G.grid; t++) {
      const int me = pos[s][t];
      int ti = -1, tj = -1;  // 128-tile whose entries this workgroup updates (none for the panel roles)
      if (t < G.nTA) {
        int si, sj;
        if (t < G.nHead) {
          const int u = t >> 2;
          sj = (u >= 6) ? 3 : ((u >= 3) ? 2 : (u >= 1 ? 1 : 0));
          si = u - sj * (sj + 1) / 2;
          ti = si >> 1;
          tj = sj >> 1;
        } else {
          ti = (t - G.nHead) & 1;
          tj = 2 + ((t - G.nHead) >> 1);
        }
      } else if (t < G.nTA + G.nd) {
        if (me < last_ta) return me + 1;  // diagonal workgroups read the head tiles
      } else if (t < G.nTA + G.nd + G.nTB) {
        tb_tile(s, t, &ti, &tj);
      } else {
        if (me < last_ta || me < last_diag) return me + 1;  // row-panel workgroups read head tiles and diagonal tiles
      }
      if (ti >= 0 && s > 0) {
        // the finished panel of this step -- for the first half of a pair this is implied by the second half's wait (panels finish in order)
        if (me < last_prev_trsm) return me + 1;
        const int before = pos[s - 1][tb_index(s - 1, ti + 2, tj + 2)];
        if (pair_tail[s][t] ? before != me : me <= before) return me + 1;  // the tile this one continues (or is paired with)
      }
    }
  }
  return 0;
}

#ifdef LMGPU_TEST_HOOKS
int lmgpu_local_group_destroy(lmgpu_local_group* g) {
  delete g;
  return LMGPU_OK;
}
int lmgpu_comm_init_local(lmgpu_handle* h, lmgpu_local_group* g) {
  if (!h) return LMGPU_INVALID;
  if (!g || g->world != h->cfg.world_size) {
    h->err = "lmgpu_comm_init_local: refused (!g || g->world != h->cfg.world_size)";
    return LMGPU_INVALID;
  }
  h->lgroup = g;
  return LMGPU_OK;
}
#endif

// ---------------------------------------------------------------- peak micro-benchmarks

__global__ __launch_bounds__(256) void peak_mfma_f64_kernel(double* out, int iters) {
  double4_t acc[4];
  for (int i = 0; i < 4; i++) acc[i] = double4_t{0, 0, 0, 0};
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// Diagnostic form of the same loop: NACC independent accumulators per wave and, around the loop, one pair of stamps per workgroup --
// s_memtime (shader-clock cycles) and s_memrealtime (constant 100 MHz) -- written to a buffer nothing else reads
// (MI355X_MICROARCH.md, DVFS give-back item 6): the clock the chip HOLDS under this load = d(memtime) / d(memrealtime) x 100 MHz.
#define PEAK_CLOCK_KERNEL(NAME, NACC)                                                                                   \
__global__ __launch_bounds__(256) void NAME(double* out, unsigned long long* stamps, int iters) { \
  double4_t acc[NACC]; \
  for (int i = 0; i < NACC; i++) acc[i] = double4_t{0, 0, 0, 0}; \
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9; \
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
  __builtin_amdgcn_sched_barrier(0); \
  for (int it = 0; it < iters; it++) { \
_Pragma("unroll") \
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0); \
  } \
  double s = 0; \
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3]; \
  out[blockIdx.x * 256 + threadIdx.x] = s; \
  __builtin_amdgcn_sched_barrier(0); \
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
  if (threadIdx.x == 0) { \
    stamps[2 * blockIdx.x] = c1 - c0; \
    stamps[2 * blockIdx.x + 1] = r1 - r0; \
  } \
}
PEAK_CLOCK_KERNEL(peak_mfma_f64_clock_kernel4, 4)
PEAK_CLOCK_KERNEL(peak_mfma_f64_clock_kernel8, 8)
// the register shape of the trailing-update tile: 4 x 4 accumulators of a 64x64 wave tile, four A and four B operand registers.
// (Round 3: the 4- and 8-accumulator loops above stop at ~49 TFLOP/s, which round 2 took for the instruction's ceiling; the update tile
// itself runs at 61-69 TFLOP/s once its memory traffic is taken away (tools/syrk4_bench.hip), and so does this loop.)
__global__ __launch_bounds__(256, 2) void peak_mfma_f64_clock_kernel16(double* out, unsigned long long* stamps, int iters) {
  double4_t acc[4][4];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) acc[i][j] = double4_t{0, 0, 0, 0};
  double a[4], b[4];
  for (int i = 0; i < 4; i++) {
    a[i] = 0.25 + threadIdx.x * 1e-3 + i * 0.01;
    b[i] = 0.5 - threadIdx.x * 1e-3 - i * 0.01;
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_sched_barrier(0);
  for (int it = 0; it < iters; it += 4) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; i++) {  // keep the operands changing (a loop-invariant operand set is not what a tile sees)
      a[i] = -a[i];
      b[i] = -b[i];
    }
  }
  double s = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

__global__ __launch_bounds__(256) void peak_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

int lmgpu_peak_mfma_f64(int32_t device, int32_t iters, double* tflops) {
  if (hipSetDevice(device) != hipSuccess) return LMGPU_HIP_ERROR;
  const int blocks = 256 * 8;
  double* out = nullptr;
  if (hipMalloc((void**)&out, (size_t)blocks * 256 * sizeof(double)) != hipSuccess) return LMGPU_HIP_ERROR;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(peak_mfma_f64_kernel, dim3(blocks), dim3(256), 0, 0, out, 16);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(peak_mfma_f64_kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 /*waves*/ * (double)iters * 4 * 2048.0;
  *tflops = flops / (ms * 1e-3) / 1e12;
  (void)hipFree(out);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return LMGPU_OK;
}

// tflops: the sustained rate of the loop; sclk_mhz: the median in-kernel shader clock while it ran (after ~1 s of back-to-back
// launches so that the power state has settled); flop_per_clk_simd = tflops / (SIMDs x clock): 32 = one v_mfma_f64_16x16x4_f64
// (2048 flop) per 64 cycles and SIMD, the rate the 78.6 TFLOP/s datasheet figure assumes AT 2.4 GHz.
int lmgpu_peak_mfma_f64_clock(int32_t device, int32_t iters, int32_t n_acc, double* tflops, double* sclk_mhz, double* flop_per_clk_simd) {
  if (hipSetDevice(device) != hipSuccess || (n_acc != 4 && n_acc != 8 && n_acc != 16) || !tflops || !sclk_mhz) return LMGPU_HIP_ERROR;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return LMGPU_HIP_ERROR;
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * (n_acc == 16 ? 2 : 8);  // 16 accumulators: two workgroups per CU, the occupancy of the update tile
  double* out = nullptr;
  unsigned long long* stamps = nullptr;
  if (hipMalloc((void**)&out, (size_t)blocks * 256 * sizeof(double)) != hipSuccess) return LMGPU_HIP_ERROR;
  if (hipMalloc((void**)&stamps, (size_t)blocks * 2 * sizeof(unsigned long long)) != hipSuccess) return LMGPU_HIP_ERROR;
  auto launch = [&](int it) {
    if (n_acc == 16) hipLaunchKernelGGL(peak_mfma_f64_clock_kernel16, dim3(blocks), dim3(256), 0, 0, out, stamps, 4 * it);
    else if (n_acc == 4) hipLaunchKernelGGL(peak_mfma_f64_clock_kernel4, dim3(blocks), dim3(256), 0, 0, out, stamps, it);
    else hipLaunchKernelGGL(peak_mfma_f64_clock_kernel8, dim3(blocks), dim3(256), 0, 0, out, stamps, it);
  };
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  launch(iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  launch(iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms1 = 0;
  (void)hipEventElapsedTime(&ms1, e0, e1);
  const int reps = std::max(1, std::min(400, (int)(1000.0 / std::max(0.05f, ms1))));  // ~1 s of back-to-back launches
  for (int r = 0; r < reps; r++) launch(iters);
  (void)hipEventRecord(e0, 0);
  launch(iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)blocks * 2);
  (void)hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::vector<double> clk;
  for (int b = 0; b < blocks; b++)
    if (h[2 * b + 1] > 0) clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0);
  std::sort(clk.begin(), clk.end());
  *sclk_mhz = clk.empty() ? 0.0 : clk[clk.size() / 2];
  const double flops = (double)blocks * 4 /*waves*/ * (double)iters * n_acc * 2048.0;  // (the 16-accumulator kernel is launched with 4 x iters and steps its loop by 4)
  *tflops = flops / (ms * 1e-3) / 1e12;
  if (flop_per_clk_simd) *flop_per_clk_simd = (*sclk_mhz > 0) ? (*tflops * 1e12) / ((double)cus * 4 * *sclk_mhz * 1e6) : 0.0;
  (void)hipFree(out);
  (void)hipFree(stamps);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return LMGPU_OK;
}

int lmgpu_peak_hbm_copy(int32_t device, int64_t bytes, int32_t iters, double* gbps) {
  if (hipSetDevice(device) != hipSuccess) return LMGPU_HIP_ERROR;
  float4 *a = nullptr, *b = nullptr;
  if (hipMalloc((void**)&a, bytes) != hipSuccess || hipMalloc((void**)&b, bytes) != hipSuccess) return LMGPU_HIP_ERROR;
  (void)hipMemset(a, 1, bytes);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const size_t n = bytes / 16;
  hipLaunchKernelGGL(peak_copy_kernel, dim3(2048), dim3(256), 0, 0, a, b, n);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL(peak_copy_kernel, dim3(2048), dim3(256), 0, 0, a, b, n);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *gbps = 2.0 * (double)bytes * iters / (ms * 1e-3) / 1e9;
  (void)hipFree(a);
  (void)hipFree(b);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return LMGPU_OK;
}

}  // extern "C"

#include "isam2.hpp"

#ifdef LDSF_STAMPS  // development aid: tools/ldsf_phases.py
extern "C" int lmgpu_debug_ldsf(unsigned long long* out16, int reset) {
  if (reset) {
    unsigned long long z[16] = {0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(lmgpu::ldsf_dbg), z, sizeof(z));
  }
  return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(lmgpu::ldsf_dbg), 16 * sizeof(unsigned long long));
}
#endif
