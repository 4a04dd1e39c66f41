// isam2_harness — drives the incremental path of liblmgpu.so (lmgpu_isam2_*) from C++ in the call order of the reference-side
// wrapper (include/lmgpu_gtsam_adapter.h: GpuISAM2::update = add_variables, add_factors in graph order, update_with; then
// calculateEstimate), without GTSAM and without Python between the updates: what an update costs a C++ caller.
// The constrained COLAMD the library asks for per update is the CALLER's (lmgpu_ccolamd_fn): here the reference's vendored CCOLAMD,
// compiled by oracle/Makefile into oracle/_ref/libccolamd_ref.so and loaded with dlopen (test infrastructure, like the tests'
// Python binding of the same library), with GTSAM's knobs (gtsam/inference/Ordering.cpp:94-97).
//
// The sequence of updates comes in a neutral text file written by the test / tool from the Python mirror's graphs:
//   ISAM2 relinearizeThreshold relinearizeSkip enableRelinearization wildfireThreshold
//   [OPT find_unused_slots 1]
//   UPDATE nv nf nremove          then nv lines  V key type store d0 d1 ...
//                                            or  W key prevkey dx dy dtheta   (a Pose2 initialised at update time as the device's
//                                                own estimate of prevkey composed with the odometry: lmgpu_isam2_get_value =
//                                                calculateEstimate(prevkey), the loop of timing/timeIncremental.cpp:84-170)
//                                 then nf lines  F type k0 k1 k2 | nmeas meas... | noise_kind nnoise noise...
//                                 then one line  R idx0 idx1 ...            (removeFactorIndices, nremove entries)
//                                 optionally     C n key group ...          (ISAM2UpdateParams::constrainedKeys)
//                                                X n key ...                (ISAM2UpdateParams::extraReelimKeys)
//                                                M n key ...                (ISAM2::marginalizeLeaves after the update: the fixed-lag
//                                                                            smoother's use, IncrementalFixedLagSmoother.cpp)
//   ... END
// usage: isam2_harness <sequence file> <device> <libccolamd_ref.so | replay:FILE> [record:FILE] [repeat:N]
//   repeat:N      run the sequence N times (fresh handle each), report the last run
//   record:FILE   additionally writes every ordering the callback returned (int32: n_cols, then the permutation) -- run once with the real
//                 CCOLAMD to make a fixture
//   replay:FILE   the callback returns the recorded orderings instead of computing them (no CCOLAMD loaded): the orderings are then a
//                 fixture-carried boundary input, like the METIS permutation of the batch benchmark (bench.py --workload isam2).  A call
//                 whose column count differs from the recording, or a recording that runs out, fails the update.
// Output: one JSON object: updates, seconds inside the library calls, ms per update, the final estimate (key, packed value).
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/lmgpu.h"

namespace {

typedef size_t (*recommended_fn)(int, int, int);
typedef void (*defaults_fn)(double*);
typedef int (*ccolamd_fn)(int, int, int, int*, int*, double*, int*, int*);
recommended_fn p_recommended = nullptr;
defaults_fn p_defaults = nullptr;
ccolamd_fn p_ccolamd = nullptr;
double g_colamd_seconds = 0;

std::FILE* g_record = nullptr;
std::vector<int32_t> g_replay;
size_t g_replay_at = 0;
bool g_use_replay = false;

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// the body of Ordering::ColamdConstrained (gtsam/inference/Ordering.cpp:86-108) on the arrays the library hands over
int colamd_cb(void*, int32_t n_rows, int32_t n_cols, const int32_t* col_ptr, const int32_t* row_idx, const int32_t* cmember, int32_t* perm_out) {
  const double t0 = now();
  if (g_use_replay) {
    if (g_replay_at + 1 + (size_t)n_cols > g_replay.size() || g_replay[g_replay_at] != n_cols) return 0;
    std::vector<char> seen((size_t)n_cols, 0);
    for (int j = 0; j < n_cols; j++) {
      const int32_t c = g_replay[g_replay_at + 1 + (size_t)j];
      if (c < 0 || c >= n_cols || seen[(size_t)c]) return 0;  // not a permutation of this call's columns
      seen[(size_t)c] = 1;
      perm_out[j] = c;
    }
    g_replay_at += 1 + (size_t)n_cols;
    g_colamd_seconds += now() - t0;
    return 1;
  }
  const size_t Alen = p_recommended(col_ptr[n_cols], n_rows, n_cols);
  std::vector<int> A(Alen), p(col_ptr, col_ptr + n_cols + 1), cm(cmember, cmember + n_cols);
  for (int i = 0; i < col_ptr[n_cols]; i++) A[i] = row_idx[i];
  double knobs[20];
  p_defaults(knobs);
  knobs[0] = -1;  // CCOLAMD_DENSE_ROW
  knobs[1] = -1;  // CCOLAMD_DENSE_COL
  int stats[20];
  const int rv = p_ccolamd(n_rows, n_cols, (int)Alen, A.data(), p.data(), knobs, stats, cm.data());
  if (rv == 1) {
    for (int j = 0; j < n_cols; j++) perm_out[j] = p[j];
    if (g_record) {
      std::fwrite(&n_cols, sizeof(int32_t), 1, g_record);
      std::fwrite(perm_out, sizeof(int32_t), (size_t)n_cols, g_record);
    }
  }
  g_colamd_seconds += now() - t0;
  return rv == 1 ? 1 : 0;
}

struct Update {
  std::vector<uint64_t> vkeys;
  std::vector<int32_t> vtypes;
  std::vector<double> vpacked;
  struct Rel {  // variable i of this update is prev (+) odometry, resolved when the update is issued
    size_t at;  // offset of its three doubles in vpacked
    uint64_t prev;
    double d[3];
  };
  std::vector<Rel> rel;
  struct F {
    int32_t type, noise_kind;
    uint64_t k[3];
    std::vector<double> meas, noise;
  };
  std::vector<F> facs;
  std::vector<uint64_t> remove;
  // ISAM2UpdateParams::constrainedKeys / extraReelimKeys of this update and the keys ISAM2::marginalizeLeaves takes out after it
  bool has_constrained = false;
  std::vector<uint64_t> ckeys, extra, marginalize;
  std::vector<int32_t> cgroups;
};

int fail(const char* what, const char* detail) {
  std::printf("{\"error\": \"%s: %s\"}\n", what, detail ? detail : "");
  return 1;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: isam2_harness <sequence file> <device> <libccolamd_ref.so | replay:FILE> [record:FILE]\n");
    return 2;
  }
  const std::string ord = argv[3];
  if (ord.rfind("replay:", 0) == 0) {
    std::FILE* f = std::fopen(ord.c_str() + 7, "rb");
    if (!f) return fail("open", ord.c_str() + 7);
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    g_replay.resize((size_t)bytes / sizeof(int32_t));
    if (std::fread(g_replay.data(), sizeof(int32_t), g_replay.size(), f) != g_replay.size()) return fail("read", ord.c_str() + 7);
    std::fclose(f);
    g_use_replay = true;
  } else {
    void* so = dlopen(argv[3], RTLD_NOW);
    if (!so) return fail("dlopen", dlerror());
    p_recommended = (recommended_fn)dlsym(so, "ccolamd_recommended");
    p_defaults = (defaults_fn)dlsym(so, "ccolamd_set_defaults");
    p_ccolamd = (ccolamd_fn)dlsym(so, "ccolamd");
    if (!p_recommended || !p_defaults || !p_ccolamd) return fail("dlsym", "ccolamd symbols");
  }
  for (int a = 4; a < argc; a++)
    if (std::string(argv[a]).rfind("record:", 0) == 0) {
      g_record = std::fopen(argv[a] + 7, "wb");
      if (!g_record) return fail("open", argv[a] + 7);
    }

  std::ifstream is(argv[1]);
  if (!is) return fail("open", argv[1]);
  std::string w;
  lmgpu_isam2_params prm{};
  if (!(is >> w) || w != "ISAM2" || !(is >> prm.relinearizeThreshold >> prm.relinearizeSkip >> prm.enableRelinearization >> prm.wildfireThreshold))
    return fail("parse", "header");
  std::vector<Update> updates;
  int find_unused_slots = 0;
  is >> w;
  while (w == "OPT") {  // OPT find_unused_slots 1  (ISAM2Params::findUnusedFactorSlots)
    std::string name;
    int val = 0;
    is >> name >> val;
    if (name == "find_unused_slots") find_unused_slots = val;
    is >> w;
  }
  while (w == "UPDATE") {
    size_t nv, nf, nr;
    if (!(is >> nv >> nf >> nr)) return fail("parse", "UPDATE");
    Update u;
    for (size_t i = 0; i < nv; i++) {
      uint64_t k;
      int32_t t;
      int store;
      if (!(is >> w >> k)) return fail("parse", "V");
      if (w == "W") {
        Update::Rel r{u.vpacked.size(), 0, {0, 0, 0}};
        if (!(is >> r.prev >> r.d[0] >> r.d[1] >> r.d[2])) return fail("parse", "W");
        u.vkeys.push_back(k);
        u.vtypes.push_back(LMGPU_POSE2);
        u.vpacked.insert(u.vpacked.end(), 3, 0.0);
        u.rel.push_back(r);
        continue;
      }
      if (w != "V" || !(is >> t >> store)) return fail("parse", "V");
      u.vkeys.push_back(k);
      u.vtypes.push_back(t);
      for (int j = 0; j < store; j++) {
        double d;
        is >> d;
        u.vpacked.push_back(d);
      }
    }
    for (size_t i = 0; i < nf; i++) {
      Update::F f{};
      size_t nm, nn;
      if (!(is >> w >> f.type >> f.k[0] >> f.k[1] >> f.k[2] >> nm) || w != "F") return fail("parse", "F");
      f.meas.resize(nm);
      for (double& d : f.meas) is >> d;
      if (!(is >> f.noise_kind >> nn)) return fail("parse", "noise");
      f.noise.resize(nn);
      for (double& d : f.noise) is >> d;
      u.facs.push_back(std::move(f));
    }
    if (!(is >> w) || w != "R") return fail("parse", "R");
    u.remove.resize(nr);
    for (uint64_t& r : u.remove) is >> r;
    for (;;) {  // optional lines, then the next UPDATE / END
      if (!(is >> w)) {
        w.clear();
        break;
      }
      size_t cnt;
      if (w == "C") {
        is >> cnt;
        u.has_constrained = true;
        u.ckeys.resize(cnt);
        u.cgroups.resize(cnt);
        for (size_t i = 0; i < cnt; i++) is >> u.ckeys[i] >> u.cgroups[i];
      } else if (w == "X") {
        is >> cnt;
        u.extra.resize(cnt);
        for (uint64_t& k : u.extra) is >> k;
      } else if (w == "M") {
        is >> cnt;
        u.marginalize.resize(cnt);
        for (uint64_t& k : u.marginalize) is >> k;
      } else {
        break;
      }
    }
    updates.push_back(std::move(u));
  }

  // repeat:N -- the whole sequence N times, each on a fresh handle, the LAST run reported: what an update costs in a process that has
  // been running for a while (code objects loaded, the runtime's launch configurations seen), as an incremental smoother's host has
  int repeats = 1;
  for (int a = 4; a < argc; a++)
    if (std::string(argv[a]).rfind("repeat:", 0) == 0) repeats = std::max(1, std::atoi(argv[a] + 7));
  auto run = [&](bool report) -> int {
    g_replay_at = 0;
    g_colamd_seconds = 0;
  lmgpu_config cfg{};
  cfg.device = std::atoi(argv[2]);
  cfg.world_size = 1;
  lmgpu_isam2* h = nullptr;
  if (lmgpu_isam2_create(&cfg, &prm, &colamd_cb, nullptr, &h) != LMGPU_OK) return fail("lmgpu_isam2_create", h ? lmgpu_isam2_last_error(h) : "");
  if (find_unused_slots && lmgpu_isam2_set_find_unused_factor_slots(h, 1) != LMGPU_OK) return fail("set_find_unused_factor_slots", lmgpu_isam2_last_error(h));
  double lib = 0, worst = 0;
  std::vector<double> per_update;
  lmgpu_isam2_result r{};
  size_t done = 0;
  double est_seconds = 0;
  size_t est_calls = 0, marginalized = 0;
  for (Update& u : updates) {
    const double t0 = now();
    int rc = LMGPU_OK;
    for (const Update::Rel& r : u.rel) {  // Pose2::compose(prev estimate, odometry)
      double a[15];
      int32_t t = -1;
      size_t own = 0, at = 0;  // the previous pose may be a variable of this very update (the first one adds poses 0 and 1)
      for (; own < u.vkeys.size() && u.vkeys[own] != r.prev; own++) at += (u.vtypes[own] == LMGPU_POSE2) ? 3 : 0;
      if (own < u.vkeys.size() && u.vtypes[own] == LMGPU_POSE2 && at < r.at) {
        for (int j = 0; j < 3; j++) a[j] = u.vpacked[at + j];
        t = LMGPU_POSE2;
      } else {
        const double te = now();
        rc = lmgpu_isam2_get_value(h, 0, r.prev, &t, a);
        est_seconds += now() - te;
        est_calls++;
      }
      if (rc != LMGPU_OK || t != LMGPU_POSE2) {
        std::printf("{\"error\": \"calculateEstimate(%" PRIu64 "): rc %d: %s\"}\n", r.prev, rc, lmgpu_isam2_last_error(h));
        lmgpu_isam2_destroy(h);
        return 1;
      }
      const double c = std::cos(a[2]), sn = std::sin(a[2]);
      u.vpacked[r.at] = a[0] + c * r.d[0] - sn * r.d[1];
      u.vpacked[r.at + 1] = a[1] + sn * r.d[0] + c * r.d[1];
      u.vpacked[r.at + 2] = a[2] + r.d[2];
    }
    if (rc == LMGPU_OK) rc = lmgpu_isam2_add_variables(h, (int32_t)u.vkeys.size(), u.vkeys.data(), u.vtypes.data(), u.vpacked.data());
    for (const Update::F& f : u.facs)
      if (rc == LMGPU_OK)
        rc = lmgpu_isam2_add_factors(h, f.type, 1, f.k, f.meas.data(), f.noise_kind, f.noise.empty() ? nullptr : f.noise.data());
    if (rc == LMGPU_OK) {
      const lmgpu_isam2_update_params up{(int32_t)u.remove.size(), u.remove.data(), u.has_constrained ? 1 : 0, (int32_t)u.ckeys.size(), u.ckeys.data(),
                                         u.cgroups.data(), 0, nullptr, (int32_t)u.extra.size(), u.extra.data(), 0, 0};
      rc = lmgpu_isam2_update_with(h, &up, &r);
    }
    if (rc == LMGPU_OK && !u.marginalize.empty()) {
      rc = lmgpu_isam2_marginalize_leaves(h, (int32_t)u.marginalize.size(), u.marginalize.data(), nullptr, nullptr);
      marginalized += u.marginalize.size();
    }
    const double dt = now() - t0;
    lib += dt;
    per_update.push_back(dt);
    worst = dt > worst ? dt : worst;
    if (rc != LMGPU_OK) {
      std::printf("{\"error\": \"update %zu: rc %d: %s\"}\n", done, rc, lmgpu_isam2_last_error(h));
      lmgpu_isam2_destroy(h);
      return 1;
    }
    done++;
  }
  const double t1 = now();
  const int n = lmgpu_isam2_num_variables(h);
  std::vector<uint64_t> keys((size_t)n);
  std::vector<int32_t> types((size_t)n);
  if (lmgpu_isam2_get_values(h, 2, keys.data(), types.data(), nullptr) != LMGPU_OK) return fail("get_values", lmgpu_isam2_last_error(h));
  static const int store[6] = {3, 12, 3, 15, 2, 5};  // lmgpu.h: doubles per packed value of each variable type
  size_t tot = 0;
  for (int32_t t : types) tot += (size_t)store[t];
  std::vector<double> packed(tot);
  if (lmgpu_isam2_get_values(h, 0, nullptr, nullptr, packed.data()) != LMGPU_OK) return fail("calculateEstimate", lmgpu_isam2_last_error(h));
  const double t_est = now() - t1;
  // the first update pays the one-time costs of a fresh handle (pool allocation, first launches): reported apart
  if (report && per_update.size() <= 64) {  // a short sequence: every update's time, in order (stderr)
    std::fprintf(stderr, "per update (ms):");
    for (double t : per_update) std::fprintf(stderr, " %.3f", 1e3 * t);
    std::fprintf(stderr, "\n");
  }
  const double first_ms = per_update.empty() ? 0.0 : 1e3 * per_update[0];
  const double steady_ms = per_update.size() > 1 ? 1e3 * (lib - per_update[0]) / (double)(per_update.size() - 1) : first_ms;
  std::sort(per_update.begin(), per_update.end());
  auto pct = [&](double q) { return per_update.empty() ? 0.0 : 1e3 * per_update[std::min(per_update.size() - 1, (size_t)(q * per_update.size()))]; };
  if (report) std::printf("{\"updates\": %zu, \"library_seconds\": %.6f, \"first_update_ms\": %.4f, \"ms_per_update_after_first\": %.6f, \"ms_per_update\": %.6f, \"p50_ms\": %.4f, \"p95_ms\": %.4f, \"p99_ms\": %.4f, \"worst_update_ms\": %.4f, \"ccolamd_callback_seconds\": %.6f, "
              "\"calculate_estimate_ms\": %.4f, \"single_estimates\": %zu, \"single_estimate_ms\": %.5f, \"variables\": %d, \"cliques\": %d, \"marginalized\": %zu, \"factor_slots\": %d, \"estimate\": [",
              done, lib, first_ms, steady_ms, done ? 1e3 * lib / done : 0.0, pct(0.50), pct(0.95), pct(0.99), 1e3 * worst, g_colamd_seconds, 1e3 * t_est, est_calls,
              est_calls ? 1e3 * est_seconds / est_calls : 0.0, n, r.cliques, marginalized, lmgpu_isam2_num_factors(h));
  const double* q = packed.data();
  for (int i = 0; report && i < n; i++) {
    std::printf("%s[%" PRIu64, i ? ", " : "", keys[(size_t)i]);
    for (int j = 0; j < store[types[(size_t)i]]; j++) std::printf(", %.17g", q[j]);
    std::printf("]");
    q += store[types[(size_t)i]];
  }
  if (report) std::printf("]}\n");
  lmgpu_isam2_destroy(h);
    return 0;
  };
  for (int rep = 0; rep < repeats; rep++) {
    const int rc = run(rep + 1 == repeats);
    if (rc) return rc;
  }
  if (g_record) std::fclose(g_record);
  return 0;
}
