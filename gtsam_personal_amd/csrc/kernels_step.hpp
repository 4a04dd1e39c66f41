// A step of the blocked Cholesky of an HBM front = trailing update with outer panel i + factorisation of panel i+1, as dataflow
// between workgroups.  Logical workgroups of a step (step_body):
//   [ head tiles: the rows of panel i+1, 32x32 quadrants for strips 0..3, 64x64 sub-tiles beyond ]  [ diagonal workgroups of panel i+1 ]
//   [ all other trailing tiles, 128x128 ]                                                           [ row-panel (trsm) workgroups of panel i+1 ]
// The head tiles publish per-strip counters; the diagonal workgroups start on them and run their potrf chain BESIDE the bulk
// of the trailing update (look-ahead without a second stream); the row-panel workgroups finish the panel.
// step_kernel = one step per launch.  chain_kernel = all fused steps of a front in one launch, ticket order from
// chain_schedule (host, below) -- see the comment there for what the order buys.  front_tail_kernel (kernels_potrf.hpp) ends the
// front when the remainder is small.  Every inter-workgroup hand-off is the release/acquire protocol of kernels_potrf.hpp.
#pragma once
#include "kernels_dense.hpp"
#include "kernels_potrf.hpp"

// development aid (tools/microbench.hip defines them): where the workgroups of a chained launch spend their time
#ifndef CHAIN_PROF_BEGIN
#define CHAIN_PROF_BEGIN()
#define CHAIN_PROF_WAITED()
#define CHAIN_PROF_END(cls)
#endif

namespace lmgpu {

struct StepArgs {
  double* A;
  int ld, n, nf;
  int p0, kp;       // finished panel: rows p0 .. p0+kp-1; trailing rows/columns r0 = p0 + kp .. n-1
  int kb_next;      // > 0: rows r0 .. r0+kb_next-1 are the next panel (a multiple of 64), factor it in this launch
  int front_id;
  int* status;
  double* inv16;
  unsigned int* flags;
  const double* S;  // != nullptr: partial-assembly buffer (same layout as A) whose rows r0 .. r0+255 are folded in by this launch
};

// 64x64 tile of the trailing update for the rows of the NEXT panel (the head of the dependency chain): four waves, 32x32 each,
// operand fragments straight from L2 (no LDS staging; 1/4 of a 128x128 tile's latency).  C[i][j] -= sum_p P[p][i] P[p][j].
// Sadd != nullptr: the rows of the next panel also receive their (so far separate) assembled contributions, C += Sadd - P^T P
__device__ __forceinline__ void syrk_subtile64(double* __restrict__ A, int ld, int n, int p0, int kp, int r0, int si, int sj,
                                               const double* __restrict__ Sadd) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kk = lane >> 4, cc = lane & 15;
  const int wr = wave >> 1, wc = wave & 1;
  if (si == sj && wr > wc) return;
  const int i0 = r0 + 64 * si + 32 * wr, j0 = r0 + 64 * sj + 32 * wc;
  if (i0 >= n || j0 >= n) return;
  const double* P = A + (size_t)p0 * ld;
  double4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) acc[a][b] = double4_t{0, 0, 0, 0};
  // operand fragments three 16-deep stages ahead of the MFMAs (a ring of four register stages): with one stage in flight the
  // loop ran at one L2 round trip per 16 k (24-40 us per tile at K = 256, the head of every step's dependency chain)
  double af[4][4][2], bf[4][4][2];
  auto load_stage = [&](int k0, double(&a_)[4][2], double(&b_)[4][2]) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int k = min(k0 + 4 * s + kk, kp - 1);
      const bool kv = k0 + 4 * s + kk < kp;
      const double* row = P + (size_t)k * ld;
#pragma unroll
      for (int a = 0; a < 2; a++) {
        const double v = row[i0 + 16 * a + cc];
        a_[s][a] = kv ? -v : 0.0;
      }
#pragma unroll
      for (int b = 0; b < 2; b++) b_[s][b] = row[j0 + 16 * b + cc];
    }
  };
#pragma unroll
  for (int st = 0; st < 3; st++) load_stage(16 * st, af[st], bf[st]);
  for (int k0 = 0; k0 < kp; k0 += 64) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      load_stage(k0 + 16 * (u + 3), af[(u + 3) & 3], bf[(u + 3) & 3]);  // past kp: clamped address, zero A operand
      if (k0 + 16 * u < kp) {
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
          for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u][s][a], bf[u][s][b], acc[a][b], 0, 0, 0);
      }
    }
  }
  double cur[2][2][4];  // all 16 loads in flight, then the stores
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const size_t at = (size_t)min(i0 + 16 * a + kk + 4 * r, n - 1) * ld + min(j0 + 16 * b + cc, n - 1);
        cur[a][b][r] = A[at];
        if (Sadd) cur[a][b][r] += Sadd[at];
      }
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = i0 + 16 * a + kk + 4 * r, col = j0 + 16 * b + cc;
        if (row < n && col < n && col >= row) A[(size_t)row * ld + col] = cur[a][b][r] + acc[a][b][r];
      }
}

// The sub-tiles of column strips 0..3 feed the diagonal workgroups (the head of the launch's dependency chain) and are cut once
// more: one workgroup per 32x32 quadrant, one 16x16 MFMA tile per wave, i.e. 64 dependent MFMAs per wave at K = 256 instead of
// 256 (in-kernel timestamps: the first potrf waited 22-40 us for its 64x64 sub-tile, MFMA-bound on SIMDs shared with update tiles).
__device__ __forceinline__ void syrk_quadrant32(double* __restrict__ A, int ld, int n, int p0, int kp, int r0, int si, int sj, int qd,
                                                const double* __restrict__ Sadd) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kk = lane >> 4, cc = lane & 15;
  const int wr = wave >> 1, wc = wave & 1, qr = qd >> 1, qc = qd & 1;
  if (si == sj && (qr > qc || (qr == qc && wr > wc))) return;
  const int i0 = r0 + 64 * si + 32 * qr + 16 * wr, j0 = r0 + 64 * sj + 32 * qc + 16 * wc;
  if (i0 >= n || j0 >= n) return;
  const double* P = A + (size_t)p0 * ld;
  double4_t acc = double4_t{0, 0, 0, 0};
  double af[4][4], bf[4][4];
  auto load_stage = [&](int k0, double(&a_)[4], double(&b_)[4]) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int k = min(k0 + 4 * s + kk, kp - 1);
      const double* row = P + (size_t)k * ld;
      const double v = row[i0 + cc];
      a_[s] = (k0 + 4 * s + kk < kp) ? -v : 0.0;
      b_[s] = row[j0 + cc];
    }
  };
#pragma unroll
  for (int st = 0; st < 3; st++) load_stage(16 * st, af[st], bf[st]);
  const size_t at[4] = {(size_t)min(i0 + kk, n - 1) * ld + min(j0 + cc, n - 1), (size_t)min(i0 + kk + 4, n - 1) * ld + min(j0 + cc, n - 1),
                        (size_t)min(i0 + kk + 8, n - 1) * ld + min(j0 + cc, n - 1), (size_t)min(i0 + kk + 12, n - 1) * ld + min(j0 + cc, n - 1)};
  double cur[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    cur[r] = A[at[r]];
    if (Sadd) cur[r] += Sadd[at[r]];
  }
  for (int k0 = 0; k0 < kp; k0 += 64) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      load_stage(k0 + 16 * (u + 3), af[(u + 3) & 3], bf[(u + 3) & 3]);  // past kp: clamped address, zero A operand
      if (k0 + 16 * u < kp) {
#pragma unroll
        for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u][s], bf[u][s], acc, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = i0 + kk + 4 * r, col = j0 + cc;
    if (row < n && col < n && col >= row) A[(size_t)row * ld + col] = cur[r] + acc[r];
  }
}

// workgroups of a step launch that factors a next panel: shared by the kernel and its launchers
__host__ __device__ inline int step_head_units(int S) { return S >= 4 ? 10 : S * (S + 1) / 2; }  // 64x64 sub-tiles of strips 0..3
// + beyond strip 3 the two 128x128 tiles (tile rows 0, 1) of every tile column from 2 on.  (Through round 2 these rows went as four 64x64
// sub-tile workgroups per strip, operands straight from the L2: in-kernel stamps showed them taking 20 % of all slot-time of the chained
// launch -- 9 112 workgroups of the C4 root at 64 GFLOP/s each, waiting as long as they worked -- for 5 % of its flop; only strips 0..3
// gate the diagonal workgroups, the row-panel workgroups need theirs a whole diagonal chain later.)
__host__ __device__ inline int step_ta_workgroups(int S) { return 4 * step_head_units(S) + (S > 4 ? 2 * ((S + 1) / 2 - 2) : 0); }
__host__ inline int step_grid(int m, int kbn) {
  const int T = (m + 127) / 128, S = (m + 63) / 64;
  return T * (T + 1) / 2 - (T >= 2 ? 2 * T - 1 : T) + step_ta_workgroups(S) + kbn / 64 + (m - kbn + 63) / 64;
}

#define STEP_LDS_BYTES (2 * 2 * SYRK_KC * SYRK_LDW * 8)
static_assert(STEP_LDS_BYTES >= PDF_LDS_BYTES, "panel roles reuse the update's LDS");

// What a chained step waits for and publishes (null / false for a step that is a launch of its own): the finished panel's
// 128-column blocks ([PDF_PR0 + q] in `prev`, counted by the row-panel workgroups of the step before) and the tile of the
// step before that covers the same entries ([PDF_TD0 + x] in `prev`); its own tiles are counted into [PDF_TD0 + x] of a.flags.
struct ChainLink {
  const unsigned int* prev;  // flag region of the finished panel (= of the step before), nullptr for the first step of a chain
  bool publish;              // a later step of the same launch consumes this step's tiles and panel
  // merged update tile: the task applies the panels of the step before AND of this step in one pass of depth 512 (half the
  // read-modify-write of C per flop, half the prologues and epilogues: 44 -> 52-56 TFLOP/s stand-alone, tools/syrk4_bench.hip);
  // the tile it continues is then the one of TWO steps back (prev2: its flag region, nullptr if that step is not in this launch)
  bool merged = false;
  const unsigned int* prev2 = nullptr;
};

// one 64-strip count per 128-column block q of a trailing matrix with S strips
__device__ __forceinline__ unsigned int chain_pr_need(int S, int q) { return (unsigned int)min(2, S - 2 * q); }
// wait for the inputs of the update of 128-tile (ti, tj) (or a part of it) of a step with T tile rows and S strips
__device__ __forceinline__ bool chain_wait_tile(const ChainLink& c, int T, int S, int ti, int tj, int* s_ok) {
  if (!c.prev) return true;
  // the tile this one continues: of the step before (trailing origin 256 rows / columns earlier), or of two steps back for a merged task
  const int back = c.merged ? 4 : 2;
  const unsigned int* tdr = c.merged ? c.prev2 : c.prev;
  const int Tp = T + back, pi = ti + back, pj = tj + back;
  const unsigned int* td = tdr ? tdr + PDF_TD0 + pi * Tp - pi * (pi - 1) / 2 + (pj - pi) : nullptr;
  return pdf_wait3(c.prev + PDF_PR0 + ti, chain_pr_need(S, ti), ti == tj ? nullptr : c.prev + PDF_PR0 + tj, chain_pr_need(S, tj), td, 1u, s_ok,
                   threadIdx.x);
}

// the work of logical workgroup t of one step
__device__ __forceinline__ void step_body(const StepArgs& a, int t, const ChainLink& c, double* sm, int* s_ok) {
  const int r0 = a.p0 + a.kp, m = a.n - r0;
  const int T = (m + 127) >> 7;  // tile rows = tile columns of the trailing matrix
  const bool next = a.kb_next > 0;
  // rows of the next panel (tile rows 0 and 1) go as 64x64 sub-tiles, strip by strip: strip sj has min(sj, 3) + 1 of them;
  // those of strips 0..3 as four 32x32 quadrant workgroups each (they gate the diagonal workgroups)
  const int S = (m + 63) >> 6;
  const int nHead = next ? 4 * step_head_units(S) : 0;
  const int nTA = next ? step_ta_workgroups(S) : 0;
  const int nTArows = next ? (T >= 2 ? 2 * T - 1 : T) : 0;  // 128x128 tiles they replace
  const int nd = next ? (a.kb_next >> 6) : 0;
  const int nTiles = T * (T + 1) / 2;
  if (t < nHead) {
    const int u = t >> 2;
    const int sj = (u >= 6) ? 3 : ((u >= 3) ? 2 : (u >= 1 ? 1 : 0));
    const int si = u - sj * (sj + 1) / 2;
    __builtin_amdgcn_s_setprio(2);
    const bool ok = chain_wait_tile(c, T, S, si >> 1, sj >> 1, s_ok);
    CHAIN_PROF_WAITED();
    if (!ok && threadIdx.x == 0) atomicExch(a.status + 1, 1 + a.front_id);  // hand-off timed out: a fault, reported apart from pivot failures
    syrk_quadrant32(a.A, a.ld, a.n, a.p0, a.kp, r0, si, sj, t & 3, a.S);
    pdf_publish(&a.flags[PDF_TA0 + sj], threadIdx.x == 0);
    CHAIN_PROF_END(0);
    return;
  }
  if (t < nTA) {  // the rest of the next panel's rows: tile rows 0 and 1, tile columns 2 ..
    const int u = t - nHead, ti = u & 1, tj = 2 + (u >> 1);
    const bool ok = chain_wait_tile(c, T, S, ti, tj, s_ok);
    CHAIN_PROF_WAITED();
    if (!ok && threadIdx.x == 0) atomicExch(a.status + 1, 1 + a.front_id);
    syrk_tile<true>(a.A, a.ld, a.n, a.p0, a.kp, r0, a.n, ti, tj, sm, a.S);
    pdf_publish2(&a.flags[PDF_TA0 + 2 * tj], 2 * tj + 1 < S ? &a.flags[PDF_TA0 + 2 * tj + 1] : nullptr, threadIdx.x == 0);
    CHAIN_PROF_END(1);
    return;
  }
  t -= nTA;
  if (t < nd) {
    panel_role(a.A, a.ld, a.n, a.nf, r0, a.kb_next, t, a.front_id, a.status, a.inv16, a.flags, sm, s_ok, true);
    CHAIN_PROF_END(2);
    return;
  }
  t -= nd;
  if (t < nTiles - nTArows) {  // remaining tiles, row-major over tile rows first..T-1
    int ti = next ? 2 : 0;
    int rem = t;
    while (rem >= T - ti) {
      rem -= T - ti;
      ti++;
    }
    const int tj = ti + rem;
    const bool ok = chain_wait_tile(c, T, S, ti, tj, s_ok);
    CHAIN_PROF_WAITED();
    if (!ok && threadIdx.x == 0) atomicExch(a.status + 1, 1 + a.front_id);  // hand-off timed out: a fault, reported apart from pivot failures
    if (c.merged)
      syrk_tile(a.A, a.ld, a.n, a.p0 - 256, a.kp + 256, r0, a.n, ti, tj, sm);  // panels i - 1 and i are consecutive rows: one operand slab of depth 512
    else
      syrk_tile(a.A, a.ld, a.n, a.p0, a.kp, r0, a.n, ti, tj, sm);
    if (c.publish) pdf_publish(&a.flags[PDF_TD0 + ti * T - ti * (ti - 1) / 2 + (tj - ti)], threadIdx.x == 0);
    CHAIN_PROF_END(3);
    return;
  }
  t -= nTiles - nTArows;
  panel_role(a.A, a.ld, a.n, a.nf, r0, a.kb_next, nd + t, a.front_id, a.status, a.inv16, a.flags, sm, s_ok, true, c.publish);
  CHAIN_PROF_END(4);
}

// a trailing update of a few tiles (the tail of a front) as 32 x 32 quadrants: grid (4 x strips, strips)
__global__ __launch_bounds__(256) void syrk_quadrants_kernel(double* __restrict__ A, int ld, int n, int p0, int kp, int r0) {
  const int sj = blockIdx.x >> 2, si = blockIdx.y;
  if (si > sj) return;
  syrk_quadrant32(A, ld, n, p0, kp, r0, si, sj, blockIdx.x & 3, nullptr);
}

__global__ __launch_bounds__(256, 2) void step_kernel(StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ int s_bid, s_ok;
  if (threadIdx.x == 0) s_bid = (int)atomicAdd(&a.flags[0], 1u);
  __syncthreads();
  step_body(a, s_bid, ChainLink{nullptr, false}, sm, &s_ok);
}

// ---------------------------------------------------------------- a run of consecutive steps as ONE launch
// Steps i0 .. i0 + nsteps - 1 of a front (every one with a full 256-row finished panel and a next panel of whole 64-blocks).
// Ticket k runs task tasks[k] = (step, logical workgroup of that step); a workgroup of step s + 1 additionally waits for what a
// launch boundary used to guarantee: the 128-column blocks of panel s + 1 it multiplies with (counted by the row-panel
// workgroups of step s) and the tile of step s that holds the entries it updates.  The task list (chain_schedule below,
// built once per front on the host) is a topological order of those dependencies, and a ticket is drawn when a workgroup
// starts, so a waiting workgroup only ever waits for workgroups that are resident or done.
// What the order buys.  Right-looking step by step, the early steps have 2000+ update tiles each and the late steps almost
// none, so the late steps run at the speed of their dependency chain (four tile stages + a head tile, ~85 us per 256 columns)
// with the matrix cores idle.  The schedule instead gives every step about the same number of update tiles: tiles of the
// early steps that lie far from the diagonal are deferred (earliest deadline first: a tile row must be complete when it
// becomes the next panel) and fill the late steps; the chain then runs under the update work from the first panel to the last.
#define CHAIN_TASK_MERGED (1 << 30)  // in tasks[k].x: update tile of step s that also applies the panel of step s - 1 (which then has no task for this tile)
struct ChainArgs {
  double* A;
  int ld, n, nf;
  int i0, nsteps;
  int front_id;
  int* status;
  double* inv16;        // 16 x 256 doubles per panel: panel p at inv16 + p * 4096
  unsigned int* flags;  // region of panel p at flags + p * PDF_FLAG_WORDS; the ticket counter is word 0 of panel i0 + 1
  const int2* tasks;    // (step - i0, logical workgroup of that step) per ticket
  const double* S;      // multi-rank: the partial-assembly buffer (summed over the ranks for every row chunk this launch touches: the host
                        // waits for those chunks' events before the launch) whose rows the head tiles fold in; else nullptr
};

__global__ __launch_bounds__(256, 2) void chain_kernel(ChainArgs ca) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ int s_bid, s_ok;
  CHAIN_PROF_BEGIN();
  if (threadIdx.x == 0) s_bid = (int)atomicAdd(&ca.flags[(size_t)(ca.i0 + 1) * PDF_FLAG_WORDS], 1u);
  __syncthreads();
  const int2 task = ca.tasks[s_bid];
  const bool merged = (task.x & CHAIN_TASK_MERGED) != 0;
  const int s = task.x & ~CHAIN_TASK_MERGED, i = ca.i0 + s;
  const int kbn = min(ca.nf, (i + 2) * 256) - (i + 1) * 256;
  StepArgs a{ca.A, ca.ld, ca.n, ca.nf, 256 * i, 256, kbn, ca.front_id, ca.status, ca.inv16 + (size_t)(i + 1) * 4096,
             ca.flags + (size_t)(i + 1) * PDF_FLAG_WORDS, ca.S};
  const ChainLink c{s > 0 ? ca.flags + (size_t)i * PDF_FLAG_WORDS : nullptr, s + 1 < ca.nsteps, merged,
                    (merged && s > 1) ? ca.flags + (size_t)(i - 1) * PDF_FLAG_WORDS : nullptr};
  step_body(a, task.y, c, sm, &s_ok);
}

}  // namespace lmgpu

// ---- host side: the ticket order of a chained launch
#include <algorithm>
#include <vector>
namespace lmgpu {

// n, nf: front size and frontal rows; steps i0 .. i0 + nsteps - 1.  Per step s the logical workgroups are laid out as in
// step_body: [head tiles][diagonal workgroups][update tiles, tile rows 2.. row-major][row-panel workgroups].
// Tile rows are numbered chain-absolutely: tile row ti of step s is R = 2 s + ti; as an update-tile row (ti >= 2) it is touched
// by the steps 0 .. last(R) = (R - 2) / 2 and must be complete at the end of block last(R), because the head tiles of the
// next step read it.  Block s of the list:
//   head tiles + diagonal workgroups of step s,
//   FAR rows (R >= far_row), first unit           -- a unit = the tiles of one step in one tile row
//   NEAR rows, their unit of step s (right-looking: always up to date),
//   FAR rows, second unit,
//   row-panel workgroups of step s.
// A far row idles through the first half of its life and then takes two units per block ("as late as two per block allows":
// completed(R, s) >= last(R) + 1 - 2 (last(R) - s)): the bottom rows of the matrix, which the chain reaches last, keep
// update work for the late steps, whose own tiles no longer fill the machine.  The two units of a row in one block depend on
// each other and are placed at the two ends of the block's tile list so that the first is (almost) done when the second is
// dispatched.  Order of dependencies: a unit (s', R) is listed after (s' - 1, R) and in a block s >= s', i.e. after the
// row-panel workgroups of step s' - 1; head tiles of step s + 1 come after every unit of the rows they read (mandatory below).
// far_pct = 100: no far rows, plain step order.
//
// merge = true (round 3, the default): update tiles go as PAIRS of steps in one pass of depth 512 (CHAIN_TASK_MERGED).  Stand-alone the
// tile is bound by what it does per pass, not by the matrix instruction (tools/syrk4_bench.hip: 61-69 TFLOP/s with neither operand
// fetch nor the read-modify-write of C, 43-44 with both at K = 256, 52-56 at K = 512).  A tile row R lives through the steps
// 0 .. L = last(R); its units are paired from the END -- (L-1, L), (L-3, L-2), ... and a single unit of step 0 if L + 1 is odd -- so
// that the last pair is exactly what block L must deliver (rows 2 L + 2, 2 L + 3 are the head rows of step L + 1).  Block s:
//   head tiles + diagonal workgroups of step s
//   X1 = the first split_pct percent of the block's update tasks, starting with the rows whose last step is s (their final pair: the
//        next step's head tiles read them)
//   row-panel workgroups of step s -- behind enough update work that the diagonal chain (~100-160 us) is mostly done when they are
//        drawn: drawn right behind the diagonal workgroups they hold ~130 slots spinning on it (in-kernel stamps: 11 % of all slot-time)
//   X2 = the rest of the update tasks: what runs while the row-panel workgroups finish panel s + 1, so that the head tiles of step
//        s + 1 find it complete
// The update tasks of a block: the pairs (s-1, s) of the aligned near rows (row R is aligned in block s when s = L mod 2), every other
// one postponed to the next block, and the pairs a deferred FAR row owes.  Two tasks of one tile are always about a block apart in the
// list (a pair takes ~120 us: a dependent task right behind it would hold its slot spinning that long).  A far row idles until it has
// to deliver one pair per block to finish in block L.
inline std::vector<int2> chain_schedule(int n, int nf, int i0, int nsteps, int far_pct = 100, bool merge = false, int split_pct = 60) {
  struct StepGeo { int T, S, nTA, nd, nTB, ntrsm; };
  std::vector<StepGeo> g(nsteps);
  for (int s = 0; s < nsteps; s++) {
    const int i = i0 + s, m = n - (i + 1) * 256, kbn = std::min(nf, (i + 2) * 256) - (i + 1) * 256;
    StepGeo& G = g[s];
    G.T = (m + 127) / 128;
    G.S = (m + 63) / 64;
    G.nTA = step_ta_workgroups(G.S);
    G.nd = kbn / 64;
    G.nTB = G.T * (G.T + 1) / 2 - (G.T >= 2 ? 2 * G.T - 1 : G.T);
    G.ntrsm = (m - kbn + 63) / 64;
  }
  const int T0 = g[0].T;
  const int far_row = std::max(4, (int)((long)T0 * far_pct / 100));
  std::vector<int> next_step(T0 + 2, 0);  // per tile row: the next step whose unit is to be listed
  std::vector<int2> tasks;
  auto last_of = [&](int R) { return std::min(nsteps - 1, (R - 2) / 2); };
  auto emit_tiles = [&](int R, int sp, bool mg) {  // the tiles of tile row R as tasks of step sp (mg: applying the panel of step sp - 1 as well)
    const int ti = R - 2 * sp, T = g[sp].T;
    int off = g[sp].nTA + g[sp].nd;
    for (int r = 2; r < ti; r++) off += T - r;
    for (int tj = ti; tj < T; tj++) tasks.push_back(int2{sp | (mg ? CHAIN_TASK_MERGED : 0), off + (tj - ti)});
  };
  if (merge) {
    // the next group of row R: a single unit when an odd number of units remains (only ever the first), else a pair
    auto group_len = [&](int R) { return ((last_of(R) + 1 - next_step[R]) & 1) ? 1 : 2; };
    auto emit_group = [&](int R) {
      const int len = group_len(R), a = next_step[R];
      emit_tiles(R, a + len - 1, len == 2);
      next_step[R] += len;
    };
    // the group's panels are all <= `panel_max`
    auto available = [&](int R, int panel_max) { return next_step[R] <= last_of(R) && next_step[R] + group_len(R) - 1 <= panel_max; };
    auto groups_left = [&](int R) { return (last_of(R) + 1 - next_step[R] + 1) / 2; };
    std::vector<char> postponed(T0 + 2, 0);
    std::vector<int2> head, upd;
    for (int s = 0; s < nsteps; s++) {
      // the block's update tasks into `upd`: C first (the next step's head tiles read those rows), then A and B interleaved row by row
      std::swap(head, tasks);  // (emit_* append to `tasks`)
      tasks.clear();
      for (int R = 2 * s + 2; R < T0; R++)
        if (last_of(R) == s)
          while (next_step[R] <= s) emit_group(R);
      int alt = 0;
      for (int R = 2 * s + 4; R < T0; R++) {
        if (last_of(R) <= s) continue;
        if (R >= far_row && last_of(R) - s >= 1) {
          if (available(R, s - 1) && groups_left(R) >= last_of(R) - s + 1) emit_group(R);  // a far row on its final run: one group per block
          continue;
        }
        if (postponed[R]) {
          if (available(R, s - 1)) emit_group(R);
          postponed[R] = 0;
        } else if (available(R, s) && next_step[R] + group_len(R) - 1 == s) {
          if ((alt++ & 1) && last_of(R) > s + 1)
            postponed[R] = 1;  // every other aligned near row delivers this group in the next block: two blocks' worth of rows to draw from
          else
            emit_group(R);
        }
      }
      std::swap(upd, tasks);
      std::swap(head, tasks);
      for (int t = 0; t < g[s].nTA + g[s].nd; t++) tasks.push_back(int2{s, t});
      const size_t cut = upd.size() * (size_t)split_pct / 100;
      tasks.insert(tasks.end(), upd.begin(), upd.begin() + cut);
      const int first_trsm = g[s].nTA + g[s].nd + g[s].nTB;
      for (int t = 0; t < g[s].ntrsm; t++) tasks.push_back(int2{s, first_trsm + t});
      tasks.insert(tasks.end(), upd.begin() + cut, upd.end());
      upd.clear();
    }
    return tasks;
  }
  auto emit_unit = [&](int R) { emit_tiles(R, next_step[R]++, false); };  // the tiles of step next_step[R] in tile row R
  auto pending = [&](int R, int s) { return next_step[R] <= std::min(s, last_of(R)); };  // a unit of row R that block s may list
  auto far_need = [&](int R, int s) { return last_of(R) + 1 - 2 * (last_of(R) - s); };   // units a far row must have after block s
  for (int s = 0; s < nsteps; s++) {
    for (int t = 0; t < g[s].nTA + g[s].nd; t++) tasks.push_back(int2{s, t});
    const bool last_block = s + 1 == nsteps;
    for (int R = std::max(far_row, 2 * s + 2); R < T0; R++)
      if (pending(R, s) && next_step[R] < far_need(R, s)) emit_unit(R);
    for (int R = 2 * s + 2; R < std::min(far_row, T0); R++)
      while (pending(R, s)) emit_unit(R);
    for (int R = std::max(far_row, 2 * s + 2); R < T0; R++)
      while (pending(R, s) && (next_step[R] < far_need(R, s) || R <= 2 * s + 3 || last_block)) emit_unit(R);
    const int first_trsm = g[s].nTA + g[s].nd + g[s].nTB;
    for (int t = 0; t < g[s].ntrsm; t++) tasks.push_back(int2{s, first_trsm + t});
  }
  return tasks;
}

}  // namespace lmgpu
