// LDS fronts: one workgroup assembles, partially factors and emits one small front of the multifrontal
// Cholesky.  Restates per clique what EliminateCholesky does on the CPU:
//   a11 Scatter + HessianFactor(gfg, scatter)    gtsam/linear/HessianFactor.cpp:240-253, Scatter.cpp:39-73
//   a12 updateHessian ([A b]^T [A b] into the upper triangle; child separator Hessians added block-wise)
//                                                gtsam/linear/JacobianFactor.cpp:586-624, BinaryJacobianFactor.h:51-83,
//                                                HessianFactor.cpp:349-373
//   a1  damping prior lambda*I (or lambda*diag)  gtsam/nonlinear/internal/LevenbergMarquardtState.h:125-156
//   a13 choleskyPartial + split                  gtsam/base/cholesky.cpp:108-159, SymmetricBlockMatrix.cpp:83-107
//   a14 back-substitution per clique             gtsam/linear/linearAlgorithms-inst.h:54-116
// Storage: the front is ROW-major upper (row k of R is contiguous); [R S d] is written nf x n row-major.
// The update (separator Hessian) either goes to HBM ((n-nf)^2 row-major upper, ld = n-nf) for the parent to gather,
// or — when the parent is an HBM front — is scattered straight into the parent with FP64 atomics (no 33-KB
// round trip per BAL point).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_factors.hpp"

namespace lmgpu {

struct FrontFac {
  int32_t fac;         // index into FacDesc[]
  int32_t c0, c1, c2;  // column offsets of the factor's variables inside the front (c2: third variable of a ternary factor)
};
struct ChildRef {
  int64_t u_off;      // pool offset of the child's update matrix
  int32_t ld, m;      // leading dimension and size (child's ns + 1)
  int32_t map_begin;  // into cmap[]: child update index -> parent front column
  int32_t pad;
};
struct FrontDesc {
  int32_t n, nf;
  int32_t fac_begin, fac_count;
  int32_t child_begin, child_count;
  int32_t fx_begin;  // into fxoff[]: delta offset of each frontal scalar (nf entries)
  int32_t sx_begin;  // into sxoff[]: delta offset of each separator scalar (n-nf-1 entries)
  int64_t rsd_off;   // [R S d], row-major nf x ld_rsd
  int64_t u_off;     // update matrix, row-major (n-nf) x ld_u, upper (a gather leaf: its transposed [S d] instead, see par_ld)
  int32_t ld_rsd, ld_u;
  int32_t id;        // front index (failure report)
  int32_t pad;       // bit 0: replicated over ranks (all-reduce after assembly); bit 1: this rank skips own factors + damping
  int64_t par_off;   // (ISAM2's device tree: pool offset of a wide clique's delta offsets)
  int32_t par_ld;    // -1: gather leaf -- the parent assembles this front's update itself from [R S d] (kernels_schur.hpp); 0 otherwise
  int32_t par_map;   // into cmap[]: this front's update index -> parent column
};

__device__ __forceinline__ double readlane_dyn(double v, int src /* wave-uniform */) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ int frexp_exp(double x) {
  int e;
  (void)frexp(x, &e);
  return e;
}

// r = sqrt(piv) and inv = 1 / r for a pivot of the LDS Cholesky (piv > 0, normal): v_rsq_f64 + two Newton steps + one correction of r, a
// dozen dependent fused multiply-adds instead of the IEEE sqrt and divide sequences (~600 cycles per pivot, and the pivots of a clique are a
// chain: 48 of them were half of what VisualISAM2Example's root clique costs).  r within an ulp of sqrt(piv); inv = 1 / r to the same.
__device__ __forceinline__ void pivot_sqrt_inv(double piv, double* r_out, double* inv_out) {
  double y = __builtin_amdgcn_rsq(piv);
  const double h = 0.5 * piv;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  double r = piv * y;
  r = fma(0.5 * y, fma(-r, r, piv), r);
  // (inv against the corrected r: one more step of 1 / r)
  y = fma(y, fma(-r, y, 1.0), y);
  *r_out = r;
  *inv_out = y;
}

// staged descriptor of one own factor of the front being assembled
struct LFac {
  int64_t joff;
  int32_t c0, c1;
  int16_t rows, d0, d1, d2;
  int32_t c2;
  int16_t off, sz;  // staging offset (< LDSF_JCAP) and size (rows x columns, < 100) in doubles
};
// local column q of a factor's [A1 A2 A3 b] (q <= d0 + d1 + d2; the last one is b) -> column of the front with n columns
template <typename FD>
__device__ __forceinline__ int fac_col(const FD& d, int c0, int c1, int c2, int q, int n) {
  return q < d.d0 ? c0 + q : (q < d.d0 + d.d1 ? c1 + (q - d.d0) : (q < d.d0 + d.d1 + d.d2 ? c2 + (q - d.d0 - d.d1) : n - 1));
}

// grid: one block per front in `list`; dynamic LDS = srows*nmax (+8) doubles (the front) + LDSF_JCAP doubles (staged
// Jacobians) + LDSF_MAXB staged factor descriptors + 4 ints.  Lanes run along the contiguous (column) index of the
// row-major front, waves along rows.  GATHER fronts (leaves whose HBM parent gathers their update itself,
// kernels_schur.hpp) only keep their nf frontal rows and the (rhs, rhs) corner: srows = max nf of the launch, which
// lets ~7 workgroups share a CU instead of 2.
// waves a workgroup needs for the blocked Cholesky (groups of four / eight pivots, trailing update on the matrix core): any number -- with
// 4 (rounds 2-3) the one- and two-wave launches of small fronts took their pivots one by one (city10000 2.63 -> 2.58 ms, victoria_park
// 6.92 -> 6.78)
#ifndef LDSF_BLOCKED_MIN_WAVES
#define LDSF_BLOCKED_MIN_WAVES 1
#endif
#define LDSF_JCAP 704
#define LDSF_MAXB 32
// Packed descriptor of an LDS front (one record per workgroup of a launch, `stride` bytes apart): everything the workgroup
// otherwise collects through list -> front -> front-factor -> factor descriptor (four dependent reads) and
// front -> fxoff -> damping weight (two more), laid out by the host once:
//   [0]   FrontDesc
//   [80]  int32 nstage (0: take the general path), tot (doubles of Jacobians), contig, has_xo
//   [96]  int32 xo[8]: delta offsets of the frontal scalars (has_xo: nf <= 8)
//   [128] LFac[nstage] with their LDS offsets filled in
#define LEAFPACK_HDR 80
#define LEAFPACK_XO 96
#define LEAFPACK_FAC 128
#define LEAFPACK_MAXNF 8
static_assert(sizeof(FrontDesc) == LEAFPACK_HDR, "leaf records embed a FrontDesc");
static_assert(sizeof(LFac) == 32, "leaf records embed LFac entries");
#define LDSF_EXTRA_BYTES (LDSF_JCAP * 8 + LDSF_MAXB * 32 + 16)
// development aid (tools/ldsf_phases.py builds a library of its own with it): 100 MHz wall-clock time per phase of the first workgroup of
// every launch of at most eight fronts (= the upper levels of a clique tree, where a launch lasts as long as its slowest front)
#ifdef LDSF_STAMPS
// which workgroups are sampled: the first one of launches of at most eight fronts (upper tree levels) -- or, with LDSF_STAMPS_LEAVES, one
// in the middle of every launch of more than 1000 fronts (a leaf level under full load)
#ifdef LDSF_STAMPS_LEAVES
#define LDSF_SAMPLED (gridDim.x > 1000 && blockIdx.x == gridDim.x / 2)
#else
#define LDSF_SAMPLED (DATAFLOW ? ((work - flow.seg_begin) * 4 > 3 * (flow.seg_end - flow.seg_begin)) : (gridDim.x <= 8 && blockIdx.x == 0))
#endif
__device__ unsigned long long ldsf_dbg[16];
#define LDSF_STAMP(i)                                                  \
  do {                                                                 \
    if (LDSF_SAMPLED && threadIdx.x == 0) {                            \
      const unsigned long long now_ = wall_clock64();                  \
      atomicAdd(&ldsf_dbg[i], now_ - ldsf_last);                       \
      ldsf_last = now_;                                                \
    }                                                                  \
  } while (0)
#else
#define LDSF_STAMP(i) do { } while (0)
#endif
// A front of at most sixteen columns as ONE wave with the whole front in registers (the 16 x 16 accumulator layout of
// v_mfma_f64_16x16x4_f64: entry (kk + 4 r, cc) in component r of lane 16 kk + cc): own factors as J^T J on the matrix core, the children's
// update matrices fetched by the lanes that own their destinations, the pivots by lane shuffles -- no LDS, no barrier.  The general
// body spends ~7.5 us per tree level on a chain of such fronts (victoria_park's upper 112 levels, a fixed-lag window, the cliques of an
// incremental pose-graph update): four waves in lock step through a dozen barriers for a 10 x 10 matrix.  Merged launches only
// (POLL: the children's entries are awaited by value, see lds_front_body).  The symmetric matrix is kept in full.
template <bool POLL>
__device__ __forceinline__ void lds_front_tiny(const FrontDesc& F, const FrontFac* __restrict__ ffac, const FacDesc* __restrict__ fd,
                                               const ChildRef* __restrict__ childs, const int32_t* __restrict__ cmap, const int32_t* __restrict__ fxoff,
                                               double* __restrict__ pool, double lambda, const double* __restrict__ dampw, int* __restrict__ status,
                                               const double* __restrict__ gex) {
  typedef double d4_t __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, kk = lane >> 4, cc = lane & 15;
  const int n = F.n, nf = F.nf;
  d4_t S = d4_t{0, 0, 0, 0};
  // ---- own factors: S += [A b]^T [A b], four rows of a factor per matrix instruction
  for (int k = 0; k < F.fac_count; k++) {
    const FrontFac ff = ffac[F.fac_begin + k];
    const FacDesc d = fd[ff.fac];
    const int m = d.rows, nc = d.d0 + d.d1 + d.d2 + 1;
    int q = -1;  // the factor's local column that lands on front column cc
    if (cc >= ff.c0 && cc < ff.c0 + d.d0)
      q = cc - ff.c0;
    else if (d.d1 > 0 && cc >= ff.c1 && cc < ff.c1 + d.d1)
      q = d.d0 + cc - ff.c1;
    else if (d.d2 > 0 && cc >= ff.c2 && cc < ff.c2 + d.d2)
      q = d.d0 + d.d1 + cc - ff.c2;
    else if (cc == n - 1)
      q = nc - 1;
    const double* J = pool + d.joff;
    for (int r0 = 0; r0 < m; r0 += 4) {
      const int r = r0 + kk;
      const double v = (q >= 0 && r < m) ? J[q * m + r] : 0.0;
      S = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, S, 0, 0, 0);
    }
  }
  // damping and the extra gradient term, by the lanes that own the entries
  {
    const int xo = fxoff[F.fx_begin + min(cc, nf - 1)];
    const double dw = lambda * dampw[xo];
#pragma unroll
    for (int rr = 0; rr < 4; rr++)
      if (kk + 4 * rr == cc && cc < nf) S[rr] += dw;
    if (gex && cc == n - 1) {
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const int I = kk + 4 * rr;
        if (I < nf) {
          const double gv = gex[fxoff[F.fx_begin + I]];
          S[rr] += gv;
        }
      }
    }
    if (gex && cc < nf) {  // (and its mirror: the matrix is kept in full)
      const double gv = gex[xo];
#pragma unroll
      for (int rr = 0; rr < 4; rr++)
        if (kk + 4 * rr == n - 1) S[rr] += gv;
    }
  }
  // ---- children: the lane that owns (I, J) fetches the child's entry that lands there
  bool timeout = false;
  for (int k = 0; k < F.child_count; k++) {
    const ChildRef c = childs[F.child_begin + k];
    const double* U = pool + c.u_off;
    const int mt = cmap[c.map_begin + min(lane, c.m - 1)];  // lane t: where the child's index t lands
    int iJ = -1, iI[4] = {-1, -1, -1, -1};
    for (int t = 0; t < c.m; t++) {
      const int dst = __builtin_amdgcn_readlane(mt, t);
      if (dst == cc) iJ = t;
#pragma unroll
      for (int rr = 0; rr < 4; rr++)
        if (dst == kk + 4 * rr) iI[rr] = t;
    }
    double u[4] = {0.0, 0.0, 0.0, 0.0};
    long spins = 0;
    bool again;
    do {
      again = false;
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const bool valid = iJ >= 0 && iI[rr] >= 0;
        const int a = valid ? min(iI[rr], iJ) : 0, b = valid ? max(iI[rr], iJ) : 0;
        if (POLL) {
          const unsigned long long bits = __hip_atomic_load((const unsigned long long*)(U + (size_t)a * c.ld + b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          u[rr] = __longlong_as_double((long long)bits);
          if (valid && bits == ~0ull) again = true;
        } else {
          u[rr] = U[(size_t)a * c.ld + b];
        }
        if (!valid) u[rr] = 0.0;
      }
      if (again) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 2000000L) {
          timeout = true;
          again = false;
        }
      }
    } while (again);
#pragma unroll
    for (int rr = 0; rr < 4; rr++) S[rr] += u[rr];
  }
  if (timeout) atomicExch(status + 1, 1 + F.id);  // never expected: spin bound hit (reported apart from pivot failures)
  // ---- partial Cholesky: row k lives in component k >> 2 of lanes 16 (k & 3) .. + 15
  bool failed = false;
  double dlast = 1.0, dprev = 1.0;  // the last two pivots' R_kk (pivot-exponent test)
  for (int k = 0; k < nf; k++) {
    const int pk = k >> 2, lk = (k & 3) << 4;
    const double rowk = pk == 0 ? S[0] : (pk == 1 ? S[1] : (pk == 2 ? S[2] : S[3]));
    double piv = __shfl(rowk, lk + k);
    if (!(piv > 0.0)) {
      if (piv <= 0.0) failed = true;  // Eigen LLT: pivot <= 0 -> NumericalIssue (NaN passes, like Eigen)
      piv = (piv == piv && piv != 0.0) ? fabs(piv) : 1.0;
    }
    double r, inv;
    pivot_sqrt_inv(piv, &r, &inv);
    dprev = dlast;
    dlast = r;
    const double Rkj = __shfl(rowk, lk + cc) * inv;  // R[k][cc]
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      const int I = kk + 4 * rr;
      const double RkI = __shfl(rowk, lk + I) * inv;  // R[k][I]
      if (I > k && cc > k) S[rr] -= RkI * Rkj;
      if (I == k) S[rr] = cc > k ? Rkj : (cc == k ? r : 0.0);
    }
  }
  if (lane == 0) {
    // pivot-exponent test, gtsam/base/cholesky.cpp:146-158
    if (nf >= 2) {
      if (!(frexp_exp(dprev) - frexp_exp(dlast) < 12)) failed = true;
    } else if (nf == 1) {
      if (!(frexp_exp(dlast) > -12)) failed = true;
    }
  }
  if (failed) atomicMin(status, F.id);
  // ---- emit [R S d] (strictly-lower zeroed) and the update matrix (upper)
  double* RSd = pool + F.rsd_off;
  double* Uo = pool + F.u_off;
#pragma unroll
  for (int rr = 0; rr < 4; rr++) {
    const int I = kk + 4 * rr;
    if (I < nf && cc < n) RSd[(size_t)I * F.ld_rsd + cc] = cc >= I ? S[rr] : 0.0;
    if (I >= nf && I < n && cc >= I && cc < n) {
      if (POLL)
        __hip_atomic_store((unsigned long long*)(Uo + (size_t)(I - nf) * F.ld_u + (cc - nf)), (unsigned long long)__double_as_longlong(S[rr]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      else
        Uo[(size_t)(I - nf) * F.ld_u + (cc - nf)] = S[rr];
    }
  }
}

// DATAFLOW (lds_front_merged_kernel below): the fronts of several consecutive tree levels in one launch; a front's extend-add polls the
// values of its children's update matrices (see there).
struct FrontFlow {
  int seg_begin, seg_end;  // positions of this launch's fronts in the level lists
};
template <bool GATHER, int MAXT, bool DATAFLOW, bool EIGHT = (MAXT > 256) || DATAFLOW>
__device__ __forceinline__ void lds_front_body(const int work, const int32_t* __restrict__ list, const FrontDesc* __restrict__ fronts,
                                               const FrontFac* __restrict__ ffac, const FacDesc* __restrict__ fd,
                                               const ChildRef* __restrict__ childs, const int32_t* __restrict__ cmap,
                                               const int32_t* __restrict__ fxoff, double* __restrict__ pool, double lambda_v, const double* __restrict__ lambda_p,
                                               const double* __restrict__ dampw, int* __restrict__ status, int nmax, int srows,
                                               double* __restrict__ gcorner, int jcap, const double* __restrict__ gex,
                                               const char* __restrict__ pack, int pack_stride, const FrontFlow flow) {
  extern __shared__ double S[];
#ifdef LDSF_STAMPS
  unsigned long long ldsf_last = wall_clock64();
  if (LDSF_SAMPLED && threadIdx.x == 0) atomicAdd(&ldsf_dbg[15], 1ull);
#endif
  double* corner_g = S + (size_t)srows * nmax;  // GATHER: the (rhs, rhs) entry lives here
  double* Jb = corner_g + 8;
  LFac* LF = (LFac*)(Jb + jcap);  // jcap <= LDSF_JCAP doubles of staged Jacobians: the launch's largest front (fewer for small ones => more fronts per CU)
  int* meta = (int*)(LF + LDSF_MAXB);
  const char* pk = pack ? pack + (size_t)work * pack_stride : nullptr;
  const FrontDesc F = pk ? *(const FrontDesc*)pk : fronts[list[work]];
  const int pk_n = pk ? ((const int*)(pk + LEAFPACK_HDR))[0] : 0;  // > 0: factor descriptors, staging offsets and damping offsets come from the record
  const int pk_tot = pk_n ? ((const int*)(pk + LEAFPACK_HDR))[1] : 0, pk_contig = pk_n ? ((const int*)(pk + LEAFPACK_HDR))[2] : 0;
  const int n = F.n, nf = F.nf, tid = threadIdx.x, nt = blockDim.x;
  if constexpr (DATAFLOW && !GATHER) {
    if (n <= 16) {  // one wave, the front in registers (the other waves leave before any barrier of this body)
      if (tid < 64) lds_front_tiny<true>(F, ffac, fd, childs, cmap, fxoff, pool, lambda_p ? *lambda_p : lambda_v, dampw, status, gex);
      return;
    }
  }
  const bool pk_xo = pk_n && ((const int*)(pk + LEAFPACK_HDR))[3] != 0;  // the record also carries the frontal delta offsets (nf <= 8)
  double damp_pre = 0.0, gex_pre = 0.0;
  if (pk_xo && tid < nf) {
    const int xo = ((const int*)(pk + LEAFPACK_XO))[tid];
    damp_pre = dampw[xo];
    if (gex) gex_pre = gex[xo];
  }
  const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
  // Everything the later phases read from memory that does not depend on the children is requested NOW, under the own-factor phase (and, in
  // a merged launch, under the wait for the children): the damping weights (fxoff -> dampw: two dependent reads), lambda, and the
  // descriptors and column maps of the first two children (childs -> cmap: two more).  Behind the wait each of them was a memory round
  // trip of its own on the critical path of a tree level (~9 us per level of victoria_park's chain of small fronts before).
  const double lambda = lambda_p ? *lambda_p : lambda_v;
  const bool damp_regs = !pk_xo && nf <= nt;
  if (damp_regs && tid < nf) {
    const int xo = fxoff[F.fx_begin + tid];
    damp_pre = dampw[xo];
    if (gex) gex_pre = gex[xo];
  }
  ChildRef pc0{}, pc1{};
  int pm0 = 0, pm1 = 0;
  if constexpr (!GATHER) {
    if (F.child_count > 0) {
      pc0 = childs[F.child_begin];
      pm0 = cmap[pc0.map_begin + min(tid, pc0.m - 1)];
    }
    if (F.child_count > 1) {
      pc1 = childs[F.child_begin + 1];
      pm1 = cmap[pc1.map_begin + min(tid, pc1.m - 1)];
    }
  }
  // gather mode (par_ld < 0): the parent assembles this leaf's update itself from [R S d] (kernels_schur.hpp); only the
  // frontal rows and the (rhs, rhs) corner of the trailing block are needed here
  constexpr bool gather = GATHER;
  const int urows = gather ? nf : n;
  double* corner = gather ? corner_g : &S[(n - 1) * n + n - 1];
  for (int i = tid; i < urows * n; i += nt) S[i] = 0.0;
  if (gather && tid == 0) *corner = 0.0;
  // ---- own factors: S += [A b]^T [A b].  Descriptors and Jacobians are staged through LDS in batches: all the
  //      dependent HBM reads (list -> front -> factor -> Jacobian) of a batch are in flight together.
  for (int k0 = 0; k0 < F.fac_count;) {
    __syncthreads();
    int B, tot, contig;
    if (pk_n) {  // one batch, laid out by the host
      for (int b = tid; b < pk_n; b += nt) LF[b] = ((const LFac*)(pk + LEAFPACK_FAC))[b];
      B = pk_n;
      tot = pk_tot;
      contig = pk_contig;
      __syncthreads();
    } else {
    const int cand = min(LDSF_MAXB, F.fac_count - k0);
    for (int b = tid; b < cand; b += nt) {
      const FrontFac ff = ffac[F.fac_begin + k0 + b];
      const FacDesc d = fd[ff.fac];
      LFac l;
      l.joff = d.joff;
      l.c0 = ff.c0;
      l.c1 = ff.c1;
      l.c2 = ff.c2;
      l.rows = d.rows;
      l.d0 = d.d0;
      l.d1 = d.d1;
      l.d2 = d.d2;
      l.off = 0;
      l.sz = (int16_t)(d.rows * (d.d0 + d.d1 + d.d2 + 1));
      LF[b] = l;
    }
    __syncthreads();
    if (tid == 0) {
      int o = 0, b = 0;
      while (b < cand && o + LF[b].sz <= jcap) {
        LF[b].off = o;
        o += LF[b].sz;
        b++;
      }
      meta[0] = b;  // >= 1: a single factor is at most 90 doubles
      meta[1] = o;  // doubles staged in this batch
      bool contig = true;  // the batch's Jacobians back to back in the pool (factors of one variable added together)?
      for (int q = 1; q < b; q++) contig = contig && (LF[q].joff == LF[0].joff + LF[q].off);
      meta[2] = contig ? 1 : 0;
    }
    __syncthreads();
    B = meta[0];
    tot = meta[1];
    contig = meta[2];
    }
    if (contig) {  // one flat copy: every load independent of the others
      const double* J = pool + LF[0].joff;
      for (int i0 = tid; i0 < tot; i0 += 8 * nt) {  // eight loads in flight per thread, then their stores
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = (i0 + u * nt < tot) ? J[i0 + u * nt] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (i0 + u * nt < tot) Jb[i0 + u * nt] = v[u];
      }
    } else {
      for (int b = wave; b < B; b += nw) {
        const double* J = pool + LF[b].joff;
        const int o = LF[b].off, sz = LF[b].sz;
        for (int i = lane; i < sz; i += 64) Jb[o + i] = J[i];
      }
    }
    __syncthreads();
    for (int b = 0; b < B; b++) {
      const LFac d = LF[b];
      const double* J = Jb + d.off;
      const int m = d.rows, nc = d.d0 + d.d1 + d.d2 + 1;
      if (gather) {  // (gather leaves only carry factors of at most two variables: the host checks)
        // only the frontal rows are kept: pairs (p, q) with p a FRONTAL column of this factor and q any column (q frontal too:
        // once, gq >= gp), all of them in one pass of the workgroup (BAL: 3 x 13); the (b, b) corner by one lane
        const bool f0 = d.c0 < nf, f1 = d.d1 > 0 && d.c1 < nf;
        const int nfc = (f0 ? d.d0 : 0) + (f1 ? d.d1 : 0);
        for (int idx = tid; idx < nfc * nc; idx += nt) {
          const int pf = idx / nc, q = idx - pf * nc;
          const int p = (f0 && pf < d.d0) ? pf : (f0 ? pf : d.d0 + pf);  // f0: frontal columns start at 0; else they are var1's
          const int gp = (p < d.d0) ? d.c0 + p : d.c1 + (p - d.d0);
          const int gq = (q < d.d0) ? d.c0 + q : (q < d.d0 + d.d1 ? d.c1 + (q - d.d0) : n - 1);
          if (gq < nf && gq < gp) continue;  // both frontal: counted from the other side
          double v = 0;
          for (int r = 0; r < m; r++) v += J[p * m + r] * J[q * m + r];
          S[gp * n + gq] += v;
        }
        if (tid == 0) {
          double v = 0;
          for (int r = 0; r < m; r++) v += J[(nc - 1) * m + r] * J[(nc - 1) * m + r];
          *corner += v;
        }
        __syncthreads();
        continue;
      }
      // thread (p = tid / 16 + k nt/16, q = p + tid % 16 + 16 l): every pair p <= q exactly once, no two threads on one entry
      for (int p = tid >> 4; p < nc; p += (nt >> 4)) {
        const int gp = fac_col(d, d.c0, d.c1, d.c2, p, n);
        for (int q = p + (tid & 15); q < nc; q += 16) {
          double v = 0;
          for (int r = 0; r < m; r++) v += J[p * m + r] * J[q * m + r];
          const int gq = fac_col(d, d.c0, d.c1, d.c2, q, n);
          const int lo = gp < gq ? gp : gq, hi = gp < gq ? gq : gp;
          if (!gather || lo < nf)
            S[lo * n + hi] += v;
          else if (lo == n - 1)
            *corner += v;  // (b, b): exactly one lane per factor
        }
      }
      __syncthreads();
    }
    k0 += B;
  }
  __syncthreads();
  LDSF_STAMP(0);  // descriptors, clear, own factors
  bool flow_timeout = false;
  // ---- children: extend-add of their update matrices (row i of U contiguous: lanes along j).  The child's column map is staged
  //      in LDS once and four (sixteen-wave form: eight) rows per wave are fetched before the first is added -- row by row, every row was a memory round
  //      trip of its own (25-35 of them per child of ~100 columns: most of the 27-60 us an upper-level front took).  The loads are
  //      unconditional on clamped addresses (the lower triangle and the padding of U are finite: the pool is cleared once).
  if constexpr (!GATHER) {  // (gather leaves have no children; keeping the block out of that instantiation keeps its 41 VGPRs)
    constexpr int EAB = (MAXT > 256 || DATAFLOW) ? 8 : 4;  // rows per wave in flight (eight cost the four-wave form occupancy on leaf levels; a merged launch holds upper levels only)
    int* cm = (int*)Jb;  // jcap >= 96 doubles: room for 139 ints (a child's update matrix is at most as wide as this front)
    for (int k = 0; k < F.child_count; k++) {
      const ChildRef c = k == 0 ? pc0 : (k == 1 ? pc1 : childs[F.child_begin + k]);
      const double* U = pool + c.u_off;
      const int32_t* map = cmap + c.map_begin;
      if (k < 2 && c.m <= nt) {  // fetched at the top of the kernel
        if (tid < c.m) cm[tid] = k == 0 ? pm0 : pm1;
      } else {
        for (int i = tid; i < c.m; i += nt) cm[i] = map[i];
      }
      __syncthreads();
      int gjs[3];
#pragma unroll
      for (int q = 0; q < 3; q++) gjs[q] = cm[min(lane + 64 * q, c.m - 1)];
      for (int i0 = wave; i0 < c.m; i0 += EAB * nw) {
        double u[EAB][3];
        if constexpr (DATAFLOW) {
          // a child of the same launch publishes its update matrix VALUE BY VALUE (agent-scope stores over the all-ones pattern the host
          // filled its upper triangle with): the load of an entry is the wait for it -- no flag, no write-back, no invalidate on the path
          // from one tree level to the next.  (A child finished by an earlier launch simply has its values there.)
          long spins = 0;
          bool again;
          do {
            again = false;
#pragma unroll
            for (int r = 0; r < EAB; r++)
#pragma unroll
              for (int q = 0; q < 3; q++) {
                const int i = i0 + r * nw, j = lane + 64 * q;
                const unsigned long long bits = __hip_atomic_load((const unsigned long long*)(U + (size_t)min(i, c.m - 1) * c.ld + min(j, c.m - 1)),
                                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                u[r][q] = __longlong_as_double((long long)bits);
                if (bits == ~0ull && i < c.m && j >= i && j < c.m) again = true;
              }
            if (again) {
              __builtin_amdgcn_s_sleep(1);
              if (++spins > 2000000L) {
                flow_timeout = true;
                again = false;
              }
            }
          } while (again);
        } else {
#pragma unroll
          for (int r = 0; r < EAB; r++)
#pragma unroll
            for (int q = 0; q < 3; q++) u[r][q] = U[(size_t)min(i0 + r * nw, c.m - 1) * c.ld + min(lane + 64 * q, c.m - 1)];
        }
#pragma unroll
        for (int r = 0; r < EAB; r++) {
          const int i = i0 + r * nw;
          if (i < c.m) {
            const int gi = cm[i];
#pragma unroll
            for (int q = 0; q < 3; q++) {
              const int j = lane + 64 * q;
              if (j >= i && j < c.m) {
                const int gj = gjs[q];
                const int lo = gi < gj ? gi : gj, hi = gi < gj ? gj : gi;
                S[lo * n + hi] += u[r][q];
              }
            }
          }
        }
      }
      __syncthreads();
    }
  }
  if (DATAFLOW && flow_timeout) atomicExch(status + 1, 1 + F.id);  // never expected: spin bound hit (reported apart from pivot failures)
  LDSF_STAMP(1);  // extend-add of the children
  // ---- damping on the frontal diagonal
  // lambda_p != nullptr: the value lives in device memory so that a captured launch sequence can be replayed with a new one
  if (pk_xo || damp_regs) {  // the weights were fetched at the top (pk_xo: nf <= LEAFPACK_MAXNF <= nt, offsets from the record)
    if (tid < nf) {
      S[tid * n + tid] += lambda * damp_pre;
      if (gex) S[tid * n + n - 1] += gex_pre;
    }
  } else {
    for (int i = tid; i < nf; i += nt) {
      S[i * n + i] += lambda * dampw[fxoff[F.fx_begin + i]];
      if (gex) S[i * n + n - 1] += gex[fxoff[F.fx_begin + i]];  // extra gradient term eta += g (marginal covariances: unit vectors)
    }
  }
  LDSF_STAMP(2);  // damping
  // ---- partial Cholesky
  bool failed = false;
  // Fronts with many frontal columns (upper levels of general sparse graphs: n ~ 100, nf ~ 20-60) take the pivots FOUR at a
  // time: the four pivot rows are finished against each other (little work between the barriers), then the trailing matrix
  // receives one rank-4 update on the matrix cores, 16x16 tile by tile (v_mfma_f64_16x16x4_f64: operands and the tile from
  // LDS).  Row by row, every trailing entry is read-modify-written in LDS once per pivot -- ~0.75 us per pivot at n = 139 from
  // LDS bandwidth alone, 100-200 us per front, which is what a narrow tree level costs.  Same arithmetic up to the order of
  // the four subtractions.
  const bool blocked = !gather && nf >= 2 && nw >= (LDSF_BLOCKED_MIN_WAVES);  // (from two pivots on: one group of four is three barriers and one pass over the trailing
                                                         //  matrix on the matrix core, where pivot by pivot takes two barriers and a pass per pivot)
  // (smaller fronts: the four-pivot groups below -- every thread factors the 4 x 4 block itself; and only the launch forms of upper levels carry
  //  the eight-pivot code: its registers would cost the per-level launches of leaf levels their occupancy)
  constexpr bool kEightOk = EIGHT;
  const bool eight = kEightOk && nf >= 12;
  if constexpr (kEightOk)
  if (blocked && eight) {
    typedef double d4_t __attribute__((ext_vector_type(4)));
    const int kk = lane >> 4, cc = lane & 15;
    double* inv8 = corner_g;  // (eight doubles behind the front that only gather leaves use)
    // Pivots EIGHT at a time (round 3; four before): wave 0 factors the 8 x 8 diagonal block in registers -- a chain of eight dependent
    // sqrt / divide pairs, ~0.15 us each, which is what a group cannot go below -- and leaves it (in place) and the eight reciprocals in
    // LDS; every thread then solves its columns of the eight-row panel against it.  Per group: three barriers, as with four pivots -- a
    // clique of 72 frontal scalars (VisualISAM2Example's root) paid for eighteen groups at 2-3 us each.  Element by element the operations
    // and their order are the ones of the row-by-row form.
    // Two levels: inside a PANEL of sixteen pivots the first group's rank-8 update only reaches the panel's own remaining eight rows (one
    // row of tiles); the rows below the panel receive the whole panel as ONE rank-16 update (four MFMAs on a tile that is read and written
    // once) after its second group.
    for (int k0 = 0; k0 < nf; k0 += 8) {
      const int kb = min(8, nf - k0);
      __syncthreads();  // previous trailing update complete
      if (wave == 0) {
        double d[8][8];
#pragma unroll
        for (int q = 0; q < 8; q++)
#pragma unroll
          for (int c = q; c < 8; c++) d[q][c] = (c < kb) ? S[(k0 + q) * n + k0 + c] : ((q == c) ? 1.0 : 0.0);
        double inv[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
          inv[q] = 1.0;
          if (q < kb) {
            double piv = d[q][q];
            if (!(piv > 0.0)) {
              if (piv <= 0.0) failed = true;  // Eigen LLT: pivot <= 0 -> NumericalIssue (NaN passes, like Eigen)
              piv = (piv == piv && piv != 0.0) ? fabs(piv) : 1.0;
            }
            double r;
            pivot_sqrt_inv(piv, &r, &inv[q]);
            d[q][q] = r;
#pragma unroll
            for (int c = q + 1; c < 8; c++) d[q][c] *= inv[q];
#pragma unroll
            for (int i = q + 1; i < 8; i++)
#pragma unroll
              for (int c = i; c < 8; c++) d[i][c] -= d[q][i] * d[q][c];
          }
        }
        // lane (q, c) = (lane >> 3, lane & 7) puts its entry back; lanes 0 .. 7 the reciprocals
        {
          const int q = lane >> 3, c = lane & 7;
          double v = 0.0, iv = 0.0;
#pragma unroll
          for (int qq = 0; qq < 8; qq++)
#pragma unroll
            for (int c2 = 0; c2 < 8; c2++)
              if (qq == q && c2 == c && c2 >= qq) v = d[qq][c2];
#pragma unroll
          for (int qq = 0; qq < 8; qq++)
            if (qq == lane) iv = inv[qq];
          if (q <= c && c < kb) S[(k0 + q) * n + k0 + c] = v;
          if (lane < 8) inv8[lane] = iv;
        }
      }
      __syncthreads();  // the factored block and the reciprocals are in LDS
      LDSF_STAMP(7);  // (eight-pivot groups) diagonal block by wave 0
      if (k0 + kb + tid < n) {
        // the factored block and the reciprocals into registers FIRST (36 broadcast reads in flight together): read where they are used,
        // every step of the chain below waited for its own LDS round trip (2.2 us per group of eight pivots)
        double Rq[8][8], iv[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
          iv[q] = inv8[q];
#pragma unroll
          for (int i = q + 1; i < 8; i++) Rq[q][i] = S[(k0 + min(q, kb - 1)) * n + k0 + min(i, kb - 1)];
        }
        for (int j = k0 + kb + tid; j < n; j += nt) {
          double x[8];
#pragma unroll
          for (int q = 0; q < 8; q++) x[q] = (q < kb) ? S[(k0 + q) * n + j] : 0.0;
#pragma unroll
          for (int q = 0; q < 8; q++)
            if (q < kb) {
              x[q] *= iv[q];
#pragma unroll
              for (int i = q + 1; i < 8; i++)
                if (i < kb) x[i] -= Rq[q][i] * x[q];
            }
#pragma unroll
          for (int q = 0; q < 8; q++)
            if (q < kb) S[(k0 + q) * n + j] = x[q];
        }
      }
      __syncthreads();
      LDSF_STAMP(8);  // (eight-pivot groups) panel solve
      const int t0 = k0 + kb;
      const int p0 = k0 & ~15, pend = min(p0 + 16, nf);
      if (t0 < pend) {  // rows t0 .. pend - 1 of the panel, all columns from t0: rank-kb
        const int T = (n - t0 + 15) >> 4;
        for (int tj = wave; tj < T; tj += nw) {
          const int row0 = t0, col0 = t0 + 16 * tj;
          d4_t c;
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int row = row0 + kk + 4 * rr, col = col0 + cc;
            c[rr] = (row < pend && col < n) ? S[row * n + col] : 0.0;
          }
#pragma unroll
          for (int sx = 0; sx < 2; sx++) {
            const bool kv = 4 * sx + kk < kb;
            const double a = (kv && row0 + cc < pend) ? -S[(k0 + 4 * sx + kk) * n + row0 + cc] : 0.0;
            const double b = (kv && col0 + cc < n) ? S[(k0 + 4 * sx + kk) * n + col0 + cc] : 0.0;
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
          }
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int row = row0 + kk + 4 * rr, col = col0 + cc;
            if (row < pend && col < n && col >= row) S[row * n + col] = c[rr];
          }
        }
        LDSF_STAMP(9);  // (eight-pivot groups) rank-8 update inside the panel
          continue;
      }
      // the panel p0 .. pend - 1 is finished: everything below it
      const int m = n - pend;
      if (m <= 0) continue;
      const int K = pend - p0;
      const int T = (m + 15) >> 4, ntile = T * (T + 1) / 2;
      for (int t = wave; t < ntile; t += nw) {
        int ti = 0, rem = t;
        while (rem >= T - ti) {
          rem -= T - ti;
          ti++;
        }
        const int row0 = pend + 16 * ti, col0 = pend + 16 * (ti + rem);
        d4_t c;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const int row = row0 + kk + 4 * rr, col = col0 + cc;
          c[rr] = (row < n && col < n) ? S[row * n + col] : 0.0;
        }
#pragma unroll
        for (int sx = 0; sx < 4; sx++) {
          // A[i = cc][k = kk] = -R[p0 + 4 sx + kk][row0 + cc],  B[k = kk][j = cc] = R[p0 + 4 sx + kk][col0 + cc]
          const bool kv = 4 * sx + kk < K;
          const double a = (kv && row0 + cc < n) ? -S[(p0 + 4 * sx + kk) * n + row0 + cc] : 0.0;
          const double b = (kv && col0 + cc < n) ? S[(p0 + 4 * sx + kk) * n + col0 + cc] : 0.0;
          c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const int row = row0 + kk + 4 * rr, col = col0 + cc;
          if (row < n && col < n && col >= row) S[row * n + col] = c[rr];
        }
      }
      LDSF_STAMP(10);  // (eight-pivot groups) rank-16 update below the panel
    }
  }
  if (blocked && !eight) {
    typedef double d4_t __attribute__((ext_vector_type(4)));
    const int kk = lane >> 4, cc = lane & 15;
    for (int k0 = 0; k0 < nf; k0 += 4) {
      const int kb = min(4, nf - k0);
      // The four pivot rows against each other: every thread factors the kb x kb diagonal block in registers (ten broadcast LDS
      // reads; the same arithmetic in every thread), then one thread per column solves that column of the four-row panel against
      // it.  Three barriers per four pivots; pivot by pivot (scale the row, update the rows below it, each behind a barrier of its
      // own and a dependent sqrt / divide) the panel took ~1.6 us of the ~3 us a group of four costs.  Element by element the
      // operations and their order are the ones of the row-by-row form.
      __syncthreads();  // previous trailing update complete
      double d[4][4], inv[4];
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int c = q; c < 4; c++) d[q][c] = (c < kb) ? S[(k0 + q) * n + k0 + c] : ((q == c) ? 1.0 : 0.0);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        inv[q] = 1.0;
        if (q < kb) {
          double piv = d[q][q];
          if (!(piv > 0.0)) {
            if (piv <= 0.0) failed = true;  // Eigen LLT: pivot <= 0 -> NumericalIssue (NaN passes, like Eigen)
            piv = (piv == piv && piv != 0.0) ? fabs(piv) : 1.0;
          }
          double r;
          pivot_sqrt_inv(piv, &r, &inv[q]);
          d[q][q] = r;
#pragma unroll
          for (int c = q + 1; c < 4; c++) d[q][c] *= inv[q];
#pragma unroll
          for (int i = q + 1; i < 4; i++)
#pragma unroll
            for (int c = i; c < 4; c++) d[i][c] -= d[q][i] * d[q][c];
        }
      }
      __syncthreads();  // every thread has read the block
      if (tid < 16) {
        const int q = tid >> 2, c = tid & 3;
        if (q <= c && c < kb) {
          double v = 0.0;
#pragma unroll
          for (int qq = 0; qq < 4; qq++)
#pragma unroll
            for (int cc2 = 0; cc2 < 4; cc2++)
              if (qq == q && cc2 == c && cc2 >= qq) v = d[qq][cc2];
          S[(k0 + q) * n + k0 + c] = v;
        }
      }
      for (int j = k0 + kb + tid; j < n; j += nt) {
        double x[4];
#pragma unroll
        for (int q = 0; q < 4; q++) x[q] = (q < kb) ? S[(k0 + q) * n + j] : 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (q < kb) {
            x[q] *= inv[q];
#pragma unroll
            for (int i = q + 1; i < 4; i++) x[i] -= d[q][i] * x[q];
          }
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (q < kb) S[(k0 + q) * n + j] = x[q];
      }
      __syncthreads();
      // Two levels (round 3): inside a PANEL of sixteen pivots the rank-4 update only reaches the panel's own remaining rows (at most
      // twelve: one row of tiles); the rows below the panel receive the whole panel as ONE rank-16 update (four MFMAs on a tile that is
      // read and written once) after its last group.  Group by group over the whole trailing matrix, every tile went LDS -> registers
      // -> LDS four times per sixteen pivots: the Cholesky of a 70-row clique was 50 of the 70 us its front took.
      const int t0 = k0 + kb;
      const int p0 = k0 & ~15, pend = min(p0 + 16, nf);
      if (t0 < pend) {  // rows t0 .. pend - 1 of the panel, all columns from t0
        const int T = (n - t0 + 15) >> 4;
        for (int tj = wave; tj < T; tj += nw) {
          const int row0 = t0, col0 = t0 + 16 * tj;
          const bool kv = kk < kb;
          const double a = (kv && row0 + cc < pend) ? -S[(k0 + kk) * n + row0 + cc] : 0.0;
          const double b = (kv && col0 + cc < n) ? S[(k0 + kk) * n + col0 + cc] : 0.0;
          d4_t c;
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int row = row0 + kk + 4 * rr, col = col0 + cc;
            c[rr] = (row < pend && col < n) ? S[row * n + col] : 0.0;
          }
          c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int row = row0 + kk + 4 * rr, col = col0 + cc;
            if (row < pend && col < n && col >= row) S[row * n + col] = c[rr];
          }
        }
        continue;
      }
      // the panel p0 .. pend - 1 is finished: everything below it
      const int m = n - pend;
      if (m <= 0) continue;
      const int K = pend - p0;
      const int T = (m + 15) >> 4, ntile = T * (T + 1) / 2;
      for (int t = wave; t < ntile; t += nw) {
        int ti = 0, rem = t;
        while (rem >= T - ti) {
          rem -= T - ti;
          ti++;
        }
        const int row0 = pend + 16 * ti, col0 = pend + 16 * (ti + rem);
        d4_t c;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const int row = row0 + kk + 4 * rr, col = col0 + cc;
          c[rr] = (row < n && col < n) ? S[row * n + col] : 0.0;
        }
#pragma unroll
        for (int sx = 0; sx < 4; sx++) {
          // A[i = cc][k = kk] = -R[p0 + 4 sx + kk][row0 + cc],  B[k = kk][j = cc] = R[p0 + 4 sx + kk][col0 + cc]
          const bool kv = 4 * sx + kk < K;
          const double a = (kv && row0 + cc < n) ? -S[(p0 + 4 * sx + kk) * n + row0 + cc] : 0.0;
          const double b = (kv && col0 + cc < n) ? S[(p0 + 4 * sx + kk) * n + col0 + cc] : 0.0;
          c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const int row = row0 + kk + 4 * rr, col = col0 + cc;
          if (row < n && col < n && col >= row) S[row * n + col] = c[rr];
        }
      }
    }
  }
  for (int k = blocked ? nf : 0; k < nf; k++) {
    __syncthreads();  // previous trailing update (or assembly) complete
    double piv = S[k * n + k];
    if (!(piv > 0.0)) {
      if (piv <= 0.0) failed = true;  // Eigen LLT: pivot <= 0 -> NumericalIssue (NaN passes, like Eigen)
      piv = (piv == piv && piv != 0.0) ? fabs(piv) : 1.0;
    }
    const double r = sqrt(piv), inv = 1.0 / r;
    for (int j = k + 1 + tid; j < n; j += nt) S[k * n + j] *= inv;
    __syncthreads();  // row k scaled; every thread has read the pivot
    if (tid == 0) {
      S[k * n + k] = r;
      if (gather) *corner -= S[k * n + n - 1] * S[k * n + n - 1];
    }
    for (int i = k + 1 + wave; i < urows; i += nw) {
      const double rki = S[k * n + i];
      for (int j = i + lane; j < n; j += 64) S[i * n + j] -= rki * S[k * n + j];
    }
  }
  LDSF_STAMP(5);  // (up to the barrier that ends the Cholesky)
  __syncthreads();
  LDSF_STAMP(3);  // partial Cholesky
  if (tid == 0) {
    // pivot-exponent test, gtsam/base/cholesky.cpp:146-158
    if (nf >= 2) {
      if (!(frexp_exp(S[(nf - 2) * n + nf - 2]) - frexp_exp(S[(nf - 1) * n + nf - 1]) < 12)) failed = true;
    } else if (nf == 1) {
      if (!(frexp_exp(S[0]) > -12)) failed = true;
    }
    if (failed) atomicMin(status, F.id);
  }
  // ---- emit [R S d] (strictly-lower zeroed) and the update matrix
  double* RSd = pool + F.rsd_off;
  for (int i = wave; i < nf; i += nw)
    for (int j = lane; j < n; j += 64) RSd[(size_t)i * F.ld_rsd + j] = (j >= i) ? S[i * n + j] : 0.0;
  const int m = n - nf;
  if (gather) {
    if (tid == 0) gcorner[F.par_map] = *corner;
    // [S d] transposed, (n - nf) x nf: each separator variable's block is contiguous for the parent's gather
    double* St = pool + F.u_off;
    for (int idx = tid; idx < m * nf; idx += nt) {
      const int c = idx / nf, r = idx - c * nf;
      St[idx] = S[r * n + nf + c];
    }
  } else {
    double* U = pool + F.u_off;
    for (int i = wave; i < m; i += nw)
      for (int j = i + lane; j < m; j += 64) {
        if constexpr (DATAFLOW)
          __hip_atomic_store((unsigned long long*)(U + (size_t)i * F.ld_u + j), (unsigned long long)__double_as_longlong(S[(nf + i) * n + nf + j]),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
          U[(size_t)i * F.ld_u + j] = S[(nf + i) * n + nf + j];
      }
  }
#ifdef LDSF_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  LDSF_STAMP(4);  // [R S d] and the update matrix written
#endif
}

// "not published yet" over the upper triangles of the update matrices of the fronts of the merged launches: grid = entries of `list`
struct FillUpper {
  int64_t u_off;
  int32_t m, ld;
};
__global__ __launch_bounds__(256) void fill_upper_kernel(const FillUpper* __restrict__ list, double* __restrict__ pool) {
  const FillUpper f = list[blockIdx.x];
  unsigned long long* U = (unsigned long long*)(pool + f.u_off);
  for (int i = threadIdx.x >> 6; i < f.m; i += 4)
    for (int j = i + (threadIdx.x & 63); j < f.m; j += 64) U[(size_t)i * f.ld + j] = ~0ull;
}

template <bool GATHER, int MAXT = 256>
__global__ __launch_bounds__(MAXT) void lds_front_kernel(const int32_t* __restrict__ list, const FrontDesc* __restrict__ fronts,
                                                         const FrontFac* __restrict__ ffac, const FacDesc* __restrict__ fd,
                                                         const ChildRef* __restrict__ childs, const int32_t* __restrict__ cmap,
                                                         const int32_t* __restrict__ fxoff, double* __restrict__ pool, double lambda_v, const double* __restrict__ lambda_p,
                                                         const double* __restrict__ dampw, int* __restrict__ status, int nmax, int srows,
                                                         double* __restrict__ gcorner, int jcap, const double* __restrict__ gex,
                                                         const char* __restrict__ pack, int pack_stride) {
  lds_front_body<GATHER, MAXT, false>((int)blockIdx.x, list, fronts, ffac, fd, childs, cmap, fxoff, pool, lambda_v, lambda_p, dampw, status, nmax, srows, gcorner, jcap,
                                      gex, pack, pack_stride, FrontFlow{});
}

// The LDS fronts of SEVERAL consecutive tree levels in one launch, bottom-up (deep clique trees: an upper level is a handful of fronts, a
// launch of ~15-25 us of which most is latency -- descriptor chains, Jacobian staging -- that does not depend on the children at all).
// list[seg_begin, seg_end) holds the levels bottom-up; ticket t takes list[seg_begin + t], so children hold lower tickets than their
// parents: whatever a workgroup waits for has been started before it (no residency assumption).  The host fills the upper triangles of the
// update matrices of the launch's fronts with the all-ones pattern first (fill_upper_kernel).
template <int MAXT>
__global__ __launch_bounds__(MAXT) void lds_front_merged_kernel(const int32_t* __restrict__ list, int seg_begin, int seg_end, const FrontDesc* __restrict__ fronts,
                                                                const FrontFac* __restrict__ ffac, const FacDesc* __restrict__ fd,
                                                                const ChildRef* __restrict__ childs, const int32_t* __restrict__ cmap,
                                                                const int32_t* __restrict__ fxoff, double* __restrict__ pool, double lambda_v,
                                                                const double* __restrict__ lambda_p, const double* __restrict__ dampw, int* __restrict__ status,
                                                                int nmax, int jcap, const double* __restrict__ gex, unsigned int* __restrict__ ticket,
                                                                int* __restrict__ relay = nullptr) {
  __shared__ int s_ticket;
  if (threadIdx.x == 0) s_ticket = (int)atomicAdd(ticket, 1u);
  __syncthreads();
  lds_front_body<false, MAXT, true>(seg_begin + s_ticket, list, fronts, ffac, fd, childs, cmap, fxoff, pool, lambda_v, lambda_p, dampw, status, nmax, nmax,
                                    (double*)nullptr, jcap, gex, (const char*)nullptr, 0, FrontFlow{seg_begin, seg_end});
  // relay (ISAM2): the last workgroup to finish hands the status word to the host's pinned word (ticket[1] counts the finished ones; the
  // host resets it with the ticket) -- a kernel or a copy command of its own did that before
  if (relay) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (atomicAdd(ticket + 1, 1u) == gridDim.x - 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(relay, __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// back-substitution for LDS-class fronts with at most LDSB_SMALL_NF frontal scalars (leaves and the levels just above them: the host
// takes the workgroup-per-front kernels below for anything larger): x_F = R^-1 (d - S x_S).  One wave per front, 4 fronts per block.
// Built around the number of dependent memory round trips (each ~2-3 us, and BAL's 100 000 leaf fronts are ~12 rounds of resident
// waves): list -> descriptor -> {separator offsets, frontal offsets, the rows of S and d four at a time, lane i's row of R} in ONE
// batch of unconditional loads on clamped indices -> x_S -> arithmetic -> store.  The nf x nf solve is a register chain: lane i
// carries y_i / R_ii and its own row of R scaled to a unit diagonal; a step is one v_readlane pair and one fma.
// (Round 1's form gathered x_S through two dependent loads under a branch per 64 columns and solved row by row with a six-shuffle
//  reduction, an LDS round trip and a divide per unknown.)
#define LDSB_SMALL_NF 12
__global__ __launch_bounds__(256) void lds_backsub_kernel(const int32_t* __restrict__ list, int nlist, const FrontDesc* __restrict__ fronts,
                                                           const int32_t* __restrict__ fxoff, const int32_t* __restrict__ sxoff,
                                                           const double* __restrict__ pool, double* __restrict__ delta, int* __restrict__ status) {
  __shared__ double rhs_s[4][16];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = blockIdx.x * 4 + w;
  if (li >= nlist) return;
  const FrontDesc F = fronts[list[li]];
  const int n = F.n, nf = F.nf, ns = n - nf - 1;
  const double* RSd = pool + F.rsd_off;
  double* rhs = rhs_s[w];
  const int fo = fxoff[F.fx_begin + min(lane, nf - 1)];
  int so[3] = {0, 0, 0};
  if (ns > 0) {  // (wave-uniform; a front without separator is a root)
#pragma unroll
    for (int q = 0; q < 3; q++) so[q] = sxoff[F.sx_begin + min(lane + 64 * q, ns - 1)];
  }
  // lane i's row of R (columns 0 .. LDSB_SMALL_NF - 1, clamped): the coefficients of the register solve
  double rk[LDSB_SMALL_NF];
  {
    const double* row = RSd + (size_t)min(lane, nf - 1) * F.ld_rsd;
#pragma unroll
    for (int k = 0; k < LDSB_SMALL_NF; k++) rk[k] = row[min(k, nf - 1)];
  }
  double xs[3] = {0.0, 0.0, 0.0};
  // rhs_i = d_i - sum_j S_ij x_S[j], four rows at a time          (n <= 139  =>  at most three 64-column chunks per row)
  for (int i0 = 0; i0 < nf; i0 += 4) {
    double sv[4][3], dv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const double* row = RSd + (size_t)min(i0 + u, nf - 1) * F.ld_rsd;
#pragma unroll
      for (int q = 0; q < 3; q++) sv[u][q] = row[nf + max(min(lane + 64 * q, ns - 1), 0)];
      dv[u] = row[n - 1];
    }
    if (i0 == 0 && ns > 0) {  // behind the first batch of rows in program order: both are in flight together
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const double v = delta[so[q]];
        xs[q] = (lane + 64 * q < ns) ? v : 0.0;
      }
    }
    double acc[4];
#pragma unroll
    for (int u = 0; u < 4; u++) acc[u] = sv[u][0] * xs[0] + sv[u][1] * xs[1] + sv[u][2] * xs[2];
    // four rows reduced together: 2 + 1 + 4 shuffles instead of 24
    {
      const bool h5 = (lane & 32) != 0, h4 = (lane & 16) != 0;
      const double k0 = h5 ? acc[2] : acc[0], s0 = h5 ? acc[0] : acc[2], k1 = h5 ? acc[3] : acc[1], s1 = h5 ? acc[1] : acc[3];
      const double a0 = k0 + __shfl_xor(s0, 32), a1 = k1 + __shfl_xor(s1, 32);
      const double kk = h4 ? a1 : a0, ss = h4 ? a0 : a1;
      double t = kk + __shfl_xor(ss, 16);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o);
      const int slot = (h5 ? 2 : 0) + (h4 ? 1 : 0);
      // the lane that holds row `slot` also needs that row's d: dv[slot] by selects (dv is indexed statically)
      const double dsel = h5 ? (h4 ? dv[3] : dv[2]) : (h4 ? dv[1] : dv[0]);
      if ((lane & 15) == 0 && i0 + slot < nf) rhs[i0 + slot] = dsel - t;
    }
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // R x = rhs in registers
  double diag = 1.0;
#pragma unroll
  for (int k = 0; k < LDSB_SMALL_NF; k++)
    if (k == lane && k < nf) diag = rk[k];
  const double rd = 1.0 / diag;
  double yi = (lane < nf) ? rhs[lane] * rd : 0.0;
#pragma unroll
  for (int k = LDSB_SMALL_NF - 1; k >= 1; k--) {
    if (k < nf) {  // wave-uniform
      const double c = (lane < k) ? rk[k] * rd : 0.0;
      yi = fma(-c, readlane_dyn(yi, k), yi);
    }
  }
  if (lane < nf) {
    delta[fo] = yi;
    if (yi != yi) atomicMin(status, F.id);  // NaN -> IndeterminantLinearSystemException (linearAlgorithms-inst.h:99)
  }
}

// ---- back-substitution of an LDS front by one WORKGROUP, the whole [R S d] of the front staged in LDS
// (tree levels that hold fronts with more than a dozen frontal columns: the upper levels of general sparse graphs)
//   stage   : a flat, batched copy of the nf x n block -- before anything the front depends on, so that it overlaps the wait for the parent
//   x_S     : one gather of the separator part of delta (offsets fetched before the wait)
//   y       : d - S x_S, eight rows per wave and reduction (wave_reduce8 below), everything from LDS
//   solve   : 64 unknowns at a time by wave 0: lane i carries y_i / R_ii, a step is one v_readlane pair and one fma with the row scaled
//             to a unit diagonal (coefficients read eight ahead); the rows above a block are folded by all four waves
// Row by row with a six-shuffle reduction and a divide per unknown (the round-1 form) a front of 30 unknowns took ~10 us after its
// parent; this form leaves one memory round trip (x_S) and ~2 us of arithmetic on the parent-to-child chain.
// Dynamic LDS: (max over the launch of nf x (n | 1)) + 2 x 144 doubles.
#define LDSB_TAIL 288
__device__ __forceinline__ double ldsb_reduce8(double (&s)[8], int lane, int& slot) {  // as wave_reduce_slots<8> (kernels_dense.hpp)
  int o = 32;
  slot = 0;
#pragma unroll
  for (int half = 4; half >= 1; half >>= 1, o >>= 1) {
    const bool hi = (lane & o) != 0;
#pragma unroll
    for (int k = 0; k < half; k++) {
      const double keep = hi ? s[k + half] : s[k], send = hi ? s[k] : s[k + half];
      s[k] = keep + __shfl_xor(send, o);
    }
    slot += hi ? half : 0;
  }
#pragma unroll
  for (; o >= 1; o >>= 1) s[0] += __shfl_xor(s[0], o);
  return s[0];
}
__device__ __forceinline__ void ldsb_stage(const FrontDesc& F, const double* __restrict__ pool, double* Ls, int tid) {
  const int n = F.n, nf = F.nf, nl = n | 1, total = nf * n;
  const double* RSd = pool + F.rsd_off;
  for (int base = 0; base < total; base += 8 * 256) {
    double v[8];
    int at[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int idx = min(base + u * 256 + tid, total - 1), i = idx / n, j = idx - i * n;
      v[u] = RSd[(size_t)i * F.ld_rsd + j];  // unconditional on a clamped index: a batch of loads, not a chain
      at[u] = i * nl + j;
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (base + u * 256 + tid < total) Ls[at[u]] = v[u];
  }
}
// after the front's ancestors are visible.  so = this thread's separator offset (sxoff[sx_begin + min(tid, ns - 1)]); Ls staged by
// ldsb_stage (a barrier follows here).  Leaves x_F in LDS (the pointer returned, nf entries; a barrier precedes the return) and
// whether this thread saw a NaN in it.
// POLL (merged launches): delta was filled with the all-ones pattern before the back-substitution started and is written with agent-scope
// stores; a separator value that is still the pattern has not been published by its front yet -- the load of x_S IS the wait (no flag, no
// fence: the hand-off is one store and one load).  *timed_out: the bounded spin gave up (never expected).
template <bool POLL = false>
__device__ __forceinline__ double* ldsb_solve_core(const FrontDesc& F, double* Ls, int so, const double* __restrict__ delta, bool* bad_out,
                                                   bool* timed_out = nullptr, const double* xv_pre = nullptr /* this thread's x_S value, already fetched */) {
  const int n = F.n, nf = F.nf, ns = n - nf - 1, nl = n | 1;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  double* xsl = Ls + (size_t)nf * nl;  // [144]
  double* y = xsl + 144;               // [144]
  double xv;
  if constexpr (POLL) {
    unsigned long long bits = 0ull;
    if (ns > 0) {
      long spins = 0;
      while ((bits = __hip_atomic_load((const unsigned long long*)(delta + so), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == ~0ull) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 4000000L) {
          *timed_out = true;
          break;
        }
      }
    }
    xv = __longlong_as_double((long long)bits);
  } else {
    xv = xv_pre ? *xv_pre : delta[so];
  }
  if (tid < ns) xsl[tid] = xv;
  __syncthreads();  // Ls and x_S in LDS
  for (int i0 = w; i0 < nf; i0 += 32) {  // rows i0, i0 + 4, ..., i0 + 28
    double acc[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int i = min(i0 + 4 * r, nf - 1);
      double a = 0.0;
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const int j = lane + 64 * q;
        if (j < ns) a += Ls[i * nl + nf + j] * xsl[j];
      }
      acc[r] = a;
    }
    int slot;
    const double tot = ldsb_reduce8(acc, lane, slot);
    const int i = i0 + 4 * slot;
    if ((lane & 7) == 0 && i < nf) y[i] = Ls[i * nl + n - 1] - tot;
  }
  __syncthreads();
  bool bad = false;
  const int nblk = (nf + 63) >> 6;
  for (int b = nblk - 1; b >= 0; b--) {
    const int r0 = 64 * b, nb = min(64, nf - r0);
    if (w == 0) {
      const int il = r0 + min(lane, nb - 1);
      const double rd = (lane < nb) ? 1.0 / Ls[il * nl + il] : 1.0;
      double yi = (lane < nb) ? y[il] * rd : 0.0;
      for (int k0 = (nb - 1) | 7; k0 >= 0; k0 -= 8) {  // (columns k >= nb carry zero coefficients: a 3-row front takes one group of eight, not eight)
        double cf[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int k = k0 - u;
          const double c = Ls[il * nl + r0 + min(k, nb - 1)];
          cf[u] = (lane < k && k < nb) ? c * rd : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) yi = fma(-cf[u], readlane_dyn(yi, k0 - u), yi);
      }
      if (lane < nb) {
        y[r0 + lane] = yi;
        if (yi != yi) bad = true;
      }
    }
    if (b == 0) break;
    __syncthreads();  // x of this block in y
    for (int i0 = w; i0 < r0; i0 += 32) {
      double acc[8];
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const int i = min(i0 + 4 * r, r0 - 1);
        acc[r] = (lane < nb) ? Ls[i * nl + r0 + lane] * y[r0 + lane] : 0.0;
      }
      int slot;
      const double tot = ldsb_reduce8(acc, lane, slot);
      const int i = i0 + 4 * slot;
      if ((lane & 7) == 0 && i < r0) y[i] -= tot;
    }
    __syncthreads();
  }
  __syncthreads();
  *bad_out = bad;
  return y;
}
// the same, stored: fo = this thread's frontal offset (fxoff[fx_begin + min(tid, nf - 1)])
__device__ __forceinline__ void ldsb_solve(const FrontDesc& F, double* Ls, int so, int fo, double* __restrict__ delta, int* __restrict__ status) {
  bool bad;
  const double* y = ldsb_solve_core(F, Ls, so, delta, &bad);
  if ((int)threadIdx.x < F.nf) delta[fo] = y[threadIdx.x];
  if (bad && (threadIdx.x & 63) == 0) atomicMin(status, F.id);  // NaN -> IndeterminantLinearSystemException (linearAlgorithms-inst.h:99)
}

__global__ __launch_bounds__(256) void lds_backsub_wide_kernel(const int32_t* __restrict__ list, int nlist, const FrontDesc* __restrict__ fronts,
                                                                const int32_t* __restrict__ fxoff, const int32_t* __restrict__ sxoff,
                                                                const double* __restrict__ pool, double* __restrict__ delta,
                                                                int* __restrict__ status) {
  extern __shared__ double Ls[];
  const FrontDesc F = fronts[list[blockIdx.x]];
  const int tid = threadIdx.x, ns = F.n - F.nf - 1;
  const int so = sxoff[F.sx_begin + max(min(tid, ns - 1), 0)], fo = fxoff[F.fx_begin + min(tid, F.nf - 1)];
  ldsb_stage(F, pool, Ls, tid);
  ldsb_solve(F, Ls, ns > 0 ? so : fo, fo, delta, status);
}

// The LDS fronts of SEVERAL consecutive tree levels in one launch (deep clique trees: a level's launch does ~10-25 us of work
// and costs about as much again in launch latency).  list[seg_begin, seg_end) holds the levels bottom-up; ticket t takes
// list[seg_end - 1 - t], so parents hold lower tickets than their children.  Hand-off through delta itself (round 3; a flag per front
// with release / acquire fences before): the host fills delta with the all-ones pattern before the back-substitution, a front stores its
// x_F with agent-scope stores, and a front that needs a value polls the value (ldsb_solve_core<true>): one store and one load per tree
// level instead of store, write-back, flag, poll, invalidate, load.  Same per-front work as lds_backsub_wide_kernel, the staging of
// [R S d] under the wait; tickets drawn at workgroup start (whatever a workgroup waits for has started before it), bounded spin.
__global__ __launch_bounds__(256) void lds_backsub_merged_kernel(const int32_t* __restrict__ list, int seg_begin, int seg_end,
                                                                  const FrontDesc* __restrict__ fronts, const int32_t* __restrict__ fxoff,
                                                                  const int32_t* __restrict__ sxoff, const double* __restrict__ pool,
                                                                  double* __restrict__ delta, int* __restrict__ status,
                                                                  const int32_t* __restrict__ parent_of, const int32_t* __restrict__ pos_of,
                                                                  unsigned int* __restrict__ done, unsigned int* __restrict__ ticket) {
  extern __shared__ double Ls[];
  __shared__ int s_ticket, s_ok;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) s_ticket = (int)atomicAdd(ticket, 1u);
  __syncthreads();
  const int fi = list[seg_end - 1 - s_ticket];
  const FrontDesc F = fronts[fi];
  const int ns = F.n - F.nf - 1;
  const int so = sxoff[F.sx_begin + max(min(tid, ns - 1), 0)], fo = fxoff[F.fx_begin + min(tid, F.nf - 1)];
  // [R S d] does not depend on the ancestors: it is staged before the first look at x_S
  ldsb_stage(F, pool, Ls, tid);
  bool bad, timed_out = false;
  const double* y = ldsb_solve_core<true>(F, Ls, ns > 0 ? so : fo, delta, &bad, &timed_out);
  if (tid < F.nf)  // published value by value: whoever needs it polls the value itself
    __hip_atomic_store((unsigned long long*)(delta + fo), (unsigned long long)__double_as_longlong(y[tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (bad && lane == 0) atomicMin(status, F.id);  // NaN -> IndeterminantLinearSystemException (linearAlgorithms-inst.h:99)
  if (timed_out) atomicExch(status + 1, 1 + F.id);  // never expected: spin bound hit (a fault, reported apart from pivot failures)
  (void)s_ok;
  (void)parent_of;
  (void)pos_of;
  (void)done;
  (void)lane;
}

}  // namespace lmgpu
