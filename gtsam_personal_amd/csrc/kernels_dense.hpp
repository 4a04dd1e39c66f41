// HBM fronts: fronts too large for one workgroup's LDS live in HBM (row-major upper, leading dimension ld = n rounded up to 16)
// and are processed by a blocked right-looking partial Cholesky with outer panels of 256 rows:
//   panel 0                     -> panel_dataflow_kernel (kernels_potrf.hpp)
//   all following panels        -> chain_kernel (kernels_step.hpp): per outer panel i the trailing update with panel i (syrk_tile
//                                  below, K = 256) + the factorisation of panel i+1, all steps of the front in ONE launch
//                                  (step_kernel: one step per launch, for runs that cannot be chained)
// so the big trailing matrix is read-modified-written once per 256 eliminated rows.
// This is choleskyPartial (gtsam/base/cholesky.cpp:108-159: LLT(A); S = R^-T B; C -= S^T S; pivot-exponent test)
// in blocked form; the root of a BAL problem (all cameras, 9001 x 9001) spends most of the solve here.
// This file: assembly of an HBM front (own factors / non-leaf children: FP64 atomics; leaf children: kernels_schur.hpp),
// the trailing-update tile, and the back-substitution kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_front.hpp"

namespace lmgpu {

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- assembly
// ---- atomic-free, bitwise-reproducible assembly of an HBM front: one WAVE owns one row of the front
// Row R of the upper triangle receives every contribution whose smaller front index is R: from a child's update matrix U (row i of
// the child maps to R) the entries U[min(i,j)][max(i,j)] for every child index j with map[j] >= R, and from an own factor whose
// local column p maps to R the products sum_r J[r][p] J[r][q] for its columns q with column(q) >= R.  The sources of a row are a
// list built once by the host (RowSrc: children in child order, then factors in graph order); the wave walks it sequentially, its
// lanes take the j / q of one source (distinct destinations inside one source: a child's column map is injective) -- so every entry
// of the front is a sum in a FIXED order with no atomics: two solves give bitwise the same front.
struct RowSrc {
  int32_t idx;  // >= 0: child reference (index into childs[]); < 0: own factor, -(index into ffac[]) - 1
  int32_t i;    // child: row / column index inside its update matrix; factor: local column p of [A b]
};
__device__ __forceinline__ void assemble_row_body(const FrontDesc& F, int64_t f_off, int ld, const int32_t* __restrict__ rowptr, const RowSrc* __restrict__ src,
                                                  const ChildRef* __restrict__ childs, const int32_t* __restrict__ cmap,
                                                  const FrontFac* __restrict__ ffac, const FacDesc* __restrict__ fd, double* __restrict__ pool,
                                                  int R, bool with_factors) {
  const int lane = threadIdx.x & 63;
  double* Arow = pool + f_off + (size_t)R * ld;
  for (int e = rowptr[R]; e < rowptr[R + 1]; e++) {
    const RowSrc sr = src[e];
    if (sr.idx >= 0) {
      const ChildRef c = childs[sr.idx];
      const double* U = pool + c.u_off;
      const int32_t* map = cmap + c.map_begin;
      const int i = sr.i;
      for (int j = lane; j < c.m; j += 64) {
        const int gj = map[j];
        if (gj >= R) Arow[gj] += (j >= i) ? U[(size_t)i * c.ld + j] : U[(size_t)j * c.ld + i];
      }
    } else if (with_factors) {
      const FrontFac ff = ffac[-sr.idx - 1];
      const FacDesc d = fd[ff.fac];
      const double* J = pool + d.joff;
      const int m = d.rows, nc = d.d0 + d.d1 + d.d2 + 1, p = sr.i;
      for (int q = lane; q < nc; q += 64) {
        const int gq = fac_col(d, ff.c0, ff.c1, ff.c2, q, F.n);
        if (gq > R || (gq == R && q == p)) {
          double v = 0;
          for (int r = 0; r < m; r++) v += J[p * m + r] * J[q * m + r];
          Arow[gq] += v;
        }
      }
    }
    // the next source of this row may touch the same entries from other lanes of this wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}
// grid: ceil(n / 4) blocks of 4 waves
__global__ __launch_bounds__(256) void hbm_assemble_rows_kernel(FrontDesc F, int64_t f_off, int ld, const int32_t* __restrict__ rowptr,
                                                                const RowSrc* __restrict__ src, const ChildRef* __restrict__ childs,
                                                                const int32_t* __restrict__ cmap, const FrontFac* __restrict__ ffac,
                                                                const FacDesc* __restrict__ fd, double* __restrict__ pool, int with_factors) {
  const int R = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (R >= F.n) return;
  assemble_row_body(F, f_off, ld, rowptr, src, childs, cmap, ffac, fd, pool, R, with_factors != 0);
}

__global__ __launch_bounds__(256) void hbm_damp_kernel(FrontDesc F, int64_t f_off, int ld, const int32_t* __restrict__ fxoff,
                                                        double* __restrict__ pool, double lambda_v, const double* __restrict__ lambda_p, const double* __restrict__ dampw,
                                                        const double* __restrict__ gex) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= F.nf) return;
  pool[f_off + (size_t)i * ld + i] += (lambda_p ? *lambda_p : lambda_v) * dampw[fxoff[F.fx_begin + i]];
  if (gex) pool[f_off + (size_t)i * ld + F.n - 1] += gex[fxoff[F.fx_begin + i]];
}

// Clears a list of ranges of the pool in ONE launch: grid (slices, ranges).  A general sparse graph has dozens of HBM fronts scattered
// through the pool; a memset node per front cost ~5 us each at the head of every solve (41 of them on city10000).
struct ZeroRange {
  int64_t off, count;  // doubles; both multiples of 2
};
__global__ __launch_bounds__(256) void zero_ranges_kernel(const ZeroRange* __restrict__ ranges, double* __restrict__ pool) {
  const ZeroRange r = ranges[blockIdx.y];
  double2* p = (double2*)(pool + r.off);
  const int64_t n2 = r.count >> 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) p[i] = double2{0.0, 0.0};
}

// Every small clear / sentinel fill of a phase as ONE launch (a launch of a few hundred bytes costs ~5 us of stream time like any other:
// sphere2500 spent 12 of its 101 launches per solve on them).  One workgroup per chunk of at most FILL_CHUNK_BYTES.
#define FILL_CHUNK_BYTES 32768
struct FillChunk {
  unsigned long long dst;  // device address, 4-byte aligned
  uint32_t bytes;          // multiple of 4
  uint32_t value;          // the 32-bit pattern
};
__global__ __launch_bounds__(256) void fill_chunks_kernel(const FillChunk* __restrict__ table) {
  const FillChunk c = table[blockIdx.x];
  if (((c.dst | c.bytes) & 15ull) == 0) {
    uint4* p = (uint4*)c.dst;
    const uint4 v{c.value, c.value, c.value, c.value};
    for (uint32_t i = threadIdx.x; i < (c.bytes >> 4); i += 256) p[i] = v;
  } else {
    uint32_t* p = (uint32_t*)c.dst;
    for (uint32_t i = threadIdx.x; i < (c.bytes >> 2); i += 256) p[i] = c.value;
  }
}

// ---------------------------------------------------------------- trailing update on the matrix cores
// C[i][j] -= sum_{p < kp} P[p][i] P[p][j]   for r0 <= i < r1, i <= j < n,   P = rows p0 .. p0+kp-1 of A (finished [R S d] rows).
// Block = 4 waves computing a 128x128 tile (each wave 64x64 = 4x4 MFMA 16x16x4 f64 tiles, 128 accumulator VGPRs).
// The two 16-row x 128-column operand slabs of each K-chunk are DMA'd HBM/L2 -> LDS with global_load_lds (one 1-KiB
// row per wave-instruction, no VGPR staging), double-buffered; fragments are ds_read_b64 (lane l: P[k + (l>>4)][col + (l&15)]).
// LDS row stride 144 doubles: rows k and k+1 of a fragment land on opposite halves of the 64-bank row (conflict-free).
#define SYRK_KC 16
#define SYRK_LDW 144
__device__ __forceinline__ void glds_row(const double* g, double* lds_row) {
  // 64 lanes x 16 B = one 128-double row; LDS destination = wave-uniform base + lane * 16
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_row, 16, 0, 0);
}

// Sadd != nullptr (multi-rank, rows of the next panel): the tile also receives its so far separate assembled contributions,
// C += Sadd - P^T P (same layout as A)
template <bool HAS_S = false>
__device__ __forceinline__ void syrk_tile(double* __restrict__ A, int ld, int n, int p0, int kp, int r0, int r1, int ti, int tj,
                                          double* sm /* [2 stages][2 operands][SYRK_KC][SYRK_LDW] */, const double* __restrict__ Sadd = nullptr) {
  if (tj < ti) return;
  const int it0 = r0 + ti * 128, jt0 = r0 + tj * 128;  // tile origins
  if (it0 >= r1 || jt0 >= n) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const bool diag = (ti == tj);
  const bool active = !(diag && wr > wc) && (it0 + wr * 64 < r1) && (jt0 + wc * 64 < n);  // wave has something to store
  const int kk = lane >> 4, cc = lane & 15;
  double4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = double4_t{0, 0, 0, 0};
  const int nchunk = (kp + SYRK_KC - 1) / SYRK_KC;
  const double* P = A + (size_t)p0 * ld;
  // stage loader: 2 operands x 16 rows = 32 row-DMAs per chunk, 8 per wave.  Rows >= kp are zero-filled.
  auto issue = [&](int c, int buf) {
    double* base = sm + (size_t)buf * 2 * SYRK_KC * SYRK_LDW;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int rr = wave * 8 + q;           // 0..31
      const int op = rr >> 4, row = rr & 15;  // operand (0 = A side / tile rows, 1 = B side / tile cols), k-row in chunk
      double* dst = base + ((size_t)op * SYRK_KC + row) * SYRK_LDW;
      const int krow = c * SYRK_KC + row;
      const int col0 = (op == 0) ? it0 : jt0;
      if (krow < kp) {
        glds_row(P + (size_t)krow * ld + col0 + lane * 2, dst);
      } else {
        dst[lane * 2] = 0.0;
        dst[lane * 2 + 1] = 0.0;
      }
    }
  };
  issue(0, 0);
  for (int c = 0; c < nchunk; c++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (c + 1 < nchunk) issue(c + 1, (c + 1) & 1);
    if (active) {
      const double* sA = sm + (size_t)(c & 1) * 2 * SYRK_KC * SYRK_LDW + wr * 64 + cc;
      const double* sB = sm + (size_t)(c & 1) * 2 * SYRK_KC * SYRK_LDW + (size_t)SYRK_KC * SYRK_LDW + wc * 64 + cc;
#pragma unroll
      for (int ks = 0; ks < SYRK_KC; ks += 4) {
        double af[4], bf[4];
#pragma unroll
        for (int a = 0; a < 4; a++) af[a] = sA[(ks + kk) * SYRK_LDW + a * 16];  // acc = +P^T P, subtracted in the epilogue
#pragma unroll
        for (int b = 0; b < 4; b++) bf[b] = sB[(ks + kk) * SYRK_LDW + b * 16];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int b = 0; b < 4; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
      }
    }
  }
  if (!active) return;
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg.
  // Read-modify-write of the wave's 64x64 block in four batches of 16 independent loads (a load -> wait -> add -> store
  // chain per element would put 64 memory round trips behind every tile).
  const int i0 = it0 + wr * 64, j0 = jt0 + wc * 64;
  const bool full = (i0 + 64 <= r1) && (j0 + 64 <= n) && (j0 >= i0 + 63);  // wave-uniform: every element valid
  if (full) {
#pragma unroll
    for (int a = 0; a < 4; a++) {
      double c[4][4];
#pragma unroll
      for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) c[b][r] = A[(size_t)(i0 + a * 16 + kk + 4 * r) * ld + j0 + b * 16 + cc];
      if constexpr (HAS_S) {
        if (Sadd) {
#pragma unroll
          for (int b = 0; b < 4; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) c[b][r] += Sadd[(size_t)(i0 + a * 16 + kk + 4 * r) * ld + j0 + b * 16 + cc];
        }
      }
#pragma unroll
      for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) A[(size_t)(i0 + a * 16 + kk + 4 * r) * ld + j0 + b * 16 + cc] = c[b][r] - acc[a][b][r];
    }
  } else {
#pragma unroll
    for (int a = 0; a < 4; a++) {
      double c[4][4];
#pragma unroll
      for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {  // clamped (always mapped) address; the value is only used where valid
          const int row = min(i0 + a * 16 + kk + 4 * r, r1 - 1), col = min(j0 + b * 16 + cc, n - 1);
          c[b][r] = A[(size_t)row * ld + col];
          if constexpr (HAS_S) {
            if (Sadd) c[b][r] += Sadd[(size_t)row * ld + col];
          }
        }
#pragma unroll
      for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = i0 + a * 16 + kk + 4 * r, col = j0 + b * 16 + cc;
          if (row < r1 && col < n && col >= row) A[(size_t)row * ld + col] = c[b][r] - acc[a][b][r];
        }
    }
  }
}

__global__ __launch_bounds__(256, 2) void syrk_mfma_kernel(double* __restrict__ A, int ld, int n, int p0, int kp, int r0, int r1) {
  extern __shared__ double sm[];
  syrk_tile(A, ld, n, p0, kp, r0, r1, blockIdx.y, blockIdx.x, sm);
}

// ---------------------------------------------------------------- back-substitution on an HBM front
// y_i = d_i - sum_j S_ij x_S[j]     (one wave per row)
__global__ __launch_bounds__(64) void hbm_rhs_init_kernel(FrontDesc F, int64_t f_off, int ld, const int32_t* __restrict__ sxoff,
                                                           const double* __restrict__ pool, const double* __restrict__ delta,
                                                           double* __restrict__ y) {
  const int i = blockIdx.x, lane = threadIdx.x;
  const int n = F.n, nf = F.nf, ns = n - nf - 1;
  const double* row = pool + f_off + (size_t)i * ld;
  double s = 0;
  for (int j = lane; j < ns; j += 64) s += row[nf + j] * delta[sxoff[F.sx_begin + j]];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) y[i] = row[n - 1] - s;
}

#define BSS_MAX_NF 1024  // fronts up to this many frontal rows take the block kernel below; larger ones the per-front dataflow path

// Sums over the 64 lanes of a wave of R per-lane values at once: each halving step trades half of the values with the lane `o` away,
// so R values cost R - 1 + log2(64 / R) shuffles instead of 6 R dependent ones (a chain of 96 ds_bpermute pairs for 16 rows was
// 8 us of every 64-row block).  Returns, in EVERY lane, the total of value number `slot`.
template <int R>
__device__ __forceinline__ double wave_reduce_slots(double (&s)[R], int lane, int& slot) {
  int o = 32;
  slot = 0;
#pragma unroll
  for (int half = R / 2; half >= 1; half >>= 1, o >>= 1) {
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

// ---------------------------------------------------------------- back-substitution of the smaller HBM fronts, one workgroup per 64-row block
// General sparse graphs (SLAM) have a handful of fronts of a few hundred columns on each of their upper levels.  A single workgroup
// reads such a front at 13-20 GB/s (one CU's worth of requests in flight against a ~3 us round trip on an otherwise idle device),
// which is what round 1's one-workgroup-per-front kernel was bound by even after its reductions and its solve chain were cleaned up:
// 131 -> 80 us for a 330-column root, 0.92 of the 3.1 ms of an LM iteration on sphere2500 (profiles/r02/backsolve_bench.txt).
// Here every 64-row block of every such front of a tree level is a workgroup of its own, all in ONE
// launch: block b folds the separator part into its right-hand side, then the blocks x_j (j > b) of its own front as their owners
// publish them, solves its diagonal block and publishes x_b.  Wave w owns rows 16 w .. 16 w + 15 of the block, a lane one column:
// row reads are whole 512-byte segments, the per-lane partial sums of a row stay in registers over all hops and are reduced ONCE
// (wave_reduce_slots), and every wave polls x_j itself -- no barrier and no shuffle on a hop.  The data is the flag: xbuf is preset
// to all ones (a NaN pattern no computation produces; producers canonicalise their NaNs).  Order: a ticket indexes a table in which
// the blocks of a front appear from the last to the first, so a workgroup only waits for workgroups that started before it.
struct BsdBlock {  // everything a workgroup needs, in one record: it is ONE dependent load away from its data
  int64_t f_off;
  int32_t ld, n, nf, fx_begin, sx_begin, id;
  int32_t b, xoff;  // xoff: where the front's x starts in xbuf (64 entries per block, the last block zero-padded)
};
#define BSD_SPIN_LIMIT 4000000L
__global__ __launch_bounds__(256) void hbm_backsolve_blocks_kernel(const BsdBlock* __restrict__ table, unsigned int* __restrict__ ticket,
                                                                    const int32_t* __restrict__ fxoff, const int32_t* __restrict__ sxoff,
                                                                    const double* __restrict__ pool, double* __restrict__ delta,
                                                                    double* __restrict__ xbuf, int* __restrict__ status, int poll = 0) {
  // poll != 0: the blocks of SEVERAL consecutive tree levels in this launch (levels that hold nothing but such fronts: the upper ten
  // levels of sphere2500 are one or two mid-size fronts each).  delta was preset to all ones by the host and a front's separator values
  // are awaited BY VALUE (the table lists the levels top-down, so a front's ancestors hold lower tickets); x_F is published with
  // agent-scope stores.
  __shared__ double Db[64][65];
  __shared__ double yb[64];
  __shared__ int s_t;
  // ticket == nullptr: the whole grid is resident at once (the host checks: at most one workgroup per CU), so no order of dispatch
  // can leave a waiting workgroup without its producer, and the atomic's round trip is saved
  int t = blockIdx.x;
  if (ticket) {
    if (threadIdx.x == 0) s_t = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    t = s_t;
  }
  const BsdBlock B = table[t];
  struct { int32_t fx_begin, sx_begin, id; } F{B.fx_begin, B.sx_begin, B.id};
  const double* A = pool + B.f_off;
  const int ld = B.ld, n = B.n, nf = B.nf, ns = n - nf - 1;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nblk = (nf + 63) >> 6, b = B.b, r0 = 64 * b, nb = min(64, nf - r0);
  double* xb = xbuf + B.xoff;
  // (every load is unconditional on a clamped address: the compiler waits for a load issued under a branch at the end of that branch)
  const double* Arow[16];
#pragma unroll
  for (int k = 0; k < 16; k++) Arow[k] = A + (size_t)min(r0 + 16 * wave + k, nf - 1) * ld;
  // the slot (row 16 wave + slot) whose total wave_reduce_slots leaves in this lane, its right-hand side and its delta offset
  const int myslot = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
  const int myrow = 16 * wave + myslot;
  const double dv = A[(size_t)min(r0 + myrow, nf - 1) * ld + n - 1];
  const int fo = fxoff[F.fx_begin + r0 + min(lane, nb - 1)];
  double dbr[16], tl[16];
  {
    const int c = r0 + min(lane, nb - 1);
#pragma unroll
    for (int k = 0; k < 16; k++) dbr[k] = Arow[k][c];
  }
  auto load_tile = [&](int j, double(&t)[16]) {  // R[rows of this block, columns of block j]; columns past nf are clamped (they meet x = 0)
    const int c = min(64 * j + lane, nf - 1);
#pragma unroll
    for (int k = 0; k < 16; k++) t[k] = Arow[k][c];
  };
  load_tile(nblk - 1, tl);  // (b == nblk - 1: loaded and not used)
  double acc[16];
#pragma unroll
  for (int k = 0; k < 16; k++) acc[k] = 0.0;
  // separator part: acc[row] += S[row][j] x_S[j], two chunks of 64 columns in flight
  for (int c0 = 0; c0 < ns; c0 += 128) {
    const int j0 = c0 + lane, j1 = c0 + 64 + lane;
    const int o0 = sxoff[F.sx_begin + min(j0, ns - 1)], o1 = sxoff[F.sx_begin + min(j1, ns - 1)];
    double v0[16], v1[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v0[k] = Arow[k][nf + min(j0, ns - 1)];
#pragma unroll
    for (int k = 0; k < 16; k++) v1[k] = Arow[k][nf + min(j1, ns - 1)];
    double d0, d1;
    if (poll) {
      long spins = 0;
      for (;;) {
        d0 = __hip_atomic_load(&delta[o0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        d1 = __hip_atomic_load(&delta[o1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__double_as_longlong(d0) != -1LL && __double_as_longlong(d1) != -1LL) break;
        if (++spins > BSD_SPIN_LIMIT) {
          atomicExch(status + 1, 1 + F.id);  // never expected: spin bound hit (a fault, reported apart from pivot failures)
          d0 = d1 = 0.0;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    } else {
      d0 = delta[o0];
      d1 = delta[o1];
    }
    const double x0 = (j0 < ns) ? d0 : 0.0, x1 = (j1 < ns) ? d1 : 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) acc[k] += v0[k] * x0 + v1[k] * x1;
  }
  // the diagonal block into LDS (identity-padded), long since arrived
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int p = 16 * wave + k;
    Db[p][lane] = (p < nb && lane < nb && lane >= p) ? dbr[k] : ((p == lane) ? 1.0 : 0.0);
  }
  bool dead = false;
  for (int j = nblk - 1; j > b; j--) {
    double xj;
    long spins = 0;
    for (;;) {
      xj = __hip_atomic_load(&xb[64 * j + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__double_as_longlong(xj) != -1LL) break;
      if (dead || ++spins > BSD_SPIN_LIMIT) {
        dead = true;
        xj = 0.0;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    double tn[16];
    load_tile(j - 1, tn);  // behind the poll in program order: in flight while x_j is folded in and x_{j-1} is awaited
#pragma unroll
    for (int k = 0; k < 16; k++) acc[k] += tl[k] * xj;
#pragma unroll
    for (int k = 0; k < 16; k++) tl[k] = tn[k];
  }
  if (__any(dead) && lane == 0) atomicExch(status + 1, 1 + F.id);  // never expected: spin bound hit (a fault, reported apart from pivot failures)
  int slot;
  const double tot = wave_reduce_slots<16>(acc, lane, slot);
  if ((lane & 3) == 0) yb[myrow] = (myrow < nb) ? dv - tot : 0.0;
  __syncthreads();
  if (wave == 0) {
    // lane i carries y_i / R_ii and row i of the block scaled to a unit diagonal: a step is one v_readlane pair and one fma
    const double rd = 1.0 / Db[lane][lane];
    double yi = yb[lane] * rd;
    for (int k0 = 63; k0 >= 0; k0 -= 8) {
      double cf[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const double c = Db[lane][k0 - u];
        cf[u] = (lane < k0 - u) ? c * rd : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) yi = fma(-cf[u], readlane_dyn(yi, k0 - u), yi);
    }
    const double pub = (lane < nb) ? ((yi != yi) ? __longlong_as_double(0x7ff8000000000000LL) : yi) : 0.0;  // never the sentinel
    __hip_atomic_store(&xb[r0 + lane], pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane < nb) {
      if (poll)
        __hip_atomic_store(&delta[fo], pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        delta[fo] = yi;
      if (yi != yi) atomicMin(status, F.id);  // NaN -> IndeterminantLinearSystemException (linearAlgorithms-inst.h:99)
    }
  }
}

// ---------------------------------------------------------------- dataflow back-substitution (one launch per HBM front)
// Inverse of every NB x NB diagonal block of R (upper triangular), all blocks in parallel: one block per workgroup,
// thread j back-substitutes column j of the inverse in LDS.  Partial last block is identity-padded.
template <int NB>
__global__ __launch_bounds__(64) void hbm_invert_diag_kernel(const double* __restrict__ A, int ld, int nf, double* __restrict__ inv) {
  __shared__ double T[NB][NB];  // T[i][k] is read as a broadcast, X[k][j] lane-contiguous: no padding needed (2 x 32 KB)
  __shared__ double X[NB][NB];
  const int b = blockIdx.x, r0 = b * NB, nb = min(NB, nf - r0), j = threadIdx.x;
  for (int idx = j; idx < NB * NB; idx += NB) {
    const int p = idx / NB, q = idx - p * NB;
    double v = (p == q) ? 1.0 : 0.0;
    if (p < nb && q < nb && q >= p) v = A[(size_t)(r0 + p) * ld + r0 + q];
    T[p][q] = v;
  }
  __syncthreads();
  // column j of X = T^-1:  X[j][j] = 1/T[j][j];  X[i][j] = -(sum_{k=i+1..j} T[i][k] X[k][j]) / T[i][i]
  for (int i = NB - 1; i >= 0; i--) {
    double s = (i == j) ? 1.0 : 0.0;
    if (i <= j) {
      for (int k = i + 1; k <= j; k++) s -= T[i][k] * X[k][j];
      s /= T[i][i];
    } else {
      s = 0.0;
    }
    X[i][j] = s;
  }
  __syncthreads();
  double* out = inv + (size_t)b * NB * NB;
  for (int idx = j; idx < NB * NB; idx += NB) out[idx] = X[idx / NB][idx % NB];
}

// The same inverses from what the factorisation left behind: the 16x16 inverses of the diagonal tiles (inv16, by-product of the
// register Cholesky, kernels_potrf.hpp; 16 per 256-row panel, panel p at inv16 + p * 4096).  Block back-substitution on the
// matrix cores, one wave per 64x64 block:  X_hh = I_h;  X_gh = -I_g (sum_{k = g+1..h} R_gk X_kh)  for g < h -- 64 MFMAs instead of
// 2048 dependent LDS round trips per column (82 -> a few us for the 141 blocks of the C4 root).
__global__ __launch_bounds__(64) void hbm_invert_diag64_from16_kernel(const double* __restrict__ A, int ld, int nf, const double* __restrict__ inv16,
                                                                       double* __restrict__ inv) {
  typedef double d4_t __attribute__((ext_vector_type(4)));
  __shared__ double Rl[64][65];
  __shared__ double Xl[64][65];
  __shared__ double Il[4][16][17];
  const int bb = blockIdx.x, r0 = 64 * bb, nb = min(64, nf - r0), lane = threadIdx.x, kk = lane >> 4, cc = lane & 15;
  for (int idx = lane; idx < 64 * 64; idx += 64) {
    const int p = idx >> 6, q = idx & 63;
    double v = (p == q) ? 1.0 : 0.0;
    if (p < nb && q < nb && q >= p) v = A[(size_t)(r0 + p) * ld + r0 + q];
    Rl[p][q] = v;
    Xl[p][q] = 0.0;
  }
  const double* I16 = inv16 + (size_t)(bb >> 2) * 4096 + (size_t)(4 * (bb & 3)) * 256;
  for (int idx = lane; idx < 4 * 256; idx += 64) Il[idx >> 8][(idx >> 4) & 15][idx & 15] = I16[idx];
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int h = 0; h < 4; h++) {
#pragma unroll
    for (int r = 0; r < 4; r++) Xl[16 * h + kk + 4 * r][16 * h + cc] = Il[h][kk + 4 * r][cc];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int g = h - 1; g >= 0; g--) {
      d4_t m = d4_t{0, 0, 0, 0};
#pragma unroll
      for (int k = g + 1; k <= h; k++)
#pragma unroll
        for (int r = 0; r < 4; r++) m = __builtin_amdgcn_mfma_f64_16x16x4f64(Rl[16 * g + cc][16 * k + 4 * r + kk], Xl[16 * k + 4 * r + kk][16 * h + cc], m, 0, 0, 0);
      d4_t x = d4_t{0, 0, 0, 0};
#pragma unroll
      for (int r = 0; r < 4; r++) x = __builtin_amdgcn_mfma_f64_16x16x4f64(-Il[g][cc][4 * r + kk], m[r], x, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; r++) Xl[16 * g + kk + 4 * r][16 * h + cc] = x[r];
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  double* out = inv + (size_t)bb * 4096;
  for (int idx = lane; idx < 64 * 64; idx += 64) out[idx] = Xl[idx >> 6][idx & 63];
}

// R x = y, R = rows 0..nf-1 of the front (upper).  Workgroup b owns row block b:  it folds x_j (j > b) into its right-hand side
// as the blocks are published, then x_b = inv(R_bb) rhs and publishes x_b.  Hand-off: 8-byte agent-scope atomics for the payload and the flag on both sides
// (cdna_hip_programming.md Guideline 16, "8-B agent atomics both sides"), bounded spin.
template <int NB>
__global__ __launch_bounds__(256) void hbm_backsolve_dataflow_kernel(FrontDesc F, int64_t f_off, int ld, const int32_t* __restrict__ fxoff,
                                                                      const double* __restrict__ pool, const double* __restrict__ inv,
                                                                      const double* __restrict__ y, double* __restrict__ xbuf,
                                                                      unsigned int* __restrict__ flags, double* __restrict__ delta,
                                                                      int* __restrict__ status) {
  __shared__ double acc[NB];
  __shared__ double xs[NB];
  __shared__ int ok, s_ticket;
  const int nblk = gridDim.x;
  // logical order from a ticket (flags[nblk], zeroed with the flags): a workgroup only waits for workgroups that started before it
  if (threadIdx.x == 0) s_ticket = (int)atomicAdd(&flags[nblk], 1u);
  __syncthreads();
  const int b = nblk - 1 - s_ticket;  // the last row block (first to finish) goes to the first workgroup that starts
  const int r0 = b * NB, nb = min(NB, F.nf - r0);
  const int tid = threadIdx.x, row = tid >> 2, quarter = tid & 3;
  const double* A = pool + f_off;
  if (tid < NB) acc[tid] = (tid < nb) ? y[r0 + tid] : 0.0;
  if (tid == 0) ok = 1;
  __syncthreads();
  const double* arow = A + (size_t)(r0 + row) * ld;
  double ibv[16];  // this thread's part of inv(R_bb): independent of x, so not on the hop-to-hop chain
  {
    const double* ib = inv + (size_t)b * NB * NB + row * NB + quarter * 16;
#pragma unroll
    for (int k = 0; k < 16; k++) ibv[k] = ib[k];
  }
  // this thread's 16 entries of R[b rows, j cols], one block AHEAD of the poll: vector memory operations of a wave retire in
  // order, so a poll issued behind 16 strided row reads could not return before them (they were ~1 us of every hop)
  auto load_rv = [&](int j, double(&r)[16]) {
    const int c0 = j * NB, ncol = min(NB, F.nf - c0);
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int c = quarter * 16 + k;
      r[k] = (j > b && row < nb && c < ncol) ? arow[c0 + c] : 0.0;
    }
  };
  double rv[16], rv_next[16];
  load_rv(nblk - 1, rv);
  for (int j = nblk - 1; j > b; j--) {
    const int c0 = j * NB, ncol = min(NB, F.nf - c0);
    // the data is the flag: xbuf is preset to a sentinel bit pattern (all ones, a NaN no computation produces: the producer
    // canonicalises its NaNs); wave 0 polls its 64 values until none is the sentinel -- one round trip per hop instead of two
    if (tid < NB) {
      double v = 0.0;
      if (tid < ncol) {
        long spins = 0;
        for (;;) {
          v = __hip_atomic_load(&xbuf[c0 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__double_as_longlong(v) != -1LL) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spins > 2000000L) {
            ok = 0;
            v = 0.0;
            break;
          }
        }
      }
      xs[tid] = v;
    }
    load_rv(j - 1, rv_next);  // behind the poll in program order: in flight while x_j is folded in and x_{j-1} is awaited
    __syncthreads();
    if (!ok) break;
    double s = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) s += rv[k] * xs[quarter * 16 + k];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (quarter == 0) acc[row] -= s;
#pragma unroll
    for (int k = 0; k < 16; k++) rv[k] = rv_next[k];
    __syncthreads();
  }
  if (!ok) {
    if (tid == 0) atomicExch(status + 1, 1 + F.id);  // never expected: spin bound hit (a fault, reported apart from pivot failures)
    // still publish something so that waiters terminate
  }
  // x_b = inv(R_bb) acc   (thread (row, quarter): 16 columns of the row; the inverse was fetched before the first hop)
  double s = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) s += ibv[k] * acc[quarter * 16 + k];
  s += __shfl_xor(s, 1);
  s += __shfl_xor(s, 2);
  if (quarter == 0 && row < nb) {
    const double pub = (s != s) ? __longlong_as_double(0x7ff8000000000000LL) : s;  // never the sentinel
    __hip_atomic_store(&xbuf[r0 + row], pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    delta[fxoff[F.fx_begin + r0 + row]] = s;
    if (s != s) atomicMin(status, F.id);
  }
}

// ---- round 3: the same dataflow back-substitution with what sat on its hop-to-hop chain taken off it.
// (1) Workgroup b builds inv(R_bb) itself while it waits for its turn (from the 16 x 16 inverses the factorisation left behind, as
//     hbm_invert_diag64_from16_kernel does for all blocks in a launch of its own: 40 us in front of the chain before) and, with it,
//     M_b = inv(R_bb) R_{b,b+1}.  (2) The hop then is  x_b = u_b - M_b x_{b+1}  with  u_b = inv(R_bb) (y_b - sum_{j > b+1} R_bj x_j)
//     finished BEFORE x_{b+1} arrives: one 64 x 64 matrix-vector product behind the poll instead of the fold, an LDS round trip, a
//     barrier and the product with the inverse.  (3) y = d is read from the front's right-hand-side column when the front has no
//     separator (a root): no launch to copy it.  Measured on the C4 root (141 hops): see DESIGN.md section 6, round 3.
__global__ __launch_bounds__(256) void hbm_backsolve_dataflow2_kernel(FrontDesc F, int64_t f_off, int ld, const int32_t* __restrict__ fxoff,
                                                                       const double* __restrict__ pool, const double* __restrict__ inv16,
                                                                       const double* __restrict__ y /* nullptr: the rhs column itself */,
                                                                       double* __restrict__ xbuf, unsigned int* __restrict__ flags,
                                                                       double* __restrict__ delta, int* __restrict__ status) {
  typedef double d4_t __attribute__((ext_vector_type(4)));
  constexpr int NB = 64;
  __shared__ double Rl[64][65];   // R_bb, then R_{b,b+1}
  __shared__ double Xl[64][65];   // inv(R_bb)
  __shared__ double Ml[64][65];   // M_b
  __shared__ double Il[4][16][17];
  __shared__ double acc[NB], xs[NB], ub[NB];
  __shared__ int ok, s_ticket;
  const int nblk = gridDim.x;
  if (threadIdx.x == 0) s_ticket = (int)atomicAdd(&flags[nblk], 1u);
  __syncthreads();
  const int b = nblk - 1 - s_ticket;
  const int r0 = b * NB, nb = min(NB, F.nf - r0);
  const int tid = threadIdx.x, row = tid >> 2, quarter = tid & 3, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, cc = lane & 15;
  const double* A = pool + f_off;
  const int n = F.n;
  if (tid < NB) acc[tid] = (tid < nb) ? (y ? y[r0 + tid] : A[(size_t)(r0 + tid) * ld + n - 1]) : 0.0;
  if (tid == 0) ok = 1;
  // ---- inv(R_bb): identity-padded partial last block
  for (int idx = tid; idx < 64 * 64; idx += 256) {
    const int p = idx >> 6, q = idx & 63;
    double v = (p == q) ? 1.0 : 0.0;
    if (p < nb && q < nb && q >= p) v = A[(size_t)(r0 + p) * ld + r0 + q];
    Rl[p][q] = v;
    Xl[p][q] = 0.0;
  }
  const double* I16 = inv16 + (size_t)(b >> 2) * 4096 + (size_t)(4 * (b & 3)) * 256;
  for (int idx = tid; idx < 4 * 256; idx += 256) Il[idx >> 8][(idx >> 4) & 15][idx & 15] = I16[idx];
  __syncthreads();
  if (wave == 0) {  // block back-substitution on the matrix cores: X_hh = I_h;  X_gh = -I_g (sum_{k = g+1..h} R_gk X_kh)
#pragma unroll
    for (int h = 0; h < 4; h++) {
#pragma unroll
      for (int r = 0; r < 4; r++) Xl[16 * h + kk + 4 * r][16 * h + cc] = Il[h][kk + 4 * r][cc];
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int g = h - 1; g >= 0; g--) {
        d4_t m = d4_t{0, 0, 0, 0};
#pragma unroll
        for (int k = g + 1; k <= h; k++)
#pragma unroll
          for (int r = 0; r < 4; r++) m = __builtin_amdgcn_mfma_f64_16x16x4f64(Rl[16 * g + cc][16 * k + 4 * r + kk], Xl[16 * k + 4 * r + kk][16 * h + cc], m, 0, 0, 0);
        d4_t x = d4_t{0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 4; r++) x = __builtin_amdgcn_mfma_f64_16x16x4f64(-Il[g][cc][4 * r + kk], m[r], x, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) Xl[16 * g + kk + 4 * r][16 * h + cc] = x[r];
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  }
  __syncthreads();
  // ---- M_b = inv(R_bb) R_{b,b+1} (b < nblk - 1): the tile into Rl, one 16-row slice of the product per wave
  if (b + 1 < nblk) {
    const int c0n = (b + 1) * NB, ncoln = min(NB, F.nf - c0n);
    for (int idx = tid; idx < 64 * 64; idx += 256) {
      const int p = idx >> 6, q = idx & 63;
      Rl[p][q] = (p < nb && q < ncoln) ? A[(size_t)(r0 + p) * ld + c0n + q] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 4; h++) {  // rows 16 wave .., columns 16 h ..
      d4_t m = d4_t{0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 16; k++) m = __builtin_amdgcn_mfma_f64_16x16x4f64(Xl[16 * wave + cc][4 * k + kk], Rl[4 * k + kk][16 * h + cc], m, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; r++) Ml[16 * wave + kk + 4 * r][16 * h + cc] = m[r];
    }
  }
  __syncthreads();
  double ibv[16], mv[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    ibv[k] = Xl[row][quarter * 16 + k];
    mv[k] = Ml[row][quarter * 16 + k];
  }
  const double* arow = A + (size_t)(r0 + row) * ld;
  auto load_rv = [&](int j, double(&r)[16]) {
    const int c0 = j * NB, ncol = min(NB, F.nf - c0);
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int c = quarter * 16 + k;
      r[k] = (j > b + 1 && row < nb && c < ncol) ? arow[c0 + c] : 0.0;
    }
  };
  auto poll = [&](int j) {  // wave 0: the 64 values of x_j (the data is the flag: sentinel-preset buffer) into xs
    const int c0 = j * NB, ncol = min(NB, F.nf - c0);
    if (tid < NB) {
      double v = 0.0;
      if (tid < ncol) {
        long spins = 0;
        for (;;) {
          v = __hip_atomic_load(&xbuf[c0 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__double_as_longlong(v) != -1LL) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spins > 2000000L) {
            ok = 0;
            v = 0.0;
            break;
          }
        }
      }
      xs[tid] = v;
    }
  };
  double rv[16], rv_next[16];
  load_rv(nblk - 1, rv);
  for (int j = nblk - 1; j > b + 1; j--) {  // the folds that are not on the chain: every block but the neighbour
    poll(j);
    load_rv(j - 1, rv_next);
    __syncthreads();
    if (!ok) break;
    double s = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) s += rv[k] * xs[quarter * 16 + k];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (quarter == 0) acc[row] -= s;
#pragma unroll
    for (int k = 0; k < 16; k++) rv[k] = rv_next[k];
    __syncthreads();
  }
  // u_b = inv(R_bb) acc, before the neighbour's x is there
  {
    double s = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) s += ibv[k] * acc[quarter * 16 + k];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (quarter == 0) ub[row] = s;
  }
  double xr;
  if (b + 1 < nblk && ok) {
    poll(b + 1);
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) s += mv[k] * xs[quarter * 16 + k];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    xr = ub[row] - s;
  } else {
    __syncthreads();
    xr = ub[row];
  }
  if (!ok && tid == 0) atomicExch(status + 1, 1 + F.id);  // never expected: spin bound hit (a fault, reported apart from pivot failures)
  if (quarter == 0 && row < nb) {
    const double pub = (xr != xr) ? __longlong_as_double(0x7ff8000000000000LL) : xr;  // never the sentinel
    __hip_atomic_store(&xbuf[r0 + row], pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    delta[fxoff[F.fx_begin + r0 + row]] = xr;
    if (xr != xr) atomicMin(status, F.id);
  }
}

}  // namespace lmgpu
