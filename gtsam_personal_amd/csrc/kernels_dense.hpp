// HBM fronts: fronts too large for one workgroup's LDS live in HBM (row-major upper, leading dimension ld)
// and are processed by a two-level blocked right-looking partial Cholesky:
//   outer panel = 256 rows:   4 x [ potrf (64x64 diagonal block, LDS) + trsm (row panel)      -> potrf_trsm_kernel
//                                   + update of the remaining rows of the outer panel (K = 64)  -> syrk_mfma_kernel ]
//                             then ONE trailing update of everything below with K = 256          -> syrk_mfma_kernel
// so the big trailing matrix is read-modified-written once per 256 eliminated rows (HBM traffic / 4 vs. 64-row steps).
// This is choleskyPartial (gtsam/base/cholesky.cpp:108-159: LLT(A); S = R^-T B; C -= S^T S; pivot-exponent test)
// in blocked form; the root of a BAL problem (all cameras, 9001 x 9001) spends >90 % of the solve here.
// Assembly (a11/a12) into an HBM front uses FP64 global atomics (children and factors scatter concurrently).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_front.hpp"

namespace lmgpu {

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- assembly
// one 64-lane block per own factor: F += [A b]^T [A b]
__global__ __launch_bounds__(64) void hbm_assemble_factors_kernel(FrontDesc F, int64_t f_off, int ld, const FrontFac* __restrict__ ffac,
                                                                   const FacDesc* __restrict__ fd, double* __restrict__ pool) {
  const FrontFac ff = ffac[F.fac_begin + blockIdx.x];
  const FacDesc d = fd[ff.fac];
  const double* J = pool + d.joff;
  double* A = pool + f_off;
  const int n = F.n, m = d.rows, nc = d.d0 + d.d1 + 1;
  const int npair = nc * (nc + 1) / 2;
  for (int pidx = threadIdx.x; pidx < npair; pidx += 64) {
    int p = 0, rem = pidx, rowlen = nc;
    while (rem >= rowlen) {
      rem -= rowlen;
      rowlen--;
      p++;
    }
    const int q = p + rem;
    double v = 0;
    for (int r = 0; r < m; r++) v += J[p * m + r] * J[q * m + r];
    const int gp = (p < d.d0) ? ff.c0 + p : (p < d.d0 + d.d1 ? ff.c1 + (p - d.d0) : n - 1);
    const int gq = (q < d.d0) ? ff.c0 + q : (q < d.d0 + d.d1 ? ff.c1 + (q - d.d0) : n - 1);
    const int lo = gp < gq ? gp : gq, hi = gp < gq ? gq : gp;
    atomicAdd(&A[(size_t)lo * ld + hi], v);
  }
}

// one block per child: extend-add of its update matrix
__global__ __launch_bounds__(256) void hbm_assemble_children_kernel(FrontDesc F, int64_t f_off, int ld, const ChildRef* __restrict__ childs,
                                                                    const int32_t* __restrict__ cmap, double* __restrict__ pool) {
  __shared__ int32_t smap[160];
  const ChildRef c = childs[F.child_begin + blockIdx.x];
  const double* U = pool + c.u_off;
  const int32_t* map = cmap + c.map_begin;
  double* A = pool + f_off;
  const bool small = c.m <= 160;
  if (small) {
    for (int i = threadIdx.x; i < c.m; i += 256) smap[i] = map[i];
    __syncthreads();
  }
  for (int i = threadIdx.x >> 6; i < c.m; i += 4) {  // one wave per row: row i of U is contiguous
    const int gi = small ? smap[i] : map[i];
    const double* Ui = U + (size_t)i * c.ld;
    for (int j = i + (threadIdx.x & 63); j < c.m; j += 64) {
      const int gj = small ? smap[j] : map[j];
      const int lo = gi < gj ? gi : gj, hi = gi < gj ? gj : gi;
      atomicAdd(&A[(size_t)lo * ld + hi], Ui[j]);
    }
  }
}

__global__ __launch_bounds__(256) void hbm_damp_kernel(FrontDesc F, int64_t f_off, int ld, const int32_t* __restrict__ fxoff,
                                                        double* __restrict__ pool, double lambda, const double* __restrict__ dampw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= F.nf) return;
  pool[f_off + (size_t)i * ld + i] += lambda * dampw[fxoff[F.fx_begin + i]];
}

// ---------------------------------------------------------------- panel: potrf + trsm
// potrf of one 64x64 diagonal block by 256 lanes with ONE barrier per pivot.  Lane t owns column b = t % 64 and the
// rows a = 4 i + t / 64 (i = 0..15) of it, in registers.  Step k (fully unrolled, so every register index is static):
// the wave that owns row k publishes it (unscaled) through a double-buffered LDS row; every lane then applies
//   A'[a][b] -= A'[k][a] * (A'[k][b] / p_k)     to its rows a > k
// reading A'[k][a] as a wave-uniform (broadcast) LDS word.  R[a][b] = A'[a][b] / sqrt(p_a) at the end.
// Slots below the diagonal (a > b) carry don't-care values that are never stored.  Blocks smaller than 64 are
// identity-padded.
#define POTRF_LDW 66  // row stride of the factor in LDS: even, so that 16-byte ds_read_b128 pairs are aligned
__device__ __forceinline__ double fast_rcp(double p) {
  double r = __builtin_amdgcn_rcp(p);
  r = fma(fma(-p, r, 1.0), r, r);
  r = fma(fma(-p, r, 1.0), r, r);
  return r;
}

template <int NB>
__device__ __forceinline__ bool potrf64_lds(double (*D)[POTRF_LDW], double (*rowbuf)[NB], double* piv) {
  static_assert(NB == 64, "register distribution below assumes a 64x64 block and 256 lanes");
  const int tid = threadIdx.x, b = tid & 63, w = tid >> 6;
  double v[16];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = D[4 * i + w][b];  // the block (upper triangle valid, identity-padded) is already in LDS
  __syncthreads();
  bool failed = false;
#pragma unroll
  for (int k = 0; k < NB; k++) {
    constexpr int dummy = 0;
    (void)dummy;
    const int ks = k >> 2, kw = k & 3;
    double* row = rowbuf[k & 1];
    if (w == kw) row[b] = v[ks];
    __syncthreads();
    double p = row[k];
    if (!(p > 0.0)) {
      if (p <= 0.0) failed = true;  // Eigen LLT: pivot <= 0 -> failure (NaN passes like Eigen)
      p = (p == p && p != 0.0) ? fabs(p) : 1.0;
    }
    if (tid == 0) piv[k] = p;
    const double rb = row[b] * fast_rcp(p);
    if (w > kw) v[ks] -= row[4 * ks + w] * rb;
#pragma unroll
    for (int i = ks + 1; i < 16; i++) v[i] -= row[4 * i + w] * rb;
  }
  __syncthreads();
  if (tid < NB) piv[tid] = 1.0 / sqrt(piv[tid]);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int a = 4 * i + w;
    if (a <= b) D[a][b] = v[i] * piv[a];
  }
  __syncthreads();
  return failed;
}

// One 64-row step of the panel factorisation, LEFT-LOOKING inside the current 256-row outer panel [ko, ko+256):
// rows [k, k+nb) have received the trailing updates of all previous OUTER panels, but not yet those of the inner
// steps ko..k of this outer panel (kprev = k - ko finished rows).  Every workgroup (4 waves, 64 columns) does:
//   1. stage  Pj = R[ko..k, k..k+64)  (kprev x 64, <= 96 KB) in LDS
//   2. D = A[k.., k..] - Pj^T Pj  on the matrix cores (redundantly per workgroup), potrf64 in registers/LDS,
//      invert the four 16x16 diagonal blocks; workgroup 0 writes R_kk back
//   3. its 64 columns:  T = A[k.., cols] - Pj^T R[ko..k, cols]   (MFMA, K = kprev),  X_g = inv(R_gg)^T (T_g - sum_{i<g} R_ig^T X_i)
//      with finished X tiles fed back as B operands straight from the accumulator registers (f64 C/D layout
//      row = (lane>>4) + 4*reg, col = lane&15  ==  B layout of k-step `reg`).
// This removes the separate K = 64 "strip" updates (one launch + one pass over the panel rows each).
#define PANEL_MAXPREV 192
template <int NB>
__global__ __launch_bounds__(256) void panel_fused_kernel(double* __restrict__ A, int ld, int n, int nf, int ko, int k, int nb, int front_id,
                                                           int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) double psm[];
  double(*D)[POTRF_LDW] = (double(*)[POTRF_LDW])psm;                 // [NB][POTRF_LDW]
  double(*rowbuf)[NB] = (double(*)[NB])(psm + NB * POTRF_LDW);       // [2][NB]
  double* piv = psm + NB * POTRF_LDW + 2 * NB;                        // [NB]
  double(*I16)[16][17] = (double(*)[16][17])(piv + NB);               // [4][16][17]
  double(*Pj)[NB + 2] = (double(*)[NB + 2])(piv + NB + 4 * 16 * 17);  // [kprev][NB + 2]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kk = lane >> 4, cc = lane & 15;
  const int kprev = k - ko;
  const double* Pko = A + (size_t)ko * ld;
  // ---- 1. stage Pj (columns k..k+63 of the finished rows of this outer panel); columns >= k+nb are never used un-masked
  for (int idx = tid; idx < kprev * NB; idx += 256) {
    const int q = idx >> 6, c = idx & 63;
    Pj[q][c] = (c < nb) ? Pko[(size_t)q * ld + k + c] : 0.0;
  }
  __syncthreads();
  // ---- 2. updated diagonal block: wave w computes tile row w (tiles (w, bt), bt >= w), writes it into D (identity-padded)
  {
    double4_t acc[4];
#pragma unroll
    for (int bt = 0; bt < 4; bt++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int a = 16 * wave + kk + 4 * r, b = 16 * bt + cc;
        const double v = A[(size_t)(k + min(a, nb - 1)) * ld + k + min(b, nb - 1)];  // branch-free: clamped address + select
        acc[bt][r] = (a < nb && b < nb) ? v : ((a == b) ? 1.0 : 0.0);
      }
    }
    for (int q = 0; q < kprev; q += 4) {
      const double af = -Pj[q + kk][16 * wave + cc];
#pragma unroll
      for (int bt = 0; bt < 4; bt++) {
        const double bf = Pj[q + kk][16 * bt + cc];
        acc[bt] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc[bt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int bt = 0; bt < 4; bt++)
#pragma unroll
      for (int r = 0; r < 4; r++) D[16 * wave + kk + 4 * r][16 * bt + cc] = acc[bt][r];
  }
  __syncthreads();
  bool failed = potrf64_lds<NB>(D, rowbuf, piv);
  if (blockIdx.x == 0) {
    for (int idx = tid; idx < nb * nb; idx += 256) {
      const int p = idx / nb, q = idx - p * nb;
      if (q >= p) A[(size_t)(k + p) * ld + k + q] = D[p][q];
    }
    if (tid == 0) {
      if (k + nb >= nf) {  // last panel: pivot-exponent test, gtsam/base/cholesky.cpp:146-158
        if (nf >= 2) {
          const double r1 = D[nb - 1][nb - 1];
          const double r2 = (nb >= 2) ? D[nb - 2][nb - 2] : A[(size_t)(nf - 2) * ld + nf - 2];
          if (!(frexp_exp(r2) - frexp_exp(r1) < 12)) failed = true;
        } else {
          if (!(frexp_exp(D[0][0]) > -12)) failed = true;
        }
      }
      if (failed) atomicMin(status, front_id);
    }
  }
  // inverses of the four 16x16 diagonal blocks: lane (blk, j) back-substitutes column j of inv(R_blk)
  if (tid < 64) {
    const int blk = tid >> 4, j = tid & 15, base = 16 * blk;
    double x[16];
#pragma unroll
    for (int i = 15; i >= 0; i--) {
      double sacc = (i == j) ? 1.0 : 0.0;
#pragma unroll
      for (int kq = i + 1; kq < 16; kq++) sacc -= D[base + i][base + kq] * x[kq];
      x[i] = (i <= j) ? sacc / D[base + i][base + i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) I16[blk][i][j] = x[i];
  }
  __syncthreads();
  // ---- 3. this wave's 16 columns of the row panel
  const int c0 = k + nb + blockIdx.x * 64 + wave * 16;
  if (c0 >= n) return;
  const int col = min(c0 + cc, n - 1);
  double* P = A + (size_t)k * ld;
  double4_t T[4];
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = 16 * g + kk + 4 * r;
      const double v = P[(size_t)min(row, nb - 1) * ld + col];
      T[g][r] = (row < nb) ? v : 0.0;
    }
  // left-looking update with the finished rows of this outer panel: B fragments straight from HBM/L2, 16 k-steps in flight
  for (int q0 = 0; q0 < kprev; q0 += 64) {
    double bq[16];
#pragma unroll
    for (int sx = 0; sx < 16; sx++) bq[sx] = Pko[(size_t)(q0 + 4 * sx + kk) * ld + col];
#pragma unroll
    for (int sx = 0; sx < 16; sx++) {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const double af = -Pj[q0 + 4 * sx + kk][16 * g + cc];
        T[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bq[sx], T[g], 0, 0, 0);
      }
    }
  }
  double4_t X[4];
#pragma unroll
  for (int g = 0; g < 4; g++) {
    double4_t acc = T[g];
#pragma unroll
    for (int i = 0; i < g; i++)
#pragma unroll
      for (int sx = 0; sx < 4; sx++) {
        const double af = -D[16 * i + 4 * sx + kk][16 * g + cc];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, X[i][sx], acc, 0, 0, 0);
      }
    double4_t out = double4_t{0, 0, 0, 0};
#pragma unroll
    for (int sx = 0; sx < 4; sx++) out = __builtin_amdgcn_mfma_f64_16x16x4f64(I16[g][4 * sx + kk][cc], acc[sx], out, 0, 0, 0);
    X[g] = out;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = 16 * g + kk + 4 * r;
      if (row < nb && c0 + cc < n) P[(size_t)row * ld + c0 + cc] = out[r];
    }
  }
}
#define PANEL_LDS_BYTES ((64 * POTRF_LDW + 2 * 64 + 64 + 4 * 16 * 17 + PANEL_MAXPREV * 66) * 8)

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

__device__ __forceinline__ void syrk_tile(double* __restrict__ A, int ld, int n, int p0, int kp, int r0, int r1, int ti, int tj,
                                          double* sm /* [2 stages][2 operands][SYRK_KC][SYRK_LDW] */) {
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
        for (int a = 0; a < 4; a++) af[a] = -sA[(ks + kk) * SYRK_LDW + a * 16];
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
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  const int i0 = it0 + wr * 64, j0 = jt0 + wc * 64;
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = i0 + a * 16 + kk + 4 * r;
        const int col = j0 + b * 16 + cc;
        if (row < r1 && col < n && col >= row) A[(size_t)row * ld + col] += acc[a][b][r];
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

// one 64-row block step of the backward solve R x = y:  every workgroup solves the diagonal block in LDS,
// workgroup 0 publishes x, and each wave folds x into 8 rows above:  y_row -= R[row, r0:r0+nb] x
template <int NB>
__global__ __launch_bounds__(256) void hbm_backsolve_step_kernel(FrontDesc F, int64_t f_off, int ld, int r0, int nb,
                                                                  const int32_t* __restrict__ fxoff, const double* __restrict__ pool,
                                                                  double* __restrict__ y, double* __restrict__ delta, int* __restrict__ status) {
  __shared__ double T[NB][NB + 1];
  __shared__ double x[NB];
  const int tid = threadIdx.x;
  const double* A = pool + f_off;
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int p = idx / NB, q = idx - p * NB;
    T[p][q] = (p < nb && q < nb && q >= p) ? A[(size_t)(r0 + p) * ld + r0 + q] : 0.0;
  }
  if (tid < NB) x[tid] = (tid < nb) ? y[r0 + tid] : 0.0;
  __syncthreads();
  for (int i = nb - 1; i >= 0; i--) {
    if (tid == 0) x[i] = x[i] / T[i][i];
    __syncthreads();
    if (tid < i) x[tid] -= T[tid][i] * x[i];
    __syncthreads();
  }
  if (blockIdx.x == 0 && tid < nb) {
    const double v = x[tid];
    delta[fxoff[F.fx_begin + r0 + tid]] = v;
    if (v != v) atomicMin(status, F.id);
  }
  // rows above
  const int wave = tid >> 6, lane = tid & 63;
  const double xl = x[lane];
  for (int rr = 0; rr < 8; rr++) {
    const int row = blockIdx.x * 32 + wave * 8 + rr;
    if (row >= r0) break;
    double s = (lane < nb) ? A[(size_t)row * ld + r0 + lane] * xl : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) y[row] -= s;
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

// R x = y, R = rows 0..nf-1 of the front (upper).  Workgroup b owns row block b:  it folds x_j (j > b) into its right-hand side
// as the blocks are published, then x_b = inv(R_bb) rhs, publishes x_b and raises flag[b].  All workgroups are co-resident
// (grid <= #CUs, checked on the host).  Hand-off: 8-byte agent-scope atomics for the payload and the flag on both sides
// (cdna_hip_programming.md Guideline 16, "8-B agent atomics both sides"), bounded spin.
template <int NB>
__global__ __launch_bounds__(256) void hbm_backsolve_dataflow_kernel(FrontDesc F, int64_t f_off, int ld, const int32_t* __restrict__ fxoff,
                                                                      const double* __restrict__ pool, const double* __restrict__ inv,
                                                                      const double* __restrict__ y, double* __restrict__ xbuf,
                                                                      unsigned int* __restrict__ flags, double* __restrict__ delta,
                                                                      int* __restrict__ status) {
  __shared__ double acc[NB];
  __shared__ double xs[NB];
  __shared__ int ok;
  const int nblk = gridDim.x;
  const int b = nblk - 1 - blockIdx.x;  // the last row block (first to finish) gets the first-dispatched workgroup
  const int r0 = b * NB, nb = min(NB, F.nf - r0);
  const int tid = threadIdx.x, row = tid >> 2, quarter = tid & 3;
  const double* A = pool + f_off;
  if (tid < NB) acc[tid] = (tid < nb) ? y[r0 + tid] : 0.0;
  if (tid == 0) ok = 1;
  __syncthreads();
  const double* arow = A + (size_t)(r0 + row) * ld;
  for (int j = nblk - 1; j > b; j--) {
    const int c0 = j * NB, ncol = min(NB, F.nf - c0);
    // prefetch this thread's 16 entries of R[b rows, j cols] while the producer is still working
    double rv[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int c = quarter * 16 + k;
      rv[k] = (row < nb && c < ncol) ? arow[c0 + c] : 0.0;
    }
    if (tid == 0) {
      long spins = 0;
      while (__hip_atomic_load(&flags[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > 20000000L) {
          ok = 0;
          break;
        }
      }
    }
    __syncthreads();
    if (!ok) break;
    if (tid < NB) xs[tid] = __hip_atomic_load(&xbuf[c0 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) s += rv[k] * xs[quarter * 16 + k];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (quarter == 0) acc[row] -= s;
    __syncthreads();
  }
  if (!ok) {
    if (tid == 0) atomicMin(status, F.id);  // never expected: spin bound hit
    // still publish something so that waiters terminate
  }
  // x_b = inv(R_bb) acc   (thread (row, quarter): 16 columns of the row)
  const double* ib = inv + (size_t)b * NB * NB;
  double s = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int c = quarter * 16 + k;
    s += ib[row * NB + c] * acc[c];
  }
  s += __shfl_xor(s, 1);
  s += __shfl_xor(s, 2);
  if (quarter == 0 && row < nb) {
    __hip_atomic_store(&xbuf[r0 + row], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    delta[fxoff[F.fx_begin + r0 + row]] = s;
    if (s != s) atomicMin(status, F.id);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_store(&flags[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace lmgpu
