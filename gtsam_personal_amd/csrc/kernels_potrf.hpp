// Single-wave Cholesky of a 64x64 diagonal block held entirely in registers, in the accumulator layout of
// v_mfma_f64_16x16x4_f64: tile T[g][h] (16x16, h >= g) keeps element (row 16g + (lane>>4) + 4r, column 16h + (lane&15))
// in register r.  No LDS and no barriers.  A 16-row strip is eliminated four pivots at a time (potrf64_wave_g4 below), its
// off-diagonal tiles come out already solved (R_gh = R_gg^-T A_gh: no explicit triangular inverse), and the remaining strips
// are updated on the matrix cores with both operands taken straight from accumulator registers (C/D layout of k-step s  ==
// A/B operand layout:  A[i = lane&15][k = lane>>4]).  A first version took the pivots one at a time with ds_bpermute shuffles
// of the pivot row (9.5-11 us per block in isolation, slower inside the fused step launch where other waves share the LDS pipe).
// Arithmetic = Eigen::LLT on the block (gtsam/base/cholesky.cpp:108-159) up to rounding; a pivot <= 0 reports failure.
#pragma once
#include <hip/hip_runtime.h>

namespace lmgpu {

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_d(double v, int src_lane) {  // src_lane: compile-time constant
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), src_lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double fast_rsqrt(double p) {
  double y = __builtin_amdgcn_rsq(p);
  const double h = 0.5 * p;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  return y;
}

// T[g][h], h >= g: in = upper triangle of the SPD block (entries below the diagonal inside diagonal tiles: don't care),
// out = R with R^T R = A.  Returns true if a pivot was <= 0 (the factor is then garbage, like Eigen's info() != Success).
// The pivots are taken FOUR at a time.  A group = the four rows held by register q of a strip.  Its 4x4
// diagonal mini-block is read with v_readlane (10 values) and factored by every lane redundantly in plain scalar-like code
// (R4 = chol(M), W = R4^-1); the group's four rows of every tile of the strip are then solved by ONE MFMA per tile
// (D = W^T [rows], A operand built from the ten W values by lane selects) and the rows below them in the strip updated by one
// more (C -= [rows]^T [rows], A operand = the solved rows of the diagonal tile, masked to the rows still to do).  No cross-lane
// shuffles, and 16 dependent steps per 64x64 block instead of 64.
// INV: E[g] leaves as R_gg^-T of the g-th 16x16 diagonal tile (an identity tile appended to the strip and taken through the same
// solves and updates), i.e. register r of lane (kk, cc) = Inv_g[cc][kk + 4 r] -- the 16x16 inverses the tile solves of the
// panel kernels multiply with, at the price of two more MFMAs per group instead of a 16-step substitution after the factorisation.
// nrows (wave-uniform): the block's real rows; the rest is identity padding, whose groups of four pivots are skipped (they would
// factor an identity against zeros: ~0.55 us per group, half of the 8.8 us of a 64-row block for a 30-column front)
template <bool INV>
__device__ __forceinline__ bool potrf64_wave_g4(double4_t (&T)[4][4], double4_t (&E)[4], int nrows = 64) {
  const int lane = threadIdx.x & 63, kk = lane >> 4, cc = lane & 15;
  bool failed = false;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    if (INV) {
#pragma unroll
      for (int r = 0; r < 4; r++) E[g][r] = (kk + 4 * r == cc) ? 1.0 : 0.0;
    }
    if (16 * g >= nrows) continue;  // an identity strip: R = I, inverse = I, nothing below it to update
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (16 * g + 4 * q >= nrows) continue;  // identity rows inside the strip: already R = I with zeros beside them
      // M[a][b], a <= b: lane (kk = a, cc = 4q + b) of register q of the diagonal tile
      double m[4][4];
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = a; b < 4; b++) m[a][b] = readlane_d(T[g][g][q], 16 * a + 4 * q + b);
      double R[4][4], inv[4];
#pragma unroll
      for (int a = 0; a < 4; a++) {
        double p = m[a][a];
#pragma unroll
        for (int k = 0; k < a; k++) p -= R[k][a] * R[k][a];
        const bool pok = p > 0.0;  // one select, no branch: after a failure the factor is garbage either way
        failed |= !pok;
        p = pok ? p : 1.0;
        const double rs = fast_rsqrt(p);
        inv[a] = rs;
        R[a][a] = p * rs;
#pragma unroll
        for (int b = a + 1; b < 4; b++) {
          double v = m[a][b];
#pragma unroll
          for (int k = 0; k < a; k++) v -= R[k][a] * R[k][b];
          R[a][b] = v * rs;
        }
      }
      // W = R^-1 (upper): W[a][a] = 1 / R[a][a];  W[a][b] = -(sum_{k=a}^{b-1} W[a][k] R[k][b]) / R[b][b]
      double W[4][4];
#pragma unroll
      for (int a = 0; a < 4; a++) {
        W[a][a] = inv[a];
#pragma unroll
        for (int b = a + 1; b < 4; b++) {
          double v = 0;
#pragma unroll
          for (int k = a; k < b; k++) v += W[a][k] * R[k][b];
          W[a][b] = -v * inv[b];
        }
      }
      // A operand of the solve: A[i = cc][k = kk] = W[k][i]  (i < 4, k <= i), zero elsewhere
      double aop = 0.0;
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int i = k; i < 4; i++) aop = (kk == k && cc == i) ? W[k][i] : aop;
      // the diagonal tile first: the next group's mini-block only waits for these two
      double bop = 0.0;
      {
        const double4_t x = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, T[g][g][q], double4_t{0, 0, 0, 0}, 0, 0, 0);
        T[g][g][q] = x[0];  // rows i = 0..3 of D = register 0, lane (kk = i, cc): the layout of register q
        if (q < 3) {        // rows of the strip still to do: A[i = cc][k = kk] = R_new[k][i] for i >= 4 (q + 1)
          bop = (cc >= 4 * (q + 1)) ? -T[g][g][q] : 0.0;
          T[g][g] = __builtin_amdgcn_mfma_f64_16x16x4f64(bop, T[g][g][q], T[g][g], 0, 0, 0);
        }
      }
#pragma unroll
      for (int h = g + 1; h < 4; h++) {
        const double4_t x = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, T[g][h][q], double4_t{0, 0, 0, 0}, 0, 0, 0);
        T[g][h][q] = x[0];
        if (q < 3) T[g][h] = __builtin_amdgcn_mfma_f64_16x16x4f64(bop, T[g][h][q], T[g][h], 0, 0, 0);
      }
      if (INV) {
        const double4_t x = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, E[g][q], double4_t{0, 0, 0, 0}, 0, 0, 0);
        E[g][q] = x[0];
        if (q < 3) E[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(bop, E[g][q], E[g], 0, 0, 0);
      }
    }
#pragma unroll
    for (int g2 = g + 1; g2 < 4; g2++)
#pragma unroll
      for (int h = g2; h < 4; h++)
#pragma unroll
        for (int s = 0; s < 4; s++) T[g2][h] = __builtin_amdgcn_mfma_f64_16x16x4f64(-T[g][g2][s], T[g][h][s], T[g2][h], 0, 0, 0);
  }
  return failed;
}
__device__ __forceinline__ bool potrf64_wave_g4(double4_t (&T)[4][4]) {
  double4_t E[4];
  return potrf64_wave_g4<false>(T, E);
}

// ---------------------------------------------------------------- one 256-row outer panel of an HBM front, two launches
// diag_potrf_kernel  (ONE workgroup): Cholesky of the kb x kb diagonal block A[ko.., ko..] (kb <= 256), right-looking over
//   64-blocks: wave 0 factors the 64x64 diagonal tile in registers (potrf64_wave_g4), all four waves then solve the tiles to
//   its right on the matrix cores (R_j,jj = R_jj^-T A_j,jj through the four 16x16 triangular inverses) and update the
//   remaining tiles of the block with both operands from LDS / accumulator registers.  Every lane re-reads from global
//   memory only what the same lane wrote (wave w <-> columns 16w..16w+15 of every tile), so the block needs no device-scope
//   fences.  Also writes the sixteen 16x16 inverses of the diagonal tiles (inv16, by-product of the register Cholesky) for the panel solve.
// panel_trsm_kernel  (one wave per 16 columns): the row panel right of the diagonal block,
//   X_j = R_jj^-T (A_j,cols - sum_{i<j} R_ij^T X_i), all 256 rows of the wave's 16 columns held in accumulator registers
//   (finished X tiles are fed back as B operands straight from those registers).
// Together they replace four launches of a fused 64-row panel step in which EVERY workgroup re-factored the diagonal tile.
#define DP_LDW 66
#define DIAG_LDS_DOUBLES (64 * DP_LDW + 4 * 16 * 17 + 3 * 64 * DP_LDW)
#define DIAG_LDS_BYTES (DIAG_LDS_DOUBLES * 8)

__device__ __forceinline__ int frexp_exp_d(double x) {
  int e;
  frexp(x, &e);
  return e;
}

// X = R_jj^-T T for one wave's 16 columns (4 row tiles), D = R_jj (LDS, identity-padded), I16 = inverses of its diagonal 16x16 tiles
__device__ __forceinline__ void trsm64_wave(double4_t (&T)[4], const double (*D)[DP_LDW], const double (*I16)[16][17], int kk, int cc) {
#pragma unroll
  for (int g = 0; g < 4; g++) {
    double4_t acc = T[g];
#pragma unroll
    for (int i = 0; i < g; i++)
#pragma unroll
      for (int sx = 0; sx < 4; sx++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-D[16 * i + 4 * sx + kk][16 * g + cc], T[i][sx], acc, 0, 0, 0);
    double4_t out = double4_t{0, 0, 0, 0};
#pragma unroll
    for (int sx = 0; sx < 4; sx++) out = __builtin_amdgcn_mfma_f64_16x16x4f64(I16[g][4 * sx + kk][cc], acc[sx], out, 0, 0, 0);
    T[g] = out;
  }
}

__device__ __forceinline__ void diag_potrf_body(double* __restrict__ A, int ld, int nf, int ko, int kb, int front_id,
                                                int* __restrict__ status, double* __restrict__ inv16, double* dsm) {
  double(*D)[DP_LDW] = (double(*)[DP_LDW])dsm;                                   // [64][DP_LDW]   R_jj
  double(*I16)[16][17] = (double(*)[16][17])(dsm + 64 * DP_LDW);                // [4][16][17]
  double(*XB)[64][DP_LDW] = (double(*)[64][DP_LDW])(dsm + 64 * DP_LDW + 4 * 16 * 17);  // [3][64][DP_LDW]  R_j,jj  (jj = j+1 ..)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kk = lane >> 4, cc = lane & 15;
  const int nblk = (kb + 63) >> 6;
  double* Ab = A + (size_t)ko * ld + ko;
  const int wc = 16 * wave + cc;  // this lane's column inside every 64-wide tile
  for (int j = 0; j < nblk; j++) {
    const int nbj = min(64, kb - 64 * j);
    double* Aj = Ab + (size_t)(64 * j) * ld;  // row block j
    // ---- 1. diagonal tile -> D (identity-padded), each wave its own 16 columns
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 16 * g + kk + 4 * r;
        const double v = Aj[(size_t)min(row, nbj - 1) * ld + 64 * j + min(wc, nbj - 1)];
        D[row][wc] = (row < nbj && wc < nbj) ? v : ((row == wc) ? 1.0 : 0.0);
      }
    __syncthreads();
    // ---- 2. wave 0 factors it in registers
    if (wave == 0) {
      double4_t T[4][4];
#pragma unroll
      for (int g = 0; g < 4; g++)
#pragma unroll
        for (int h = g; h < 4; h++)
#pragma unroll
          for (int r = 0; r < 4; r++) T[g][h][r] = D[16 * g + kk + 4 * r][16 * h + cc];
      double4_t E[4];
      bool failed = potrf64_wave_g4<true>(T, E, nbj);
#pragma unroll
      for (int g = 0; g < 4; g++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
          for (int h = g; h < 4; h++) D[16 * g + kk + 4 * r][16 * h + cc] = T[g][h][r];
          // the 16x16 inverses come out of the factorisation (E = R_gg^-T; identity padding inverts to identity)
          I16[g][cc][kk + 4 * r] = E[g][r];
          inv16[(size_t)(4 * j + g) * 256 + cc * 16 + kk + 4 * r] = E[g][r];
        }
      if (ko + 64 * j + nbj >= nf) {  // last rows of the frontal part: pivot-exponent test, gtsam/base/cholesky.cpp:146-158
        // R[nf-1][nf-1] is element (nbj-1, nbj-1) of this tile: tile (3,3) if nbj == 64, else read back below
        __builtin_amdgcn_s_waitcnt(0);
        const double r1 = D[nbj - 1][nbj - 1];
        if (nf >= 2) {
          const double r2 = (nbj >= 2) ? D[nbj - 2][nbj - 2] : A[(size_t)(nf - 2) * ld + nf - 2];
          if (!(frexp_exp_d(r2) - frexp_exp_d(r1) < 12)) failed = true;
        } else {
          if (!(frexp_exp_d(r1) > -12)) failed = true;
        }
      }
      if (failed && lane == 0) atomicMin(status, front_id);
    }
    __syncthreads();
    // ---- 3. R_jj -> global
    for (int idx = tid; idx < nbj * 64; idx += 256) {
      const int p = idx >> 6, q = idx & 63;
      if (q >= p && q < nbj) Aj[(size_t)p * ld + 64 * j + q] = D[p][q];
    }
    __syncthreads();
    if (j + 1 == nblk) break;
    // ---- 4. tiles to the right of the diagonal one (inside the block)
    double4_t X[3][4];
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const int jj = j + 1 + t;
      if (jj < nblk) {
        const int col = 64 * jj + wc;  // column inside the block
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int row = 16 * g + kk + 4 * r;
            const double v = Aj[(size_t)min(row, nbj - 1) * ld + min(col, kb - 1)];
            X[t][g][r] = (row < nbj && col < kb) ? v : 0.0;
          }
        trsm64_wave(X[t], D, I16, kk, cc);
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int row = 16 * g + kk + 4 * r;
            XB[t][row][wc] = X[t][g][r];
            if (row < nbj && col < kb) Aj[(size_t)row * ld + col] = X[t][g][r];
          }
      }
    }
    __syncthreads();
    // ---- 5. update the remaining tiles (i, jj), j < i <= jj:  C -= R_j,i^T R_j,jj   (this wave's 16 columns of each)
#pragma unroll
    for (int ti = 0; ti < 3; ti++)
#pragma unroll
      for (int tj = ti; tj < 3; tj++) {
        const int i = j + 1 + ti, jj = j + 1 + tj;
        if (jj < nblk) {
          const int nbi = min(64, kb - 64 * i), col = 64 * jj + wc;
          double* Ai = Ab + (size_t)(64 * i) * ld;
          double4_t C[4];
#pragma unroll
          for (int g = 0; g < 4; g++)
#pragma unroll
            for (int r = 0; r < 4; r++) C[g][r] = Ai[(size_t)min(16 * g + kk + 4 * r, nbi - 1) * ld + min(col, kb - 1)];
#pragma unroll
          for (int s = 0; s < 16; s++)
#pragma unroll
            for (int g = 0; g < 4; g++)
              C[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(-XB[ti][4 * s + kk][16 * g + cc], X[tj][s >> 2][s & 3], C[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; g++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int row = 16 * g + kk + 4 * r;
              if (row < nbi && col < kb) Ai[(size_t)row * ld + col] = C[g][r];
            }
        }
      }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void diag_potrf_kernel(double* __restrict__ A, int ld, int nf, int ko, int kb, int front_id,
                                                          int* __restrict__ status, double* __restrict__ inv16) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  diag_potrf_body(A, ld, nf, ko, kb, front_id, status, inv16, dsm);
}

// grid = ceil(cols / 64) workgroups of 4 independent waves; cols = n - ko - kb columns right of the diagonal block
#define PTRSM_LDS_BYTES (16 * 16 * 17 * 8)
__device__ __forceinline__ void panel_trsm_body(double* __restrict__ A, int ld, int n, int ko, int kb, const double* __restrict__ inv16,
                                                double (*I16)[16][17], int bx) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kk = lane >> 4, cc = lane & 15;
  const int nblk = (kb + 63) >> 6;
  for (int idx = tid; idx < nblk * 4 * 256; idx += 256) I16[idx >> 8][(idx >> 4) & 15][idx & 15] = inv16[idx];
  __syncthreads();
  const int c0 = ko + kb + (bx * 4 + wave) * 16;
  if (c0 >= n) return;
  const int col = min(c0 + cc, n - 1);
  const bool cvalid = c0 + cc < n;
  const double* Ab = A + (size_t)ko * ld + ko;  // diagonal block (finished R)
  double* P = A + (size_t)ko * ld;              // row panel
  double4_t X[4][4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (j < nblk) {
      const int nbj = min(64, kb - 64 * j);
      double4_t T[4];
#pragma unroll
      for (int g = 0; g < 4; g++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = 16 * g + kk + 4 * r;
          const double v = P[(size_t)(64 * j + min(row, nbj - 1)) * ld + col];
          T[g][r] = (row < nbj) ? v : 0.0;
        }
      // T -= R_ij^T X_i for the finished row blocks (A fragments of R_ij straight from L2)
#pragma unroll
      for (int i = 0; i < j; i++) {
        const double* Rij = Ab + (size_t)(64 * i) * ld + 64 * j;
#pragma unroll
        for (int s0 = 0; s0 < 16; s0 += 4) {
          double af[4][4];
#pragma unroll
          for (int s = 0; s < 4; s++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
              const int q = 16 * g + cc;  // column inside tile (i, j): valid iff < nbj (rows of block i are always full)
              const double v = Rij[(size_t)(4 * (s0 + s) + kk) * ld + min(q, nbj - 1)];
              af[s][g] = (q < nbj) ? -v : 0.0;
            }
#pragma unroll
          for (int s = 0; s < 4; s++)
#pragma unroll
            for (int g = 0; g < 4; g++) T[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s][g], X[i][(s0 + s) >> 2][(s0 + s) & 3], T[g], 0, 0, 0);
        }
      }
      // solve with R_jj: sub-tiles from L2, inverses from LDS
      const double* Rjj = Ab + (size_t)(64 * j) * ld + 64 * j;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        double4_t acc = T[g];
#pragma unroll
        for (int i = 0; i < g; i++)
#pragma unroll
          for (int sx = 0; sx < 4; sx++) {
            const int p = 16 * i + 4 * sx + kk, q = 16 * g + cc;
            const double v = Rjj[(size_t)min(p, nbj - 1) * ld + min(q, nbj - 1)];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64((p < nbj && q < nbj) ? -v : 0.0, T[i][sx], acc, 0, 0, 0);
          }
        double4_t out = double4_t{0, 0, 0, 0};
#pragma unroll
        for (int sx = 0; sx < 4; sx++) out = __builtin_amdgcn_mfma_f64_16x16x4f64(I16[4 * j + g][4 * sx + kk][cc], acc[sx], out, 0, 0, 0);
        T[g] = out;
        X[j][g] = out;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = 16 * g + kk + 4 * r;
          if (row < nbj && cvalid) P[(size_t)(64 * j + row) * ld + c0 + cc] = out[r];
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void panel_trsm_kernel(double* __restrict__ A, int ld, int n, int ko, int kb, const double* __restrict__ inv16) {
  __shared__ double I16[16][16][17];
  panel_trsm_body(A, ld, n, ko, kb, inv16, I16, blockIdx.x);
}

// ---------------------------------------------------------------- the end of a front in one small launch
// When what is left after an outer panel is small (m = n - r0 <= TAIL_MAX_M columns: the last, partial panel of the frontal part
// plus the separator and the right-hand side -- BAL's 9001-column root leaves 40 + 1), the remaining three steps
//   update with panel [p0, p0 + kp)  ->  factor the last nf - r0 frontal columns  ->  update what is right of them
// are one workgroup: the panel's columns and the trailing block staged in LDS, C -= P^T P, a right-looking partial Cholesky
// in LDS, everything written back (the rows below the frontal part are the front's update matrix for its parent).  Replaces
// two update launches and the two-launch panel (five launches with their gaps: ~0.15 ms of a 9 ms step) by ~35 us.
// Also leaves the 16x16 inverses of its diagonal tiles (inv16, identity-padded) like diag_potrf_kernel.
#define TAIL_MAX_M 48
#define TAIL_MAX_KP 256
#define TAIL_LDS_BYTES ((TAIL_MAX_KP * TAIL_MAX_M + TAIL_MAX_M * TAIL_MAX_M + 64 * 65) * 8)
__global__ __launch_bounds__(256) void front_tail_kernel(double* __restrict__ A, int ld, int n, int nf, int p0, int kp, int front_id,
                                                         int* __restrict__ status, double* __restrict__ inv16) {
  extern __shared__ double tsm[];
  const int r0 = p0 + kp, m = n - r0, nft = nf - r0, tid = threadIdx.x;
  double* Pl = tsm;                               // [kp][m]
  double* S = tsm + (size_t)TAIL_MAX_KP * TAIL_MAX_M;  // [m][m] row-major, upper
  double(*D)[65] = (double(*)[65])(S + TAIL_MAX_M * TAIL_MAX_M);  // 64x64 identity-padded copy of the frontal factor for the inverses
  for (int idx = tid; idx < kp * m; idx += 256) {
    const int k = idx / m, j = idx - k * m;
    Pl[idx] = A[(size_t)(p0 + k) * ld + r0 + j];
  }
  for (int idx = tid; idx < m * m; idx += 256) {
    const int i = idx / m, j = idx - i * m;
    S[idx] = (j >= i) ? A[(size_t)(r0 + i) * ld + r0 + j] : 0.0;
  }
  __syncthreads();
  for (int idx = tid; idx < m * m; idx += 256) {
    const int i = idx / m, j = idx - i * m;
    if (j < i) continue;
    double s0 = 0.0, s1 = 0.0;
    int k = 0;
    for (; k + 1 < kp; k += 2) {
      s0 += Pl[k * m + i] * Pl[k * m + j];
      s1 += Pl[(k + 1) * m + i] * Pl[(k + 1) * m + j];
    }
    if (k < kp) s0 += Pl[k * m + i] * Pl[k * m + j];
    S[idx] -= s0 + s1;
  }
  bool failed = false;
  for (int k = 0; k < nft; k++) {
    __syncthreads();
    double piv = S[k * m + k];
    if (!(piv > 0.0)) {
      if (piv <= 0.0) failed = true;  // Eigen LLT: pivot <= 0 -> NumericalIssue (NaN passes, like Eigen)
      piv = (piv == piv && piv != 0.0) ? fabs(piv) : 1.0;
    }
    const double r = sqrt(piv), inv = 1.0 / r;
    for (int j = k + 1 + tid; j < m; j += 256) S[k * m + j] *= inv;
    __syncthreads();
    if (tid == 0) S[k * m + k] = r;
    for (int idx = tid; idx < (m - k - 1) * (m - k - 1); idx += 256) {
      const int i = k + 1 + idx / (m - k - 1), j = k + 1 + idx % (m - k - 1);
      if (j >= i) S[i * m + j] -= S[k * m + i] * S[k * m + j];
    }
  }
  __syncthreads();
  if (tid == 0) {  // pivot-exponent test, gtsam/base/cholesky.cpp:146-158
    if (nft >= 2) {
      if (!(frexp_exp_d(S[(nft - 2) * m + nft - 2]) - frexp_exp_d(S[(nft - 1) * m + nft - 1]) < 12)) failed = true;
    } else if (nft == 1) {
      const double r1 = S[0];
      if (nf >= 2) {
        if (!(frexp_exp_d(A[(size_t)(nf - 2) * ld + nf - 2]) - frexp_exp_d(r1) < 12)) failed = true;
      } else if (!(frexp_exp_d(r1) > -12)) {
        failed = true;
      }
    }
    if (failed) atomicMin(status, front_id);
  }
  for (int idx = tid; idx < m * m; idx += 256) {
    const int i = idx / m, j = idx - i * m;
    if (j >= i) A[(size_t)(r0 + i) * ld + r0 + j] = S[idx];
  }
  // 16x16 inverses of the diagonal tiles of the frontal factor (identity-padded to 64)
  for (int idx = tid; idx < 64 * 64; idx += 256) {
    const int p = idx >> 6, q = idx & 63;
    D[p][q] = (p < nft && q < nft && q >= p) ? S[p * m + q] : ((p == q) ? 1.0 : 0.0);
  }
  __syncthreads();
  if (tid < 64) {
    const int blk = tid >> 4, c = tid & 15, base = 16 * blk;
    double x[16];
#pragma unroll
    for (int i = 15; i >= 0; i--) {
      double sacc = (i == c) ? 1.0 : 0.0;
#pragma unroll
      for (int kq = i + 1; kq < 16; kq++) sacc -= D[base + i][base + kq] * x[kq];
      x[i] = (i <= c) ? sacc / D[base + i][base + i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) inv16[(size_t)blk * 256 + i * 16 + c] = x[i];
  }
}

// ---------------------------------------------------------------- the same outer panel as ONE dataflow launch
// Workgroup b < nblk owns block column b of the diagonal block (and factors diagonal tile b); the others own 64 columns
// right of it.  Every workgroup runs the left-looking column algorithm of panel_trsm_kernel on its own columns and
// consumes  R_jj + its 16x16 inverses   (flag diag_ready[j], published by workgroup j after potrf64_wave_g4)  and the tiles
// R_ij of block column j (flag tile_ready[i][j], published by workgroup j) from global memory.  A diagonal workgroup keeps
// its diagonal tile in registers and folds X_j into it as soon as X_j exists, so the dependency chain of the launch is
//   potrf(0) -> hand-off -> [workgroup 1: solve X_0, one tile update, potrf(1)] -> hand-off -> ...
// i.e. four register Choleskys, three tile solves/updates and four hand-offs per 256 rows, with all other work beside it.
// Hand-offs follow cdna_hip_programming.md Guideline 16: plain stores, every wave drains (s_waitcnt vmcnt(0)), workgroup
// barrier, ONE lane: agent-scope release fence, wait, relaxed agent-scope flag store / add; consumer: ONE lane polls
// relaxed, ONE agent-scope acquire fence, wait, workgroup barrier, then plain vector loads.  Logical workgroup ids come
// from a ticket counter, so a workgroup only ever waits for workgroups that started before it (no dispatch-order assumption).
// flags (zeroed by the host before the launch): [0] ticket, [4 + j] diag_ready[j], [8 + 4 i + j] tile_ready[i][j],
// [PDF_TA0 + sj] number of finished trailing-update workgroups of column strip sj in the next panel's rows (fused step only).
// development aid (tools/microbench.hip defines it): wave 0 lane 0 of the diagonal workgroups stores s_memtime at the phase boundaries
#ifndef PDF_STAMP
#define PDF_STAMP(flags, b, slot)
#endif
#define PDF_TA0 32
#define PDF_MAX_COLTILES 992  // column strips: fronts up to 63 488 columns take the fused path
// chained steps only (chain_kernel, kernels_step.hpp):
//   [PDF_PR0 + q]  finished row-panel workgroups of the 128-column block q right of the panel (its rows there are final),
//   [PDF_TD0 + x]  trailing-update tile x = (ti, tj) of the step that factors this panel has been stored
#define PDF_PR0 1024
#define PDF_TD0 1536
#define PDF_FLAG_WORDS 8192
#define PDF_MAX_CHAIN_T 114  // (T + 1) T / 2 <= PDF_FLAG_WORDS - PDF_TD0
#define PDF_SPIN_LIMIT 2000000L  // a legitimate wait is < 1 ms; the bound (~1-2 s) only keeps a logic error from hanging the device
#define PDF_LDS_DOUBLES (2 * 64 * DP_LDW)
#define PDF_LDS_BYTES (PDF_LDS_DOUBLES * 8)

// publications a finished column strip sj of the next panel's rows has received (kernels_step.hpp): strips 0..3 are cut into
// 32x32 quadrant workgroups (sj + 1 sub-tiles x 4), the others belong to the two 128x128 tiles (tile rows 0 and 1) of their tile column
__host__ __device__ inline unsigned int pdf_ta_need(int sj) { return sj < 4 ? 4u * (unsigned int)(sj + 1) : 2u; }

// all threads call; thread 0 waits until flags[w] >= need for the (up to two) words given, then one acquire covers the workgroup
__device__ __forceinline__ bool pdf_wait(unsigned int* flags, int w0, unsigned int need0, int w1, unsigned int need1, int* s_ok, int tid) {
  if (tid == 0) {
    bool ok = true;
    for (int q = 0; q < 2 && ok; q++) {
      const int w = q ? w1 : w0;
      const unsigned int need = q ? need1 : need0;
      if (w < 0) continue;
      long spins = 0;
      while (__hip_atomic_load(&flags[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > PDF_SPIN_LIMIT) {
          ok = false;
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *s_ok = ok ? 1 : 0;
  }
  __syncthreads();
  return *s_ok != 0;
}

// the same for up to three flag words given by address (nullptr: none)
__device__ __forceinline__ bool pdf_wait3(const unsigned int* f0, unsigned int n0, const unsigned int* f1, unsigned int n1, const unsigned int* f2,
                                          unsigned int n2, int* s_ok, int tid) {
  if (tid == 0) {
    bool ok = true;
    for (int q = 0; q < 3 && ok; q++) {
      const unsigned int* f = q == 0 ? f0 : (q == 1 ? f1 : f2);
      const unsigned int need = q == 0 ? n0 : (q == 1 ? n1 : n2);
      if (!f) continue;
      long spins = 0;
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > PDF_SPIN_LIMIT) {
          ok = false;
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *s_ok = ok ? 1 : 0;
  }
  __syncthreads();
  return *s_ok != 0;
}

// every wave calls this after its last store of the payload; `signaller` = the one thread that raises the flag
__device__ __forceinline__ void pdf_publish(unsigned int* flag, bool signaller) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (signaller) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// the same for a workgroup that completes two flags at once (flag1 may be nullptr)
__device__ __forceinline__ void pdf_publish2(unsigned int* flag0, unsigned int* flag1, bool signaller) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (signaller) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(flag0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (flag1) __hip_atomic_fetch_add(flag1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The work of logical workgroup b on the outer panel [ko, ko + kb) (kb a multiple of 64).  ta_need > 0: the panel's rows
// are being produced by trailing-update tiles of the same launch; wait for the ones covering this workgroup's columns.
// publish_strips: a row-panel workgroup counts itself into [PDF_PR0 + 128-column block] when its columns are final (chained steps).
__device__ __forceinline__ void panel_role(double* A, int ld, int n, int nf, int ko, int kb, int b, int front_id, int* status, double* inv16,
                                           unsigned int* flags, double* dsm, int* s_ok, bool fused, bool publish_strips = false) {
  double(*D)[DP_LDW] = (double(*)[DP_LDW])dsm;
  double(*XB)[DP_LDW] = (double(*)[DP_LDW])(dsm + 64 * DP_LDW);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kk = lane >> 4, cc = lane & 15;
  const int nblk = kb >> 6;
  const bool diagwg = b < nblk;
  const int c0 = diagwg ? ko + 64 * b + 16 * wave : ko + kb + 64 * (b - nblk) + 16 * wave;
  const int col = min(c0 + cc, n - 1);
  const bool cvalid = c0 + cc < n;
  const int wc = 16 * wave + cc;
  double* Ab = A + (size_t)ko * ld + ko;  // diagonal block
  double* P = A + (size_t)ko * ld;        // row panel
  bool healthy = true;
  if (diagwg) __builtin_amdgcn_s_setprio(3);  // the chain of the launch: ahead of the update waves sharing its SIMDs
  PDF_STAMP(flags, b, 19);
  if (fused) {
    const int sj = diagwg ? b : nblk + (b - nblk);  // 64-column strip of the trailing update (its origin is ko)
    healthy &= pdf_wait(flags, PDF_TA0 + sj, pdf_ta_need(sj), -1, 0u, s_ok, tid);
  }
  PDF_STAMP(flags, b, 0);
  double4_t X[4][4];
  // diagonal workgroup: the ten upper 16x16 tiles of the diagonal tile (b, b), right-looking, dealt three / two per wave:
  //   wave 0: (0,0) (0,3)   wave 1: (0,1) (1,1) (1,3)   wave 2: (0,2) (1,2) (2,2)   wave 3: (2,3) (3,3)
  double4_t Td[3];
  const int td_n = (wave == 0 || wave == 3) ? 2 : 3;
  const int td_g0 = (wave == 3) ? 2 : 0, td_g1 = (wave == 0) ? 0 : (wave == 3 ? 3 : 1), td_g2 = (wave == 1) ? 1 : 2;
  const int td_h0 = wave, td_h1 = (wave == 0) ? 3 : wave, td_h2 = (wave == 1) ? 3 : 2;
  if (diagwg) {
    const double* Pd = P + (size_t)(64 * b) * ld + ko + 64 * b;
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const int tg = t == 0 ? td_g0 : (t == 1 ? td_g1 : td_g2), th = t == 0 ? td_h0 : (t == 1 ? td_h1 : td_h2);
#pragma unroll
      for (int r = 0; r < 4; r++) Td[t][r] = (t < td_n) ? Pd[(size_t)(16 * tg + kk + 4 * r) * ld + 16 * th + cc] : 0.0;
    }
  }
  const int jend = diagwg ? b : nblk;  // exclusive
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (j < jend) {
      double4_t T[4];
#pragma unroll
      for (int g = 0; g < 4; g++)
#pragma unroll
        for (int r = 0; r < 4; r++) T[g][r] = P[(size_t)(64 * j + 16 * g + kk + 4 * r) * ld + col];
      if (j > 0) {
        // tiles R_ij (i < j) of block column j, published by workgroup j in the order i = 0, 1, ..: the last one implies the others
        healthy &= pdf_wait(flags, 8 + 4 * (j - 1) + j, 1u, -1, 0u, s_ok, tid);
#pragma unroll
        for (int i = 0; i < j; i++) {
          const double* Rij = Ab + (size_t)(64 * i) * ld + 64 * j;
#pragma unroll
          for (int s0 = 0; s0 < 16; s0 += 4) {
            double af[4][4];
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
              for (int g = 0; g < 4; g++) af[s][g] = -Rij[(size_t)(4 * (s0 + s) + kk) * ld + 16 * g + cc];
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
              for (int g = 0; g < 4; g++) T[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s][g], X[i][(s0 + s) >> 2][(s0 + s) & 3], T[g], 0, 0, 0);
          }
        }
      }
      healthy &= pdf_wait(flags, 4 + j, 1u, -1, 0u, s_ok, tid);
      PDF_STAMP(flags, b, 1 + 4 * j);
      {
        // -R_jj is staged in LDS once per workgroup (D is free until the final gather): one 512-byte row per wave instruction
        const double* Rjj = Ab + (size_t)(64 * j) * ld + 64 * j;
        const double* Ij = inv16 + (size_t)(4 * j) * 256;
        double isub[4][4];
#pragma unroll
        for (int u = 0; u < 16; u++) D[wave + 4 * u][lane] = -Rjj[(size_t)(wave + 4 * u) * ld + lane];
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
          for (int sx = 0; sx < 4; sx++) isub[g][sx] = Ij[g * 256 + (4 * sx + kk) * 16 + cc];
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; g++) {
          double4_t acc = T[g];
#pragma unroll
          for (int i = 0; i < g; i++)
#pragma unroll
            for (int sx = 0; sx < 4; sx++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(D[16 * i + 4 * sx + kk][16 * g + cc], T[i][sx], acc, 0, 0, 0);
          double4_t out = double4_t{0, 0, 0, 0};
#pragma unroll
          for (int sx = 0; sx < 4; sx++) out = __builtin_amdgcn_mfma_f64_16x16x4f64(isub[g][sx], acc[sx], out, 0, 0, 0);
          T[g] = out;
          X[j][g] = out;
#pragma unroll
          for (int r = 0; r < 4; r++)
            if (cvalid) P[(size_t)(64 * j + 16 * g + kk + 4 * r) * ld + c0 + cc] = out[r];
        }
      }
      PDF_STAMP(flags, b, 2 + 4 * j);
      if (diagwg) {
        // own tile (j, b): a copy in LDS for the A operands, publish it, fold it into the diagonal tile
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
          for (int r = 0; r < 4; r++) XB[16 * g + kk + 4 * r][wc] = X[j][g][r];
        pdf_publish(&flags[8 + 4 * j + b], tid == 64);
        PDF_STAMP(flags, b, 3 + 4 * j);
#pragma unroll
        for (int s = 0; s < 16; s++) {
          Td[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-XB[4 * s + kk][16 * td_g0 + cc], X[j][s >> 2][s & 3], Td[0], 0, 0, 0);  // h == wave
          Td[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-XB[4 * s + kk][16 * td_g1 + cc], XB[4 * s + kk][16 * td_h1 + cc], Td[1], 0, 0, 0);
          if (td_n == 3) Td[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-XB[4 * s + kk][16 * td_g2 + cc], XB[4 * s + kk][16 * td_h2 + cc], Td[2], 0, 0, 0);
        }
        PDF_STAMP(flags, b, 4 + 4 * j);
      }
    }
  }
  if (!healthy && tid == 0) atomicExch(status + 1, 1 + front_id);  // never expected: spin bound hit (a fault, reported apart from pivot failures)
  if (!diagwg) {
    if (publish_strips) pdf_publish(&flags[PDF_PR0 + ((b - nblk) >> 1)], tid == 0);
    return;
  }
  // ---- diagonal tile b: gather the ten tiles, factor in wave 0, publish R_bb and its 16x16 inverses straight from registers
#pragma unroll
  for (int t = 0; t < 3; t++) {
    const int tg = t == 0 ? td_g0 : (t == 1 ? td_g1 : td_g2), th = t == 0 ? td_h0 : (t == 1 ? td_h1 : td_h2);
    if (t < td_n) {
#pragma unroll
      for (int r = 0; r < 4; r++) D[16 * tg + kk + 4 * r][16 * th + cc] = Td[t][r];
    }
  }
  __syncthreads();
  PDF_STAMP(flags, b, 20);
  if (wave == 0) {
    double4_t T[4][4], E[4];
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
      for (int h = g; h < 4; h++)
#pragma unroll
        for (int r = 0; r < 4; r++) T[g][h][r] = D[16 * g + kk + 4 * r][16 * h + cc];
    bool failed = potrf64_wave_g4<true>(T, E);
    PDF_STAMP(flags, b, 21);
    double* Aj = Ab + (size_t)(64 * b) * ld + 64 * b;
#pragma unroll
    for (int g = 0; g < 4; g++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (cc >= kk + 4 * r) Aj[(size_t)(16 * g + kk + 4 * r) * ld + 16 * g + cc] = T[g][g][r];
#pragma unroll
        for (int h = g + 1; h < 4; h++) Aj[(size_t)(16 * g + kk + 4 * r) * ld + 16 * h + cc] = T[g][h][r];
        inv16[(size_t)(4 * b + g) * 256 + cc * 16 + kk + 4 * r] = E[g][r];  // E = R_gg^-T
      }
    }
    if (ko + 64 * b + 64 >= nf) {  // last frontal rows: pivot-exponent test, gtsam/base/cholesky.cpp:146-158
      const double r1 = readlane_d(T[3][3][3], 63), r2 = readlane_d(T[3][3][3], 46);  // (63, 63) and (62, 62)
      if (!(frexp_exp_d(r2) - frexp_exp_d(r1) < 12)) failed = true;
    }
    if (failed && lane == 0) atomicMin(status, front_id);
    PDF_STAMP(flags, b, 22);
  }
  pdf_publish(&flags[4 + b], tid == 64);
  PDF_STAMP(flags, b, 23);
}

__global__ __launch_bounds__(256) void panel_dataflow_kernel(double* A, int ld, int n, int nf, int ko, int kb, int front_id, int* status,
                                                              double* inv16, unsigned int* flags) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  __shared__ int s_bid, s_ok;
  if (threadIdx.x == 0) s_bid = (int)atomicAdd(&flags[0], 1u);
  __syncthreads();
  panel_role(A, ld, n, nf, ko, kb, s_bid, front_id, status, inv16, flags, dsm, &s_ok, false);
}

}  // namespace lmgpu
