// Atomic-free, bitwise-reproducible assembly of an HBM front from its LEAF children (the Schur-complement form).
// For a leaf front c (no children) with eliminated block [R S d], its contribution to the parent is
//     U_c = sum_{own factors} [A b]_sep^T [A b]_sep  -  [S d]^T [S d]
// where a factor only touches ONE separator variable (its other variable is the leaf's frontal one), so the first term
// is block-diagonal.  Instead of every leaf scattering its dense (ns+1)^2 update (BAL: 100 000 x 4186 FP64 atomics =
// 3.35 GB of atomic traffic), the parent GATHERS:
//   schur_pairs_kernel    per destination block (variable pair a <= b of the parent, or (a, rhs)):
//                         A[a][b] -= sum over the leaves that see both of S_a^T S_b   (reads the leaves' [R S d] rows,
//                         which are in HBM anyway for the back-substitution; the 218 MB of S at C4 sit in the Infinity Cache)
//   schur_factor_kernel   per separator variable: A[v][v] += sum A_v^T A_v,  A[v][rhs] += sum A_v^T b
//   the (rhs, rhs) corner is a fixed-order reduction of one scalar per leaf.
// Every destination entry is owned by exactly one lane, source lists have a fixed order and long lists are cut into
// fixed contiguous chunks (one per wave) whose partial sums are added in chunk order => reproducible sums.
// Entries are self-contained 16-byte records, fetched 64 at a time by the lanes and broadcast with shuffles, so the only
// dependent HBM/L2 access per entry is the data itself.
// This is the same arithmetic as HessianFactor::updateHessian of the child separator factors
// (gtsam/linear/HessianFactor.cpp:349-373), grouped by destination.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_front.hpp"

namespace lmgpu {

struct GPairBlock {
  int32_t pa, pb;   // parent columns of the destination block (pa <= pb)
  int16_t da, db;   // block dims
  int32_t begin, count;
};
struct GPairEntry {  // 16 bytes
  int64_t rsd_off;   // pool offset of the leaf's TRANSPOSED [S d]  ((n - nf) x nf, written by lds_front_kernel<true>)
  int16_t sa, sb;    // element offsets of the (contiguous, [column][row]) blocks S_a / S_b inside it
  int16_t nf, ld;    // rows of [R S d]; ld unused
};
struct GVarBlock {
  int32_t pv;       // parent column of the variable
  int16_t dv, pad;
  int32_t begin, count;
};
struct GVarEntry {   // 16 bytes
  int64_t joff;      // pool offset of the factor's [A b]
  int16_t rows, c0, cb, pad;  // rows, first column of the separator variable, column of b
};

__device__ __forceinline__ long long shfl_ll(long long v, int src) {
  const int lo = __shfl((int)(v & 0xffffffffLL), src), hi = __shfl((int)(v >> 32), src);
  return ((long long)hi << 32) | (unsigned int)lo;
}

// WAVES waves per destination block (1 for short lists, 4 for long ones); grid = number of blocks
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void schur_pairs_kernel(const GPairBlock* __restrict__ blocks, const GPairEntry* __restrict__ entries,
                                                                  double* __restrict__ pool, int64_t f_off, int ld) {
  __shared__ double part[WAVES][128];
  const GPairBlock B = blocks[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nout = B.da * B.db;
  const int o0 = lane, o1 = lane + 64;  // da * db <= 81
  const int i0 = o0 / B.db, j0 = o0 - i0 * B.db, i1 = o1 / B.db, j1 = o1 - i1 * B.db;
  const bool v0 = o0 < nout, v1 = o1 < nout;
  const int per = (B.count + WAVES - 1) / WAVES;
  const int cb = B.begin + wave * per, ce = min(B.begin + B.count, cb + per);
  double s0 = 0, s1 = 0;
  for (int base = cb; base < ce; base += 64) {
    const int cnt = min(64, ce - base);
    long long moff = 0, mmeta = 0;
    if (lane < cnt) {
      const GPairEntry E = entries[base + lane];
      moff = E.rsd_off;
      mmeta = ((long long)(unsigned short)E.sa) | ((long long)(unsigned short)E.sb << 16) | ((long long)(unsigned short)E.nf << 32) |
              ((long long)(unsigned short)E.ld << 48);
    }
    for (int e = 0; e < cnt; e += 4) {  // 4 entries per trip: their loads are independent and in flight together
      double t0[4], t1[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int eu = min(e + u, cnt - 1);
        const long long off = shfl_ll(moff, eu), meta = shfl_ll(mmeta, eu);
        const int sa = (int)(meta & 0xffff), sb = (int)((meta >> 16) & 0xffff), nf = (int)((meta >> 32) & 0xffff), lde = (int)((meta >> 48) & 0xffff);
        const double* St = pool + off;
        (void)lde;
        double a0 = 0, a1 = 0;
        for (int r = 0; r < nf; r++) {
          if (v0) a0 += St[sa + i0 * nf + r] * St[sb + j0 * nf + r];
          if (v1) a1 += St[sa + i1 * nf + r] * St[sb + j1 * nf + r];
        }
        const bool valid = e + u < cnt;
        t0[u] = valid ? a0 : 0.0;
        t1[u] = valid ? a1 : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {  // fixed order
        s0 += t0[u];
        s1 += t1[u];
      }
    }
  }
  if (WAVES > 1) {
    part[wave][lane] = s0;
    part[wave][lane + 64] = s1;
    __syncthreads();
    if (wave != 0) return;
    s0 = 0;
    s1 = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) {
      s0 += part[w][lane];
      s1 += part[w][lane + 64];
    }
  }
  double* A = pool + f_off;
  const bool diag = (B.pa == B.pb);
  if (v0 && (!diag || i0 <= j0)) A[(size_t)(B.pa + i0) * ld + B.pb + j0] -= s0;
  if (v1 && (!diag || i1 <= j1)) A[(size_t)(B.pa + i1) * ld + B.pb + j1] -= s1;
}

// 16 waves per separator variable (BAL: ~1000 factors per camera)
#define SCHUR_FW 16
__global__ __launch_bounds__(64 * SCHUR_FW) void schur_factor_kernel(const GVarBlock* __restrict__ blocks, const GVarEntry* __restrict__ entries,
                                                                      double* __restrict__ pool, int64_t f_off, int ld, int n) {
  __shared__ double part[SCHUR_FW][128];
  const GVarBlock B = blocks[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, d = B.dv, nout = d * d + d;  // d x d block, then the rhs column
  const int o0 = lane, o1 = lane + 64;
  // output o < d*d: (i, j) of A_v^T A_v ; else row (o - d*d) of A_v^T b   -> columns (ci, cj) relative to the variable / b
  auto cols = [&](int o, int& ci, int& cj, bool& isb) {
    if (o < d * d) {
      ci = o / d;
      cj = o - ci * d;
      isb = false;
    } else {
      ci = o - d * d;
      cj = 0;
      isb = true;
    }
  };
  int a0, b0, a1, b1;
  bool r0, r1;
  cols(o0, a0, b0, r0);
  cols(o1, a1, b1, r1);
  const bool v0 = o0 < nout, v1 = o1 < nout;
  const int per = (B.count + SCHUR_FW - 1) / SCHUR_FW;
  const int cbeg = B.begin + wave * per, cend = min(B.begin + B.count, cbeg + per);
  double s0 = 0, s1 = 0;
  for (int base = cbeg; base < cend; base += 64) {
    const int cnt = min(64, cend - base);
    long long moff = 0, mmeta = 0;
    if (lane < cnt) {
      const GVarEntry E = entries[base + lane];
      moff = E.joff;
      mmeta = ((long long)(unsigned short)E.rows) | ((long long)(unsigned short)E.c0 << 16) | ((long long)(unsigned short)E.cb << 32);
    }
    for (int e = 0; e < cnt; e += 4) {
      double t0[4], t1[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int eu = min(e + u, cnt - 1);
        const long long off = shfl_ll(moff, eu), meta = shfl_ll(mmeta, eu);
        const int m = (int)(meta & 0xffff), c0 = (int)((meta >> 16) & 0xffff), cb = (int)((meta >> 32) & 0xffff);
        const double* J = pool + off;
        double x0 = 0, x1 = 0;
        if (v0) {
          const double* x = J + (c0 + a0) * m;
          const double* y = J + (r0 ? cb : c0 + b0) * m;
          for (int r = 0; r < m; r++) x0 += x[r] * y[r];
        }
        if (v1) {
          const double* x = J + (c0 + a1) * m;
          const double* y = J + (r1 ? cb : c0 + b1) * m;
          for (int r = 0; r < m; r++) x1 += x[r] * y[r];
        }
        const bool valid = e + u < cnt;
        t0[u] = valid ? x0 : 0.0;
        t1[u] = valid ? x1 : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        s0 += t0[u];
        s1 += t1[u];
      }
    }
  }
  part[wave][lane] = s0;
  part[wave][lane + 64] = s1;
  __syncthreads();
  if (wave != 0) return;
  s0 = 0;
  s1 = 0;
#pragma unroll
  for (int w = 0; w < SCHUR_FW; w++) {
    s0 += part[w][lane];
    s1 += part[w][lane + 64];
  }
  double* A = pool + f_off;
  if (v0) {
    if (!r0) {
      if (a0 <= b0) A[(size_t)(B.pv + a0) * ld + B.pv + b0] += s0;
    } else {
      A[(size_t)(B.pv + a0) * ld + n - 1] += s0;
    }
  }
  if (v1) {
    if (!r1) {
      if (a1 <= b1) A[(size_t)(B.pv + a1) * ld + B.pv + b1] += s1;
    } else {
      A[(size_t)(B.pv + a1) * ld + n - 1] += s1;
    }
  }
}

__global__ void add_scalar_kernel(double* __restrict__ dst, const double* __restrict__ src) { *dst += *src; }

}  // namespace lmgpu
