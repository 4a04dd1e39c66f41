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
// Entries are self-contained 16-byte records, fetched 64 at a time by the lanes and broadcast with v_readlane, so the only
// dependent HBM/L2 access per entry is the data itself; the products run on the matrix cores (one MFMA per entry).
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

typedef double double4s_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ long long readlane_ll(long long v, int src /* wave-uniform */) {
  const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffLL), src), hi = __builtin_amdgcn_readlane((int)(v >> 32), src);
  return ((long long)hi << 32) | (unsigned int)lo;
}

// an 8-byte global load that only the lanes of `mask` (wave-uniform) take part in: no branch (the compiler waits for a load issued under a
// branch at the join), and the address unit is given 27 lanes instead of 64.  The compiler does not count this load: the caller waits with
// an explicit s_waitcnt before it uses the values (the compiler's own counts only ever come out too high with these loads outstanding).
__device__ __forceinline__ double schur_load_masked(const double* p, unsigned long long mask) {
  double v;
  unsigned long long save;
  asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, %2\n\tglobal_load_dwordx2 %0, %3, off\n\ts_mov_b64 exec, %1"
               : "=&v"(v), "=&s"(save)
               : "s"(mask), "v"(p)
               : "memory");
  return v;
}

// WAVES waves per destination block (1 for short lists, 4 for long ones); grid = number of blocks.
// One v_mfma_f64_16x16x4_f64 per list entry:  D[i][j] += sum_k S_a[k][i] S_b[k][j]  with lane (i = lane & 15, k = lane >> 4)
// supplying S_a[k][i] as the A operand and S_b[k][i] as the B operand (two 8-byte loads per lane and entry; rows k >= nf and
// columns >= the block dims are zero).  Four accumulators take entries e, e+1, e+2, e+3 of every group of four and are
// added in a fixed order; waves take fixed contiguous chunks of the list.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void schur_pairs_kernel(const GPairBlock* __restrict__ blocks, const GPairEntry* __restrict__ entries,
                                                                  double* __restrict__ pool, int64_t f_off, int ld, int nblocks, int write_mode) {
  __shared__ double part[WAVES][4][64];
  // XCD-aware order: workgroup ids go round-robin over the eight XCDs, so id -> (id % 8) * ceil(N / 8) + id / 8 gives every XCD
  // a contiguous range of the (row-major sorted) destination blocks: the blocks of one camera row re-read that camera's S
  // blocks nine times between them, which only hits in L2 if they run on the same XCD
  // (the grid is nblocks rounded up to a multiple of eight, so that the map is onto)
  const int per_xcd = (nblocks + 7) >> 3;
  const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (logical >= nblocks) return;
  const GPairBlock B = blocks[logical];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kk = lane >> 4, cc = lane & 15;
  const bool va = cc < B.da, vb = cc < B.db;
  const int per = (B.count + WAVES - 1) / WAVES;
  const int cb = B.begin + wave * per, ce = min(B.begin + B.count, cb + per);
  double4s_t acc[4];
#pragma unroll
  for (int u = 0; u < 4; u++) acc[u] = double4s_t{0, 0, 0, 0};
  for (int base = cb; base < ce; base += 64) {
    const int cnt = min(64, ce - base);
    long long moff = 0, mmeta = 0;
    if (lane < cnt) {
      const GPairEntry E = entries[base + lane];
      moff = E.rsd_off;
      mmeta = ((long long)(unsigned short)E.sa) | ((long long)(unsigned short)E.sb << 16) | ((long long)(unsigned short)E.nf << 32);
    }
    for (int e = 0; e < cnt; e += 4) {
      if (write_mode & 2) {
        // the operands of four entries as masked loads (BAL: 27 of 64 lanes hold an element of a 3 x 9 block), one wait, four products
        double av[4], bv[4];
        bool kav[4], kbv[4];
        bool small = true;
        long long offs[4], metas[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int eu = min(e + u, cnt - 1);
          offs[u] = readlane_ll(moff, eu);
          metas[u] = readlane_ll(mmeta, eu);
          small = small && (int)((metas[u] >> 32) & 0xffff) <= 4;
        }
        if (small) {  // (wave-uniform)
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int sa = (int)(metas[u] & 0xffff), sb = (int)((metas[u] >> 16) & 0xffff), nf = (int)((metas[u] >> 32) & 0xffff);
            const double* St = pool + offs[u];
            const bool valid = e + u < cnt;
            kav[u] = valid && va && kk < nf;
            kbv[u] = valid && vb && kk < nf;
            av[u] = schur_load_masked(St + sa + cc * nf + kk, __builtin_amdgcn_ballot_w64(kav[u]));
            bv[u] = schur_load_masked(St + sb + cc * nf + kk, __builtin_amdgcn_ballot_w64(kbv[u]));
          }
          asm volatile("s_waitcnt vmcnt(0)"
                       : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3])::"memory");
#pragma unroll
          for (int u = 0; u < 4; u++)
            acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(kav[u] ? av[u] : 0.0, kbv[u] ? bv[u] : 0.0, acc[u], 0, 0, 0);
          continue;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int eu = min(e + u, cnt - 1);
        const long long off = readlane_ll(moff, eu), meta = readlane_ll(mmeta, eu);
        const int sa = (int)(meta & 0xffff), sb = (int)((meta >> 16) & 0xffff), nf = (int)((meta >> 32) & 0xffff);
        const double* St = pool + off;
        const bool valid = e + u < cnt;
        for (int k0 = 0; k0 < nf; k0 += 4) {
          const int k = k0 + kk;
          const bool ka = valid && va && k < nf, kb = valid && vb && k < nf;
          const double av = St[ka ? sa + cc * nf + k : sa], bv = St[kb ? sb + cc * nf + k : sb];
          acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(ka ? av : 0.0, kb ? bv : 0.0, acc[u], 0, 0, 0);
        }
      }
    }
  }
  double4s_t sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  if (WAVES > 1) {
#pragma unroll
    for (int r = 0; r < 4; r++) part[wave][r][lane] = sum[r];
    __syncthreads();
    if (wave != 0) return;
    sum = double4s_t{0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < WAVES; w++)
#pragma unroll
      for (int r = 0; r < 4; r++) sum[r] += part[w][r][lane];
  }
  double* A = pool + f_off;
  const bool diag = (B.pa == B.pb);
  // destination read-modify-write: the four loads of a lane in flight together (clamped address where the slot is unused).
  // (Issuing them before the list walk was measured slower: 1.41 vs 1.26 ms per C4 assembly.)
  // write_mode: this gather is the FIRST contribution to the front (no clear beforehand): the block is written, not added to
  double cur[4] = {0.0, 0.0, 0.0, 0.0};
  if (!(write_mode & 1)) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = min(kk + 4 * r, B.da - 1), j = min(cc, B.db - 1);
      cur[r] = A[(size_t)(B.pa + i) * ld + B.pb + j];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int i = kk + 4 * r, j = cc;
    if (i < B.da && j < B.db && (!diag || i <= j)) A[(size_t)(B.pa + i) * ld + B.pb + j] = cur[r] - sum[r];
  }
}

// 16 waves per separator variable (BAL: ~1000 factors per camera):  [A_v^T A_v | A_v^T b]  as one MFMA per factor,
// A operand lane (i, k): A_v[k][i];  B operand lane (j, k): A_v[k][j] for j < d, b[k] for j == d.
#define SCHUR_FW 16
__global__ __launch_bounds__(64 * SCHUR_FW) void schur_factor_kernel(const GVarBlock* __restrict__ blocks, const GVarEntry* __restrict__ entries,
                                                                      double* __restrict__ pool, int64_t f_off, int ld, int n, int masked) {
  __shared__ double part[SCHUR_FW][4][64];
  const GVarBlock B = blocks[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), d = B.dv;
  const int kk = lane >> 4, cc = lane & 15;
  const bool va = cc < d, vb = cc <= d;
  const int per = (B.count + SCHUR_FW - 1) / SCHUR_FW;
  const int cbeg = B.begin + wave * per, cend = min(B.begin + B.count, cbeg + per);
  double4s_t acc[4];
#pragma unroll
  for (int u = 0; u < 4; u++) acc[u] = double4s_t{0, 0, 0, 0};
  for (int base = cbeg; base < cend; base += 64) {
    const int cnt = min(64, cend - base);
    long long moff = 0, mmeta = 0;
    if (lane < cnt) {
      const GVarEntry E = entries[base + lane];
      moff = E.joff;
      mmeta = ((long long)(unsigned short)E.rows) | ((long long)(unsigned short)E.c0 << 16) | ((long long)(unsigned short)E.cb << 32);
    }
    for (int e = 0; e < cnt; e += 4) {
      if (masked) {  // (as in schur_pairs_kernel: factors of at most four rows -- BAL's have two -- take masked loads)
        double av[4], bv[4];
        bool kav[4], kbv[4];
        bool small = true;
        long long offs[4], metas[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int eu = min(e + u, cnt - 1);
          offs[u] = readlane_ll(moff, eu);
          metas[u] = readlane_ll(mmeta, eu);
          small = small && (int)(metas[u] & 0xffff) <= 4;
        }
        if (small) {  // (wave-uniform)
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int m = (int)(metas[u] & 0xffff), c0 = (int)((metas[u] >> 16) & 0xffff), cbc = (int)((metas[u] >> 32) & 0xffff);
            const double* J = pool + offs[u];
            const bool valid = e + u < cnt;
            const int bcol = (cc < d) ? c0 + cc : cbc;
            kav[u] = valid && va && kk < m;
            kbv[u] = valid && vb && kk < m;
            av[u] = schur_load_masked(J + (c0 + cc) * m + kk, __builtin_amdgcn_ballot_w64(kav[u]));
            bv[u] = schur_load_masked(J + bcol * m + kk, __builtin_amdgcn_ballot_w64(kbv[u]));
          }
          asm volatile("s_waitcnt vmcnt(0)"
                       : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3])::"memory");
#pragma unroll
          for (int u = 0; u < 4; u++)
            acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(kav[u] ? av[u] : 0.0, kbv[u] ? bv[u] : 0.0, acc[u], 0, 0, 0);
          continue;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int eu = min(e + u, cnt - 1);
        const long long off = readlane_ll(moff, eu), meta = readlane_ll(mmeta, eu);
        const int m = (int)(meta & 0xffff), c0 = (int)((meta >> 16) & 0xffff), cbc = (int)((meta >> 32) & 0xffff);
        const double* J = pool + off;
        const bool valid = e + u < cnt;
        const int bcol = (cc < d) ? c0 + cc : cbc;
        for (int k0 = 0; k0 < m; k0 += 4) {
          const int k = k0 + kk;
          const bool ka = valid && va && k < m, kb = valid && vb && k < m;
          const double av = J[ka ? (c0 + cc) * m + k : 0], bv = J[kb ? bcol * m + k : 0];
          acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(ka ? av : 0.0, kb ? bv : 0.0, acc[u], 0, 0, 0);
        }
      }
    }
  }
  double4s_t sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
  for (int r = 0; r < 4; r++) part[wave][r][lane] = sum[r];
  __syncthreads();
  if (wave != 0) return;
  sum = double4s_t{0, 0, 0, 0};
#pragma unroll
  for (int w = 0; w < SCHUR_FW; w++)
#pragma unroll
    for (int r = 0; r < 4; r++) sum[r] += part[w][r][lane];
  double* A = pool + f_off;
  double cur[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int i = min(kk + 4 * r, d - 1);
    cur[r] = A[(size_t)(B.pv + i) * ld + (cc < d ? B.pv + cc : n - 1)];
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int i = kk + 4 * r, j = cc;
    if (i < d && ((j < d && i <= j) || j == d)) A[(size_t)(B.pv + i) * ld + (j < d ? B.pv + j : n - 1)] = cur[r] + sum[r];
  }
}

__global__ void add_scalar_kernel(double* __restrict__ dst, const double* __restrict__ src) { *dst += *src; }

// upper blocks of a front that no gather list covers (write-mode gather: nothing else initialises them)
struct GZeroBlock {
  int32_t pa, pb;
  int16_t da, db;
};
__global__ __launch_bounds__(128) void zero_blocks_kernel(const GZeroBlock* __restrict__ blocks, double* __restrict__ pool, int64_t f_off, int ld) {
  const GZeroBlock B = blocks[blockIdx.x];
  double* A = pool + f_off;
  for (int idx = threadIdx.x; idx < B.da * B.db; idx += 128) {
    const int i = idx / B.db, j = idx - i * B.db;
    if (B.pa != B.pb || i <= j) A[(size_t)(B.pa + i) * ld + B.pb + j] = 0.0;
  }
}

}  // namespace lmgpu
