// Products with the Bayes tree seen as a GaussianFactorGraph of unit-noise Jacobian factors [R S | d], one per clique
// (gtsam/linear/GaussianBayesTree.cpp:73-92) -- what DoglegOptimizer needs beside the solve:
//   forward    e_c = [R S] x - alpha d ,  sum ||e_c||^2      (GaussianFactorGraph::operator* :408-415 for R g;  error(x) with alpha = 1)
//   transpose  g  -= [R S]^T d                               (gradientAtZero :369-378, JacobianFactor.cpp:716-724)
// LDS-class fronts keep [R S d] as nf x n rows (strictly-lower part zeroed) at F.rsd_off; HBM fronts as rows 0..nf-1 of the dense
// front.  The squared norms go through per-clique / per-row buffers and the fixed-order reduction (reproducible).  The gradient too is a
// FIXED-order sum (round 3; FP64 atomics before): every clique writes the terms of its columns into a slot of their own in `part`, and
// one thread per scalar of the gradient adds the slots that belong to it in the order of a list the host built (bt_gather_kernel): a
// variable receives terms from its own clique and from every clique that has it in its separator.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_front.hpp"

namespace lmgpu {

// one wave per LDS-class front, 4 per block; out[li] = ||[R S] x - alpha d||^2
__global__ __launch_bounds__(256) void bt_lds_forward_kernel(const int32_t* __restrict__ list, int nlist, const FrontDesc* __restrict__ fronts,
                                                              const int32_t* __restrict__ fxoff, const int32_t* __restrict__ sxoff,
                                                              const double* __restrict__ pool, const double* __restrict__ x, double alpha,
                                                              double* __restrict__ out) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = blockIdx.x * 4 + w;
  if (li >= nlist) return;
  const FrontDesc F = fronts[list[li]];
  const int n = F.n, nf = F.nf, ns = n - nf - 1;
  const double* RSd = pool + F.rsd_off;
  double tot = 0;
  for (int i = 0; i < nf; i++) {
    const double* row = RSd + (size_t)i * F.ld_rsd;
    double s = 0;
    for (int j = i + lane; j < nf; j += 64) s += row[j] * x[fxoff[F.fx_begin + j]];
    for (int j = lane; j < ns; j += 64) s += row[nf + j] * x[sxoff[F.sx_begin + j]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const double e = s - alpha * row[n - 1];
    tot += e * e;
  }
  if (lane == 0) out[li] = tot;
}

// g[col] -= sum_i RSd[i][col] d_i   (lanes along the columns)
__global__ __launch_bounds__(256) void bt_lds_transpose_kernel(const int32_t* __restrict__ list, int nlist, const FrontDesc* __restrict__ fronts,
                                                                const int32_t* __restrict__ fxoff, const int32_t* __restrict__ sxoff,
                                                                const double* __restrict__ pool, const int64_t* __restrict__ part_off,
                                                                double* __restrict__ part) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = blockIdx.x * 4 + w;
  if (li >= nlist) return;
  const FrontDesc F = fronts[list[li]];
  const int n = F.n, nf = F.nf;
  const double* RSd = pool + F.rsd_off;
  double* out = part + part_off[li];
  for (int j = lane; j < n - 1; j += 64) {
    double s = 0;
    for (int i = 0; i < nf && i <= j; i++) s += RSd[(size_t)i * F.ld_rsd + j] * RSd[(size_t)i * F.ld_rsd + n - 1];
    out[j] = -s;
  }
}

// g[x] = sum of the slots listed for scalar x, in list order
__global__ __launch_bounds__(256) void bt_gather_kernel(const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx, const double* __restrict__ part, int n,
                                                         double* __restrict__ g) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= n) return;
  double s = 0;
  for (int e = ptr[x]; e < ptr[x + 1]; e++) s += part[idx[e]];
  g[x] = s;
}

// HBM front: one wave per row i < nf; out[i] = e_i^2
__global__ __launch_bounds__(64) void bt_hbm_forward_kernel(FrontDesc F, int64_t f_off, int ld, const int32_t* __restrict__ fxoff,
                                                             const int32_t* __restrict__ sxoff, const double* __restrict__ pool,
                                                             const double* __restrict__ x, double alpha, double* __restrict__ out) {
  const int i = blockIdx.x, lane = threadIdx.x;
  const int n = F.n, nf = F.nf, ns = n - nf - 1;
  const double* row = pool + f_off + (size_t)i * ld;
  double s = 0;
  for (int j = i + lane; j < nf; j += 64) s += row[j] * x[fxoff[F.fx_begin + j]];
  for (int j = lane; j < ns; j += 64) s += row[nf + j] * x[sxoff[F.sx_begin + j]];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const double e = s - alpha * row[n - 1];
  if (lane == 0) out[i] = e * e;
}

// HBM front: thread = column, blockIdx.y = chunk of 64 rows; slot (chunk, col) = - sum_{i in chunk, i <= col} A[i][col] d_i
__global__ __launch_bounds__(256) void bt_hbm_transpose_kernel(FrontDesc F, int64_t f_off, int ld, const double* __restrict__ pool, double* __restrict__ part) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int n = F.n, nf = F.nf;
  if (j >= n - 1) return;
  const int i0 = blockIdx.y * 64, i1 = min(min(nf, i0 + 64), j + 1);
  const double* A = pool + f_off;
  double s = 0;
  for (int i = i0; i < i1; i++) s += A[(size_t)i * ld + j] * A[(size_t)i * ld + n - 1];
  part[(size_t)blockIdx.y * (n - 1) + j] = -s;  // (a chunk below the column's diagonal contributes a zero)
}

}  // namespace lmgpu
