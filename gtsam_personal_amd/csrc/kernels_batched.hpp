// Medium HBM fronts (one outer panel: nf <= 256, a few hundred columns) of a general sparse graph, ALL fronts of a tree
// level per launch: the same device bodies as the per-front kernels, with the front taken from blockIdx.y / blockIdx.z.
// A SLAM graph such as sphere2500 has ~80 such fronts on 20 levels; one launch sequence per front (six launches, each
// bound by launch + dependent-load latency) made them the largest cost of a solve.
#pragma once
#include "kernels_dense.hpp"
#include "kernels_potrf.hpp"
#include "kernels_step.hpp"

namespace lmgpu {

// One record per medium front, level by level: everything the kernels below need to find their data, so that a workgroup is ONE
// dependent load away from it (list -> front descriptor -> offsets was three, at ~3 us each on an otherwise idle device).
struct MedFront {
  FrontDesc F;
  int64_t f_off;
  int32_t ld, row_begin;
};
struct MedLevel {
  const MedFront* mf;  // this level's records
};

// deterministic row-owner assembly (kernels_dense.hpp: assemble_row_body) for all medium fronts of a level: grid (max rows / 4, fronts)
__global__ __launch_bounds__(256) void med_assemble_rows_kernel(MedLevel L, const int32_t* __restrict__ rowptr,
                                                                const RowSrc* __restrict__ src, const ChildRef* __restrict__ childs,
                                                                const int32_t* __restrict__ cmap, const FrontFac* __restrict__ ffac,
                                                                const FacDesc* __restrict__ fd, double* __restrict__ pool,
                                                                const int32_t* __restrict__ fxoff, double lambda_v, const double* __restrict__ lambda_p,
                                                                const double* __restrict__ dampw, const double* __restrict__ gex) {
  const MedFront M = L.mf[blockIdx.y];
  const FrontDesc& F = M.F;
  const int R = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (R >= F.n) return;
  assemble_row_body(F, M.f_off, M.ld, rowptr + M.row_begin, src, childs, cmap, ffac, fd, pool, R, true);
  // the damping of a frontal row by the wave that owns it (med_damp_kernel was a launch of its own: ~4.5 us per tree level)
  if (R < F.nf && (threadIdx.x & 63) == 0) {
    double* Arow = pool + M.f_off + (size_t)R * M.ld;
    const int xo = fxoff[F.fx_begin + R];
    Arow[R] += (lambda_p ? *lambda_p : lambda_v) * dampw[xo];
    if (gex) Arow[F.n - 1] += gex[xo];
  }
}

__global__ __launch_bounds__(256) void med_damp_kernel(MedLevel L, const int32_t* __restrict__ fxoff, double* __restrict__ pool, double lambda_v, const double* __restrict__ lambda_p,
                                                        const double* __restrict__ dampw, const double* __restrict__ gex) {
  const MedFront M = L.mf[blockIdx.y];
  const FrontDesc& F = M.F;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= F.nf) return;
  pool[M.f_off + (size_t)i * M.ld + i] += (lambda_p ? *lambda_p : lambda_v) * dampw[fxoff[F.fx_begin + i]];
  if (gex) pool[M.f_off + (size_t)i * M.ld + F.n - 1] += gex[fxoff[F.fx_begin + i]];
}

// the whole frontal block [0, nf) x [0, nf) of each front: one workgroup per front
__global__ __launch_bounds__(256) void med_diag_potrf_kernel(MedLevel L, double* __restrict__ pool, int* __restrict__ status, double* __restrict__ inv16) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  const MedFront M = L.mf[blockIdx.x];
  diag_potrf_body(pool + M.f_off, M.ld, M.F.nf, 0, M.F.nf, M.F.id, status, inv16 + (size_t)blockIdx.x * 16 * 256, dsm);
}

__global__ __launch_bounds__(256) void med_panel_trsm_kernel(MedLevel L, double* __restrict__ pool, const double* __restrict__ inv16) {
  __shared__ double I16[16][16][17];
  const MedFront M = L.mf[blockIdx.y];
  if ((int)blockIdx.x * 64 >= M.F.n - M.F.nf) return;
  panel_trsm_body(pool + M.f_off, M.ld, M.F.n, 0, M.F.nf, inv16 + (size_t)blockIdx.y * 16 * 256, I16, blockIdx.x);
}

// Trailing update of every medium front of a level, one workgroup per 32 x 32 quadrant (kernels_step.hpp: syrk_quadrant32, one
// 16 x 16 MFMA tile per wave, operands straight from L2).  A 128 x 128 tile per workgroup (syrk_tile) is the right shape for the dense
// root, where thousands of tiles share 256 CUs; here a front has one to ten of them and each is K / 4 x 16 dependent MFMAs per wave
// on a single CU (~2.7 us per 16 rows of K: 17-52 us per level of sphere2500) while 250 CUs idle.  grid: (4 x strips, strips, fronts).
__global__ __launch_bounds__(256) void med_syrk_kernel(MedLevel L, double* __restrict__ pool) {
  const MedFront M = L.mf[blockIdx.z];
  const int sj = blockIdx.x >> 2, si = blockIdx.y, S = (M.F.n - M.F.nf + 63) >> 6;
  if (si > sj || sj >= S) return;
  syrk_quadrant32(pool + M.f_off, M.ld, M.F.n, 0, M.F.nf, M.F.nf, si, sj, blockIdx.x & 3, nullptr);
}

}  // namespace lmgpu
