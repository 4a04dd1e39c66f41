// Medium HBM fronts (one outer panel: nf <= 256, a few hundred columns) of a general sparse graph, ALL fronts of a tree
// level per launch: the same device bodies as the per-front kernels, with the front taken from blockIdx.y / blockIdx.z.
// A SLAM graph such as sphere2500 has ~80 such fronts on 20 levels; one launch sequence per front (six launches, each
// bound by launch + dependent-load latency) made them the largest cost of a solve.
#pragma once
#include "kernels_dense.hpp"
#include "kernels_potrf.hpp"

namespace lmgpu {

struct MedLevel {
  const int32_t* list;       // front ids of this level's medium fronts
  const FrontDesc* fronts;
  const int64_t* f_off;
  const int32_t* f_ld;
};

__global__ __launch_bounds__(64) void med_assemble_factors_kernel(MedLevel L, const FrontFac* __restrict__ ffac, const FacDesc* __restrict__ fd,
                                                                   double* __restrict__ pool) {
  const int fi = L.list[blockIdx.y];
  const FrontDesc F = L.fronts[fi];
  if ((int)blockIdx.x >= F.fac_count) return;
  assemble_factor_body(F, L.f_off[fi], L.f_ld[fi], ffac, fd, pool, blockIdx.x);
}

__global__ __launch_bounds__(256) void med_assemble_children_kernel(MedLevel L, const ChildRef* __restrict__ childs, const int32_t* __restrict__ cmap,
                                                                    double* __restrict__ pool) {
  __shared__ int32_t smap[160];
  const int fi = L.list[blockIdx.y];
  const FrontDesc F = L.fronts[fi];
  if ((int)blockIdx.x >= F.child_count) return;
  assemble_child_body(F, L.f_off[fi], L.f_ld[fi], childs, cmap, pool, blockIdx.x, smap, blockIdx.z, gridDim.z);
}

// deterministic row-owner assembly (kernels_dense.hpp: assemble_row_body) for all medium fronts of a level: grid (max rows / 4, fronts)
__global__ __launch_bounds__(256) void med_assemble_rows_kernel(MedLevel L, const int32_t* __restrict__ row_begin, const int32_t* __restrict__ rowptr,
                                                                const RowSrc* __restrict__ src, const ChildRef* __restrict__ childs,
                                                                const int32_t* __restrict__ cmap, const FrontFac* __restrict__ ffac,
                                                                const FacDesc* __restrict__ fd, double* __restrict__ pool) {
  const int fi = L.list[blockIdx.y];
  const FrontDesc F = L.fronts[fi];
  const int R = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (R >= F.n) return;
  assemble_row_body(F, L.f_off[fi], L.f_ld[fi], rowptr + row_begin[fi], src, childs, cmap, ffac, fd, pool, R, true);
}

__global__ __launch_bounds__(256) void med_damp_kernel(MedLevel L, const int32_t* __restrict__ fxoff, double* __restrict__ pool, double lambda_v, const double* __restrict__ lambda_p,
                                                        const double* __restrict__ dampw, const double* __restrict__ gex) {
  const int fi = L.list[blockIdx.y];
  const FrontDesc F = L.fronts[fi];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= F.nf) return;
  pool[L.f_off[fi] + (size_t)i * L.f_ld[fi] + i] += (lambda_p ? *lambda_p : lambda_v) * dampw[fxoff[F.fx_begin + i]];
  if (gex) pool[L.f_off[fi] + (size_t)i * L.f_ld[fi] + F.n - 1] += gex[fxoff[F.fx_begin + i]];
}

// the whole frontal block [0, nf) x [0, nf) of each front: one workgroup per front
__global__ __launch_bounds__(256) void med_diag_potrf_kernel(MedLevel L, double* __restrict__ pool, int* __restrict__ status, double* __restrict__ inv16) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  const int fi = L.list[blockIdx.x];
  const FrontDesc F = L.fronts[fi];
  diag_potrf_body(pool + L.f_off[fi], L.f_ld[fi], F.nf, 0, F.nf, F.id, status, inv16 + (size_t)blockIdx.x * 16 * 256, dsm);
}

__global__ __launch_bounds__(256) void med_panel_trsm_kernel(MedLevel L, double* __restrict__ pool, const double* __restrict__ inv16) {
  __shared__ double I16[16][16][17];
  const int fi = L.list[blockIdx.y];
  const FrontDesc F = L.fronts[fi];
  if ((int)blockIdx.x * 64 >= F.n - F.nf) return;
  panel_trsm_body(pool + L.f_off[fi], L.f_ld[fi], F.n, 0, F.nf, inv16 + (size_t)blockIdx.y * 16 * 256, I16, blockIdx.x);
}

__global__ __launch_bounds__(256, 2) void med_syrk_kernel(MedLevel L, double* __restrict__ pool) {
  extern __shared__ double sm[];
  const int fi = L.list[blockIdx.z];
  const FrontDesc F = L.fronts[fi];
  syrk_tile(pool + L.f_off[fi], L.f_ld[fi], F.n, 0, F.nf, F.nf, F.n, blockIdx.y, blockIdx.x, sm);
}

}  // namespace lmgpu
