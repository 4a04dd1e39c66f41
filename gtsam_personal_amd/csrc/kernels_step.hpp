// One launch per outer panel i of an HBM front:   trailing update with panel i   +   factorisation of panel i+1.
// Logical workgroup ids (ticket order = start order) are laid out as
//   [ trailing tiles of tile rows 0 and 1  = the rows of panel i+1 ]  [ diagonal workgroups of panel i+1 ]
//   [ all other trailing tiles ]                                       [ row-panel (trsm) workgroups of panel i+1 ]
// The first group publishes a per-column-tile counter when a tile is stored; the diagonal workgroups start on it and run
// their potrf chain BESIDE the bulk of the trailing update (they occupy 4 of the 512 workgroup slots); the row-panel
// workgroups are dispatched last and fill the tail of the update.  The dependency chain of the whole factorisation is
// then  update_i -> (tail) -> update_{i+1}  instead of  update_i -> panel_{i+1} -> update_{i+1}  (look-ahead without a
// second stream).  Every inter-workgroup hand-off is the release/acquire protocol of kernels_potrf.hpp.
#pragma once
#include "kernels_dense.hpp"
#include "kernels_potrf.hpp"

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
};

#define STEP_LDS_BYTES (2 * 2 * SYRK_KC * SYRK_LDW * 8)
static_assert(STEP_LDS_BYTES >= PDF_LDS_BYTES, "panel roles reuse the update's LDS");

__global__ __launch_bounds__(256, 2) void step_kernel(StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ int s_bid, s_ok;
  if (threadIdx.x == 0) s_bid = (int)atomicAdd(&a.flags[0], 1u);
  __syncthreads();
  int t = s_bid;
  const int r0 = a.p0 + a.kp, m = a.n - r0;
  const int T = (m + 127) >> 7;  // tile rows = tile columns of the trailing matrix
  const bool next = a.kb_next > 0;
  const int nTA = next ? (T >= 2 ? 2 * T - 1 : T) : 0;
  const int nd = next ? (a.kb_next >> 6) : 0;
  const int nTiles = T * (T + 1) / 2;
  if (t < nTA) {  // tile rows 0 and 1, published per column tile
    const int ti = (t < T) ? 0 : 1, tj = (t < T) ? t : t - T + 1;
    syrk_tile(a.A, a.ld, a.n, a.p0, a.kp, r0, a.n, ti, tj, sm);
    pdf_publish(&a.flags[PDF_TA0 + tj], threadIdx.x == 0);
    return;
  }
  t -= nTA;
  if (t < nd) {
    panel_role(a.A, a.ld, a.n, a.nf, r0, a.kb_next, t, a.front_id, a.status, a.inv16, a.flags, sm, &s_ok, true);
    return;
  }
  t -= nd;
  if (t < nTiles - nTA) {  // remaining tiles, row-major over tile rows first..T-1
    int ti = next ? 2 : 0;
    if (!next) {
      // all tiles
    }
    int rem = t;
    while (rem >= T - ti) {
      rem -= T - ti;
      ti++;
    }
    syrk_tile(a.A, a.ld, a.n, a.p0, a.kp, r0, a.n, ti, ti + rem, sm);
    return;
  }
  t -= nTiles - nTA;
  panel_role(a.A, a.ld, a.n, a.nf, r0, a.kb_next, nd + t, a.front_id, a.status, a.inv16, a.flags, sm, &s_ok, true);
}

}  // namespace lmgpu
