// One tree level of a general sparse graph in ONE launch: its LDS fronts AND its medium dense fronts.
// The two kinds only depend on the levels below, but as launches of their own they ran one after the other -- an upper level of
// city10000 is ~45 us of LDS fronts (a handful of 100-column fronts, each a workgroup) followed by ~35 us of medium fronts (assembly ->
// diagonal block -> row panel -> trailing update, four launches of a few workgroups each) on a device that is idle but for them.  A
// second stream was measured in round 2 (every cross-stream edge of the replayed graph costs more than it hides); here both live in one
// grid: ticket k runs tasks[k], the host's list [LDS fronts | assembly rows | diagonal blocks | panel strips | update quadrants] of the
// level, and the four phases of a medium front hand over through per-front counters (release / acquire at agent scope) instead of
// launch boundaries.  A task only ever waits for tasks with lower tickets, and a ticket is drawn when the workgroup starts.
// The device bodies are the ones of the per-level launches (lds_front_body, assemble_row_body, diag_potrf_body, panel_trsm_body,
// syrk_quadrant32): same arithmetic in the same order.
#pragma once
#include "kernels_batched.hpp"

namespace lmgpu {

struct LevelTask {
  int32_t kind;   // 0 LDS front, 1 assembly of four rows, 2 diagonal block, 3 row-panel strip, 4 trailing-update quadrant
  int32_t front;  // kind 0: position in the level lists; else the medium front's index inside its level
  int32_t a, b;   // kind 1: first row; 3: strip; 4: a = si | sj << 8, b = quadrant
};
struct LevelSync {  // per medium front, cleared before every solve
  unsigned int rows_done, diag_done, strips_done, pad;
};

__device__ __forceinline__ bool level_wait_ge(const unsigned int* p, unsigned int want) {
  long spins = 0;
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > 4000000L) return false;
  }
  return true;
}

template <int MAXT>
__global__ __launch_bounds__(MAXT) void level_fused_kernel(const LevelTask* __restrict__ tasks, unsigned int* __restrict__ ticket,
                                                           const int32_t* __restrict__ list, const FrontDesc* __restrict__ fronts,
                                                           const FrontFac* __restrict__ ffac, const FacDesc* __restrict__ fd,
                                                           const ChildRef* __restrict__ childs, const int32_t* __restrict__ cmap,
                                                           const int32_t* __restrict__ fxoff, double* __restrict__ pool, double lambda_v,
                                                           const double* __restrict__ lambda_p, const double* __restrict__ dampw, int* __restrict__ status,
                                                           int nmax, int jcap, const double* __restrict__ gex, MedLevel L,
                                                           const int32_t* __restrict__ rowptr, const RowSrc* __restrict__ src, double* __restrict__ inv16,
                                                           LevelSync* __restrict__ sync) {
  extern __shared__ __attribute__((aligned(16))) double lvl_sm[];
  __shared__ int s_ticket, s_ok;
  if (threadIdx.x == 0) s_ticket = (int)atomicAdd(ticket, 1u);
  __syncthreads();
  const LevelTask t = tasks[s_ticket];
  if (t.kind == 0) {
    lds_front_body<false, MAXT, false, true>(t.front, list, fronts, ffac, fd, childs, cmap, fxoff, pool, lambda_v, lambda_p, dampw, status, nmax, nmax, (double*)nullptr, jcap,
                                       gex, (const char*)nullptr, 0, FrontFlow{});
    return;
  }
  if (threadIdx.x >= 256) return;  // the dense-front bodies are written for four waves (a wave that has left no longer counts at barriers)
  const MedFront M = L.mf[t.front];
  const FrontDesc& F = M.F;
  LevelSync* sy = sync + t.front;
  const int tid = threadIdx.x;
  auto wait_for = [&](const unsigned int* p, unsigned int want) {  // thread 0 looks, everybody acquires
    if (tid == 0) s_ok = level_wait_ge(p, want) ? 1 : 0;
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (!s_ok && tid == 0) atomicExch(status + 1, 1 + F.id);  // never expected: spin bound hit (reported apart from pivot failures)
  };
  auto publish_add = [&](unsigned int* p) {  // every wave's stores performed, then one release + count
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  if (t.kind == 1) {
    const int R = t.a + (tid >> 6);
    if (R < F.n) {
      assemble_row_body(F, M.f_off, M.ld, rowptr + M.row_begin, src, childs, cmap, ffac, fd, pool, R, true);
      if (R < F.nf && (tid & 63) == 0) {  // the damping of a frontal row by the wave that owns it
        double* Arow = pool + M.f_off + (size_t)R * M.ld;
        const int xo = fxoff[F.fx_begin + R];
        Arow[R] += (lambda_p ? *lambda_p : lambda_v) * dampw[xo];
        if (gex) Arow[F.n - 1] += gex[xo];
      }
    }
    publish_add(&sy->rows_done);
    return;
  }
  if (t.kind == 2) {
    wait_for(&sy->rows_done, (unsigned int)((F.n + 3) / 4));
    diag_potrf_body(pool + M.f_off, M.ld, F.nf, 0, F.nf, F.id, status, inv16 + (size_t)t.front * 16 * 256, lvl_sm);
    publish_add(&sy->diag_done);
    return;
  }
  if (t.kind == 3) {
    wait_for(&sy->diag_done, 1u);
    panel_trsm_body(pool + M.f_off, M.ld, F.n, 0, F.nf, inv16 + (size_t)t.front * 16 * 256, (double(*)[16][17])lvl_sm, t.a);
    publish_add(&sy->strips_done);
    return;
  }
  wait_for(&sy->strips_done, (unsigned int)((F.n - F.nf + 63) / 64));
  syrk_quadrant32(pool + M.f_off, M.ld, F.n, 0, F.nf, F.nf, t.a & 255, t.a >> 8, t.b, nullptr);
}

}  // namespace lmgpu
