// Diagnostic (not part of the product): what bounds the 128x128 trailing-update tile, stand-alone on a C4-sized front.
// Self-contained (no product header) so that variants can be tried without touching the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/syrk4_bench tools/syrk4_bench.hip
// Round 3, first pass (profiles/r03/syrk_tile_experiments.txt): the matrix instruction is NOT the bound -- with neither operand fetch nor
// the read-modify-write of C the v_mfma_f64_16x16x4_f64 tile runs at 61 (K = 256) / 69 (K = 512) TFLOP/s, the same as three forms on
// v_mfma_f64_4x4x4_4b_f64 -- what costs is per tile: the epilogue (21 us of a 96 us tile), the prologue, the operand DMA waits.
// This file keeps the 16x16x4 form and varies those.
//   EPI 0 none | 1 four batches of 16 loads (round 2) | 2 software-pipelined batches | 3 no-return FP64 atomics
//       4 = 2 + the tile's C lines touched (dummy loads) PF chunks before the end, so that the epilogue's loads hit in the L2
//   KCV  k-rows per LDS stage (16: 73.7 KB per workgroup, 2 per CU; 8: 36.9 KB)      NWG  workgroups per CU asked for
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));
#define LDW 144

__device__ __forceinline__ void glds_row(const double* g, double* lds_row) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_row, 16, 0, 0);
}

template <bool FETCH, int EPI, int KCV, int PF>
__device__ __forceinline__ void tile(double* __restrict__ A, int ld, int n, int p0, int kp, int r0, int ti, int tj, double* sm) {
  if (tj < ti) return;
  const int it0 = r0 + ti * 128, jt0 = r0 + tj * 128;
  if (it0 >= n || jt0 >= n) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const bool diag = (ti == tj);
  const bool active = !(diag && wr > wc) && (it0 + wr * 64 < n) && (jt0 + wc * 64 < n);
  const int kk = lane >> 4, cc = lane & 15;
  const int i0 = it0 + wr * 64, j0 = jt0 + wc * 64;
  double4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = double4_t{0, 0, 0, 0};
  const int nchunk = kp / KCV;
  const double* P = A + (size_t)p0 * ld;
  constexpr int RPW = 2 * KCV / 4;  // row DMAs per wave and chunk
  auto issue = [&](int c, int buf) {
    if (!FETCH && c > 1) return;
    double* base = sm + (size_t)buf * 2 * KCV * LDW;
#pragma unroll
    for (int q = 0; q < RPW; q++) {
      const int rr = wave * RPW + q;
      const int op = rr / KCV, row = rr % KCV;
      double* dst = base + ((size_t)op * KCV + row) * LDW;
      const int col0 = (op == 0) ? it0 : jt0;
      glds_row(P + (size_t)(c * KCV + row) * ld + col0 + lane * 2, dst);
    }
  };
  int pf[4] = {0, 0, 0, 0};
  issue(0, 0);
  for (int c = 0; c < nchunk; c++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (c + 1 < nchunk) issue(c + 1, (c + 1) & 1);
    if (EPI == 4 && active && c == nchunk - PF) {
      // one dword of each of the wave's 256 C lines (64 rows x 4 lines): lane l -> rows l, line q
#pragma unroll
      for (int q = 0; q < 4; q++) pf[q] = *(const int*)(A + (size_t)min(i0 + lane, n - 1) * ld + min(j0 + 16 * q, n - 1));
    }
    if (active) {
      const double* sA = sm + (size_t)(c & 1) * 2 * KCV * LDW + wr * 64 + cc;
      const double* sB = sm + (size_t)(c & 1) * 2 * KCV * LDW + (size_t)KCV * LDW + wc * 64 + cc;
#pragma unroll
      for (int ks = 0; ks < KCV; ks += 4) {
        double af[4], bf[4];
#pragma unroll
        for (int a = 0; a < 4; a++) af[a] = sA[(ks + kk) * LDW + a * 16];
#pragma unroll
        for (int b = 0; b < 4; b++) bf[b] = sB[(ks + kk) * LDW + b * 16];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int b = 0; b < 4; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
      }
    }
  }
  if (!active) return;
  if (EPI == 4) asm volatile("" ::"v"(pf[0]), "v"(pf[1]), "v"(pf[2]), "v"(pf[3]));
  if (EPI == 0) {
    double t = 0;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) t += acc[a][b][r];
    if (t == 123.456) A[0] = t;
    return;
  }
  if constexpr (EPI == 3) {
    const bool full = (i0 + 64 <= n) && (j0 + 64 <= n) && (j0 >= i0 + 63);
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = i0 + a * 16 + kk + 4 * r, col = j0 + b * 16 + cc;
          if (full || (row < n && col < n && col >= row))
            __builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double*)(A + (size_t)row * ld + col), -acc[a][b][r]);
        }
  } else if constexpr (EPI == 2 || EPI == 4) {
    // software pipeline over eight 8-row half-slices: the loads of the next are in flight while the current one is stored
    double cv[2][4][2];
    auto load = [&](int h, double(&c)[4][2]) {
#pragma unroll
      for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const int row = min(i0 + (h >> 1) * 16 + kk + 4 * (2 * (h & 1) + r), n - 1), col = min(j0 + b * 16 + cc, n - 1);
          c[b][r] = A[(size_t)row * ld + col];
        }
    };
    load(0, cv[0]);
#pragma unroll
    for (int h = 0; h < 8; h++) {
      if (h + 1 < 8) load(h + 1, cv[(h + 1) & 1]);
#pragma unroll
      for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const int row = i0 + (h >> 1) * 16 + kk + 4 * (2 * (h & 1) + r), col = j0 + b * 16 + cc;
          if (row < n && col < n && col >= row) A[(size_t)row * ld + col] = cv[h & 1][b][r] - acc[h >> 1][b][2 * (h & 1) + r];
        }
    }
  } else {
#pragma unroll
    for (int a = 0; a < 4; a++) {
      double cv[4][4];
#pragma unroll
      for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = min(i0 + a * 16 + kk + 4 * r, n - 1), col = min(j0 + b * 16 + cc, n - 1);
          cv[b][r] = A[(size_t)row * ld + col];
        }
#pragma unroll
      for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = i0 + a * 16 + kk + 4 * r, col = j0 + b * 16 + cc;
          if (row < n && col < n && col >= row) A[(size_t)row * ld + col] = cv[b][r] - acc[a][b][r];
        }
    }
  }
}

#define SR 4
__device__ __forceinline__ void blocked_tile(int logical, int T, int* ti_out, int* tj_out) {
  int g = 0, rem = logical;
  for (;; g++) {
    const int r_lo = g * SR, r_hi = min(T, r_lo + SR);
    int cnt = 0;
    for (int r = r_lo; r < r_hi; r++) cnt += T - r;
    if (rem < cnt || r_hi >= T) break;
    rem -= cnt;
  }
  const int r_lo = g * SR, r_hi = min(T, r_lo + SR), h = r_hi - r_lo;
  int tj = r_lo, ti = r_lo;
  const int head = h * (h + 1) / 2;
  if (rem < head) {
    int q = 0;
    while (rem >= q + 1) {
      rem -= q + 1;
      q++;
    }
    tj = r_lo + q;
    ti = r_lo + rem;
  } else {
    rem -= head;
    tj = r_hi + rem / h;
    ti = r_lo + rem % h;
  }
  *ti_out = ti;
  *tj_out = tj;
}

template <bool FETCH, int EPI, int KCV, int NWG, int PF>
__global__ __launch_bounds__(256, NWG) void syrk_kernel(double* __restrict__ A, int ld, int n, int p0, int kp, int r0, int ntiles) {
  extern __shared__ double sm[];
  const int per_xcd = (ntiles + 7) >> 3;
  const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (logical >= ntiles) return;
  const int T = (n - r0 + 127) / 128;
  int ti, tj;
  blocked_tile(logical, T, &ti, &tj);
  tile<FETCH, EPI, KCV, PF>(A, ld, n, p0, kp, r0, ti, tj, sm);
}

template <bool FETCH, int EPI, int KCV, int NWG, int PF>
static float run(double* A, int ld, int n, int kp, int r0, int reps) {
  const int lds = 2 * 2 * KCV * LDW * 8;
  auto k = syrk_kernel<FETCH, EPI, KCV, NWG, PF>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int m = n - r0, T = (m + 127) / 128, ntiles = T * (T + 1) / 2, grid = (ntiles + 7) & ~7;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, A, ld, n, r0 - kp, kp, r0, ntiles);
  (void)hipEventRecord(e0, 0);
  for (int w = 0; w < reps; w++) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, A, ld, n, r0 - kp, kp, r0, ntiles);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

template <int EPI, int KCV, int NWG, int PF>
static bool check(int n, int ld) {
  const int kp = 64, r0 = 64;
  std::vector<double> h((size_t)n * ld);
  for (size_t i = 0; i < h.size(); i++) h[i] = 1e-3 * (double)((i * 2654435761u) % 1000) - 0.5;
  double* A;
  (void)hipMalloc((void**)&A, h.size() * 8);
  (void)hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  const int lds = 2 * 2 * KCV * LDW * 8;
  auto k = syrk_kernel<true, EPI, KCV, NWG, PF>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int m = n - r0, T = (m + 127) / 128, ntiles = T * (T + 1) / 2, grid = (ntiles + 7) & ~7;
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, A, ld, n, r0 - kp, kp, r0, ntiles);
  std::vector<double> g(h.size());
  (void)hipMemcpy(g.data(), A, h.size() * 8, hipMemcpyDeviceToHost);
  (void)hipFree(A);
  double worst = 0;
  bool lower_ok = true;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      double want = h[(size_t)i * ld + j];
      if (i >= r0 && j >= i) {
        for (int p = 0; p < kp; p++) want -= h[(size_t)(r0 - kp + p) * ld + i] * h[(size_t)(r0 - kp + p) * ld + j];
        worst = std::max(worst, std::fabs(want - g[(size_t)i * ld + j]));
      } else if (g[(size_t)i * ld + j] != want) {
        lower_ok = false;
      }
    }
  const bool ok = worst < 1e-11 && lower_ok;
  if (!ok) std::printf("epilogue %d KC %d NWG %d check FAILED: max err %.3e, outside untouched: %s\n", EPI, KCV, NWG, worst, lower_ok ? "yes" : "NO");
  return ok;
}

template <int EPI, int KCV, int NWG, int PF>
static void line(double* A, int ld, int n, const char* what) {
  if (!check<EPI, KCV, NWG, PF>(777, 784)) return;
  std::printf("%-58s", what);
  for (int kp : {256, 512})
    for (int r0 : {1024, 4608}) {
      const int m = n - r0;
      const double fl = 2.0 * kp * ((double)m * (m + 1) / 2);
      const float t = run<true, EPI, KCV, NWG, PF>(A, ld, n, kp, r0, 6);
      std::printf("  K%d m%d %5.1f", kp, m, fl / (t * 1e-3) / 1e12);
    }
  std::printf("\n");
}

int main() {
  const int n = 9001, ld = 9008;
  double* A;
  (void)hipMalloc((void**)&A, (size_t)n * ld * 8 + 1024 * 8);
  std::vector<double> h((size_t)n * ld);
  for (size_t i = 0; i < h.size(); i++) h[i] = 1e-3 * ((i * 2654435761u) % 1000) - 0.5;
  (void)hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  line<1, 16, 2, 0>(A, ld, n, "KC16 2wg, epilogue 4x16 (round 2)");
  line<0, 16, 2, 0>(A, ld, n, "KC16 2wg, no epilogue");
  line<2, 16, 2, 0>(A, ld, n, "KC16 2wg, pipelined epilogue (8-row half slices)");
  line<4, 16, 2, 1>(A, ld, n, "KC16 2wg, pipelined + C lines touched 1 chunk early");
  line<4, 16, 2, 2>(A, ld, n, "KC16 2wg, pipelined + C lines touched 2 chunks early");
  line<4, 16, 2, 4>(A, ld, n, "KC16 2wg, pipelined + C lines touched 4 chunks early");
  line<4, 16, 2, 8>(A, ld, n, "KC16 2wg, pipelined + C lines touched 8 chunks early");
  line<3, 16, 2, 0>(A, ld, n, "KC16 2wg, atomic epilogue");
  line<1, 8, 2, 0>(A, ld, n, "KC8 2wg, epilogue 4x16");
  line<1, 8, 3, 0>(A, ld, n, "KC8 3wg, epilogue 4x16");
  line<2, 8, 3, 0>(A, ld, n, "KC8 3wg, pipelined epilogue");
  line<0, 8, 3, 0>(A, ld, n, "KC8 3wg, no epilogue");
  line<4, 8, 3, 4>(A, ld, n, "KC8 3wg, pipelined + C lines touched 4 chunks early");
  line<3, 8, 3, 0>(A, ld, n, "KC8 3wg, atomic epilogue");
  return 0;
}
