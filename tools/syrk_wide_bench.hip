// Experiment (not part of the product): a trailing-update tile built for v_mfma_f64_4x4x4_4b_f64.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Igtsam_personal_amd/csrc -o tools/syrk_wide_bench tools/syrk_wide_bench.hip
// One workgroup = 128 rows x 256 columns of C (4 waves, 64 x 128 each, 128 accumulator doubles per lane, ONE workgroup per CU):
// 21 flop per operand byte instead of 16, the operand registers double-buffered so that the LDS reads of k-step s + 1 are in
// flight under the 128 MFMAs of k-step s, the C tile transposed through LDS before its read-modify-write.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "kernels_dense.hpp"
using namespace lmgpu;

#define WKC 16
#define WLDA 144
#define WLDB 272
#define WSTAGE (WKC * (WLDA + WLDB))
#define WLDS_BYTES (2 * WSTAGE * 8)

__global__ __launch_bounds__(256, 1) void syrk_wide_kernel(double* __restrict__ A, int ld, int n, int p0, int kp, int r0, int r1) {
  extern __shared__ double sm[];
  const int ti = blockIdx.y, tjp = blockIdx.x;
  const int it0 = r0 + ti * 128, jt0 = r0 + tjp * 256;
  if (it0 >= r1 || jt0 >= n || jt0 + 255 < it0) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int i0 = it0 + wr * 64, j0 = jt0 + wc * 128;
  const bool active = (i0 < r1) && (j0 < n) && (j0 + 127 >= i0);
  const int kk = lane >> 4, cc = lane & 15;
  double acc[4][8][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 8; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[a][b][r] = 0.0;
  const int nchunk = (kp + WKC - 1) / WKC;
  const double* P = A + (size_t)p0 * ld;
  auto issue = [&](int c, int buf) {
    double* base = sm + (size_t)buf * WSTAGE;
#pragma unroll
    for (int q = 0; q < 12; q++) {
      const int id = wave * 12 + q;  // 0..47: 16 A rows, then 16 B rows in two halves
      const bool isA = id < 16;
      const int row = isA ? id : (id - 16) >> 1, half = isA ? 0 : (id - 16) & 1;
      double* dst = isA ? base + (size_t)row * WLDA : base + (size_t)WKC * WLDA + (size_t)row * WLDB + half * 128;
      const int krow = c * WKC + row;
      const int col0 = isA ? it0 : jt0 + half * 128;
      if (krow < kp) {
        glds_row(P + (size_t)krow * ld + col0 + lane * 2, dst);
      } else {
        dst[lane * 2] = 0.0;
        dst[lane * 2 + 1] = 0.0;
      }
    }
  };
  int rot[4];
#pragma unroll
  for (int r = 0; r < 4; r++) rot[r] = (cc + 4 * r) & 15;
  double af[2][4], bf[2][8][4];
  auto load_ops = [&](const double* sA, const double* sB, int ks, double(&a_)[4], double(&b_)[8][4]) {
#pragma unroll
    for (int a = 0; a < 4; a++) a_[a] = sA[(ks + kk) * WLDA + a * 16];
#pragma unroll
    for (int b = 0; b < 8; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) b_[b][r] = sB[(ks + kk) * WLDB + b * 16 + rot[r]];
  };
  auto mfmas = [&](const double(&a_)[4], const double(&b_)[8][4]) {
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 8; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[a][b][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a_[a], b_[b][r], acc[a][b][r], 0, 0, 0);
  };
  issue(0, 0);
  for (int c = 0; c < nchunk; c++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (c + 1 < nchunk) issue(c + 1, (c + 1) & 1);
    if (active) {
      const double* sA = sm + (size_t)(c & 1) * WSTAGE + wr * 64 + cc;
      const double* sB = sm + (size_t)(c & 1) * WSTAGE + (size_t)WKC * WLDA + wc * 128;
      load_ops(sA, sB, 0, af[0], bf[0]);
#pragma unroll
      for (int s4 = 0; s4 < WKC / 4; s4++) {
        if (s4 + 1 < WKC / 4) load_ops(sA, sB, 4 * (s4 + 1), af[(s4 + 1) & 1], bf[(s4 + 1) & 1]);
        mfmas(af[s4 & 1], bf[s4 & 1]);
      }
    }
  }
  __syncthreads();
  if (!active) return;
  const int di = lane >> 4, dblk = (lane >> 2) & 3, dj = lane & 3;
  double* tr = sm + (size_t)wave * (16 * 132);
  double* wr_p = tr + (4 * dblk + di) * 132 + dj;
  const double* rd_p = tr + kk * 132 + cc;
  const bool full = (i0 + 64 <= r1) && (j0 + 128 <= n) && (j0 >= i0 + 63);
#pragma unroll
  for (int a = 0; a < 4; a++) {
#pragma unroll
    for (int b = 0; b < 8; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) wr_p[b * 16 + 4 * ((dblk + r) & 3)] = acc[a][b][r];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int hb = 0; hb < 2; hb++) {
      double v[4][4], cur[4][4];
#pragma unroll
      for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) v[b][r] = rd_p[(4 * r) * 132 + (4 * hb + b) * 16];
      if (full) {
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
          for (int r = 0; r < 4; r++) cur[b][r] = A[(size_t)(i0 + a * 16 + kk + 4 * r) * ld + j0 + (4 * hb + b) * 16 + cc];
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
          for (int r = 0; r < 4; r++) A[(size_t)(i0 + a * 16 + kk + 4 * r) * ld + j0 + (4 * hb + b) * 16 + cc] = cur[b][r] - v[b][r];
      } else {
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int row = min(i0 + a * 16 + kk + 4 * r, r1 - 1), col = min(j0 + (4 * hb + b) * 16 + cc, n - 1);
            cur[b][r] = A[(size_t)row * ld + col];
          }
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int row = i0 + a * 16 + kk + 4 * r, col = j0 + (4 * hb + b) * 16 + cc;
            if (row < r1 && col < n && col >= row) A[(size_t)row * ld + col] = cur[b][r] - v[b][r];
          }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

// reference for the correctness check: plain fp64 on the device, one thread per entry
__global__ void syrk_ref_kernel(double* __restrict__ A, int ld, int n, int p0, int kp, int r0) {
  const int j = r0 + blockIdx.x * 64 + threadIdx.x, i = r0 + blockIdx.y;
  if (j >= n || j < i) return;
  double s = 0;
  for (int k = 0; k < kp; k++) s += A[(size_t)(p0 + k) * ld + i] * A[(size_t)(p0 + k) * ld + j];
  A[(size_t)i * ld + j] -= s;
}

int main() {
  const int n = 9001, ld = 9008;
  double *A, *B;
  const size_t bytes = (size_t)n * ld * 8 + 4096 * 8;
  (void)hipMalloc((void**)&A, bytes);
  (void)hipMalloc((void**)&B, bytes);
  std::vector<double> h((size_t)n * ld);
  for (size_t i = 0; i < h.size(); i++) h[i] = 1e-3 * ((i * 2654435761u) % 1000) - 0.5;
  (void)hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute((const void*)syrk_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WLDS_BYTES);
  {  // correctness on a ragged trailing matrix
    const int r0 = 7936, kp = 256, m = n - r0;
    hipLaunchKernelGGL(syrk_wide_kernel, dim3((m + 255) / 256, (m + 127) / 128), dim3(256), WLDS_BYTES, 0, A, ld, n, r0 - kp, kp, r0, n);
    hipLaunchKernelGGL(syrk_ref_kernel, dim3((m + 63) / 64, m), dim3(64), 0, 0, B, ld, n, r0 - kp, kp, r0);
    std::vector<double> ha((size_t)m * ld), hb((size_t)m * ld);
    (void)hipMemcpy(ha.data(), A + (size_t)r0 * ld, ha.size() * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hb.data(), B + (size_t)r0 * ld, hb.size() * 8, hipMemcpyDeviceToHost);
    double maxd = 0, maxv = 0;
    for (int i = 0; i < m; i++)
      for (int j = r0 + i; j < n; j++) {
        maxd = std::max(maxd, std::abs(ha[(size_t)i * ld + j] - hb[(size_t)i * ld + j]));
        maxv = std::max(maxv, std::abs(hb[(size_t)i * ld + j]));
      }
    // entries outside the upper triangle / beyond column n must be untouched
    bool untouched = true;
    for (int i = 1; i < m && untouched; i++)
      for (int j = r0; j < r0 + i; j++)
        if (ha[(size_t)i * ld + j] != h[(size_t)(r0 + i) * ld + j]) untouched = false;
    std::printf("check: max |wide - ref| = %.3e (max |ref| %.3e), lower triangle untouched: %s\n", maxd, maxv, untouched ? "yes" : "NO");
  }
  for (int kp : {256, 512})
    for (int r0 : {1024, 2304, 4608, 6912}) {
      const int m = n - r0;
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      const dim3 grid((m + 255) / 256, (m + 127) / 128);
      for (int w = 0; w < 3; w++) hipLaunchKernelGGL(syrk_wide_kernel, grid, dim3(256), WLDS_BYTES, 0, A, ld, n, r0 - kp, kp, r0, n);
      (void)hipEventRecord(e0, 0);
      const int reps = 10;
      for (int w = 0; w < reps; w++) hipLaunchKernelGGL(syrk_wide_kernel, grid, dim3(256), WLDS_BYTES, 0, A, ld, n, r0 - kp, kp, r0, n);
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      ms /= reps;
      std::printf("K = %4d m = %4d: wide tile (128 x 256, row-major grid) %7.1f us  %5.1f TFLOP/s\n", kp, m, 1e3 * ms,
                  2.0 * kp * ((double)m * (m + 1) / 2) / (ms * 1e-3) / 1e12);
    }
  return 0;
}
