// Development check + timing of the single-wave register Cholesky (kernels_potrf.hpp) against a host Cholesky.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Igtsam_personal_amd/csrc tools/potrf_bench.hip -o tools/potrf_bench
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels_potrf.hpp"
#ifndef POTRF_FN
#define POTRF_FN potrf64_wave_g4
#endif

using namespace lmgpu;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__global__ __launch_bounds__(64) void potrf_wave_kernel(const double* __restrict__ A, double* __restrict__ R, int reps, int* failed_out) {
  const int lane = threadIdx.x, kk = lane >> 4, cc = lane & 15;
  double4_t T[4][4];
  bool failed = false;
  for (int it = 0; it < reps; it++) {
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
      for (int h = g; h < 4; h++)
#pragma unroll
        for (int r = 0; r < 4; r++) T[g][h][r] = __builtin_nontemporal_load(&A[(16 * g + kk + 4 * r) * 64 + 16 * h + cc]);
    failed |= POTRF_FN(T);
    asm volatile("" ::: "memory");
  }
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int h = g; h < 4; h++)
#pragma unroll
      for (int r = 0; r < 4; r++) R[(16 * g + kk + 4 * r) * 64 + 16 * h + cc] = T[g][h][r];
  if (lane == 0) *failed_out = failed;
}

// the variant that also leaves the 16x16 inverses of the diagonal tiles: inv[g * 256 + i * 16 + c] = (R_gg^-1)[i][c]
__global__ __launch_bounds__(64) void potrf_wave_inv_kernel(const double* __restrict__ A, double* __restrict__ R, double* __restrict__ inv, int reps,
                                                           int* failed_out) {
  const int lane = threadIdx.x, kk = lane >> 4, cc = lane & 15;
  double4_t T[4][4], E[4];
  bool failed = false;
  for (int it = 0; it < reps; it++) {
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
      for (int h = g; h < 4; h++)
#pragma unroll
        for (int r = 0; r < 4; r++) T[g][h][r] = __builtin_nontemporal_load(&A[(16 * g + kk + 4 * r) * 64 + 16 * h + cc]);
    failed |= potrf64_wave_g4<true>(T, E);
    asm volatile("" ::: "memory");
  }
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int h = g; h < 4; h++)
#pragma unroll
      for (int r = 0; r < 4; r++) R[(16 * g + kk + 4 * r) * 64 + 16 * h + cc] = T[g][h][r];
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int r = 0; r < 4; r++) inv[g * 256 + cc * 16 + kk + 4 * r] = E[g][r];
  if (lane == 0) *failed_out = failed;
}

int main() {
  const int n = 64;
  std::vector<double> a(n * n), r(n * n, 0.0), ref(n * n, 0.0);
  srand(3);
  std::vector<double> m(n * n);
  for (auto& x : m) x = (rand() % 2001) / 1000.0 - 1.0;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      double s = (i == j) ? 0.5 : 0.0;
      for (int k = 0; k < n; k++) s += m[k * n + i] * m[k * n + j];
      a[i * n + j] = s;
    }
  // host upper Cholesky
  ref = a;
  for (int k = 0; k < n; k++) {
    const double d = std::sqrt(ref[k * n + k]);
    for (int j = k; j < n; j++) ref[k * n + j] /= d;
    for (int i = k + 1; i < n; i++)
      for (int j = i; j < n; j++) ref[i * n + j] -= ref[k * n + i] * ref[k * n + j];
  }
  double *dA, *dR;
  int* dF;
  CK(hipMalloc((void**)&dA, n * n * 8));
  CK(hipMalloc((void**)&dR, n * n * 8));
  CK(hipMalloc((void**)&dF, 4));
  CK(hipMemcpy(dA, a.data(), n * n * 8, hipMemcpyHostToDevice));
  CK(hipMemset(dR, 0, n * n * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int reps : {1, 1, 101}) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(potrf_wave_kernel, dim3(1), dim3(64), 0, 0, dA, dR, reps, dF);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("reps %3d: %.2f us total\n", reps, ms * 1e3);
  }
  CK(hipMemcpy(r.data(), dR, n * n * 8, hipMemcpyDeviceToHost));
  int f;
  CK(hipMemcpy(&f, dF, 4, hipMemcpyDeviceToHost));
  double maxrel = 0;
  for (int i = 0; i < n; i++)
    for (int j = i; j < n; j++) maxrel = std::fmax(maxrel, std::fabs(r[i * n + j] - ref[i * n + j]) / (std::fabs(ref[i * n + j]) + 1e-30));
  printf("failed=%d  max rel err vs host = %.3e   %s\n", f, maxrel, (maxrel < 1e-9 && !f) ? "OK" : "MISMATCH");
  bool ok = maxrel < 1e-9 && !f;
  // with inverses
  double* dI;
  CK(hipMalloc((void**)&dI, 4 * 256 * 8));
  CK(hipMemset(dR, 0, n * n * 8));
  for (int reps : {1, 1, 101}) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(potrf_wave_inv_kernel, dim3(1), dim3(64), 0, 0, dA, dR, dI, reps, dF);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("with inverses, reps %3d: %.2f us total\n", reps, ms * 1e3);
  }
  std::vector<double> inv(4 * 256);
  CK(hipMemcpy(r.data(), dR, n * n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(inv.data(), dI, 4 * 256 * 8, hipMemcpyDeviceToHost));
  double maxrel2 = 0, maxinv = 0;
  for (int i = 0; i < n; i++)
    for (int j = i; j < n; j++) maxrel2 = std::fmax(maxrel2, std::fabs(r[i * n + j] - ref[i * n + j]) / (std::fabs(ref[i * n + j]) + 1e-30));
  for (int g = 0; g < 4; g++)  // R_gg * Inv_g == I, Inv_g upper triangular with exact zeros below
    for (int i = 0; i < 16; i++)
      for (int c = 0; c < 16; c++) {
        double sacc = 0;
        for (int k = 0; k < 16; k++) sacc += (k >= i ? ref[(16 * g + i) * n + 16 * g + k] : 0.0) * inv[g * 256 + k * 16 + c];
        maxinv = std::fmax(maxinv, std::fabs(sacc - (i == c ? 1.0 : 0.0)));
        if (i > c && inv[g * 256 + i * 16 + c] != 0.0) maxinv = 1.0;
      }
  printf("with inverses: max rel err of R = %.3e, max |R Inv - I| = %.3e   %s\n", maxrel2, maxinv, (maxrel2 < 1e-9 && maxinv < 1e-9) ? "OK" : "MISMATCH");
  ok = ok && maxrel2 < 1e-9 && maxinv < 1e-9;
  return ok ? 0 : 1;
}
