// Diagnostic (not part of the product): the trailing-update kernel alone on a C4-sized front, to see what bounds it.
// Builds against the experimental v_mfma_f64_4x4x4_4b_f64 form of syrk_tile: apply tools/variants/syrk_tile_mfma4x4x4_xcd.patch first
// (git apply); results of round 2 in profiles/r02/syrk_4x4x4_experiments.txt.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Igtsam_personal_amd/csrc [-DSYRK_DBG_NO_FETCH] [-DSYRK_DBG_NO_EPILOGUE] -o tools/syrk_bench tools/syrk_bench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "kernels_dense.hpp"
using namespace lmgpu;

int main() {
  const int n = 9001, ld = 9008;
  double* A;
  (void)hipMalloc((void**)&A, (size_t)n * ld * 8 + 1024 * 8);
  std::vector<double> h((size_t)n * ld);
  for (size_t i = 0; i < h.size(); i++) h[i] = 1e-3 * ((i * 2654435761u) % 1000) - 0.5;
  (void)hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  const int lds = 2 * 2 * SYRK_KC * SYRK_LDW * 8;
  (void)hipFuncSetAttribute((const void*)syrk_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int kp : {256, 512, 1024})
  for (int r0 : {1024, 2304, 4608, 6912}) {
    const int m = n - r0, T = (m + 127) / 128;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(syrk_mfma_kernel, dim3(T, T), dim3(256), lds, 0, A, ld, n, r0 - kp, kp, r0, n);
    (void)hipEventRecord(e0, 0);
    const int reps = 10;
    for (int w = 0; w < reps; w++) hipLaunchKernelGGL(syrk_mfma_kernel, dim3(T, T), dim3(256), lds, 0, A, ld, n, r0 - kp, kp, r0, n);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::printf("K = %4d m = %4d (%4d tiles): row-major grid %7.1f us  %5.1f TFLOP/s", kp, m, T * (T + 1) / 2, 1e3 * ms, 2.0 * kp * ((double)m * (m + 1) / 2) / (ms * 1e-3) / 1e12);
    const int ntiles = T * (T + 1) / 2, grid = (ntiles + 7) & ~7;
    (void)hipFuncSetAttribute((const void*)syrk_xcd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(syrk_xcd_kernel, dim3(grid), dim3(256), lds, 0, A, ld, n, r0 - kp, kp, r0, n, ntiles);
    (void)hipEventRecord(e0, 0);
    for (int w = 0; w < reps; w++) hipLaunchKernelGGL(syrk_xcd_kernel, dim3(grid), dim3(256), lds, 0, A, ld, n, r0 - kp, kp, r0, n, ntiles);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::printf("   XCD-blocked %7.1f us  %5.1f TFLOP/s\n", 1e3 * ms, 2.0 * kp * ((double)m * (m + 1) / 2) / (ms * 1e-3) / 1e12);
  }
  return 0;
}
