// Diagnostic (not part of the product): how often can one SIMD issue v_mfma_f64_16x16x4_f64?
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_f64_rate tools/mfma_f64_rate.hip && tools/mfma_f64_rate
// For every (waves per SIMD, independent accumulators per wave) it prints the shader-clock cycles per MFMA and SIMD
// (in-kernel s_memtime), the clock held (s_memtime / s_memrealtime x 100 MHz) and the resulting TFLOP/s of the chip.
// 64 cycles per MFMA = 32 flop/clk/SIMD = the 78.6 TFLOP/s datasheet figure at 2.4 GHz.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC, bool RANDOM>
__global__ __launch_bounds__(256) void rate_kernel(double* out, unsigned long long* stamps, int iters) {
  double4_t acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = double4_t{0, 0, 0, 0};
  double a = RANDOM ? 1.0 + ((threadIdx.x * 2654435761u) % 1000) * 1.37e-4 : 0.0, b = RANDOM ? 1.0 - ((threadIdx.x * 40503u) % 997) * 2.1e-4 : 0.0;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_sched_barrier(0);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

// the same measurement for v_mfma_f64_4x4x4_4b_f64 (four 4x4x4 blocks: 512 flop) and for plain VALU v_fma_f64 (128 flop per wave instruction)
template <int NACC>
__global__ __launch_bounds__(256) void rate4x4_kernel(double* out, unsigned long long* stamps, int iters) {
  double acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = 0.0;
  double a = 1.0 + ((threadIdx.x * 2654435761u) % 1000) * 1.37e-4, b = 1.0 - ((threadIdx.x * 40503u) % 997) * 2.1e-4;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_sched_barrier(0);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}
template <int NACC>
__global__ __launch_bounds__(256) void ratefma_kernel(double* out, unsigned long long* stamps, int iters) {
  double acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = 1e-3 * i;
  double a = 1.0 + ((threadIdx.x * 2654435761u) % 1000) * 1.37e-7, b = ((threadIdx.x * 40503u) % 997) * 2.1e-7;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_sched_barrier(0);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}
template <typename K>
void run_other(const char* what, K kernel, double flop_per_wave_instr, int nacc, int cus, int wg_per_cu, int iters, double* out, unsigned long long* stamps) {
  const int blocks = cus * wg_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int r = 0; r < 30; r++) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, stamps, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, stamps, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)blocks * 2);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> clk;
  for (int b = 0; b < blocks; b++) clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0);
  std::sort(clk.begin(), clk.end());
  const double tf = (double)blocks * 4 * iters * nacc * flop_per_wave_instr / (ms * 1e-3) / 1e12;
  const double c = clk[clk.size() / 2];
  std::printf("%s, %d wave(s)/SIMD, %2d accumulators: clock %6.0f MHz, %5.1f TFLOP/s = %.1f flop/clk/SIMD\n", what, wg_per_cu, nacc, c, tf,
              tf * 1e12 / (cus * 4.0 * c * 1e6));
}

template <int NACC, bool RANDOM>
void run(int cus, int wg_per_cu, int iters, double* out, unsigned long long* stamps) {
  const int blocks = cus * wg_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int r = 0; r < 30; r++) hipLaunchKernelGGL((rate_kernel<NACC, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, stamps, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((rate_kernel<NACC, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, stamps, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)blocks * 2);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc, clk;
  for (int b = 0; b < blocks; b++) {
    cyc.push_back((double)h[2 * b] / ((double)iters * NACC * wg_per_cu));  // every SIMD of a CU carries one wave of each of its workgroups
    clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0);
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(clk.begin(), clk.end());
  const double tf = (double)blocks * 4 * iters * NACC * 2048.0 / (ms * 1e-3) / 1e12;
  std::printf("%s operands, %d wave(s)/SIMD, %2d accumulators: %6.1f cycles per MFMA and SIMD (median), clock %6.0f MHz, %5.1f TFLOP/s\n",
              RANDOM ? "random" : "zero  ", wg_per_cu, NACC, cyc[cyc.size() / 2], clk[clk.size() / 2], tf);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* out;
  unsigned long long* stamps;
  hipMalloc((void**)&out, (size_t)cus * 8 * 256 * 8);
  hipMalloc((void**)&stamps, (size_t)cus * 8 * 16);
  std::printf("%s, %d CUs\n", p.name, cus);
  for (int w : {1, 2, 4, 8}) {
    run<4, true>(cus, w, 20000 / w, out, stamps);
    run<8, true>(cus, w, 10000 / w, out, stamps);
    run<16, true>(cus, w, 5000 / w, out, stamps);
  }
  for (int w : {1, 2, 4, 8}) {
    run_other("v_mfma_f64_4x4x4_4b_f64", rate4x4_kernel<8>, 512.0, 8, cus, w, 20000, out, stamps);
    run_other("v_fma_f64 (VALU)       ", ratefma_kernel<16>, 128.0, 16, cus, w, 40000, out, stamps);
  }
  run<8, false>(cus, 1, 10000, out, stamps);
  run<8, false>(cus, 2, 5000, out, stamps);
  return 0;
}
