// Development microbenchmark (not part of the product): times the dense-front kernels on a synthetic SPD matrix and
// (trailing-update tile at several sizes).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../gtsam_personal_amd/csrc tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

// phase stamps of the diagonal workgroups: flags[512 + 64 b + 2 slot] (two words per 64-bit s_memrealtime: the 100 MHz
// counter shared by all XCDs, so the stamps of different workgroups lie on one time line)
#define PDF_STAMP(flags, b, slot)                                                                  \
  do {                                                                                             \
    if ((b) < 4 && threadIdx.x == 0) {                                                             \
      unsigned long long t_;                                                                       \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
      *(unsigned long long*)&(flags)[512 + 64 * (b) + 2 * (slot)] = t_;                            \
    }                                                                                              \
  } while (0)
// where the workgroups of a chained launch spend their time: per class (head quadrants, head 64-tiles, diagonal, update tiles, row-panel)
// [count, ticks waiting for inputs, ticks working] + slot-time per 100 us bucket of the launch (100 MHz realtime counter)
__device__ unsigned long long g_prof[16], g_hist[256], g_t0, g_begin[1 << 17], g_waited[1 << 17];
__device__ __forceinline__ unsigned long long prof_now() {
  unsigned long long t_;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
  return t_;
}
#define CHAIN_PROF_BEGIN()                                  \
  do {                                                      \
    if (threadIdx.x == 0) {                                 \
      g_begin[blockIdx.x] = prof_now();                     \
      g_waited[blockIdx.x] = 0;                             \
      atomicMin(&g_t0, g_begin[blockIdx.x]);                \
    }                                                       \
  } while (0)
#define CHAIN_PROF_WAITED()                                 \
  do {                                                      \
    if (threadIdx.x == 0) g_waited[blockIdx.x] = prof_now(); \
  } while (0)
#define CHAIN_PROF_END(cls)                                                              \
  do {                                                                                   \
    __syncthreads();                                                                     \
    if (threadIdx.x == 0 && g_begin[blockIdx.x] != 0) { /* chained launches only */      \
      const unsigned long long e_ = prof_now(), b_ = g_begin[blockIdx.x];                \
      g_begin[blockIdx.x] = 0;                                                           \
      const unsigned long long w_ = g_waited[blockIdx.x] ? g_waited[blockIdx.x] : b_;    \
      atomicAdd(&g_prof[3 * (cls)], 1ull);                                               \
      atomicAdd(&g_prof[3 * (cls) + 1], w_ - b_);                                        \
      atomicAdd(&g_prof[3 * (cls) + 2], e_ - w_);                                        \
      const unsigned long long t0_ = g_t0;                                               \
      int it_ = 0;                                                                       \
      for (unsigned long long x_ = w_ - t0_; x_ < e_ - t0_ && it_ < 300; it_++) {        \
        const unsigned long long nb_ = (x_ / 10000 + 1) * 10000;                         \
        const unsigned long long y_ = nb_ < e_ - t0_ ? nb_ : e_ - t0_;                   \
        if (x_ / 10000 < 256) atomicAdd(&g_hist[x_ / 10000], y_ - x_);                   \
        x_ = y_;                                                                         \
      }                                                                                  \
    }                                                                                    \
  } while (0)
#include "kernels_dense.hpp"
#include "kernels_potrf.hpp"
#include "kernels_step.hpp"

using namespace lmgpu;

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e = (x);                                                         \
    if (e != hipSuccess) {                                                      \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

static void fill_spd(std::vector<double>& h, int n, int ld) {
  // diagonally dominant SPD: A = 0.01 * (random symmetric) + n * I   (upper triangle used)
  srand(1);
  for (int i = 0; i < n; i++)
    for (int j = i; j < n; j++) h[(size_t)i * ld + j] = (i == j) ? (double)n : 0.01 * ((rand() % 2001) / 1000.0 - 1.0);
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 9001;
  const int ld = (n + 15) & ~15;
  std::vector<double> h((size_t)n * ld + 1024, 0.0);
  fill_spd(h, n, ld);
  double* A;
  CK(hipMalloc((void**)&A, h.size() * sizeof(double)));
  unsigned long long* st;
  CK(hipMalloc((void**)&st, 64 * sizeof(unsigned long long)));
  int* status;
  CK(hipMalloc((void**)&status, sizeof(int)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int kSyrkLds = 2 * 2 * SYRK_KC * SYRK_LDW * 8;
  CK(hipFuncSetAttribute((const void*)syrk_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kSyrkLds));
  auto reset = [&]() { CK(hipMemcpy(A, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice)); };
  auto timeit = [&](const char* name, int reps, auto&& fn) {
    float best = 1e30f, tot = 0;
    for (int r = 0; r < reps; r++) {
      reset();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      fn();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
      tot += ms;
    }
    printf("%-44s best %9.1f us   avg %9.1f us\n", name, best * 1e3, tot / reps * 1e3);
  };
  const bool chain_only = argc > 3;  // microbench <n> <far_pct> <merge 0|1>: only the chained launch
  // 1. one outer panel as a dataflow launch: duration and the phase stamps of the four diagonal workgroups
  if (!chain_only) {
    unsigned int* flags;
    double* inv16;
    CK(hipMalloc((void**)&flags, PDF_FLAG_WORDS * 4));
    CK(hipMalloc((void**)&inv16, 16 * 256 * 8));
    CK(hipFuncSetAttribute((const void*)panel_dataflow_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PDF_LDS_BYTES));
    for (int k0 : {0, 4352}) {
      const int cols = n - k0 - 256, grid = 4 + (cols + 63) / 64;
      float best = 1e30f;
      std::vector<unsigned int> hf(PDF_FLAG_WORDS);
      for (int rep = 0; rep < 3; rep++) {
        reset();
        CK(hipMemset(flags, 0, PDF_FLAG_WORDS * 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(panel_dataflow_kernel, dim3(grid), dim3(256), PDF_LDS_BYTES, 0, A, ld, n, n - 1, k0, 256, 0, status, inv16, flags);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      CK(hipMemcpy(hf.data(), flags, PDF_FLAG_WORDS * 4, hipMemcpyDeviceToHost));
      printf("panel_dataflow k0=%d grid=%d: %.1f us\n", k0, grid, best * 1e3);
      auto st = [&](int b, int slot) { return *(unsigned long long*)&hf[512 + 64 * b + 2 * slot]; };
      const char* names[24] = {"start", "w0", "x0", "p0", "u0", "w1", "x1", "p1", "u1", "w2", "x2", "p2", "u2", "w3", "x3", "p3", "u3", "", "", "",
                               "gathered", "potrf", "stored", "published"};
      for (int b = 0; b < 4; b++) {
        printf("  wg %d (us from its own start; the counters of different XCDs are not aligned):", b);
        const unsigned long long t0 = st(b, 0);
        for (int slot = 1; slot < 24; slot++) {
          const unsigned long long t = st(b, slot);
          if (t) printf(" %s=%.1f", names[slot], (double)(long long)(t - t0) / 100.0);  // 100 MHz
        }
        printf("\n");
      }
    }
  }
  // 2. the fused step launch (update with panel at p0 + factorisation of the next panel) with the same stamps
  if (!chain_only) {
    unsigned int* flags;
    double* inv16;
    CK(hipMalloc((void**)&flags, PDF_FLAG_WORDS * 4));
    CK(hipMalloc((void**)&inv16, 16 * 256 * 8));
    CK(hipFuncSetAttribute((const void*)step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, STEP_LDS_BYTES));
    for (int p0 : {0, 4352, 6400}) {
      if (p0 + 512 >= n) continue;
      const int r0 = p0 + 256, m = n - r0, kbn = 256;
      const int grid = step_grid(m, kbn);
      float best = 1e30f;
      std::vector<unsigned int> hf(PDF_FLAG_WORDS);
      for (int rep = 0; rep < 3; rep++) {
        reset();
        CK(hipMemset(flags, 0, PDF_FLAG_WORDS * 4));
        CK(hipDeviceSynchronize());
        StepArgs a{A, ld, n, n - 1, p0, 256, kbn, 0, status, inv16, flags, nullptr};
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(step_kernel, dim3(grid), dim3(256), STEP_LDS_BYTES, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      CK(hipMemcpy(hf.data(), flags, PDF_FLAG_WORDS * 4, hipMemcpyDeviceToHost));
      printf("step_kernel p0=%d m=%d grid=%d: %.1f us (%.1f TFLOP/s)\n", p0, m, grid, best * 1e3, 256.0 * m * (m + 1) / (best * 1e-3) / 1e12);
      auto st = [&](int b, int slot) { return *(unsigned long long*)&hf[512 + 64 * b + 2 * slot]; };
      const char* names[24] = {"start", "w0", "x0", "p0", "u0", "w1", "x1", "p1", "u1", "w2", "x2", "p2", "u2", "w3", "x3", "p3", "u3", "", "", "",
                               "gathered", "potrf", "stored", "published"};
      names[0] = "rows ready";
      for (int b = 0; b < 4; b++) {
        printf("  wg %d (us from its own entry):", b);
        const unsigned long long t0 = st(b, 19);
        for (int slot = 0; slot < 24; slot++) {
          if (slot == 19) continue;
          const unsigned long long t = st(b, slot);
          if (t) printf(" %s=%.1f", names[slot], (double)(long long)(t - t0) / 100.0);  // 100 MHz
        }
        printf("\n");
      }
    }
  }
  // 2b. all fused steps of the matrix as ONE chained launch: total time and the time line of the diagonal workgroups per step
  {
    const int np = (n - 1 + 255) / 256;  // nf = n - 1
    unsigned int* flags;
    double* inv16;
    CK(hipMalloc((void**)&flags, (size_t)(np + 2) * PDF_FLAG_WORDS * 4));
    CK(hipMalloc((void**)&inv16, (size_t)(np + 1) * 16 * 256 * 8));
    CK(hipFuncSetAttribute((const void*)chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, STEP_LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)panel_dataflow_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PDF_LDS_BYTES));
    auto rows_of = [&](int i) { return std::min(n - 1, (i + 1) * 256) - i * 256; };
    for (int first : {0, 17}) {
      int nsteps = 0;
      while (first + nsteps + 1 < np && rows_of(first + nsteps) == 256 && rows_of(first + nsteps + 1) % 64 == 0 && n - (first + nsteps + 1) * 256 > 0) nsteps++;
      const std::vector<int2> tasks = chain_schedule(n, n - 1, first, nsteps, argc > 2 ? atoi(argv[2]) : 50, argc > 3 ? atoi(argv[3]) != 0 : true);
      int2* d_tasks;
      CK(hipMalloc((void**)&d_tasks, tasks.size() * sizeof(int2)));
      CK(hipMemcpy(d_tasks, tasks.data(), tasks.size() * sizeof(int2), hipMemcpyHostToDevice));
      ChainArgs ca{A, ld, n, n - 1, first, nsteps, 0, status, inv16, flags, d_tasks};
      reset();
      CK(hipMemset(flags, 0, (size_t)(np + 2) * PDF_FLAG_WORDS * 4));
      // panel `first` must be factored for the first step (its trailing data are whatever the matrix holds: timing only)
      hipLaunchKernelGGL(panel_dataflow_kernel, dim3(4 + (n - first * 256 - 256 + 63) / 64), dim3(256), PDF_LDS_BYTES, 0, A, ld, n, n - 1, first * 256, 256,
                         0, status, inv16, flags + (size_t)first * PDF_FLAG_WORDS);
      CK(hipDeviceSynchronize());
      {
        std::vector<unsigned long long> z(256, 0);
        const unsigned long long big = ~0ull;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z.data(), 16 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_hist), z.data(), 256 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_t0), &big, 8));
      }
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(chain_kernel, dim3((int)tasks.size()), dim3(256), STEP_LDS_BYTES, 0, ca);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("chain_kernel steps %d..%d (%d workgroups): %.1f us\n", first, first + ca.nsteps - 1, (int)tasks.size(), ms * 1e3);
      std::vector<unsigned int> hf((size_t)(np + 2) * PDF_FLAG_WORDS);
      CK(hipMemcpy(hf.data(), flags, hf.size() * 4, hipMemcpyDeviceToHost));
      auto st = [&](int region, int b, int slot) { return *(unsigned long long*)&hf[(size_t)region * PDF_FLAG_WORDS + 512 + 64 * b + 2 * slot]; };
      const unsigned long long t0 = st(first + 1, 0, 19);
      {
        unsigned long long pr[16], hi[256];
        CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_prof), 16 * 8));
        CK(hipMemcpyFromSymbol(hi, HIP_SYMBOL(g_hist), 256 * 8));
        const char* cn[5] = {"head quadrants", "head 64-tiles", "diagonal", "update tiles", "row-panel"};
        double tot_work = 0;
        for (int c = 0; c < 5; c++) {
          printf("  class %-14s: %6llu workgroups, waiting %9.1f us, working %10.1f us (avg %6.1f us each)\n", cn[c], pr[3 * c], pr[3 * c + 1] / 100.0,
                 pr[3 * c + 2] / 100.0, pr[3 * c] ? pr[3 * c + 2] / 100.0 / pr[3 * c] : 0.0);
          tot_work += pr[3 * c + 2] / 100.0;
        }
        printf("  slot occupancy (working workgroup-time / (512 slots x launch time)): %.3f\n", tot_work / (512.0 * ms * 1e3));
        printf("  working slots per 100 us bucket (of 512):");
        for (int b = 0; b < 256 && b * 100.0 < ms * 1e3; b++) printf(" %d", (int)(hi[b] / 10000.0 + 0.5));
        printf("\n");
      }
      {  // cadence of the whole chain: when panel s + 1 was complete (last diagonal workgroup published), and the flop done by then
        double prev = 0, flop_acc = 0;
        for (int sidx = 0; sidx < ca.nsteps; sidx++) {
          const int reg = first + sidx + 1;
          const double t = (double)(long long)(st(reg, 3, 23) - t0) / 100.0;
          const double m = n - (first + sidx + 1) * 256;
          flop_acc += 256.0 * m * (m + 1);
          printf("  cadence step %2d m=%5d: panel done at %8.1f us (+%6.1f)   updates of steps <= this one: %6.1f GF = %6.1f us at 44.8 TF\n", first + sidx,
                 (int)m, t, t - prev, flop_acc / 1e9, flop_acc / 44.8e12 * 1e6);
          prev = t;
        }
      }
      for (int sidx = std::max(0, ca.nsteps - 12); sidx < ca.nsteps; sidx++) {
        const int reg = first + sidx + 1;
        auto us = [&](int b, int slot) { return (double)(long long)(st(reg, b, slot) - t0) / 100.0; };
        printf("  step %2d (m=%4d): wg0 entry %.1f rows %.1f gathered %.1f potrf %.1f pub %.1f | wg1 w0 %.1f potrf %.1f pub %.1f | wg2 pub %.1f | wg3 pub %.1f\n",
               first + sidx, n - (first + sidx + 1) * 256, us(0, 19), us(0, 0), us(0, 20), us(0, 21), us(0, 23), us(1, 1), us(1, 21), us(1, 23), us(2, 23),
               us(3, 23));
      }
    }
  }
  if (chain_only) return 0;
  // 3. strip updates (K = 64, <= 192 rows) and big updates (K = 256) at several trailing sizes
  for (int r0 : {64, 4160, 8256}) {
    char nm[128];
    const int r1 = r0 + 192;
    snprintf(nm, sizeof nm, "syrk strip K=64 rows [%d,%d) cols..n", r0, r1);
    timeit(nm, 5, [&]() {
      const int Tr = (r1 - r0 + 127) / 128, Tc = (n - r0 + 127) / 128;
      hipLaunchKernelGGL(syrk_mfma_kernel, dim3(Tc, Tr), dim3(256), kSyrkLds, 0, A, ld, n, r0 - 64, 64, r0, r1);
    });
  }
  for (int r0 : {256, 2304, 4352, 6400, 8448}) {
    char nm[128];
    snprintf(nm, sizeof nm, "syrk big K=256 rows [%d,n)", r0);
    const double m = n - r0;
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
      reset();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      const int T = (n - r0 + 127) / 128;
      hipLaunchKernelGGL(syrk_mfma_kernel, dim3(T, T), dim3(256), kSyrkLds, 0, A, ld, n, r0 - 256, 256, r0, n);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    printf("%-44s best %9.1f us   %6.1f TFLOP/s\n", nm, best * 1e3, 256.0 * m * (m + 1) / (best * 1e-3) / 1e12);
  }
  CK(hipGetLastError());
  return 0;
}
