// Latency of hbm_backsolve_blocks_kernel on ONE front, the way the upper levels of a SLAM clique tree run it (a few workgroups on an
// otherwise idle device); checked against a plain CPU back-substitution.  profiles/r02/backsolve_bench.txt has the history.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Igtsam_personal_amd/csrc -o tools/backsolve_bench tools/backsolve_bench.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "kernels_dense.hpp"
using namespace lmgpu;

int main() {
  for (auto sz : {std::pair<int, int>{330, 331}, {228, 517}, {114, 421}, {30, 199}, {546, 547}}) {
    const int nf = sz.first, n = sz.second, ns = n - nf - 1, ld = (n + 15) & ~15;
    std::vector<double> A((size_t)nf * ld, 0.0), delta(n, 0.0);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st % 2000001) / 1e6 - 1.0; };
    for (int i = 0; i < nf; i++) {
      for (int j = i; j < n; j++) A[(size_t)i * ld + j] = 0.05 * rnd();
      A[(size_t)i * ld + i] = 2.0 + rnd() * 0.5;
    }
    for (int j = 0; j < ns; j++) delta[nf + j] = rnd();
    // CPU reference
    std::vector<double> x(nf);
    for (int i = nf - 1; i >= 0; i--) {
      double s = A[(size_t)i * ld + n - 1];
      for (int j = 0; j < ns; j++) s -= A[(size_t)i * ld + nf + j] * delta[nf + j];
      for (int j = i + 1; j < nf; j++) s -= A[(size_t)i * ld + j] * x[j];
      x[i] = s / A[(size_t)i * ld + i];
    }
    FrontDesc F{};
    F.n = n; F.nf = nf; F.fx_begin = 0; F.sx_begin = 0; F.id = 0;
    std::vector<int32_t> fx(nf), sx(std::max(ns, 1));
    for (int i = 0; i < nf; i++) fx[i] = i;
    for (int j = 0; j < ns; j++) sx[j] = nf + j;
    double *dA, *dd; int32_t *dl, *dfx, *dsx, *dld; int64_t* doff; FrontDesc* dF; int* dst;
    (void)hipMalloc((void**)&dA, A.size() * 8); (void)hipMalloc((void**)&dd, n * 8);
    (void)hipMalloc((void**)&dl, 4); (void)hipMalloc((void**)&dfx, nf * 4); (void)hipMalloc((void**)&dsx, sx.size() * 4);
    (void)hipMalloc((void**)&dld, 4); (void)hipMalloc((void**)&doff, 8); (void)hipMalloc((void**)&dF, sizeof(F)); (void)hipMalloc((void**)&dst, 8);
    const int32_t zero = 0; const int64_t zero64 = 0; const int big = 0x7f7f7f7f;
    (void)hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dd, delta.data(), n * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dl, &zero, 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dfx, fx.data(), nf * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dsx, sx.data(), sx.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dld, &ld, 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(doff, &zero64, 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dF, &F, sizeof(F), hipMemcpyHostToDevice);
    (void)hipMemcpy(dst, &big, 4, hipMemcpyHostToDevice);
    std::vector<double> out(n);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    {  // the same front, one workgroup per 64-row block
      const int nblk = (nf + 63) / 64;
      std::vector<BsdBlock> tab;
      for (int b = nblk - 1; b >= 0; b--) tab.push_back(BsdBlock{0, ld, n, nf, 0, 0, 0, b, 0});
      BsdBlock* dtab; unsigned int* dtick; double* dx;
      (void)hipMalloc((void**)&dtab, tab.size() * sizeof(BsdBlock)); (void)hipMalloc((void**)&dtick, 4); (void)hipMalloc((void**)&dx, nblk * 64 * 8);
      (void)hipMemcpy(dtab, tab.data(), tab.size() * sizeof(BsdBlock), hipMemcpyHostToDevice);
      (void)hipMemcpy(dd, delta.data(), n * 8, hipMemcpyHostToDevice);
      auto launch2 = [&]() {
        (void)hipMemsetAsync(dtick, 0, 4, 0);
        (void)hipMemsetAsync(dx, 0xff, nblk * 64 * 8, 0);
        hipLaunchKernelGGL(hbm_backsolve_blocks_kernel, dim3(nblk), dim3(256), 0, 0, dtab, (unsigned int*)nullptr, dfx, dsx, dA, dd, dx, dst);
      };
      launch2();
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(out.data(), dd, n * 8, hipMemcpyDeviceToHost);
      double err2 = 0;
      for (int i = 0; i < nf; i++) err2 = std::max(err2, std::abs(out[i] - x[i]));
      for (int w = 0; w < 5; w++) launch2();
      (void)hipEventRecord(e0, 0);
      for (int w = 0; w < 50; w++) launch2();
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
      float ms0 = 0;
      (void)hipEventRecord(e0, 0);
      for (int w = 0; w < 50; w++) { (void)hipMemsetAsync(dtick, 0, 4, 0); (void)hipMemsetAsync(dx, 0xff, nblk * 64 * 8, 0); }
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms0, e0, e1);
      std::printf("nf %4d n %4d: max |x - ref| %.2e   %.1f us per launch incl. the two memsets (%.1f us for those alone)\n", nf, n, err2, 1e3 * ms / 50, 1e3 * ms0 / 50);
    }
  }
  return 0;
}
