// Diagnostic: lane layout of v_mfma_f64_4x4x4_4b_f64 (and of its CBSZ / ABID broadcast controls), found by feeding unit vectors.
// For every (A lane la, B lane lb): a = e_la, b = e_lb, c = 0 -> prints which D lane (if any) becomes 1.
//   hipcc --offload-arch=gfx950 -O1 -o tools/mfma4x4_probe tools/mfma4x4_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int CBSZ, int ABID, int BLGP>
__global__ void probe(int* table) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      const double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, BLGP);
      // every lane reports; a table entry holds a bit mask over lanes in two ints
      const unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) {
        table[2 * (la * 64 + lb)] = (int)(m & 0xffffffffull);
        table[2 * (la * 64 + lb) + 1] = (int)(m >> 32);
      }
    }
}

template <int CBSZ, int ABID, int BLGP>
void run(int* d_table) {
  hipLaunchKernelGGL((probe<CBSZ, ABID, BLGP>), dim3(1), dim3(64), 0, 0, d_table);
  std::vector<int> h(2 * 64 * 64);
  (void)hipMemcpy(h.data(), d_table, h.size() * 4, hipMemcpyDeviceToHost);
  std::printf("CBSZ %d ABID %d BLGP %d\n", CBSZ, ABID, BLGP);
  for (int la = 0; la < 64; la++) {
    std::printf("a%02d:", la);
    for (int lb = 0; lb < 64; lb++) {
      const unsigned long long m = ((unsigned long long)(unsigned)h[2 * (la * 64 + lb) + 1] << 32) | (unsigned)h[2 * (la * 64 + lb)];
      if (!m) continue;
      std::printf(" b%02d->", lb);
      bool first = true;
      for (int l = 0; l < 64; l++)
        if (m >> l & 1) {
          std::printf("%sd%02d", first ? "" : "+", l);
          first = false;
        }
    }
    std::printf("\n");
  }
}

int main() {
  int* d_table;
  (void)hipMalloc((void**)&d_table, 2 * 64 * 64 * 4);
  run<0, 0, 0>(d_table);
  run<2, 0, 0>(d_table);
  run<2, 1, 0>(d_table);
  run<2, 3, 0>(d_table);
  run<1, 0, 0>(d_table);
  run<1, 1, 0>(d_table);
  run<0, 0, 1>(d_table);
  run<0, 0, 2>(d_table);
  run<0, 0, 4>(d_table);
  return 0;
}
