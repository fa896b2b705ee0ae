// Pinned-memory PCIe bandwidth of the box: H2D alone, D2H alone, both at once (two streams).  hipcc -O2 -o pcie_bw pcie_bw.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() {
  const size_t n = 512ull << 20;
  void *h0, *h1, *d0, *d1;
  (void)hipHostMalloc(&h0, n, hipHostMallocDefault); (void)hipHostMalloc(&h1, n, hipHostMallocDefault);
  (void)hipMalloc(&d0, n); (void)hipMalloc(&d1, n);
  hipStream_t s0, s1;
  (void)hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 8; ++i) {
        if (mode != 1) (void)hipMemcpyAsync(d0, h0, n, hipMemcpyHostToDevice, s0);
        if (mode != 0) (void)hipMemcpyAsync(h1, d1, n, hipMemcpyDeviceToHost, s1);
      }
      (void)hipStreamSynchronize(s0); (void)hipStreamSynchronize(s1);
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (rep) printf("%s: %.1f GB/s per direction\n", mode == 0 ? "H2D alone" : mode == 1 ? "D2H alone" : "H2D + D2H together", 8.0 * n / dt / 1e9);
    }
  }
  return 0;
}
