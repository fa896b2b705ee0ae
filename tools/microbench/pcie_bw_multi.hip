// Pinned-memory PCIe bandwidth with the streaming path's transfer pattern: NS streams, each with its own pinned / device buffer pair of 206 MB (one
// 16cam6p batch of 8192 frames), H2D copies back to back; optionally D2H copies of 166 MB on NS more streams at the same time.
//   hipcc -O2 -o pcie_bw_multi pcie_bw_multi.hip ; ./pcie_bw_multi
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
int main() {
  const size_t nin = 206ull << 20, nout = 166ull << 20;
  for (int ns : {1, 2, 4, 8}) {
    std::vector<void *> hi(ns), di(ns), ho(ns), dob(ns);
    std::vector<hipStream_t> si(ns), so(ns);
    for (int i = 0; i < ns; ++i) {
      (void)hipHostMalloc(&hi[i], nin, hipHostMallocDefault); (void)hipMalloc(&di[i], nin);
      (void)hipHostMalloc(&ho[i], nout, hipHostMallocDefault); (void)hipMalloc(&dob[i], nout);
      (void)hipStreamCreateWithFlags(&si[i], hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&so[i], hipStreamNonBlocking);
    }
    for (int mode = 0; mode < 2; ++mode) {   // 0: H2D only, 1: H2D + D2H
      double best_in = 0, best_out = 0;
      for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        const int reps = 6;
        for (int k = 0; k < reps; ++k)
          for (int i = 0; i < ns; ++i) {
            (void)hipMemcpyAsync(di[i], hi[i], nin, hipMemcpyHostToDevice, si[i]);
            if (mode) (void)hipMemcpyAsync(ho[i], dob[i], nout, hipMemcpyDeviceToHost, so[i]);
          }
        for (int i = 0; i < ns; ++i) { (void)hipStreamSynchronize(si[i]); (void)hipStreamSynchronize(so[i]); }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (reps * ns * (double)nin / dt / 1e9 > best_in) { best_in = reps * ns * (double)nin / dt / 1e9; best_out = mode ? reps * ns * (double)nout / dt / 1e9 : 0; }
      }
      printf("%d streams per direction, %s: H2D %.1f GB/s%s", ns, mode ? "H2D + D2H" : "H2D only", best_in, mode ? "" : "\n");
      if (mode) printf(", D2H %.1f GB/s\n", best_out);
    }
    for (int i = 0; i < ns; ++i) { (void)hipHostFree(hi[i]); (void)hipFree(di[i]); (void)hipHostFree(ho[i]); (void)hipFree(dob[i]); (void)hipStreamDestroy(si[i]); (void)hipStreamDestroy(so[i]); }
  }
  return 0;
}
