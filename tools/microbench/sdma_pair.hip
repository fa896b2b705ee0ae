// Does one upload stream + one download stream overlap on this runtime, and what decides it?  (round 6: the streaming path's slow mode —
// every step 55 ms instead of 32, both directions at ~60 % of their usual rate for a whole run, one default bench.py run in ten.)
// Each trial creates a fresh pair of non-blocking streams, issues the FIRST copies of the pair in a given order, then measures the steady state
// the streaming path produces: copies of 206 MB up / 166 MB down queued back to back on the two streams.
//   hipcc -O2 -o sdma_pair sdma_pair.hip ; ./sdma_pair [trials]
// first-copy orders:
//   0  upload alone (waited for), then download alone (waited for)
//   1  download issued while the first upload is in flight
//   2  download alone (waited for), then upload alone
//   3  upload issued while the first download is in flight
// history (emulates the contexts of earlier bench passes): 0 none, 1 twelve other streams did copies of both directions and were destroyed,
//   2 the same streams are still alive
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const int trials = argc > 1 ? atoi(argv[1]) : 4;
  const size_t nin = 206ull << 20, nout = 166ull << 20;
  void *hi, *di, *ho, *dob;
  (void)hipHostMalloc(&hi, nin, hipHostMallocDefault); (void)hipMalloc(&di, nin);
  (void)hipHostMalloc(&ho, nout, hipHostMallocDefault); (void)hipMalloc(&dob, nout);
  for (int history = 0; history < 3; ++history) {
    std::vector<hipStream_t> old;
    if (history) {
      old.resize(12);
      for (auto &s : old) {
        (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        (void)hipMemcpyAsync(di, hi, 1 << 20, hipMemcpyHostToDevice, s);
        (void)hipMemcpyAsync(ho, dob, 1 << 20, hipMemcpyDeviceToHost, s);
      }
      for (auto &s : old) (void)hipStreamSynchronize(s);
      if (history == 1) { for (auto &s : old) (void)hipStreamDestroy(s); old.clear(); }
    }
    for (int order = 0; order < 4; ++order)
      for (int t = 0; t < trials; ++t) {
        hipStream_t si, so;
        (void)hipStreamCreateWithFlags(&si, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&so, hipStreamNonBlocking);
        if (order == 0) {
          (void)hipMemcpyAsync(di, hi, nin, hipMemcpyHostToDevice, si); (void)hipStreamSynchronize(si);
          (void)hipMemcpyAsync(ho, dob, nout, hipMemcpyDeviceToHost, so); (void)hipStreamSynchronize(so);
        } else if (order == 1) {
          (void)hipMemcpyAsync(di, hi, nin, hipMemcpyHostToDevice, si);
          (void)hipMemcpyAsync(ho, dob, nout, hipMemcpyDeviceToHost, so);
          (void)hipStreamSynchronize(si); (void)hipStreamSynchronize(so);
        } else if (order == 2) {
          (void)hipMemcpyAsync(ho, dob, nout, hipMemcpyDeviceToHost, so); (void)hipStreamSynchronize(so);
          (void)hipMemcpyAsync(di, hi, nin, hipMemcpyHostToDevice, si); (void)hipStreamSynchronize(si);
        } else {
          (void)hipMemcpyAsync(ho, dob, nout, hipMemcpyDeviceToHost, so);
          (void)hipMemcpyAsync(di, hi, nin, hipMemcpyHostToDevice, si);
          (void)hipStreamSynchronize(si); (void)hipStreamSynchronize(so);
        }
        // steady state A: both directions queued back to back (what eight contexts with two batches in flight produce)
        const int reps = 6;
        double t0 = now();
        for (int k = 0; k < reps; ++k) {
          (void)hipMemcpyAsync(di, hi, nin, hipMemcpyHostToDevice, si);
          (void)hipMemcpyAsync(ho, dob, nout, hipMemcpyDeviceToHost, so);
        }
        (void)hipStreamSynchronize(si); (void)hipStreamSynchronize(so);
        const double dtA = now() - t0;
        // steady state B: one pair at a time (nothing queued behind)
        t0 = now();
        for (int k = 0; k < reps; ++k) {
          (void)hipMemcpyAsync(di, hi, nin, hipMemcpyHostToDevice, si);
          (void)hipMemcpyAsync(ho, dob, nout, hipMemcpyDeviceToHost, so);
          (void)hipStreamSynchronize(si); (void)hipStreamSynchronize(so);
        }
        const double dtB = now() - t0;
        // steady state C: small copies (26 MB up / 21 MB down: one context's batch), 48 pairs queued
        t0 = now();
        for (int k = 0; k < 48; ++k) {
          (void)hipMemcpyAsync(di, hi, nin / 8, hipMemcpyHostToDevice, si);
          (void)hipMemcpyAsync(ho, dob, nout / 8, hipMemcpyDeviceToHost, so);
        }
        (void)hipStreamSynchronize(si); (void)hipStreamSynchronize(so);
        const double dtC = now() - t0;
        printf("history %d order %d trial %d: queued %.1f + %.1f GB/s | pairwise %.1f + %.1f | small queued %.1f + %.1f\n", history, order, t,
               reps * nin / dtA / 1e9, reps * nout / dtA / 1e9, reps * nin / dtB / 1e9, reps * nout / dtB / 1e9, 6 * nin / dtC / 1e9, 6 * nout / dtC / 1e9);
        (void)hipStreamDestroy(si); (void)hipStreamDestroy(so);
      }
    for (auto &s : old) (void)hipStreamDestroy(s);
  }
  return 0;
}
