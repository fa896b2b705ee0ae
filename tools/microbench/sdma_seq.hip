// A sequence of copy phases on ONE upload stream + ONE download stream (round 6: what flips the pair into the serialised state?).
//   ./sdma_seq <phases> [MB per copy] [copies per phase] [extra streams]
//   phases: B (both directions queued by one thread), U (uploads alone), D (downloads alone), T (eight threads, each a streaming context: upload -> wait ->
//   download -> wait, 12 rounds, two batches in flight), R / r (re-create the download stream / both), W (wait 50 ms)
//   extra streams: that many other non-blocking streams are created first and run a small kernel each (the contexts' compute streams)
// Prints GB/s per phase (both: upload rate + download rate over the phase's wall time).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(float *p) { p[threadIdx.x] += 1.f; }
int main(int argc, char **argv) {
  const char *ph = argc > 1 ? argv[1] : "BUDBDB";
  const size_t n = (size_t)(argc > 2 ? atoi(argv[2]) : 32) << 20;
  const int reps = argc > 3 ? atoi(argv[3]) : 4, extra = argc > 4 ? atoi(argv[4]) : 0;
  const int NT = 8;
  std::vector<void *> hi(NT), di(NT), ho(NT), dob(NT);
  for (int i = 0; i < NT; ++i) {
    (void)hipHostMalloc(&hi[i], n, hipHostMallocDefault); (void)hipMalloc(&di[i], n);
    (void)hipHostMalloc(&ho[i], n, hipHostMallocDefault); (void)hipMalloc(&dob[i], n);
  }
  std::vector<hipStream_t> others(extra);
  for (auto &s : others) { (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, (float *)di[0]); }
  for (auto &s : others) (void)hipStreamSynchronize(s);
  hipStream_t up, down;
  (void)hipStreamCreateWithFlags(&up, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&down, hipStreamNonBlocking);
  for (const char *p = ph; *p; ++p) {
    const double t0 = now();
    const char c = *p;
    if (c == 'R' || c == 'r') {
      (void)hipStreamDestroy(down); (void)hipStreamCreateWithFlags(&down, hipStreamNonBlocking);
      if (c == 'r') { (void)hipStreamDestroy(up); (void)hipStreamCreateWithFlags(&up, hipStreamNonBlocking); }
      printf("%c: re-created\n", c);
      continue;
    }
    if (c == 'W') { std::this_thread::sleep_for(std::chrono::milliseconds(50)); printf("W\n"); continue; }
    if (c == 'T') {
      const int rounds = 12;
      std::vector<std::thread> th;
      for (int i = 0; i < NT; ++i)
        th.emplace_back([&, i] {
          hipEvent_t in[2], out;
          (void)hipEventCreateWithFlags(&in[0], hipEventDisableTiming); (void)hipEventCreateWithFlags(&in[1], hipEventDisableTiming); (void)hipEventCreateWithFlags(&out, hipEventDisableTiming);
          hipStream_t cs = i < (int)others.size() ? others[i] : nullptr;
          // submit(k): upload on `up`, a small kernel on the context's stream after it; collect(k): wait for the kernel, download on `down`, wait for it
          auto submit = [&](int k) {
            (void)hipMemcpyAsync(di[i], hi[i], n, hipMemcpyHostToDevice, up);
            (void)hipEventRecord(in[k & 1], up);
            if (cs) { (void)hipStreamWaitEvent(cs, in[k & 1], 0); hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, cs, (float *)di[i]); }
          };
          submit(0);
          for (int k = 1; k <= rounds; ++k) {
            submit(k);
            if (cs) (void)hipStreamSynchronize(cs); else (void)hipEventSynchronize(in[(k - 1) & 1]);
            (void)hipMemcpyAsync(ho[i], dob[i], n, hipMemcpyDeviceToHost, down);
            (void)hipEventRecord(out, down);
            (void)hipEventSynchronize(out);
          }
          if (cs) (void)hipStreamSynchronize(cs);
          (void)hipStreamSynchronize(up);
        });
      for (auto &t : th) t.join();
      const double dt = now() - t0;
      printf("T: %.1f + %.1f GB/s\n", NT * (rounds + 1) * (double)n / 1e9 / dt, NT * rounds * (double)n / 1e9 / dt);
      continue;
    }
    for (int k = 0; k < reps; ++k) {
      if (c == 'B' || c == 'U') (void)hipMemcpyAsync(di[0], hi[0], n, hipMemcpyHostToDevice, up);
      if (c == 'B' || c == 'D') (void)hipMemcpyAsync(ho[0], dob[0], n, hipMemcpyDeviceToHost, down);
    }
    (void)hipStreamSynchronize(up); (void)hipStreamSynchronize(down);
    const double dt = now() - t0, gb = reps * (double)n / 1e9;
    if (c == 'B') printf("B: %.1f + %.1f GB/s\n", gb / dt, gb / dt);
    else printf("%c: %.1f GB/s\n", c, gb / dt);
  }
  return 0;
}
