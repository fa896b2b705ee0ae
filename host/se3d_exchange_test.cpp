// CPU-only driver of the TEST-ONLY staged all-gather (se3d_exchange.hpp): `world` threads, each a rank with its own Collective on one rendezvous
// directory and plain host memory as "device" memory, run `calls` all-gathers of pseudo-random blocks of varying size; every rank must end every call
// with the concatenation of all ranks' blocks in rank order.  Prints one line; exit code 0 on success.  (tests/test_sync_cpu.py style: the test
// builds and runs it.)
//   se3d_exchange_test <world> <calls>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "se3d_exchange.hpp"

static unsigned char byte_of(int rank, int call, size_t i) { return (unsigned char)((rank * 131 + call * 31 + (int)(i * 7)) & 0xff); }

int main(int argc, char** argv) {
  const int world = argc > 1 ? std::atoi(argv[1]) : 3, calls = argc > 2 ? std::atoi(argv[2]) : 6;
  char tmpl[] = "/tmp/se3d_exchange_test_XXXXXX";
  const char* dir = mkdtemp(tmpl);
  if (!dir) return 2;
  std::vector<int> bad(world, 0);
  std::vector<std::thread> th;
  for (int r = 0; r < world; ++r)
    th.emplace_back([&, r] {
      try {
        auto coll = se3d::make_staged_collective(r, world, dir, se3d::host_mem_ops());
        for (int k = 0; k < calls; ++k) {
          const size_t n = 1000 + 977 * (size_t)k;
          std::vector<unsigned char> send(n), recv(n * world);
          for (size_t i = 0; i < n; ++i) send[i] = byte_of(r, k, i);
          coll->all_gather(send.data(), recv.data(), n, nullptr);
          for (int q = 0; q < world; ++q)
            for (size_t i = 0; i < n; ++i)
              if (recv[(size_t)q * n + i] != byte_of(q, k, i)) { ++bad[r]; break; }
        }
      } catch (const std::exception& e) {
        std::fprintf(stderr, "rank %d: %s\n", r, e.what());
        bad[r] = 1000;
      }
    });
  for (auto& t : th) t.join();
  int total = 0;
  for (int b : bad) total += b;
  const std::string rm = std::string("rm -rf '") + dir + "'";
  if (std::system(rm.c_str()) != 0) std::fprintf(stderr, "could not remove %s\n", dir);
  std::printf("staged all-gather: %d ranks x %d calls, %d mismatching blocks\n", world, calls, total);
  return total ? 1 : 0;
}
