// se3d_sync_test.cpp — driver for tests/test_sync_cpu.py: reads "n_inputs queue_size age_penalty lower_bound_ns" then lines
// "input stamp_ns" from stdin, feeds them to se3d::ApproximateTimeSync in arrival order and prints every published tuple as one
// line of stamps (input order).  CPU only.
#include <cstdio>
#include <iostream>

#include "se3d_sync.hpp"

struct Stamped { int64_t stamp; };

int main() {
  int n;
  unsigned q;
  double pen;
  long long lb;
  if (!(std::cin >> n >> q >> pen >> lb)) return 2;
  se3d::ApproximateTimeSync<Stamped> sync(q, n, [](const Stamped& m) { return m.stamp; });
  sync.setAgePenalty(pen);
  sync.setInterMessageLowerBound(lb);
  sync.registerCallback([&](const std::vector<std::shared_ptr<const Stamped>>& t) {
    for (int i = 0; i < n; ++i) std::printf("%lld%c", (long long)t[i]->stamp, i + 1 < n ? ' ' : '\n');
  });
  int i;
  long long st;
  while (std::cin >> i >> st) sync.add(std::make_shared<const Stamped>(Stamped{st}), i);
  std::printf("warnings %d\n", sync.warnings());
  return 0;
}
