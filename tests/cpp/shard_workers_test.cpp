// The worker threads of host-sharded banks (knaster_amd/csrc/shard_workers.hpp), on the CPU and under
// ThreadSanitizer: every run() executes each index exactly once, returns only when all are done, sees the effects of the
// previous run, and the pool can be torn down at any point between runs.
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#include "../../knaster_amd/csrc/shard_workers.hpp"

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

int main() {
  for (int n : {1, 2, 3, 8}) {
    ShardWorkers w(n);
    std::vector<long> cell(static_cast<size_t>(n), 0);  // cell[k] is written by index k only: no atomics needed if run() synchronises
    std::vector<int> calls(static_cast<size_t>(n), 0);
    long expect = 0;
    for (int round = 0; round < 3000; ++round) {
      const long add = round % 7;
      w.run([&](int k) {
        cell[static_cast<size_t>(k)] += add + k;  // reads what the previous round left there
        calls[static_cast<size_t>(k)] += 1;
      });
      expect += add;
      for (int k = 0; k < n; ++k) {  // the caller reads every cell right after run(): must be complete and visible
        CHECK(cell[static_cast<size_t>(k)] == expect + static_cast<long>(round + 1) * k);
        CHECK(calls[static_cast<size_t>(k)] == round + 1);
      }
      if (round % 500 == 499) {  // a pause long enough for the workers to stop spinning and sleep
        struct timespec ts = {0, 3000000};
        nanosleep(&ts, nullptr);
      }
    }
  }
  {  // destruction without any run, and right after one
    ShardWorkers idle(4);
  }
  {
    ShardWorkers w(4);
    int hits[4] = {0, 0, 0, 0};
    w.run([&](int k) { hits[k] = 1; });
    CHECK(hits[0] + hits[1] + hits[2] + hits[3] == 4);
  }
  std::printf(failures ? "FAILED (%d)\n" : "ok   shard_workers\n", failures);
  return failures ? 1 : 0;
}
