// How fast does one host core write 24-byte records (three 8-byte stores) into memory of each kind?  (The host side of a
// parameter change for a device-resolved node is exactly that: kernels_events.hip.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
static double run(uint64_t* out, size_t recs, int reps) {
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r)
    for (size_t w = 0; w < recs; ++w) {
      out[3 * w + 0] = w | ((uint64_t)r << 32);
      out[3 * w + 1] = w * 3 + r;
      out[3 * w + 2] = w ^ 0x5555;
    }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e9 / ((double)reps * recs);
}
int main() {
  const size_t recs = 65536;
  const int reps = 200;
  uint64_t* m = (uint64_t*)malloc(recs * 24);
  printf("malloc                       %.2f ns per record\n", run(m, recs, reps));
  struct { const char* name; unsigned flags; } kinds[] = {{"hipHostMalloc default      ", hipHostMallocDefault}, {"hipHostMallocNonCoherent   ", hipHostMallocNonCoherent},
      {"hipHostMallocCoherent      ", hipHostMallocCoherent}, {"hipHostMallocWriteCombined ", hipHostMallocWriteCombined}, {"Mapped | Coherent          ", hipHostMallocMapped | hipHostMallocCoherent}};
  for (auto& k : kinds) {
    uint64_t* p = nullptr;
    if (hipHostMalloc((void**)&p, recs * 24, k.flags) != hipSuccess) { printf("%s failed\n", k.name); continue; }
    printf("%s  %.2f ns per record\n", k.name, run(p, recs, reps));
    hipHostFree(p);
  }
  return 0;
}
