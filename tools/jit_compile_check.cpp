// Compiles ONE chain signature with the library's run-time fusion (hiprtc) and says how it went -- no GPU needed: the compile
// itself is host work, only loading the result asks for a device.  A compiler crash kills this process, not a host's:
// tools/jit_compile_fuzz.py runs it over random voices.
//   make -C tests/cpp bin/jit_compile_check
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../include/knaster_hip.h"
#include "jit.hpp"
#include "jit_cache.hpp"    // --sha256: the cache's hash function against a known implementation (tests/test_jit_cache.py)
#include "stage_table.hpp"  // partition_chain: how knh_bank_init cuts a plain chain into pipeline groups
int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: jit_compile_check <signature> [f64] [fma] [pipe] | --sha256 <text>\n"); return 2; }
  if (!std::strcmp(argv[1], "--sha256")) {
    knh_jit::Sha256 h;
    const char* t = argc > 2 ? argv[2] : "";
    h.update(t, std::strlen(t));
    std::puts(h.hex().c_str());
    return 0;
  }
  struct PrintStats {  // where the kernel came from (knh_jit_stats)
    ~PrintStats() {
      uint64_t m = 0, d = 0, h = 0, i = 0;
      knh_jit_stats(&m, &d, &h, &i);
      std::printf("jit stats: memory %llu disk %llu helper %llu in-process %llu\n", (unsigned long long)m, (unsigned long long)d, (unsigned long long)h, (unsigned long long)i);
    }
  } print_stats;
  bool f64 = false, fma = false, pipe = false;
  for (int i = 2; i < argc; ++i) { f64 = f64 || !std::strcmp(argv[i], "f64"); fma = fma || !std::strcmp(argv[i], "fma"); pipe = pipe || !std::strcmp(argv[i], "pipe"); }
  std::string err;
  const knh::JitKernel* k = nullptr;
  if (pipe) {  // the pipelined form a plain chain of up to 512 voice groups gets at init
    unsigned cuts[2] = {0, 0};
    const unsigned n_cuts = partition_chain(argv[1], cuts);
    k = knh::jit_pipe_kernel(argv[1], cuts, n_cuts, f64, fma, &err);
  } else {
    k = knh::jit_voice_kernel(argv[1], f64, fma, &err);
  }
  if (k) { std::puts("compiled and loaded"); return 0; }
  // with no device the compile is followed by a failing load: that is a successful compile
  if (err.rfind("hipGetDevice", 0) == 0 || err.rfind("hipModuleLoadData", 0) == 0) { std::puts("compiled"); return 0; }
  std::printf("FAILED: %.1500s\n", err.c_str());
  return 1;
}
