// Compiles ONE chain signature with the library's run-time fusion (hiprtc) and says how it went -- no GPU needed: the compile
// itself is host work, only loading the result asks for a device.  A compiler crash kills this process, not a host's:
// tools/jit_compile_fuzz.py runs it over random voices.
//   make -C tests/cpp bin/jit_compile_check
#include <cstdio>
#include <cstring>
#include <string>
#include "jit.hpp"
int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: jit_compile_check <signature> [f64] [fma]\n"); return 2; }
  bool f64 = false, fma = false;
  for (int i = 2; i < argc; ++i) { f64 = f64 || !std::strcmp(argv[i], "f64"); fma = fma || !std::strcmp(argv[i], "fma"); }
  std::string err;
  const knh::JitKernel* k = knh::jit_voice_kernel(argv[1], f64, fma, &err);
  if (k) { std::puts("compiled and loaded"); return 0; }
  // with no device the compile is followed by a failing load: that is a successful compile
  if (err.rfind("hipGetDevice", 0) == 0 || err.rfind("hipModuleLoadData", 0) == 0) { std::puts("compiled"); return 0; }
  std::printf("FAILED: %.1500s\n", err.c_str());
  return 1;
}
