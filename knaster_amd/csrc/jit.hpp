// jit.hpp -- run-time fusion of chains that have no pre-built kernel: the device header
// (voice_chain.hpp, embedded in the library at build time) is handed to hiprtc with the chain's
// stage list as template arguments, compiled for gfx950 and cached per process.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace knh {

struct JitKernel {
  hipFunction_t fn = nullptr;  // voice_kernel<F, FMA, 1, Stages...>
  std::string lowered_name;
};

// signature: kernel_registry.hpp's one-character-per-stage string.  Returns nullptr and sets *error on failure.
const JitKernel* jit_voice_kernel(const char* signature, bool f64, bool fma, std::string* error);

// Launch helper: args points at a VoiceKernelArgs<F>.
hipError_t jit_launch(const JitKernel* k, const void* args, size_t args_size, unsigned n_wavefronts, hipStream_t stream);

}  // namespace knh
