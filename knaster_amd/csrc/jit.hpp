// jit.hpp -- run-time fusion of chains that have no pre-built kernel: the device header
// (voice_chain.hpp, embedded in the library at build time) is handed to hiprtc with the chain's
// stage list as template arguments, compiled for gfx950 and cached per process.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace knh {

struct JitKernel {
  hipFunction_t fn = nullptr;  // voice_kernel<F, FMA, 1, Stages...> or voice_pipe_kernel<F, FMA, Group<...>...>
  std::string lowered_name;
  unsigned block_threads = 64;  // one wavefront, or (groups + 1) wavefronts for a pipelined kernel
};

// signature: kernel_registry.hpp's one-character-per-stage string.  Returns nullptr and sets *error on failure.
const JitKernel* jit_voice_kernel(const char* signature, bool f64, bool fma, std::string* error);

// The same chain as a wave pipeline: `cuts` holds the index of the first stage of every group after the first
// (ascending, inside the signature), e.g. "WmSA" with cuts {2, 3} = Group<W,m>, Group<S>, Group<A>.
const JitKernel* jit_pipe_kernel(const char* signature, const unsigned* cuts, unsigned n_cuts, bool f64, bool fma, std::string* error);

// Launch helper: args points at a VoiceKernelArgs<F>.
hipError_t jit_launch(const JitKernel* k, const void* args, size_t args_size, unsigned n_wavefronts, hipStream_t stream);

}  // namespace knh
