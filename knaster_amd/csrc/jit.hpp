// jit.hpp -- run-time fusion of chains that have no pre-built kernel: the device header
// (voice_stages.hpp + voice_chain.hpp, embedded in the library at build time) is handed to hiprtc with the chain's
// stage list as template arguments, compiled for gfx950 -- in a helper process, so that a compiler crash is a status for the host
// (jit_cache.hpp) -- and cached per process and on disk.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace knh {

struct JitKernel {
  hipFunction_t fn = nullptr;  // voice_kernel<F, FMA, 1, Stages...> or voice_pipe_kernel<F, FMA, Group<...>...>
  std::string lowered_name;
  unsigned block_threads = 64;  // one wavefront, or (groups + 1) wavefronts for a pipelined kernel
  unsigned groups_per_workgroup = 1;  // 64-voice groups one workgroup renders (whole-chain wavefronts: 4, 8 or 16 of them)
};

// Where this process's kernels came from so far (knh_jit_stats): the in-memory table, the code-object cache on disk, a
// compile in the helper process, a compile in this process (no helper beside the library, or KNH_JIT_INPROCESS=1).
struct JitStats { unsigned long memory_hits = 0, disk_hits = 0, helper_runs = 0, in_process = 0; };
JitStats jit_stats();

// signature: kernel_registry.hpp's one-character-per-stage string.  Returns nullptr and sets *error on failure (an error
// that starts with "JIT_CRASH: " = the compiler died or hung in the helper process: KNH_ERR_INTERNAL, not a bad chain).
// waves: 1 = the one-wavefront kernel; 4 / 8 / 16 = that many whole-chain wavefronts (64-voice groups) per workgroup, sharing
// one staged sine table: the form of banks with more groups than the pipeline covers (voice_chain.hpp, voice_kernel's WAVES).
const JitKernel* jit_voice_kernel(const char* signature, bool f64, bool fma, std::string* error, unsigned waves = 1);

// The same chain as a wave pipeline: `cuts` holds the index of the first stage of every group after the first
// (ascending, inside the signature), e.g. "WmSA" with cuts {2, 3} = Group<W,m>, Group<S>, Group<A>.
const JitKernel* jit_pipe_kernel(const char* signature, const unsigned* cuts, unsigned n_cuts, bool f64, bool fma, std::string* error);

// A graph-shaped voice of SinWt oscillators and arithmetic as ONE frame-parallel kernel (voice_frame.hpp): the stages written
// out as straight-line code, a statement per stage (`ops`: kernel_registry.hpp's InterpOp list).  vpw voices per workgroup,
// threads_per_voice lanes each (a multiple of 64 >= block_size); n_state_words per voice; n_sig signal slots; out_sig the
// voice's signal.  block_threads of the result = vpw * threads_per_voice.
struct FrameOp { unsigned kind, a, b, o, slot; };
const JitKernel* jit_frame_kernel(const FrameOp* ops, unsigned n_ops, unsigned n_sig, unsigned out_sig, unsigned n_state_words, unsigned vpw,
                                  unsigned threads_per_voice, bool f64, std::string* error);
hipError_t jit_frame_launch(const JitKernel* k, const void* args, size_t args_size, unsigned n_workgroups, hipStream_t stream);

// Launch helper: args points at a VoiceKernelArgs<F>.
hipError_t jit_launch(const JitKernel* k, const void* args, size_t args_size, unsigned n_wavefronts, hipStream_t stream);

}  // namespace knh
