// voice_frame.hpp -- the frame-parallel form of a voice: a lane per FRAME, the voice's stages evaluated one after the other
// for all frames of the block at once.  Open to voices whose every stage is a pure function of the frame index (SinWt with
// its integer phase, x (op) value, a (op) b: kernels_interp.hip says why).  The stages themselves are not in this file: the
// host writes them as straight-line code, one statement per stage on the voice's signal variables, and hiprtc compiles it
// (jit.hip, jit_frame_kernel) -- no templates to unroll, so a voice of 1 500 stages builds in seconds -- or, where that is
// not wanted, kernels_interp.hip interprets them.  This is what surrounds them: staging of the table and of the voice's
// state words, parameter changes at block boundaries, the phases moving on, the write-back.
#pragma once
#include "voice_chain.hpp"

namespace knh_dev {

// sine: 16 384 floats of LDS; pw_all: [vpw][nw_padded] state words of LDS (nw_padded = n_state_words rounded up to 4);
// body(pw, sine, n) -> the voice's sample at frame n of the processed range; sin_slots: first state word of every SinWt stage.
template <typename F, typename Body>
__device__ __forceinline__ void frame_parallel_run(const VoiceKernelArgs<F>& a, F* rows, float* sine, typename WordOf<F>::type* pw_all,
                                                    u32 n_state_words, u32 nw_padded, u32 vpw, u32 threads_per_voice, const u32* sin_slots,
                                                    u32 n_sin, Body body) {
  typedef typename WordOf<F>::type W;
  const u32 n_frames = a.frame_end - a.frame_begin;
  const u32 vi = threadIdx.x / threads_per_voice;  // the voice of this thread within the workgroup
  const u32 tid = threadIdx.x % threads_per_voice, nthreads = threads_per_voice;
  W* pw = pw_all + (size_t)vi * nw_padded;
  const u32 voice_raw = blockIdx.x * vpw + vi;
  const bool have_voice = voice_raw < a.n_voices;
  const u32 voice = have_voice ? voice_raw : a.n_voices - 1;  // (spare threads of the last workgroup shadow the last voice and write nothing)
  {  // the table, 16 bytes per lane and step
    typedef float V4 __attribute__((ext_vector_type(4)));
    const V4* src = reinterpret_cast<const V4*>(a.sine_table);
    V4* dst = reinterpret_cast<V4*>(sine);
    for (u32 i = threadIdx.x; i < 4096u; i += blockDim.x) dst[i] = src[i];
  }
  for (u32 i = tid; i < n_state_words; i += nthreads) pw[i] = a.state[(long)i * a.stride + voice];
  u32 ev_i = 0, ev_end = 0;
  if (a.ev_start) { ev_i = a.ev_start[voice]; ev_end = a.ev_start[voice + 1]; }
  __syncthreads();
  const u32 n = tid;
  const bool live = n < n_frames;
  u32 base = 0;
  // parameter changes land at block boundaries (no stage of such a voice is wrapped in WrPreciseTiming)
  auto apply_changes_upto = [&](u32 frame_abs) {
    while (ev_i < ev_end && a.events[ev_i].frame <= frame_abs) {
      const Event e = a.events[ev_i];
      const u32 op = e.slot_op >> 24, slot = e.slot_op & 0xFFFFFFu;
      if ((op & 0x7Fu) == EV_SET && slot < n_state_words && tid == 0 && have_voice) {
        pw[slot] = (W)e.bits;
        a.state[(long)slot * a.stride + voice] = (W)e.bits;
      }
      ++ev_i;
    }
  };
  for (u32 blk = 0; blk < a.n_blocks; ++blk, base += a.block_size) {
    apply_changes_upto(base + a.frame_begin);
    __syncthreads();
    if (live) {
      const F y = body(pw, sine, n);
      if (have_voice) rows[((long)blk * a.n_voices + voice) * a.block_size + a.frame_begin + n] = y;
    }
    __syncthreads();
    for (u32 i = tid; i < n_sin; i += nthreads) {  // the oscillators move on by the frames just rendered
      const u32 s = sin_slots[i];
      pw[s] = (W)((u32)pw[s] + n_frames * (u32)pw[s + 2]);
    }
    __syncthreads();
    apply_changes_upto(base + a.frame_end);  // those due exactly at the end of the range (precise_timing.rs:85-103)
    __syncthreads();
  }
  if (have_voice)
    for (u32 i = tid; i < n_sin; i += nthreads) a.state[(long)sin_slots[i] * a.stride + voice] = pw[sin_slots[i]];
}

}  // namespace knh_dev
