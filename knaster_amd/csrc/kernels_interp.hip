// kernels_interp.hip -- voices that are LARGE graphs of free-running oscillators and arithmetic (the reference's own
// "256 FM cascade" bench, knaster_benchmarks/benches/graph_dsp_performance.rs:37-72: 256 SinWt, 1 275 MathUGen/Constant
// nodes, ONE voice), evaluated frame-parallel by a small interpreter instead of being fused into one kernel per chain.
//
// Why not the fused form: a voice's stages unroll into one kernel with the voice in a lane, so a thousand stages are a
// thousand stages of straight-line code per sample (hiprtc does not finish the 1 531-stage cascade, and a single voice
// would use one lane of one wavefront).  What makes another mapping possible: every stage kind such a graph is made of is
// a pure function of the frame index --
//   SinWt (osc.rs:97-168): the phase is a u32 that advances by a constant increment, so frame j of the block reads
//                          table[((phase0 + j * inc + offset) >> 16) & 16383]: no sample depends on the one before;
//   x (op) value, a (op) b (math.rs:22-85, wrappers_core/math.rs): element-wise.
// So: one workgroup per voice, a lane per FRAME, and the stages in order, each reading and writing whole rows of signal
// slots in LDS (the host hands the slots out like registers: bank.hip build_signature).  The operations per sample are the
// same as in the fused kernel and in the reference, in the same order: results are bit-identical.
//
// Per block and voice: the voice's parameter words and the program (16 bytes per stage) are staged into LDS once; then per
// stage one broadcast read of the instruction, its parameter words, the operand rows, the result row.
#include <hip/hip_runtime.h>

#include "kernel_registry.hpp"
#include "voice_chain.hpp"

namespace knh_dev {

// vpw voices per workgroup (each with threads_per_voice = a multiple of 64 lanes): they share the table and the program, and
// their wavefronts hide each other's LDS latency (a stage is a chain of dependent LDS round trips).
template <typename F>
__global__ void __launch_bounds__(1024) voice_interp_kernel(VoiceKernelArgs<F> a, const InterpOp* prog, u32 n_ops, u32 n_state_words,
                                                             u32 n_sig, u32 out_sig, F* rows, u32 vpw, u32 threads_per_voice) {
  typedef typename WordOf<F>::type W;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // layout: sine table | program | per voice: its state words | per voice: signal rows [n_sig][frames]
  float* sine = reinterpret_cast<float*>(lds);
  InterpOp* ops = reinterpret_cast<InterpOp*>(lds + 16384 * sizeof(float));
  const u32 n_frames = a.frame_end - a.frame_begin;
  const size_t pw_bytes = ((size_t)n_state_words * sizeof(W) + 15u) & ~(size_t)15u;
  const u32 vi = threadIdx.x / threads_per_voice;      // the voice of this thread within the workgroup
  const u32 tid = threadIdx.x % threads_per_voice, nthreads = threads_per_voice;
  unsigned char* after_ops = reinterpret_cast<unsigned char*>(ops) + (size_t)n_ops * sizeof(InterpOp);
  W* pw = reinterpret_cast<W*>(after_ops + (size_t)vi * pw_bytes);
  F* sig = reinterpret_cast<F*>(after_ops + (size_t)vpw * pw_bytes) + (size_t)vi * n_sig * n_frames;
  const u32 voice_raw = blockIdx.x * vpw + vi;
  const bool have_voice = voice_raw < a.n_voices;
  const u32 voice = have_voice ? voice_raw : a.n_voices - 1;  // (a workgroup past the last voice: its spare threads shadow it and write nothing)
  {  // the table, 16 bytes per lane and step
    typedef float V4 __attribute__((ext_vector_type(4)));
    const V4* src = reinterpret_cast<const V4*>(a.sine_table);
    V4* dst = reinterpret_cast<V4*>(sine);
    for (u32 i = threadIdx.x; i < 4096u; i += blockDim.x) dst[i] = src[i];
  }
  for (u32 i = threadIdx.x; i < n_ops; i += blockDim.x) ops[i] = prog[i];
  for (u32 i = tid; i < n_state_words; i += nthreads) pw[i] = a.state[(long)i * a.stride + voice];
  u32 ev_i = 0, ev_end = 0;
  if (a.ev_start) { ev_i = a.ev_start[voice]; ev_end = a.ev_start[voice + 1]; }
  __syncthreads();
  const u32 n = tid;  // this lane's frame of the processed range
  const bool live = n < n_frames;
  u32 base = 0;
  // Parameter changes land at block boundaries (no stage of such a voice is wrapped in WrPreciseTiming): every thread walks
  // its voice's list, one of them writes the new word where the stages read it and where the next launch does.
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
      F* row = sig + n;  // signal s at frame n: row[s * n_frames]
      InterpOp nxt = ops[0];
      for (u32 i = 0; i < n_ops; ++i) {
        const InterpOp op = nxt;
        if (i + 1 < n_ops) nxt = ops[i + 1];  // (the next instruction is on its way while this one runs)
        const u32 kind = op.kind;
        F r;
        if (kind == INTERP_SIN_WT) {
          const u32 phase0 = (u32)pw[op.slot], off = (u32)pw[op.slot + 1], inc = (u32)pw[op.slot + 2];
          r = (F)sine[((phase0 + n * inc + off) >> 16) & 16383u];
        } else if (kind <= INTERP_VAL_LAST) {
          const F x = row[(u32)op.a * n_frames];
          const F v = word_to_f<F>(pw[op.slot]);
          switch (kind) {
            case INTERP_VAL_MUL: r = x * v; break;
            case INTERP_VAL_ADD: r = x + v; break;
            case INTERP_VAL_SUB: r = x - v; break;
            case INTERP_VAL_DIV: r = x / v; break;
            case INTERP_VAL_VSUB: r = v - x; break;
            default: r = v / x; break;  // INTERP_VAL_VDIV
          }
        } else {
          const F x = row[(u32)op.a * n_frames], y = row[(u32)op.b * n_frames];
          switch (kind) {
            case INTERP_MATH_MUL: r = x * y; break;
            case INTERP_MATH_ADD: r = x + y; break;
            case INTERP_MATH_SUB: r = x - y; break;
            default: r = x / y; break;  // INTERP_MATH_DIV
          }
        }
        row[(u32)op.o * n_frames] = r;
      }
      if (have_voice) rows[((long)blk * a.n_voices + voice) * a.block_size + a.frame_begin + n] = row[out_sig * n_frames];
    }
    __syncthreads();
    // the oscillators move on by the frames just rendered
    for (u32 i = tid; i < n_ops; i += nthreads)
      if (ops[i].kind == INTERP_SIN_WT) pw[ops[i].slot] = (W)((u32)pw[ops[i].slot] + n_frames * (u32)pw[ops[i].slot + 2]);
    __syncthreads();
    apply_changes_upto(base + a.frame_end);  // those due exactly at the end of the range (the fused kernels: precise_timing.rs:85-103)
    __syncthreads();
  }
  if (have_voice)
    for (u32 i = tid; i < n_ops; i += nthreads)
      if (ops[i].kind == INTERP_SIN_WT) a.state[(long)ops[i].slot * a.stride + voice] = pw[ops[i].slot];
}

}  // namespace knh_dev

namespace knh {
using namespace knh_dev;

size_t interp_lds_bytes(unsigned n_ops, unsigned n_state_words, unsigned n_sig, unsigned n_frames, bool f64, unsigned vpw) {
  const size_t w = f64 ? 8 : 4;
  return 16384 * sizeof(float) + (size_t)n_ops * sizeof(InterpOp) + (size_t)vpw * ((((size_t)n_state_words * w + 15u) & ~(size_t)15u) + (size_t)n_sig * n_frames * w);
}

template <typename F>
static hipError_t launch_interp(const VoiceKernelArgs<F>& a, const InterpOp* prog, unsigned n_ops, unsigned n_state_words, unsigned n_sig,
                                unsigned out_sig, F* rows, hipStream_t s) {
  const unsigned n_frames = a.frame_end - a.frame_begin;
  if (a.n_voices == 0 || n_frames == 0) return hipSuccess;
  const unsigned tpv = ((n_frames + 63u) / 64u) * 64u;  // threads per voice
  // voices per workgroup: as many as keep every CU busy (256 workgroups), fit 1 024 threads and the 160 KiB of LDS
  unsigned vpw = a.n_voices / 256u;
  if (vpw < 1u) vpw = 1u;
  if (vpw > 1024u / tpv) vpw = 1024u / tpv;
  while (vpw > 1u && interp_lds_bytes(n_ops, n_state_words, n_sig, n_frames, sizeof(F) == 8, vpw) > 158u * 1024u) --vpw;
  const size_t lds = interp_lds_bytes(n_ops, n_state_words, n_sig, n_frames, sizeof(F) == 8, vpw);
  if (lds > 64 * 1024) {  // more than 64 KiB of dynamic LDS has to be asked for (per device: set where the launch goes)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&voice_interp_kernel<F>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((voice_interp_kernel<F>), dim3((a.n_voices + vpw - 1u) / vpw), dim3(vpw * tpv), lds, s, a, prog, n_ops, n_state_words, n_sig,
                     out_sig, rows, vpw, tpv);
  return hipGetLastError();
}
hipError_t launch_interp_f32(const VoiceKernelArgs<float>& a, const InterpOp* prog, unsigned n_ops, unsigned n_state_words, unsigned n_sig,
                             unsigned out_sig, float* rows, hipStream_t s) {
  return launch_interp<float>(a, prog, n_ops, n_state_words, n_sig, out_sig, rows, s);
}
hipError_t launch_interp_f64(const VoiceKernelArgs<double>& a, const InterpOp* prog, unsigned n_ops, unsigned n_state_words, unsigned n_sig,
                             unsigned out_sig, double* rows, hipStream_t s) {
  return launch_interp<double>(a, prog, n_ops, n_state_words, n_sig, out_sig, rows, s);
}

}  // namespace knh
