// kernel_registry.hpp -- maps a chain signature to its pre-instantiated fused kernels.
//
// Signature: one character per device stage, in chain order:
//   W SinWt   R SinWt.ar_params() driven by the running signal   N SinNumeric
//   S SvfFilter   L OnePoleLpf   H OnePoleHpf   A x*EnvAsr   E x*EnvAr   V x*Envelope (segments)   D SampleDelay   P Phasor   X SafetyLimiter   B PolyBlep   Y AllpassDelay   Z AllpassFeedbackDelay   F BufferReader   U WhiteNoise   K PinkNoise   O BrownNoise   G RandomLin
//   J Pan2 (last stage only: two output channels)   I an input channel of the bank node (a source shared by all voices)
//   + - * / ^  MathUGen of two signals; "@a" / "@a,b" after a stage: the stage(s) whose output it reads, when not the one before it
//   m x*value   a x+value   s x-value   d x/value   v value-x   q value/x   p x.powf(value)   i x.powi(n)
#pragma once
#include <hip/hip_runtime.h>

#include "voice_chain.hpp"

namespace knh_dev {
// One stage of a voice evaluated by the frame-parallel interpreter (kernels_interp.hip): what it computes, the signal
// slots it reads and writes, and the first of its state words (the stage's slot base).
enum { INTERP_VAL_MUL = 0, INTERP_VAL_ADD, INTERP_VAL_SUB, INTERP_VAL_DIV, INTERP_VAL_VSUB, INTERP_VAL_VDIV, INTERP_VAL_LAST = INTERP_VAL_VDIV,
       INTERP_MATH_MUL, INTERP_MATH_ADD, INTERP_MATH_SUB, INTERP_MATH_DIV, INTERP_SIN_WT };
struct InterpOp { u32 kind; unsigned short a, b, o, pad; u32 slot; };  // 16 bytes
static_assert(sizeof(InterpOp) == 16, "one 16-byte LDS read per stage");
// Device-side resolution of WrPreciseTiming change queues (kernels_events.hip).  DevRec is the host's 24-byte record of one
// call (Bank::QRec: the same bytes); DevStage what the resolver needs to know of a stage.
struct DevRec {
  u32 voice;
  unsigned short delay;   // set_delay_within_block_for_param value, when the arm bit is set
  unsigned short stage;
  unsigned char param;
  unsigned char kb;       // bits 0-3 ParameterValue kind, bit 4 arm, bit 5 has a value
  unsigned short block;   // block of the launch the call is addressed to
  u32 pad;
  u64 value;              // f64 bits (Float) or the integer
};
static_assert(sizeof(DevRec) == 24, "one 24-byte record per call");
struct DevStage {
  unsigned short kind, dcpb, slot_base, param_base, flags, ar_param;
  short widx;             // index among the device-resolved wrapped stages (its queue state), -1: not one
  unsigned short pad;
};
struct EventResolveArgs {
  const DevRec* recs;       // pinned host memory, arrival order
  u32 n_recs;
  const DevStage* stages;   // device
  u32 n_voices, block_size, frame_begin, frame_end, n_blocks, sample_rate, f64;
  double f2pi;
  unsigned short* armed;    // device, [n_params_total][n_voices]: WrPreciseTiming::next_delay of every parameter (persistent)
  const u32* host_start;    // the host-made event list of the launch (pinned), or null
  const Event* host_events;
  u32 *cnt, *val_cnt, *cursor;  // device scratch, [n_voices] each, contiguous from cnt
  u32* rec_start;           // [n_voices + 1]
  u64* keys;                // [n_recs]
  DevRec* dev_recs;         // [n_recs]: the records in device memory (the counting kernel copies them: one pass over PCIe)
  u32* out_start;           // [n_voices + 1]: the launch's ev_start
  Event* out_events;        // [host events + value records]
  u32* overflow;            // mapped pinned host word: set when a change found its node's WrPreciseTiming queue full and was dropped
                            // (the reference logs "Not enough space for scheduled changes", precise_timing.rs:129-134; so does the host)
};
}  // namespace knh_dev

namespace knh {
hipError_t launch_resolve_events(const knh_dev::EventResolveArgs& a, hipStream_t s);

template <typename F>
using VoiceLaunchFn = hipError_t (*)(const knh_dev::VoiceKernelArgs<F>& args, unsigned n_wavefronts, hipStream_t stream);

struct KernelEntry {
  const char* signature;
  int n_slots;
  VoiceLaunchFn<float> f32[2];   // [allow_fma]
  VoiceLaunchFn<double> f64[2];  // [allow_fma]
};
// Wave-specialised (pipelined) variant of a chain, when one is built (voice_pipe.hpp).
struct PipeEntry {
  const char* signature;
  int n_groups;
  int form;  // knh_dev::PIPE_MIXER (32-sample tiles, f64: 16), PIPE_FOLD or PIPE_INPLACE (64-sample tiles, f64: 32): voice_pipe.hpp
  int gpw;   // 64-voice groups per workgroup: 1, or 2 (PIPE_INPLACE with the short tiles; for banks of more groups than CUs)
  int long_tiles;  // 1: PipeTile<F>::big frames per tile (64, f64: 32), 0: PipeTile<F>::value (32, f64: 16)
  VoiceLaunchFn<float> f32[2];
  VoiceLaunchFn<double> f64[2];
};
// the first entry for the chain whose form is in `forms` (bit i = form i)
const PipeEntry* find_pipe(const char* signature, unsigned forms = 7u, int groups_per_workgroup = 1);
// Five-role (dependence-cut) pipeline for source -> SVF -> x*envelope -> post chains, f32 banks (voice_dag.hpp).
struct DagEntry {
  const char* signature;
  VoiceLaunchFn<float> f32[2];
};
const DagEntry* find_dag(const char* signature);
// Many-wave builds of the single-wave kernel (4 or 8 voice groups per workgroup) for large banks.
struct WideEntry {
  const char* signature;
  VoiceLaunchFn<float> f32_w4[2], f32_w8[2];
  VoiceLaunchFn<double> f64_w4[2], f64_w8[2];
  VoiceLaunchFn<float> f32_w16[2];    // sixteen groups per workgroup: four wavefronts per SIMD (banks of thousands of groups)
  VoiceLaunchFn<double> f64_w16[2];
};
const WideEntry* find_wide(const char* signature);

// Voices that are large graphs of SinWt oscillators and arithmetic, a lane per frame (kernels_interp.hip).  rows:
// [n_blocks][n_voices][block_size], every voice's signal (the fold kernels take it from there).
size_t interp_lds_bytes(unsigned n_ops, unsigned n_state_words, unsigned n_sig, unsigned n_frames, bool f64, unsigned voices_per_workgroup = 1);
hipError_t launch_interp_f32(const knh_dev::VoiceKernelArgs<float>& a, const knh_dev::InterpOp* prog, unsigned n_ops, unsigned n_state_words,
                             unsigned n_sig, unsigned out_sig, float* rows, hipStream_t s);
hipError_t launch_interp_f64(const knh_dev::VoiceKernelArgs<double>& a, const knh_dev::InterpOp* prog, unsigned n_ops, unsigned n_state_words,
                             unsigned n_sig, unsigned out_sig, double* rows, hipStream_t s);

const KernelEntry* find_kernel(const char* signature);
int kernel_count();
const KernelEntry* kernel_at(int i);

// n_blocks consecutive [n_rows][row_len] row sets -> n_blocks consecutive [channels][out_stride] blocks.
// tree = false: exact left fold of the rows in order; tree = true: 16-ary two-level fold (deterministic)
// zero_flags (or null): two words the kernel also clears -- the flag set of the bank's next launch
// host (or null): `out` is mapped pinned host memory and the kernel signals the host when it is written (knh_dev::HostDone)
hipError_t launch_res_server_f32(const knh_dev::ResServerArgs<float>& a, hipStream_t s);
hipError_t launch_res_server_f64(const knh_dev::ResServerArgs<double>& a, hipStream_t s);
hipError_t launch_fold_f32(bool tree, const float* rows, unsigned n_rows, unsigned row_len, unsigned frame_begin,
                           unsigned frame_end, float* out, unsigned channels, unsigned out_stride, unsigned n_blocks, bool accumulate,
                           unsigned* zero_flags, hipStream_t s, const knh_dev::HostDone* host = nullptr);
hipError_t launch_fold_f64(bool tree, const double* rows, unsigned n_rows, unsigned row_len, unsigned frame_begin,
                           unsigned frame_end, double* out, unsigned channels, unsigned out_stride, unsigned n_blocks, bool accumulate,
                           unsigned* zero_flags, hipStream_t s, const knh_dev::HostDone* host = nullptr);

// Host-sharded banks (host_shards.hpp): out[i] (+)= shards[0][i] + shards[1][i] + ... in shard order, for the frames
// [frame_begin, frame_end) of every block of `block_size` frames; n = elements per shard, shard k starts at k * shard_stride.
hipError_t launch_sum_shards_f32(const float* shards, unsigned n_shards, size_t shard_stride, size_t n, unsigned block_size,
                                 unsigned frame_begin, unsigned frame_end, float* out, bool accumulate, hipStream_t s);
hipError_t launch_sum_shards_f64(const double* shards, unsigned n_shards, size_t shard_stride, size_t n, unsigned block_size,
                                 unsigned frame_begin, unsigned frame_end, double* out, bool accumulate, hipStream_t s);

}  // namespace knh
