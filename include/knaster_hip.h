/* knaster_hip.h -- C ABI of the MI355X voice-bank engine for Knaster's UGen hot path.
 *
 * One `knh_bank` IS one UGen from the host graph's point of view
 * (Inputs = U0, Outputs = U1|U2): it evaluates N independent voices that share
 * one chain topology, fused into a single gfx950 kernel, and returns their
 * mixed block.  A Rust shim implements `knaster_core::UGen` by forwarding to
 * these entry points.  Every function cites the
 * reference interface it replaces; paths are relative to the knaster repo.
 *
 * Conventions
 *   - all entry points return a knh_status (0 = ok) and never throw or abort
 *     across the ABI; the message of the last failure on a handle is available
 *     from knh_last_error() (reference: "log and continue", never a Result on
 *     the audio path -- knaster_core/src/ugen.rs:328,340).
 *   - single caller, non-reentrant per handle (the reference calls init on the
 *     control thread and everything else from the one audio thread:
 *     knaster_graph/src/graph_gen.rs:110-200).  A handle may be destroyed from
 *     a different thread than the one that processed with it.
 *   - the library owns all device memory; `out` pointers are only used during
 *     the call (knaster_graph/src/task.rs:26-29).
 *   - plain C types only: no HIP or torch types appear in a signature; a HIP
 *     stream is passed as `void*`.
 */
#ifndef KNASTER_HIP_H
#define KNASTER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KNH_ABI_VERSION 4

typedef enum knh_status {
  KNH_OK = 0,
  KNH_ERR_INVALID_ARGUMENT = 1,  /* null handle, bad enum, wrong state            */
  KNH_ERR_OUT_OF_RANGE = 2,      /* voice/stage/param index out of bounds: no-op  */
  KNH_ERR_UNSUPPORTED_CHAIN = 3, /* run-time fusion (hiprtc) of the chain failed   */
  KNH_ERR_DEVICE = 4,            /* a HIP call failed; see knh_last_error          */
  KNH_ERR_NOT_INITIALISED = 5,   /* process/param before knh_bank_init             */
  KNH_ERR_NO_DEVICE = 6,         /* no gfx950 device visible: the product has no CPU path */
  KNH_ERR_WRONG_VALUE_KIND = 7,  /* e.g. Trigger sent to a Float parameter
                                    (reference panics: knaster_macros/src/lib.rs:601-606) */
  KNH_ERR_OUT_OF_MEMORY = 8,     /* host memory ran out inside the library (std::bad_alloc caught at the boundary) */
  KNH_ERR_INTERNAL = 9           /* any other C++ exception caught at the boundary: never unwinds into the caller */
} knh_status;

/* Sample type F of the bank: knaster_primitives/src/float.rs:97-175 */
typedef enum knh_sample_type { KNH_F32 = 0, KNH_F64 = 1 } knh_sample_type;

/* ParameterValue kinds: knaster_core/src/parameters/types.rs:25-36 */
typedef enum knh_value_kind {
  KNH_VALUE_FLOAT = 0,
  KNH_VALUE_TRIGGER = 1,
  KNH_VALUE_INTEGER = 2,
  KNH_VALUE_BOOL = 3,
  /* ParameterValue::Smoothing(ParameterSmoothing, Rate): fvalue = seconds of ParameterSmoothing::Linear
   * (as f32), ivalue = 0 None, 1 Linear at Rate::BlockRate, 2 Linear at Rate::AudioRate */
  KNH_VALUE_SMOOTHING = 4
} knh_value_kind;

/* A voice chain is a short list of stages evaluated in order on one running
 * signal `x` (one sample per frame).  Each stage stands for the reference
 * nodes/wrappers listed beside it; parameter indices are the reference's own
 * (#[param] declaration order, knaster_macros/src/lib.rs:594-613).
 *
 * kind                      reference construct                                   nodes  ctor args
 * KNH_STAGE_SIN_WT          g.push(SinWt::new(freq))            osc.rs:97-168        1    freq
 *     params: 0 freq, 1 phase_offset, 2 reset_phase(trigger)
 *     with KNH_STAGE_FLAG_AR_FREQ: pushed as SinWt::new(freq).ar_params() and
 *     `.link("freq", x)` -- the running signal drives param 0 every sample
 *     (audio_rate.rs:42-57, graph_edit.rs:735-754); x is replaced by its output.
 * KNH_STAGE_SIN_NUMERIC     g.push(SinNumeric::new(freq))       osc.rs:222-271       1    freq
 *     params: 0 freq, 1 phase_offset, 2 reset_phase(trigger)
 * KNH_STAGE_SVF             x >> SvfFilter::new(ty,cutoff,q,gain_db)  svf.rs:44-281  1    type, cutoff, q, gain_db
 *     params: 0 cutoff_freq, 1 q, 2 gain, 3 filter(integer), 4 t_calculate_coefficients(trigger)
 * KNH_STAGE_ONEPOLE_LPF     x >> OnePoleLpf::new(cutoff)        onepole.rs:111-140   1    cutoff
 * KNH_STAGE_ONEPOLE_HPF     x >> OnePoleHpf::new()              onepole.rs:144-177   1    (none)
 *     params: 0 cutoff_freq
 * KNH_STAGE_MUL_ENV_ASR     x * g.push(EnvAsr::new(a, r))       envelopes.rs:19-163  2    attack_s, release_s
 *     (EnvAsr node + MathUGen<_,U1,Mul>, math.rs:39-49,94-165)
 *     params: 0 attack_time, 1 release_time, 2 t_release(trigger), 3 t_restart(trigger)
 * KNH_STAGE_MUL_ENV_AR      x * g.push(EnvAr::new(a, r))        envelopes.rs:174-303 2    attack_s, release_s
 *     params: 0 attack_time, 1 release_time, 2 t_restart(trigger)
 * KNH_STAGE_MUL_ENVELOPE    x * g.push(Envelope::new(start, segments).time_scale(ts).looping(l))
 *                           envelopes.rs:359-527                                    2    start_value, time_scale,
 *     looping (0/1), n_segments, then n_max x (duration_s, value); n_max = (n_args - 4) / 2 is the same for
 *     every voice of the bank, n_segments <= n_max per voice.  All of its state is f64 for any F, as in the
 *     reference.  At most one per chain.
 *     params: 0 time_scale, 1 jump_to_segment(integer), 2 t_restart(trigger), 3 t_stop(trigger)
 * KNH_STAGE_MUL_CONST       x * c   (Constant + MathUGen Mul)   graph_edit.rs:1036-1066, util.rs:37-64   2   c
 * KNH_STAGE_ADD_CONST       x + c   (Constant + MathUGen Add)                        2    c
 * KNH_STAGE_SUB_CONST       x - c   (Constant + MathUGen Sub)                        2    c
 * KNH_STAGE_DIV_CONST       x / c   (Constant + MathUGen Div)                        2    c
 *     params: 0 value (the Constant's)
 * KNH_STAGE_WR_MUL          previous_node.wr_mul(v)             wrappers_core/math.rs:15-113   0   v
 * KNH_STAGE_WR_ADD          previous_node.wr_add(v)             wrappers_core/math.rs:116-191  0   v
 * KNH_STAGE_WR_SUB          previous_node.wr_sub(v)             wrappers_core/math.rs:194-269  0   v
 * KNH_STAGE_WR_VSUB         previous_node.wr_v_sub_gen(v)  v - x   wrappers_core/math.rs:272-348  0   v
 * KNH_STAGE_WR_DIV          previous_node.wr_div(v)        x / v   wrappers_core/math.rs:351-426  0   v
 * KNH_STAGE_WR_VDIV         previous_node.wr_v_div_gen(v)  v / x   wrappers_core/math.rs:429-505  0   v
 * KNH_STAGE_WR_POWF         previous_node.wr_powf(v)       x.powf(v)  wrappers_core/math.rs:508-584  0   v
 * KNH_STAGE_WR_POWI         previous_node.wr_powi(n)       x.powi(n)  wrappers_core/math.rs:587-661  0   n (i32)
 * KNH_STAGE_POW_CONST       x.powf(c) (Constant + MathUGen Pow, math.rs:75-85)      2    c
 * KNH_STAGE_SAMPLE_DELAY    x >> g.push(SampleDelay::new(Seconds::from_secs_f64(max_delay)))   delay.rs:14-50   1   max_delay (s)
 *     params: 0 delay_time (seconds; delay_samples = (seconds * sample_rate) as usize).  The ring holds
 *     (Seconds::to_secs_f64() * sample_rate) as usize samples per voice, in HBM.  A delay_time longer than the ring
 *     is refused with KNH_ERR_OUT_OF_RANGE (the reference indexes out of bounds there); a zero-length ring fails init.
 *     At most one per chain.
 * KNH_STAGE_ALLPASS_DELAY   x >> g.push(AllpassDelay::new(Seconds::from_secs_f64(max_delay)))  delay.rs:93-206  1  max_delay (s)
 *     params: 0 delay_time (seconds).  The ring holds Seconds::to_samples(sample_rate) samples per voice (HBM); the
 *     fractional part of the delay goes through the reference's first-order allpass interpolator.  A delay_time of
 *     the ring length or more is ignored, as in the reference.  At most one delay stage (of either kind) per chain.
 * KNH_STAGE_ALLPASS_FB_DELAY  x >> g.push(AllpassFeedbackDelay::new(Seconds::from_secs_f64(max_delay)))  delay.rs:210-306  1  max_delay (s)
 *     the Schroeder allpass around an AllpassDelay.  params: 0 delay_time (seconds; longer than the ring: ignored -- the
 *     reference does not check and would index out of bounds), 1 feedback
 * KNH_STAGE_BUFFER_READER   g.push(BufferReader::<F, U1>::new(buffer, rate, looping).start_at(start))  buffer.rs:19-191  1  rate, looping (0/1), start (s)
 *     a source: plays the single-channel Buffer given to knh_bank_set_buffer (shared by every voice) with linear
 *     interpolation; one-shot voices mark done at the frame after their last one and are silent afterwards.
 *     params: 0 rate, 1 looping(bool), 2 start_s, 3 duration_s, 4 end_s, 5 t_restart(trigger).  Not combinable with
 *     delayed_changes_per_block (the reference's block loop ignores partial blocks).  At most one per chain.
 * KNH_STAGE_PHASOR          g.push(Phasor::new(freq))           osc.rs:172-214       1    freq
 *     a source like SinWt: a 0..1 ramp, f64 phase and step whatever F is.  params: 0 freq
 * KNH_STAGE_POLYBLEP        g.push(PolyBlep::new(waveform, freq))   polyblep.rs:123-508   1    waveform (0..13), freq
 *     a source: band-limited saw, sine, cosine, triangle, square, rectangle, ramp, modified triangle / square, half- and
 *     full-wave rectified sine, triangular pulse, fixed and variable trapezoid (the reference's Waveform order; any other
 *     value is Sawtooth).  params: 0 freq, 1 pulse_width, 2 waveform(integer).  Bit-exact except where the reference
 *     calls sin/cos (waveforms 1, 2, 9, 10, and every waveform at or above sample_rate / 4): device libm there.
 * KNH_STAGE_WHITE_NOISE     g.push(WhiteNoise::new())   noise.rs:26-47     1    seed
 * KNH_STAGE_PINK_NOISE      g.push(PinkNoise::new())    noise.rs:49-111    1    seed
 * KNH_STAGE_BROWN_NOISE     g.push(BrownNoise::new())   noise.rs:119-156   1    seed
 * KNH_STAGE_RANDOM_LIN      g.push(RandomLin::new(freq))  noise.rs:158-230   1    seed, freq
 *     params: 0 freq.  Random values in 0..1 joined by straight lines, a new value freq times a second.
 *     (The reference seeds this one with next_randomness_seed() * 94 + 53: the shim passes the counter value.)
 *     sources; no parameters.  seed = the value the UGen's constructor got from next_randomness_seed()
 *     (noise.rs:11-22: a process-wide counter, 0, 1, 2, ... in construction order).  The generator is the
 *     `fastrand` crate's (2.3.0, not vendored with the reference): restated from its published algorithm, parity
 *     unpinned.
 * KNH_STAGE_SAFETY_LIMITER  x >> g.push(SafetyLimiter::new())   dynamics.rs:9-31     1    (none)
 *     clamp to [-1, 1], NaN -> 0; no parameters, no state
 * KNH_STAGE_MATH_ADD/_SUB/_MUL/_DIV/_POW    a + b, a - b, a * b, a / b, a.pow(b) of TWO SIGNALS of the voice
 *                           (MathUGen<_, U1, Op>, math.rs:17-165, as graph_edit.rs:936-971 creates it)   1   (none)
 *     a = the output of stage `input`, b = the output of stage `input2` (both required).  No parameters, no state.  Pow
 *     runs the device libm (tolerance only).  With these, with `input` on any other stage, and with source stages allowed
 *     anywhere in the list (each starts a new signal), a voice is a small feed-forward graph rather than a chain: the
 *     reference's "256 FM cascade" shape (knaster_benchmarks/benches/graph_dsp_performance.rs:37-72) is one.  Such voices
 *     run in the single-wave kernel form, fused at knh_bank_init time: every stage unrolls into the one kernel, so a fused
 *     voice may hold at most 512 stages (KNH_ERR_UNSUPPORTED_CHAIN beyond; 91 stages fuse in 2 s, 379 in a minute).
 *     A voice made of SIN_WT sources, the *_CONST / WR_* arithmetic and MATH_ADD/_SUB/_MUL/_DIV only (nothing wrapped in
 *     WrPreciseTiming or WrSmoothParams) is parallel in time -- every such stage is a pure function of the frame index --
 *     and runs a lane per FRAME instead of a lane per voice, up to 4 096 stages, with the same results (the 256-oscillator
 *     cascade itself: 1 531 stages in ONE voice, 11 us per 128-frame block).
 * KNH_STAGE_INPUT           (graph input) >> ...   the bank NODE's input channel `channel`: UGen::Inputs > 0,
 *                           `input.read(channel, frame)` in process_block (ugen.rs:263-284)               0    channel
 *     a source whose signal is the same for every voice: whatever the host graph connected to that input of the bank
 *     (another node's output feeding every voice's filter; or, with KNH_STAGE_FLAG_AR_FREQ on a following SIN_WT behind
 *     e.g. a MUL_CONST, the buffer UGen::set_ar_param_buffer hands over, ugen.rs:309-329).  knh_bank_desc.in_channels says
 *     how many there are; the samples of the next launch come from knh_bank_set_input[_device].  No parameters.
 * KNH_STAGE_PAN2            x >> g.push(Pan2::new(pan))         pan.rs:12-37         1    pan (-1 .. 1)
 *     mono -> stereo with the cos/sin pan law: the voice's signal times left_gain goes to graph out 0, times
 *     right_gain to graph out 1 (`(voice >> pan).to_graph_out()`, knaster/examples/many_sines.rs:51-63).  Must be the
 *     LAST stage, and the bank must have out_channels = 2; per-voice output (knh_bank_process_block_voices) is then
 *     [2][n_voices][block_size].  params: 0 pan.  The gains are fastapprox::fast::cos / sin (crate fastapprox 0.3.1,
 *     not vendored with the reference) of (pan * 0.5 + 0.5) * pi/2, restated from the published algorithm: parity
 *     unpinned.  Cannot be wrapped in WrPreciseTiming here (delayed_changes_per_block must be 0).
 *     WrMul: params: 0 = the reference's "wr_mul" (index T::Parameters of the
 *     wrapped node, math.rs:69-98).  No other wrapper adds a parameter.  POW_CONST: params: 0 value.
 *     powi is the multiply-by-squaring loop of compiler-builtins (exact, bit-identical to the oracle);
 *     powf runs the device libm: within a few ulp of the host's powf, never bit-exact by contract.
 */
typedef enum knh_stage_kind {
  KNH_STAGE_SIN_WT = 0,
  KNH_STAGE_SIN_NUMERIC = 1,
  KNH_STAGE_SVF = 2,
  KNH_STAGE_ONEPOLE_LPF = 3,
  KNH_STAGE_ONEPOLE_HPF = 4,
  KNH_STAGE_MUL_ENV_ASR = 5,
  KNH_STAGE_MUL_ENV_AR = 6,
  KNH_STAGE_MUL_CONST = 7,
  KNH_STAGE_ADD_CONST = 8,
  KNH_STAGE_SUB_CONST = 9,
  KNH_STAGE_DIV_CONST = 10,
  KNH_STAGE_WR_MUL = 11,
  KNH_STAGE_WR_ADD = 12,
  KNH_STAGE_WR_SUB = 13,
  KNH_STAGE_MUL_ENVELOPE = 14,
  KNH_STAGE_WR_VSUB = 15,
  KNH_STAGE_WR_DIV = 16,
  KNH_STAGE_WR_VDIV = 17,
  KNH_STAGE_WR_POWF = 18,
  KNH_STAGE_WR_POWI = 19,
  KNH_STAGE_POW_CONST = 20,
  KNH_STAGE_SAMPLE_DELAY = 21,
  KNH_STAGE_PHASOR = 22,
  KNH_STAGE_SAFETY_LIMITER = 23,
  KNH_STAGE_POLYBLEP = 24,
  KNH_STAGE_ALLPASS_DELAY = 25,
  KNH_STAGE_ALLPASS_FB_DELAY = 26,
  KNH_STAGE_BUFFER_READER = 27,
  KNH_STAGE_WHITE_NOISE = 28,
  KNH_STAGE_PINK_NOISE = 29,
  KNH_STAGE_BROWN_NOISE = 30,
  KNH_STAGE_RANDOM_LIN = 31,
  KNH_STAGE_PAN2 = 32,
  KNH_STAGE_MATH_ADD = 33,
  KNH_STAGE_MATH_SUB = 34,
  KNH_STAGE_MATH_MUL = 35,
  KNH_STAGE_MATH_DIV = 36,
  KNH_STAGE_MATH_POW = 37,
  KNH_STAGE_INPUT = 38,
  KNH_STAGE_KIND_COUNT = 39
} knh_stage_kind;

/* SvfFilterType: knaster_core_dsp/src/ugens/svf.rs:19-39 (out-of-range -> Low,
 * knaster_macros/src/lib.rs:44-47) */
typedef enum knh_svf_type {
  KNH_SVF_LOW = 0, KNH_SVF_HIGH = 1, KNH_SVF_BAND = 2, KNH_SVF_NOTCH = 3, KNH_SVF_PEAK = 4,
  KNH_SVF_ALL = 5, KNH_SVF_BELL = 6, KNH_SVF_LOW_SHELF = 7, KNH_SVF_HIGH_SHELF = 8
} knh_svf_type;

enum {
  /* SinWt pushed as .ar_params() with its "freq" linked to the running signal */
  KNH_STAGE_FLAG_AR_FREQ = 1u << 0,
  /* the stage's parameterised node is wrapped in WrSmoothParams (innermost wrapper): a
   * KNH_VALUE_SMOOTHING value selects linear smoothing for one Float parameter, after which new
   * Float values are reached by a block-rate linear ramp (smooth_params.rs:12-311).  The ramp lives
   * on the host, as it does on the reference's audio thread: one param_apply per node per block. */
  KNH_STAGE_FLAG_SMOOTH_PARAMS = 1u << 1
};

typedef struct knh_stage_desc {
  uint16_t kind;  /* knh_stage_kind */
  uint16_t flags; /* KNH_STAGE_FLAG_* */
  /* > 0: the stage's parameterised node is wrapped (outermost) in
   * WrPreciseTiming<N, _> with N = this value, so set_delay_within_block
   * takes effect (precise_timing.rs:14-149).  0: delays are ignored with a
   * warning, exactly like an unwrapped reference UGen (ugen.rs:339-341). */
  uint16_t delayed_changes_per_block;
  /* Audio-rate parameter (WrArParams, knaster_core_dsp/src/wrappers_core/audio_rate.rs:11-85; graph_edit.rs:735-754 `link`):
   * 0 = none; k > 0: the stage's parameterised node is pushed as `.ar_params()` and its FLOAT parameter k - 1 is linked to
   * the signal `input2` names (required then): every sample the node's setter for that parameter runs with the driving
   * signal's sample, then the node renders one sample (the wrapper forces frame-by-frame processing, as in the reference);
   * an ordinary param_apply to that parameter is ignored while the link stands (audio_rate.rs:70-74).  Supported:
   *     SIN_WT 0 freq, 1 phase_offset      SIN_NUMERIC 0 freq, 1 phase_offset      *_CONST / POW_CONST 0 value
   *     WR_MUL 0 "wr_mul"                  MUL_ENV_ASR / MUL_ENV_AR 0 attack_time, 1 release_time      -- all bit-exact --
   *     SVF 0 cutoff_freq, 1 q, 2 gain     ONEPOLE_LPF / _HPF 0 cutoff_freq      -- the setter's tan / pow / sqrt / exp run in
   *     the device library: within a tolerance of the reference, not bit for bit.
   * A voice with such a stage is a graph (explicit operands): it runs in the single-wave kernel form, fused at init.
   * (KNH_STAGE_FLAG_AR_FREQ is the older spelling of "SIN_WT, ar_param = 1, driven by the running signal".)
   * An envelope under the wrapper marks done at frame 0, as the reference's does (envelopes.rs:153-156). */
  uint16_t ar_param;
  /* Which signal the stage reads.  0 (what a zero-initialised descriptor says): the output of the stage before it -- a
   * chain on one running signal.  k > 0: the output of stage k - 1 of this list (an earlier one), so that a signal can feed
   * several stages and stages need not follow their input directly.  Source stages read nothing (except SIN_WT with
   * KNH_STAGE_FLAG_AR_FREQ, whose frequency this signal drives); wrapper stages (KNH_STAGE_WR_*) wrap the stage before
   * them and keep 0.  `input2` is the second operand of the KNH_STAGE_MATH_* stages and 0 everywhere else.
   * (A voice with several envelope stages reports, as its done frame, the mark of the last of them in the reference's
   * TASK order that finished in the block -- one UGenFlags for all tasks of a graph, graph_gen.rs:196-200.  For a chain
   * that is list order; for a graph it is the order Graph::calculate_node_order gives: depth first from the voice's output,
   * first operand first.)
   * "The output of stage k - 1" is the output of the NODE that stage stands for: if wrapper stages follow it, what the
   * reader gets is the last wrapper's output (the reference's wr_mul() etc. are part of the UGen they wrap). */
  uint16_t input;
  uint16_t input2;
} knh_stage_desc;

/* How the N per-voice signals are folded into the output block. */
typedef enum knh_mix_mode {
  /* Pairwise sum in voice order, a binary tree over the voice index: level by level
   * neighbours are added in pairs (v0+v1, v2+v3, ...), a node without a right
   * neighbour passes through unchanged.  The order of the additions depends on the
   * number of voices only (not on the kernel form, the grouping of voices into
   * wavefronts or the device): same result on every run; differs from the reference's
   * single left fold only by reassociation (bounded; see DESIGN.md "mixdown"). */
  KNH_MIX_TREE = 0,
  /* Bit-exact reference order ((v0+v1)+v2)+... in sample precision
   * (knaster_graph/src/graph.rs:827-872).  Serial in the voice axis: slower. */
  KNH_MIX_LEFT_FOLD = 1
} knh_mix_mode;

typedef struct knh_bank_desc {
  uint32_t abi_version;  /* KNH_ABI_VERSION */
  uint32_t n_voices;
  uint32_t sample_type;  /* knh_sample_type */
  uint32_t n_stages;
  const knh_stage_desc* stages;
  /* 1: mono signal -> graph out 0.  2: `.out([0,0]).to_graph_out()`, the same
   * mono signal additively to out 0 and out 1 (graph_edit.rs:280-292,363-369) -- or, for a
   * chain that ends in KNH_STAGE_PAN2, the Pan2's left and right outputs to out 0 and out 1. */
  uint32_t out_channels;
  uint32_t mix_mode;     /* knh_mix_mode */
  int32_t device;        /* HIP device ordinal, -1 = current device */
  /* 0 = exact: every a*b+c is two roundings, per-voice output bit-identical to
   * the reference's scalar code.  1 = allow fused multiply-add (faster, last-bit
   * differences). */
  uint32_t allow_fma;
  /* UGen::Inputs of the bank node (0 .. 16): input channels every voice can read through KNH_STAGE_INPUT stages. */
  uint32_t in_channels;
} knh_bank_desc;

typedef struct knh_bank knh_bank;

/* Flags returned by process (summary of UGenFlags, knaster_core/src/ugen.rs:121-219) */
enum {
  KNH_FLAG_ANY_DONE = 1u << 0, /* some voice's envelope called mark_done this block */
  KNH_FLAG_ALL_DONE = 1u << 1  /* every voice's last envelope is Stopped            */
};

/* Library / device discovery. */
uint32_t knh_abi_version(void);
/* Number of usable gfx950 devices (0 if none).  Never fails. */
int32_t knh_device_count(void);
/* Static message for a status code. */
const char* knh_status_string(int32_t status);
/* Message of the last failure on `bank` (or of the last failed create when
 * bank == NULL).  Valid until the next call on that handle. */
const char* knh_last_error(const knh_bank* bank);

/* Number of reference UGen nodes one voice of this chain stands for (the
 * "UGens" factor of the UGen-samples/s metric). */
int32_t knh_chain_ugen_count(const knh_stage_desc* stages, uint32_t n_stages);

/* Construction = the reference's `SomeUGen::new(args)` for every node of every
 * voice (osc.rs:110, svf.rs:64, envelopes.rs:33,187, util.rs:43).  */
int32_t knh_bank_create(const knh_bank_desc* desc, knh_bank** out_bank);
/* The same bank with its HOST work on `host_threads` threads.  WrPreciseTiming's change queues
 * (precise_timing.rs:65-135) and the per-block assembly of the device event lists are per node, hence per
 * voice: the bank is cut into up to `host_threads` contiguous ranges of whole 64-voice groups, a worker thread
 * per range does that range's share of knh_bank_param_apply_many[_at] and of process, the ranges' kernels run
 * side by side on their own streams, and their mixes are summed in range order on the caller's stream.  Every
 * call keeps its meaning and stays single-caller.  Meant for banks whose voices all receive sample-accurate
 * changes (BASELINE config C5), where one core assembling the lists is slower than the kernel.
 * Measured on C5: 2.4e10 -> 4.8e10 UGen-samples/s with 4 threads; more ranges than the runtime has hardware queues
 * (4) run their kernels one after another and lose (8: 2.5e10).
 * host_threads 0 or 1, KNH_MIX_LEFT_FOLD banks and banks of at most 64 voices: identical to knh_bank_create.
 * The mix is a sum of per-range tree mixes: deterministic, within the tree mix's tolerance of the left fold. */
int32_t knh_bank_create_sharded(const knh_bank_desc* desc, uint32_t host_threads, knh_bank** out_bank);
/* Constructor arguments of stage `stage` for `count` voices starting at
 * `first_voice`; `args` is [count][n_args] row-major, n_args as in the table
 * above.  Must be called before knh_bank_init; unset stages use zeros. */
int32_t knh_bank_set_ctor_args(knh_bank* bank, uint32_t stage, uint32_t first_voice, uint32_t count,
                               const double* args, uint32_t n_args);
/* UGen::init(sample_rate, block_size) -- knaster_core/src/ugen.rs:242-246; runs
 * on the control thread, may allocate (graph.rs:462-475).  Uploads all state. */
/* Buffer::from_vec(samples, sample_rate) (dsp/buffer.rs:58-66) for the chain's BufferReader stage: `n_frames` samples of
 * the bank's sample type, copied into device memory at init.  Before knh_bank_init. */
int32_t knh_bank_set_buffer(knh_bank* bank, uint32_t stage, const void* samples, size_t n_frames, double buffer_sample_rate);
int32_t knh_bank_init(knh_bank* bank, uint32_t sample_rate, size_t block_size);
void knh_bank_destroy(knh_bank* bank);

/* UGen::inputs()/outputs()/parameters() of the bank node -- dynugen.rs:23-63 */
uint16_t knh_bank_inputs(const knh_bank* bank);
uint16_t knh_bank_outputs(const knh_bank* bank);
/* Number of parameters of one stage (0 for unknown stage). */
uint16_t knh_bank_stage_parameters(const knh_bank* bank, uint32_t stage);
/* UGen::param_descriptions() of one stage; NULL when out of range. */
const char* knh_bank_stage_param_description(const knh_bank* bank, uint32_t stage, uint32_t param);

/* UGen::param_apply(ctx, index, value) -- knaster_core/src/ugen.rs:308 -- on
 * the node of `stage` in `voice`.  Takes effect at the start of the next
 * processed block, or at the armed in-block delay if the stage is wrapped in
 * WrPreciseTiming and a delay is armed for that parameter
 * (precise_timing.rs:126-135).  `fvalue` is used for FLOAT, `ivalue` for
 * INTEGER/BOOL; TRIGGER uses neither. */
int32_t knh_bank_param_apply(knh_bank* bank, uint32_t voice, uint32_t stage, uint32_t param,
                             uint32_t kind, double fvalue, int64_t ivalue);
/* UGen::set_delay_within_block_for_param(ctx, index, delay) -- ugen.rs:330-341,
 * precise_timing.rs:146-148.  The armed delay stays armed for later changes of
 * that parameter until re-armed (reference behaviour). */
int32_t knh_bank_set_delay_within_block_for_param(knh_bank* bank, uint32_t voice, uint32_t stage,
                                                  uint32_t param, uint16_t delay);
/* The same two calls for many (voice, stage, param) at once, applied in array
 * order: one SchedulingEvent each, as GraphGen::apply_parameter_change does
 * (graph_gen.rs:269-305): if delays[i] > 0 the delay is armed first, then the
 * value is applied.  `delays` may be NULL (all zero). */
int32_t knh_bank_param_apply_many(knh_bank* bank, size_t count, const uint32_t* voices,
                                  const uint32_t* stages, const uint32_t* params,
                                  const uint32_t* kinds, const double* fvalues,
                                  const int64_t* ivalues, const uint16_t* delays);
/* knh_bank_param_apply(voice, stage, param, kind, fvalue, ivalue) for every voice of
 * [voice_begin, voice_end) in rising order -- a bank's note-on, "this cutoff for
 * all of them" -- without an array per call: the SchedulingEvents a host would
 * send one by one (graph_gen.rs:269-305), same checks, same result.  An envelope
 * trigger sent this way (or through knh_bank_param_apply_many with the voices in
 * rising order) reaches a resident kernel as one 32-byte range event. */
int32_t knh_bank_param_apply_range(knh_bank* bank, uint32_t voice_begin, uint32_t voice_end, uint32_t stage,
                                   uint32_t param, uint32_t kind, double fvalue, int64_t ivalue);

/* UGen::process_block(ctx, flags, input, output) -- knaster_core/src/ugen.rs:263-284
 * as called by Task::run (knaster_graph/src/task.rs:25-31).
 *   frames_to_process / block_start_offset / frame_clock = ctx.block (ugen.rs:57-112);
 *   a top-level graph always passes (block_size, 0, clock).
 *   out: host pointer, channel-major [out_channels][block_size] of F
 *        (RawContiguousBlock, knaster_graph/src/block.rs:19-78); frames
 *        [block_start_offset, block_start_offset+frames_to_process) are written.
 *   out_flags: optional KNH_FLAG_* summary.
 * Blocks until the block is on the host (non-realtime driver only). */
int32_t knh_bank_process_block(knh_bank* bank, size_t frames_to_process, size_t block_start_offset,
                               uint64_t frame_clock, void* out, uint32_t* out_flags);
/* The same call for an output block that is handed over as the reference's `Block` trait hands it over: one slice per
 * channel (knaster_primitives/src/block.rs:33-197, `channel_as_slice_mut(channel)`), with no promise that channel c + 1
 * follows channel c in memory.  `out_channels[c]` points at the FIRST FRAME THIS CALL WRITES of channel c and
 * frames_to_process samples are written there.  Under a splitting wrapper (WrPreciseTiming, precise_timing.rs:98-110) the
 * output is a PartialBlockMut whose slices already start at the partial block's offset (block.rs:307-339) while
 * ctx.block_start_offset() carries the same offset (BlockMetadata::make_partial, ugen.rs:87-93): a shim passes
 * `output.channel_as_slice_mut(c).as_mut_ptr()` per channel and the ctx's offset, and nothing is offset twice.
 * block_start_offset still tells the bank where in its block the frames lie (its frame counter, the in-block delays of
 * queued changes); it is not applied to the pointers. */
int32_t knh_bank_process_block_channels(knh_bank* bank, size_t frames_to_process, size_t block_start_offset,
                                        uint64_t frame_clock, void* const* out_channels, uint32_t* out_flags);
/* How knh_bank_process_block[_channels] -- the call the reference makes once per block -- is served (round 4).  Where the bank's
 * kernel form allows it (the pipelined forms: up to 256 voice groups, KNH_MIX_TREE, no bank inputs) the first such call
 * launches a RESIDENT kernel: it keeps the voices' state in registers and the sine table in LDS between calls, takes each
 * call as one command word the host stores, mixes across its workgroups itself and writes the block into host memory -- no
 * launch, no staging, no second kernel per call.  It leaves when anything else needs the device state or the device (any
 * other entry point of this bank that reads state or launches; another bank launching on the device) or when the host has
 * not called for KNH_RESIDENT_IDLE_US microseconds (default 5 000; every wait on the device is bounded by a clock), and the
 * next call launches it again.  Results are those of a launch per call, bit for bit.  KNH_RESIDENT=0: a launch per call.
 * calls: process calls served by a resident kernel so far; launches: how many times one was launched.  Either may be NULL. */
int32_t knh_bank_resident_stats(knh_bank* bank, uint64_t* calls, uint64_t* launches);
/* Diagnostics of the last call a resident kernel served, on the device's constant 100 MHz clock (ticks of 10 ns): [0] the
 * voice kernel saw the command, [1] the fold server's root did, [2] the first tile's rows had all arrived, [3] the last
 * tile's, [4] the root had written the block and the flags.  Zeros when no resident kernel has served a call. */
int32_t knh_bank_resident_trace(knh_bank* bank, uint64_t* ticks5);
/* Same, but the mixed block is left in device memory at `out_device`
 * ([out_channels][block_size] of F) and the work is only enqueued on
 * `hip_stream` (a hipStream_t, NULL = the bank's own stream).  Used for the
 * multi-GPU reduce and for back-to-back blocks without host round trips. */
int32_t knh_bank_process_block_device(knh_bank* bank, size_t frames_to_process,
                                      size_t block_start_offset, uint64_t frame_clock,
                                      void* out_device, void* hip_stream);
/* The input block(s) of the bank node for the NEXT process call: `in` = host [n_blocks][in_channels][block_size] of F,
 * channel-major per block like the output (what Task::run hands a node as its input channels,
 * knaster_graph/src/task.rs:17-32); copied.  Needed before every process call of a bank with KNH_STAGE_INPUT stages;
 * n_blocks must match that call.  The _device form takes device memory of the same layout, read in stream order. */
int32_t knh_bank_set_input(knh_bank* bank, uint32_t n_blocks, const void* in);
int32_t knh_bank_set_input_device(knh_bank* bank, uint32_t n_blocks, const void* in_device);
/* Parity/debug: also materialise every voice's own signal,
 * voices_out = host [n_voices][block_size] of F ([2][n_voices][block_size] for a chain that ends in
 * KNH_STAGE_PAN2: every voice's left signals, then every voice's right signals).  `out` may be NULL. */
int32_t knh_bank_process_block_voices(knh_bank* bank, size_t frames_to_process,
                                      size_t block_start_offset, uint64_t frame_clock, void* out,
                                      void* voices_out, uint32_t* out_flags);
/* Several consecutive whole blocks in ONE launch (non-realtime rendering): the voices' state stays in
 * registers across the blocks, so launch overhead and the sine-table staging are paid once.  Results
 * are identical, bit for bit, to n_blocks calls of knh_bank_process_block.  This is
 * AudioProcessor::run_without_inputs() called n_blocks times (knaster_graph/src/processor.rs:142-179)
 * with every SchedulingEvent of those blocks known up front; changes for block k > 0 of the launch are
 * queued with knh_bank_param_apply_many_at.
 *   out / out_device: [n_blocks][out_channels][block_size] of F. */
int32_t knh_bank_process_blocks(knh_bank* bank, uint32_t n_blocks, uint64_t frame_clock, void* out,
                                uint32_t* out_flags);
int32_t knh_bank_process_blocks_device(knh_bank* bank, uint32_t n_blocks, uint64_t frame_clock,
                                       void* out_device, void* hip_stream);
/* As knh_bank_process_blocks_device, but the bank's mix is ADDED to what `out_device` already holds
 * (out = out + mix): a further additive source on the same graph outputs, the MathUGen<Add> that
 * connect_to_output_internal inserts (knaster_graph/src/graph.rs:850-864).  Lets a host with several
 * banks (voices of different shapes) keep one output buffer in HBM and read it back once. */
int32_t knh_bank_process_blocks_device_add(knh_bank* bank, uint32_t n_blocks, uint64_t frame_clock,
                                           void* out_device, void* hip_stream);
/* knh_bank_process_blocks in two halves, for a host that renders launch after launch to host memory (offline rendering:
 * AudioProcessor::run_without_inputs in a loop, processor.rs:142-179): _begin enqueues the launch and the copy of its mixed
 * blocks into pinned memory and returns; _end waits for the OLDEST outstanding launch and copies its blocks to `out`
 * ([n_blocks][out_channels][block_size] of F).  Up to two launches may be outstanding, so that the host's work for launch
 * k + 1 (parameter calls, event assembly) and its kernels run while launch k's blocks cross PCIe.  Same samples as
 * knh_bank_process_blocks; no flag summary (knh_bank_read_done_frames, or a blocking call, reports those). */
int32_t knh_bank_process_blocks_begin(knh_bank* bank, uint32_t n_blocks, uint64_t frame_clock);
int32_t knh_bank_process_blocks_end(knh_bank* bank, void* out);
/* Device-memory helpers for hosts that do not link HIP themselves (the library owns the allocations):
 * zero-initialised allocation on `device` (-1: current), free, and a synchronising device-to-host read. */
void* knh_device_malloc(size_t bytes, int32_t device);
void knh_device_free(void* device_ptr);
int32_t knh_device_read(void* dst_host, const void* src_device, size_t bytes, void* hip_stream);
/* knh_bank_param_apply_many addressed to block `block_offset` (0 = the next block) of the next
 * launch: what GraphGen does with a SchedulingEvent whose Time is not yet due -- it keeps it and
 * applies it in the block where to_samples_until_due() < block_size
 * (knaster_graph/src/graph_gen.rs:110-166, scheduling.rs:95-121). */
int32_t knh_bank_param_apply_many_at(knh_bank* bank, uint32_t block_offset, size_t count,
                                     const uint32_t* voices, const uint32_t* stages,
                                     const uint32_t* params, const uint32_t* kinds,
                                     const double* fvalues, const int64_t* ivalues,
                                     const uint16_t* delays);
/* Per-voice done frame of the last processed block (UGenFlags::done,
 * ugen.rs:169-175): done_frames[v] = frame in block, or UINT32_MAX. */
int32_t knh_bank_read_done_frames(knh_bank* bank, uint32_t* done_frames);
/* Diagnostics: 16 device words of the last launch (0,1: the done/running summary; 4..8: per-role busy
 * cycles per tile when the library was built with -DKNH_DAG_STAMPS, zero otherwise). */
int32_t knh_bank_debug_words(knh_bank* bank, uint32_t* out16);
/* Diagnostics: the chain as the device code names it (one character per stage, "@a,b,o" signal slots and "#R" for a
 * voice that is a graph, "%P" for an audio-rate parameter): what a chain without a pre-built kernel is fused from at
 * knh_bank_init.  Valid while the bank lives; available before init (tools/jit_compile_fuzz.py compiles such strings
 * without a GPU). */
const char* knh_bank_debug_signature(const knh_bank* bank);
/* Run-time fusion (chains without a pre-built kernel are fused by hiprtc at knh_bank_init): where this process's kernels have
 * come from so far -- its in-memory table, the code-object cache on disk (KNH_JIT_CACHE_DIR; default $XDG_CACHE_HOME/knaster_hip,
 * $HOME/.cache/knaster_hip or /tmp/knaster_hip-<uid>; KNH_JIT_CACHE=0: none), a compile in the helper process (knh_jit_helper
 * beside the library, or $KNH_JIT_HELPER: a compiler crash there is KNH_ERR_INTERNAL + knh_last_error for the host, never the
 * host's own death), or a compile in this process (no helper found, or KNH_JIT_INPROCESS=1).  Any pointer may be NULL. */
void knh_jit_stats(uint64_t* memory_hits, uint64_t* disk_hits, uint64_t* helper_compiles, uint64_t* in_process_compiles);
/* Wait for everything enqueued by *_device calls. */
int32_t knh_bank_synchronize(knh_bank* bank);

/* Measurement hooks (bench.py): device time in milliseconds and launch count
 * of the voice kernel accumulated since the last reset, measured with HIP
 * events on the stream the kernel runs on. */
int32_t knh_bank_timing_reset(knh_bank* bank, int32_t enable);
int32_t knh_bank_timing_read(knh_bank* bank, double* kernel_ms, uint64_t* launches);
/* The same for the exchange between GPUs of a bank made by knh_bank_create_rank[_custom]: time and count of the sums of the
 * ranks' mixed blocks since the last knh_bank_timing_reset (RCCL: device time of the ncclReduce calls on the communicator's
 * stream; a host reduce function: wall time of its calls).  0 / 0 for any other bank.  A first multi-GPU run explains itself
 * with the two: kernel time per launch and reduce time per launch, per rank (bench.py prints both lists). */
int32_t knh_bank_collective_timing_read(knh_bank* bank, double* reduce_ms, uint64_t* reduces);
/* Algorithmic HBM bytes one voice moves per processed block (state read once,
 * mutable state written once); see DESIGN.md. */
int32_t knh_bank_algorithmic_bytes_per_voice_block(const knh_bank* bank, uint32_t* read_bytes,
                                                   uint32_t* write_bytes);

/* ---------------------------------------------------------------------------------------------------------------
 * Several GPUs of one node.  The voices of a graph are independent -- no voice reads another voice's
 * state or output (knaster/examples/many_sines.rs:51-63); the only cross-voice operation is the additive graph output
 * (knaster_graph/src/graph.rs:827-872) -- so a bank shards by contiguous voice ranges of whole 64-voice groups, the
 * chain and the sine table are replicated, parameter changes are routed on the host by voice index, and the one exchange
 * is the sum of the GPUs' mixed blocks.  Both forms return an ordinary knh_bank: every entry point above keeps its
 * meaning and takes GLOBAL voice indices.  KNH_MIX_TREE only (the sum re-associates, like the tree mix); per-voice output
 * is not available.
 * --------------------------------------------------------------------------------------------------------------- */

/* ONE PROCESS, several GPUs: voice range k lives on devices[k] with a host thread and a stream of its own; each range's
 * mixed blocks are copied peer-to-peer (xGMI) to devices[0] and summed there in range order -- the one-shot direct sum,
 * deterministic, no ring.  The mix (out / out_device) belongs to devices[0].  A device may appear more than once.
 * This is what a Rust host owning all GPUs of the node calls instead of knh_bank_create. */
int32_t knh_bank_create_multi_device(const knh_bank_desc* desc, const int32_t* devices, uint32_t n_devices, knh_bank** out_bank);

/* ONE PROCESS PER GPU: this process is `rank` of `world` and owns the voices knh_shard_voice_range() gives it, on
 * desc->device; desc->n_voices is the TOTAL.  A call for a voice of another rank is checked and otherwise ignored, so
 * every rank may be handed the same parameter stream.  After every launch the ranks' mixed blocks are summed to rank 0
 * by RCCL (ncclReduce over xGMI) on a stream of the communicator's own: the next launch overlaps it when the host
 * alternates two out_device buffers.  The mix is on rank 0 (the other ranks' `out` holds their own share of it), once
 * knh_bank_synchronize (or a blocking process call) has returned.  `comm_id`: KNH_COMM_ID_BYTES made by rank 0 with
 * knh_comm_unique_id and handed to the other ranks by whatever channel the host has.  world == 1 needs no RCCL. */
#define KNH_COMM_ID_BYTES 128
int32_t knh_comm_unique_id(uint8_t* id);
int32_t knh_bank_create_rank(const knh_bank_desc* desc, uint32_t rank, uint32_t world, const uint8_t* comm_id, knh_bank** out_bank);
/* The same with the host's own sum-reduce instead of RCCL (another transport; or tests, which run two ranks on one GPU,
 * where RCCL refuses a second rank): called after every launch as reduce(user, device_buf, count, sample_type, root,
 * hip_stream); on return the sum over all ranks must be in `device_buf` on `root`, in the order of `hip_stream`
 * (the function may block).  Returns a knh_status. */
typedef int32_t (*knh_reduce_fn)(void* user, void* device_buf, size_t count, uint32_t sample_type, uint32_t root, void* hip_stream);
int32_t knh_bank_create_rank_custom(const knh_bank_desc* desc, uint32_t rank, uint32_t world, knh_reduce_fn reduce, void* user,
                                    knh_bank** out_bank);
/* Voice range [first, first + count) of `rank`: the 64-voice groups dealt out as evenly as they go.  Never fails for
 * rank < world (count may be 0); returns KNH_ERR_INVALID_ARGUMENT otherwise. */
int32_t knh_shard_voice_range(uint32_t n_voices, uint32_t rank, uint32_t world, uint32_t* first, uint32_t* count);
/* Number of ranks RCCL reports for a bank made by knh_bank_create_rank (ncclCommCount; 1 when world == 1; the number of
 * voice ranges for a multi-device or host-sharded bank; 1 for any other bank). */
uint32_t knh_bank_ranks(const knh_bank* bank);

/* The communicator on its own, for hosts that keep their banks separate: sum-reduce `count` samples at `buf` (device
 * memory, in place on `root`) over all ranks, enqueued on the communicator's stream after everything `after_stream`
 * holds so far.  knh_comm_wait_buffer / knh_comm_wait make `stream` wait for the last reduce of `buf` / for all of them;
 * knh_comm_synchronize waits on the host. */
typedef struct knh_comm knh_comm;
int32_t knh_comm_create(uint32_t rank, uint32_t world, const uint8_t* id, int32_t device, knh_comm** out_comm);
void knh_comm_destroy(knh_comm* comm);
const char* knh_comm_last_error(const knh_comm* comm);
uint32_t knh_comm_world(const knh_comm* comm);
int32_t knh_comm_rccl_version(void);
int32_t knh_comm_reduce_sum(knh_comm* comm, void* buf, size_t count, uint32_t sample_type, uint32_t root, void* after_stream);
int32_t knh_comm_wait_buffer(knh_comm* comm, const void* buf, void* stream);
int32_t knh_comm_wait(knh_comm* comm, void* stream);
int32_t knh_comm_synchronize(knh_comm* comm);
/* Measurement hooks: device time in milliseconds and count of the reduces since the last reset (HIP events on the
 * communicator's stream). */
int32_t knh_comm_timing_reset(knh_comm* comm, int32_t enable);
int32_t knh_comm_timing_read(knh_comm* comm, double* reduce_ms, uint64_t* reduces);

#ifdef __cplusplus
}
#endif
#endif /* KNASTER_HIP_H */
