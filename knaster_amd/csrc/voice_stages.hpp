// voice_stages.hpp -- device-side UGen stages (gfx950): what one stage of a voice does to one sample and to a tile of samples.
// (The chain that strings them together and the kernels are in voice_chain.hpp.)
//
// One lane = one voice.  A chain is a compile-time list of stages evaluated in
// order on one running sample x; all per-voice state lives in registers for the
// whole block and is read from / written back to a struct-of-arrays in HBM once
// per launch (coalesced: lane i touches word i of each slot row).
//
// Arithmetic contract: with FMA == false every a*b+c below is a separate
// multiply and add in source order (the translation unit is also built with
// -ffp-contract=off), which makes each voice's signal bit-identical to the
// reference's scalar Rust.  Citations are file:line in the knaster repo.
//
// This header is self-contained (no libc/libstdc++ includes) so the same text
// can be handed to hiprtc for chains that are not pre-instantiated.
#pragma once

// Accurate (not v_sin_f32) sine from the ROCm device library, linked by hipcc and hiprtc alike.
extern "C" __device__ float __ocml_sin_f32(float);
extern "C" __device__ double __ocml_sin_f64(double);
extern "C" __device__ float __ocml_cos_f32(float);
extern "C" __device__ double __ocml_cos_f64(double);
extern "C" __device__ float __ocml_pow_f32(float, float);
extern "C" __device__ double __ocml_pow_f64(double, double);
extern "C" __device__ float __ocml_tan_f32(float);
extern "C" __device__ double __ocml_tan_f64(double);
extern "C" __device__ float __ocml_exp_f32(float);
extern "C" __device__ double __ocml_exp_f64(double);
extern "C" __device__ float __ocml_sqrt_f32(float);
extern "C" __device__ double __ocml_sqrt_f64(double);

namespace knh_dev {

typedef unsigned int u32;
typedef unsigned long long u64;

// Slot word: u32 for an f32 bank, u64 for an f64 bank.
template <typename F> struct WordOf;
template <> struct WordOf<float> { typedef u32 type; };
template <> struct WordOf<double> { typedef u64 type; };

template <typename F> __device__ __forceinline__ F word_to_f(typename WordOf<F>::type w);
template <> __device__ __forceinline__ float word_to_f<float>(u32 w) { return __builtin_bit_cast(float, w); }
template <> __device__ __forceinline__ double word_to_f<double>(u64 w) { return __builtin_bit_cast(double, w); }
__device__ __forceinline__ u32 f_to_word(float f) { return __builtin_bit_cast(u32, f); }
__device__ __forceinline__ u64 f_to_word(double f) { return __builtin_bit_cast(u64, f); }

template <bool FMA> __device__ __forceinline__ float mad(float a, float b, float c) {
  if constexpr (FMA) return __builtin_fmaf(a, b, c);
  else return a * b + c;
}
template <bool FMA> __device__ __forceinline__ double mad(double a, double b, double c) {
  if constexpr (FMA) return __builtin_fma(a, b, c);
  else return a * b + c;
}

// Rust `as u32` from f64: NaN -> 0, negative -> 0, too large -> u32::MAX (osc.rs:129).  That is what the hardware
// conversion does by itself (v_cvt_u32_f64: out-of-range values saturate, NaN gives 0); spelled as compares in C++ it
// becomes two branches per sample (an out-of-range cast is undefined there, so the compiler guards the instruction).
__device__ __forceinline__ u32 sat_u32(double v) {
  u32 r;
  asm("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(v));
  return r;
}

// What a stage is unless it says otherwise: none of its slots can be switched in the middle of a tile (see take_params).
struct StageDefaults {
  // Slots (bit k = the stage's slot k) that are PARAMETERS: read by tick, never written by it, so that a sample-accurate
  // change of one is "use the new value from frame f on" and nothing else.  A tile in which voices have such changes
  // runs stage by stage like any other tile, each sample taking over the new parameter values at its voice's frame
  // (take_params, one compare and one select per parameter and sample) instead of dropping to the per-sample path.
  static constexpr u32 kParamMask = 0u;
  template <typename R> static __device__ __forceinline__ void take_params(R&, const R&, bool) {}
  static constexpr bool kBinary = false;  // a MathUGen of two signals (Math2): no tick, an apply(a, b)
  // >= 0: the node is pushed as `.ar_params()` and this float parameter is linked to a second signal of the voice
  // (WrArParams, audio_rate.rs:11-85: every sample `param_apply(p, buf[i])`, then one sample of the node): ArP<S, P> below
  static constexpr int kArParam = -1;
  static constexpr bool kUsesRing = false;  // a delay: the stage moves tiles of a per-voice ring in HBM (RingLines)
};
// A stage whose parameter P is driven at audio rate by another signal of the voice (graph-shaped voices: DagChain reads the
// driver through the stage's second operand and calls S::ar_set<F, P> in front of every sample).  What the setter of each
// parameter does per sample is restated in the stage (ar_set); the parameters that are + - x / only are bit-exact, the
// SvfFilter's and the one-pole filters' cutoff go through the device's tan / pow / sqrt / exp (tolerance).
template <typename S, int P> struct ArP : S {
  static constexpr int kArParam = P;
};

// Event opcodes (host -> device state patches, applied at an in-block frame).
enum { EV_SET = 0, EV_ENV_ASR_RELEASE = 1, EV_NOP = 2, EV_SEGENV_STOP = 3, EV_ALLPASS_DELAY = 4, EV_SPLIT = 0x80 /* flag: change came out of a WrPreciseTiming queue */ };

struct Event {   // 16 bytes
  u32 frame;     // absolute frame within the launch: block_index * block_size + frame_in_block
  u32 slot_op;   // slot (low 24 bits: absolute slot index of the patched word, or the stage's first
                 // slot for ops) | op << 24 (EV_* incl. the EV_SPLIT flag)
  u64 bits;      // new word (low 32 bits for an f32 bank)
};

// Uniform per-launch context.
struct Ctx {
  const float* sine;        // LDS copy of the 16384-entry sine table
  double f2pi;              // SinWt::freq_to_phase_inc (osc.rs:144-145)
  const double* seg_table;  // segment Envelope: [voice][seg_max][3] = (duration, 1/duration, value)
  u32 seg_max;
  void* delay_ring;         // SampleDelay: [voice][delay_stride] samples of F, each voice's ring contiguous
  u32 delay_stride;
  const void* buffer;       // BufferReader: the bank's shared single-channel Buffer, samples of F
  u32 buffer_frames;
  const void* input_block;  // the bank node's input channels for the block being processed: [in_channels][in_stride] of F
  u32 in_stride;            // = block_size
  u32 sample_rate;          // ctx.sample_rate(), for setters that run on the device (audio-rate parameters)
  __attribute__((address_space(3))) char* ring_tile;  // this wavefront's RingLines tile in LDS (whole-chain kernels of up to eight wavefronts), or null
  u32 ring_sink_row;        // the spare ring behind the last voice's (= the number of voices): where lanes without a voice move their lines
};

// ---------------------------------------------------------------------------
// Stages.  Each has: kSlots, kMutableMask (slots written back), Regs<F>,
// load/store, tick (one sample), on_event.
// ---------------------------------------------------------------------------

// SinWt -- knaster_core_dsp/src/ugens/osc.rs:97-168, wavetable.rs:21-60,322-324
// slots: 0 phase, 1 phase_offset, 2 phase_increment
template <bool AR_FREQ>
struct SinWtT : StageDefaults {
  static constexpr int kSlots = 3;
  static constexpr u32 kMutableMask = AR_FREQ ? 0b101u : 0b001u;
  static constexpr bool kUsesSine = true;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = false;
  static constexpr bool kHasSeg = false;
  template <typename F> struct Regs { u32 phase, off, inc; };
  static constexpr u32 kParamMask = AR_FREQ ? 0b010u : 0b110u;  // phase_offset, and the increment unless the signal drives it
  template <typename R> static __device__ __forceinline__ void take_params(R& r, const R& n, bool c) {
    r.off = c ? n.off : r.off;
    if (!AR_FREQ) r.inc = c ? n.inc : r.inc;
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long stride) {
    r.phase = (u32)s[0]; r.off = (u32)s[stride]; r.inc = (u32)s[2 * stride];
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void store(const Regs<F>& r, W* s, long stride) {
    s[0] = (W)r.phase;
    if (AR_FREQ) s[2 * stride] = (W)r.inc;
  }
  // SinWt::freq (osc.rs:127-130) / ::phase_offset (:133-135) applied to one sample of the driving signal
  template <typename F, int P>
  static __device__ __forceinline__ void ar_set(Regs<F>& r, F v, const Ctx& c) {
    if (P == 0) r.inc = sat_u32((double)v * c.f2pi);
    else r.off = sat_u32((double)v * 65536.0);
  }
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F x, const Ctx& c, u32, u32&) {
    if (AR_FREQ) {
      // WrArParams::process (audio_rate.rs:42-57): param_apply(freq, x as f64) then process.
      r.inc = sat_u32((double)x * c.f2pi);
    }
    float s = c.sine[((r.phase + r.off) >> 16) & 16383u];
    r.phase += r.inc;
    return (F)s;
  }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
    if constexpr (AR_FREQ) {
      // the same running sum with the driving signal's increment per sample (audio_rate.rs:42-57: the setter, then the sample)
      u32 q = r.phase + r.off, inc = r.inc;
#pragma unroll
      for (int j = 0; j < T; ++j) {
        inc = sat_u32((double)x[j] * c.f2pi);
        x[j] = (F)c.sine[(q >> 16) & 16383u];
        q += inc;
      }
      r.inc = inc;
      r.phase = q - r.off;
    } else {
      // phase + phase_offset as ONE running sum over the tile (u32 arithmetic wraps, so the phase afterwards is that sum
      // minus the offset): add, shift, mask, read per sample
      u32 q = r.phase + r.off;
#pragma unroll
      for (int j = 0; j < T; ++j) {
        x[j] = (F)c.sine[(q >> 16) & 16383u];
        q += r.inc;
      }
      r.phase = q - r.off;
    }
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits, u32 /*frame*/) {
    if ((op & 0x7Fu) != EV_SET) return;
    if (rel == 0) r.phase = (u32)bits;
    else if (rel == 1) r.off = (u32)bits;
    else r.inc = (u32)bits;
  }
};
typedef SinWtT<false> SinWt;
typedef SinWtT<true> SinWtAr;

// Phasor -- osc.rs:172-214: out = phase; phase += step; while phase >= 1 { phase -= 1 }.  Phase and step are f64 for
// any F.  slots: 0,1 phase (low, high word)  2,3 step
struct Phasor : StageDefaults {
  static constexpr int kSlots = 4;
  static constexpr u32 kMutableMask = 0b0011u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = false;
  static constexpr bool kHasSeg = false;
  template <typename F> struct Regs { double phase, step; };
  template <typename W> static __device__ __forceinline__ double ld2(const W* s, long st, int k) {
    const u64 lo = (u32)s[(long)k * st], hi = (u32)s[(long)(k + 1) * st];
    return __builtin_bit_cast(double, lo | (hi << 32));
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long st) { r.phase = ld2(s, st, 0); r.step = ld2(s, st, 2); }
  template <typename F, typename W>
  static __device__ __forceinline__ void store(const Regs<F>& r, W* s, long st) {
    const u64 b = __builtin_bit_cast(u64, r.phase);
    s[0] = (W)(u32)b;
    s[st] = (W)(u32)(b >> 32);
  }
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F, const Ctx&, u32, u32&) {
    const F out = (F)r.phase;
    r.phase += r.step;
    while (r.phase >= 1.0) r.phase -= 1.0;
    return out;
  }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits, u32) {
    if ((op & 0x7Fu) != EV_SET) return;
    const u32 w = (u32)bits;
    auto lo = [](double d, u32 v) { return __builtin_bit_cast(double, (__builtin_bit_cast(u64, d) & 0xFFFFFFFF00000000ull) | (u64)v); };
    auto hi = [](double d, u32 v) { return __builtin_bit_cast(double, (__builtin_bit_cast(u64, d) & 0x00000000FFFFFFFFull) | ((u64)v << 32)); };
    switch (rel) {
      case 0: r.phase = lo(r.phase, w); break;
      case 1: r.phase = hi(r.phase, w); break;
      case 2: r.step = lo(r.step, w); break;
      default: r.step = hi(r.step, w); break;
    }
  }
};

// WhiteNoise / PinkNoise / BrownNoise -- knaster_core_dsp/src/ugens/noise.rs:26-156.  Sources.  Their random numbers come
// from the `fastrand` crate (Cargo.lock: 2.3.0), which is not vendored in the reference tree: restated here from its
// published algorithm (PARITY UNPINNED): Rng(seed) is a u64; every draw does
//   s += 0x2d358dccaa6c78a5;  t = (u128)s * (s ^ 0x8bb84b93962eacc9);  r = lo64(t) ^ hi64(t)
// (wyrand, final v4.2 constants); u32() takes the low 32 bits; f32() = from_bits(0x3F800000 + (u32() >> 9)) - 1.0.
// The reference seeds each UGen with next_randomness_seed() (a process-wide counter, noise.rs:11-22): the constructor
// argument.  Every sample is `F::new(rng.f32() * 2.0 - 1.0)`: f32 arithmetic, then the cast.
struct NoiseRng {
  u64 s;
  __device__ __forceinline__ float f32() {
    s += 0x2d358dccaa6c78a5ull;
    const u64 b = s ^ 0x8bb84b93962eacc9ull;
    const u64 lo = s * b;
    const u64 hi = __umul64hi(s, b);
    const u32 r = (u32)(lo ^ hi);
    return __builtin_bit_cast(float, 0x3F800000u + (r >> 9)) - 1.0f;
  }
  __device__ __forceinline__ float bipolar() { return f32() * 2.0f - 1.0f; }
  template <typename W> __device__ __forceinline__ void load(const W* st, long stride) {
    s = (u64)(u32)st[0] | ((u64)(u32)st[stride] << 32);
  }
  template <typename W> __device__ __forceinline__ void store(W* st, long stride) const {
    st[0] = (W)(u32)s;
    st[stride] = (W)(u32)(s >> 32);
  }
  __device__ __forceinline__ void patch(u32 rel, u32 w) {
    s = rel == 0 ? ((s & 0xFFFFFFFF00000000ull) | (u64)w) : ((s & 0x00000000FFFFFFFFull) | ((u64)w << 32));
  }
};
// slots: 0,1 rng state (low, high word)
struct WhiteNoise : StageDefaults {
  static constexpr int kSlots = 2;
  static constexpr u32 kMutableMask = 0b11u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = false;
  static constexpr bool kHasSeg = false;
  template <typename F> struct Regs { NoiseRng rng; };
  template <typename F, typename W> static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long st) { r.rng.load(s, st); }
  template <typename F, typename W> static __device__ __forceinline__ void store(const Regs<F>& r, W* s, long st) { r.rng.store(s, st); }
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F, const Ctx&, u32, u32&) { return (F)r.rng.bipolar(); }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits, u32) {
    if ((op & 0x7Fu) == EV_SET) r.rng.patch(rel, (u32)bits);
  }
};
// BrownNoise -- noise.rs:119-156: last += white * 0.1; clamp to [-1, 1].  slots: 0,1 rng  2 last_output
struct BrownNoise : StageDefaults {
  static constexpr int kSlots = 3;
  static constexpr u32 kMutableMask = 0b111u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = false;
  static constexpr bool kHasSeg = false;
  template <typename F> struct Regs { NoiseRng rng; F last; };
  template <typename F, typename W> static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long st) {
    r.rng.load(s, st);
    r.last = word_to_f<F>(s[2 * st]);
  }
  template <typename F, typename W> static __device__ __forceinline__ void store(const Regs<F>& r, W* s, long st) {
    r.rng.store(s, st);
    s[2 * st] = f_to_word(r.last);
  }
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F, const Ctx&, u32, u32&) {
    const F white = (F)r.rng.bipolar();
    F v = r.last + white * (F)0.1;  // F::new(0.1): the f64 literal cast to F
    // f32::clamp / f64::clamp: NaN stays NaN
    v = v < (F)-1 ? (F)-1 : v;
    v = v > (F)1 ? (F)1 : v;
    r.last = v;
    return v;
  }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits, u32) {
    if ((op & 0x7Fu) != EV_SET) return;
    if (rel < 2) r.rng.patch(rel, (u32)bits);
    else r.last = word_to_f<F>((typename WordOf<F>::type)bits);
  }
};
// RandomLin -- noise.rs:158-230: random values in 0..1, a new one whenever the phase reaches 1, straight lines between.
// slots: 0,1 rng  2 current_value  3 current_change_width  4 phase  5 phase_step (= freq / sample_rate, host side)
struct RandomLin : StageDefaults {
  static constexpr int kSlots = 6;
  static constexpr u32 kMutableMask = 0b011111u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = false;
  static constexpr bool kHasSeg = false;
  template <typename F> struct Regs { NoiseRng rng; F value, width, phase, step; };
  template <typename F, typename W> static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long st) {
    r.rng.load(s, st);
    r.value = word_to_f<F>(s[2 * st]); r.width = word_to_f<F>(s[3 * st]);
    r.phase = word_to_f<F>(s[4 * st]); r.step = word_to_f<F>(s[5 * st]);
  }
  template <typename F, typename W> static __device__ __forceinline__ void store(const Regs<F>& r, W* s, long st) {
    r.rng.store(s, st);
    s[2 * st] = f_to_word(r.value); s[3 * st] = f_to_word(r.width); s[4 * st] = f_to_word(r.phase);
  }
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F, const Ctx&, u32, u32&) {
    const F out = mad<FMA>(r.phase, r.width, r.value);  // current_value + phase * current_change_width
    r.phase += r.step;
    if (r.phase >= (F)1) {  // new_value(), noise.rs:186-192
      const F old_target = r.value + r.width;
      const F fresh = (F)r.rng.f32();
      r.value = old_target;
      r.width = fresh - old_target;
      r.phase = (F)0;
    }
    return out;
  }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits, u32) {
    if ((op & 0x7Fu) != EV_SET) return;
    const F v = word_to_f<F>((typename WordOf<F>::type)bits);
    if (rel < 2) r.rng.patch(rel, (u32)bits);
    else if (rel == 2) r.value = v; else if (rel == 3) r.width = v; else if (rel == 4) r.phase = v; else r.step = v;
  }
};
// PinkNoise -- noise.rs:49-111 (Voss-McCartney, nine octaves).  slots: 0,1 rng  2 counter  3 pink  4 always_on
// 5..13 white_noises[0..8].  The nine rows live in registers; the row to replace (counter.trailing_zeros()) is picked
// with selects, not with an indexed access.
struct PinkNoise : StageDefaults {
  static constexpr int kSlots = 14;
  static constexpr u32 kMutableMask = 0x3FFFu;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = false;
  static constexpr bool kHasSeg = false;
  template <typename F> struct Regs { NoiseRng rng; u32 counter; F pink, always_on, white[9]; };
  template <typename F, typename W> static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long st) {
    r.rng.load(s, st);
    r.counter = (u32)s[2 * st];
    r.pink = word_to_f<F>(s[3 * st]);
    r.always_on = word_to_f<F>(s[4 * st]);
#pragma unroll
    for (int k = 0; k < 9; ++k) r.white[k] = word_to_f<F>(s[(long)(5 + k) * st]);
  }
  template <typename F, typename W> static __device__ __forceinline__ void store(const Regs<F>& r, W* s, long st) {
    r.rng.store(s, st);
    s[2 * st] = (W)r.counter;
    s[3 * st] = f_to_word(r.pink);
    s[4 * st] = f_to_word(r.always_on);
#pragma unroll
    for (int k = 0; k < 9; ++k) s[(long)(5 + k) * st] = f_to_word(r.white[k]);
  }
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F, const Ctx&, u32, u32&) {
    const u32 index = (u32)__builtin_ctz(r.counter);  // counter is in 1..=256: index 0..8
    F old = (F)0;
#pragma unroll
    for (int k = 0; k < 9; ++k) old = index == (u32)k ? r.white[k] : old;
    r.pink -= old;
    const F fresh = (F)r.rng.bipolar();
#pragma unroll
    for (int k = 0; k < 9; ++k) r.white[k] = index == (u32)k ? fresh : r.white[k];
    r.pink += fresh;
    r.pink -= r.always_on;
    r.always_on = (F)r.rng.bipolar();
    r.pink += r.always_on;
    r.counter = (r.counter & 255u) + 1u;  // counter &= mask - 1; counter += 1   (mask = 2^8)
    return r.pink / (F)10;                 // / (PINK_NOISE_OCTAVES + 1)
  }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits, u32) {
    if ((op & 0x7Fu) != EV_SET) return;
    const F v = word_to_f<F>((typename WordOf<F>::type)bits);
    if (rel < 2) r.rng.patch(rel, (u32)bits);
    else if (rel == 2) r.counter = (u32)bits;
    else if (rel == 3) r.pink = v;
    else if (rel == 4) r.always_on = v;
    else {
#pragma unroll
      for (int k = 0; k < 9; ++k) r.white[k] = rel == (u32)(5 + k) ? v : r.white[k];
    }
  }
};

// One input channel of the bank NODE (UGen::Inputs > 0, ugen.rs:232-284: `input.read(channel, frame)`; also what an
// audio-rate parameter buffer, ugen.rs:309-329, amounts to): a source whose signal is the same for every voice -- whatever
// the host graph connected to that input.  slot 0: the channel.
struct InputCh : StageDefaults {
  static constexpr int kSlots = 1;
  static constexpr u32 kMutableMask = 0u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = false;
  static constexpr bool kHasSeg = false;
  template <typename F> struct Regs { u32 ch; };
  template <typename F, typename W> static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long) { r.ch = (u32)s[0]; }
  template <typename F, typename W> static __device__ __forceinline__ void store(const Regs<F>&, W*, long) {}
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F, const Ctx& c, u32 frame, u32&) {
    return reinterpret_cast<const F*>(c.input_block)[r.ch * c.in_stride + frame];
  }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
  }
  template <typename F> static __device__ __forceinline__ void on_event(Regs<F>&, u32, u32, u64, u32) {}
};

// SafetyLimiter -- dynamics.rs:9-31: clamp to [-1, 1] (a NaN passes the clamp), then NaN -> 0.  No state.
struct SafetyLimiter : StageDefaults {
  static constexpr int kSlots = 0;
  static constexpr u32 kMutableMask = 0u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = false;
  static constexpr bool kHasSeg = false;
  template <typename F> struct Regs {};
  template <typename F, typename W> static __device__ __forceinline__ void load(Regs<F>&, const W*, long) {}
  template <typename F, typename W> static __device__ __forceinline__ void store(const Regs<F>&, W*, long) {}
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>&, F x, const Ctx&, u32, u32&) {
    F s = x;
    if (s < (F)-1) s = (F)-1;  // f32::clamp: comparisons, so -0.0 and NaN come through unchanged
    if (s > (F)1) s = (F)1;
    return s != s ? (F)0 : s;
  }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
  }
  template <typename F> static __device__ __forceinline__ void on_event(Regs<F>&, u32, u32, u64, u32) {}
};

// PolyBlep -- polyblep.rs:123-508: fourteen waveforms with polynomial band-limiting of their steps (blep) and corners
// (blamp).  Every waveform is + - * / and comparisons in the reference's order, except the four that call sin
// (Sine, Cosine, Half/FullWaveRectifiedSine, and every waveform above sample_rate / 4): device libm, tolerance only.
// slots: 0 t (phase 0..1)  1 dt = freq / sample_rate  2 pulse_width  3 waveform (u32)  4 dt * sample_rate >= sample_rate / 4
struct PolyBlepOsc : StageDefaults {
  static constexpr int kSlots = 5;
  static constexpr u32 kMutableMask = 0b00001u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = false;
  static constexpr bool kHasSeg = false;
  template <typename F> struct Regs { F t, dt, pw; u32 wf, fast; };
  template <typename F, typename W>
  static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long st) {
    r.t = word_to_f<F>(s[0]); r.dt = word_to_f<F>(s[st]); r.pw = word_to_f<F>(s[2 * st]);
    r.wf = (u32)s[3 * st]; r.fast = (u32)s[4 * st];
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void store(const Regs<F>& r, W* s, long) { s[0] = f_to_word(r.t); }
  static __device__ __forceinline__ float trunc_f(float v) { return __builtin_truncf(v); }
  static __device__ __forceinline__ double trunc_f(double v) { return __builtin_trunc(v); }
  static __device__ __forceinline__ float sin_f(float v) { return __ocml_sin_f32(v); }
  static __device__ __forceinline__ double sin_f(double v) { return __ocml_sin_f64(v); }
  static __device__ __forceinline__ float cos_f(float v) { return __ocml_cos_f32(v); }
  static __device__ __forceinline__ double cos_f(double v) { return __ocml_cos_f64(v); }
  template <typename F> static __device__ __forceinline__ F wrap(F v) { return v - trunc_f(v); }  // t -= bitwise_or_zero(t)
  template <typename F> static __device__ __forceinline__ F sq(F v) { return v * v; }
  template <typename F> static __device__ __forceinline__ F blep(F t, F dt) {  // :49-57
    if (t < dt) return -sq<F>(t / dt - (F)1);
    if (t > (F)1 - dt) return sq<F>((t - (F)1) / dt + (F)1);
    return (F)0;
  }
  template <typename F> static __device__ __forceinline__ F blamp(F t, F dt) {  // :60-70
    if (t < dt) { t = t / dt - (F)1; return ((F)-1 / (F)3) * sq<F>(t) * t; }
    if (t > (F)1 - dt) { t = (t - (F)1) / dt + (F)1; return ((F)1 / (F)3) * sq<F>(t) * t; }
    return (F)0;
  }
  template <typename F> static __device__ __forceinline__ F clamp1(F v) { return v < (F)-1 ? (F)-1 : (v > (F)1 ? (F)1 : v); }
  template <typename F> static __device__ __forceinline__ F fold_tri(F y) {  // the 4t triangle fold shared by tri / trap / trap2
    if (y >= (F)3) return y - (F)4;
    if (y > (F)1) return (F)2 - y;
    return y;
  }
  // One waveform, one sample (the reference's method of the same name; line numbers in polyblep.rs).
  template <typename F, int WF> static __device__ __forceinline__ F wave(F t, F dt, F pw_in) {
    constexpr F TAU = (F)6.28318530717958647692528676655900577, PI = (F)3.14159265358979323846264338327950288;
    if constexpr (WF == 1) {
      return sin_f(t * TAU);
    } else if constexpr (WF == 2) {
      return cos_f(t * TAU);
    } else if constexpr (WF == 3) {  // tri, :264-285
      const F t1 = wrap<F>(t + (F)0.25), t2 = wrap<F>(t + (F)0.75);
      F y = fold_tri<F>(t * (F)4);
      return y + (F)4 * dt * (blamp<F>(t1, dt) - blamp<F>(t2, dt));
    } else if constexpr (WF == 4) {  // sqr, :428-441
      const F t2 = wrap<F>(t + (F)0.5);
      const F y = t < (F)0.5 ? (F)1 : (F)-1;
      return y + (blep<F>(t, dt) - blep<F>(t2, dt));
    } else if constexpr (WF == 5) {  // rect, :471-484
      const F t2 = wrap<F>(t + (F)1 - pw_in);
      F y = (F)-2 * pw_in;
      if (t < pw_in) y = y + (F)2;
      return y + (blep<F>(t, dt) - blep<F>(t2, dt));
    } else if constexpr (WF == 6) {  // ramp, :496-504
      const F u = wrap<F>(t);
      const F y = (F)1 - (F)2 * u;
      return y + blep<F>(u, dt);
    } else if constexpr (WF == 7) {  // tri2, :287-311
      F pw = pw_in < (F)0.9999 ? pw_in : (F)0.9999;  // f32::min / max: a NaN pulse width turns into the bound
      if (!(pw_in == pw_in)) pw = (F)0.9999;
      pw = pw > (F)0.0001 ? pw : (F)0.0001;
      const F t1 = wrap<F>(t + (F)0.5 * pw), t2 = wrap<F>(t + (F)1 - (F)0.5 * pw);
      F y = t * (F)2;
      if (y >= (F)2 - pw) y = (y - (F)2) / pw;
      else if (y >= pw) y = (F)1 - (y - pw) / ((F)1 - pw);
      else y = y / pw;
      return y + dt / (pw - pw * pw) * (blamp<F>(t1, dt) - blamp<F>(t2, dt));
    } else if constexpr (WF == 8) {  // sqr2, :443-469
      F t1 = wrap<F>(t + (F)0.875 + (F)0.25 * (pw_in - (F)0.5));
      F t2 = wrap<F>(t + (F)0.375 + (F)0.25 * (pw_in - (F)0.5));
      F y = t1 < (F)0.5 ? (F)1 : (F)-1;
      y = y + (blep<F>(t1, dt) - blep<F>(t2, dt));
      t1 = wrap<F>(t1 + (F)0.5 * ((F)1 - pw_in));
      t2 = wrap<F>(t2 + (F)0.5 * ((F)1 - pw_in));
      y = y + (t1 < (F)0.5 ? (F)1 : (F)-1);
      y = y + (blep<F>(t1, dt) - blep<F>(t2, dt));
      return (F)0.5 * y;
    } else if constexpr (WF == 9) {  // half, :231-247
      const F t2 = wrap<F>(t + (F)0.5);
      F y = t < (F)0.5 ? (F)2 * sin_f(t * TAU) - (F)2 / PI : (F)-2 / PI;
      return y + TAU * dt * (blamp<F>(t, dt) + blamp<F>(t2, dt));
    } else if constexpr (WF == 10) {  // full, :249-257
      const F u = wrap<F>(t + (F)0.25);
      const F y = (F)2 * sin_f(u * PI) - (F)4 / PI;
      return y + TAU * dt * blamp<F>(u, dt);
    } else if constexpr (WF == 11) {  // trip, :313-351
      const F pw = pw_in;
      const F t1 = wrap<F>(t + (F)0.75 + (F)0.5 * pw);
      F y;
      if (t1 >= pw) {
        y = -pw;
      } else {
        y = (F)4 * t1;
        y = y >= (F)2 * pw ? (F)4 - y / pw - pw : y / pw - pw;
      }
      if (pw > (F)0) {
        const F t2 = wrap<F>(t1 + (F)1 - (F)0.5 * pw), t3 = wrap<F>(t1 + (F)1 - pw);
        y = y + (F)2 * dt / pw * (blamp<F>(t1, dt) - (F)2 * blamp<F>(t2, dt) + blamp<F>(t3, dt));
      }
      return y;
    } else if constexpr (WF == 12) {  // trap, :353-386
      F y = fold_tri<F>((F)4 * t);
      y = clamp1<F>((F)2 * y);
      F t1 = wrap<F>(t + (F)0.125), t2 = wrap<F>(t1 + (F)0.5);
      y = y + (F)4 * dt * (blamp<F>(t1, dt) - blamp<F>(t2, dt));
      t1 = wrap<F>(t + (F)0.375);
      t2 = wrap<F>(t1 + (F)0.5);
      return y + (F)4 * dt * (blamp<F>(t1, dt) - blamp<F>(t2, dt));
    } else if constexpr (WF == 13) {  // trap2, :388-426
      F pw = pw_in < (F)0.9999 ? pw_in : (F)0.9999;
      if (!(pw_in == pw_in)) pw = (F)0.9999;
      const F scale = (F)1 / ((F)1 - pw);
      F y = fold_tri<F>((F)4 * t);
      y = clamp1<F>(scale * y);
      F t1 = wrap<F>(t + (F)0.25 - (F)0.25 * pw), t2 = wrap<F>(t1 + (F)0.5);
      y = y + scale * (F)2 * dt * (blamp<F>(t1, dt) - blamp<F>(t2, dt));
      t1 = wrap<F>(t + (F)0.25 + (F)0.25 * pw);
      t2 = wrap<F>(t1 + (F)0.5);
      return y + scale * (F)2 * dt * (blamp<F>(t1, dt) - blamp<F>(t2, dt));
    } else {  // saw (0, and every out-of-range value), :486-494
      const F u = wrap<F>(t + (F)0.5);
      const F y = (F)2 * u - (F)1;
      return y - blep<F>(u, dt);
    }
  }
  // N consecutive samples of one waveform: get_and_inc, :224-228, N times
  template <typename F, int N> struct Run { F y[N]; F t; };
  template <typename F, int N, int WF> static __device__ __forceinline__ void run(Run<F, N>& q, F t, F dt, F pw) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      q.y[k] = wave<F, WF>(t, dt, pw);
      t = t + dt;  // inc, :219-222
      t = t - trunc_f(t);
    }
    q.t = t;
  }
  // One instance of each N per kernel, called, with the waveform dispatched once per call: inlined into every sample
  // of an unrolled tile the fourteen waveforms would be ~400 instructions x 32, and called once per sample the entry
  // (a full s_waitcnt, the dispatch tree, two far jumps) costs more than most waveforms.
  template <typename F, int N> static __device__ __attribute__((noinline)) Run<F, N> samples(F t, F dt, F pw, u32 wf, u32 fast) {
    Run<F, N> q;
    if (fast) wf = 1u;  // next_sample, :210-212: a sine at or above sample_rate / 4
    switch (wf) {
      case 1u: run<F, N, 1>(q, t, dt, pw); break;
      case 2u: run<F, N, 2>(q, t, dt, pw); break;
      case 3u: run<F, N, 3>(q, t, dt, pw); break;
      case 4u: run<F, N, 4>(q, t, dt, pw); break;
      case 5u: run<F, N, 5>(q, t, dt, pw); break;
      case 6u: run<F, N, 6>(q, t, dt, pw); break;
      case 7u: run<F, N, 7>(q, t, dt, pw); break;
      case 8u: run<F, N, 8>(q, t, dt, pw); break;
      case 9u: run<F, N, 9>(q, t, dt, pw); break;
      case 10u: run<F, N, 10>(q, t, dt, pw); break;
      case 11u: run<F, N, 11>(q, t, dt, pw); break;
      case 12u: run<F, N, 12>(q, t, dt, pw); break;
      case 13u: run<F, N, 13>(q, t, dt, pw); break;
      default: run<F, N, 0>(q, t, dt, pw); break;
    }
    return q;
  }
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F, const Ctx&, u32, u32&) {
    const Run<F, 1> q = samples<F, 1>(r.t, r.dt, r.pw, r.wf, r.fast);
    r.t = q.t;
    return q.y[0];
  }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
    static_assert(T % 8 == 0, "tiles are multiples of eight samples");
#pragma unroll
    for (int j = 0; j < T; j += 8) {
      const Run<F, 8> q = samples<F, 8>(r.t, r.dt, r.pw, r.wf, r.fast);
#pragma unroll
      for (int k = 0; k < 8; ++k) x[j + k] = q.y[k];
      r.t = q.t;
    }
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits, u32) {
    if ((op & 0x7Fu) != EV_SET) return;
    const F v = word_to_f<F>((typename WordOf<F>::type)bits);
    switch (rel) {
      case 0: r.t = v; break; case 1: r.dt = v; break; case 2: r.pw = v; break;
      case 3: r.wf = (u32)bits; break; default: r.fast = (u32)bits; break;
    }
  }
};

// BufferReader<F, U1> -- buffer.rs:19-191: plays the bank's shared Buffer (dsp/buffer.rs) from an f64 read pointer with
// linear interpolation (Buffer::get_linear_interp_f64, :100-110), per-voice rate, start and end, looping or one-shot
// (mark_done(i + 1) at the frame after the last one, then silence).  All positions are f64 for any F.
// slots: 0,1 read_pointer  2,3 step (= base_rate * rate)  4,5 start_frame  6,7 end_frame  8 finished  9 looping
struct BufferReader : StageDefaults {
  static constexpr int kSlots = 10;
  static constexpr u32 kMutableMask = 0b0100000011u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = true;  // it marks done and can be "stopped"
  static constexpr bool kNeedsBind = true;
  static constexpr bool kHasSeg = true;
  template <typename F> struct Regs { double rp, step, start, end; u32 finished, looping, seg; const F* buf; u32 n; };
  template <typename F> static __device__ __forceinline__ bool is_stopped(const Regs<F>& r) { return r.finished != 0u; }
  template <typename W> static __device__ __forceinline__ double ld2(const W* s, long st, int k) {
    const u64 lo = (u32)s[(long)k * st], hi = (u32)s[(long)(k + 1) * st];
    return __builtin_bit_cast(double, lo | (hi << 32));
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long st) {
    r.rp = ld2(s, st, 0); r.step = ld2(s, st, 2); r.start = ld2(s, st, 4); r.end = ld2(s, st, 6);
    r.finished = (u32)s[8 * st]; r.looping = (u32)s[9 * st];
    r.seg = 0; r.buf = nullptr; r.n = 0;
  }
  template <typename F>
  static __device__ __forceinline__ void bind(Regs<F>& r, const Ctx& c) {
    r.buf = reinterpret_cast<const F*>(c.buffer);
    r.n = c.buffer_frames;
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void store(const Regs<F>& r, W* s, long st) {
    const u64 b = __builtin_bit_cast(u64, r.rp);
    s[0] = (W)(u32)b; s[st] = (W)(u32)(b >> 32); s[8 * st] = (W)r.finished;
  }
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F, const Ctx& c, u32 frame, u32& done_frame) {
    bind<F>(r, c);
    if (r.finished || r.n == 0u) return (F)0;
    // get_linear_interp_f64: mix = fract(index); buffer[i] * (1 - mix) + buffer[(i + 1) % len] * mix.  `index as usize`
    // saturates at 0; an index past the end is undefined behaviour in the reference, clamped here.
    const double ip = __builtin_trunc(r.rp);
    const F mix = (F)(r.rp - ip);
    u32 i = r.rp > 0.0 ? (r.rp < 4294967040.0 ? (u32)r.rp : 0xFFFFFFFFu) : 0u;
    if (i >= r.n) i = r.n - 1u;
    const u32 i1 = i + 1u == r.n ? 0u : i + 1u;
    const F y = r.buf[i] * ((F)1 - mix) + r.buf[i1] * mix;
    r.rp += r.step;
    if (r.rp >= r.end) {  // process_block, :166-173
      if (r.looping) { r.rp = r.start; }
      else { r.finished = 1u; done_frame = frame + 1u - r.seg; }
    }
    return y;
  }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits, u32 frame) {
    if (op & EV_SPLIT) r.seg = frame;
    if ((op & 0x7Fu) != EV_SET) return;
    const u32 w = (u32)bits;
    auto lo = [](double d, u32 v) { return __builtin_bit_cast(double, (__builtin_bit_cast(u64, d) & 0xFFFFFFFF00000000ull) | (u64)v); };
    auto hi = [](double d, u32 v) { return __builtin_bit_cast(double, (__builtin_bit_cast(u64, d) & 0x00000000FFFFFFFFull) | ((u64)v << 32)); };
    switch (rel) {
      case 0: r.rp = lo(r.rp, w); break;      case 1: r.rp = hi(r.rp, w); break;
      case 2: r.step = lo(r.step, w); break;  case 3: r.step = hi(r.step, w); break;
      case 4: r.start = lo(r.start, w); break; case 5: r.start = hi(r.start, w); break;
      case 6: r.end = lo(r.end, w); break;    case 7: r.end = hi(r.end, w); break;
      case 8: r.finished = w; break;
      default: r.looping = w; break;
    }
  }
};

// SinNumeric -- osc.rs:222-271.  slots: 0 phase, 1 phase_offset, 2 phase_increment
struct SinNum : StageDefaults {
  static constexpr int kSlots = 3;
  static constexpr u32 kMutableMask = 0b001u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = false;
  static constexpr bool kHasSeg = false;
  template <typename F> struct Regs { F phase, off, inc; };
  static constexpr u32 kParamMask = 0b110u;
  template <typename R> static __device__ __forceinline__ void take_params(R& r, const R& n, bool c) {
    r.off = c ? n.off : r.off;
    r.inc = c ? n.inc : r.inc;
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long stride) {
    r.phase = word_to_f<F>(s[0]); r.off = word_to_f<F>(s[stride]); r.inc = word_to_f<F>(s[2 * stride]);
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void store(const Regs<F>& r, W* s, long) { s[0] = f_to_word(r.phase); }
  static __device__ __forceinline__ float sin_f(float v) { return __ocml_sin_f32(v); }
  static __device__ __forceinline__ double sin_f(double v) { return __ocml_sin_f64(v); }
  // (p * TAU).sin() for a phase p in revolutions (osc.rs:264).  f32: v_sin_f32 IS sin(2 pi x) of an argument in revolutions,
  // one instruction for the device library's forty.  Over every f32 p of [0, 2) it is within 8.7e-7 of the reference's value
  // -- glibc's sinf of the f32-rounded product -- and within 1.3e-7 of the exact sine (the reference's own rounding of p * TAU
  // is the larger part of the difference): tools/micro/hw_sin.hip, profiles/r03_micro_hw_sin.txt.  Beyond |p| < 2 (a phase
  // offset of several turns, a negative frequency running away) the reference's argument rounding grows with |p| and must be
  // reproduced to stay within tolerance: those samples take the library's sinf of the rounded product, as before.
  // (the choice is per voice and sample: which voices share a wavefront does not change anyone's value)
  static __device__ __forceinline__ float sin_turns(float p) {
    const bool far = !(__builtin_fabsf(p) < 2.0f);
    float y = __builtin_amdgcn_sinf(p);
    if (__builtin_amdgcn_ballot_w64(far) != 0) {
      const float lib = __ocml_sin_f32(p * 6.28318530717958647692f);
      y = far ? lib : y;
    }
    return y;
  }
  static __device__ __forceinline__ double sin_turns(double p) { return __ocml_sin_f64(p * 6.28318530717958647692); }
  // SinNumeric::freq (osc.rs:240-242: F::new(freq) / F::new(sample_rate as f32)) / ::phase_offset (:244-247)
  template <typename F, int P>
  static __device__ __forceinline__ void ar_set(Regs<F>& r, F v, const Ctx& c) {
    if (P == 0) r.inc = v / (F)(float)c.sample_rate;
    else r.off = v;
  }
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F, const Ctx&, u32, u32&) {
    F out = sin_turns(r.phase + r.off);
    r.phase += r.inc;
    if (r.phase > (F)1) r.phase -= (F)1;
    return out;
  }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits, u32 /*frame*/) {
    if ((op & 0x7Fu) != EV_SET) return;
    F v = word_to_f<F>((typename WordOf<F>::type)bits);
    if (rel == 0) r.phase = v;
    else if (rel == 1) r.off = v;
    else r.inc = v;
  }
};

// SinNumeric cut in two for the pipeline kernels (voice_pipe.hpp): the SERIAL part -- the phase accumulator, three
// instructions per sample -- and the part that is a pure function of one sample, sin(p * TAU), some forty instructions.
// SinPhase hands on p = phase + phase_offset (the reference's first operation on it, osc.rs:264), SinMap finishes
// ((p) * TAU).sin(): the same operations in the same order as SinNum::tick, so the same bits.  SinMap has no state and
// can therefore be spread over several wavefronts, each taking a slice of every tile (Fan groups).
// SinPhase slots: 0 phase, 1 phase_offset, 2 phase_increment (= SinNum's); SinMap: none
struct SinPhase : StageDefaults {
  static constexpr int kSlots = 3;
  static constexpr u32 kMutableMask = 0b001u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = false;
  static constexpr bool kHasSeg = false;
  template <typename F> struct Regs { F phase, off, inc; };
  static constexpr u32 kParamMask = 0b110u;
  template <typename R> static __device__ __forceinline__ void take_params(R& r, const R& n, bool c) {
    r.off = c ? n.off : r.off;
    r.inc = c ? n.inc : r.inc;
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long stride) {
    r.phase = word_to_f<F>(s[0]); r.off = word_to_f<F>(s[stride]); r.inc = word_to_f<F>(s[2 * stride]);
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void store(const Regs<F>& r, W* s, long) { s[0] = f_to_word(r.phase); }
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F, const Ctx&, u32, u32&) {
    const F p = r.phase + r.off;
    r.phase += r.inc;
    if (r.phase > (F)1) r.phase -= (F)1;
    return p;
  }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits, u32 /*frame*/) {
    if ((op & 0x7Fu) != EV_SET) return;
    F v = word_to_f<F>((typename WordOf<F>::type)bits);
    if (rel == 0) r.phase = v;
    else if (rel == 1) r.off = v;
    else r.inc = v;
  }
};
struct SinMap : StageDefaults {
  static constexpr int kSlots = 0;
  static constexpr u32 kMutableMask = 0u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = false;
  static constexpr bool kHasSeg = false;
  template <typename F> struct Regs {};
  template <typename F, typename W> static __device__ __forceinline__ void load(Regs<F>&, const W*, long) {}
  template <typename F, typename W> static __device__ __forceinline__ void store(const Regs<F>&, W*, long) {}
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>&, F p, const Ctx&, u32, u32&) {
    return SinNum::sin_turns(p);
  }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
  }
  template <typename F> static __device__ __forceinline__ void on_event(Regs<F>&, u32, u32, u64, u32) {}
};

// SvfFilter tick -- svf.rs:272-278.  slots: 0 ic1eq, 1 ic2eq, 2 a1, 3 a2, 4 a3, 5 m0, 6 m1, 7 m2
// All nine filter types share this tick; the type only changes the coefficients (host side).
struct Svf : StageDefaults {
  static constexpr int kSlots = 8;
  static constexpr u32 kMutableMask = 0b11u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = false;
  static constexpr bool kHasSeg = false;
  // nl ("not low"): zero iff the output mix is the low-pass one (m0 = 0, m1 = 0, m2 = 1 to the bit), kept beside the three so
  // that the per-tile choice of the step (low_pass() below) looks at one register, not at three that are otherwise idle in
  // that step.  m0 is NOT next to m1, m2 on purpose: as neighbours the three were accessed as overlapping two-float vectors
  // -- (m0, m1) where they are loaded, (m1, m2) where the packed step wants them -- and a struct slice with overlapping vector
  // accesses is not promoted to registers: it became a 12-byte object in scratch memory, a memory round trip per tile away.
  template <typename F> struct Regs { F ic1, ic2, a1, a2, a3, m1, m2; typename WordOf<F>::type nl; F m0; };
  template <typename R> static __device__ __forceinline__ void note_mix(R& r) {
    r.nl = (f_to_word(r.m0) | f_to_word(r.m1)) | (f_to_word(r.m2) ^ f_to_word((decltype(r.m2))1));
  }
  static constexpr u32 kParamMask = 0b11111100u;  // the six coefficients (every setter recomputes them on the host)
  template <typename R> static __device__ __forceinline__ void take_params(R& r, const R& n, bool c) {
    r.a1 = c ? n.a1 : r.a1; r.a2 = c ? n.a2 : r.a2; r.a3 = c ? n.a3 : r.a3;
    r.m0 = c ? n.m0 : r.m0; r.m1 = c ? n.m1 : r.m1; r.m2 = c ? n.m2 : r.m2;
    r.nl = c ? n.nl : r.nl;
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long st) {
    r.ic1 = word_to_f<F>(s[0]); r.ic2 = word_to_f<F>(s[st]); r.a1 = word_to_f<F>(s[2 * st]);
    r.a2 = word_to_f<F>(s[3 * st]); r.a3 = word_to_f<F>(s[4 * st]); r.m0 = word_to_f<F>(s[5 * st]);
    r.m1 = word_to_f<F>(s[6 * st]); r.m2 = word_to_f<F>(s[7 * st]);
    note_mix(r);
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void store(const Regs<F>& r, W* s, long st) {
    s[0] = f_to_word(r.ic1); s[st] = f_to_word(r.ic2);
  }
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F v0, const Ctx&, u32, u32&) {
    const F v3 = v0 - r.ic2;
    if constexpr (FMA) {
      const F v1 = mad<true>(r.a2, v3, r.a1 * r.ic1);
      const F v2 = mad<true>(r.a3, v3, mad<true>(r.a2, r.ic1, r.ic2));
      r.ic1 = mad<true>((F)2, v1, -r.ic1);
      r.ic2 = mad<true>((F)2, v2, -r.ic2);
      return mad<true>(r.m2, v2, mad<true>(r.m1, v1, r.m0 * v0));
    } else {
      const F v1 = r.a1 * r.ic1 + r.a2 * v3;
      const F v2 = r.ic2 + r.a2 * r.ic1 + r.a3 * v3;
      // 2*v is exact in binary floating point, so round(2*v - ic) is the same value whether the
      // product is rounded first or not: one FMA gives the reference's two-instruction result bit
      // for bit (the only exception, |v| >= 2^127 where 2*v alone would overflow, is far outside
      // any filter state that is not already garbage).
      r.ic1 = mad<true>((F)2, v1, -r.ic1);
      r.ic2 = mad<true>((F)2, v2, -r.ic2);
      return r.m0 * v0 + r.m1 * v1 + r.m2 * v2;
    }
  }
  // f32, exact arithmetic: the same fifteen roundings as tick(), issued as ten instructions per sample.  Independent pairs share
  // a packed instruction -- (a1*ic1, a2*ic1), (a2*v3, a3*v3), (v1, v2), (ic1', ic2'), (m1*v1, m2*v2) -- with the state kept in
  // an aligned register pair so that no moves are needed.  A wavefront alone on its SIMD is issue-bound (tools/micro/
  // valu_issue.hip, svf_low_variants.hip: ~4.15 cycles per instruction whatever it is, an s_nop included), so the instruction
  // count of this loop IS the block time of the filter wave.  Halves of a pair cannot be named through asm operands, hence the
  // fixed registers.  (Rounds 1-2 kept packed results away from the instruction right behind them and carried the output's
  // three terms over to the next sample's step to fill those places -- nine and a half instructions; the hardware needs no
  // such wait state, and the carried two-float values are what the compiler's subregister renaming pass crashed on once a
  // second filter path sat beside this one.)
  // KNH_SVF_NOP=1 (a -D of the build, KNH_EXTRA_FLAGS; for run-time fused kernels the environment variable of the same name
  // at knh_bank_init): the wait state the compiler's hazard table would put between a packed-f32 instruction and an
  // instruction that reads its result goes back in, at the three places where the hand-written steps have none.  The default
  // build relies on the hardware needing none (measured: same bits, profiles/r03_micro_svf_low_variants.txt); the switch is
  // what a part or a ROCm on which that stops holding is diagnosed with -- tests/test_gpu_properties.py::
  // test_svf_steps_with_and_without_the_wait_state compares the two bit for bit at one, two and four wavefronts per SIMD.
#ifdef KNH_SVF_NOP
#define KNH_SVF_WAIT "s_nop 0\n\t"
#else
#define KNH_SVF_WAIT
#endif
  // The fixed registers v100 .. v122 of the two steps below need a kernel with at least 123 VGPRs: every kernel form has 128 or
  // more (the 1 024-thread sixteen-groups form exactly 128: __launch_bounds__(1024) on a CU with 512 VGPRs per SIMD lane set);
  // a form with a smaller budget (more than 1 024 threads per workgroup does not exist; a waves-per-SIMD attribute above 4
  // would be one) must not include them.
  static constexpr int kHighestFixedVgpr = 122;
  template <int T>
  static __device__ __forceinline__ void tick_tile_packed(Regs<float>& r, float (&x)[T]) {
    static_assert(T % 8 == 0, "the filter tile is unrolled in blocks of eight samples");
    static_assert(kHighestFixedVgpr < 128, "the fixed registers stay inside the smallest register budget of any kernel form (128: __launch_bounds__(1024))");
    // Registers, all named and all below v128 (the sixteen-groups-per-workgroup kernels have 128) -- every operand of the asm is
    // a single 32-bit register; see tick_tile_low for why:
    //   v[100:101] (ic1, ic2)  v[102:103] P1  v[104:105] P2  v[106:107] (v1, v2)  v[108:109] (m1*v1, m2*v2)  v112 the output sum
    //   v114 v3  v[116:117] (a1, a2)  v[118:119] (a2, a3)  v[120:121] (m1, m2)  v122 m0
    float ic1 = r.ic1, ic2 = r.ic2;
    const float a1 = r.a1, a2 = r.a2, a2b = r.a2, a3 = r.a3, m0 = r.m0, m1 = r.m1, m2 = r.m2;
#define KNH_SVF_STEP(K)                                                                                            \
      "v_sub_f32 v114, %[x" #K "], v101\n\t"                              /* v3 = x - ic2                     */   \
      "v_pk_mul_f32 v[102:103], v[116:117], v[100:101] op_sel_hi:[1,0]\n\t"   /* (a1*ic1, a2*ic1)             */   \
      "v_pk_mul_f32 v[104:105], v[118:119], v[114:115] op_sel_hi:[1,0]\n\t"   /* (a2*v3, a3*v3)               */   \
      "v_add_f32 v103, v101, v103\n\t"                                    /* ic2 + a2*ic1                     */   \
      "v_pk_add_f32 v[106:107], v[102:103], v[104:105]\n\t"               /* (v1, v2)                         */   \
      "v_mul_f32 v112, v122, %[x" #K "]\n\t"                              /* m0*x                             */   \
      "v_pk_fma_f32 v[100:101], v[106:107], 2.0, v[100:101] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"    \
      "v_pk_mul_f32 v[108:109], v[120:121], v[106:107]\n\t"               /* (m1*v1, m2*v2)                   */   \
      KNH_SVF_WAIT                                                                                                     \
      "v_add_f32 v112, v112, v108\n\t"                                    /* m0*x + m1*v1                     */   \
      "v_add_f32 %[y" #K "], v112, v109\n\t"                              /* ... + m2*v2 -> output            */
#pragma unroll
    for (int j = 0; j < T; j += 8) {
      float y0, y1, y2, y3, y4, y5, y6, y7;
      asm volatile(KNH_SVF_STEP(0) KNH_SVF_STEP(1) KNH_SVF_STEP(2) KNH_SVF_STEP(3) KNH_SVF_STEP(4) KNH_SVF_STEP(5) KNH_SVF_STEP(6) KNH_SVF_STEP(7)
                   : [y0] "=&v"(y0), [y1] "=&v"(y1), [y2] "=&v"(y2), [y3] "=&v"(y3), [y4] "=&v"(y4), [y5] "=&v"(y5),
                     [y6] "=&v"(y6), [y7] "=&v"(y7), "+{v100}"(ic1), "+{v101}"(ic2)
                   : [x0] "v"(x[j]), [x1] "v"(x[j + 1]), [x2] "v"(x[j + 2]), [x3] "v"(x[j + 3]), [x4] "v"(x[j + 4]),
                     [x5] "v"(x[j + 5]), [x6] "v"(x[j + 6]), [x7] "v"(x[j + 7]), "{v116}"(a1), "{v117}"(a2), "{v118}"(a2b), "{v119}"(a3),
                     "{v120}"(m1), "{v121}"(m2), "{v122}"(m0)
                   : "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v112", "v114", "v115");
      x[j] = y0; x[j + 1] = y1; x[j + 2] = y2; x[j + 3] = y3; x[j + 4] = y4; x[j + 5] = y5; x[j + 6] = y6; x[j + 7] = y7;
    }
#undef KNH_SVF_STEP
    r.ic1 = ic1; r.ic2 = ic2;
  }
  // The low-pass output (svf.rs:148-157: m0 = 0, m1 = 0, m2 = 1) without its three multiplies and two adds, bit for bit:
  //   (0*x + 0*v1) + 1*v2 = z + v2 with z = +-0 while x and v1 are finite, and z + v2 = v2 unless v2 = -0;
  //   v2 = (ic2 + a2*ic1) + a3*v3 is -0 only if both terms are, so only if ic2 = -0; and ic2' = 2*v2 - ic2 is never -0 (it
  //   would need v2 = -0 with ic2 = +0), so v2 = -0 can only happen in the first sample after ic2 was SET to -0: low_pass()
  //   refuses that state and the general step runs.  When x or v1 is not finite the reference returns NaN (0*inf); x not
  //   finite makes v1 = a1*ic1 + a2*(x - ic2) not finite too, so fma(0, v1, v2) -- v2 when v1 is finite, NaN when it is not,
  //   and +0 + v2 = v2 -- is the reference's value in every case, in one instruction.
  // The step is then seven instructions: v3, (a1*ic1, a2*ic1), (a2*v3, a3*v3), ic2 + a2*ic1, (v1, v2), the output,
  // (ic1', ic2').  Packed results are read by the very next instruction (the output reads the packed add's, the next
  // sample's first instruction the packed fma's).  The compiler would put an s_nop there (its hazard table takes op_sel_hi of a packed f32 instruction
  // for a half-register write: the dst_sel forwarding rule of gfx940); the hardware needs none -- the step with and without
  // it gives the same bits over 1.3e9 samples, and 29.25 against 33.25 cycles per sample for a wavefront alone on its SIMD
  // (tools/micro/svf_low_variants.hip, profiles/r03_micro_svf_low_variants.txt): an s_nop costs an issue slot like any
  // instruction.
  template <typename F> static __device__ __forceinline__ bool low_pass(const Regs<F>& r) {
    typedef typename WordOf<F>::type W;
    const W neg0 = (W)1 << (sizeof(F) * 8 - 1);
    const bool mine = r.nl == 0 && f_to_word(r.ic2) != neg0;
    return __builtin_amdgcn_ballot_w64(!mine) == 0;
  }
  template <int T>
  static __device__ __forceinline__ void tick_tile_low(Regs<float>& r, float (&x)[T]) {
    static_assert(T % 8 == 0, "the filter tile is unrolled in blocks of eight samples");
    // Registers, all named and all below v128: v[100:101] (ic1, ic2)  v[102:103] P1  v[104:105] P2  v[106:107] (v1, v2)  v114 v3
    // v[116:117] (a1, a2)  v[118:119] (a2, a3).  Every operand of the asm is a single 32-bit register: no 64-bit C++ value is
    // tied to it.  (With the state and the coefficients as two-float vectors, as in tick_tile_packed, the compiler's "rename
    // disconnected subregister components" pass crashed on voices with two filters once this step and the general one sat
    // in one function -- inside hiprtc, and therefore inside the host process.)
    float ic1 = r.ic1, ic2 = r.ic2;
    const float a1 = r.a1, a2 = r.a2, a2b = r.a2, a3 = r.a3;
#define KNH_SVF_LOW(K)                                                                                             \
      "v_sub_f32 v114, %[x" #K "], v101\n\t"                              /* v3 = x - ic2                     */   \
      "v_pk_mul_f32 v[102:103], v[116:117], v[100:101] op_sel_hi:[1,0]\n\t"   /* (a1*ic1, a2*ic1)             */   \
      "v_pk_mul_f32 v[104:105], v[118:119], v[114:115] op_sel_hi:[1,0]\n\t"   /* (a2*v3, a3*v3)               */   \
      "v_add_f32 v103, v101, v103\n\t"                                    /* ic2 + a2*ic1                     */   \
      "v_pk_add_f32 v[106:107], v[102:103], v[104:105]\n\t"               /* (v1, v2)                         */   \
      KNH_SVF_WAIT                                                                                                     \
      "v_fma_f32 %[y" #K "], 0, v106, v107\n\t"                           /* v2 (NaN if v1 is not finite)     */   \
      "v_pk_fma_f32 v[100:101], v[106:107], 2.0, v[100:101] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"    \
      KNH_SVF_WAIT
#pragma unroll
    for (int j = 0; j < T; j += 8) {
      float y0, y1, y2, y3, y4, y5, y6, y7;
      asm volatile(KNH_SVF_LOW(0) KNH_SVF_LOW(1) KNH_SVF_LOW(2) KNH_SVF_LOW(3) KNH_SVF_LOW(4) KNH_SVF_LOW(5) KNH_SVF_LOW(6) KNH_SVF_LOW(7)
                   : [y0] "=&v"(y0), [y1] "=&v"(y1), [y2] "=&v"(y2), [y3] "=&v"(y3), [y4] "=&v"(y4), [y5] "=&v"(y5),
                     [y6] "=&v"(y6), [y7] "=&v"(y7), "+{v100}"(ic1), "+{v101}"(ic2)
                   : [x0] "v"(x[j]), [x1] "v"(x[j + 1]), [x2] "v"(x[j + 2]), [x3] "v"(x[j + 3]), [x4] "v"(x[j + 4]),
                     [x5] "v"(x[j + 5]), [x6] "v"(x[j + 6]), [x7] "v"(x[j + 7]), "{v116}"(a1), "{v117}"(a2), "{v118}"(a2b), "{v119}"(a3)
                   : "v102", "v103", "v104", "v105", "v106", "v107", "v114", "v115");
      x[j] = y0; x[j + 1] = y1; x[j + 2] = y2; x[j + 3] = y3; x[j + 4] = y4; x[j + 5] = y5; x[j + 6] = y6; x[j + 7] = y7;
    }
#undef KNH_SVF_LOW
    r.ic1 = ic1; r.ic2 = ic2;
  }
  // the same in f64: eleven instructions for the general step's fifteen, every operand at least two instructions old
  template <int T>
  static __device__ __forceinline__ void tick_tile_low_f64(Regs<double>& r, double (&x)[T]) {
    double ic1 = r.ic1, ic2 = r.ic2;
    const double a1 = r.a1, a2 = r.a2, a3 = r.a3;
    double v1p = 0.0, v2p = 0.0;
#define KNH_STEP(stmt) stmt; __builtin_amdgcn_sched_barrier(0)
#pragma unroll
    for (int j = 0; j < T; ++j) {
      KNH_STEP(const double v3 = x[j] - ic2);
      KNH_STEP(const double p2 = a2 * ic1);
      KNH_STEP(const double p1 = a1 * ic1);
      KNH_STEP(const double q2 = a3 * v3);
      KNH_STEP(const double q1 = a2 * v3);
      KNH_STEP(const double t = ic2 + p2);
      if (j > 0) { KNH_STEP(x[j - 1] = __builtin_fma(0.0, v1p, v2p)); }
      KNH_STEP(const double v2 = t + q2);
      KNH_STEP(const double v1 = p1 + q1);
      KNH_STEP(ic2 = __builtin_fma(2.0, v2, -ic2));
      KNH_STEP(ic1 = __builtin_fma(2.0, v1, -ic1));
      v1p = v1; v2p = v2;
    }
    x[T - 1] = __builtin_fma(0.0, v1p, v2p);
#undef KNH_STEP
    r.ic1 = ic1; r.ic2 = ic2;
  }
  // f64, exact arithmetic: the same fifteen roundings per sample as tick(), in an order fixed by hand.  An f64 instruction
  // occupies the SIMD for four cycles (half the f32 rate), so a wavefront alone on its SIMD could run the step in 15 x 4 = 60
  // cycles -- unless an instruction reads the result of the one before it, which holds the issue for about twice that.  The
  // compiler's own schedule has four such pairs per sample (q1 -> v1, p2 -> t, q2 -> v2, o2 -> out: 111 cycles per sample
  // measured, profiles/r02).  Here the four output instructions of sample j - 1 are woven into the recurrence of sample j,
  // and every instruction reads results that are at least three instructions old:
  //    1 v3 = x - ic2      2 p1 = a1 ic1     3 p2 = a2 ic1     4 o1 = m1 v1'      5 q1 = a2 v3
  //    6 q2 = a3 v3        7 t = ic2 + p2    8 o2 = m2 v2'     9 v1 = p1 + q1    10 v2 = t + q2
  //   11 s = o0' + o1     12 ic1 = 2 v1 - ic1   13 ic2 = 2 v2 - ic2   14 out' = s + o2   15 o0 = m0 x      (' = of the sample before)
  // Plain C++, one operation per statement, with a scheduling barrier behind each: the compiler keeps this order (and, unlike
  // with inline asm, knows what the instructions are: it pads nothing).
  template <int T>
  static __device__ __forceinline__ void tick_tile_f64(Regs<double>& r, double (&x)[T]) {
    double ic1 = r.ic1, ic2 = r.ic2;
    const double a1 = r.a1, a2 = r.a2, a3 = r.a3, m0 = r.m0, m1 = r.m1, m2 = r.m2;
    double v1p = 0.0, v2p = 0.0, o0p = 0.0;
#define KNH_STEP(stmt) stmt; __builtin_amdgcn_sched_barrier(0)
#pragma unroll
    for (int j = 0; j < T; ++j) {
      double o1 = 0.0, o2 = 0.0, s = 0.0;
      KNH_STEP(const double v3 = x[j] - ic2);
      KNH_STEP(const double p1 = a1 * ic1);
      KNH_STEP(const double p2 = a2 * ic1);
      if (j > 0) { KNH_STEP(o1 = m1 * v1p); }
      KNH_STEP(const double q1 = a2 * v3);
      KNH_STEP(const double q2 = a3 * v3);
      KNH_STEP(const double t = ic2 + p2);
      if (j > 0) { KNH_STEP(o2 = m2 * v2p); }
      KNH_STEP(const double v1 = p1 + q1);
      KNH_STEP(const double v2 = t + q2);
      if (j > 0) { KNH_STEP(s = o0p + o1); }
      // 2*v is exact, so one FMA gives the reference's two roundings' result bit for bit (see tick())
      KNH_STEP(ic1 = __builtin_fma(2.0, v1, -ic1));
      KNH_STEP(ic2 = __builtin_fma(2.0, v2, -ic2));
      KNH_STEP(const double o0 = m0 * x[j]);
      if (j > 0) { KNH_STEP(x[j - 1] = s + o2); }
      v1p = v1; v2p = v2; o0p = o0;
    }
    x[T - 1] = (o0p + m1 * v1p) + m2 * v2p;  // the last sample's output
#undef KNH_STEP
    r.ic1 = ic1; r.ic2 = ic2;
  }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
    if constexpr (sizeof(F) == 4 && !FMA) {
      if (low_pass(r)) tick_tile_low<T>(r, x); else tick_tile_packed<T>(r, x);
    } else if constexpr (sizeof(F) == 8 && !FMA) {
      if (low_pass(r)) tick_tile_low_f64<T>(r, x); else tick_tile_f64<T>(r, x);
    } else {
#pragma unroll
      for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
    }
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits, u32 /*frame*/) {
    if ((op & 0x7Fu) != EV_SET) return;
    F v = word_to_f<F>((typename WordOf<F>::type)bits);
    // (selects, not a switch: a switch over the fields becomes an indexed store into the struct, which then lives in
    // scratch memory -- and every tile waits for a scratch load of its coefficients)
    r.ic1 = rel == 0 ? v : r.ic1; r.ic2 = rel == 1 ? v : r.ic2; r.a1 = rel == 2 ? v : r.a1; r.a2 = rel == 3 ? v : r.a2;
    r.a3 = rel == 4 ? v : r.a3; r.m0 = rel == 5 ? v : r.m0; r.m1 = rel == 6 ? v : r.m1; r.m2 = rel >= 7 ? v : r.m2;
    note_mix(r);
  }
};

static __device__ __forceinline__ float dev_pow(float a, float b) { return __ocml_pow_f32(a, b); }
static __device__ __forceinline__ double dev_pow(double a, double b) { return __ocml_pow_f64(a, b); }
// SvfFilter with a parameter driven at audio rate (ArP<SvfP, P>: P = 0 cutoff_freq, 1 q, 2 gain): every sample runs the
// setter, which recomputes the coefficients from cutoff, q, gain and the type (svf.rs:81-133 -> set_coeffs :146-242).  The
// three values and the type are therefore device state too (slots 8..11); tan / pow / sqrt are the device library's, so a
// voice with such a filter is compared with the reference within a tolerance, not bit for bit.
// slots: 0..7 as Svf, 8 cutoff, 9 q, 10 gain_db, 11 type
struct SvfP : Svf {
  static constexpr int kSlots = 12;
  static constexpr u32 kParamMask = 0u;  // (graph-shaped voices only: no mid-tile parameter switching there)
  template <typename F> struct Regs : Svf::Regs<F> { F cutoff, q, gain; u32 ty; };
  template <typename F, typename W>
  static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long st) {
    Svf::load<F, W>(r, s, st);
    r.cutoff = word_to_f<F>(s[8 * st]); r.q = word_to_f<F>(s[9 * st]); r.gain = word_to_f<F>(s[10 * st]); r.ty = (u32)s[11 * st];
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void store(const Regs<F>& r, W* s, long st) {
    Svf::store<F, W>(r, s, st);
    // the coefficients and the three values moved with the driving signal: what the next launch starts from
    s[2 * st] = f_to_word(r.a1); s[3 * st] = f_to_word(r.a2); s[4 * st] = f_to_word(r.a3);
    s[5 * st] = f_to_word(r.m0); s[6 * st] = f_to_word(r.m1); s[7 * st] = f_to_word(r.m2);
    s[8 * st] = f_to_word(r.cutoff); s[9 * st] = f_to_word(r.q); s[10 * st] = f_to_word(r.gain);
  }
  static __device__ __forceinline__ float tan_f(float v) { return __ocml_tan_f32(v); }
  static __device__ __forceinline__ double tan_f(double v) { return __ocml_tan_f64(v); }
  static __device__ __forceinline__ float sqrt_f(float v) { return __ocml_sqrt_f32(v); }
  static __device__ __forceinline__ double sqrt_f(double v) { return __ocml_sqrt_f64(v); }
  template <typename F>
  static __device__ __forceinline__ void set_coeffs(Regs<F>& r, const Ctx& c) {  // svf.rs:146-242, as bank.hip's svf_coeffs
    const F one = (F)1, sr = (F)(float)c.sample_rate;  // F::new(sample_rate as f32)
    F g = tan_f(((F)3.14159265358979323846 * r.cutoff) / sr);
    F k = one / r.q;
    F m0 = (F)0, m1 = (F)0, m2 = (F)0, amp = (F)0;
    if (r.ty >= 6u && r.ty <= 8u) amp = dev_pow((F)10, r.gain / (F)40);
    switch (r.ty) {
      default: m2 = one; break;                                       // Low
      case 2u: m1 = one; break;                                       // Band
      case 1u: m0 = one; m1 = -k; m2 = -one; break;                   // High
      case 3u: m0 = one; m1 = -k; break;                              // Notch
      case 4u: m0 = one; m1 = -k; m2 = -(F)2; break;                  // Peak
      case 5u: m0 = one; m1 = -(F)2 * k; break;                       // All
      case 6u: g = g / sqrt_f(amp); k = one / (r.q * amp); m0 = one; m1 = k * (amp * amp - one); break;           // Bell
      case 7u: g = g / sqrt_f(amp); m0 = one; m1 = k * (amp - one); m2 = amp * amp - one; break;                    // LowShelf
      case 8u: g = g * sqrt_f(amp); m0 = amp * amp; m1 = k * (one - amp) * amp; m2 = one - amp * amp; break;        // HighShelf
    }
    r.a1 = one / (one + g * (g + k));
    r.a2 = g * r.a1;
    r.a3 = g * r.a2;
    r.m0 = m0; r.m1 = m1; r.m2 = m2;
  }
  template <typename F, int P>
  static __device__ __forceinline__ void ar_set(Regs<F>& r, F v, const Ctx& c) {
    if (P == 0) r.cutoff = v; else if (P == 1) r.q = v; else r.gain = v;
    set_coeffs<F>(r, c);
  }
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F v0, const Ctx& c, u32 n, u32& d) { return Svf::tick<F, FMA>(r, v0, c, n, d); }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = Svf::tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits, u32 frame) {
    if ((op & 0x7Fu) != EV_SET) return;
    if (rel < 8u) { Svf::on_event<F>(r, op, rel, bits, frame); return; }
    const F v = word_to_f<F>((typename WordOf<F>::type)bits);
    if (rel == 8u) r.cutoff = v; else if (rel == 9u) r.q = v; else if (rel == 10u) r.gain = v; else r.ty = (u32)bits;
  }
};

// OnePoleLpf / OnePoleHpf tick -- onepole.rs:64-92.  slots: 0 last_output, 1 a0, 2 b1
template <bool HIGHPASS>
struct OnePoleT : StageDefaults {
  static constexpr int kSlots = 3;
  static constexpr u32 kMutableMask = 0b1u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = false;
  static constexpr bool kHasSeg = false;
  template <typename F> struct Regs { F y, a0, b1; };
  static constexpr u32 kParamMask = 0b110u;
  template <typename R> static __device__ __forceinline__ void take_params(R& r, const R& n, bool c) {
    r.a0 = c ? n.a0 : r.a0;
    r.b1 = c ? n.b1 : r.b1;
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long st) {
    r.y = word_to_f<F>(s[0]); r.a0 = word_to_f<F>(s[st]); r.b1 = word_to_f<F>(s[2 * st]);
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void store(const Regs<F>& r, W* s, long) { s[0] = f_to_word(r.y); }
  // OnePoleLpf/Hpf::cutoff_freq -> set_freq_lowpass (onepole.rs:35-46,135-139,172-176): device exp, tolerance only
  static __device__ __forceinline__ float exp_f(float v) { return __ocml_exp_f32(v); }
  static __device__ __forceinline__ double exp_f(double v) { return __ocml_exp_f64(v); }
  template <typename F, int P>
  static __device__ __forceinline__ void ar_set(Regs<F>& r, F v, const Ctx& c) {
    const F f = v / (F)c.sample_rate;
    r.b1 = exp_f((F)-2.0 * (F)3.14159265358979323846 * f);
    r.a0 = (F)1.0 - r.b1;
  }
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F x, const Ctx&, u32, u32&) {
    if constexpr (FMA) r.y = mad<true>(x, r.a0, r.y * r.b1);
    else r.y = x * r.a0 + r.y * r.b1;
    return HIGHPASS ? x - r.y : r.y;
  }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits, u32 /*frame*/) {
    if ((op & 0x7Fu) != EV_SET) return;
    F v = word_to_f<F>((typename WordOf<F>::type)bits);
    if (rel == 0) r.y = v; else if (rel == 1) r.a0 = v; else r.b1 = v;
  }
};
typedef OnePoleT<false> OnePoleLp;
typedef OnePoleT<true> OnePoleHp;

// x * EnvAsr / x * EnvAr -- envelopes.rs:52-81,113-128 / 205-233 and MathUGen Mul (math.rs:39-49).
// slots: 0 state, 1 t, 2 attack_rate, 3 release_rate, 4 release_scale
// state: 0 Stopped, 1 Attacking, 2 Sustaining, 3 Releasing
template <bool AR>
struct MulEnvT : StageDefaults {
  static constexpr int kSlots = 5;
  static constexpr u32 kMutableMask = 0b10011u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = true;
  static constexpr bool kNeedsBind = false;
  static constexpr bool kHasSeg = true;  // r.seg: start of the (partial) block, for mark_done
  template <typename F> struct Regs { u32 state; F t, ar, rr, scale; u32 seg; };
  // the two rates; `seg` (where the node's partial block began, set by any change that came out of a WrPreciseTiming
  // queue) moves with them
  static constexpr u32 kParamMask = 0b01100u;
  template <typename R> static __device__ __forceinline__ void take_params(R& r, const R& n, bool c) {
    r.ar = c ? n.ar : r.ar;
    r.rr = c ? n.rr : r.rr;
    r.seg = c ? n.seg : r.seg;
  }
  template <typename F> static __device__ __forceinline__ bool is_stopped(const Regs<F>& r) { return r.state == 0u; }
  template <typename F, typename W>
  static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long st) {
    r.state = (u32)s[0]; r.t = word_to_f<F>(s[st]); r.ar = word_to_f<F>(s[2 * st]);
    r.rr = word_to_f<F>(s[3 * st]); r.scale = word_to_f<F>(s[4 * st]);
    r.seg = 0;
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void store(const Regs<F>& r, W* s, long st) {
    s[0] = (W)r.state; s[st] = f_to_word(r.t); s[4 * st] = f_to_word(r.scale);
  }
  // EnvAsr/EnvAr::attack_time / ::release_time (envelopes.rs:85-110 / :236-261): the rate is a function of the new time
  // alone (the reference's "skip when unchanged" recomputes the same number), F::from(sample_rate) is the u32 as F
  template <typename F, int P>
  static __device__ __forceinline__ void ar_set(Regs<F>& r, F v, const Ctx& c) {
    const F rate = v == (F)0 ? (F)1 : (F)1 / (v * (F)c.sample_rate);
    if (P == 0) r.ar = rate; else r.rr = rate;
  }
  // One sample of the envelope itself (EnvAsr/EnvAr::next_sample, envelopes.rs:52-81 / 205-233).
  template <typename F>
  static __device__ __forceinline__ F env_next(Regs<F>& r, u32 frame, u32& done_frame) {
    const u32 st = r.state;
    const F t = r.t;
    // powi(3) = t*(t*t) (num-traits pow by squaring); then * release_scale
    const F rel_out = (t * (t * t)) * r.scale;
    F env = (F)0;
    if (st == 1u) env = t;
    if (st == 2u) env = (F)1;
    if (st == 3u) env = rel_out;
    if (st == 1u) {
      F nt = t + r.ar;
      r.t = nt;
      if (nt >= (F)1) {
        if (AR) { r.scale = (F)1; r.state = 3u; r.t = (F)1; }
        else r.state = 2u;
      }
    } else if (st == 3u) {
      F nt = t - r.rr;
      r.t = nt;
      // mark_done(i): i counts from the start of the (partial) block the envelope was handed
      // (envelopes.rs:158-162 under precise_timing.rs:104-110)
      if (nt <= (F)0) { r.state = 0u; r.t = (F)0; done_frame = frame - r.seg; }
    }
    return env;
  }
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F x, const Ctx&, u32 frame, u32& done_frame) {
    return x * env_next<F>(r, frame, done_frame);
  }
  // T envelope samples at once.  State transitions are rare (at most three per note), so the tile is
  // first evaluated branch-free under the assumption that no lane changes state inside it; every
  // sample's transition test is still evaluated, and if any lane would have changed state the tile is
  // redone with the exact per-sample state machine from the untouched registers.  Same values either way.
  template <typename F, int T>
  static __device__ __forceinline__ void env_tile(Regs<F>& r, F (&e)[T], u32 frame0, u32& done_frame) {
    const u32 st = r.state;
    const bool isA = st == 1u, isR = st == 3u;
    const bool anyA = __builtin_amdgcn_ballot_w64(isA) != 0, anyR = __builtin_amdgcn_ballot_w64(isR) != 0;
    const F konst = st == 2u ? (F)1 : (F)0;
    if (!anyA && !anyR) {  // every lane Sustaining or Stopped
#pragma unroll
      for (int j = 0; j < T; ++j) e[j] = konst;
      return;
    }
    const F step = isA ? r.ar : (isR ? -r.rr : (F)0);  // t - rr == t + (-rr) exactly
    F t = r.t;
    bool hit_hi = false, hit_lo = false;
    if (!anyR) {
#pragma unroll
      for (int j = 0; j < T; ++j) {
        e[j] = isA ? t : konst;
        t = t + step;
        hit_hi |= t >= (F)1;
      }
    } else {
      const F scale = r.scale;
#pragma unroll
      for (int j = 0; j < T; ++j) {
        const F cube = (t * (t * t)) * scale;
        e[j] = isA ? t : (isR ? cube : konst);
        t = t + step;
        hit_hi |= t >= (F)1;
        hit_lo |= t <= (F)0;
      }
    }
    const bool hit = (isA && hit_hi) || (isR && hit_lo);
    if (__builtin_amdgcn_ballot_w64(hit) == 0) {
      r.t = t;
      return;
    }
#pragma unroll
    for (int j = 0; j < T; ++j) e[j] = env_next<F>(r, frame0 + j, done_frame);
  }
  // Tiles with Releasing lanes (none Attacking) in which a lane may run out: x[j] *= max((t*(t*t))*scale, 0), t += step,
  // straight-line for every lane.  The clamp is what Releasing -> Stopped means for this sequence: t falls
  // monotonically, a sample whose t is <= 0 lies after the stop (envelopes.rs:72-78) and its cube times a non-negative
  // scale is <= 0.  t_block[k] = t of the first sample after the k-th block of eight (for mark_done's frame).
  // (Two hand-scheduled packed versions of this loop were measured slower than what the compiler makes of it:
  // 4 060 against 3 500 cycles per 64-sample tile for the envelope wave, tools/env_stamps.py.)
  template <typename F, int T>
  static __device__ __forceinline__ void release_tile_clamped(F (&x)[T], F& t_io, F step, F scale, F (&t_block)[T / 8]) {
    static_assert(T % 8 == 0, "blocks of eight samples");
    F t = t_io;
#pragma unroll
    for (int j = 0; j < T; ++j) {
      const F tj = t;
      t = t + step;
      const F e = (tj * (tj * tj)) * scale;
      x[j] = x[j] * (e > (F)0 ? e : (F)0);
      if (j % 8 == 7) t_block[j / 8] = t;
    }
    t_io = t;
  }
  // x[j] *= envelope, T samples at once.  The tile is first run under the assumption that no lane changes state
  // inside it, as straight-line code with no per-sample select: in a tile without Releasing lanes every lane's
  // envelope is its t (lanes that are not Attacking hold t = 1 or 0 with step 0); in a tile without Attacking
  // lanes it is (t*(t*t))*scale (holding lanes use t = scale = 1 or 0, which gives exactly 1 or 0).  t moves by
  // a constant step, so the sequence is monotone and a threshold is crossed inside the tile iff it is crossed at
  // its first or its last sample.  If any lane would change state (or Attacking and Releasing lanes share the
  // tile) the exact per-sample state machine runs instead, from the untouched registers.  Same values either way.
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx&, u32 frame0, u32& done_frame) {
    const u32 st = r.state;
    const bool isA = st == 1u, isR = st == 3u;
    const bool anyA = __builtin_amdgcn_ballot_w64(isA) != 0, anyR = __builtin_amdgcn_ballot_w64(isR) != 0;
    const F konst = st == 2u ? (F)1 : (F)0;
    if (!anyA && !anyR) {  // every lane Sustaining or Stopped
#pragma unroll
      for (int j = 0; j < T; ++j) x[j] = x[j] * konst;
      return;
    }
    if (!(anyA && anyR)) {
      const bool moving = isA || isR;
      const F step = isA ? r.ar : (isR ? -r.rr : (F)0);  // t - rr == t + (-rr) exactly
      const F scale = isR ? r.scale : konst;
      F t = moving ? r.t : konst;
#ifndef KNH_CLAMPED_RELEASE_MIN_TILE
#define KNH_CLAMPED_RELEASE_MIN_TILE 32  // the pipeline kernels' tiles; 8-sample tiles keep the code below
#endif
      if constexpr (T % 8 == 0 && T >= KNH_CLAMPED_RELEASE_MIN_TILE) {
        // Releasing lanes, none Attacking, and some lane may run out inside the tile: one straight-line clamped pass.
        // NaN and negative-scale cases (never produced by the setters) keep to the exact per-sample code further down.
        const bool odd = isR && !(t == t && step == step && scale >= (F)0);
        const F reach = t + (F)(T + 1) * step;
        const bool near = isR && !(reach > (F)0.0009765625 && t > (F)0.0009765625);
        if (anyR && __builtin_amdgcn_ballot_w64(odd) == 0 && __builtin_amdgcn_ballot_w64(near) != 0) {
          const F t0 = t;
          F tb[T / 8];  // t of the first sample after each block of eight
          release_tile_clamped<F, T>(x, t, step, scale, tb);
          const F t1 = t0 + step;
          const bool hit = isR && (t1 <= (F)0 || t <= (F)0);
          if (__builtin_amdgcn_ballot_w64(hit) != 0) {
            // mark_done's frame: the samples before the stop are those whose t is positive.  Blocks whose next block
            // still starts above zero count eight, blocks that start at or below zero none (the sequence falls);
            // only the block in which a stopping lane crosses is walked again sample by sample (adds only).
            u32 alive = 0;
#pragma unroll
            for (int k = 0; k < T / 8; ++k) {
              const F ts = k == 0 ? t0 : tb[k - 1];
              const bool full = !(tb[k] <= (F)0), none = ts <= (F)0;
              const bool cross = hit && !full && !none;
              alive += full ? 8u : 0u;
              if (__builtin_amdgcn_ballot_w64(cross) != 0) {
                u32 c = 0;
                F tq = ts;
#pragma unroll
                for (int i = 0; i < 8; ++i) { c += !(tq <= (F)0) ? 1u : 0u; tq = tq + step; }
                alive += cross ? c : 0u;
              }
            }
            if (hit) { r.state = 0u; r.t = (F)0; done_frame = frame0 + alive - 1u - r.seg; }
            else if (moving) r.t = t;
          } else if (moving) {
            r.t = t;
          }
          return;
        }
      }
      {
        // Most moving tiles are nowhere near a threshold: t after the tile, estimated in one step, is further from it
        // than any T roundings could carry the real sequence (|error| < T * 2^-24 * 2 << 2^-10).  Those run here with
        // no per-sample test and nothing kept but the running t; anything closer (or NaN) takes the exact code below.
        const F reach = t + (F)(T + 1) * step;
        const F margin = (F)0.0009765625;  // 2^-10
        const bool maybe = isA ? !(reach < (F)1 - margin && t < (F)1 - margin) : (isR ? !(reach > margin && t > margin) : false);
        if (__builtin_amdgcn_ballot_w64(maybe) == 0) {
          if (anyR) {
#pragma unroll
            for (int j = 0; j < T; ++j) { const F tj = t; t = t + step; x[j] = x[j] * ((tj * (tj * tj)) * scale); }
          } else {
#pragma unroll
            for (int j = 0; j < T; ++j) { const F tj = t; t = t + step; x[j] = x[j] * tj; }
          }
          if (moving) r.t = t;
          return;
        }
      }
      F tt[T];
#pragma unroll
      for (int j = 0; j < T; ++j) { tt[j] = t; t = t + step; }
      const F t1 = T > 1 ? tt[1] : t;
      const bool hit = isA ? (t1 >= (F)1 || t >= (F)1) : (isR ? (t1 <= (F)0 || t <= (F)0) : false);
      if (__builtin_amdgcn_ballot_w64(hit) == 0) {
        if (anyR) {
#pragma unroll
          for (int j = 0; j < T; ++j) x[j] = x[j] * ((tt[j] * (tt[j] * tt[j])) * scale);
        } else {
#pragma unroll
          for (int j = 0; j < T; ++j) x[j] = x[j] * tt[j];
        }
        if (moving) r.t = t;
        return;
      }
      // Some lane changes state in this tile.  The two common cases stay tile-wise, at one select per sample:
      if (anyR) {
        // a release runs out (envelopes.rs:72-78 / :224-230): t keeps falling in the speculative sequence, so the
        // samples after the last positive t are exactly the Stopped ones (env 0), and their count gives the frame
        // of mark_done.  `!(t <= 0)` keeps a NaN t on the cubic branch, as the reference's `nt <= 0` test does.
        u32 alive = 0;
#pragma unroll
        for (int j = 0; j < T; ++j) {
          const bool pos = !(tt[j] <= (F)0);
          const F e = pos ? (tt[j] * (tt[j] * tt[j])) * scale : (F)0;
          alive += pos ? 1u : 0u;
          x[j] = x[j] * e;
        }
        if (isR && hit) { r.state = 0u; r.t = (F)0; done_frame = frame0 + alive - 1u - r.seg; }
        else if (moving) r.t = t;
        return;
      }
      if (!AR) {
        // an EnvAsr attack arrives (envelopes.rs:58-64): from the first t >= 1 on the envelope is 1 (Sustaining)
        // and t stays at that first value.  Sample 0 outputs its t whatever it is (a restart can come in above 1).
        F first = t;  // the sequence rises, so the first t >= 1 is the smallest one; the tile's last step is a candidate too
        x[0] = x[0] * tt[0];
#pragma unroll
        for (int j = T - 1; j >= 1; --j) {
          const bool ge = tt[j] >= (F)1;
          first = ge ? tt[j] : first;
          x[j] = x[j] * (ge ? (F)1 : tt[j]);
        }
        if (isA && hit) { r.state = 2u; r.t = first; }
        else if (moving) r.t = t;
        return;
      }
    }
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = x[j] * env_next<F>(r, frame0 + j, done_frame);
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits, u32 frame) {
    if (op & EV_SPLIT) r.seg = frame;  // a queued WrPreciseTiming change starts a new partial block here
    op &= 0x7Fu;
    if (op == EV_ENV_ASR_RELEASE) {  // EnvAsr::t_release, envelopes.rs:113-128
      if (r.state == 1u) { r.scale = r.t; r.state = 3u; r.t = (F)1; }
      else if (r.state == 2u) { r.scale = (F)1; r.state = 3u; r.t = (F)1; }
      return;
    }
    if ((op & 0x7Fu) != EV_SET) return;
    if (rel == 0) { r.state = (u32)bits; return; }
    F v = word_to_f<F>((typename WordOf<F>::type)bits);
    if (rel == 1) r.t = v; else if (rel == 2) r.ar = v; else if (rel == 3) r.rr = v; else r.scale = v;
  }
};
typedef MulEnvT<false> MulAsr;
typedef MulEnvT<true> MulAr;

// x * Envelope (segment envelope) -- envelopes.rs:359-527 with MathUGen Mul.  Every quantity is f64 whatever F is.
// slots (one word each; doubles take two, low word first):
//   0 running  1 current_segment  2,3 current_time  4,5 from_value  6,7 dt (= time_scale * base_scale)
//   8 n_segments  9 looping  10 row of this voice in the segment table
struct MulSegEnv : StageDefaults {
  static constexpr int kSlots = 11;
  static constexpr u32 kMutableMask = 0b111111u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = true;
  static constexpr bool kNeedsBind = true;
  static constexpr bool kHasSeg = false;
  template <typename F> struct Regs {
    u32 running, cur, n_seg, looping, seg;
    double time, from, dt, dur, recip, val;
    const double* rows;
  };
  template <typename F> static __device__ __forceinline__ bool is_stopped(const Regs<F>& r) { return r.running == 0u; }
  template <typename W> static __device__ __forceinline__ double ld2(const W* s, long st, int k) {
    const u64 lo = (u32)s[(long)k * st], hi = (u32)s[(long)(k + 1) * st];
    return __builtin_bit_cast(double, lo | (hi << 32));
  }
  template <typename W> static __device__ __forceinline__ void st2(W* s, long st, int k, double v) {
    const u64 b = __builtin_bit_cast(u64, v);
    s[(long)k * st] = (W)(u32)b;
    s[(long)(k + 1) * st] = (W)(u32)(b >> 32);
  }
  template <typename F> static __device__ __forceinline__ void fetch(Regs<F>& r) {
    const double* p = r.rows + (long)r.cur * 3;
    r.dur = p[0]; r.recip = p[1]; r.val = p[2];
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long st) {
    r.running = (u32)s[0]; r.cur = (u32)s[st];
    r.time = ld2(s, st, 2); r.from = ld2(s, st, 4); r.dt = ld2(s, st, 6);
    r.n_seg = (u32)s[8 * st]; r.looping = (u32)s[9 * st];
    r.seg = 0;
    r.rows = nullptr;  // bound on first use (needs the launch context)
    r.dur = r.recip = r.val = 0.0;
    r.seg = (u32)s[10 * st];  // table row, consumed by bind()
  }
  template <typename F>
  static __device__ __forceinline__ void bind(Regs<F>& r, const Ctx& c) {
    if (r.rows == nullptr) {
      r.rows = c.seg_table + (long)r.seg * c.seg_max * 3;
      fetch<F>(r);
    }
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void store(const Regs<F>& r, W* s, long st) {
    s[0] = (W)r.running; s[st] = (W)r.cur;
    st2(s, st, 2, r.time); st2(s, st, 4, r.from);
  }
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F x, const Ctx& c, u32, u32& done_frame) {
    bind<F>(r, c);
    F out;
    if (!r.running) {
      out = (F)r.from;
    } else {
      const double t = r.time;
      if (t < r.dur) {
        out = (F)(r.from + (t * r.recip) * (r.val - r.from));
        r.time = t + r.dt;
      } else if (r.cur + 1u < r.n_seg) {
        r.from = r.val;
        out = (F)(r.from + (t * r.recip) * (r.val - r.from));
        r.time = t - r.dur + r.dt;
        r.cur += 1u;
        fetch<F>(r);
      } else {
        r.from = r.val;
        out = (F)r.from;
        if (r.looping) {
          r.cur = 0u;
          r.time = 0.0;
          fetch<F>(r);
        } else {
          r.running = 0u;
          done_frame = 0u;  // flags.mark_done(0), envelopes.rs:458
        }
      }
    }
    return x * out;
  }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits, u32) {
    op &= 0x7Fu;
    if (op == EV_SEGENV_STOP) {  // t_stop, envelopes.rs:510-522
      if (r.running && r.rows) r.from = r.from + (r.time * r.recip) * (r.val - r.from);
      r.running = 0u;
      return;
    }
    if (op != EV_SET) return;
    const u32 w = (u32)bits;
    auto lo = [](double d, u32 v) { return __builtin_bit_cast(double, (__builtin_bit_cast(u64, d) & 0xFFFFFFFF00000000ull) | (u64)v); };
    auto hi = [](double d, u32 v) { return __builtin_bit_cast(double, (__builtin_bit_cast(u64, d) & 0x00000000FFFFFFFFull) | ((u64)v << 32)); };
    switch (rel) {
      case 0: r.running = w; break;
      case 1: r.cur = w; if (r.rows) fetch<F>(r); break;
      case 2: r.time = lo(r.time, w); break;
      case 3: r.time = hi(r.time, w); break;
      case 4: r.from = lo(r.from, w); break;
      case 5: r.from = hi(r.from, w); break;
      case 6: r.dt = lo(r.dt, w); break;
      case 7: r.dt = hi(r.dt, w); break;
      default: break;
    }
  }
};

// Ring traffic as whole lines (round 4).  A voice's ring is contiguous, so the T samples of a tile are T * sizeof(F)
// contiguous bytes of it -- but with a voice per lane, a 16-byte access of the wavefront touches 64 different lines and
// uses 16 bytes of each: the memory system sees eight requests per line where one would do, and the delay chains ran at
// 2.0-2.1 TB/s of ring traffic whatever the size of the bank (tools/micro/ring_lines.hip: the same pattern without any
// arithmetic 3.2 TB/s, whole lines 4.9-5.5).  Here the wavefront moves a tile as lines: in instruction i, i = 0..7, lane l
// moves chunk (l & 7) -- 16 bytes -- of the voice in lane slot 8 i + (l >> 3): eight lanes to one voice's 128 contiguous
// bytes.  Between the two layouts (a voice's samples in its own lane's registers / a line across eight lanes) sits a tile
// in LDS private to the wavefront: 64 rows of 128 + 16 bytes (144 = 9 x 16: 16-byte accesses of 64 lanes at that stride
// spread evenly over the banks); the 16 bytes behind a row hold the voice's ring row, its two positions and the ring's
// length for the lanes that move its lines.  Longer tiles go 128 bytes per voice at a time.  A tile may cross the end of its
// ring: every chunk wraps by itself, and the one chunk of a lap that straddles the end goes sample
// by sample (rounds 1-3 sent the whole wavefront to the sample-by-sample path whenever any of its 64 voices crossed the
// end of its ring inside the tile: with 64 different delays that was every third 64-sample tile of the D3 bank).
// LDS executes a wavefront's instructions in order, so the tile needs no waits between its phases -- only the compiler
// must be kept from reordering them.
template <typename F>
struct RingLines {
  static constexpr int kLine = 128;
  static constexpr int TS = kLine / (int)sizeof(F);  // samples of a voice per line
  static constexpr int kRow = kLine + 16;
  static constexpr int kTileBytes = 64 * kRow;        // 9 216 per wavefront
  static constexpr int VW = 16 / (int)sizeof(F);
  typedef typename WordOf<F>::type W;
  typedef u32 U4 __attribute__((ext_vector_type(4)));
  typedef u32 U4u __attribute__((ext_vector_type(4), aligned(4)));  // (a ring position is sample-aligned, no more)
  typedef __attribute__((address_space(3))) char* tile_t;
  typedef __attribute__((address_space(3))) U4* lds_u4;
  typedef __attribute__((address_space(1))) U4u* glb_u4;
  typedef __attribute__((address_space(1))) W* glb_w;
  struct Lines { U4 v[8]; };
  // The eight voices whose lines this lane moves, for the tile at hand: ring address, positions, length.
  struct Owned { u64 ring[8]; u32 wp[8], rp[8], len[8]; };
  static __device__ __forceinline__ void fence() { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
  static __device__ __forceinline__ u32 wrapped(u32 pos, u32 len) { return pos - len < pos ? pos - len : pos; }  // pos < 2 len (pos - len wraps to a huge number below len)
  // Every lane posts its voice's ring row, positions (samples) and ring length; then reads those of the voices it moves
  // lines for.  A lane without a voice posts the bank's sink row (one spare ring behind the last voice's: Ctx::ring_sink_row),
  // so that no access below needs to ask whether its voice exists.
  static __device__ __forceinline__ void exchange(tile_t tile, int lane, bool live, u32 row, u32 wp, u32 rp, u32 len, const Ctx& c, Owned& o) {
    U4 h;
    h[0] = live ? row : c.ring_sink_row; h[1] = live ? wp : 0u; h[2] = live ? rp : 0u; h[3] = live ? len : 0x7FFFFFFFu;
    *reinterpret_cast<lds_u4>(tile + lane * kRow + kLine) = h;
    fence();
    U4 g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = *reinterpret_cast<lds_u4>(tile + (8 * i + (lane >> 3)) * kRow + kLine);
    const u64 row_bytes = (u64)c.delay_stride * sizeof(F);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const u32 r0 = g[i][0], r1 = g[i][1], r2 = g[i][2], r3 = g[i][3];
      o.ring[i] = reinterpret_cast<u64>(c.delay_ring) + (u64)r0 * row_bytes;
      o.wp[i] = r1 + (u32)((lane & 7) * VW);  // (this lane's chunk of the line)
      o.rp[i] = r2 + (u32)((lane & 7) * VW);
      o.len[i] = r3;
    }
    fence();
  }
  // the lines at read position + sample_offset, requested (the loads are in flight when this returns)
  static __device__ __forceinline__ void load(const Owned& o, u32 sample_offset, Lines& l) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const u32 len = o.len[i], pos = wrapped(o.rp[i] + sample_offset, len);
      U4 v;
      if (pos + (u32)VW <= len) {
        v = *reinterpret_cast<glb_u4>(o.ring[i] + (u64)pos * sizeof(F));
      } else {  // the chunk that straddles the end of the ring: sample by sample
#pragma unroll
        for (int e = 0; e < VW; ++e) {
          const W q = reinterpret_cast<glb_w>(o.ring[i])[wrapped(pos + (u32)e, len)];
          if constexpr (sizeof(F) == 4) v[e] = (u32)q;
          else { v[2 * e] = (u32)q; v[2 * e + 1] = (u32)((u64)q >> 32); }
        }
      }
      l.v[i] = v;
    }
  }
  // lines -> this lane's voice's TS samples
  static __device__ __forceinline__ void to_samples(tile_t tile, int lane, const Lines& l, F* y) {
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<lds_u4>(tile + (8 * i + (lane >> 3)) * kRow + (lane & 7) * 16) = l.v[i];
    fence();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const U4 q = *reinterpret_cast<lds_u4>(tile + lane * kRow + j * 16);
#pragma unroll
      for (int k = 0; k < VW; ++k) {
        // (a copy of the element first: __builtin_bit_cast of a vector element reads element 0, voice_chain.hpp res_get)
        if constexpr (sizeof(F) == 4) { const u32 e = q[k]; y[j * VW + k] = __builtin_bit_cast(F, e); }
        else { const u32 lo = q[2 * k], hi = q[2 * k + 1]; const u64 e = (u64)lo | ((u64)hi << 32); y[j * VW + k] = __builtin_bit_cast(F, e); }
      }
    }
    fence();
  }
  // this lane's voice's TS samples -> lines, stored at write position + sample_offset
  static __device__ __forceinline__ void store(tile_t tile, int lane, const Owned& o, const F* x, u32 sample_offset) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      U4 q;
#pragma unroll
      for (int k = 0; k < VW; ++k) {
        const F e = x[j * VW + k];
        if constexpr (sizeof(F) == 4) q[k] = __builtin_bit_cast(u32, e);
        else { const u64 b = __builtin_bit_cast(u64, e); q[2 * k] = (u32)b; q[2 * k + 1] = (u32)(b >> 32); }
      }
      *reinterpret_cast<lds_u4>(tile + lane * kRow + j * 16) = q;
    }
    fence();
    U4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<lds_u4>(tile + (8 * i + (lane >> 3)) * kRow + (lane & 7) * 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const u32 len = o.len[i], pos = wrapped(o.wp[i] + sample_offset, len);
      if (pos + (u32)VW <= len) {
        *reinterpret_cast<glb_u4>(o.ring[i] + (u64)pos * sizeof(F)) = v[i];
      } else {
#pragma unroll
        for (int e = 0; e < VW; ++e) {
          W q;
          if constexpr (sizeof(F) == 4) { const u32 t = v[i][e]; q = (W)t; }
          else { const u32 lo = v[i][2 * e], hi = v[i][2 * e + 1]; q = (W)((u64)lo | ((u64)hi << 32)); }
          reinterpret_cast<glb_w>(o.ring[i])[wrapped(pos + (u32)e, len)] = q;
        }
      }
    }
    fence();
  }
};

// SampleDelay -- delay.rs:14-50.  process: buffer[wp] = x; out = buffer[(wp + len - delay) % len]; wp = (wp + 1) % len.
// Each voice's ring is a contiguous run of HBM ([voice][delay_stride]); slots: 0 write_position, 1 off = len - delay_samples
// (0..len), 2 len, 3 the voice's ring row.  A tile whose reads cannot meet its own writes (delay >= T) and that does not
// cross the end of the ring moves its T samples with 16-byte loads, then 16-byte stores; any other tile runs sample by
// sample in the reference's order (store, then load).
struct SampleDelay : StageDefaults {
  static constexpr int kSlots = 4;
  static constexpr u32 kMutableMask = 0b1u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = true;
  static constexpr bool kHasSeg = false;
  static constexpr bool kUsesRing = true;
  static constexpr int kPrefetch = 32;  // largest tile that is read one tile ahead
  static constexpr int kLinesAhead = 2; // ... as whole lines (RingLines): 256 bytes per voice
  template <typename F> struct Regs {
    u32 wp, off, len, row;
    F* ring;
    u32 pre_pos;       // ring position the tile in `pre` / `lines` was read from, 0xFFFFFFFF: none
    F pre[kPrefetch];  // the next tile's samples, requested while this tile is being processed
    typename RingLines<F>::Lines lines[kLinesAhead];  // the same where the wavefront moves its rings as lines: this lane's share of them
  };
  template <typename F, typename W>
  static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long st) {
    r.wp = (u32)s[0]; r.off = (u32)s[st]; r.len = (u32)s[2 * st]; r.row = (u32)s[3 * st];
    r.ring = nullptr;
    r.pre_pos = 0xFFFFFFFFu;
  }
  template <typename F>
  static __device__ __forceinline__ void bind(Regs<F>& r, const Ctx& c) {
    if (r.ring == nullptr) r.ring = reinterpret_cast<F*>(c.delay_ring) + (long)r.row * c.delay_stride;
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void store(const Regs<F>& r, W* s, long) { s[0] = (W)r.wp; }
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F x, const Ctx& c, u32, u32&) {
    bind<F>(r, c);
    if (r.len == 0u) return x;  // a lane past the last voice (its state words are zero): no memory access
    r.ring[r.wp] = x;
    u32 rp = r.wp + r.off;  // < 2 * len
    if (rp >= r.len) rp -= r.len;
    // the sample just stored is forwarded from the register (delay 0 or len); anything else comes from memory
    const F y = rp == r.wp ? x : r.ring[rp];
    r.wp = r.wp + 1u == r.len ? 0u : r.wp + 1u;
    return y;
  }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
    bind<F>(r, c);
    constexpr int VW = 16 / (int)sizeof(F);
    typedef F Vec __attribute__((ext_vector_type(VW), aligned(sizeof(F))));
    u32 rp = r.wp + r.off;
    if (rp >= r.len) rp -= r.len;
    const bool dead = r.len == 0u;  // a lane past the last voice: takes part in nothing
    {
      typedef RingLines<F> RL;
      if constexpr (T % RL::TS == 0 && T / RL::TS <= kLinesAhead) {
        if (c.ring_tile != nullptr) {  // (known per kernel: the other vector path is not in its code then)
          // The tile as whole lines (RingLines): wherever every voice of the wavefront has a delay of a tile or more (the
          // reads cannot meet this tile's writes; off == 0 is a delay of 0 or of the whole ring: tick() forwards that
          // sample) -- either pointer may cross the end of its ring.  The next tile's lines are requested before this
          // tile's are stored and wait in registers through everything the other stages do in between, if no voice's delay
          // is shorter than two tiles (those loads cannot meet this tile's stores then).
          const bool lines_ok = dead || (r.len >= (u32)T && r.off != 0u && r.off <= r.len - (u32)T);
          if (__builtin_amdgcn_ballot_w64(!lines_ok) == 0) {
            constexpr int NS = T / RL::TS;
            const int lane = (int)(threadIdx.x & 63u);
            u32 np = rp + (u32)T;
            if (np >= r.len) np -= r.len;
            const bool have = !dead && r.pre_pos == rp;
            const bool ahead = !dead && r.len >= 2u * (u32)T && r.off <= r.len - 2u * (u32)T;
            const bool all_have = __builtin_amdgcn_ballot_w64(!(have || dead)) == 0;
            const bool all_ahead = __builtin_amdgcn_ballot_w64(!(ahead || dead)) == 0;
            typename RL::Owned own;
            RL::exchange(c.ring_tile, lane, !dead, r.row, r.wp, rp, r.len, c, own);
            F y[T];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
              if (all_have) {
                RL::to_samples(c.ring_tile, lane, r.lines[s], &y[s * RL::TS]);
              } else {
                typename RL::Lines l;
                RL::load(own, (u32)(s * RL::TS), l);
                RL::to_samples(c.ring_tile, lane, l, &y[s * RL::TS]);
              }
            }
            if (all_ahead) {
#pragma unroll
              for (int s = 0; s < NS; ++s) RL::load(own, (u32)(T + s * RL::TS), r.lines[s]);
            }
            r.pre_pos = all_ahead && !dead ? np : 0xFFFFFFFFu;
#pragma unroll
            for (int s = 0; s < NS; ++s) RL::store(c.ring_tile, lane, own, &x[s * RL::TS], (u32)(s * RL::TS));
            if (!dead) {
#pragma unroll
              for (int j = 0; j < T; ++j) x[j] = y[j];
              r.wp += (u32)T;
              if (r.wp >= r.len) r.wp -= r.len;
            }
            return;
          }
          r.pre_pos = 0xFFFFFFFFu;
#pragma unroll
          for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
          return;
        }
      }
    }
    // delay >= T  <=>  off <= len - T;  neither the T stores nor the T loads may cross the end of the ring
    // (off == 0 is a delay of 0 or of the whole ring: the sample just stored comes straight back -- tick() forwards it)
    const bool vec_ok = dead || (r.len >= (u32)T && r.off != 0u && r.off <= r.len - (u32)T && r.wp <= r.len - (u32)T && rp <= r.len - (u32)T);
    if (__builtin_amdgcn_ballot_w64(!vec_ok) == 0) {
      // Large tiles (the pipelined kernels, one wavefront per SIMD: nothing else hides HBM latency) are read one
      // tile ahead: this tile comes out of registers filled during the previous one, and the next tile's loads go
      // out before this tile's stores.  They cannot meet those stores when delay >= 2T (off <= len - 2T).
      constexpr bool kAhead = T >= 16 && T <= kPrefetch;
      bool have = false, ahead = false;
      u32 np = 0;
      if constexpr (kAhead) {
        have = !dead && r.pre_pos == rp;
        np = rp + (u32)T;
        ahead = !dead && r.len >= 2u * (u32)T && r.off <= r.len - 2u * (u32)T && np <= r.len - (u32)T && r.wp + (u32)T <= r.len - (u32)T;
      }
      const bool all_have = kAhead && __builtin_amdgcn_ballot_w64(!(have || dead)) == 0;
      if (!dead) {
        F y[T];
        if (all_have) {
          if constexpr (kAhead) {
#pragma unroll
            for (int j = 0; j < T; ++j) y[j] = r.pre[j];
          }
        } else {
          const Vec* src = reinterpret_cast<const Vec*>(r.ring + rp);
#pragma unroll
          for (int j = 0; j < T / VW; ++j) {
            const Vec v = src[j];
#pragma unroll
            for (int k = 0; k < VW; ++k) y[j * VW + k] = v[k];
          }
        }
        if constexpr (kAhead) {
          if (ahead) {
            const Vec* nsrc = reinterpret_cast<const Vec*>(r.ring + np);
#pragma unroll
            for (int j = 0; j < T / VW; ++j) {
              const Vec v = nsrc[j];
#pragma unroll
              for (int k = 0; k < VW; ++k) r.pre[j * VW + k] = v[k];
            }
          }
          r.pre_pos = ahead ? np : 0xFFFFFFFFu;
        } else {
          r.pre_pos = 0xFFFFFFFFu;
        }
        Vec* dst = reinterpret_cast<Vec*>(r.ring + r.wp);
#pragma unroll
        for (int j = 0; j < T / VW; ++j) {
          Vec v;
#pragma unroll
          for (int k = 0; k < VW; ++k) v[k] = x[j * VW + k];
          dst[j] = v;
        }
#pragma unroll
        for (int j = 0; j < T; ++j) x[j] = y[j];
        r.wp = r.wp + (u32)T == r.len ? 0u : r.wp + (u32)T;
      }
      return;
    }
    r.pre_pos = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits, u32) {
    if ((op & 0x7Fu) != EV_SET) return;
    if (rel == 1) r.off = (u32)bits;
    else if (rel == 0) r.wp = (u32)bits;
  }
};

// AllpassDelay -- delay.rs:93-206: out = allpass(buffer[read]); read += 1; buffer[write] = x; write += 1 (both modulo the
// ring), with a first-order allpass interpolator (delay.rs:53-90: out = coeff * (in - prev_out) + prev_in) for the
// fractional part of the delay.  Ring layout as SampleDelay.  slots: 0 write_frame  1 read_frame  2 ring length
// 3 ring row  4 coeff  5 prev_input  6 prev_output  (7 feedback).  delay_time arrives as a coeff patch plus
// EV_ALLPASS_DELAY carrying the whole number of frames: read_frame is derived from the live write_frame
// (set_delay_in_frames, :160-174).
// FB = true: AllpassFeedbackDelay, the Schroeder allpass around it (delay.rs:210-306): d = read(); w = d * feedback + x;
// write(w); out = d - feedback * w.
template <bool FB>
struct AllpassDelayT : StageDefaults {
  static constexpr int kSlots = FB ? 8 : 7;
  static constexpr bool kUsesRing = true;
  static constexpr u32 kMutableMask = 0b1100011u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = true;
  static constexpr bool kHasSeg = false;
  template <typename F> struct Regs { u32 wp, rp, len, row; F coeff, pin, pout, fb; F* ring; };
  template <typename F, typename W>
  static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long st) {
    r.wp = (u32)s[0]; r.rp = (u32)s[st]; r.len = (u32)s[2 * st]; r.row = (u32)s[3 * st];
    r.coeff = word_to_f<F>(s[4 * st]); r.pin = word_to_f<F>(s[5 * st]); r.pout = word_to_f<F>(s[6 * st]);
    r.fb = FB ? word_to_f<F>(s[7 * st]) : (F)0;
    r.ring = nullptr;
  }
  template <typename F>
  static __device__ __forceinline__ void bind(Regs<F>& r, const Ctx& c) {
    if (r.ring == nullptr) r.ring = reinterpret_cast<F*>(c.delay_ring) + (long)r.row * c.delay_stride;
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void store(const Regs<F>& r, W* s, long st) {
    s[0] = (W)r.wp; s[st] = (W)r.rp; s[5 * st] = f_to_word(r.pin); s[6 * st] = f_to_word(r.pout);
  }
  template <typename F> static __device__ __forceinline__ F allpass(Regs<F>& r, F in) {  // :78-83
    const F out = r.coeff * (in - r.pout) + r.pin;
    r.pout = out;
    r.pin = in;
    return out;
  }
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F x, const Ctx& c, u32, u32&) {
    bind<F>(r, c);
    if (r.len == 0u) return x;  // a lane past the last voice
    const F y = allpass<F>(r, r.ring[r.rp]);
    r.rp = r.rp + 1u == r.len ? 0u : r.rp + 1u;
    const F w = FB ? y * r.fb + x : x;  // process_sample, :263-269
    r.ring[r.wp] = w;
    r.wp = r.wp + 1u == r.len ? 0u : r.wp + 1u;
    return FB ? y - r.fb * w : y;
  }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
    bind<F>(r, c);
    constexpr int VW = 16 / (int)sizeof(F);
    typedef F Vec __attribute__((ext_vector_type(VW), aligned(sizeof(F))));
    const bool dead = r.len == 0u;
    // a read meets a store of the same tile only when the write pointer is 1 .. T-1 frames ahead of the read pointer
    const u32 ahead = r.wp >= r.rp ? r.wp - r.rp : r.wp + r.len - r.rp;
    {
      typedef RingLines<F> RL;
      if constexpr (T % RL::TS == 0) {
        if (c.ring_tile != nullptr) {  // the tile as whole lines (RingLines, above): either pointer may cross the end of the ring
          const bool lines_ok = dead || (r.len >= (u32)T && (ahead == 0u || ahead >= (u32)T));
          if (__builtin_amdgcn_ballot_w64(!lines_ok) == 0) {
            constexpr int NS = T / RL::TS;
            const int lane = (int)(threadIdx.x & 63u);
            typename RL::Owned own;
            RL::exchange(c.ring_tile, lane, !dead, r.row, r.wp, r.rp, r.len, c, own);
            F y[T];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
              typename RL::Lines l;
              RL::load(own, (u32)(s * RL::TS), l);
              RL::to_samples(c.ring_tile, lane, l, &y[s * RL::TS]);
            }
            F wr[T];  // what goes into the ring: the input, or the input plus the fed-back delayed signal
#pragma unroll
            for (int j = 0; j < T; ++j) wr[j] = x[j];
            if (!dead) {
#pragma unroll
              for (int j = 0; j < T; ++j) {
                const F d = allpass<F>(r, y[j]);
                wr[j] = FB ? d * r.fb + x[j] : x[j];
                x[j] = FB ? d - r.fb * wr[j] : d;
              }
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) RL::store(c.ring_tile, lane, own, &wr[s * RL::TS], (u32)(s * RL::TS));
            if (!dead) {
              r.rp += (u32)T;
              if (r.rp >= r.len) r.rp -= r.len;
              r.wp += (u32)T;
              if (r.wp >= r.len) r.wp -= r.len;
            }
            return;
          }
#pragma unroll
          for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
          return;
        }
      }
    }
    const bool vec_ok = dead || (r.len >= (u32)T && (ahead == 0u || ahead >= (u32)T) && r.wp <= r.len - (u32)T && r.rp <= r.len - (u32)T);
    if (__builtin_amdgcn_ballot_w64(!vec_ok) == 0) {
      if (!dead) {
        F y[T];
        const Vec* src = reinterpret_cast<const Vec*>(r.ring + r.rp);
#pragma unroll
        for (int j = 0; j < T / VW; ++j) {
          const Vec v = src[j];
#pragma unroll
          for (int k = 0; k < VW; ++k) y[j * VW + k] = v[k];
        }
        F wr[T];  // what goes into the ring: the input, or the input plus the fed-back delayed signal
#pragma unroll
        for (int j = 0; j < T; ++j) {
          const F d = allpass<F>(r, y[j]);
          wr[j] = FB ? d * r.fb + x[j] : x[j];
          x[j] = FB ? d - r.fb * wr[j] : d;
        }
        Vec* dst = reinterpret_cast<Vec*>(r.ring + r.wp);
#pragma unroll
        for (int j = 0; j < T / VW; ++j) {
          Vec v;
#pragma unroll
          for (int k = 0; k < VW; ++k) v[k] = wr[j * VW + k];
          dst[j] = v;
        }
        r.rp = r.rp + (u32)T == r.len ? 0u : r.rp + (u32)T;
        r.wp = r.wp + (u32)T == r.len ? 0u : r.wp + (u32)T;
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits, u32) {
    op &= 0x7Fu;
    if (op == EV_ALLPASS_DELAY) {  // set_delay_in_frames, :168-172
      const u32 nf = (u32)bits;
      r.rp = r.wp >= nf ? r.wp - nf : r.len - nf + r.wp;
      return;
    }
    if (op != EV_SET) return;
    if (rel == 4) r.coeff = word_to_f<F>((typename WordOf<F>::type)bits);
    else if (rel == 7) r.fb = word_to_f<F>((typename WordOf<F>::type)bits);
    else if (rel == 0) r.wp = (u32)bits;
    else if (rel == 1) r.rp = (u32)bits;
  }
};
typedef AllpassDelayT<false> AllpassDelay;
typedef AllpassDelayT<true> AllpassFbDelay;

// x (op) value: Constant + MathUGen (util.rs:61-63, math.rs:22-85) and WrMul/WrAdd/WrSub
// (wrappers_core/math.rs:62-67).  slot 0: value.  OP: 0 mul, 1 add, 2 sub, 3 div
template <int OP>
struct ValT : StageDefaults {
  static constexpr int kSlots = 1;
  static constexpr u32 kMutableMask = 0u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = false;
  static constexpr bool kHasSeg = false;
  template <typename F> struct Regs { F v; };
  static constexpr u32 kParamMask = 0b1u;
  template <typename R> static __device__ __forceinline__ void take_params(R& r, const R& n, bool c) { r.v = c ? n.v : r.v; }
  template <typename F, typename W>
  static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long) { r.v = word_to_f<F>(s[0]); }
  template <typename F, typename W>
  static __device__ __forceinline__ void store(const Regs<F>&, W*, long) {}
  // Constant::value (util.rs:47-50) / WrMul's "wr_mul" (wrappers_core/math.rs:92-98): value = F::new(v)
  template <typename F, int P>
  static __device__ __forceinline__ void ar_set(Regs<F>& r, F v, const Ctx&) { r.v = v; }
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F x, const Ctx&, u32, u32&) {
    if (OP == 0) return x * r.v;
    if (OP == 1) return x + r.v;
    if (OP == 2) return x - r.v;
    if (OP == 3) return x / r.v;
    if (OP == 4) return r.v - x;  // WrVSub, wrappers_core/math.rs:297-299
    if (OP == 5) return r.v / x;  // WrVDiv, wrappers_core/math.rs:454-456
    return dev_pow(x, r.v);       // WrPowf / MathUGen Pow: device libm, tolerance only
  }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
    if constexpr (sizeof(F) == 4 && OP <= 2 && T % 2 == 0) {
      // two neighbouring samples to a packed instruction (v_pk_mul_f32 / v_pk_add_f32: the same roundings)
      typedef float f2 __attribute__((ext_vector_type(2)));
      const f2 v = {r.v, r.v};
#pragma unroll
      for (int j = 0; j < T; j += 2) {
        f2 p = {x[j], x[j + 1]};
        p = OP == 0 ? p * v : (OP == 1 ? p + v : p - v);
        x[j] = p.x; x[j + 1] = p.y;
      }
    } else {
#pragma unroll
      for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
    }
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32, u64 bits, u32 /*frame*/) {
    if ((op & 0x7Fu) == EV_SET) r.v = word_to_f<F>((typename WordOf<F>::type)bits);
  }
};
typedef ValT<0> MulVal;
typedef ValT<1> AddVal;
typedef ValT<2> SubVal;
typedef ValT<3> DivVal;
typedef ValT<4> VSubVal;
typedef ValT<5> VDivVal;
typedef ValT<6> PowVal;

// x.powi(n): WrPowi (wrappers_core/math.rs:587-661).  f32::powi / f64::powi with a run-time exponent lower to
// compiler-builtins' __powisf2 / __powidf2: multiply by squaring, reciprocal at the end for n < 0.  slot 0: n (i32)
struct PowiVal : StageDefaults {
  static constexpr int kSlots = 1;
  static constexpr u32 kMutableMask = 0u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = false;
  static constexpr bool kHasSeg = false;
  template <typename F> struct Regs { int n; };
  static constexpr u32 kParamMask = 0b1u;
  template <typename R> static __device__ __forceinline__ void take_params(R& r, const R& n, bool c) { r.n = c ? n.n : r.n; }
  template <typename F, typename W>
  static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long) { r.n = (int)(u32)s[0]; }
  template <typename F, typename W>
  static __device__ __forceinline__ void store(const Regs<F>&, W*, long) {}
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>& r, F x, const Ctx&, u32, u32&) {
    u32 p = r.n < 0 ? 0u - (u32)r.n : (u32)r.n;
    F a = x, m = (F)1;
    for (;;) {
      if (p & 1u) m *= a;
      p >>= 1;
      if (p == 0u) break;
      a *= a;
    }
    return r.n < 0 ? (F)1 / m : m;
  }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = tick<F, FMA>(r, x[j], c, frame0 + j, done_frame);
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32, u64 bits, u32) {
    if ((op & 0x7Fu) == EV_SET) r.n = (int)(u32)bits;
  }
};

// Pan2 -- knaster_core_dsp/src/ugens/pan.rs:12-37: [x * left_gain, x * right_gain] with the two gains a function of the
// `pan` parameter alone (fastapprox::fast::cos / sin of pan' * pi/2, recomputed by the reference every sample from the
// stored pan: the same two numbers each time), so they are computed on the host (bank.hip) and live here as two words.
// The stage is the END of a chain: the running signal passes through unchanged and the two products are formed where a
// voice's signal leaves the chain, in the per-wave fold (left = ((x0*l0 + x1*l1) + x2*l2) + ..., the reference's
// Pan2 outputs summed by its chain of Add nodes, one chain per output channel).  Gains change at block boundaries
// only (the stage cannot be wrapped in WrPreciseTiming here), which are tile boundaries in every kernel form.
// slots: 0 left_gain  1 right_gain
struct Pan2 : StageDefaults {
  static constexpr int kSlots = 2;
  static constexpr u32 kMutableMask = 0u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = false;
  static constexpr bool kHasSeg = false;
  template <typename F> struct Regs { F l, r; };
  template <typename F, typename W>
  static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long st) { r.l = word_to_f<F>(s[0]); r.r = word_to_f<F>(s[st]); }
  template <typename F, typename W>
  static __device__ __forceinline__ void store(const Regs<F>&, W*, long) {}
  template <typename F, bool FMA>
  static __device__ __forceinline__ F tick(Regs<F>&, F x, const Ctx&, u32, u32&) { return x; }
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(Regs<F>&, F (&)[T], const Ctx&, u32, u32&) {}
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits, u32) {
    if ((op & 0x7Fu) != EV_SET) return;
    const F v = word_to_f<F>((typename WordOf<F>::type)bits);
    if (rel == 0) r.l = v; else r.r = v;
  }
};
template <typename S> struct IsPan { static constexpr bool value = false; };
template <> struct IsPan<Pan2> { static constexpr bool value = true; };

}  // namespace knh_dev
