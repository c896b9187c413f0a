// knaster_oracle.hpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
//
// A C++17 restatement of the Rust reference's per-sample / per-block UGen hot
// path (ErikNatanael/knaster), written from the sources read as text.  Only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it;
// nothing under knaster_amd/ links, includes or calls it.
//
// PINNING STATUS
//   * Pinned by the reference's own known-answer tests (ported in
//     oracle/oracle_kat.cpp, run by tests/test_oracle_kat.py): wrapper
//     arithmetic, WrPreciseTiming sequences, MathUGen, graph plumbing,
//     additive outputs, disconnect, bench asserts, Seconds conversions,
//     the implement_a_gen sine check.
//   * PARITY UNPINNED for Pan2's gains (fastapprox::fast::cos/sin, an un-vendored
//     crate, restated from its published algorithm) and for the noise UGens'
//     generator (fastrand, likewise).
//   * PARITY UNPINNED for the waveform-level output of SinWt, SinNumeric,
//     SvfFilter, OnePole*, EnvAsr, EnvAr: no reference test or fixture pins
//     those numbers and no Rust toolchain exists in the build image, so those
//     are correct "by construction of the restatement" only.
//
// All citations are file:line under /root/reference/.
// Arithmetic rules: built with -ffp-contract=off -fno-fast-math so every
// a*b+c is two roundings, exactly as rustc emits it.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <string>
#include <unordered_set>
#include <utility>
#include <vector>

namespace kno {

// ---------------------------------------------------------------------------
// Float helpers -- knaster_primitives/src/float.rs:11-56,97-175
// F::new(v) is a plain `as` cast through to_f32()/to_f64() (round to nearest).
// ---------------------------------------------------------------------------
using PFloat = double;  // knaster_primitives/src/parameters.rs:6

template <typename F, typename V>
inline F fnew(V v) {
  return static_cast<F>(v);
}
template <typename F>
struct FloatConst;
template <>
struct FloatConst<float> {
  static constexpr float PI = 3.14159265358979323846f;   // f32::consts::PI
  static constexpr float TAU = 6.28318530717958647692f;  // f32::consts::TAU
};
template <>
struct FloatConst<double> {
  static constexpr double PI = 3.14159265358979323846;
  static constexpr double TAU = 6.28318530717958647692;
};

// Rust `as u32` from a float saturates: NaN -> 0, <0 -> 0, >MAX -> MAX.
inline uint32_t sat_u32(double v) {
  if (!(v > 0.0)) return 0u;  // also NaN
  if (v >= 4294967295.0) return 0xFFFFFFFFu;
  return static_cast<uint32_t>(v);
}

// ---------------------------------------------------------------------------
// XOrShift32Rng -- knaster_core_dsp/src/dsp/xorrng.rs:9-50
// ---------------------------------------------------------------------------
struct XOrShift32Rng {
  uint32_t fpd;
  explicit XOrShift32Rng(uint32_t seed = 17) : fpd(seed == 0 ? 17u : seed) {}
  uint32_t gen_u32() {
    fpd ^= fpd << 13;
    fpd ^= fpd >> 17;
    fpd ^= fpd << 5;
    return fpd;
  }
  float gen_f32() { return static_cast<float>(gen_u32()) / static_cast<float>(0xFFFFFFFFu); }
  double gen_f64() { return static_cast<double>(gen_u32()) / static_cast<double>(0xFFFFFFFFu); }
};

// ---------------------------------------------------------------------------
// Seconds -- knaster_primitives/src/time.rs:11-157
// ---------------------------------------------------------------------------
constexpr uint32_t SUBSECOND_TESIMALS_PER_SECOND = 282240000u;
struct Seconds {
  uint32_t seconds = 0;
  uint32_t subsecond_tesimals = 0;
  static Seconds zero() { return Seconds{0, 0}; }
  static Seconds from_secs_f64(double s) {  // time.rs:58-63
    Seconds r;
    r.seconds = sat_u32(std::floor(s));
    double fract = s - std::trunc(s);  // f64::fract
    r.subsecond_tesimals = sat_u32(fract * static_cast<double>(SUBSECOND_TESIMALS_PER_SECOND));
    return r;
  }
  static Seconds from_samples(uint64_t samples, uint64_t sample_rate) {  // time.rs:76-84
    Seconds r;
    r.seconds = static_cast<uint32_t>(samples / sample_rate);
    r.subsecond_tesimals = static_cast<uint32_t>(
        (samples % sample_rate) * static_cast<uint64_t>(SUBSECOND_TESIMALS_PER_SECOND) / sample_rate);
    return r;
  }
  static Seconds from_subsample_tesimals_u64(uint64_t t) {  // time.rs:48-53
    Seconds r;
    r.seconds = static_cast<uint32_t>(t / SUBSECOND_TESIMALS_PER_SECOND);
    r.subsecond_tesimals = static_cast<uint32_t>(t - static_cast<uint64_t>(r.seconds) * SUBSECOND_TESIMALS_PER_SECOND);
    return r;
  }
  uint64_t to_subsample_tesimals_u64() const {  // time.rs:55-57
    return static_cast<uint64_t>(seconds) * SUBSECOND_TESIMALS_PER_SECOND + subsecond_tesimals;
  }
  // From<core::time::Duration> (time.rs:135-143); Duration::from_secs_f64 truncates to whole nanoseconds
  static Seconds from_duration(uint64_t secs, uint32_t subsec_nanos) {
    const double conversion_factor = static_cast<double>(SUBSECOND_TESIMALS_PER_SECOND) / 1000000000.0;
    Seconds r;
    r.seconds = static_cast<uint32_t>(secs);
    r.subsecond_tesimals = sat_u32(static_cast<double>(subsec_nanos) * conversion_factor);
    return r;
  }
  double to_samples_f64(double sample_rate) const {  // time.rs:92-96
    return static_cast<double>(seconds) * sample_rate +
           (static_cast<double>(subsecond_tesimals) * sample_rate) / static_cast<double>(SUBSECOND_TESIMALS_PER_SECOND);
  }
  double to_secs_f64() const {  // time.rs:71-74
    return static_cast<double>(seconds) + (static_cast<double>(subsecond_tesimals) / static_cast<double>(SUBSECOND_TESIMALS_PER_SECOND));
  }
  uint64_t to_samples(uint64_t sample_rate) const {  // time.rs:86-90
    return static_cast<uint64_t>(seconds) * sample_rate +
           (static_cast<uint64_t>(subsecond_tesimals) * sample_rate) /
               static_cast<uint64_t>(SUBSECOND_TESIMALS_PER_SECOND);
  }
  bool operator==(const Seconds& o) const {
    return seconds == o.seconds && subsecond_tesimals == o.subsecond_tesimals;
  }
  bool le(const Seconds& o) const {  // Ord, time.rs:149-157
    if (seconds == o.seconds) return subsecond_tesimals <= o.subsecond_tesimals;
    return seconds < o.seconds;
  }
  Seconds saturating_sub(const Seconds& rhs) const {  // time.rs:119-133
    if (le(rhs)) return zero();
    if (subsecond_tesimals >= rhs.subsecond_tesimals)
      return Seconds{seconds - rhs.seconds, subsecond_tesimals - rhs.subsecond_tesimals};
    return Seconds{seconds - rhs.seconds - 1,
                   SUBSECOND_TESIMALS_PER_SECOND - (rhs.subsecond_tesimals - subsecond_tesimals)};
  }
  Seconds add(const Seconds& rhs) const {  // time.rs:158-171
    uint32_t s = seconds + rhs.seconds;
    uint32_t t = subsecond_tesimals + rhs.subsecond_tesimals;
    while (t >= SUBSECOND_TESIMALS_PER_SECOND) {
      s += 1;
      t -= SUBSECOND_TESIMALS_PER_SECOND;
    }
    return Seconds{s, t};
  }
};

// ---------------------------------------------------------------------------
// AudioCtx / BlockMetadata / UGenFlags -- knaster_core/src/ugen.rs:8-219
// ---------------------------------------------------------------------------
struct BlockMetadata {
  size_t block_start_offset = 0;
  size_t frames_to_process = 0;
  uint64_t frame_clock = 0;
  BlockMetadata make_partial(size_t start_offset, size_t length) const {  // ugen.rs:87-93
    BlockMetadata b;
    b.block_start_offset = block_start_offset + start_offset;
    b.frames_to_process = length;
    b.frame_clock = frame_clock + start_offset;
    return b;
  }
};
struct AudioCtx {
  uint32_t sample_rate_;
  size_t block_size_;
  BlockMetadata block;
  std::vector<std::string> log;  // stands in for the rt_log! ring buffer
  AudioCtx(uint32_t sr, size_t bs) : sample_rate_(sr), block_size_(bs) {
    block.frames_to_process = bs;
  }
  uint32_t sample_rate() const { return sample_rate_; }
  size_t block_size() const { return block_size_; }
  size_t frames_to_process() const { return block.frames_to_process; }
  size_t block_start_offset() const { return block.block_start_offset; }
  uint64_t frame_clock() const { return block.frame_clock; }
  void rt_log(const std::string& s) {
    if (log.size() < 64) log.push_back(s);
  }
};
struct UGenFlags {
  bool remove_self_supported = false;
  bool done_ = false;
  uint32_t done_frame_in_block = 0xFFFFFFFFu;
  bool remove_self_ = false;
  bool remove_parent = false;
  uint32_t remove_parent_from_frame_in_block = 0xFFFFFFFFu;
  void mark_done(uint32_t frame) {  // ugen.rs:199-202
    done_ = true;
    done_frame_in_block = frame;
  }
  bool done(uint32_t* frame) const {
    if (done_ && frame) *frame = done_frame_in_block;
    return done_;
  }
};

// ---------------------------------------------------------------------------
// ParameterValue -- knaster_core/src/parameters/types.rs:25-36
// ---------------------------------------------------------------------------
enum class Rate : uint8_t { BlockRate = 0, AudioRate = 1 };  // knaster_core/src/lib.rs:54-61 (default BlockRate)
struct ParameterSmoothing {  // knaster_core/src/parameters/types.rs:107-114
  bool linear = false;       // false = ParameterSmoothing::None
  float seconds = 0.f;       // Linear(f32)
};
struct ParameterValue {
  enum Kind : uint8_t { Float = 0, Trigger = 1, Integer = 2, Bool = 3, Smoothing = 4 } kind = Float;
  PFloat f = 0.0;
  uint64_t i = 0;
  bool b = false;
  ParameterSmoothing smoothing;
  Rate rate = Rate::BlockRate;
  static ParameterValue Smooth(ParameterSmoothing s, Rate r = Rate::BlockRate) {
    ParameterValue p;
    p.kind = Smoothing;
    p.smoothing = s;
    p.rate = r;
    return p;
  }
  static ParameterValue Flt(PFloat v) {
    ParameterValue p;
    p.kind = Float;
    p.f = v;
    return p;
  }
  static ParameterValue Trig() {
    ParameterValue p;
    p.kind = Trigger;
    return p;
  }
  static ParameterValue Int(uint64_t v) {
    ParameterValue p;
    p.kind = Integer;
    p.i = v;
    return p;
  }
  // `.float().expect(..)` in macro-generated code (knaster_macros/src/lib.rs:601-606)
  PFloat float_or_panic() const {
    if (kind != Float) throw std::runtime_error("parameter value is expected to be a float");
    return f;
  }
  uint64_t integer_or_panic() const {
    if (kind != Integer) throw std::runtime_error("parameter value is expected to be an integer");
    return i;
  }
  bool bool_or_panic() const {
    if (kind != Bool) throw std::runtime_error("parameter value is expected to be a bool");
    return b;
  }
};

enum class ParameterError { Ok = 0, ParameterIndexOutOfBounds = 1, DescriptionNotFound = 2 };

// ---------------------------------------------------------------------------
// Blocks -- knaster_primitives/src/block.rs:33-339, knaster_graph/src/block.rs
// Channel-major; a view is an array of per-channel pointers + a frame count.
// A partial block is the same view advanced by `offset` frames.
// ---------------------------------------------------------------------------
template <typename F>
struct BlockView {
  std::vector<F*> ch;
  size_t frames = 0;
  BlockView() = default;
  BlockView(F* contiguous, size_t channels, size_t block_size) : frames(block_size) {
    for (size_t c = 0; c < channels; ++c) ch.push_back(contiguous + c * block_size);
  }
  size_t channels() const { return ch.size(); }
  F read(size_t c, size_t f) const {
    assert(c < ch.size() && f < frames);
    return ch[c][f];
  }
  void write(F v, size_t c, size_t f) {
    assert(c < ch.size() && f < frames);
    ch[c][f] = v;
  }
  F* channel(size_t c) { return ch[c]; }
  const F* channel(size_t c) const { return ch[c]; }
  BlockView partial(size_t offset, size_t length) const {  // block.rs:269-339
    assert(offset + length <= frames);
    BlockView p;
    p.frames = length;
    for (F* c : ch) p.ch.push_back(c + offset);
    return p;
  }
};

// ---------------------------------------------------------------------------
// UGen -- knaster_core/src/ugen.rs:232-369 (object-safe form, as DynUGen in
// knaster_graph/src/dynugen.rs:23-63)
// ---------------------------------------------------------------------------
template <typename F>
struct UGen {
  virtual ~UGen() = default;
  virtual size_t inputs() const = 0;
  virtual size_t outputs() const = 0;
  virtual size_t parameters() const = 0;
  virtual void init(uint32_t /*sample_rate*/, size_t /*block_size*/) {}
  // in: inputs() samples, out: outputs() samples
  virtual void process(AudioCtx& ctx, UGenFlags& flags, const F* in, F* out) = 0;
  // Default = frame loop, ugen.rs:263-284
  virtual void process_block(AudioCtx& ctx, UGenFlags& flags, const BlockView<F>& input,
                             BlockView<F>& output) {
    const size_t ni = inputs(), no = outputs();
    F in_frame[16], out_frame[16];
    assert(ni <= 16 && no <= 16);
    for (size_t frame = 0; frame < ctx.block.frames_to_process; ++frame) {
      for (size_t i = 0; i < ni; ++i) in_frame[i] = input.read(i, frame);
      process(ctx, flags, in_frame, out_frame);
      for (size_t i = 0; i < no; ++i) output.write(out_frame[i], i, frame);
    }
  }
  virtual std::vector<std::string> param_descriptions() const { return {}; }
  virtual void param_apply(AudioCtx& ctx, size_t index, ParameterValue value) = 0;
  virtual void set_ar_param_buffer(AudioCtx& ctx, size_t /*index*/, const F* /*buffer*/) {
    ctx.rt_log("Warning: Audio rate parameter buffer set, but did not reach a WrArParams");
  }
  virtual void set_delay_within_block_for_param(AudioCtx& ctx, size_t /*index*/, uint16_t /*delay*/) {
    ctx.rt_log("Warning: Parameter delay set, but did not reach a WrHiResParams");
  }
  // ugen.rs:344-368
  ParameterError param(AudioCtx& ctx, size_t index, ParameterValue value) {
    if (index >= parameters()) return ParameterError::ParameterIndexOutOfBounds;
    param_apply(ctx, index, value);
    return ParameterError::Ok;
  }
  ParameterError param(AudioCtx& ctx, const std::string& desc, ParameterValue value) {
    auto d = param_descriptions();
    for (size_t i = 0; i < d.size(); ++i)
      if (d[i] == desc) {
        param_apply(ctx, i, value);
        return ParameterError::Ok;
      }
    return ParameterError::DescriptionNotFound;
  }
};
template <typename F>
using UGenPtr = std::unique_ptr<UGen<F>>;

// ---------------------------------------------------------------------------
// Wavetable -- knaster_core_dsp/src/dsp/wavetable.rs:8-60,130-139,322-324
// ---------------------------------------------------------------------------
constexpr uint32_t TABLE_POWER = 14;
constexpr size_t TABLE_SIZE = size_t(1) << TABLE_POWER;
constexpr uint32_t TABLE_HIGH_MASK = uint32_t(TABLE_SIZE) - 1;
constexpr uint32_t FRACTIONAL_PART = 65536;

struct WavetablePhase {
  uint32_t v = 0;
  size_t integer_component() const { return (v >> 16) & TABLE_HIGH_MASK; }
  void increase(uint32_t add) { v = v + add; }  // wrapping
  WavetablePhase operator+(WavetablePhase o) const { return WavetablePhase{v + o.v}; }
};

// NonAaWavetable::<f32>::sine(): buf[i] = (f32) sin((i/16384) * PI * 2.0), f64 sin.
inline const std::vector<float>& sine_wavetable_f32() {
  static const std::vector<float> table = [] {
    std::vector<float> t(TABLE_SIZE);
    const double PI = 3.14159265358979323846;
    for (size_t i = 0; i < TABLE_SIZE; ++i)
      t[i] = static_cast<float>(
          std::sin((static_cast<double>(i) / static_cast<double>(TABLE_SIZE)) * PI * 2.0));
    return t;
  }();
  return table;
}

// ---------------------------------------------------------------------------
// SinWt -- knaster_core_dsp/src/ugens/osc.rs:97-168
// params: 0 freq, 1 phase_offset, 2 reset_phase (trigger)
// ---------------------------------------------------------------------------
template <typename F>
struct SinWt : UGen<F> {
  WavetablePhase phase, phase_offset_;
  uint32_t phase_increment = 0;
  double freq_to_phase_inc = 0.0;
  F freq_;
  const float* wavetable;
  explicit SinWt(F freq) : freq_(freq), wavetable(sine_wavetable_f32().data()) {}
  size_t inputs() const override { return 0; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 3; }
  std::vector<std::string> param_descriptions() const override {
    return {"freq", "phase_offset", "reset_phase"};
  }
  void freq(PFloat f) {  // osc.rs:127-130
    freq_ = fnew<F>(f);
    phase_increment = sat_u32(static_cast<double>(freq_) * freq_to_phase_inc);
  }
  void phase_offset(PFloat o) {  // osc.rs:133-135
    phase_offset_.v = sat_u32(o * static_cast<double>(FRACTIONAL_PART));
  }
  void reset_phase() { phase.v = 0; }
  void init(uint32_t sample_rate, size_t) override {  // osc.rs:142-147
    reset_phase();
    freq_to_phase_inc = static_cast<double>(TABLE_SIZE) * static_cast<double>(FRACTIONAL_PART) *
                        (1.0 / static_cast<double>(sample_rate));
    freq(static_cast<double>(freq_));
  }
  F next_sample() {  // osc.rs:151-156
    float s = wavetable[(phase + phase_offset_).integer_component()];
    phase.increase(phase_increment);
    return fnew<F>(s);
  }
  void process(AudioCtx&, UGenFlags&, const F*, F* out) override { out[0] = next_sample(); }
  void process_block(AudioCtx& ctx, UGenFlags&, const BlockView<F>&, BlockView<F>& output) override {
    (void)ctx;
    F* o = output.channel(0);
    for (size_t i = 0; i < output.frames; ++i) o[i] = next_sample();
  }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue v) override {
    switch (index) {
      case 0: freq(v.float_or_panic()); break;
      case 1: phase_offset(v.float_or_panic()); break;
      case 2: reset_phase(); break;
      default: ctx.rt_log("Unknown parameter set for SinWt");
    }
  }
};

// ---------------------------------------------------------------------------
// SinNumeric -- osc.rs:222-271.  Only `process` (default frame loop).
// ---------------------------------------------------------------------------
template <typename F>
struct SinNumeric : UGen<F> {
  F phase, phase_offset_ = 0, phase_increment = 0;
  explicit SinNumeric(F freq) : phase(freq) {}
  size_t inputs() const override { return 0; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 3; }
  std::vector<std::string> param_descriptions() const override {
    return {"freq", "phase_offset", "reset_phase"};
  }
  void init(uint32_t sample_rate, size_t) override {  // osc.rs:253-261
    if (phase_increment == F(0)) phase_increment = phase / fnew<F>(static_cast<float>(sample_rate));
    phase = F(0);
  }
  void process(AudioCtx&, UGenFlags&, const F*, F* out) override {  // osc.rs:263-270
    out[0] = std::sin((phase + phase_offset_) * FloatConst<F>::TAU);
    phase += phase_increment;
    if (phase > F(1)) phase -= F(1);
  }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue v) override {
    switch (index) {
      case 0:  // osc.rs:240-242
        phase_increment =
            fnew<F>(v.float_or_panic()) / fnew<F>(static_cast<float>(ctx.sample_rate()));
        break;
      case 1: phase_offset_ = fnew<F>(v.float_or_panic()); break;
      case 2: phase = F(0); break;
      default: ctx.rt_log("Unknown parameter set for SinNumeric");
    }
  }
};

// ---------------------------------------------------------------------------
// SvfFilter -- knaster_core_dsp/src/ugens/svf.rs:19-281
// params: 0 cutoff_freq, 1 q, 2 gain, 3 filter, 4 t_calculate_coefficients
// ---------------------------------------------------------------------------
enum SvfFilterType : uint8_t { Low = 0, High, Band, Notch, Peak, All, Bell, LowShelf, HighShelf };

template <typename F>
struct SvfCoeffs {
  F a1 = 0, a2 = 0, a3 = 0, m0 = 0, m1 = 0, m2 = 0;
};
// svf.rs:146-242
template <typename F>
inline SvfCoeffs<F> svf_set_coeffs(SvfFilterType ty, F cutoff, F q, F gain_db, F sample_rate) {
  const F PI = FloatConst<F>::PI, ONE = 1, ZERO = 0;
  SvfCoeffs<F> c;
  auto common = [&](F g, F k) {
    c.a1 = ONE / (ONE + g * (g + k));
    c.a2 = g * c.a1;
    c.a3 = g * c.a2;
  };
  switch (ty) {
    case Low: {
      F g = std::tan((PI * cutoff) / sample_rate), k = ONE / q;
      common(g, k);
      c.m0 = ZERO; c.m1 = ZERO; c.m2 = ONE;
    } break;
    case Band: {
      F g = std::tan((PI * cutoff) / sample_rate), k = ONE / q;
      common(g, k);
      c.m0 = ZERO; c.m1 = ONE; c.m2 = ZERO;
    } break;
    case High: {
      F g = std::tan((PI * cutoff) / sample_rate), k = ONE / q;
      common(g, k);
      c.m0 = ONE; c.m1 = -k; c.m2 = -ONE;
    } break;
    case Notch: {
      F g = std::tan((PI * cutoff) / sample_rate), k = ONE / q;
      common(g, k);
      c.m0 = ONE; c.m1 = -k; c.m2 = ZERO;
    } break;
    case Peak: {
      F g = std::tan((PI * cutoff) / sample_rate), k = ONE / q;
      common(g, k);
      c.m0 = ONE; c.m1 = -k; c.m2 = -F(2);
    } break;
    case All: {
      F g = std::tan((PI * cutoff) / sample_rate), k = ONE / q;
      common(g, k);
      c.m0 = ONE; c.m1 = -F(2) * k; c.m2 = ZERO;
    } break;
    case Bell: {
      F amp = std::pow(F(10), gain_db / F(40));
      F g = std::tan((PI * cutoff) / sample_rate) / std::sqrt(amp);
      F k = ONE / (q * amp);
      common(g, k);
      c.m0 = ONE; c.m1 = k * (amp * amp - ONE); c.m2 = ZERO;
    } break;
    case LowShelf: {
      F amp = std::pow(F(10), gain_db / F(40));
      F g = std::tan((PI * cutoff) / sample_rate) / std::sqrt(amp);
      F k = ONE / q;
      common(g, k);
      c.m0 = ONE; c.m1 = k * (amp - ONE); c.m2 = amp * amp - ONE;
    } break;
    case HighShelf: {
      F amp = std::pow(F(10), gain_db / F(40));
      F g = std::tan((PI * cutoff) / sample_rate) * std::sqrt(amp);
      F k = ONE / q;
      common(g, k);
      c.m0 = amp * amp; c.m1 = k * (ONE - amp) * amp; c.m2 = ONE - amp * amp;
    } break;
  }
  return c;
}
inline SvfFilterType svf_type_from_pinteger(uint64_t v) {  // knaster_macros/src/lib.rs:44-47
  return v <= 8 ? static_cast<SvfFilterType>(v) : Low;
}

template <typename F>
struct SvfFilter : UGen<F> {
  SvfFilterType ty;
  F cutoff_freq, q, gain_db;
  F ic1eq = 0, ic2eq = 0;
  SvfCoeffs<F> c;
  SvfFilter(SvfFilterType ty_, F cutoff, F q_, F gain) : ty(ty_), cutoff_freq(cutoff), q(q_), gain_db(gain) {}
  size_t inputs() const override { return 1; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 5; }
  std::vector<std::string> param_descriptions() const override {
    return {"cutoff_freq", "q", "gain", "filter", "t_calculate_coefficients"};
  }
  void recalc(uint32_t sr) {
    c = svf_set_coeffs<F>(ty, cutoff_freq, q, gain_db, fnew<F>(static_cast<float>(sr)));
  }
  void init(uint32_t sample_rate, size_t) override { recalc(sample_rate); }
  F process_sample(F v0) {  // svf.rs:272-278
    F v3 = v0 - ic2eq;
    F v1 = c.a1 * ic1eq + c.a2 * v3;
    F v2 = ic2eq + c.a2 * ic1eq + c.a3 * v3;
    ic1eq = F(2) * v1 - ic1eq;
    ic2eq = F(2) * v2 - ic2eq;
    return c.m0 * v0 + c.m1 * v1 + c.m2 * v2;
  }
  void process(AudioCtx&, UGenFlags&, const F* in, F* out) override { out[0] = process_sample(in[0]); }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue v) override {
    switch (index) {
      case 0: cutoff_freq = fnew<F>(v.float_or_panic()); recalc(ctx.sample_rate()); break;
      case 1: q = fnew<F>(v.float_or_panic()); recalc(ctx.sample_rate()); break;
      case 2: gain_db = fnew<F>(v.float_or_panic()); recalc(ctx.sample_rate()); break;
      case 3: ty = svf_type_from_pinteger(v.integer_or_panic()); recalc(ctx.sample_rate()); break;
      case 4: recalc(ctx.sample_rate()); break;
      default: ctx.rt_log("Unknown parameter set for SvfFilter");
    }
  }
};

// ---------------------------------------------------------------------------
// OnePole / OnePoleLpf / OnePoleHpf -- knaster_core_dsp/src/ugens/onepole.rs
// ---------------------------------------------------------------------------
template <typename T>
struct OnePole {
  T last_output = T(0.0), a0 = T(1.0), b1 = T(0.0);
  void set_freq_lowpass(T freq, T sample_rate) {  // onepole.rs:35-46
    T f = freq / sample_rate;
    T b_tmp = std::exp(T(-2.0) * FloatConst<T>::PI * f);
    b1 = b_tmp;
    a0 = T(1.0) - b1;
  }
  void set_freq_highpass(T freq, T sample_rate) { set_freq_lowpass(freq, sample_rate); }
  T process_lp(T input) {  // onepole.rs:64-77
    last_output = input * a0 + last_output * b1;
    return last_output;
  }
  T process_hp(T input) {  // onepole.rs:80-92
    last_output = input * a0 + last_output * b1;
    return input - last_output;
  }
};
template <typename F>
struct OnePoleLpf : UGen<F> {
  OnePole<F> op;
  explicit OnePoleLpf(F cutoff) { op.b1 = cutoff; }  // onepole.rs:118-122
  size_t inputs() const override { return 1; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 1; }
  std::vector<std::string> param_descriptions() const override { return {"cutoff_freq"}; }
  void init(uint32_t sample_rate, size_t) override {  // onepole.rs:123-129
    if (op.a0 == F(1)) {
      F freq = op.b1;
      op.set_freq_lowpass(freq, fnew<F>(static_cast<float>(sample_rate)));
    }
  }
  void process(AudioCtx&, UGenFlags&, const F* in, F* out) override { out[0] = op.process_lp(in[0]); }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue v) override {
    if (index == 0)
      op.set_freq_lowpass(fnew<F>(v.float_or_panic()), static_cast<F>(ctx.sample_rate()));
    else
      ctx.rt_log("Unknown parameter set for OnePoleLpf");
  }
};
template <typename F>
struct OnePoleHpf : UGen<F> {
  OnePole<F> op;
  OnePoleHpf() = default;
  size_t inputs() const override { return 1; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 1; }
  std::vector<std::string> param_descriptions() const override { return {"cutoff_freq"}; }
  void init(uint32_t sample_rate, size_t) override {  // onepole.rs:161-167
    if (op.a0 == F(1)) {
      F freq = op.b1;
      op.set_freq_highpass(freq, fnew<F>(static_cast<float>(sample_rate)));
    }
  }
  void process(AudioCtx&, UGenFlags&, const F* in, F* out) override { out[0] = op.process_hp(in[0]); }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue v) override {
    if (index == 0)
      op.set_freq_highpass(fnew<F>(v.float_or_panic()), static_cast<F>(ctx.sample_rate()));
    else
      ctx.rt_log("Unknown parameter set for OnePoleHpf");
  }
};

// ---------------------------------------------------------------------------
// EnvAsr / EnvAr -- knaster_core_dsp/src/ugens/envelopes.rs:19-303
// ---------------------------------------------------------------------------
template <typename F>
inline F powi3(F t) {  // num-traits powi(3): acc = t; base = t*t; acc = acc*base
  return t * (t * t);
}
enum class AsrState : uint32_t { Stopped = 0, Attacking = 1, Sustaining = 2, Releasing = 3 };

template <typename F>
struct EnvAsr : UGen<F> {
  AsrState state = AsrState::Stopped;
  F t = 0, attack_seconds, attack_rate = 1, release_seconds, release_rate = 1, release_scale = 1;
  EnvAsr(F attack_time, F release_time) : attack_seconds(attack_time), release_seconds(release_time) {}
  size_t inputs() const override { return 0; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 4; }
  std::vector<std::string> param_descriptions() const override {
    return {"attack_time", "release_time", "t_release", "t_restart"};
  }
  F next_sample(UGenFlags& flags, uint32_t sample_in_block) {  // envelopes.rs:52-81
    F out;
    switch (state) {
      case AsrState::Stopped: out = F(0); break;
      case AsrState::Attacking:
        out = t;
        t += attack_rate;
        if (t >= F(1)) state = AsrState::Sustaining;
        break;
      case AsrState::Sustaining: out = F(1); break;
      case AsrState::Releasing:
      default:
        out = powi3(t) * release_scale;
        t -= release_rate;
        if (t <= F(0)) {
          state = AsrState::Stopped;
          t = F(0);
          flags.mark_done(sample_in_block);
        }
        break;
    }
    return out;
  }
  void init(uint32_t sample_rate, size_t) override {  // envelopes.rs:135-151
    if (attack_rate == F(1)) {
      if (attack_seconds == F(0)) attack_rate = F(1);
      else attack_rate = F(1) / (attack_seconds * static_cast<F>(sample_rate));
    }
    if (release_rate == F(1)) {
      if (release_seconds == F(0)) release_rate = F(1);
      else release_rate = F(1) / (release_seconds * static_cast<F>(sample_rate));
    }
  }
  void process(AudioCtx&, UGenFlags& flags, const F*, F* out) override { out[0] = next_sample(flags, 0); }
  void process_block(AudioCtx&, UGenFlags& flags, const BlockView<F>&, BlockView<F>& output) override {
    F* o = output.channel(0);
    for (size_t i = 0; i < output.frames; ++i) o[i] = next_sample(flags, static_cast<uint32_t>(i));
  }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue v) override {
    switch (index) {
      case 0: {  // envelopes.rs:85-96
        F atk = fnew<F>(v.float_or_panic());
        if (attack_seconds != atk) {
          attack_seconds = atk;
          if (atk == F(0)) attack_rate = F(1);
          else attack_rate = F(1) / (attack_seconds * static_cast<F>(ctx.sample_rate()));
        }
      } break;
      case 1: {  // envelopes.rs:99-110
        F rel = fnew<F>(v.float_or_panic());
        if (release_seconds != rel) {
          release_seconds = rel;
          if (rel == F(0)) release_rate = F(1);
          else release_rate = F(1) / (release_seconds * static_cast<F>(ctx.sample_rate()));
        }
      } break;
      case 2:  // t_release, envelopes.rs:113-128
        switch (state) {
          case AsrState::Stopped: break;
          case AsrState::Attacking:
            release_scale = t;
            state = AsrState::Releasing;
            t = F(1);
            break;
          case AsrState::Sustaining:
            release_scale = F(1);
            state = AsrState::Releasing;
            t = F(1);
            break;
          case AsrState::Releasing: break;
        }
        break;
      case 3: state = AsrState::Attacking; break;  // t_restart: t is NOT reset
      default: ctx.rt_log("Unknown parameter set for EnvAsr");
    }
  }
};

enum class ArState : uint32_t { Stopped = 0, Attacking = 1, Releasing = 3 };
template <typename F>
struct EnvAr : UGen<F> {
  ArState state = ArState::Stopped;
  F t = 0, attack_seconds, attack_rate = 1, release_seconds, release_rate = 1, release_scale = 1;
  EnvAr(F attack_time, F release_time) : attack_seconds(attack_time), release_seconds(release_time) {}
  size_t inputs() const override { return 0; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 3; }
  std::vector<std::string> param_descriptions() const override {
    return {"attack_time", "release_time", "t_restart"};
  }
  F next_sample(UGenFlags& flags, uint32_t sample_in_block) {  // envelopes.rs:205-233
    F out;
    switch (state) {
      case ArState::Stopped: out = F(0); break;
      case ArState::Attacking:
        out = t;
        t += attack_rate;
        if (t >= F(1)) {
          release_scale = F(1);
          state = ArState::Releasing;
          t = F(1);
        }
        break;
      case ArState::Releasing:
      default:
        out = powi3(t) * release_scale;
        t -= release_rate;
        if (t <= F(0)) {
          state = ArState::Stopped;
          t = F(0);
          flags.mark_done(sample_in_block);
        }
        break;
    }
    return out;
  }
  void init(uint32_t sample_rate, size_t) override {  // envelopes.rs:268-284
    if (attack_rate == F(1)) {
      if (attack_seconds == F(0)) attack_rate = F(1);
      else attack_rate = F(1) / (attack_seconds * static_cast<F>(sample_rate));
    }
    if (release_rate == F(1)) {
      if (release_seconds == F(0)) release_rate = F(1);
      else release_rate = F(1) / (release_seconds * static_cast<F>(sample_rate));
    }
  }
  void process(AudioCtx&, UGenFlags& flags, const F*, F* out) override { out[0] = next_sample(flags, 0); }
  void process_block(AudioCtx&, UGenFlags& flags, const BlockView<F>&, BlockView<F>& output) override {
    F* o = output.channel(0);
    for (size_t i = 0; i < output.frames; ++i) o[i] = next_sample(flags, static_cast<uint32_t>(i));
  }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue v) override {
    switch (index) {
      case 0: {
        F atk = fnew<F>(v.float_or_panic());
        if (attack_seconds != atk) {
          attack_seconds = atk;
          if (atk == F(0)) attack_rate = F(1);
          else attack_rate = F(1) / (attack_seconds * static_cast<F>(ctx.sample_rate()));
        }
      } break;
      case 1: {
        F rel = fnew<F>(v.float_or_panic());
        if (release_seconds != rel) {
          release_seconds = rel;
          if (rel == F(0)) release_rate = F(1);
          else release_rate = F(1) / (release_seconds * static_cast<F>(ctx.sample_rate()));
        }
      } break;
      case 2: state = ArState::Attacking; break;
      default: ctx.rt_log("Unknown parameter set for EnvAr");
    }
  }
};

// ---------------------------------------------------------------------------
// Envelope (segment envelope) -- knaster_core_dsp/src/ugens/envelopes.rs:319-527
// All state is f64 regardless of F; only `process` exists (default frame loop).
// params: 0 time_scale, 1 jump_to_segment (integer), 2 t_restart, 3 t_stop
// ---------------------------------------------------------------------------
struct EnvelopeSegment {
  double reciprocal_duration, duration, value;
  EnvelopeSegment(double duration_, double value_) : reciprocal_duration(1.0 / duration_), duration(duration_), value(value_) {}
};
template <typename F>
struct Envelope : UGen<F> {
  bool running = false;
  size_t run_segment = 0;     // EnvelopeState::Running { current_segment, current_time }
  double run_time = 0.0;
  std::vector<EnvelopeSegment> segments;
  double start_value, from_value;
  size_t current_segment = 0;
  double time_scale = 1.0, base_scale = 0.0;
  bool looping = false;
  Envelope(double start, std::vector<EnvelopeSegment> segs) : segments(std::move(segs)), start_value(start), from_value(start) {}
  size_t inputs() const override { return 0; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 4; }
  std::vector<std::string> param_descriptions() const override { return {"time_scale", "jump_to_segment", "t_restart", "t_stop"}; }
  void init(uint32_t sample_rate, size_t) override { base_scale = 1.0 / static_cast<double>(sample_rate); }
  void process(AudioCtx&, UGenFlags& flags, const F*, F* out) override {  // envelopes.rs:407-463
    if (!running) {
      out[0] = fnew<F>(from_value);
      return;
    }
    const double t = run_time;
    const size_t cs = run_segment;
    if (t < segments[cs].duration) {
      const EnvelopeSegment& seg = segments[cs];
      out[0] = fnew<F>(from_value + (t * seg.reciprocal_duration) * (seg.value - from_value));
      run_time = t + (time_scale * base_scale);
    } else if (cs + 1 < segments.size()) {
      from_value = segments[cs].value;
      const EnvelopeSegment& seg = segments[cs];
      out[0] = fnew<F>(from_value + (t * seg.reciprocal_duration) * (seg.value - from_value));
      run_segment = cs + 1;
      run_time = t - seg.duration + (time_scale * base_scale);
    } else {
      from_value = segments[cs].value;
      out[0] = fnew<F>(from_value);
      if (looping) {
        run_segment = 0;
        run_time = 0.0;
      } else {
        running = false;
        flags.mark_done(0);
      }
    }
  }
  void param_apply(AudioCtx&, size_t index, ParameterValue v) override {  // envelopes.rs:478-524
    switch (index) {
      case 0: time_scale = static_cast<double>(fnew<F>(v.float_or_panic())); break;
      case 1: {
        size_t j = static_cast<size_t>(v.integer_or_panic());
        if (j >= segments.size()) j = segments.size() - 1;
        running = true;
        run_segment = j;
        run_time = 0.0;
        current_segment = j;
      } break;
      case 2:
        running = true;
        run_segment = 0;
        run_time = 0.0;
        from_value = start_value;
        break;
      case 3:
        if (running) {
          const EnvelopeSegment& seg = segments[run_segment];
          from_value = from_value + (run_time * seg.reciprocal_duration) * (seg.value - from_value);
        }
        running = false;
        break;
      default: break;
    }
  }
};

// ---------------------------------------------------------------------------
// Buffer (single channel) -- knaster_core_dsp/src/dsp/buffer.rs:38-133;  BufferReader<F, U1> -- ugens/buffer.rs:19-191
// ---------------------------------------------------------------------------
template <typename F>
struct Buffer {
  std::vector<F> buffer;
  double sample_rate = 48000.0;
  double num_frames() const { return static_cast<double>(buffer.size()); }
  double buf_rate_scale(uint32_t server_sample_rate) const { return sample_rate / static_cast<double>(server_sample_rate); }
  double length_seconds() const { return num_frames() / sample_rate; }
  F get_linear_interp_f64(double index) const {  // :100-110 with num_channels = 1, channel = 0
    const F mix = fnew<F>(index - std::trunc(index));
    size_t index_u = index > 0.0 ? static_cast<size_t>(index) : 0;  // `as usize` saturates
    if (index_u >= buffer.size()) index_u = buffer.size() - 1;       // undefined behaviour in the reference; the product clamps
    return buffer[index_u] * (F(1) - mix) + buffer[(index_u + 1) % buffer.size()] * mix;
  }
};
template <typename F>
struct BufferReader : UGen<F> {
  std::shared_ptr<const Buffer<F>> buffer;
  double read_pointer = 0.0, rate, base_rate = 0.0;
  bool finished = false, looping;
  double start_frame = 0.0, dur_frame, end_frame;
  BufferReader(std::shared_ptr<const Buffer<F>> b, double rate_, bool looping_, double start_at_seconds = 0.0)
      : buffer(std::move(b)), rate(rate_), looping(looping_), start_frame(start_at_seconds) {
    dur_frame = end_frame = buffer->length_seconds();
  }
  size_t inputs() const override { return 0; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 6; }
  std::vector<std::string> param_descriptions() const override { return {"rate", "looping", "start_s", "duration_s", "end_s", "t_restart"}; }
  void jump_to(double p) { read_pointer = p; finished = false; }
  void init(uint32_t sample_rate, size_t) override {  // :106-115
    base_rate = buffer->buf_rate_scale(sample_rate);
    start_frame = Seconds::from_secs_f64(start_frame).to_samples_f64(buffer->sample_rate);
    dur_frame = Seconds::from_secs_f64(dur_frame).to_samples_f64(buffer->sample_rate);
    end_frame = start_frame + dur_frame;
    jump_to(start_frame);
  }
  void process(AudioCtx&, UGenFlags& flags, const F*, F* out) override {  // :119-141 (unused by the graph: it calls process_block)
    if (finished) { out[0] = F(0); return; }
    out[0] = buffer->get_linear_interp_f64(read_pointer);
    read_pointer += base_rate * rate;
    if (read_pointer >= end_frame) {
      finished = true;
      if (looping) jump_to(start_frame);
      else flags.mark_done(0);
    }
  }
  void process_block(AudioCtx& ctx, UGenFlags& flags, const BlockView<F>&, BlockView<F>& output) override {  // :143-190
    F* o = output.channel(0);
    size_t stop_sample = SIZE_MAX;
    if (!finished) {
      for (size_t i = 0; i < ctx.block_size(); ++i) {
        o[i] = buffer->get_linear_interp_f64(read_pointer);
        read_pointer += base_rate * rate;
        if (read_pointer >= end_frame) {
          finished = true;
          if (looping) jump_to(start_frame);
          else flags.mark_done(static_cast<uint32_t>(i + 1));
        }
        if (finished) { stop_sample = i + 1; break; }
      }
    } else {
      stop_sample = 0;
    }
    if (stop_sample != SIZE_MAX)
      for (size_t i = stop_sample; i < ctx.block_size(); ++i) o[i] = F(0);
  }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue v) override {  // :62-103
    switch (index) {
      case 0: rate = v.float_or_panic(); break;
      case 1: looping = v.bool_or_panic(); break;
      case 2:
        start_frame = Seconds::from_secs_f64(v.float_or_panic()).to_samples_f64(buffer->sample_rate);
        end_frame = start_frame + dur_frame;
        break;
      case 3:
        dur_frame = Seconds::from_secs_f64(v.float_or_panic()).to_samples_f64(buffer->sample_rate);
        end_frame = start_frame + dur_frame;
        break;
      case 4: end_frame = Seconds::from_secs_f64(v.float_or_panic()).to_samples_f64(buffer->sample_rate); break;
      case 5: jump_to(start_frame); break;
      default: ctx.rt_log("Unknown parameter set for BufferReader");
    }
  }
};

// ---------------------------------------------------------------------------
// PolyBlep -- knaster_core_dsp/src/ugens/polyblep.rs:41-508
// ---------------------------------------------------------------------------
enum class Waveform : uint8_t {
  Sawtooth = 0, Sine, Cosine, Triangle, Square, Rectangle, Ramp, ModifiedTriangle, ModifiedSquare,
  HalfWaveRectifiedSine, FullWaveRectifiedSine, TriangularPulse, TrapezoidFixed, TrapezoidVariable
};
inline Waveform waveform_from_pinteger(uint64_t v) { return v < 14 ? static_cast<Waveform>(v) : Waveform::Sawtooth; }  // macro: unwrap_or(default)
template <typename F>
struct PolyBlep : UGen<F> {
  Waveform waveform;
  F sample_rate = F(0), freq_in_hz, dt = F(0), pulse_width = F(0.5), t = F(0);
  PolyBlep(Waveform wf, F freq) : waveform(wf), freq_in_hz(freq) {}
  size_t inputs() const override { return 0; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 3; }
  std::vector<std::string> param_descriptions() const override { return {"freq", "pulse_width", "waveform"}; }
  static F tau() { return static_cast<F>(6.28318530717958647692528676655900577); }
  static F pi() { return static_cast<F>(3.14159265358979323846264338327950288); }
  static F square_number(F v) { return v * v; }
  static F bitwise_or_zero(F v) { return std::trunc(v); }
  static F blep(F t, F dt) {
    if (t < dt) return -square_number(t / dt - F(1));
    if (t > F(1) - dt) return square_number((t - F(1)) / dt + F(1));
    return F(0);
  }
  static F blamp(F t, F dt) {
    if (t < dt) {
      t = t / dt - F(1);
      return -F(1) / F(3) * square_number(t) * t;
    }
    if (t > F(1) - dt) {
      t = (t - F(1)) / dt + F(1);
      return F(1) / F(3) * square_number(t) * t;
    }
    return F(0);
  }
  static F fmin_rs(F a, F b) { return std::isnan(a) ? b : (std::isnan(b) ? a : (a < b ? a : b)); }  // f32::min
  static F fmax_rs(F a, F b) { return std::isnan(a) ? b : (std::isnan(b) ? a : (a > b ? a : b)); }
  static F clamp_rs(F v, F lo, F hi) { return v < lo ? lo : (v > hi ? hi : v); }                    // f32::clamp
  void set_freq(F f) { freq_in_hz = f; dt = f / sample_rate; }
  F get_freq_in_hz() const { return dt * sample_rate; }
  void init(uint32_t sr, size_t) override {
    sample_rate = static_cast<F>(sr);
    if (dt == F(0) && freq_in_hz != F(0)) set_freq(freq_in_hz);
  }
  F wrap(F v) const { return v - bitwise_or_zero(v); }
  F sin_w() const { return std::sin(t * tau()); }
  F cos_w() const { return std::cos(t * tau()); }
  F half() const {
    F t2 = wrap(t + F(0.5));
    F y = t < F(0.5) ? F(2) * std::sin(t * tau()) - F(2) / pi() : -F(2) / pi();
    y += tau() * dt * (blamp(t, dt) + blamp(t2, dt));
    return y;
  }
  F full() const {
    F u = wrap(t + F(0.25));
    F y = F(2) * std::sin(u * pi()) - F(4) / pi();
    y += tau() * dt * blamp(u, dt);
    return y;
  }
  static F fold(F y) {
    if (y >= F(3)) y -= F(4);
    else if (y > F(1)) y = F(2) - y;
    return y;
  }
  F tri() const {
    F t1 = wrap(t + F(0.25)), t2 = wrap(t + F(0.75));
    F y = fold(t * F(4));
    y += F(4) * dt * (blamp(t1, dt) - blamp(t2, dt));
    return y;
  }
  F tri2() const {
    F pw = fmax_rs(fmin_rs(pulse_width, F(0.9999)), F(0.0001));
    F t1 = wrap(t + F(0.5) * pw), t2 = wrap(t + F(1) - F(0.5) * pw);
    F y = t * F(2);
    if (y >= F(2) - pw) y = (y - F(2)) / pw;
    else if (y >= pw) y = F(1) - (y - pw) / (F(1) - pw);
    else y /= pw;
    y += dt / (pw - pw * pw) * (blamp(t1, dt) - blamp(t2, dt));
    return y;
  }
  F trip() const {
    F t1 = wrap(t + F(0.75) + F(0.5) * pulse_width);
    F y;
    if (t1 >= pulse_width) {
      y = -pulse_width;
    } else {
      y = F(4) * t1;
      y = y >= F(2) * pulse_width ? F(4) - y / pulse_width - pulse_width : y / pulse_width - pulse_width;
    }
    if (pulse_width > F(0)) {
      F t2 = wrap(t1 + F(1) - F(0.5) * pulse_width), t3 = wrap(t1 + F(1) - pulse_width);
      y += F(2) * dt / pulse_width * (blamp(t1, dt) - F(2) * blamp(t2, dt) + blamp(t3, dt));
    }
    return y;
  }
  F trap() const {
    F y = fold(F(4) * t);
    y = clamp_rs(F(2) * y, -F(1), F(1));
    F t1 = wrap(t + F(0.125)), t2 = wrap(t1 + F(0.5));
    y += F(4) * dt * (blamp(t1, dt) - blamp(t2, dt));
    t1 = wrap(t + F(0.375));
    t2 = wrap(t1 + F(0.5));
    y += F(4) * dt * (blamp(t1, dt) - blamp(t2, dt));
    return y;
  }
  F trap2() const {
    F pw = fmin_rs(pulse_width, F(0.9999));
    F scale = F(1) / (F(1) - pw);
    F y = fold(F(4) * t);
    y = clamp_rs(scale * y, -F(1), F(1));
    F t1 = wrap(t + F(0.25) - F(0.25) * pw), t2 = wrap(t1 + F(0.5));
    y += scale * F(2) * dt * (blamp(t1, dt) - blamp(t2, dt));
    t1 = wrap(t + F(0.25) + F(0.25) * pw);
    t2 = wrap(t1 + F(0.5));
    y += scale * F(2) * dt * (blamp(t1, dt) - blamp(t2, dt));
    return y;
  }
  F sqr() const {
    F t2 = wrap(t + F(0.5));
    F y = t < F(0.5) ? F(1) : -F(1);
    y += blep(t, dt) - blep(t2, dt);
    return y;
  }
  F sqr2() const {
    F t1 = wrap(t + F(0.875) + F(0.25) * (pulse_width - F(0.5)));
    F t2 = wrap(t + F(0.375) + F(0.25) * (pulse_width - F(0.5)));
    F y = t1 < F(0.5) ? F(1) : -F(1);
    y += blep(t1, dt) - blep(t2, dt);
    t1 = wrap(t1 + F(0.5) * (F(1) - pulse_width));
    t2 = wrap(t2 + F(0.5) * (F(1) - pulse_width));
    y += t1 < F(0.5) ? F(1) : -F(1);
    y += blep(t1, dt) - blep(t2, dt);
    return F(0.5) * y;
  }
  F rect() const {
    F t2 = wrap(t + F(1) - pulse_width);
    F y = -F(2) * pulse_width;
    if (t < pulse_width) y += F(2);
    y += blep(t, dt) - blep(t2, dt);
    return y;
  }
  F saw() const {
    F u = wrap(t + F(0.5));
    F y = F(2) * u - F(1);
    y -= blep(u, dt);
    return y;
  }
  F ramp() const {
    F u = wrap(t);
    F y = F(1) - F(2) * u;
    y += blep(u, dt);
    return y;
  }
  F next_sample() const {
    if (get_freq_in_hz() >= sample_rate / F(4)) return sin_w();
    switch (waveform) {
      case Waveform::Sine: return sin_w();
      case Waveform::Cosine: return cos_w();
      case Waveform::Triangle: return tri();
      case Waveform::Square: return sqr();
      case Waveform::Rectangle: return rect();
      case Waveform::Sawtooth: return saw();
      case Waveform::Ramp: return ramp();
      case Waveform::ModifiedTriangle: return tri2();
      case Waveform::ModifiedSquare: return sqr2();
      case Waveform::HalfWaveRectifiedSine: return half();
      case Waveform::FullWaveRectifiedSine: return full();
      case Waveform::TriangularPulse: return trip();
      case Waveform::TrapezoidFixed: return trap();
      case Waveform::TrapezoidVariable: return trap2();
    }
    return F(0);
  }
  void process(AudioCtx&, UGenFlags&, const F*, F* out) override {
    out[0] = next_sample();
    t += dt;
    t -= bitwise_or_zero(t);
  }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue v) override {
    switch (index) {
      case 0: set_freq(fnew<F>(v.float_or_panic())); break;
      case 1: pulse_width = fnew<F>(v.float_or_panic()); break;
      case 2: waveform = waveform_from_pinteger(static_cast<uint64_t>(v.integer_or_panic())); break;
      default: ctx.rt_log("Unknown parameter set for PolyBlep");
    }
  }
};

// ---------------------------------------------------------------------------
// WhiteNoise / PinkNoise / BrownNoise -- knaster_core_dsp/src/ugens/noise.rs:26-156
//
// PARITY UNPINNED.  The random numbers come from the `fastrand` crate (Cargo.lock:937-940: version 2.3.0), which is a
// crates.io dependency and not in the reference tree, and the reference has no test or golden vector for these UGens.
// FastRng restates fastrand 2.3.0's published algorithm: `Rng::with_seed(seed)` = `Rng(seed)` (a u64); `gen_u64`:
//   s = state.wrapping_add(0x2d35_8dcc_aa6c_78a5); state = s; t = (s as u128) * ((s ^ 0x8bb8_4b93_962e_acc9) as u128);
//   (t as u64) ^ (t >> 64) as u64                              (wyrand with the final v4.2 constants)
// `u32(..)` over the full range = `gen_u64() as u32`; `f32()` = f32::from_bits((1 << 30) - (1 << 23) + (u32(..) >> 9)) - 1.0.
// What IS anchored on the reference: the call sites -- one `rng.f32() * 2.0 - 1.0` (f32 arithmetic) per draw, cast with
// F::new, the order of draws inside PinkNoise::process, the seed = next_randomness_seed() (a process-wide counter
// from 0, noise.rs:11-22; here the constructor argument).
// ---------------------------------------------------------------------------
struct FastRng {
  uint64_t state;
  explicit FastRng(uint64_t seed) : state(seed) {}
  uint64_t gen_u64() {
    const uint64_t s = state + 0x2d358dccaa6c78a5ull;
    state = s;
    const unsigned __int128 t = static_cast<unsigned __int128>(s) * static_cast<unsigned __int128>(s ^ 0x8bb84b93962eacc9ull);
    return static_cast<uint64_t>(t) ^ static_cast<uint64_t>(t >> 64);
  }
  uint32_t u32() { return static_cast<uint32_t>(gen_u64()); }
  float f32() {
    const uint32_t bits = (1u << 30) - (1u << 23) + (u32() >> 9);
    float f;
    std::memcpy(&f, &bits, 4);
    return f - 1.0f;
  }
};
template <typename F>
struct WhiteNoise : UGen<F> {
  FastRng rng;
  explicit WhiteNoise(uint64_t seed) : rng(seed) {}
  size_t inputs() const override { return 0; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 0; }
  void param_apply(AudioCtx&, size_t, ParameterValue) override {}  // no parameters
  void process(AudioCtx&, UGenFlags&, const F*, F* out) override { out[0] = fnew<F>(rng.f32() * 2.0f - 1.0f); }
};
template <typename F>
struct PinkNoise : UGen<F> {  // noise.rs:49-111
  static constexpr uint32_t kOctaves = 9;
  F white_noises[kOctaves] = {}, always_on_white_noise = F(0), pink = F(0);
  uint32_t counter = 1, mask = 1u << (kOctaves - 1);
  FastRng rng;
  explicit PinkNoise(uint64_t seed) : rng(seed) {}
  size_t inputs() const override { return 0; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 0; }
  void param_apply(AudioCtx&, size_t, ParameterValue) override {}  // no parameters
  void process(AudioCtx&, UGenFlags&, const F*, F* out) override {
    if (!(counter > 0 && counter <= mask)) throw std::runtime_error("PinkNoise: counter out of range");  // the asserts
    const uint32_t index = static_cast<uint32_t>(__builtin_ctz(counter));
    if (index >= kOctaves) throw std::runtime_error("PinkNoise: index out of range");
    pink -= white_noises[index];
    white_noises[index] = fnew<F>(rng.f32() * 2.0f - 1.0f);
    pink += white_noises[index];
    pink -= always_on_white_noise;
    always_on_white_noise = fnew<F>(rng.f32() * 2.0f - 1.0f);
    pink += always_on_white_noise;
    counter &= mask - 1;
    counter += 1;
    out[0] = pink / (F(kOctaves) + F(1));
  }
};
template <typename F>
struct RandomLin : UGen<F> {  // noise.rs:158-230
  FastRng rng;
  F current_value, current_change_width = F(0), phase = F(0), phase_step, freq_to_phase_inc = F(0);
  RandomLin(uint64_t counter, F freq) : rng(counter * 94u + 53u), current_value(fnew<F>(rng.f32())), phase_step(freq) {}
  size_t inputs() const override { return 0; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 1; }
  std::vector<std::string> param_descriptions() const override { return {"freq"}; }
  void new_value() {
    const F old_target = current_value + current_change_width;
    const F fresh = fnew<F>(rng.f32());
    current_value = old_target;
    current_change_width = fresh - old_target;
    phase = F(0);
  }
  void init(uint32_t sample_rate, size_t) override {
    freq_to_phase_inc = F(1) / static_cast<F>(sample_rate);
    phase_step *= freq_to_phase_inc;  // freq is stored in phase_step until init
    new_value();
  }
  void process(AudioCtx&, UGenFlags&, const F*, F* out) override {
    out[0] = current_value + phase * current_change_width;
    phase += phase_step;
    if (phase >= F(1)) new_value();
  }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue v) override {
    if (index != 0) { ctx.rt_log("Unknown parameter set for RandomLin"); return; }
    const F value = fnew<F>(v.float_or_panic());
    phase_step = freq_to_phase_inc == F(0) ? value : value * freq_to_phase_inc;
  }
};
template <typename F>
struct BrownNoise : UGen<F> {  // noise.rs:119-156
  FastRng rng;
  F last_output = F(0);
  explicit BrownNoise(uint64_t seed) : rng(seed) {}
  size_t inputs() const override { return 0; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 0; }
  void param_apply(AudioCtx&, size_t, ParameterValue) override {}  // no parameters
  void process(AudioCtx&, UGenFlags&, const F*, F* out) override {
    const F white = fnew<F>(rng.f32() * 2.0f - 1.0f);
    last_output += white * fnew<F>(0.1);  // F::new(0.1): the f64 literal cast to F
    if (last_output < F(-1)) last_output = F(-1);  // clamp: NaN stays
    if (last_output > F(1)) last_output = F(1);
    out[0] = last_output;
  }
};

// ---------------------------------------------------------------------------
// Phasor -- knaster_core_dsp/src/ugens/osc.rs:172-214;  SafetyLimiter -- ugens/dynamics.rs:9-31
// ---------------------------------------------------------------------------
template <typename F>
struct Phasor : UGen<F> {
  double phase = 0.0, step, freq_to_phase_step_mult = 0.0;
  explicit Phasor(double freq) : step(freq) {}
  size_t inputs() const override { return 0; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 1; }
  std::vector<std::string> param_descriptions() const override { return {"freq"}; }
  void set_freq(double freq) { step = freq_to_phase_step_mult == 0.0 ? freq : freq * freq_to_phase_step_mult; }
  void init(uint32_t sample_rate, size_t) override {
    freq_to_phase_step_mult = 1.0 / static_cast<double>(sample_rate);
    set_freq(step);
  }
  void process(AudioCtx&, UGenFlags&, const F*, F* out) override {
    out[0] = fnew<F>(phase);
    phase += step;
    while (phase >= 1.0) phase -= 1.0;
  }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue v) override {
    if (index == 0) set_freq(v.float_or_panic());
    else ctx.rt_log("Unknown parameter set for Phasor");
  }
};
template <typename F>
struct SafetyLimiter : UGen<F> {
  size_t inputs() const override { return 1; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 0; }
  void process(AudioCtx&, UGenFlags&, const F* in, F* out) override {
    F s = in[0];
    if (s < F(-1)) s = F(-1);  // f32::clamp(min, max): NaN in, NaN out
    if (s > F(1)) s = F(1);
    out[0] = std::isnan(s) ? F(0) : s;
  }
  void param_apply(AudioCtx&, size_t, ParameterValue) override {}
};

// ---------------------------------------------------------------------------
// Pan2 -- knaster_core_dsp/src/ugens/pan.rs:12-37.  One input, two outputs; the gains come from
// `fastapprox::fast::{cos, sin}` (Cargo.lock:931, fastapprox 0.3.1 -- a crates.io dependency that is NOT in the
// reference tree).  PARITY UNPINNED: the two functions are restated from the algorithm the crate ports (P. Mineiro's
// fastapprox, fasttrig.h: `fastsin` is a parabola q = 4/pi x - 4/pi^2 x|x| refined by an odd polynomial in q whose
// coefficients take the sign of x by bit operations; `fastcos(x) = fastsin(x + pi/2)`, wrapped by -3pi/2 above pi/2);
// no reference test holds a Pan2 output.  Anchored on the call site: pan stored as pan * 0.5 + 0.5 (:20,:28), the
// angle pan * FRAC_PI_2 in f32 (:33), gains cast with F::new (:34-35), out = [x * left, x * right] (:36).
// ---------------------------------------------------------------------------
namespace fastapprox_fast {
inline float bits_to_f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint32_t f_to_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float sin(float x) {
  const float fouroverpi = 1.2732395447351627f;
  const float fouroverpisq = 0.40528473456935109f;
  const float q = 0.78444488374548933f;
  uint32_t p = f_to_bits(0.20363937680730309f);
  uint32_t r = f_to_bits(0.015124940802184233f);
  uint32_t s = f_to_bits(-0.0032225901625579573f);
  const uint32_t vx = f_to_bits(x);
  const uint32_t sign = vx & 0x80000000u;
  const float abs_x = bits_to_f(vx & 0x7FFFFFFFu);
  const float qpprox = fouroverpi * x - fouroverpisq * x * abs_x;
  const float qpproxsq = qpprox * qpprox;
  p |= sign;
  r |= sign;
  s ^= sign;
  return q * qpprox + qpproxsq * (bits_to_f(p) + qpproxsq * (bits_to_f(r) + qpproxsq * bits_to_f(s)));
}
inline float cos(float x) {
  const float halfpi = 1.5707963267948966f;
  const float halfpiminustwopi = -4.7123889803846899f;
  const float offset = (x > halfpi) ? halfpiminustwopi : halfpi;
  return sin(x + offset);
}
}  // namespace fastapprox_fast

template <typename F>
struct Pan2 : UGen<F> {
  float pan;
  explicit Pan2(float p) : pan(p * 0.5f + 0.5f) {}  // :18-23
  size_t inputs() const override { return 1; }
  size_t outputs() const override { return 2; }
  size_t parameters() const override { return 1; }
  std::vector<std::string> param_descriptions() const override { return {"pan"}; }
  void process(AudioCtx&, UGenFlags&, const F* in, F* out) override {  // :31-37
    const F signal = in[0];
    const float pan_pos_radians = pan * 1.57079632679489661923132169163975144f;
    const F left_gain = fnew<F>(fastapprox_fast::cos(pan_pos_radians));
    const F right_gain = fnew<F>(fastapprox_fast::sin(pan_pos_radians));
    out[0] = signal * left_gain;
    out[1] = signal * right_gain;
  }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue value) override {
    if (index == 0) pan = static_cast<float>(value.f) * 0.5f + 0.5f;  // :26-29 (`pan: f32`: the macro casts the PFloat)
    else ctx.rt_log("Unknown parameter set for Pan2");
  }
};

// ---------------------------------------------------------------------------
// AllpassInterpolator / AllpassDelay -- knaster_core_dsp/src/ugens/delay.rs:53-206
// ---------------------------------------------------------------------------
template <typename F>
struct AllpassInterpolator {
  F coeff = F(1), prev_input = F(1), prev_output = F(1);  // new(): all ONE
  void clear() { prev_input = F(1); prev_output = F(1); }
  void set_delta(F delta) { coeff = (F(1) - delta) / (F(1) + delta); }
  F process_sample(F input) {
    const F output = coeff * (input - prev_output) + prev_input;
    prev_output = output;
    prev_input = input;
    return output;
  }
};
template <typename F>
struct AllpassDelay : UGen<F> {
  std::vector<F> buffer;
  size_t write_frame = 0, read_frame = 0, num_frames = 1, clear_nr_of_samples_left = 0;
  AllpassInterpolator<F> allpass;
  Seconds max_delay_seconds;
  explicit AllpassDelay(Seconds max_delay) : max_delay_seconds(max_delay) {}
  size_t inputs() const override { return 1; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 1; }
  std::vector<std::string> param_descriptions() const override { return {"delay_time"}; }
  void init(uint32_t sample_rate, size_t) override { buffer.assign(static_cast<size_t>(max_delay_seconds.to_samples(sample_rate)), F(0)); }
  F read() {  // :145-157
    if (clear_nr_of_samples_left > 0) {
      clear_nr_of_samples_left -= 1;
      read_frame = (read_frame + 1) % buffer.size();
      return F(0);
    }
    const F v = allpass.process_sample(buffer[read_frame]);
    read_frame = (read_frame + 1) % buffer.size();
    return v;
  }
  void write_and_advance(F input) {  // :202-205
    buffer[write_frame] = input;
    write_frame = (write_frame + 1) % buffer.size();
  }
  void set_delay_in_frames(F nf) {  // :160-174
    const F nf_floor = std::floor(nf);
    num_frames = static_cast<size_t>(nf_floor);
    F delta = nf - nf_floor;
    if (nf > F(0.5) && delta < F(0.5)) {
      delta += F(1);
      num_frames -= 1;
    }
    read_frame = write_frame >= num_frames ? write_frame - num_frames : buffer.size() - num_frames + write_frame;
    allpass.set_delta(delta);
  }
  void process(AudioCtx&, UGenFlags&, const F* in, F* out) override {  // :125-134
    out[0] = read();
    write_and_advance(in[0]);
  }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue v) override {  // :136-143
    if (index != 0) { ctx.rt_log("Unknown parameter set for AllpassDelay"); return; }
    const double delay_frames = v.float_or_panic() * static_cast<double>(ctx.sample_rate());
    const size_t as_usize = delay_frames > 0.0 ? (delay_frames >= 18446744073709551615.0 ? SIZE_MAX : static_cast<size_t>(delay_frames)) : 0;
    if (as_usize < buffer.size()) set_delay_in_frames(fnew<F>(delay_frames));
  }
};

// AllpassFeedbackDelay -- delay.rs:210-306
template <typename F>
struct AllpassFeedbackDelay : UGen<F> {
  F feedback = F(0);
  AllpassDelay<F> allpass_delay;
  explicit AllpassFeedbackDelay(Seconds max_delay) : allpass_delay(max_delay) {}
  size_t inputs() const override { return 1; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 2; }
  std::vector<std::string> param_descriptions() const override { return {"delay_time", "feedback"}; }
  void init(uint32_t sample_rate, size_t bs) override { allpass_delay.init(sample_rate, bs); }
  void process(AudioCtx&, UGenFlags&, const F* in, F* out) override {  // process_sample, :263-269
    const F delayed_sig = allpass_delay.read();
    const F delay_write = delayed_sig * feedback + in[0];
    allpass_delay.write_and_advance(delay_write);
    out[0] = delayed_sig - feedback * delay_write;
  }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue v) override {
    if (index == 0) {  // :231-236; the product ignores a delay that does not fit the ring (the reference indexes out of bounds)
      const double delay_frames = v.float_or_panic() * static_cast<double>(ctx.sample_rate());
      if (delay_frames < static_cast<double>(allpass_delay.buffer.size())) allpass_delay.set_delay_in_frames(fnew<F>(delay_frames));
      else ctx.rt_log("AllpassFeedbackDelay: delay_time longer than the buffer, ignored");
    } else if (index == 1) {
      feedback = fnew<F>(v.float_or_panic());
    } else {
      ctx.rt_log("Unknown parameter set for AllpassFeedbackDelay");
    }
  }
};

// ---------------------------------------------------------------------------
// SampleDelay -- knaster_core_dsp/src/ugens/delay.rs:14-50
// ---------------------------------------------------------------------------
template <typename F>
struct SampleDelay : UGen<F> {
  std::vector<F> buffer;
  size_t write_position = 0, delay_samples = 0;
  Seconds max_delay_length;
  explicit SampleDelay(Seconds max_delay) : max_delay_length(max_delay) {}
  size_t inputs() const override { return 1; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 1; }
  std::vector<std::string> param_descriptions() const override { return {"delay_time"}; }
  void init(uint32_t sample_rate, size_t) override {  // :45-49
    buffer.assign(static_cast<size_t>(max_delay_length.to_secs_f64() * static_cast<double>(sample_rate)), F(0));
    write_position = 0;
  }
  void process(AudioCtx&, UGenFlags&, const F* in, F* out) override {  // :37-44
    buffer[write_position] = in[0];
    out[0] = buffer[(write_position + buffer.size() - delay_samples) % buffer.size()];
    write_position = (write_position + 1) % buffer.size();
  }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue v) override {  // :33-36
    if (index != 0) { ctx.rt_log("Unknown parameter set for SampleDelay"); return; }
    const double d = v.float_or_panic() * static_cast<double>(ctx.sample_rate());
    const size_t n = d > 0.0 ? (d >= 18446744073709551615.0 ? SIZE_MAX : static_cast<size_t>(d)) : 0;  // `as usize` saturates
    if (n > buffer.size()) {  // the reference's index arithmetic would underflow here; the product ignores the change
      ctx.rt_log("SampleDelay: delay_time longer than the buffer, ignored");
      return;
    }
    delay_samples = n;
  }
};

// ---------------------------------------------------------------------------
// Constant -- knaster_core_dsp/src/ugens/util.rs:37-64
// ---------------------------------------------------------------------------
template <typename F>
struct Constant : UGen<F> {
  F value;
  explicit Constant(F v) : value(v) {}
  size_t inputs() const override { return 0; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 1; }
  std::vector<std::string> param_descriptions() const override { return {"value"}; }
  void process(AudioCtx&, UGenFlags&, const F*, F* out) override { out[0] = value; }
  void process_block(AudioCtx&, UGenFlags&, const BlockView<F>&, BlockView<F>& output) override {
    std::fill(output.channel(0), output.channel(0) + output.frames, value);
  }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue v) override {
    if (index == 0) value = fnew<F>(v.float_or_panic());
    else ctx.rt_log("Unknown parameter set for Constant");
  }
};

// ---------------------------------------------------------------------------
// MathUGen -- knaster_core_dsp/src/ugens/math.rs:17-165
// inputs a0..a{N-1}, b0..b{N-1}; out[ch] = a[ch] op b[ch]
// ---------------------------------------------------------------------------
enum class MathOp { Add, Sub, Mul, Div, Pow };
template <typename F>
inline F math_apply(MathOp op, F a, F b) {
  switch (op) {
    case MathOp::Add: return a + b;
    case MathOp::Sub: return a - b;
    case MathOp::Mul: return a * b;
    case MathOp::Div: return a / b;
    case MathOp::Pow: return std::pow(a, b);
  }
  return F(0);
}
template <typename F>
struct MathUGen : UGen<F> {
  size_t channels;
  MathOp op;
  MathUGen(size_t channels_, MathOp op_) : channels(channels_), op(op_) {}
  size_t inputs() const override { return channels * 2; }
  size_t outputs() const override { return channels; }
  size_t parameters() const override { return 0; }
  void process(AudioCtx&, UGenFlags&, const F* in, F* out) override {
    for (size_t c = 0; c < channels; ++c) out[c] = math_apply(op, in[c], in[c + channels]);
  }
  void process_block(AudioCtx&, UGenFlags&, const BlockView<F>& input, BlockView<F>& output) override {
    for (size_t c = 0; c < channels; ++c) {
      const F* a = input.channel(c);
      const F* b = input.channel(c + channels);
      F* o = output.channel(c);
      for (size_t i = 0; i < output.frames; ++i) o[i] = math_apply(op, a[i], b[i]);
    }
  }
  void param_apply(AudioCtx&, size_t, ParameterValue) override {}
};

// ---------------------------------------------------------------------------
// Test UGens -- knaster_core_dsp/src/test_utils.rs:8-86,
// knaster_graph/src/tests/utils.rs:4-67
// ---------------------------------------------------------------------------
template <typename F>
struct TestNumUGen : UGen<F> {
  F number;
  explicit TestNumUGen(F n) : number(n) {}
  size_t inputs() const override { return 0; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 0; }
  void process(AudioCtx&, UGenFlags&, const F*, F* out) override { out[0] = number; }
  void param_apply(AudioCtx&, size_t, ParameterValue) override {}
};
template <typename F>
struct TestInPlusParamUGen : UGen<F> {
  F number = 0;
  size_t inputs() const override { return 1; }
  size_t outputs() const override { return 1; }
  size_t parameters() const override { return 1; }
  std::vector<std::string> param_descriptions() const override { return {"number"}; }
  void process(AudioCtx&, UGenFlags&, const F* in, F* out) override { out[0] = number + in[0]; }
  void param_apply(AudioCtx&, size_t index, ParameterValue v) override {
    if (index == 0) number = fnew<F>(v.float_or_panic());
  }
};

// ---------------------------------------------------------------------------
// Math wrappers -- knaster_core_dsp/src/wrappers_core/math.rs:15-661
// All run the inner UGen, then transform every output sample in place.
// Only WrMul adds a parameter ("wr_mul", index = inner parameters()).
// ---------------------------------------------------------------------------
enum class WrOp { Mul, Add, Sub, VSub, Div, VDiv, Powf, Powi };
template <typename F>
struct WrMath : UGen<F> {
  UGenPtr<F> ugen;
  WrOp op;
  F value;
  int32_t ivalue = 0;
  WrMath(UGenPtr<F> inner, WrOp op_, F v) : ugen(std::move(inner)), op(op_), value(v) {}
  WrMath(UGenPtr<F> inner, int32_t exponent)
      : ugen(std::move(inner)), op(WrOp::Powi), value(0), ivalue(exponent) {}
  size_t inputs() const override { return ugen->inputs(); }
  size_t outputs() const override { return ugen->outputs(); }
  size_t parameters() const override { return ugen->parameters() + (op == WrOp::Mul ? 1 : 0); }
  std::vector<std::string> param_descriptions() const override {
    auto d = ugen->param_descriptions();
    d.resize(ugen->parameters());
    if (op == WrOp::Mul) d.push_back("wr_mul");
    return d;
  }
  void init(uint32_t sr, size_t bs) override { ugen->init(sr, bs); }
  F apply(F s) const {
    switch (op) {
      case WrOp::Mul: return s * value;
      case WrOp::Add: return s + value;
      case WrOp::Sub: return s - value;
      case WrOp::VSub: return value - s;
      case WrOp::Div: return s / value;
      case WrOp::VDiv: return value / s;
      case WrOp::Powf: return std::pow(s, value);
      case WrOp::Powi: {  // num-traits pow by squaring
        F base = s, acc = F(1);
        int32_t n = ivalue < 0 ? -ivalue : ivalue;
        bool first = true;
        while (n > 0) {
          if (n & 1) { acc = first ? base : acc * base; first = false; }
          n >>= 1;
          if (n) base = base * base;
        }
        return ivalue < 0 ? F(1) / acc : acc;
      }
    }
    return s;
  }
  void process(AudioCtx& ctx, UGenFlags& flags, const F* in, F* out) override {
    ugen->process(ctx, flags, in, out);
    for (size_t i = 0; i < outputs(); ++i) out[i] = apply(out[i]);
  }
  void process_block(AudioCtx& ctx, UGenFlags& flags, const BlockView<F>& input, BlockView<F>& output) override {
    ugen->process_block(ctx, flags, input, output);  // math.rs:62-67
    for (size_t c = 0; c < output.channels(); ++c) {
      F* o = output.channel(c);
      for (size_t i = 0; i < output.frames; ++i) o[i] = apply(o[i]);
    }
  }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue v) override {
    if (op == WrOp::Mul && index == ugen->parameters()) value = fnew<F>(v.float_or_panic());
    else ugen->param_apply(ctx, index, v);
  }
  void set_ar_param_buffer(AudioCtx& ctx, size_t index, const F* buffer) override {
    ugen->set_ar_param_buffer(ctx, index, buffer);
  }
  void set_delay_within_block_for_param(AudioCtx& ctx, size_t index, uint16_t delay) override {
    ugen->set_delay_within_block_for_param(ctx, index, delay);
  }
};

// ---------------------------------------------------------------------------
// WrArParams -- knaster_core_dsp/src/wrappers_core/audio_rate.rs:11-85
// Only `process` is defined -> default frame loop -> inner runs per sample.
// ---------------------------------------------------------------------------
template <typename F>
struct WrArParams : UGen<F> {
  UGenPtr<F> ugen;
  std::vector<const F*> buffers;
  size_t block_index = 0;
  explicit WrArParams(UGenPtr<F> inner) : ugen(std::move(inner)), buffers(ugen->parameters(), nullptr) {}
  size_t inputs() const override { return ugen->inputs(); }
  size_t outputs() const override { return ugen->outputs(); }
  size_t parameters() const override { return ugen->parameters(); }
  std::vector<std::string> param_descriptions() const override { return ugen->param_descriptions(); }
  void init(uint32_t sr, size_t bs) override { ugen->init(sr, bs); }
  void process(AudioCtx& ctx, UGenFlags& flags, const F* in, F* out) override {  // :42-57
    for (size_t p = 0; p < buffers.size(); ++p)
      if (buffers[p]) {
        PFloat value = static_cast<double>(buffers[p][block_index]);
        ugen->param_apply(ctx, p, ParameterValue::Flt(value));
      }
    block_index = (block_index + 1) % ctx.block_size();
    ugen->process(ctx, flags, in, out);
  }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue v) override {  // :70-74
    if (buffers[index] == nullptr) ugen->param_apply(ctx, index, v);
  }
  void set_ar_param_buffer(AudioCtx&, size_t index, const F* buffer) override {
    assert(index < buffers.size());
    buffers[index] = buffer;
  }
};

// ---------------------------------------------------------------------------
// WrPreciseTiming -- knaster_core_dsp/src/wrappers_core/precise_timing.rs:14-149
// ---------------------------------------------------------------------------
template <typename F>
struct WrPreciseTiming : UGen<F> {
  struct Change {
    bool some = false;
    uint16_t delay = 0;
    size_t index = 0;
    ParameterValue value;
  };
  UGenPtr<F> ugen;
  std::vector<Change> waiting_changes;  // capacity = DELAYED_CHANGES_PER_BLOCK
  std::vector<uint16_t> next_delay;     // per parameter; NOT reset after a block
  size_t next_delay_i = 0;
  WrPreciseTiming(size_t delayed_changes_per_block, UGenPtr<F> inner)
      : ugen(std::move(inner)), waiting_changes(delayed_changes_per_block), next_delay(ugen->parameters(), 0) {}
  size_t inputs() const override { return ugen->inputs(); }
  size_t outputs() const override { return ugen->outputs(); }
  size_t parameters() const override { return ugen->parameters(); }
  std::vector<std::string> param_descriptions() const override { return ugen->param_descriptions(); }
  void init(uint32_t sr, size_t bs) override { ugen->init(sr, bs); }
  void process(AudioCtx& ctx, UGenFlags& flags, const F* in, F* out) override {  // :50-64
    for (auto& wc : waiting_changes)
      if (wc.some) {
        wc.some = false;
        ugen->param_apply(ctx, wc.index, wc.value);
      }
    ugen->process(ctx, flags, in, out);
  }
  void process_block(AudioCtx& ctx, UGenFlags& flags, const BlockView<F>& input, BlockView<F>& output) override {
    // precise_timing.rs:65-114
    size_t block_i = 0, change_i = 0;
    const BlockMetadata org_block = ctx.block;
    const size_t num_changes_scheduled = next_delay_i;
    for (;;) {
      size_t local_frames_to_process = ctx.frames_to_process() - block_i;
      while (change_i < num_changes_scheduled) {
        Change& wc = waiting_changes[change_i];
        if (wc.some) {
          if (static_cast<size_t>(wc.delay) <= block_i + ctx.block_start_offset()) {
            ugen->param_apply(ctx, wc.index, wc.value);
            wc.some = false;
          } else {
            local_frames_to_process = std::min(
                local_frames_to_process, static_cast<size_t>(wc.delay) - ctx.block_start_offset() - block_i);
            break;
          }
        }
        change_i += 1;
      }
      if (block_i >= ctx.frames_to_process()) break;
      if (local_frames_to_process == ctx.frames_to_process()) {
        ugen->process_block(ctx, flags, input, output);
      } else {
        BlockView<F> pin = input.partial(block_i, local_frames_to_process);
        BlockView<F> pout = output.partial(block_i, local_frames_to_process);
        ctx.block = org_block.make_partial(block_i, local_frames_to_process);
        ugen->process_block(ctx, flags, pin, pout);
        ctx.block = org_block;
      }
      block_i += local_frames_to_process;
    }
    ctx.block = org_block;
    next_delay_i = 0;
  }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue v) override {  // :126-135
    if (next_delay[index] == 0) {
      ugen->param_apply(ctx, index, v);
    } else if (next_delay_i < waiting_changes.size()) {
      Change c;
      c.some = true;
      c.delay = next_delay[index];
      c.index = index;
      c.value = v;
      waiting_changes[next_delay_i] = c;
      next_delay_i += 1;
    } else {
      ctx.rt_log("Warning: Not enough space for scheduled changes in WrPreciseTiming, change ignored");
    }
  }
  void set_ar_param_buffer(AudioCtx& ctx, size_t index, const F* buffer) override {
    ugen->set_ar_param_buffer(ctx, index, buffer);
  }
  void set_delay_within_block_for_param(AudioCtx&, size_t index, uint16_t delay) override {
    next_delay[index] = delay;
  }
};

// ---------------------------------------------------------------------------
// WrSmoothParams -- knaster_core_dsp/src/wrappers_core/smooth_params.rs:12-311
// The per-parameter `Rate` table (`parameters`) is never written in the reference ("TODO: set the Rate
// of a parameter", :149), so the audio-rate branch of process_block is unreachable; only the
// block-rate path is restated.  Note next_value() advances by ctx.block_size() per process_block CALL,
// also when the call covers a partial block.
// ---------------------------------------------------------------------------
struct ParameterSmoothingState {
  bool linear = false;
  PFloat current_value = 0.0;  // None { current_value }
  PFloat start_value = 0.0, end_value = 0.0;
  size_t duration_frames = 0, frames_elapsed = 0;
  Rate rate = Rate::BlockRate;
  bool done = true;
  PFloat interpolated() const {
    PFloat current_mix = static_cast<PFloat>(frames_elapsed) / static_cast<PFloat>(duration_frames);
    return (end_value - start_value) * current_mix + start_value;
  }
  bool next_value(size_t block_size, size_t frame_in_block, PFloat* out) {  // :263-300
    if (!linear) return false;
    if (rate == Rate::BlockRate && frame_in_block != 0) return false;
    if (done) return false;
    PFloat current_value_ = interpolated();
    if (frames_elapsed == duration_frames) done = true;
    else if (rate == Rate::BlockRate) frames_elapsed = std::min(frames_elapsed + block_size, duration_frames);
    else frames_elapsed += 1;
    *out = current_value_;
    return true;
  }
};
template <typename F>
struct WrSmoothParams : UGen<F> {
  UGenPtr<F> ugen;
  std::vector<ParameterSmoothingState> smoothing_state;
  explicit WrSmoothParams(UGenPtr<F> inner) : ugen(std::move(inner)), smoothing_state(ugen->parameters()) {}
  size_t inputs() const override { return ugen->inputs(); }
  size_t outputs() const override { return ugen->outputs(); }
  size_t parameters() const override { return ugen->parameters(); }
  std::vector<std::string> param_descriptions() const override { return ugen->param_descriptions(); }
  void init(uint32_t sr, size_t bs) override { ugen->init(sr, bs); }
  void set_smoothing(size_t index, ParameterSmoothing smoothing, Rate new_rate, double sample_rate) {  // :30-102
    ParameterSmoothingState& st = smoothing_state[index];
    if (!smoothing.linear) {
      if (st.linear) {
        PFloat cv = st.interpolated();
        st = ParameterSmoothingState{};
        st.current_value = cv;
      }
      return;
    }
    size_t new_duration = static_cast<size_t>(static_cast<double>(smoothing.seconds) * sample_rate);
    if (!st.linear) {
      PFloat cv = st.current_value;
      st.linear = true;
      st.start_value = cv;
      st.end_value = cv;
      st.duration_frames = new_duration;
      st.frames_elapsed = 0;
      st.rate = new_rate;
      st.done = true;
    } else if (st.done) {
      st.start_value = st.end_value;
      st.duration_frames = new_duration;
      st.frames_elapsed = 0;
      st.rate = new_rate;
      st.done = true;
    } else {
      PFloat cv = st.interpolated();
      st.start_value = cv;
      st.duration_frames = new_duration;
      st.rate = new_rate;
      st.done = true;  // (sic) :96 -- the ramp in flight is frozen until the next value arrives
    }
  }
  void process(AudioCtx& ctx, UGenFlags& flags, const F* in, F* out) override {  // :116-131
    for (size_t j = 0; j < smoothing_state.size(); ++j) {
      PFloat v;
      if (smoothing_state[j].next_value(1, 0, &v)) ugen->param_apply(ctx, j, ParameterValue::Flt(v));
    }
    ugen->process(ctx, flags, in, out);
  }
  void process_block(AudioCtx& ctx, UGenFlags& flags, const BlockView<F>& input, BlockView<F>& output) override {
    for (size_t j = 0; j < smoothing_state.size(); ++j) {  // :188-197
      PFloat v;
      if (smoothing_state[j].next_value(ctx.block_size(), 0, &v)) ugen->param_apply(ctx, j, ParameterValue::Flt(v));
    }
    ugen->process_block(ctx, flags, input, output);
  }
  void param_apply(AudioCtx& ctx, size_t index, ParameterValue value) override {  // :210-259
    if (index >= ugen->parameters()) return;
    switch (value.kind) {
      case ParameterValue::Integer: case ParameterValue::Trigger: case ParameterValue::Bool:
        ugen->param_apply(ctx, index, value);
        break;
      case ParameterValue::Float: {
        ParameterSmoothingState& st = smoothing_state[index];
        if (!st.linear) {
          ugen->param_apply(ctx, index, value);
        } else {
          if (st.done) st.start_value = st.end_value;
          else st.start_value = st.interpolated();
          st.end_value = value.f;
          st.done = false;
          st.frames_elapsed = 0;
        }
      } break;
      case ParameterValue::Smoothing:
        set_smoothing(index, value.smoothing, value.rate, static_cast<double>(ctx.sample_rate()));
        break;
    }
  }
  // set_ar_param_buffer / set_delay_within_block_for_param are NOT forwarded by the reference wrapper
  // (they fall to the trait defaults, ugen.rs:322-341): WrPreciseTiming and WrArParams must sit outside.
};

// ---------------------------------------------------------------------------
// BufferAllocator -- knaster_graph/src/buffer_allocator.rs:50-166
// Works on offsets; the backing store is a std::vector owned by the Graph.
// ---------------------------------------------------------------------------
struct BufferAllocator {
  struct AllocatedBlock {
    size_t start_offset, len, outstanding_borrows;
  };
  size_t next_free_pos = 0;
  std::vector<AllocatedBlock> allocated_blocks;
  std::vector<size_t> return_order;
  size_t virtual_allocation_size = 0;
  size_t assign_new_block(size_t num_channels, size_t block_size, size_t num_borrows) {
    if (next_free_pos + num_channels * block_size > virtual_allocation_size)
      virtual_allocation_size = next_free_pos + num_channels * block_size;
    size_t start = next_free_pos;
    allocated_blocks.push_back({start, num_channels * block_size, num_borrows});
    next_free_pos += num_channels * block_size;
    return start;
  }
  void return_block(size_t start_offset) {
    for (size_t i = 0; i < allocated_blocks.size(); ++i)
      if (allocated_blocks[i].start_offset == start_offset) {
        allocated_blocks[i].outstanding_borrows -= 1;
        if (allocated_blocks[i].outstanding_borrows == 0) return_order.push_back(i);
        break;
      }
  }
  size_t get_block(size_t num_channels, size_t block_size, size_t num_borrows) {
    size_t len = num_channels * block_size;
    for (size_t k = 0; k < return_order.size(); ++k) {
      size_t i = return_order.size() - (k + 1);
      size_t index = return_order[i];
      if (allocated_blocks[index].outstanding_borrows == 0 && allocated_blocks[index].len >= len) {
        allocated_blocks[index].outstanding_borrows = num_borrows;
        return_order.erase(return_order.begin() + static_cast<long>(i));
        return allocated_blocks[index].start_offset;
      }
    }
    return assign_new_block(num_channels, block_size, num_borrows);
  }
  void reset(size_t block_size) {
    next_free_pos = 0;
    virtual_allocation_size = 0;
    allocated_blocks.clear();
    return_order.clear();
    get_block(1, block_size, SIZE_MAX);  // the shared zero channel at offset 0
  }
};

// ---------------------------------------------------------------------------
// Scheduling -- knaster_graph/src/scheduling.rs:29-139
// ---------------------------------------------------------------------------
struct Time {
  Seconds seconds;
  bool absolute = false;
  static Time at(Seconds s) { return Time{s, true}; }
  static Time after(Seconds s) { return Time{s, false}; }
  static Time asap() { return Time{Seconds::zero(), false}; }
  uint64_t to_samples_until_due(uint64_t block_size, uint64_t sample_rate, uint64_t frame_clock) {
    if (absolute) {
      uint64_t t = seconds.to_samples(sample_rate);
      return t >= frame_clock ? t - frame_clock : 0;  // saturating_sub
    }
    if (seconds == Seconds::zero()) return 0;
    uint64_t samples = seconds.to_samples(sample_rate);
    seconds = seconds.saturating_sub(Seconds::from_samples(block_size, sample_rate));
    return samples;
  }
};
using NodeKey = size_t;
constexpr NodeKey GRAPH_KEY = SIZE_MAX;
struct SchedulingEvent {
  NodeKey node_key = 0;
  size_t parameter = 0;
  bool has_value = false;
  ParameterValue value;
  bool has_time = false;
  Time time;
};

// ---------------------------------------------------------------------------
// Graph + GraphGen + AudioProcessor
//   knaster_graph/src/graph.rs:768-881 (additive connects), :1588-1704
//   (allocate_node_buffers), :1938-2067 (node order), task.rs:17-32,
//   graph_gen.rs:77-305, processor.rs:47-197.
// Feedback edges, subgraphs, node freeing and the lock-free rings are out of
// scope (control plane); commit() applies the new schedule synchronously.
// ---------------------------------------------------------------------------
struct Edge {
  bool some = false;
  NodeKey source = GRAPH_KEY;  // GRAPH_KEY = graph input
  uint16_t channel_in_source = 0;
};
struct ParameterEdge {
  NodeKey source;
  uint16_t channel_in_source;
  uint16_t parameter_index;
};

template <typename F>
struct Task {
  UGen<F>* ugen;
  std::vector<const F*> in_buffers;
  F* out_buffer;
  size_t output_channels;
  void run(AudioCtx& ctx, UGenFlags& flags) {  // task.rs:25-31
    BlockView<F> input;
    input.frames = ctx.block_size();
    for (const F* p : in_buffers) input.ch.push_back(const_cast<F*>(p));
    BlockView<F> output(out_buffer, output_channels, ctx.block_size());
    ugen->process_block(ctx, flags, input, output);
  }
};

template <typename F>
class Graph {
 public:
  struct Node {
    UGenPtr<F> ugen;
    size_t inputs, outputs;
    size_t num_output_dependents = 0;
    size_t output_offset = 0;
    bool auto_math_node = false;
    bool alive = true;
  };
  Graph(size_t inputs, size_t outputs, size_t block_size, uint32_t sample_rate)
      : num_inputs(inputs), num_outputs(outputs), block_size_(block_size), sample_rate_(sample_rate),
        output_edges(outputs), ctx(sample_rate, block_size) {
    blocks_to_keep_scheduled_changes = sample_rate / static_cast<uint32_t>(block_size);
  }
  // graph.rs:462-475: init() runs at push time
  NodeKey push(UGenPtr<F> ugen) {
    ugen->init(sample_rate_, block_size_);
    Node n;
    n.inputs = ugen->inputs();
    n.outputs = ugen->outputs();
    n.ugen = std::move(ugen);
    nodes.push_back(std::move(n));
    node_input_edges.emplace_back(nodes.back().inputs);
    node_parameter_edges.emplace_back();
    recalculation_required = true;
    return nodes.size() - 1;
  }
  UGen<F>* ugen(NodeKey k) { return nodes[k].ugen.get(); }
  // graph.rs:768-822
  void connect_to_node(NodeKey source, uint16_t so_channel, uint16_t si_channel, NodeKey sink, bool additive) {
    recalculation_required = true;
    Edge e{true, source, so_channel};
    if (!additive || !node_input_edges[sink][si_channel].some) {
      node_input_edges[sink][si_channel] = e;
      return;
    }
    Edge existing = node_input_edges[sink][si_channel];
    NodeKey add_node = new_additive_node();
    node_input_edges[add_node][0] = existing;
    node_input_edges[add_node][1] = e;
    node_input_edges[sink][si_channel] = Edge{true, add_node, 0};
  }
  // graph.rs:827-872
  void connect_to_output(NodeKey source, uint16_t so_channel, uint16_t si_channel, bool additive) {
    recalculation_required = true;
    Edge e{true, source, so_channel};
    if (!additive || !output_edges[si_channel].some) {
      output_edges[si_channel] = e;
      return;
    }
    Edge existing = output_edges[si_channel];
    NodeKey add_node = new_additive_node();
    node_input_edges[add_node][0] = existing;
    node_input_edges[add_node][1] = e;
    output_edges[si_channel] = Edge{true, add_node, 0};
  }
  // graph.rs:629-726 (replace form): audio-rate parameter edge
  void connect_to_parameter(NodeKey source, uint16_t so_channel, uint16_t parameter, NodeKey sink) {
    recalculation_required = true;
    auto& pe = node_parameter_edges[sink];
    for (auto& e : pe)
      if (e.parameter_index == parameter) {
        e.source = source;
        e.channel_in_source = so_channel;
        return;
      }
    pe.push_back(ParameterEdge{source, so_channel, parameter});
  }
  void disconnect_output_from_source(NodeKey source, uint16_t so_channel) {
    recalculation_required = true;
    for (auto& edges : node_input_edges)
      for (auto& e : edges)
        if (e.some && e.source == source && e.channel_in_source == so_channel) e.some = false;
    for (auto& e : output_edges)
      if (e.some && e.source == source && e.channel_in_source == so_channel) e.some = false;
  }
  void disconnect_input_to_sink(uint16_t si_channel, NodeKey sink) {
    recalculation_required = true;
    node_input_edges[sink][si_channel].some = false;
  }
  // `sig * c` / `sig + c`: graph_edit.rs:1036-1066 pushes Constant first, then the MathUGen,
  // connects ch0 <- sig, ch1 <- constant.
  NodeKey math_with_constant(NodeKey sig, uint16_t sig_channel, MathOp op, F c) {
    NodeKey cn = push(std::make_unique<Constant<F>>(c));
    NodeKey m = push(std::make_unique<MathUGen<F>>(1, op));
    connect_to_node(sig, sig_channel, 0, m, false);
    connect_to_node(cn, 0, 1, m, false);
    return m;
  }
  // `a * b` between two single-channel sources (graph_edit.rs:936-971)
  NodeKey math_nodes(NodeKey a, uint16_t a_ch, MathOp op, NodeKey b, uint16_t b_ch) {
    NodeKey m = push(std::make_unique<MathUGen<F>>(1, op));
    connect_to_node(a, a_ch, 0, m, false);
    connect_to_node(b, b_ch, 1, m, false);
    return m;
  }

  // param.set(v) / set_at(v, t): graph_edit.rs:1708-1753
  void schedule(const SchedulingEvent& ev) { scheduling_queue.push_back(ev); }
  void set(NodeKey node, size_t param, ParameterValue v) {
    SchedulingEvent ev;
    ev.node_key = node;
    ev.parameter = param;
    ev.has_value = true;
    ev.value = v;
    schedule(ev);
  }
  void set_at(NodeKey node, size_t param, ParameterValue v, Time t) {
    SchedulingEvent ev;
    ev.node_key = node;
    ev.parameter = param;
    ev.has_value = true;
    ev.value = v;
    ev.has_time = true;
    ev.time = t;
    schedule(ev);
  }

  // graph.rs:1707-1726
  void commit_changes() {
    if (!recalculation_required) return;
    calculate_node_order();
    allocate_node_buffers();
    generate_tasks();
    recalculation_required = false;
  }

  // AudioProcessor::run_raw_ptr_inputs, processor.rs:142-179 + GraphGen::process_block
  void run(const std::vector<const F*>& input_pointers, F* output /* [outputs][block_size] */) {
    assert(input_pointers.size() == num_inputs);
    ctx.block = BlockMetadata{};
    ctx.block.frames_to_process = block_size_;
    ctx.block.frame_clock = frame_clock;
    // (ii) parameter changes, graph_gen.rs:110-166
    size_t n_waiting = waiting_parameter_changes.size();
    for (size_t i = 0; i < n_waiting; ++i) {
      auto [event, num_blocks_waiting] = waiting_parameter_changes.front();
      waiting_parameter_changes.pop_front();
      if (num_blocks_waiting > blocks_to_keep_scheduled_changes) continue;
      SchedulingEvent un;
      if (!apply_parameter_change(event, &un)) waiting_parameter_changes.emplace_back(un, num_blocks_waiting + 1);
    }
    for (auto& ev : scheduling_queue) {
      SchedulingEvent un;
      if (!apply_parameter_change(ev, &un)) waiting_parameter_changes.emplace_back(un, 0);
    }
    scheduling_queue.clear();
    // (iii) patch graph-input pointers, graph_gen.rs:187-194
    for (auto& [task_index, pairs] : graph_input_channels_to_nodes)
      for (auto& [graph_input, node_input] : pairs) tasks[task_index].in_buffers[node_input] = input_pointers[graph_input];
    // (iv) HOT LOOP graph_gen.rs:196-200
    UGenFlags new_flags;
    for (auto& task : tasks) task.run(ctx, new_flags);
    last_flags = new_flags;
    // (v) outputs graph_gen.rs:202-224
    for (size_t c = 0; c < num_outputs; ++c) {
      F* out_channel = output + c * block_size_;
      const Edge& e = output_edges[c];
      if (e.some) {
        const F* src = e.source == GRAPH_KEY ? input_pointers[e.channel_in_source]
                                             : buffer.data() + nodes[e.source].output_offset +
                                                   static_cast<size_t>(e.channel_in_source) * block_size_;
        std::memcpy(out_channel, src, block_size_ * sizeof(F));
      } else {
        std::fill(out_channel, out_channel + block_size_, F(0));
      }
    }
    frame_clock += block_size_;
  }
  const std::vector<NodeKey>& order() const { return node_order; }
  size_t buffer_len() const { return buffer.size(); }
  size_t num_tasks() const { return tasks.size(); }
  AudioCtx& audio_ctx() { return ctx; }
  UGenFlags last_flags;
  uint64_t frame_clock = 0;

 private:
  NodeKey new_additive_node() {  // graph.rs:874-881
    NodeKey k = push(std::make_unique<MathUGen<F>>(1, MathOp::Add));
    nodes[k].auto_math_node = true;
    return k;
  }
  // graph.rs:1938-1980
  std::vector<NodeKey> depth_first_search(std::unordered_set<NodeKey>& visited, std::vector<NodeKey>& nodes_to_process) {
    std::vector<NodeKey> order;
    while (!nodes_to_process.empty()) {
      NodeKey node_key = nodes_to_process.back();
      bool found_unvisited = false;
      for (const Edge& e : node_input_edges[node_key]) {
        if (!e.some) continue;
        if (e.source != GRAPH_KEY && !visited.count(e.source)) {
          nodes_to_process.push_back(e.source);
          visited.insert(e.source);
          found_unvisited = true;
          break;
        }
      }
      if (!found_unvisited) {
        for (const ParameterEdge& e : node_parameter_edges[node_key]) {
          if (!visited.count(e.source)) {
            nodes_to_process.push_back(e.source);
            visited.insert(e.source);
            found_unvisited = true;
            break;
          }
        }
      }
      if (!found_unvisited) {
        order.push_back(nodes_to_process.back());
        nodes_to_process.pop_back();
      }
    }
    return order;
  }
  // graph.rs:1984-2018
  NodeKey get_deepest_output_node(NodeKey start_node, const std::unordered_set<NodeKey>& visited) {
    NodeKey last_connected = start_node, last_connected_output = start_node;
    for (;;) {
      bool found_later = false;
      for (NodeKey key = 0; key < node_input_edges.size() && !found_later; ++key) {
        for (const Edge& ie : node_input_edges[key]) {
          if (!ie.some || ie.source == GRAPH_KEY) continue;
          if (ie.source == last_connected && !visited.count(ie.source)) {
            last_connected = key;
            found_later = true;
            for (const Edge& oe : output_edges)
              if (oe.some && oe.source != GRAPH_KEY && oe.source == last_connected) last_connected_output = last_connected;
            break;
          }
        }
      }
      if (!found_later) break;
    }
    return last_connected_output;
  }
  // graph.rs:2022-2067
  void calculate_node_order() {
    node_order.clear();
    std::unordered_set<NodeKey> visited;
    std::vector<NodeKey> nodes_to_process;
    for (const Edge& e : output_edges) {
      if (!e.some || e.source == GRAPH_KEY) continue;
      NodeKey deepest = get_deepest_output_node(e.source, visited);
      if (!visited.count(deepest)) {
        nodes_to_process.push_back(deepest);
        visited.insert(deepest);
      }
    }
    auto stack = depth_first_search(visited, nodes_to_process);
    node_order.insert(node_order.end(), stack.begin(), stack.end());
    for (NodeKey k = 0; k < nodes.size(); ++k)
      if (!visited.count(k)) node_order.push_back(k);
  }
  // graph.rs:1588-1704
  void allocate_node_buffers() {
    for (auto& n : nodes) n.num_output_dependents = 0;
    for (auto& edges : node_input_edges)
      for (const Edge& e : edges)
        if (e.some && e.source != GRAPH_KEY) nodes[e.source].num_output_dependents += 1;
    for (auto& edges : node_parameter_edges)
      for (const ParameterEdge& e : edges) nodes[e.source].num_output_dependents += 1;
    graph_input_channels_to_nodes.clear();
    allocator.reset(block_size_);
    for (size_t order_index = 0; order_index < node_order.size(); ++order_index) {
      NodeKey key = node_order[order_index];
      size_t offset = allocator.get_block(nodes[key].outputs, block_size_, nodes[key].num_output_dependents);
      nodes[key].output_offset = offset;
      std::vector<std::pair<size_t, size_t>> from_graph;
      for (size_t ch = 0; ch < node_input_edges[key].size(); ++ch) {
        const Edge& e = node_input_edges[key][ch];
        if (!e.some) continue;
        if (e.source != GRAPH_KEY) allocator.return_block(nodes[e.source].output_offset);
        else from_graph.emplace_back(e.channel_in_source, ch);
      }
      for (const ParameterEdge& e : node_parameter_edges[key]) allocator.return_block(nodes[e.source].output_offset);
      if (!from_graph.empty()) graph_input_channels_to_nodes.emplace_back(order_index, from_graph);
    }
    if (allocator.virtual_allocation_size > buffer.size()) buffer.assign(allocator.virtual_allocation_size, F(0));
    // offset 0..block_size is the shared zero channel: keep it cleared
    std::fill(buffer.begin(), buffer.begin() + static_cast<long>(block_size_), F(0));
  }
  // graph.rs:1497-1562 generate_tasks + generate_ar_parameter_changes; task.rs:101-131
  void generate_tasks() {
    tasks.clear();
    for (NodeKey key : node_order) {
      Task<F> t;
      t.ugen = nodes[key].ugen.get();
      t.out_buffer = buffer.data() + nodes[key].output_offset;
      t.output_channels = nodes[key].outputs;
      t.in_buffers.assign(nodes[key].inputs, buffer.data());  // zero channel by default
      for (size_t ch = 0; ch < node_input_edges[key].size(); ++ch) {
        const Edge& e = node_input_edges[key][ch];
        if (e.some && e.source != GRAPH_KEY)
          t.in_buffers[ch] = buffer.data() + nodes[e.source].output_offset +
                             static_cast<size_t>(e.channel_in_source) * block_size_;
      }
      tasks.push_back(std::move(t));
    }
    for (NodeKey key = 0; key < nodes.size(); ++key)
      for (const ParameterEdge& e : node_parameter_edges[key]) {
        const F* buf = buffer.data() + nodes[e.source].output_offset +
                       static_cast<size_t>(e.channel_in_source) * block_size_;
        nodes[key].ugen->set_ar_param_buffer(ctx, e.parameter_index, buf);
      }
  }
  // graph_gen.rs:269-305.  Returns true if applied.
  bool apply_parameter_change(SchedulingEvent event, SchedulingEvent* unapplied) {
    bool ready = true;
    uint64_t delay_in_block = 0;
    if (event.has_time) {
      delay_in_block = event.time.to_samples_until_due(block_size_, sample_rate_, ctx.frame_clock());
      ready = ready && (delay_in_block < block_size_);
    }
    if (ready) {
      for (NodeKey key : node_order) {
        if (key == event.node_key) {
          UGen<F>* g = nodes[key].ugen.get();
          if (delay_in_block > 0)
            g->set_delay_within_block_for_param(ctx, event.parameter, static_cast<uint16_t>(delay_in_block));
          if (event.has_value) g->param_apply(ctx, event.parameter, event.value);
          return true;
        }
      }
    }
    *unapplied = event;
    return false;
  }

  size_t num_inputs, num_outputs, block_size_;
  uint32_t sample_rate_;
  std::vector<Node> nodes;
  std::vector<std::vector<Edge>> node_input_edges;
  std::vector<std::vector<ParameterEdge>> node_parameter_edges;
  std::vector<Edge> output_edges;
  std::vector<NodeKey> node_order;
  BufferAllocator allocator;
  std::vector<F> buffer;
  std::vector<Task<F>> tasks;
  std::vector<std::pair<size_t, std::vector<std::pair<size_t, size_t>>>> graph_input_channels_to_nodes;
  std::vector<SchedulingEvent> scheduling_queue;
  std::deque<std::pair<SchedulingEvent, uint32_t>> waiting_parameter_changes;
  uint32_t blocks_to_keep_scheduled_changes;
  bool recalculation_required = true;
  AudioCtx ctx;
};

}  // namespace kno
