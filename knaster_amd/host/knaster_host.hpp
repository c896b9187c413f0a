// knaster_host.hpp -- C++ host-side mirror of Knaster's graph API for the accelerated path.
//
// The reference host is Rust; this image has no Rust toolchain, so the layer above the C ABI
// (include/knaster_hip.h) is written in C++ with the reference's names and argument meaning:
//
//   auto [graph, processor] = AudioProcessor<float>::create(/*outputs*/ 2, {.block_size = 64});
//   graph.edit([&](GraphEdit<float>& g) {                 // knaster_graph/src/graph.rs:1410
//     auto s = g.push(SinWt(440.f));                      // graph_edit.rs:88
//     (s * 0.2f).out({0, 0}).to_graph_out();              // graph_edit.rs:1170-1207, 280-292, 363-369
//   });                                                   // commit on scope exit (graph_edit.rs:258-262)
//   processor.run_without_inputs();                       // processor.rs:142-179
//   float l = processor.output_block().read(0, 0);
//
// At commit, every signal connected to the graph output is traced back to its source; voices whose
// chains have the same shape are gathered into one knh_bank (one fused gfx950 kernel per shape)
// instead of ~6 nodes per voice.  Parameter handles (`node.param("freq").set(v)`, `.set_at(v, t)`,
// `.trig()`) and `carrier.link("freq", signal)` keep their meaning; GraphGen's event handling
// (graph_gen.rs:110-166, scheduling.rs:95-121) is restated in `AudioProcessor::run_*`.
// A voice may be a small graph, not only a chain: `a * b` of two oscillators, one signal feeding two filters
// (knh_stage_desc.input / KNH_STAGE_MATH_*).  Anything that is not a supported per-voice graph is rejected at
// commit (the reference would run it on the CPU; that is outside this engine).  Header-only; needs libknaster_hip.so.
#pragma once
#include <algorithm>
#include <atomic>
#include <array>
#include <cmath>
#include <cstdint>
#include <deque>
#include <functional>
#include <initializer_list>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/knaster_hip.h"

namespace knaster {

// ---- Seconds / Time: knaster_primitives/src/time.rs:25-134, knaster_graph/src/scheduling.rs:73-139 ----
struct Seconds {
  static constexpr uint32_t kTesimalsPerSecond = 282240000u;
  uint32_t seconds = 0, subsecond_tesimals = 0;
  static Seconds zero() { return {}; }
  static Seconds from_secs_f64(double s) {
    Seconds r;
    r.seconds = static_cast<uint32_t>(std::floor(s));
    r.subsecond_tesimals = static_cast<uint32_t>((s - std::trunc(s)) * static_cast<double>(kTesimalsPerSecond));
    return r;
  }
  static Seconds from_samples(uint64_t samples, uint64_t sample_rate) {
    return {static_cast<uint32_t>(samples / sample_rate),
            static_cast<uint32_t>((samples % sample_rate) * kTesimalsPerSecond / sample_rate)};
  }
  uint64_t to_samples(uint64_t sample_rate) const {
    return static_cast<uint64_t>(seconds) * sample_rate + (static_cast<uint64_t>(subsecond_tesimals) * sample_rate) / kTesimalsPerSecond;
  }
  bool operator==(const Seconds& o) const { return seconds == o.seconds && subsecond_tesimals == o.subsecond_tesimals; }
  bool le(const Seconds& o) const { return seconds == o.seconds ? subsecond_tesimals <= o.subsecond_tesimals : seconds < o.seconds; }
  Seconds saturating_sub(const Seconds& r) const {
    if (le(r)) return zero();
    if (subsecond_tesimals >= r.subsecond_tesimals) return {seconds - r.seconds, subsecond_tesimals - r.subsecond_tesimals};
    return {seconds - r.seconds - 1, kTesimalsPerSecond - (r.subsecond_tesimals - subsecond_tesimals)};
  }
};
struct Time {
  Seconds seconds;
  bool absolute = false;
  static Time at(Seconds s) { return {s, true}; }
  static Time after(Seconds s) { return {s, false}; }
  static Time asap() { return {Seconds::zero(), false}; }
  uint64_t to_samples_until_due(uint64_t block_size, uint64_t sample_rate, uint64_t frame_clock) {
    if (absolute) {
      uint64_t t = seconds.to_samples(sample_rate);
      return t >= frame_clock ? t - frame_clock : 0;
    }
    if (seconds == Seconds::zero()) return 0;
    uint64_t samples = seconds.to_samples(sample_rate);
    seconds = seconds.saturating_sub(Seconds::from_samples(block_size, sample_rate));
    return samples;
  }
};
struct PTrigger {};

// ---- UGen descriptions (constructor arguments only; the DSP lives in the fused kernels) ----------
enum class SvfFilterType : uint8_t { Low = 0, High, Band, Notch, Peak, All, Bell, LowShelf, HighShelf };

struct UGenSpec {
  UGenSpec() = default;
  UGenSpec(uint16_t kind_, std::vector<double> args_) : kind(kind_), args(std::move(args_)) {}
  uint16_t kind = 0;                 // knh_stage_kind of the core UGen
  std::vector<double> args;          // constructor arguments
  std::vector<std::pair<uint16_t, double>> wrappers;  // (KNH_STAGE_WR_*, value) in application order
  bool ar_params_ = false, smooth_params_ = false;
  uint16_t precise_timing_ = 0;
  bool is_constant = false, is_env = false;
  uint16_t outputs = 1;              // UGen::Outputs (Pan2: 2)
  // wrappers_core.rs:26-111
  UGenSpec wr_mul(double v) && { wrappers.emplace_back(KNH_STAGE_WR_MUL, v); return std::move(*this); }
  UGenSpec wr_add(double v) && { wrappers.emplace_back(KNH_STAGE_WR_ADD, v); return std::move(*this); }
  UGenSpec wr_sub(double v) && { wrappers.emplace_back(KNH_STAGE_WR_SUB, v); return std::move(*this); }
  UGenSpec wr_v_sub_gen(double v) && { wrappers.emplace_back(KNH_STAGE_WR_VSUB, v); return std::move(*this); }
  UGenSpec wr_div(double v) && { wrappers.emplace_back(KNH_STAGE_WR_DIV, v); return std::move(*this); }
  UGenSpec wr_v_div_gen(double v) && { wrappers.emplace_back(KNH_STAGE_WR_VDIV, v); return std::move(*this); }
  UGenSpec wr_powf(double v) && { wrappers.emplace_back(KNH_STAGE_WR_POWF, v); return std::move(*this); }
  UGenSpec wr_powi(int32_t n) && { wrappers.emplace_back(KNH_STAGE_WR_POWI, static_cast<double>(n)); return std::move(*this); }
  UGenSpec ar_params() && { ar_params_ = true; return std::move(*this); }
  UGenSpec smooth_params() && { smooth_params_ = true; return std::move(*this); }
  UGenSpec precise_timing(uint16_t max_changes_per_block) && { precise_timing_ = max_changes_per_block; return std::move(*this); }
};
inline UGenSpec SinWt(double freq) { return UGenSpec(KNH_STAGE_SIN_WT, {freq}); }
inline UGenSpec SinNumeric(double freq) { return UGenSpec(KNH_STAGE_SIN_NUMERIC, {freq}); }
// PolyBlep::new(waveform, freq) -- polyblep.rs:136-145; Waveform in the reference's order (Sawtooth = 0 ... TrapezoidVariable = 13)
inline UGenSpec PolyBlep(int waveform, double freq) { return UGenSpec(KNH_STAGE_POLYBLEP, {static_cast<double>(waveform), freq}); }
inline UGenSpec Phasor(double freq) { return UGenSpec(KNH_STAGE_PHASOR, {freq}); }        // osc.rs:172-214
inline UGenSpec SafetyLimiter() { return UGenSpec(KNH_STAGE_SAFETY_LIMITER, {}); }        // dynamics.rs:9-31
// Pan2::new(pan) -- pan.rs:18-23: one input, two outputs; `(voice >> pan).to_graph_out()` sends them to graph outputs 0 and 1
inline UGenSpec Pan2(double pan) { UGenSpec s(KNH_STAGE_PAN2, {pan}); s.outputs = 2; return s; }
// noise.rs:11-22: every randomness UGen takes its seed from one process-wide counter, in construction order, so a
// graph built in the same order makes the same noise.  WhiteNoise / PinkNoise / BrownNoise::new() -- noise.rs:33,65,133
inline uint64_t next_randomness_seed() {
  static std::atomic<uint64_t> next{0};
  return next.fetch_add(1, std::memory_order_seq_cst);
}
inline UGenSpec WhiteNoise() { return UGenSpec(KNH_STAGE_WHITE_NOISE, {static_cast<double>(next_randomness_seed())}); }
inline UGenSpec PinkNoise() { return UGenSpec(KNH_STAGE_PINK_NOISE, {static_cast<double>(next_randomness_seed())}); }
inline UGenSpec BrownNoise() { return UGenSpec(KNH_STAGE_BROWN_NOISE, {static_cast<double>(next_randomness_seed())}); }
inline UGenSpec RandomLin(double freq) { return UGenSpec(KNH_STAGE_RANDOM_LIN, {static_cast<double>(next_randomness_seed()), freq}); }  // noise.rs:172
inline UGenSpec SvfFilter(SvfFilterType ty, double cutoff, double q, double gain_db) {
  return UGenSpec(KNH_STAGE_SVF, {static_cast<double>(ty), cutoff, q, gain_db});
}
inline UGenSpec OnePoleLpf(double cutoff) { return UGenSpec(KNH_STAGE_ONEPOLE_LPF, {cutoff}); }
inline UGenSpec OnePoleHpf() { return UGenSpec(KNH_STAGE_ONEPOLE_HPF, {}); }
// SampleDelay::new(Seconds::from_secs_f64(max_delay_seconds)) -- delay.rs:24-31
inline UGenSpec AllpassDelay(double max_delay_seconds) { return UGenSpec(KNH_STAGE_ALLPASS_DELAY, {max_delay_seconds}); }  // delay.rs:107-117
inline UGenSpec AllpassFeedbackDelay(double max_delay_seconds) { return UGenSpec(KNH_STAGE_ALLPASS_FB_DELAY, {max_delay_seconds}); }  // delay.rs:221-229
inline UGenSpec SampleDelay(double max_delay_seconds) { return UGenSpec(KNH_STAGE_SAMPLE_DELAY, {max_delay_seconds}); }
inline UGenSpec EnvAsr(double attack, double release) { UGenSpec s(KNH_STAGE_MUL_ENV_ASR, {attack, release}); s.is_env = true; return s; }
inline UGenSpec EnvAr(double attack, double release) { UGenSpec s(KNH_STAGE_MUL_ENV_AR, {attack, release}); s.is_env = true; return s; }
// Envelope::new(start_value, segments).time_scale(..).looping(..) -- envelopes.rs:373-400
struct EnvelopeSegment { double duration, value; };
inline UGenSpec Envelope(double start_value, const std::vector<EnvelopeSegment>& segments, double time_scale = 1.0, bool looping = false) {
  if (segments.empty()) throw std::runtime_error("Envelope needs at least one segment");
  std::vector<double> a{start_value, time_scale, looping ? 1.0 : 0.0, static_cast<double>(segments.size())};
  for (const EnvelopeSegment& sg : segments) { a.push_back(sg.duration); a.push_back(sg.value); }
  UGenSpec s(KNH_STAGE_MUL_ENVELOPE, std::move(a));
  s.is_env = true;
  return s;
}
inline UGenSpec Constant(double value) { UGenSpec s(KNH_STAGE_MUL_CONST, {value}); s.is_constant = true; return s; }

struct GraphError : std::runtime_error { using std::runtime_error::runtime_error; };
enum class ParameterError { Ok = 0, ParameterIndexOutOfBounds, DescriptionNotFound };

template <typename F> class Graph;
template <typename F> class GraphEdit;
template <typename F> class AudioProcessor;

// A node of the (host-side) graph table.
struct NodeRec {
  enum Type { UGEN, MATH } type = UGEN;
  UGenSpec spec;            // UGEN
  uint16_t math_kind = 0;   // MATH: KNH_STAGE_{MUL,ADD,SUB,DIV}_CONST for op with a Constant; 0xFFFF = signal (op) signal
  uint16_t math2_kind = KNH_STAGE_MATH_MUL;  // MATH of two signals: KNH_STAGE_MATH_{ADD,SUB,MUL,DIV,POW}
  int in0 = -1, in1 = -1;   // input edges (node ids)
  int link_source = -1, link_param = -1;  // audio-rate parameter edge (graph_edit.rs:735-754)
  // where the node ended up after commit
  int bank = -1, voice = -1, stage = -1;
};

struct SchedulingEvent {  // scheduling.rs:29-36
  int node = -1;
  uint32_t parameter = 0;
  uint32_t kind = KNH_VALUE_FLOAT;
  double f = 0;
  int64_t i = 0;
  bool has_time = false;
  Time time;
};

// What a voice chain looks like to the planner: the stage list plus the node behind every stage.
struct ChainPlan {
  std::vector<knh_stage_desc> stages;
  std::vector<int> stage_node;                  // node id whose parameters stage s exposes (-1: none)
  std::vector<std::vector<double>> stage_args;  // constructor arguments per stage
  std::string signature() const {
    std::string s;
    for (auto& st : stages) {
      s += std::to_string(st.kind) + "." + std::to_string(st.flags) + "." + std::to_string(st.delayed_changes_per_block) + "." +
           std::to_string(st.input) + "." + std::to_string(st.input2) + "." + std::to_string(st.ar_param) + ";";
    }
    return s;
  }
};

// ---- signal handles --------------------------------------------------------------------------------
template <typename F>
class Sig {  // SH / DH of graph_edit.rs:266-277: one or more output channels of nodes
 public:
  Sig(GraphEdit<F>* g, std::vector<int> nodes) : g_(g), nodes_(std::move(nodes)), chans_(nodes_.size(), 0) {}
  Sig(GraphEdit<F>* g, std::vector<int> nodes, std::vector<int> chans) : g_(g), nodes_(std::move(nodes)), chans_(std::move(chans)) {}
  Sig operator*(double c) const { return g_->math_const(*this, KNH_STAGE_MUL_CONST, c); }
  Sig operator+(double c) const { return g_->math_const(*this, KNH_STAGE_ADD_CONST, c); }
  Sig operator-(double c) const { return g_->math_const(*this, KNH_STAGE_SUB_CONST, c); }
  Sig operator/(double c) const { return g_->math_const(*this, KNH_STAGE_DIV_CONST, c); }
  Sig pow(double c) const { return g_->math_const(*this, KNH_STAGE_POW_CONST, c); }  // graph_edit.rs:451 with a Constant
  // two signals: MathUGen<_, U1, Op> with both inputs connected (graph_edit.rs:936-971)
  Sig operator*(const Sig& o) const { return g_->math_sig(*this, o, KNH_STAGE_MATH_MUL); }
  Sig operator+(const Sig& o) const { return g_->math_sig(*this, o, KNH_STAGE_MATH_ADD); }
  Sig operator-(const Sig& o) const { return g_->math_sig(*this, o, KNH_STAGE_MATH_SUB); }
  Sig operator/(const Sig& o) const { return g_->math_sig(*this, o, KNH_STAGE_MATH_DIV); }
  Sig pow(const Sig& o) const { return g_->math_sig(*this, o, KNH_STAGE_MATH_POW); }
  Sig operator>>(const Sig& sink) const { return g_->connect(*this, sink); }  // graph_edit.rs:1347-1417
  // .out([0,0]): the same channel twice (graph_edit.rs:280-292)
  Sig out(std::initializer_list<int> channels) const {
    std::vector<int> n, ch;
    for (int c : channels) {
      if (c < 0 || c >= static_cast<int>(nodes_.size())) throw GraphError("out(): channel out of range");
      n.push_back(nodes_[static_cast<size_t>(c)]);
      ch.push_back(chans_[static_cast<size_t>(c)]);
    }
    return Sig(g_, n, ch);
  }
  void to_graph_out() const { g_->to_graph_out(*this); }  // additive (graph_edit.rs:363-369)
  int node() const { return nodes_.at(0); }
  const std::vector<int>& nodes() const { return nodes_; }
  const std::vector<int>& channels() const { return chans_; }  // output channel of nodes()[i] that channel i of this handle carries

  // node.param("freq") -> Parameter (graph_edit.rs:761, 1700-1886)
  class Parameter {
   public:
    Parameter(Graph<F>* graph, int node, uint32_t index) : graph_(graph), node_(node), index_(index) {}
    void set(double v) { graph_->schedule({node_, index_, KNH_VALUE_FLOAT, v, 0, false, {}}); }
    void set(int64_t v) { graph_->schedule({node_, index_, KNH_VALUE_INTEGER, 0, v, false, {}}); }
    void set_at(double v, Time t) { graph_->schedule({node_, index_, KNH_VALUE_FLOAT, v, 0, true, t}); }
    void trig() { graph_->schedule({node_, index_, KNH_VALUE_TRIGGER, 0, 0, false, {}}); }
    // param.smooth(ParameterSmoothing::Linear(seconds)) / ::None (graph_edit.rs:1773-1786); needs .smooth_params()
    void smooth_linear(float seconds) { graph_->schedule({node_, index_, KNH_VALUE_SMOOTHING, seconds, 1, false, {}}); }
    void smooth_none() { graph_->schedule({node_, index_, KNH_VALUE_SMOOTHING, 0, 0, false, {}}); }
    void trig_at(Time t) { graph_->schedule({node_, index_, KNH_VALUE_TRIGGER, 0, 0, true, t}); }
    void set(PTrigger) { trig(); }
   private:
    Graph<F>* graph_;
    int node_;
    uint32_t index_;
  };
  Parameter param(const std::string& name) const { return g_->param(node(), name); }
  Parameter param(uint32_t index) const { return g_->param(node(), index); }
  // carrier.link("freq", modulator): audio-rate parameter edge; needs .ar_params() on the carrier
  Sig link(const std::string& name, const Sig& source) const { g_->link(node(), name, source); return *this; }

 private:
  GraphEdit<F>* g_;
  std::vector<int> nodes_, chans_;
};

inline const char* const* stage_param_names(uint16_t kind, int* n) {
  static const char* sin[] = {"freq", "phase_offset", "reset_phase"};
  static const char* svf[] = {"cutoff_freq", "q", "gain", "filter", "t_calculate_coefficients"};
  static const char* op[] = {"cutoff_freq"};
  static const char* asr[] = {"attack_time", "release_time", "t_release", "t_restart"};
  static const char* ar[] = {"attack_time", "release_time", "t_restart"};
  static const char* val[] = {"value"};
  static const char* seg[] = {"time_scale", "jump_to_segment", "t_restart", "t_stop"};
  static const char* dly[] = {"delay_time"};
  static const char* frq[] = {"freq"};
  switch (kind) {
    case KNH_STAGE_PHASOR: case KNH_STAGE_RANDOM_LIN: *n = 1; return frq;
    case KNH_STAGE_PAN2: { static const char* pan[] = {"pan"}; *n = 1; return pan; }
    case KNH_STAGE_POLYBLEP: { static const char* pb[] = {"freq", "pulse_width", "waveform"}; *n = 3; return pb; }
    case KNH_STAGE_SAFETY_LIMITER: case KNH_STAGE_WHITE_NOISE: case KNH_STAGE_PINK_NOISE: case KNH_STAGE_BROWN_NOISE: *n = 0; return frq;
    case KNH_STAGE_SAMPLE_DELAY: case KNH_STAGE_ALLPASS_DELAY: *n = 1; return dly;
    case KNH_STAGE_ALLPASS_FB_DELAY: { static const char* fbd[] = {"delay_time", "feedback"}; *n = 2; return fbd; }
    case KNH_STAGE_MUL_ENVELOPE: *n = 4; return seg;
    case KNH_STAGE_SIN_WT: case KNH_STAGE_SIN_NUMERIC: *n = 3; return sin;
    case KNH_STAGE_SVF: *n = 5; return svf;
    case KNH_STAGE_ONEPOLE_LPF: case KNH_STAGE_ONEPOLE_HPF: *n = 1; return op;
    case KNH_STAGE_MUL_ENV_ASR: *n = 4; return asr;
    case KNH_STAGE_MUL_ENV_AR: *n = 3; return ar;
    default: *n = 1; return val;
  }
}

// ---- Graph -----------------------------------------------------------------------------------------
template <typename F>
class Graph {
 public:
  struct Bank {
    knh_bank* h = nullptr;
    ChainPlan plan;
    uint32_t n_voices = 0;
    std::vector<F> out;  // [channels][block_size] of the last block(s)
  };
  Graph(uint32_t outputs, uint32_t sample_rate, size_t block_size) : outputs_(outputs), sample_rate_(sample_rate), block_size_(block_size) {
    if (outputs < 1 || outputs > 2) throw GraphError("this engine renders mono or stereo graphs");
  }
  ~Graph() {
    for (auto& b : banks_) knh_bank_destroy(b.h);
  }
  Graph(const Graph&) = delete;
  Graph& operator=(const Graph&) = delete;

  // graph.edit(|g| ...): changes are committed when the closure returns (graph.rs:1410, graph_edit.rs:258-262)
  template <typename Fn>
  void edit(Fn&& fn) {
    GraphEdit<F> g(this);
    fn(g);
    commit_changes();
  }
  size_t num_banks() const { return banks_.size(); }
  const Bank& bank(size_t i) const { return banks_.at(i); }
  size_t num_nodes() const { return nodes_.size(); }
  uint32_t sample_rate() const { return sample_rate_; }
  size_t block_size() const { return block_size_; }
  uint32_t outputs() const { return outputs_; }
  // Set `plan_only` before the first edit to build the plan without touching a device (tests on CPU).
  bool plan_only = false;

 private:
  friend class GraphEdit<F>;
  friend class AudioProcessor<F>;
  template <typename> friend class Sig;

  void schedule(const SchedulingEvent& ev) { events_.push_back(ev); }

  // ---- voice recognition -------------------------------------------------------------------------
  // Walks back from `node` and appends the stages that produce its signal; returns that signal's name for a later
  // stage's `input` (1 + the index of the stage whose output it is).  A node already met in this voice (one signal
  // feeding two consumers) is not walked again.  A stage that reads the stage right before it says so with input = 0,
  // so that plain chains keep the descriptors (and the pre-built kernels) they always had.
  uint16_t trace(int node, ChainPlan& p, std::vector<int>& visited, std::map<int, uint16_t>& done) {
    if (node < 0) throw GraphError("unconnected input in a voice chain");
    auto seen = done.find(node);
    if (seen != done.end()) return seen->second;
    NodeRec& n = nodes_[static_cast<size_t>(node)];
    if (n.bank >= 0) throw GraphError("a node feeds more than one voice (fan-out across voices is not a per-voice graph)");
    visited.push_back(node);
    auto in_of = [&](uint16_t src) -> uint16_t { return src == p.stages.size() ? 0 : src; };  // the stage before: the default
    auto finish = [&]() -> uint16_t { return done[node] = static_cast<uint16_t>(p.stages.size()); };
    // node.link(param, signal) on a node pushed as .ar_params() (graph_edit.rs:735-754, audio_rate.rs:11-85): the driving
    // signal is traced first (so that the node's own input can still be "the stage before"), the stage then names it
    // (knh_stage_desc.ar_param / .input2).  -> {ar_param, input2}
    auto ar_edge = [&](const NodeRec& x) -> std::pair<uint16_t, uint16_t> {
      if (x.link_source < 0) return {0, 0};
      if (!x.spec.ar_params_) throw GraphError("link(): the node was not pushed as .ar_params(); the edge would have no effect (ugen.rs:309-329)");
      const uint16_t drv = trace(x.link_source, p, visited, done);
      return {static_cast<uint16_t>(x.link_param + 1), drv};
    };
    if (n.type == NodeRec::MATH) {
      if (n.math_kind != 0xFFFF) {  // signal (op) Constant: graph_edit.rs:1036-1066
        const NodeRec& c = nodes_[static_cast<size_t>(n.in1)];
        const auto ar = ar_edge(c);
        const uint16_t src = trace(n.in0, p, visited, done);
        push_stage(p, n.math_kind, c.spec.smooth_params_ ? KNH_STAGE_FLAG_SMOOTH_PARAMS : 0, c.spec.precise_timing_, n.in1, c.spec.args, in_of(src), ar.second, ar.first);
        visited.push_back(n.in1);
        return finish();
      }
      const NodeRec& a = nodes_[static_cast<size_t>(n.in0)];
      const NodeRec& b = nodes_[static_cast<size_t>(n.in1)];
      // signal * envelope (either operand order; multiplication commutes exactly)
      int env = -1;
      if (n.math2_kind == KNH_STAGE_MATH_MUL)
        env = (b.type == NodeRec::UGEN && b.spec.is_env) ? n.in1 : (a.type == NodeRec::UGEN && a.spec.is_env) ? n.in0 : -1;
      if (env >= 0) {
        const NodeRec& e = nodes_[static_cast<size_t>(env)];
        if (!e.spec.wrappers.empty()) throw GraphError("wrappers on an envelope inside a product are not fused");
        const auto ar = ar_edge(e);
        const uint16_t src = trace(env == n.in1 ? n.in0 : n.in1, p, visited, done);
        push_stage(p, e.spec.kind, e.spec.smooth_params_ ? KNH_STAGE_FLAG_SMOOTH_PARAMS : 0, e.spec.precise_timing_, env, e.spec.args, in_of(src), ar.second, ar.first);
        visited.push_back(env);
        return finish();
      }
      // signal (op) pushed Constant: the same Constant + MathUGen pair as `signal (op) number`
      // (a Constant on the left only where the operation commutes exactly)
      const bool a_const = a.type == NodeRec::UGEN && a.spec.is_constant, b_const = b.type == NodeRec::UGEN && b.spec.is_constant;
      const bool commutes = n.math2_kind == KNH_STAGE_MATH_MUL || n.math2_kind == KNH_STAGE_MATH_ADD;
      if (b_const || (a_const && commutes)) {
        const int cn = b_const ? n.in1 : n.in0;
        const NodeRec& c = nodes_[static_cast<size_t>(cn)];
        if (!c.spec.wrappers.empty()) throw GraphError("wrappers on a Constant operand are not fused");
        const auto ar = ar_edge(c);
        const uint16_t src = trace(b_const ? n.in0 : n.in1, p, visited, done);
        const uint16_t kind = n.math2_kind == KNH_STAGE_MATH_MUL ? KNH_STAGE_MUL_CONST : n.math2_kind == KNH_STAGE_MATH_ADD ? KNH_STAGE_ADD_CONST
                            : n.math2_kind == KNH_STAGE_MATH_SUB ? KNH_STAGE_SUB_CONST : n.math2_kind == KNH_STAGE_MATH_DIV ? KNH_STAGE_DIV_CONST : KNH_STAGE_POW_CONST;
        push_stage(p, kind, c.spec.smooth_params_ ? KNH_STAGE_FLAG_SMOOTH_PARAMS : 0, c.spec.precise_timing_, cn, c.spec.args, in_of(src), ar.second, ar.first);
        visited.push_back(cn);
        return finish();
      }
      // two signals of the voice: MathUGen<_, U1, Op> with both operands named
      const uint16_t sa = trace(n.in0, p, visited, done);
      const uint16_t sb = trace(n.in1, p, visited, done);
      push_stage(p, n.math2_kind, 0, 0, -1, {}, sa, sb);
      return finish();
    }
    const UGenSpec& s = n.spec;
    if (s.is_env || s.is_constant) throw GraphError("an envelope/constant must be an operand of * + - /");
    const bool source = s.kind == KNH_STAGE_SIN_WT || s.kind == KNH_STAGE_SIN_NUMERIC || s.kind == KNH_STAGE_PHASOR || s.kind == KNH_STAGE_POLYBLEP ||
                        s.kind == KNH_STAGE_BUFFER_READER || s.kind == KNH_STAGE_WHITE_NOISE || s.kind == KNH_STAGE_PINK_NOISE ||
                        s.kind == KNH_STAGE_BROWN_NOISE || s.kind == KNH_STAGE_RANDOM_LIN;
    uint16_t flags = 0, input = 0;
    std::pair<uint16_t, uint16_t> ar{0, 0};
    if (source && n.link_source >= 0 && s.kind == KNH_STAGE_SIN_WT && s.ar_params_ && n.link_param == 0) {
      // SinWt(..).ar_params().link("freq", ..): the spelling with pre-built kernels (BASELINE config C5)
      input = in_of(trace(n.link_source, p, visited, done));
      flags = KNH_STAGE_FLAG_AR_FREQ;
    } else {
      ar = ar_edge(n);  // any other float parameter of any node the library can drive at audio rate (knh_stage_desc.ar_param)
      if (!source) input = in_of(trace(n.in0, p, visited, done));
    }
    if (s.smooth_params_) flags |= KNH_STAGE_FLAG_SMOOTH_PARAMS;
    push_stage(p, s.kind, flags, s.precise_timing_, node, s.args, input, ar.second, ar.first);
    for (auto& w : s.wrappers) push_stage(p, w.first, 0, 0, node, {w.second});
    return finish();
  }
  static void push_stage(ChainPlan& p, uint16_t kind, uint16_t flags, uint16_t dcpb, int node, std::vector<double> args, uint16_t input = 0,
                         uint16_t input2 = 0, uint16_t ar_param = 0) {
    p.stages.push_back(knh_stage_desc{kind, flags, dcpb, ar_param, input, input2});
    p.stage_node.push_back(node);
    p.stage_args.push_back(std::move(args));
  }

  // graph.rs:1707-1726: plan the newly connected voices, create their banks, upload their state.
  void commit_changes() {
    struct Voice { ChainPlan plan; std::vector<int> nodes; };
    std::vector<Voice> voices;
    for (auto& conn : new_outputs_) {
      // conn = the per-channel (node, output channel) of one to_graph_out(): either the same mono signal on every
      // graph output (`.out([0,0])`), or the two outputs of a Pan2 on the two graph outputs, in order
      if (conn.size() != outputs_) throw GraphError("to_graph_out(): channel count does not match the graph outputs");
      const NodeRec& last = nodes_[static_cast<size_t>(conn[0].first)];
      const bool pan = last.type == NodeRec::UGEN && last.spec.kind == KNH_STAGE_PAN2;
      for (size_t c = 0; c < conn.size(); ++c) {
        if (conn[c].first != conn[0].first) throw GraphError("different signals per output channel are not a voice chain");
        if (conn[c].second != (pan ? static_cast<int>(c) : 0)) throw GraphError("a Pan2's outputs go to graph outputs 0 and 1, in order");
      }
      if (pan && outputs_ != 2) throw GraphError("a Pan2 voice needs a stereo graph");
      Voice v;
      std::map<int, uint16_t> done;
      trace(conn[0].first, v.plan, v.nodes, done);
      voices.push_back(std::move(v));
    }
    new_outputs_.clear();
    // group by chain shape, first appearance first
    std::vector<std::string> order;
    std::map<std::string, std::vector<size_t>> groups;
    for (size_t i = 0; i < voices.size(); ++i) {
      std::string sig = voices[i].plan.signature();
      if (!groups.count(sig)) order.push_back(sig);
      groups[sig].push_back(i);
    }
    for (const std::string& sig : order) {
      const std::vector<size_t>& members = groups[sig];
      Bank b;
      b.plan = voices[members[0]].plan;
      b.n_voices = static_cast<uint32_t>(members.size());
      const int bank_index = static_cast<int>(banks_.size());
      for (size_t vi = 0; vi < members.size(); ++vi) {
        const Voice& v = voices[members[vi]];
        for (size_t s = 0; s < v.plan.stages.size(); ++s) {
          int node = v.plan.stage_node[s];
          if (node < 0) continue;  // a MathUGen of two signals: no parameters, found through v.nodes below
          NodeRec& nr = nodes_[static_cast<size_t>(node)];
          if (nr.bank < 0) { nr.bank = bank_index; nr.voice = static_cast<int>(vi); nr.stage = static_cast<int>(s); }
        }
        for (int node : v.nodes) {
          NodeRec& nr = nodes_[static_cast<size_t>(node)];
          if (nr.bank < 0) { nr.bank = bank_index; nr.voice = static_cast<int>(vi); nr.stage = -1; }
        }
      }
      if (!plan_only) {
        knh_bank_desc d{};
        d.abi_version = KNH_ABI_VERSION;
        d.n_voices = b.n_voices;
        d.sample_type = sizeof(F) == 8 ? KNH_F64 : KNH_F32;
        d.n_stages = static_cast<uint32_t>(b.plan.stages.size());
        d.stages = b.plan.stages.data();
        d.out_channels = outputs_;
        d.mix_mode = KNH_MIX_TREE;
        d.device = -1;
        d.allow_fma = 0;
        if (knh_bank_create(&d, &b.h) != KNH_OK) throw GraphError(std::string("knh_bank_create: ") + knh_last_error(nullptr));
        for (size_t s = 0; s < b.plan.stages.size(); ++s) {
          // voices of one bank may hold Envelopes of different lengths: pad to the longest (unused rows, duration 1)
          size_t n_args = 0;
          for (size_t vi = 0; vi < members.size(); ++vi) n_args = std::max(n_args, voices[members[vi]].plan.stage_args[s].size());
          if (!n_args) continue;
          std::vector<double> args(members.size() * n_args, 1.0);
          for (size_t vi = 0; vi < members.size(); ++vi)
            std::copy(voices[members[vi]].plan.stage_args[s].begin(), voices[members[vi]].plan.stage_args[s].end(), args.begin() + static_cast<long>(vi * n_args));
          check(b, knh_bank_set_ctor_args(b.h, static_cast<uint32_t>(s), 0, b.n_voices, args.data(), static_cast<uint32_t>(n_args)));
        }
        // UGen::init runs at push time in the reference (graph.rs:462-475); the bank's voices all init here
        check(b, knh_bank_init(b.h, sample_rate_, block_size_));
      }
      b.out.assign(static_cast<size_t>(outputs_) * block_size_, F(0));
      banks_.push_back(std::move(b));
    }
  }
  static void check(const Bank& b, int rc) {
    if (rc != KNH_OK) throw GraphError(std::string("knaster_hip: ") + knh_last_error(b.h));
  }

  uint32_t outputs_, sample_rate_;
  size_t block_size_;
  std::vector<NodeRec> nodes_;
  std::vector<std::vector<std::pair<int, int>>> new_outputs_;  // per to_graph_out(): (node, output channel) per graph output
  std::vector<Bank> banks_;
  std::vector<SchedulingEvent> events_;  // the rtrb channel of graph.rs:225-230, drained by the processor
};

// ---- GraphEdit: the builder handed to graph.edit() -------------------------------------------------
template <typename F>
class GraphEdit {
 public:
  explicit GraphEdit(Graph<F>* g) : graph_(g) {}
  Sig<F> push(UGenSpec spec) {  // graph_edit.rs:88-98
    NodeRec n;
    n.type = NodeRec::UGEN;
    n.spec = std::move(spec);
    const uint16_t outs = n.spec.outputs;
    graph_->nodes_.push_back(std::move(n));
    const int id = static_cast<int>(graph_->nodes_.size()) - 1;
    std::vector<int> nodes(outs, id), chans(outs);
    for (uint16_t c = 0; c < outs; ++c) chans[c] = c;
    return Sig<F>(this, nodes, chans);
  }

 private:
  template <typename> friend class Sig;
  Sig<F> math_const(const Sig<F>& s, uint16_t kind, double c) {  // graph_edit.rs:1036-1066
    std::vector<int> outs;
    int cn = push(Constant(c)).node();
    for (int src : s.nodes()) {
      NodeRec m;
      m.type = NodeRec::MATH;
      m.math_kind = kind;
      m.in0 = src;
      m.in1 = cn;
      graph_->nodes_.push_back(m);
      outs.push_back(static_cast<int>(graph_->nodes_.size()) - 1);
    }
    return Sig<F>(this, outs);
  }
  Sig<F> math_sig(const Sig<F>& a, const Sig<F>& b, uint16_t kind) {  // graph_edit.rs:936-971
    NodeRec m;
    m.type = NodeRec::MATH;
    m.math_kind = 0xFFFF;
    m.math2_kind = kind;
    m.in0 = a.node();
    m.in1 = b.node();
    graph_->nodes_.push_back(m);
    return Sig<F>(this, {static_cast<int>(graph_->nodes_.size()) - 1});
  }
  Sig<F> connect(const Sig<F>& src, const Sig<F>& sink) {
    NodeRec& n = graph_->nodes_[static_cast<size_t>(sink.node())];
    if (n.type != NodeRec::UGEN) throw GraphError(">>: the sink must be a pushed UGen");
    n.in0 = src.node();
    return sink;
  }
  void to_graph_out(const Sig<F>& s) {
    std::vector<std::pair<int, int>> conn;
    for (size_t c = 0; c < s.nodes().size(); ++c) conn.emplace_back(s.nodes()[c], s.channels()[c]);
    graph_->new_outputs_.push_back(std::move(conn));
  }
  typename Sig<F>::Parameter param(int node, const std::string& name) {
    const NodeRec& n = graph_->nodes_[static_cast<size_t>(node)];
    if (n.type != NodeRec::UGEN) throw GraphError("param(): not a UGen node");
    int cnt = 0;
    const char* const* names = stage_param_names(n.spec.kind, &cnt);
    for (int i = 0; i < cnt; ++i)
      if (name == names[i]) return typename Sig<F>::Parameter(graph_, node, static_cast<uint32_t>(i));
    // "wr_mul" = index T::Parameters of the wrapped node (wrappers_core/math.rs:69-98)
    uint32_t idx = static_cast<uint32_t>(cnt);
    for (auto& w : n.spec.wrappers) {
      if (w.first == KNH_STAGE_WR_MUL) {
        if (name == "wr_mul") return typename Sig<F>::Parameter(graph_, node, idx);
        ++idx;
      }
    }
    throw GraphError("DescriptionNotFound(" + name + ")");  // ParameterError::DescriptionNotFound, ugen.rs:365
  }
  typename Sig<F>::Parameter param(int node, uint32_t index) { return typename Sig<F>::Parameter(graph_, node, index); }
  void link(int node, const std::string& name, const Sig<F>& source) {
    NodeRec& n = graph_->nodes_[static_cast<size_t>(node)];
    int cnt = 0;
    const char* const* names = stage_param_names(n.spec.kind, &cnt);
    for (int i = 0; i < cnt; ++i)
      if (name == names[i]) { n.link_source = source.node(); n.link_param = i; return; }
    throw GraphError("DescriptionNotFound(" + name + ")");
  }
  Graph<F>* graph_;
};

// ---- AudioProcessor: the non-realtime driver (processor.rs:47-197) ---------------------------------
struct AudioProcessorOptions {  // processor.rs:23-45
  size_t block_size = 64;
  uint32_t sample_rate = 48000;
};
template <typename F>
class OutputBlock {  // RawContiguousBlock view, knaster_graph/src/block.rs:19-78
 public:
  OutputBlock(const F* data, size_t channels, size_t block_size) : d_(data), ch_(channels), bs_(block_size) {}
  F read(size_t channel, size_t frame) const {
    if (channel >= ch_ || frame >= bs_) throw std::out_of_range("OutputBlock::read");
    return d_[channel * bs_ + frame];
  }
  const F* channel_as_slice(size_t channel) const { return d_ + channel * bs_; }
  size_t channels() const { return ch_; }
  size_t block_size() const { return bs_; }
 private:
  const F* d_;
  size_t ch_, bs_;
};

template <typename F>
class AudioProcessor {
 public:
  // AudioProcessor::<F>::new::<U0, Outputs>(options) -> (Graph, AudioProcessor)
  static std::pair<std::unique_ptr<Graph<F>>, std::unique_ptr<AudioProcessor<F>>> create(uint32_t outputs, AudioProcessorOptions o = {}) {
    if (o.block_size == 0) throw GraphError("The block size must not be 0");
    auto g = std::make_unique<Graph<F>>(outputs, o.sample_rate, o.block_size);
    auto p = std::unique_ptr<AudioProcessor<F>>(new AudioProcessor<F>(g.get()));
    return {std::move(g), std::move(p)};
  }
  ~AudioProcessor() { if (dev_out_) knh_device_free(dev_out_); }
  AudioProcessor(const AudioProcessor&) = delete;
  AudioProcessor& operator=(const AudioProcessor&) = delete;
  size_t block_size() const { return graph_->block_size(); }
  uint16_t inputs() const { return 0; }
  uint16_t outputs() const { return static_cast<uint16_t>(graph_->outputs()); }
  uint64_t frame_clock() const { return frame_clock_; }

  // One block: GraphGen::process_block steps (ii), (iv), (v) of SURVEY call stack B.
  void run_without_inputs() { run_blocks(1); }
  // k blocks rendered in one launch per bank; events are resolved per block exactly as k calls would.
  void run_blocks(uint32_t k) {
    Graph<F>& g = *graph_;
    const uint64_t bs = g.block_size(), sr = g.sample_rate();
    const uint32_t keep = static_cast<uint32_t>(sr / bs);  // graph_gen.rs:73-75
    // (ii) parameter changes: waiting ones first, then the new ones (graph_gen.rs:110-166)
    std::vector<SchedulingEvent> fresh;
    fresh.swap(g.events_);
    for (uint32_t b = 0; b < k; ++b) {
      const uint64_t clock = frame_clock_ + b * bs;
      size_t n_waiting = waiting_.size();
      for (size_t i = 0; i < n_waiting; ++i) {
        auto [ev, blocks_waiting] = waiting_.front();
        waiting_.pop_front();
        if (blocks_waiting > keep) continue;
        if (!apply_parameter_change(ev, b, clock)) waiting_.emplace_back(ev, blocks_waiting + 1);
      }
      if (b == 0)
        for (SchedulingEvent& ev : fresh)
          if (!apply_parameter_change(ev, b, clock)) waiting_.emplace_back(ev, 0);
    }
    // (iv) run the banks, (v) sum their blocks in bank order: the mix stays in one HBM buffer that every
    // bank after the first adds into (the Add chain of graph.rs:850-864), and is read back once.
    const size_t n_out = static_cast<size_t>(k) * g.outputs() * bs;
    out_.assign(n_out, F(0));
    if (!g.banks_.empty() && g.banks_[0].h) {
      if (n_out > dev_out_elems_) {
        if (dev_out_) knh_device_free(dev_out_);
        dev_out_ = knh_device_malloc(n_out * sizeof(F), -1);
        if (!dev_out_) throw GraphError("knaster_hip: device allocation failed");
        dev_out_elems_ = n_out;
      }
      bool first = true;
      for (auto& bank : g.banks_) {
        if (!bank.h) continue;
        int rc = first ? knh_bank_process_blocks_device(bank.h, k, frame_clock_, dev_out_, nullptr)
                       : knh_bank_process_blocks_device_add(bank.h, k, frame_clock_, dev_out_, nullptr);
        if (rc != KNH_OK) throw GraphError(std::string("knaster_hip: ") + knh_last_error(bank.h));
        // each bank enqueues on its own stream: order them (the next bank adds to what this one wrote)
        if (knh_bank_synchronize(bank.h) != KNH_OK) throw GraphError(std::string("knaster_hip: ") + knh_last_error(bank.h));
        first = false;
      }
      if (knh_device_read(out_.data(), dev_out_, n_out * sizeof(F), nullptr) != KNH_OK) throw GraphError("knaster_hip: read back failed");
    }
    frame_clock_ += static_cast<uint64_t>(k) * bs;
    last_blocks_ = k;
  }
  // The last processed block (block `b` of the last run_blocks call)
  OutputBlock<F> output_block(uint32_t b = 0) const {
    if (b >= last_blocks_) throw std::out_of_range("output_block");
    const size_t n = static_cast<size_t>(graph_->outputs()) * graph_->block_size();
    return OutputBlock<F>(out_.data() + b * n, graph_->outputs(), graph_->block_size());
  }

 private:
  explicit AudioProcessor(Graph<F>* g) : graph_(g) {}
  // graph_gen.rs:269-305: returns true when applied (or unroutable), false to keep waiting
  bool apply_parameter_change(SchedulingEvent& ev, uint32_t block_offset, uint64_t clock) {
    Graph<F>& g = *graph_;
    uint64_t delay = 0;
    if (ev.has_time) {
      delay = ev.time.to_samples_until_due(g.block_size(), g.sample_rate(), clock);
      if (delay >= g.block_size()) return false;
    }
    const NodeRec& n = g.nodes_.at(static_cast<size_t>(ev.node));
    if (n.bank < 0 || n.stage < 0) return true;  // not part of a committed voice chain: dropped
    typename Graph<F>::Bank& bank = g.banks_[static_cast<size_t>(n.bank)];
    if (!bank.h) return true;
    // a parameter of the wrapped node, or one of its WrMul wrappers' "wr_mul"
    int cnt = 0;
    stage_param_names(n.spec.kind, &cnt);
    uint32_t stage = static_cast<uint32_t>(n.stage), param = ev.parameter;
    if (n.type == NodeRec::UGEN && static_cast<int>(ev.parameter) >= cnt) {
      uint32_t k = ev.parameter - static_cast<uint32_t>(cnt), s = stage;
      for (auto& w : n.spec.wrappers) {
        ++s;
        if (w.first == KNH_STAGE_WR_MUL) {
          if (k == 0) { stage = s; param = 0; break; }
          --k;
        }
      }
    }
    const uint32_t voice = static_cast<uint32_t>(n.voice);
    const uint16_t d16 = static_cast<uint16_t>(delay);
    const uint32_t kind = ev.kind;
    knh_bank_param_apply_many_at(bank.h, block_offset, 1, &voice, &stage, &param, &kind, &ev.f, &ev.i, delay > 0 ? &d16 : nullptr);
    return true;
  }
  Graph<F>* graph_;
  uint64_t frame_clock_ = 0;
  uint32_t last_blocks_ = 0;
  std::vector<F> out_;
  void* dev_out_ = nullptr;
  size_t dev_out_elems_ = 0;
  std::deque<std::pair<SchedulingEvent, uint32_t>> waiting_;
};

}  // namespace knaster
