// oracle_kat.cpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
// The reference's own known-answer tests for the hot path, restated against
// the oracle.  Every check cites the reference test it pins the oracle to.
// Exit code 0 = all passed; prints one line per test.
#include <cstdio>
#include <cstdlib>
#include <limits>

#include "oracle_bank.hpp"

using namespace kno;

static int g_fail = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      std::printf("  FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);        \
      ++g_fail;                                                            \
    }                                                                      \
  } while (0)
#define RUN(name)                       \
  do {                                  \
    int before = g_fail;                \
    name();                             \
    std::printf("%s %s\n", g_fail == before ? "ok  " : "FAIL", #name); \
  } while (0)

template <typename F>
static F process1(UGen<F>& g, AudioCtx& ctx, UGenFlags& flags) {
  F out[4] = {0, 0, 0, 0};
  g.process(ctx, flags, nullptr, out);
  return out[0];
}
template <typename F>
static UGenPtr<F> num(F v) {
  return std::make_unique<TestNumUGen<F>>(v);
}

// knaster_core_dsp/src/wrappers_core.rs:124-164
static void wrapper_arithmetic() {
  AudioCtx ctx(48000, 4);
  UGenFlags flags;
  using W = WrMath<double>;
  { W g(num(2.5), WrOp::Add, 2.5); CHECK(process1(g, ctx, flags) == 5.0); }
  { W g(num(2.5), WrOp::Mul, 3.); CHECK(process1(g, ctx, flags) == 7.5); }
  { W g(num(2.5), WrOp::Div, 5.); CHECK(process1(g, ctx, flags) == 0.5); }
  { W g(num(2.5), WrOp::VDiv, 5.); CHECK(process1(g, ctx, flags) == 2.); }
  { W g(num(6.0), WrOp::Sub, 7.); CHECK(process1(g, ctx, flags) == -1.0); }
  { W g(num(6.0), WrOp::VSub, 7.); CHECK(process1(g, ctx, flags) == 1.0); }
  { W g(num(6.0), WrOp::Powf, 2.);
    double s = process1(g, ctx, flags);
    CHECK(std::fabs(s - 36.0) < double(std::numeric_limits<float>::epsilon()) * 10.); }
  { W g(num(6.0), 2);
    double s = process1(g, ctx, flags);
    CHECK(std::fabs(s - 36.0) <= double(std::numeric_limits<float>::epsilon()) * 2.); }
}

static void run_precise(UGen<float>& g, AudioCtx& ctx, float* o16) {
  UGenFlags flags;
  const uint16_t d[5] = {5, 6, 8, 9, 10};
  for (uint16_t x : d) {
    g.set_delay_within_block_for_param(ctx, 0, x);
    CHECK(g.param(ctx, size_t(0), ParameterValue::Flt(double(x))) == ParameterError::Ok);
  }
  std::vector<float> in(2 * 16, 0.f), out(2 * 16, 0.f);
  BlockView<float> ib(in.data(), 2, 16), ob(out.data(), 2, 16);
  g.process_block(ctx, flags, ib, ob);
  for (int i = 0; i < 16; ++i) o16[i] = out[i];
}
// knaster_core_dsp/src/wrappers_core.rs:167-200
static void sample_accurate_parameters_test() {
  AudioCtx ctx(48000, 16);
  WrPreciseTiming<float> g(10, std::make_unique<TestInPlusParamUGen<float>>());
  float o[16];
  run_precise(g, ctx, o);
  const float expect[16] = {0., 0., 0., 0., 0., 5., 6., 6., 8., 9., 10., 10., 10., 10., 10., 10.};
  for (int i = 0; i < 16; ++i) CHECK(o[i] == expect[i]);
}
// knaster_core_dsp/src/wrappers_core.rs:202-250 (closure wrapper omitted: out of scope)
static void sample_accurate_parameters_with_wrappers_test() {
  AudioCtx ctx(48000, 16);
  UGenPtr<float> g = std::make_unique<WrPreciseTiming<float>>(10, std::make_unique<TestInPlusParamUGen<float>>());
  g = std::make_unique<WrMath<float>>(std::move(g), WrOp::Add, 0.0f);
  g = std::make_unique<WrMath<float>>(std::move(g), WrOp::Sub, 0.0f);
  g = std::make_unique<WrMath<float>>(std::move(g), WrOp::Div, 1.0f);
  g = std::make_unique<WrMath<float>>(std::move(g), WrOp::Mul, 1.0f);
  g = std::make_unique<WrMath<float>>(std::move(g), WrOp::Powf, 1.0f);
  g = std::make_unique<WrMath<float>>(std::move(g), 1);
  float o[16];
  run_precise(*g, ctx, o);
  const float expect[16] = {0., 0., 0., 0., 0., 5., 6., 6., 8., 9., 10., 10., 10., 10., 10., 10.};
  for (int i = 0; i < 16; ++i) CHECK(std::fabs(o[i] - expect[i]) <= 0.0002f);
}

// knaster_core_dsp/src/ugens/math.rs:317-359
static void gen_arithmetics() {
  AudioCtx ctx(48000, 4);
  UGenFlags flags;
  std::vector<float> b0(2 * 4), b1(2 * 4);
  std::fill(b0.begin(), b0.begin() + 4, 3.0f);
  std::fill(b0.begin() + 4, b0.end(), 2.0f);
  struct { MathOp op; float expect; } cases[] = {{MathOp::Add, 5.0f}, {MathOp::Sub, 1.0f}, {MathOp::Div, 1.5f}, {MathOp::Mul, 6.0f}};
  for (auto& c : cases) {
    MathUGen<float> m(1, c.op);
    float in[2] = {3.0f, 2.0f}, out[1];
    m.process(ctx, flags, in, out);
    CHECK(out[0] == c.expect);
    BlockView<float> ib(b0.data(), 2, 4), ob(b1.data(), 2, 4);
    m.process_block(ctx, flags, ib, ob);
    for (int i = 0; i < 4; ++i) CHECK(b1[i] == c.expect);
  }
}
// knaster_core_dsp/src/ugens/math.rs:360-389
static void gen_arithmetics_multichannel() {
  AudioCtx ctx(48000, 4);
  UGenFlags flags;
  std::vector<double> b0(4 * 4), b1(2 * 4);
  const double fills[4] = {3.0, 7.0, 2.0, 4.0};
  for (int c = 0; c < 4; ++c) std::fill(b0.begin() + c * 4, b0.begin() + (c + 1) * 4, fills[c]);
  MathUGen<double> m(2, MathOp::Add);
  double in[4] = {3.0, 7.0, 2.0, 4.0}, out[2];
  m.process(ctx, flags, in, out);
  CHECK(out[0] == 5.0 && out[1] == 11.0);
  BlockView<double> ib(b0.data(), 4, 4), ob(b1.data(), 2, 4);
  m.process_block(ctx, flags, ib, ob);
  for (int i = 0; i < 4; ++i) CHECK(b1[i] == 5.0 && b1[4 + i] == 11.0);
}

// knaster_graph/src/tests/graph_tests.rs:13-47
static void graph_empty_graph_zero_output() {
  Graph<float> g(0, 4, 16, 48000);
  g.commit_changes();
  std::vector<float> out(4 * 16, 1.f);
  g.run({}, out.data());
  for (float s : out) CHECK(s == 0.0f);
}
// graph_tests.rs:49-80
static void graph_inputs_to_outputs() {
  Graph<float> g(3, 3, 16, 48000);
  g.connect_to_output(GRAPH_KEY, 1, 0, true);
  g.connect_to_output(GRAPH_KEY, 2, 1, true);
  g.commit_changes();
  std::vector<float> in(16 * 3, 1.0f), out(3 * 16, 9.f);
  g.run({in.data(), in.data() + 16, in.data() + 32}, out.data());
  CHECK(out[0] == 1.0f && out[16] == 1.0f && out[32] == 0.0f);
}
// graph_tests.rs:82-127
static void graph_inputs_to_nodes_to_outputs() {
  Graph<float> g(3, 3, 16, 48000);
  g.connect_to_output(GRAPH_KEY, 0, 1, true);
  g.connect_to_output(GRAPH_KEY, 0, 2, true);
  NodeKey g0 = g.push(std::make_unique<TestInPlusParamUGen<float>>());
  NodeKey g1 = g.push(std::make_unique<TestInPlusParamUGen<float>>());
  g.set(g0, 0, ParameterValue::Flt(0.75));
  g.set(g1, 0, ParameterValue::Flt(0.5));
  g.connect_to_output(g0, 0, 2, true);  // additive onto graph-in 0 already at out 2
  g.connect_to_node(GRAPH_KEY, 2, 0, g1, false);
  g.connect_to_output(g1, 0, 0, true);
  g.commit_changes();
  std::vector<float> in(16 * 3, 2.0f), out(3 * 16, 9.f);
  g.run({in.data(), in.data() + 16, in.data() + 32}, out.data());
  CHECK(out[0] == 2.5f);
  CHECK(out[16] == 2.0f);
  CHECK(out[32] == 2.75f);
}
// graph_tests.rs:129-184
static void multichannel_nodes() {
  Graph<double> g(3, 2, 16, 48000);
  NodeKey v0_0 = g.push(num(0.125)), v0_1 = g.push(num(1.)), v1_0 = g.push(num(0.5)), v1_1 = g.push(num(4.125));
  NodeKey m = g.push(std::make_unique<MathUGen<double>>(2, MathOp::Add));
  g.connect_to_node(v0_0, 0, 0, m, false);
  g.connect_to_node(v0_1, 0, 1, m, false);
  g.connect_to_node(v1_0, 0, 2, m, false);
  g.connect_to_node(v1_1, 0, 3, m, false);
  g.connect_to_output(m, 0, 0, true);
  g.connect_to_output(m, 1, 1, true);
  g.commit_changes();
  std::vector<double> in(16 * 3, 1.0), out(2 * 16);
  g.run({in.data(), in.data() + 16, in.data() + 32}, out.data());
  CHECK(out[0] == 0.625 && out[16] == 5.125);
  NodeKey m2 = g.push(std::make_unique<MathUGen<double>>(1, MathOp::Mul));
  NodeKey m3 = g.push(std::make_unique<MathUGen<double>>(1, MathOp::Mul));
  g.connect_to_node(m, 0, 0, m2, false);
  g.connect_to_node(v1_0, 0, 1, m2, false);
  g.connect_to_node(m, 1, 0, m3, false);
  g.connect_to_node(v0_0, 0, 1, m3, false);
  g.connect_to_output(m2, 0, 0, false);  // to_graph_out_replace
  g.connect_to_output(m3, 0, 1, false);
  g.commit_changes();
  g.run({in.data(), in.data() + 16, in.data() + 32}, out.data());
  CHECK(out[0] == 0.625 * 0.5 && out[16] == 5.125 * 0.125);
}
// graph_tests.rs:256-297
static void disconnect() {
  Graph<float> g(0, 1, 16, 48000);
  NodeKey n1 = g.push(std::make_unique<TestInPlusParamUGen<float>>());
  g.set(n1, 0, ParameterValue::Flt(0.5));
  NodeKey n2 = g.push(std::make_unique<TestInPlusParamUGen<float>>());
  g.set(n2, 0, ParameterValue::Flt(1.25));
  NodeKey n3 = g.push(std::make_unique<TestInPlusParamUGen<float>>());
  g.set(n3, 0, ParameterValue::Flt(0.125));
  g.connect_to_node(n1, 0, 0, n2, true);
  g.connect_to_node(n2, 0, 0, n3, true);
  g.connect_to_output(n3, 0, 0, true);
  g.commit_changes();
  std::vector<float> out(16);
  g.run({}, out.data());
  CHECK(out[0] == 0.5f + 1.25f + 0.125f);
  g.disconnect_output_from_source(n1, 0);
  g.commit_changes();
  g.run({}, out.data());
  CHECK(out[0] == 1.25f + 0.125f);
  g.disconnect_input_to_sink(0, n3);
  g.commit_changes();
  g.run({}, out.data());
  CHECK(out[0] == 0.125f);
}
// knaster_benchmarks/benches/wrappers_vs_nodes.rs:25-28,51-54,79-82,108-111
static void bench_asserts() {
  {  // TestNum(2.0).wr_mul(0.5) -> 1.0
    Graph<float> g(0, 1, 32, 48000);
    NodeKey n = g.push(std::make_unique<WrMath<float>>(num(2.0f), WrOp::Mul, 0.5f));
    g.connect_to_output(n, 0, 0, true);
    g.commit_changes();
    std::vector<float> out(32);
    g.run({}, out.data());
    CHECK(out[0] == 1.0f && out[31] == 1.0f);
  }
  {  // 100 of them summed through the additive output -> 100.0
    Graph<float> g(0, 1, 32, 48000);
    for (int i = 0; i < 100; ++i) {
      NodeKey n = g.push(std::make_unique<WrMath<float>>(num(2.0f), WrOp::Mul, 0.5f));
      g.connect_to_output(n, 0, 0, true);
    }
    g.commit_changes();
    std::vector<float> out(32);
    g.run({}, out.data());
    CHECK(out[0] == 100.0f);
    CHECK(g.num_tasks() == 100 + 99);  // 99 auto Add nodes: the linear chain
  }
  {  // same via MathUGen Mul node with a constant
    Graph<float> g(0, 1, 32, 48000);
    for (int i = 0; i < 100; ++i) {
      NodeKey n = g.push(num(2.0f));
      NodeKey m = g.math_with_constant(n, 0, MathOp::Mul, 0.5f);
      g.connect_to_output(m, 0, 0, true);
    }
    g.commit_changes();
    std::vector<float> out(32);
    g.run({}, out.data());
    CHECK(out[0] == 100.0f);
  }
}
// knaster_core/examples/implement_a_gen.rs:24-35 (its Osc is SinNumeric's arithmetic)
static void implement_a_gen_sine() {
  AudioCtx ctx(48000, 64);
  UGenFlags flags;
  SinNumeric<float> osc(0.f);
  osc.init(48000, 64);
  osc.param(ctx, std::string("freq"), ParameterValue::Flt(200.));
  CHECK(process1(osc, ctx, flags) == 0.0f);
  std::vector<float> out(64);
  BlockView<float> ib, ob(out.data(), 1, 64);
  ib.frames = 64;
  osc.process_block(ctx, flags, ib, ob);
  float expect = std::sin((200.0f / 48000.0f) * 6.28318530717958647692f * 64.f);
  CHECK(std::fabs(out[63] - expect) < std::numeric_limits<float>::epsilon());
}
// knaster_primitives/src/time.rs:474-503
static void seconds_sample_conversion() {
  CHECK(Seconds::from_samples(1, 44100).to_samples(88200) == 2);
  CHECK(Seconds::from_samples(1, 44100).to_samples(44100) == 1);
  CHECK(Seconds::from_samples(2, 44100).to_samples(44100) == 2);
  CHECK(Seconds::from_samples(3, 44100).to_samples(44100) == 3);
  CHECK(Seconds::from_samples(4, 44100).to_samples(44100) == 4);
  CHECK(Seconds::from_samples(22050, 44100) == Seconds::from_secs_f64(0.5));
  CHECK(Seconds::from_samples(44100, 44100).to_samples(88200) == 88200);
  CHECK(Seconds::from_samples(44100 * 3 + 1, 44100).to_samples(88200) == 3 * 88200 + 2);
  CHECK(Seconds::from_samples(96000 * 3 + 8, 96000).to_samples(88200) == 3 * 88200 + 7);
  CHECK(Seconds::zero().to_samples(48000) == 0);
  CHECK(Seconds::from_secs_f64(0.).to_samples(48000) == 0);
  Seconds s = Seconds{0, SUBSECOND_TESIMALS_PER_SECOND - 1}.add(Seconds{1, 1});
  CHECK((s == Seconds{2, 0}));
}
// knaster_graph/src/graph.rs:2483-2513: EnvAsr(0,0) restarted+released is done within 10 blocks of 16
// knaster_primitives/src/time.rs:460-472, 497-503
static void seconds_tesimals_duration_arithmetic() {
  Seconds original{8347, SUBSECOND_TESIMALS_PER_SECOND - 5};
  CHECK(original == Seconds::from_subsample_tesimals_u64(original.to_subsample_tesimals_u64()));
  // duration_to_subsample_time: the two conversions need not agree in the last tesimal for arbitrary inputs, the
  // reference asserts they do for these four (Duration::from_secs_f64 = whole seconds + nanoseconds, truncated)
  for (double s : {73.73, 10.832, 10000.25, 84923.399}) {
    const double whole = std::floor(s);
    const uint32_t nanos = static_cast<uint32_t>((s - whole) * 1e9);
    CHECK(Seconds::from_secs_f64(s) == Seconds::from_duration(static_cast<uint64_t>(whole), nanos));
  }
  Seconds a{0, SUBSECOND_TESIMALS_PER_SECOND - 1}, b{1, 1};
  CHECK(a.add(b) == (Seconds{2, 0}));
}

static void free_node_when_done() {
  Graph<float> g(0, 2, 16, 48000);
  NodeKey asr = g.push(std::make_unique<EnvAsr<float>>(0.0f, 0.0f));
  g.set(asr, 0, ParameterValue::Flt(0.0));
  g.set(asr, 1, ParameterValue::Flt(0.0));
  g.set(asr, 3, ParameterValue::Trig());
  g.set(asr, 2, ParameterValue::Trig());
  g.commit_changes();
  bool done = false;
  std::vector<float> out(32);
  for (int i = 0; i < 10; ++i) {
    g.run({}, out.data());
    done = done || g.last_flags.done_;
  }
  CHECK(done);
}
// README.md:34-51 / config C1: SinWt(440) * 0.2 -> both outputs; channels identical, first sample 0
static void readme_example_shape() {
  Graph<float> g(0, 2, 64, 48000);
  NodeKey s = g.push(std::make_unique<SinWt<float>>(440.f));
  NodeKey m = g.math_with_constant(s, 0, MathOp::Mul, 0.2f);
  g.connect_to_output(m, 0, 0, true);
  g.connect_to_output(m, 0, 1, true);
  g.commit_changes();
  std::vector<float> out(128);
  g.run({}, out.data());
  CHECK(out[0] == 0.0f);
  for (int i = 0; i < 64; ++i) CHECK(out[i] == out[64 + i]);
  CHECK(g.num_tasks() == 3);
  // sample i is table[(i*inc)>>16] * 0.2 with inc = (u32)(440 * 16384*65536/48000)
  uint32_t inc = sat_u32(double(440.f) * (16384.0 * 65536.0 * (1.0 / 48000.0)));
  for (uint32_t i = 0; i < 64; ++i) CHECK(out[i] == sine_wavetable_f32()[((i * inc) >> 16) & 16383] * 0.2f);
}
// Buffer reuse: a linear chain of k single-channel nodes needs a constant number of blocks
// (buffer_allocator.rs:106-135, graph.rs:1588-1704)
static void buffer_reuse_linear_chain() {
  size_t lens[2];
  int ks[2] = {4, 64};
  for (int t = 0; t < 2; ++t) {
    Graph<float> g(0, 1, 16, 48000);
    NodeKey prev = g.push(num(1.0f));
    for (int i = 0; i < ks[t]; ++i) {
      NodeKey n = g.push(std::make_unique<TestInPlusParamUGen<float>>());
      g.connect_to_node(prev, 0, 0, n, false);
      prev = n;
    }
    g.connect_to_output(prev, 0, 0, true);
    g.commit_changes();
    lens[t] = g.buffer_len();
    std::vector<float> out(16);
    g.run({}, out.data());
    CHECK(out[0] == 1.0f);
  }
  CHECK(lens[0] == lens[1]);
  CHECK(lens[0] <= 16 * 4);
}
// Time::to_samples_until_due + GraphGen re-queueing (scheduling.rs:95-121, graph_gen.rs:110-166):
// a change scheduled for absolute frame 16*3+5 lands in block 3 at in-block frame 5 under WrPreciseTiming.
static void scheduled_change_lands_sample_accurately() {
  Graph<float> g(0, 1, 16, 48000);
  NodeKey n = g.push(std::make_unique<WrPreciseTiming<float>>(4, std::make_unique<TestInPlusParamUGen<float>>()));
  g.connect_to_output(n, 0, 0, true);
  g.commit_changes();
  g.set_at(n, 0, ParameterValue::Flt(7.0), Time::at(Seconds::from_samples(16 * 3 + 5, 48000)));
  std::vector<float> out(16);
  for (int b = 0; b < 5; ++b) {
    g.run({}, out.data());
    for (int i = 0; i < 16; ++i) {
      float expect = (b > 3 || (b == 3 && i >= 5)) ? 7.0f : 0.0f;
      CHECK(out[i] == expect);
    }
  }
}

int main() {
  RUN(wrapper_arithmetic);
  RUN(sample_accurate_parameters_test);
  RUN(sample_accurate_parameters_with_wrappers_test);
  RUN(gen_arithmetics);
  RUN(gen_arithmetics_multichannel);
  RUN(graph_empty_graph_zero_output);
  RUN(graph_inputs_to_outputs);
  RUN(graph_inputs_to_nodes_to_outputs);
  RUN(multichannel_nodes);
  RUN(disconnect);
  RUN(bench_asserts);
  RUN(implement_a_gen_sine);
  RUN(seconds_sample_conversion);
  RUN(seconds_tesimals_duration_arithmetic);
  RUN(free_node_when_done);
  RUN(readme_example_shape);
  RUN(buffer_reuse_linear_chain);
  RUN(scheduled_change_lands_sample_accurately);
  std::printf("%s (%d failures)\n", g_fail ? "KAT FAILED" : "KAT PASSED", g_fail);
  return g_fail ? 1 : 0;
}
