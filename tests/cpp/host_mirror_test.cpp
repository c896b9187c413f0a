// Tests of the C++ host mirror, written the way the reference's own graph tests read
// (knaster_graph/src/tests/graph_tests.rs, README.md:34-51, knaster/examples/many_sines.rs).
//   host_mirror_test --plan   : no device needed (chain recognition, grouping, parameter names, Time)
//   host_mirror_test --gpu    : end to end on a MI355X, checked against the CPU oracle
#include <cstdio>
#include <cstring>

#include "../../knaster_amd/host/knaster_host.hpp"
#include "../../oracle/knaster_oracle.hpp"  // test-side checker only

using namespace knaster;

static int g_fail = 0;
#define CHECK(cond)                                                   \
  do {                                                                \
    if (!(cond)) {                                                    \
      std::printf("  FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
      ++g_fail;                                                       \
    }                                                                 \
  } while (0)
#define RUN(name)                                                         \
  do {                                                                    \
    int before = g_fail;                                                  \
    try { name(); } catch (const std::exception& e) { std::printf("  EXCEPTION %s\n", e.what()); ++g_fail; } \
    std::printf("%s %s\n", g_fail == before ? "ok  " : "FAIL", #name);    \
  } while (0)

static std::vector<double> voice_freqs(int n) {
  kno::XOrShift32Rng rng(0x9E3779B9u);
  std::vector<double> f;
  for (int i = 0; i < n; ++i) f.push_back(55.0 * std::exp2(6.0 * static_cast<double>(rng.gen_f32())));
  return f;
}

// ---------------------------------------------------------------------------------------------------
static void plan_readme_example() {
  auto [graph, processor] = AudioProcessor<float>::create(2, {64, 48000});
  graph->plan_only = true;
  graph->edit([&](GraphEdit<float>& g) {
    auto s = g.push(SinWt(440.));
    (s * 0.2).out({0, 0}).to_graph_out();
  });
  CHECK(graph->num_banks() == 1);
  CHECK(graph->num_nodes() == 3);  // SinWt, Constant, MathUGen Mul (graph_edit.rs:1047-1049)
  const auto& b = graph->bank(0);
  CHECK(b.n_voices == 1 && b.plan.stages.size() == 2);
  CHECK(b.plan.stages[0].kind == KNH_STAGE_SIN_WT && b.plan.stages[1].kind == KNH_STAGE_MUL_CONST);
  CHECK(knh_chain_ugen_count(b.plan.stages.data(), 2) == 3);
  CHECK(processor->outputs() == 2 && processor->inputs() == 0 && processor->block_size() == 64);
}
static void plan_groups_voices_by_chain_shape() {
  auto [graph, processor] = AudioProcessor<float>::create(2, {128, 48000});
  (void)processor;
  graph->plan_only = true;
  graph->edit([&](GraphEdit<float>& g) {
    for (int i = 0; i < 5; ++i) {  // many_sines.rs:51-63 without the Pan2
      auto env = g.push(EnvAr(0.01, 0.1));
      auto sine = g.push(SinWt(300. + i).wr_mul(0.0125));
      (env * sine).out({0, 0}).to_graph_out();
    }
    for (int i = 0; i < 3; ++i) {
      auto s = g.push(SinWt(100. + i).wr_mul(0.1));
      auto f = g.push(SvfFilter(SvfFilterType::Low, 1000., 1., 0.));
      auto e = g.push(EnvAsr(0.01, 0.2));
      ((s >> f) * e).out({0, 0}).to_graph_out();
    }
    auto m = g.push(SinWt(3.));
    auto c = g.push(SinWt(440.).ar_params().precise_timing(4));
    c.link("freq", m * 50. + 440.);
    (c * 0.1).out({0, 0}).to_graph_out();
  });
  CHECK(graph->num_banks() == 3);
  CHECK(graph->bank(0).n_voices == 5 && graph->bank(1).n_voices == 3 && graph->bank(2).n_voices == 1);
  const auto& b0 = graph->bank(0).plan.stages;
  CHECK(b0.size() == 3 && b0[0].kind == KNH_STAGE_SIN_WT && b0[1].kind == KNH_STAGE_WR_MUL && b0[2].kind == KNH_STAGE_MUL_ENV_AR);
  const auto& b1 = graph->bank(1).plan.stages;
  CHECK(b1.size() == 4 && b1[2].kind == KNH_STAGE_SVF && b1[3].kind == KNH_STAGE_MUL_ENV_ASR);
  const auto& b2 = graph->bank(2).plan.stages;
  CHECK(b2.size() == 5 && b2[3].kind == KNH_STAGE_SIN_WT && (b2[3].flags & KNH_STAGE_FLAG_AR_FREQ) && b2[3].delayed_changes_per_block == 4);
  CHECK(b2[1].kind == KNH_STAGE_MUL_CONST && b2[2].kind == KNH_STAGE_ADD_CONST && b2[4].kind == KNH_STAGE_MUL_CONST);
}
// node.link(param, signal) to parameters other than SinWt's freq (graph_edit.rs:735-754; WrArParams, audio_rate.rs:11-85)
static void plan_link_to_any_parameter() {
  auto [graph, processor] = AudioProcessor<float>::create(2, {64, 48000});
  (void)processor;
  graph->plan_only = true;
  graph->edit([&](GraphEdit<float>& g) {
    auto lfo = g.push(SinWt(3.));
    auto s = g.push(SinWt(220.));
    auto f = g.push(SvfFilter(SvfFilterType::Low, 1000., 1., 0.).ar_params());
    f.link("cutoff_freq", lfo * 500. + 1000.);
    ((s >> f) * 0.1).out({0, 0}).to_graph_out();
  });
  CHECK(graph->num_banks() == 1);
  const auto& st = graph->bank(0).plan.stages;  // lfo, * 500, + 1000, s, svf(cutoff <- stage 2), * 0.1
  CHECK(st.size() == 6);
  CHECK(st[0].kind == KNH_STAGE_SIN_WT && st[1].kind == KNH_STAGE_MUL_CONST && st[2].kind == KNH_STAGE_ADD_CONST && st[3].kind == KNH_STAGE_SIN_WT);
  CHECK(st[4].kind == KNH_STAGE_SVF && st[4].ar_param == 1 && st[4].input2 == 3 && st[4].input == 0 && st[4].flags == 0);
  CHECK(st[5].kind == KNH_STAGE_MUL_CONST);
  bool threw = false;
  try {  // without .ar_params() the edge would reach no WrArParams (ugen.rs:309-329): refused rather than silently ignored
    graph->edit([&](GraphEdit<float>& g) {
      auto lfo = g.push(SinWt(3.));
      auto c = g.push(SinWt(220.));
      c.link("phase_offset", lfo * 100.);
      (c * 0.1).out({0, 0}).to_graph_out();
    });
  } catch (const GraphError&) { threw = true; }
  CHECK(threw);
}
// noise.rs:11-22: WhiteNoise / PinkNoise / BrownNoise::new() draw their seeds from one process-wide counter, in
// construction order; the mirror hands that seed to the bank as the stage's constructor argument.
static void plan_noise_sources_take_seeds_in_construction_order() {
  auto [graph, processor] = AudioProcessor<float>::create(2, {64, 48000});
  (void)processor;
  graph->plan_only = true;
  const uint64_t first = next_randomness_seed() + 1;
  graph->edit([&](GraphEdit<float>& g) {
    (g.push(WhiteNoise()) * 0.1).out({0, 0}).to_graph_out();
    (g.push(PinkNoise()) * 0.1).out({0, 0}).to_graph_out();
    auto b = g.push(BrownNoise().wr_mul(0.5));
    auto f = g.push(OnePoleLpf(800.));
    (b >> f).out({0, 0}).to_graph_out();
    (g.push(WhiteNoise()) * 0.2).out({0, 0}).to_graph_out();  // a second voice of the first bank
  });
  CHECK(graph->num_banks() == 3);
  CHECK(graph->bank(0).n_voices == 2 && graph->bank(0).plan.stages[0].kind == KNH_STAGE_WHITE_NOISE);
  CHECK(graph->bank(1).plan.stages[0].kind == KNH_STAGE_PINK_NOISE && graph->bank(2).plan.stages[0].kind == KNH_STAGE_BROWN_NOISE);
  CHECK(graph->bank(0).plan.stage_args[0].size() == 1 && graph->bank(0).plan.stage_args[0][0] == double(first));
  CHECK(graph->bank(1).plan.stage_args[0][0] == double(first + 1));
  CHECK(graph->bank(2).plan.stage_args[0][0] == double(first + 2));
  CHECK(next_randomness_seed() == first + 4);
}
// knaster/examples/many_sines.rs:51-63 as written: `((env * sine) >> pan).to_graph_out()` -- the Pan2's two outputs go to
// graph outputs 0 and 1, and the 600 voices become one bank whose chain ends in KNH_STAGE_PAN2.
static void plan_many_sines_with_pan2() {
  auto [graph, processor] = AudioProcessor<float>::create(2, {64, 48000});
  (void)processor;
  graph->plan_only = true;
  graph->edit([&](GraphEdit<float>& g) {
    for (int i = 0; i < 600; ++i) {
      auto env = g.push(EnvAr(0.01, 0.1));
      auto sine = g.push(SinWt(3000. + i).wr_mul(0.0125));
      auto pan = g.push(Pan2(-1.0 + i / 300.0));
      ((env * sine) >> pan).to_graph_out();
    }
  });
  CHECK(graph->num_banks() == 1 && graph->bank(0).n_voices == 600);
  const auto& st = graph->bank(0).plan.stages;
  CHECK(st.size() == 4 && st[0].kind == KNH_STAGE_SIN_WT && st[1].kind == KNH_STAGE_WR_MUL && st[2].kind == KNH_STAGE_MUL_ENV_AR && st[3].kind == KNH_STAGE_PAN2);
  CHECK(knh_chain_ugen_count(st.data(), 4) == 4);  // SinWt, EnvAr, MathUGen Mul, Pan2
  // a Pan2 needs a stereo graph, and its outputs go to graph outputs 0 and 1 in that order
  bool threw = false;
  try {
    auto [g1, p1] = AudioProcessor<float>::create(1, {64, 48000});
    (void)p1;
    g1->plan_only = true;
    g1->edit([&](GraphEdit<float>& g) { (g.push(SinWt(440.)) >> g.push(Pan2(0.))).to_graph_out(); });
  } catch (const GraphError&) { threw = true; }
  CHECK(threw);
  threw = false;
  try {
    auto [g2, p2] = AudioProcessor<float>::create(2, {64, 48000});
    (void)p2;
    g2->plan_only = true;
    g2->edit([&](GraphEdit<float>& g) { (g.push(SinWt(440.)) >> g.push(Pan2(0.))).out({1, 0}).to_graph_out(); });
  } catch (const GraphError&) { threw = true; }
  CHECK(threw);
}
// A voice that is a graph: ring modulation of two oscillators, and the reference's "FM cascade"
// (knaster_benchmarks/benches/graph_dsp_performance.rs:37-72) written with the same operators.
static void plan_voices_that_are_graphs() {
  auto [graph, processor] = AudioProcessor<float>::create(2, {64, 48000});
  (void)processor;
  graph->plan_only = true;
  graph->edit([&](GraphEdit<float>& g) {
    for (int i = 0; i < 3; ++i) {
      auto a = g.push(SinWt(100. + i)), b = g.push(SinWt(200. + i).wr_mul(0.5));
      ((a * b) * 0.1).out({0, 0}).to_graph_out();  // ring modulation
    }
  });
  CHECK(graph->num_banks() == 1 && graph->bank(0).n_voices == 3);
  const auto& st = graph->bank(0).plan.stages;
  CHECK(st.size() == 5 && st[0].kind == KNH_STAGE_SIN_WT && st[1].kind == KNH_STAGE_SIN_WT && st[2].kind == KNH_STAGE_WR_MUL);
  CHECK(st[3].kind == KNH_STAGE_MATH_MUL && st[3].input == 1 && st[3].input2 == 3 && st[4].kind == KNH_STAGE_MUL_CONST && st[4].input == 0);
}
static void plan_rejects_what_is_not_a_voice_chain() {
  auto [graph, processor] = AudioProcessor<float>::create(2, {64, 48000});
  (void)processor;
  graph->plan_only = true;
  bool threw = false;
  try {
    graph->edit([&](GraphEdit<float>& g) {
      auto a = g.push(SinWt(100.)), e = g.push(EnvAsr(0.1, 0.1));
      (a >> e).out({0, 0}).to_graph_out();  // an envelope is an operand of *, not a processor
    });
  } catch (const GraphError&) { threw = true; }
  CHECK(threw);
  auto [g2, p2] = AudioProcessor<float>::create(2, {64, 48000});
  (void)p2;
  g2->plan_only = true;
  threw = false;
  try {
    g2->edit([&](GraphEdit<float>& g) { g.push(SinWt(1.)).param("no_such_param"); });
  } catch (const GraphError& e) { threw = std::strstr(e.what(), "DescriptionNotFound") != nullptr; }
  CHECK(threw);
}
static void time_and_seconds() {
  // knaster_primitives/src/time.rs:474-503
  CHECK(Seconds::from_samples(1, 44100).to_samples(88200) == 2);
  CHECK(Seconds::from_samples(44100 * 3 + 1, 44100).to_samples(88200) == 3 * 88200 + 2);
  CHECK(Seconds::from_samples(96000 * 3 + 8, 96000).to_samples(88200) == 3 * 88200 + 7);
  CHECK((Seconds::from_samples(22050, 44100) == Seconds::from_secs_f64(0.5)));
  // scheduling.rs:95-121 against the oracle's restatement, absolute and relative
  for (uint64_t due : {0ull, 5ull, 63ull, 64ull, 200ull, 100000ull}) {
    Time t = Time::at(Seconds::from_samples(due, 48000));
    kno::Time ot = kno::Time::at(kno::Seconds::from_samples(due, 48000));
    for (uint64_t clock = 0; clock < 400; clock += 64) CHECK(t.to_samples_until_due(64, 48000, clock) == ot.to_samples_until_due(64, 48000, clock));
    Time r = Time::after(Seconds::from_samples(due, 48000));
    kno::Time orr = kno::Time::after(kno::Seconds::from_samples(due, 48000));
    for (int k = 0; k < 6; ++k) CHECK(r.to_samples_until_due(64, 48000, 0) == orr.to_samples_until_due(64, 48000, 0));
  }
}

// ---------------------------------------------------------------------------------------------------
static void gpu_readme_example() {  // BASELINE.json configs[0] through the graph API
  auto [graph, processor] = AudioProcessor<float>::create(2, {64, 48000});
  graph->edit([&](GraphEdit<float>& g) {
    auto s = g.push(SinWt(440.));
    (s * 0.2).out({0, 0}).to_graph_out();
  });
  const auto& table = kno::sine_wavetable_f32();
  uint32_t inc = kno::sat_u32(double(440.f) * (16384.0 * 65536.0 * (1.0 / 48000.0)));
  uint32_t phase = 0;
  for (int block = 0; block < 3; ++block) {
    processor->run_without_inputs();
    auto out = processor->output_block();
    for (size_t i = 0; i < 64; ++i) {
      float want = table[(phase >> 16) & 16383] * 0.2f;
      CHECK(out.read(0, i) == want && out.read(1, i) == want);
      phase += inc;
    }
  }
  CHECK(processor->frame_clock() == 192);
}

struct C3Voice { double freq, gain, cutoff, q, atk, rel; };
static std::vector<C3Voice> c3_voices(int n) {
  kno::XOrShift32Rng rng(0x9E3779B9u);
  std::vector<C3Voice> v;
  for (int i = 0; i < n; ++i) {
    double u[7];
    for (double& x : u) x = static_cast<double>(rng.gen_f32());
    v.push_back({55.0 * std::exp2(6.0 * u[0]), 1.0 / n, 200.0 + 7800.0 * u[1], 0.5 + 3.5 * u[2], 0.002 + 0.02 * u[3], 0.05 + 0.25 * u[4]});
  }
  return v;
}
static void gpu_voice_graph_matches_reference_shaped_graph() {
  const int N = 300, B = 128;
  auto voices = c3_voices(N);
  auto [graph, processor] = AudioProcessor<float>::create(2, {B, 48000});
  std::vector<Sig<float>::Parameter> restart, release, cutoff;
  graph->edit([&](GraphEdit<float>& g) {
    for (const auto& v : voices) {
      auto s = g.push(SinWt(v.freq).wr_mul(v.gain));
      auto f = g.push(SvfFilter(SvfFilterType::Low, v.cutoff, v.q, 0.).precise_timing(4));
      auto e = g.push(EnvAsr(v.atk, v.rel));
      ((s >> f) * e).out({0, 0}).to_graph_out();
      restart.push_back(e.param("t_restart"));
      release.push_back(e.param("t_release"));
      cutoff.push_back(f.param("cutoff_freq"));
    }
  });
  CHECK(graph->num_banks() == 1 && graph->bank(0).n_voices == N);
  // the same patch as the reference would build it: one node per UGen, Add chain on both outputs
  kno::Graph<float> ref(0, 2, B, 48000);
  std::vector<kno::NodeKey> r_env, r_svf;
  for (const auto& v : voices) {
    auto s = ref.push(std::make_unique<kno::WrMath<float>>(std::make_unique<kno::SinWt<float>>(float(v.freq)), kno::WrOp::Mul, float(v.gain)));
    auto f = ref.push(std::make_unique<kno::WrPreciseTiming<float>>(4, std::make_unique<kno::SvfFilter<float>>(kno::Low, float(v.cutoff), float(v.q), 0.f)));
    auto e = ref.push(std::make_unique<kno::EnvAsr<float>>(float(v.atk), float(v.rel)));
    ref.connect_to_node(s, 0, 0, f, false);
    auto m = ref.math_nodes(f, 0, kno::MathOp::Mul, e, 0);
    ref.connect_to_output(m, 0, 0, true);
    ref.connect_to_output(m, 0, 1, true);
    r_env.push_back(e);
    r_svf.push_back(f);
  }
  ref.commit_changes();
  std::vector<float> want(2 * B);
  double worst = 0, peak = 0;
  for (int block = 0; block < 8; ++block) {
    if (block == 0)
      for (int i = 0; i < N; ++i) { restart[i].trig(); ref.set(r_env[i], 3, kno::ParameterValue::Trig()); }
    if (block == 2)  // sample-accurate: absolute time 2*B + 37 frames (Time::at), lands at in-block frame 37
      for (int i = 0; i < N; i += 2) {
        cutoff[i].set_at(500.0 + i, Time::at(Seconds::from_samples(2 * B + 37, 48000)));
        ref.set_at(r_svf[i], 0, kno::ParameterValue::Flt(500.0 + i), kno::Time::at(kno::Seconds::from_samples(2 * B + 37, 48000)));
      }
    if (block == 3)  // scheduled well ahead: due in block 5 at frame 9
      for (int i = 1; i < N; i += 2) {
        cutoff[i].set_at(3000.0 - i, Time::at(Seconds::from_samples(5 * B + 9, 48000)));
        ref.set_at(r_svf[i], 0, kno::ParameterValue::Flt(3000.0 - i), kno::Time::at(kno::Seconds::from_samples(5 * B + 9, 48000)));
      }
    if (block == 6)
      for (int i = 0; i < N; ++i) { release[i].trig(); ref.set(r_env[i], 2, kno::ParameterValue::Trig()); }
    processor->run_without_inputs();
    ref.run({}, want.data());
    auto out = processor->output_block();
    for (int c = 0; c < 2; ++c)
      for (int i = 0; i < B; ++i) {
        worst = std::max(worst, std::fabs(double(out.read(c, i)) - double(want[c * B + i])));
        peak = std::max(peak, std::fabs(double(want[c * B + i])));
      }
  }
  std::printf("  max |gpu - reference-shaped graph| = %.3g (peak %.3g)\n", worst, peak);
  CHECK(worst <= 1e-5);   // tree fold vs the reference's left fold: the north-star tolerance
  CHECK(peak > 1e-3);
}
static void gpu_run_blocks_equals_block_by_block() {
  const int N = 130, B = 64;
  auto voices = c3_voices(N);
  std::vector<std::vector<float>> results;
  for (int mode = 0; mode < 2; ++mode) {
    auto [graph, processor] = AudioProcessor<float>::create(2, {B, 48000});
    std::vector<Sig<float>::Parameter> restart, freq;
    graph->edit([&](GraphEdit<float>& g) {
      for (const auto& v : voices) {
        auto s = g.push(SinWt(v.freq).wr_mul(v.gain).precise_timing(2));
        auto e = g.push(EnvAr(v.atk, v.rel * 0.1));
        (e * s).out({0, 0}).to_graph_out();
        restart.push_back(e.param("t_restart"));
        freq.push_back(s.param("freq"));
      }
    });
    for (int i = 0; i < N; ++i) {
      restart[i].trig();
      freq[i].set_at(voices[i].freq * 2, Time::at(Seconds::from_samples(3 * B + (i % B), 48000)));
      restart[i].trig_at(Time::at(Seconds::from_samples(5 * B, 48000)));
    }
    std::vector<float> all;
    if (mode == 0) {
      for (int b = 0; b < 8; ++b) {
        processor->run_without_inputs();
        auto o = processor->output_block();
        all.insert(all.end(), o.channel_as_slice(0), o.channel_as_slice(0) + 2 * B);
      }
    } else {
      processor->run_blocks(8);
      for (uint32_t b = 0; b < 8; ++b) {
        auto o = processor->output_block(b);
        all.insert(all.end(), o.channel_as_slice(0), o.channel_as_slice(0) + 2 * B);
      }
    }
    results.push_back(all);
  }
  CHECK(results[0].size() == results[1].size());
  CHECK(std::memcmp(results[0].data(), results[1].data(), results[0].size() * sizeof(float)) == 0);
  float peak = 0;
  for (float x : results[0]) peak = std::max(peak, std::fabs(x));
  CHECK(peak > 1e-3f);
}

// carrier.link("phase_offset", lfo * depth + offset) and an envelope whose release_time follows a signal, through the graph
// API, against the reference-shaped graph with the reference's wrapper and parameter edges: bit for bit.
static void gpu_link_to_any_parameter() {
  const int N = 40, B = 64;
  auto voices = c3_voices(N);
  auto [graph, processor] = AudioProcessor<float>::create(2, {B, 48000});
  kno::Graph<float> ref(0, 2, B, 48000);
  std::vector<Sig<float>::Parameter> restart;
  std::vector<std::pair<kno::NodeKey, size_t>> ref_restart;
  graph->edit([&](GraphEdit<float>& g) {
    for (int i = 0; i < N; ++i) {
      const auto& v = voices[i];
      auto lfo = g.push(SinWt(2.0 + 0.1 * i));
      auto c = g.push(SinWt(v.freq).ar_params());
      c.link("phase_offset", lfo * 4000. + 8192.);
      auto e = g.push(EnvAr(0.001, 0.004).ar_params());
      e.link("release_time", lfo * 0.001 + 0.003);
      ((c * e) * v.gain).out({0, 0}).to_graph_out();
      restart.push_back(e.param("t_restart"));
      // the same voice with the oracle's graph API
      auto r_lfo = ref.push(std::make_unique<kno::SinWt<float>>(float(2.0 + 0.1 * i)));
      auto r_off = ref.math_with_constant(ref.math_with_constant(r_lfo, 0, kno::MathOp::Mul, 4000.f), 0, kno::MathOp::Add, 8192.f);
      auto r_c = ref.push(std::make_unique<kno::WrArParams<float>>(std::make_unique<kno::SinWt<float>>(float(v.freq))));
      ref.connect_to_parameter(r_off, 0, 1, r_c);
      auto r_rel = ref.math_with_constant(ref.math_with_constant(r_lfo, 0, kno::MathOp::Mul, 0.001f), 0, kno::MathOp::Add, 0.003f);
      auto r_e = ref.push(std::make_unique<kno::WrArParams<float>>(std::make_unique<kno::EnvAr<float>>(0.001f, 0.004f)));
      ref.connect_to_parameter(r_rel, 0, 1, r_e);
      auto r_m = ref.push(std::make_unique<kno::MathUGen<float>>(1, kno::MathOp::Mul));
      ref.connect_to_node(r_c, 0, 0, r_m, false);
      ref.connect_to_node(r_e, 0, 1, r_m, false);
      auto r_g = ref.math_with_constant(r_m, 0, kno::MathOp::Mul, float(v.gain));
      ref.connect_to_output(r_g, 0, 0, true);
      ref.connect_to_output(r_g, 0, 1, true);
      ref_restart.emplace_back(r_e, 2);
    }
  });
  CHECK(graph->num_banks() == 1);
  ref.commit_changes();
  std::vector<float> want(2 * B);
  float peak = 0;
  for (int b = 0; b < 6; ++b) {
    if (b == 0 || b == 3)
      for (int i = 0; i < N; ++i) { restart[i].trig(); ref.set(ref_restart[i].first, ref_restart[i].second, kno::ParameterValue::Trig()); }
    processor->run_without_inputs();
    ref.run({}, want.data());
    auto o = processor->output_block();
    // per voice the arithmetic is the reference's; the mixes differ by the order of their additions only (tree against
    // the reference's left fold): compare within the mix tolerance
    for (size_t i = 0; i < size_t(B); ++i) {
      CHECK(std::fabs(o.read(0, i) - want[i]) <= 1e-5f);
      peak = std::max(peak, std::fabs(o.read(0, i)));
    }
  }
  CHECK(peak > 1e-3f);
}

// Voices of three different shapes in one graph: three banks, one device-resident mix (bank 2 and 3 add
// into the buffer bank 1 wrote), compared with the reference-shaped graph holding all of them.
static void gpu_heterogeneous_voices_mix_on_device() {
  const int B = 96;
  auto voices = c3_voices(90);
  auto [graph, processor] = AudioProcessor<float>::create(2, {B, 48000});
  kno::Graph<float> ref(0, 2, B, 48000);
  std::vector<Sig<float>::Parameter> trig;
  std::vector<std::pair<kno::NodeKey, size_t>> ref_trig;
  graph->edit([&](GraphEdit<float>& g) {
    for (int i = 0; i < 90; ++i) {
      const auto& v = voices[i];
      if (i % 3 == 0) {  // plain sine * gain
        (g.push(SinWt(v.freq)) * v.gain).out({0, 0}).to_graph_out();
        auto s = ref.push(std::make_unique<kno::SinWt<float>>(float(v.freq)));
        auto m = ref.math_with_constant(s, 0, kno::MathOp::Mul, float(v.gain));
        ref.connect_to_output(m, 0, 0, true); ref.connect_to_output(m, 0, 1, true);
      } else if (i % 3 == 1) {  // many_sines shape
        auto e = g.push(EnvAr(v.atk, v.rel * 0.05));
        auto s = g.push(SinWt(v.freq).wr_mul(v.gain));
        (e * s).out({0, 0}).to_graph_out();
        trig.push_back(e.param("t_restart"));
        auto re = ref.push(std::make_unique<kno::EnvAr<float>>(float(v.atk), float(v.rel * 0.05)));
        auto rs = ref.push(std::make_unique<kno::WrMath<float>>(std::make_unique<kno::SinWt<float>>(float(v.freq)), kno::WrOp::Mul, float(v.gain)));
        auto m = ref.math_nodes(re, 0, kno::MathOp::Mul, rs, 0);
        ref.connect_to_output(m, 0, 0, true); ref.connect_to_output(m, 0, 1, true);
        ref_trig.emplace_back(re, 2);
      } else {  // filtered voice
        auto s = g.push(SinWt(v.freq).wr_mul(v.gain));
        auto f = g.push(OnePoleLpf(v.cutoff));
        auto e = g.push(EnvAsr(v.atk, v.rel));
        ((s >> f) * e).out({0, 0}).to_graph_out();
        trig.push_back(e.param("t_restart"));
        auto rs = ref.push(std::make_unique<kno::WrMath<float>>(std::make_unique<kno::SinWt<float>>(float(v.freq)), kno::WrOp::Mul, float(v.gain)));
        auto rf = ref.push(std::make_unique<kno::OnePoleLpf<float>>(float(v.cutoff)));
        auto re = ref.push(std::make_unique<kno::EnvAsr<float>>(float(v.atk), float(v.rel)));
        ref.connect_to_node(rs, 0, 0, rf, false);
        auto m = ref.math_nodes(rf, 0, kno::MathOp::Mul, re, 0);
        ref.connect_to_output(m, 0, 0, true); ref.connect_to_output(m, 0, 1, true);
        ref_trig.emplace_back(re, 3);
      }
    }
  });
  ref.commit_changes();
  CHECK(graph->num_banks() == 3);
  for (auto& t : trig) t.trig();
  for (auto& rt : ref_trig) ref.set(rt.first, rt.second, kno::ParameterValue::Trig());
  std::vector<float> want(2 * B);
  double worst = 0, peak = 0;
  for (int block = 0; block < 6; ++block) {
    processor->run_without_inputs();
    ref.run({}, want.data());
    auto out = processor->output_block();
    for (int c = 0; c < 2; ++c)
      for (int i = 0; i < B; ++i) {
        worst = std::max(worst, std::fabs(double(out.read(c, i)) - double(want[c * B + i])));
        peak = std::max(peak, std::fabs(double(want[c * B + i])));
      }
  }
  std::printf("  3 banks: max |gpu - reference-shaped graph| = %.3g (peak %.3g)\n", worst, peak);
  CHECK(worst <= 1e-5 && peak > 1e-3);
}

// Segment Envelopes of different lengths share one bank (constructor rows padded to the longest).
static void gpu_segment_envelopes_of_ragged_length() {
  const int B = 64, N = 40;
  auto voices = c3_voices(N);
  auto [graph, processor] = AudioProcessor<float>::create(2, {B, 48000});
  kno::Graph<float> ref(0, 2, B, 48000);
  std::vector<Sig<float>::Parameter> trig;
  std::vector<kno::NodeKey> ref_env;
  graph->edit([&](GraphEdit<float>& g) {
    for (int i = 0; i < N; ++i) {
      const auto& v = voices[i];
      std::vector<EnvelopeSegment> segs;
      std::vector<kno::EnvelopeSegment> rsegs;
      for (int k = 0; k < 1 + i % 4; ++k) {
        segs.push_back({0.0008 * (k + 1) + v.atk * 0.01, k % 2 ? 0.1 : 1.0});
        rsegs.emplace_back(segs.back().duration, segs.back().value);
      }
      auto s = g.push(SinWt(v.freq).wr_mul(v.gain));
      auto e = g.push(Envelope(0.0, segs, 1.0 + 0.1 * (i % 3), i % 5 == 0));
      (s * e).out({0, 0}).to_graph_out();
      trig.push_back(e.param("t_restart"));
      auto rs = ref.push(std::make_unique<kno::WrMath<float>>(std::make_unique<kno::SinWt<float>>(float(v.freq)), kno::WrOp::Mul, float(v.gain)));
      auto env = std::make_unique<kno::Envelope<float>>(0.0, rsegs);
      env->time_scale = 1.0 + 0.1 * (i % 3);
      env->looping = i % 5 == 0;
      auto re = ref.push(std::move(env));
      auto m = ref.math_nodes(rs, 0, kno::MathOp::Mul, re, 0);
      ref.connect_to_output(m, 0, 0, true); ref.connect_to_output(m, 0, 1, true);
      ref_env.push_back(re);
    }
  });
  ref.commit_changes();
  CHECK(graph->num_banks() == 1);
  for (auto& t : trig) t.trig();
  for (auto k : ref_env) ref.set(k, 2, kno::ParameterValue::Trig());
  std::vector<float> want(2 * B);
  double worst = 0, peak = 0;
  for (int block = 0; block < 8; ++block) {
    processor->run_without_inputs();
    ref.run({}, want.data());
    auto out = processor->output_block();
    for (int c = 0; c < 2; ++c)
      for (int i = 0; i < B; ++i) {
        worst = std::max(worst, std::fabs(double(out.read(c, i)) - double(want[c * B + i])));
        peak = std::max(peak, std::fabs(double(want[c * B + i])));
      }
  }
  std::printf("  ragged Envelopes: max |gpu - reference-shaped graph| = %.3g (peak %.3g)\n", worst, peak);
  CHECK(worst <= 1e-5 && peak > 1e-3);
}

// The UGens added on the same skeleton, through the graph API: a band-limited oscillator into a Schroeder allpass
// and a limiter, and a Phasor voice through a sample delay, against the same nodes in the reference-shaped graph.
static void gpu_polyblep_delay_limiter_voices() {
  const int B = 64, N = 50;
  auto voices = c3_voices(N);
  auto [graph, processor] = AudioProcessor<float>::create(2, {B, 48000});
  kno::Graph<float> ref(0, 2, B, 48000);
  std::vector<Sig<float>::Parameter> delay_params;
  std::vector<std::pair<kno::NodeKey, double>> ref_delays;
  graph->edit([&](GraphEdit<float>& g) {
    for (int i = 0; i < N; ++i) {
      const auto& v = voices[i];
      const double dly = 0.0005 + 0.00003 * i;
      if (i % 2 == 0) {
        const int wf = (i / 2) % 3 == 0 ? 0 : ((i / 2) % 3 == 1 ? 4 : 12);  // saw, square, fixed trapezoid: the exact ones
        auto o = g.push(PolyBlep(wf, v.freq).wr_mul(3.0 * v.gain * N));
        auto d = g.push(AllpassFeedbackDelay(0.004));
        auto l = g.push(SafetyLimiter());
        ((o >> d >> l) * (1.0 / N)).out({0, 0}).to_graph_out();
        delay_params.push_back(d.param("delay_time"));
        auto ro = ref.push(std::make_unique<kno::WrMath<float>>(
            std::make_unique<kno::PolyBlep<float>>(kno::waveform_from_pinteger(wf), float(v.freq)), kno::WrOp::Mul, float(3.0 * v.gain * N)));
        auto rd = ref.push(std::make_unique<kno::AllpassFeedbackDelay<float>>(kno::Seconds::from_secs_f64(0.004)));
        auto rl = ref.push(std::make_unique<kno::SafetyLimiter<float>>());
        ref.connect_to_node(ro, 0, 0, rd, false);
        ref.connect_to_node(rd, 0, 0, rl, false);
        auto m = ref.math_with_constant(rl, 0, kno::MathOp::Mul, float(1.0 / N));
        ref.connect_to_output(m, 0, 0, true); ref.connect_to_output(m, 0, 1, true);
        ref_delays.emplace_back(rd, dly);
      } else {
        auto o = g.push(Phasor(v.freq * 0.25));
        auto d = g.push(SampleDelay(0.004));
        ((o >> d) * v.gain).out({0, 0}).to_graph_out();
        delay_params.push_back(d.param("delay_time"));
        auto ro = ref.push(std::make_unique<kno::Phasor<float>>(v.freq * 0.25));
        auto rd = ref.push(std::make_unique<kno::SampleDelay<float>>(kno::Seconds::from_secs_f64(0.004)));
        ref.connect_to_node(ro, 0, 0, rd, false);
        auto m = ref.math_with_constant(rd, 0, kno::MathOp::Mul, float(v.gain));
        ref.connect_to_output(m, 0, 0, true); ref.connect_to_output(m, 0, 1, true);
        ref_delays.emplace_back(rd, dly);
      }
    }
  });
  ref.commit_changes();
  CHECK(graph->num_banks() == 2);
  for (size_t i = 0; i < delay_params.size(); ++i) {
    delay_params[i].set(ref_delays[i].second);
    ref.set(ref_delays[i].first, 0, kno::ParameterValue::Flt(ref_delays[i].second));
  }
  std::vector<float> want(2 * B);
  double worst = 0, peak = 0;
  for (int block = 0; block < 8; ++block) {
    processor->run_without_inputs();
    ref.run({}, want.data());
    auto out = processor->output_block();
    for (int c = 0; c < 2; ++c)
      for (int i = 0; i < B; ++i) {
        worst = std::max(worst, std::fabs(double(out.read(c, i)) - double(want[c * B + i])));
        peak = std::max(peak, std::fabs(double(want[c * B + i])));
      }
  }
  std::printf("  PolyBlep/allpass/limiter + Phasor/delay voices: max |gpu - reference-shaped graph| = %.3g (peak %.3g)\n", worst, peak);
  CHECK(worst <= 1e-5 && peak > 1e-3);
}

// many_sines.rs:51-92 end to end: 600 panned voices, envelope restarts and new frequencies while running, against the
// reference-shaped graph (one node per UGen, an Add chain per output channel) of the oracle.
static void gpu_many_sines_with_pan2() {
  const int N = 600, B = 64;
  kno::XOrShift32Rng rng(0x1234567u);
  struct V { double freq, amp, pan; };
  std::vector<V> voices;
  for (int i = 0; i < N; ++i) {
    const double a = rng.gen_f32(), b = rng.gen_f32(), c = rng.gen_f32();
    voices.push_back({3000.0 + 7000.0 * a, 0.01 + 0.005 * b, -1.0 + 2.0 * c});
  }
  auto [graph, processor] = AudioProcessor<float>::create(2, {B, 48000});
  std::vector<Sig<float>::Parameter> restart, freq, panp;
  graph->edit([&](GraphEdit<float>& g) {
    for (const auto& v : voices) {
      auto env = g.push(EnvAr(0.01, 0.1));
      auto sine = g.push(SinWt(v.freq).wr_mul(v.amp));
      auto pan = g.push(Pan2(v.pan));
      ((env * sine) >> pan).to_graph_out();
      restart.push_back(env.param("t_restart"));
      freq.push_back(sine.param("freq"));
      panp.push_back(pan.param("pan"));
    }
  });
  CHECK(graph->num_banks() == 1 && graph->bank(0).n_voices == N);
  kno::Graph<float> ref(0, 2, B, 48000);
  std::vector<kno::NodeKey> r_env, r_sine, r_pan;
  for (const auto& v : voices) {
    auto e = ref.push(std::make_unique<kno::EnvAr<float>>(0.01f, 0.1f));
    auto s = ref.push(std::make_unique<kno::WrMath<float>>(std::make_unique<kno::SinWt<float>>(float(v.freq)), kno::WrOp::Mul, float(v.amp)));
    auto p = ref.push(std::make_unique<kno::Pan2<float>>(float(v.pan)));
    auto m = ref.math_nodes(e, 0, kno::MathOp::Mul, s, 0);
    ref.connect_to_node(m, 0, 0, p, false);
    ref.connect_to_output(p, 0, 0, true);
    ref.connect_to_output(p, 1, 1, true);
    r_env.push_back(e); r_sine.push_back(s); r_pan.push_back(p);
  }
  ref.commit_changes();
  std::vector<float> want(2 * B);
  double worst = 0, peak = 0, stereo = 0;
  const double ratios[7] = {1.0, 9. / 8., 6. / 5., 3. / 2., 8. / 5., 16. / 9., 2.};
  for (int block = 0; block < 12; ++block) {
    for (int i = block % 3; i < N; i += 3) {  // many_sines.rs:73-88: a new frequency, then the envelope restarts
      const double f = 220.0 * ratios[i % 7] * (1 + block % 4);
      freq[i].set(f); ref.set(r_sine[i], 0, kno::ParameterValue::Flt(f));
      restart[i].trig(); ref.set(r_env[i], 2, kno::ParameterValue::Trig());
    }
    if (block == 5)
      for (int i = 0; i < N; i += 2) { panp[i].set(-voices[i].pan); ref.set(r_pan[i], 0, kno::ParameterValue::Flt(-voices[i].pan)); }
    processor->run_without_inputs();
    ref.run({}, want.data());
    auto out = processor->output_block();
    for (int c = 0; c < 2; ++c)
      for (int i = 0; i < B; ++i) {
        worst = std::max(worst, std::fabs(double(out.read(c, i)) - double(want[c * B + i])));
        peak = std::max(peak, std::fabs(double(want[c * B + i])));
      }
    for (int i = 0; i < B; ++i) stereo = std::max(stereo, std::fabs(double(out.read(0, i)) - double(out.read(1, i))));
  }
  std::printf("  many_sines with Pan2: max |gpu - reference-shaped graph| = %.3g (peak %.3g, max |L - R| %.3g)\n", worst, peak, stereo);
  CHECK(worst <= 1e-5 && peak > 1e-3 && stereo > 1e-4);
}
// Graph-shaped voices end to end: ring modulation with fan-out, and eight-oscillator FM cascades built with the
// operators of graph_dsp_performance.rs:37-72 -- each cascade's additive outputs are separate to_graph_out() calls, i.e.
// separate voices of the mirror that share nodes: that sharing across voices is what the planner refuses, so here every
// cascade sums its nodes itself (acc = acc + node, the Add chain the reference inserts).
// N voices, each the reference's "FM cascade" of D oscillators (graph_dsp_performance.rs:37-72); ADD = the constant the bench
// multiplies `last` by (440 there: the signal then grows 22-fold per oscillator and is not finite for long)
static void fm_cascade_voices(const int N, const int D, const float ADD) {
  const int B = 64;
  auto [graph, processor] = AudioProcessor<float>::create(2, {B, 48000});
  kno::Graph<float> ref(0, 2, B, 48000);
  graph->edit([&](GraphEdit<float>& g) {
    for (int v = 0; v < N; ++v) {
      const double det = 1.0 + 0.003 * v;
      auto s0 = g.push(SinWt(220. * det));
      auto acc = s0 * 0.05;
      auto last = s0;
      for (int i = 1; i < D; ++i) {
        auto s = g.push(SinWt((220. + i) * det));
        auto add = last * double(ADD);
        auto mul = s * last;
        auto node = mul + add;
        acc = acc + node;
        if (i + 1 < D) last = node * 0.05;
      }
      (acc * (1.0 / N)).out({0, 0}).to_graph_out();
    }
  });
  CHECK(graph->num_banks() == 1 && graph->bank(0).n_voices == N);
  for (int v = 0; v < N; ++v) {
    const float det = float(1.0 + 0.003 * v);
    auto s0 = ref.push(std::make_unique<kno::SinWt<float>>(float(220. * double(det))));
    auto c0 = ref.push(std::make_unique<kno::Constant<float>>(0.05f));
    auto acc = ref.math_nodes(s0, 0, kno::MathOp::Mul, c0, 0);
    auto last = s0;
    for (int i = 1; i < D; ++i) {
      auto s = ref.push(std::make_unique<kno::SinWt<float>>(float((220. + i) * double(det))));
      auto k440 = ref.push(std::make_unique<kno::Constant<float>>(ADD));
      auto add = ref.math_nodes(last, 0, kno::MathOp::Mul, k440, 0);
      auto mul = ref.math_nodes(s, 0, kno::MathOp::Mul, last, 0);
      auto node = ref.math_nodes(mul, 0, kno::MathOp::Add, add, 0);
      acc = ref.math_nodes(acc, 0, kno::MathOp::Add, node, 0);
      if (i + 1 < D) {
        auto c = ref.push(std::make_unique<kno::Constant<float>>(0.05f));
        last = ref.math_nodes(node, 0, kno::MathOp::Mul, c, 0);
      }
    }
    auto g1 = ref.push(std::make_unique<kno::Constant<float>>(float(1.0 / N)));
    auto out = ref.math_nodes(acc, 0, kno::MathOp::Mul, g1, 0);
    ref.connect_to_output(out, 0, 0, true);
    ref.connect_to_output(out, 0, 1, true);
  }
  ref.commit_changes();
  std::vector<float> want(2 * B);
  double worst = 0, peak = 0;
  for (int block = 0; block < 4; ++block) {
    processor->run_without_inputs();
    ref.run({}, want.data());
    auto out = processor->output_block();
    for (int c = 0; c < 2; ++c)
      for (int i = 0; i < B; ++i) {
        worst = std::max(worst, std::fabs(double(out.read(c, i)) - double(want[c * B + i])));
        peak = std::max(peak, std::fabs(double(want[c * B + i])));
      }
  }
  std::printf("  %d FM cascades of %d oscillators as graph voices: max |gpu - reference-shaped graph| = %.3g (peak %.3g)\n", N, D, worst, peak);
  CHECK(worst <= 1e-5 * std::max(1.0, peak) && peak > 1e-3);
}
static void gpu_voices_that_are_graphs() { fm_cascade_voices(70, 8, 440.0f); }
// the bench's own depth: 256 oscillators in one voice (1 531 stages: run frame-parallel, not fused -- kernels_interp.hip)
static void gpu_reference_fm_cascade_256() { fm_cascade_voices(2, 256, 19.0f); }

int main(int argc, char** argv) {
  bool plan = false, gpu = false;
  for (int i = 1; i < argc; ++i) {
    plan = plan || !std::strcmp(argv[i], "--plan");
    gpu = gpu || !std::strcmp(argv[i], "--gpu");
  }
  if (!plan && !gpu) plan = true;
  (void)voice_freqs;
  if (plan) {
    RUN(plan_readme_example);
    RUN(plan_groups_voices_by_chain_shape);
    RUN(plan_link_to_any_parameter);
    RUN(plan_noise_sources_take_seeds_in_construction_order);
    RUN(plan_many_sines_with_pan2);
    RUN(plan_voices_that_are_graphs);
    RUN(plan_rejects_what_is_not_a_voice_chain);
    RUN(time_and_seconds);
  }
  if (gpu) {
    if (knh_device_count() < 1) { std::printf("no gfx950 device\n"); return 2; }
    RUN(gpu_readme_example);
    RUN(gpu_voice_graph_matches_reference_shaped_graph);
    RUN(gpu_run_blocks_equals_block_by_block);
    RUN(gpu_link_to_any_parameter);
    RUN(gpu_heterogeneous_voices_mix_on_device);
    RUN(gpu_segment_envelopes_of_ragged_length);
    RUN(gpu_polyblep_delay_limiter_voices);
    RUN(gpu_many_sines_with_pan2);
    RUN(gpu_voices_that_are_graphs);
    RUN(gpu_reference_fm_cascade_256);
  }
  std::printf("%s (%d failures)\n", g_fail ? "HOST MIRROR FAILED" : "HOST MIRROR PASSED", g_fail);
  return g_fail ? 1 : 0;
}
