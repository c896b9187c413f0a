// Tests and timing of the C ABI driven exactly as the Rust shim drives it (shim_twin.hpp = lib.rs in C++), one
// process_block call per block, the way the reference runs a node (knaster_graph/src/task.rs:25-31,
// processor.rs:142-179, graph_gen.rs:110-200).
//   shim_twin_test --cpu                 : what needs no device (the exception guard of the ABI)
//   shim_twin_test --gpu                 : the call sequences on a MI355X, checked against the oracle / against one launch
//   shim_twin_test --bench C3|C1 [blocks] [batched] [ktime]: the per-block boundary rate, one JSON line
#include <sys/resource.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../oracle/knaster_oracle.hpp"  // test-side checker only
#include "shim_twin.hpp"

using namespace shim_twin;

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

struct C3Voice { double freq, gain, cutoff, q, atk, rel; };
static std::vector<C3Voice> c3_voices(int n) {  // the reference's xorshift32, seven draws per voice
  kno::XOrShift32Rng rng(0x9E3779B9u);
  std::vector<C3Voice> v;
  for (int i = 0; i < n; ++i) {
    double u[7];
    for (double& x : u) x = static_cast<double>(rng.gen_f32());
    v.push_back({55.0 * std::exp2(6.0 * u[0]), 1.0 / n, 200.0 + 7800.0 * u[1], 0.5 + 3.5 * u[2], 0.002 + 0.02 * u[3], 0.05 + 0.25 * u[4]});
  }
  return v;
}
static std::vector<knh_stage_desc> c3_chain() {
  return {stage(KNH_STAGE_SIN_WT), stage(KNH_STAGE_WR_MUL), stage(KNH_STAGE_SVF), stage(KNH_STAGE_MUL_ENV_ASR)};
}
static std::vector<std::vector<double>> c3_ctor(int n) {
  auto v = c3_voices(n);
  std::vector<std::vector<double>> c(4);
  for (const auto& x : v) {
    c[0].push_back(x.freq);
    c[1].push_back(x.gain);
    c[2].insert(c[2].end(), {double(KNH_SVF_LOW), x.cutoff, x.q, 0.0});
    c[3].insert(c[3].end(), {x.atk, x.rel});
  }
  return c;
}

// ---------------------------------------------------------------------------------------------------
// No C++ exception crosses the C ABI: with the address space capped, a bank whose constructor tables cannot be allocated
// comes back as KNH_ERR_OUT_OF_MEMORY with a message -- not as a std::bad_alloc unwinding into the caller.
static void cpu_no_exception_crosses_the_abi() {
  rlimit old{};
  getrlimit(RLIMIT_AS, &old);
  rlimit cap = old;
  cap.rlim_cur = 6ull << 30;
  CHECK(setrlimit(RLIMIT_AS, &cap) == 0);
  std::vector<knh_stage_desc> st = c3_chain();
  knh_bank_desc desc{};
  desc.abi_version = KNH_ABI_VERSION;
  desc.n_voices = 0x7FFFFFFFu;  // x 4 constructor arguments x 8 bytes = 64 GiB for the SVF's table alone
  desc.sample_type = KNH_F32;
  desc.n_stages = static_cast<uint32_t>(st.size());
  desc.stages = st.data();
  desc.out_channels = 2;
  desc.mix_mode = KNH_MIX_TREE;
  desc.device = -1;
  knh_bank* b = reinterpret_cast<knh_bank*>(1);
  const int32_t rc = knh_bank_create(&desc, &b);
  CHECK(rc == KNH_ERR_OUT_OF_MEMORY);
  CHECK(b == nullptr);
  CHECK(std::strstr(knh_last_error(nullptr), "memory") != nullptr);
  CHECK(std::strcmp(knh_status_string(KNH_ERR_OUT_OF_MEMORY), "out of host memory") == 0);
  setrlimit(RLIMIT_AS, &old);
  // and the library is as usable as before
  desc.n_voices = 64;
  CHECK(knh_bank_create(&desc, &b) == KNH_OK && b != nullptr);
  knh_bank_destroy(b);
}

// ---------------------------------------------------------------------------------------------------
static void gpu_readme_example_block_by_block() {  // BASELINE.json configs[0]: README.md:34-51, block 64
  GpuVoiceBank<float> bank({stage(KNH_STAGE_SIN_WT), stage(KNH_STAGE_MUL_CONST)}, 1, {{440.0}, {0.2}});
  bank.init(48000, 64);
  CHECK(bank.init_error().empty());
  AudioCtx ctx;
  UGenFlags flags;
  const auto& table = kno::sine_wavetable_f32();
  const uint32_t inc = kno::sat_u32(double(440.f) * (16384.0 * 65536.0 * (1.0 / 48000.0)));
  uint32_t phase = 0;
  std::vector<float> out(2 * 64);
  for (int block = 0; block < 8; ++block) {  // AudioProcessor::run_without_inputs, processor.rs:142-179
    CHECK(bank.process_block(ctx, flags, nullptr, out.data()) == KNH_OK);
    ctx.frame_clock += 64;
    for (size_t i = 0; i < 64; ++i) {
      const float want = table[(phase >> 16) & 16383] * 0.2f;
      CHECK(out[i] == want && out[64 + i] == want);
      phase += inc;
    }
  }
  CHECK(!flags.done);  // a chain without an envelope never finishes
}

// The note cycle of the bench through single calls, block by block, against ONE launch of the same blocks with the events
// scheduled ahead: the two ways a host may drive a bank give the same samples, bit for bit.
static void gpu_c3_block_by_block_equals_one_launch() {
  const int N = 320, B = 128, BLOCKS = 8;
  GpuVoiceBank<float> a(c3_chain(), N, c3_ctor(N)), b(c3_chain(), N, c3_ctor(N));
  a.init(48000, B);
  b.init(48000, B);
  CHECK(a.init_error().empty() && b.init_error().empty());
  AudioCtx ctx;
  ctx.block_size = ctx.frames_to_process = B;
  UGenFlags flags;
  std::vector<float> blocks_a(size_t(BLOCKS) * 2 * B), blocks_b(blocks_a.size());
  for (int blk = 0; blk < BLOCKS; ++blk) {
    // graph_gen.rs:110-166: the block's events first (one SchedulingEvent per voice = one param_apply each) ...
    if (blk == 0) for (int v = 0; v < N; ++v) CHECK(a.param_apply(ctx, a.index(v, 3, "t_restart"), Value::Trigger) == KNH_OK);
    if (blk == 4) for (int v = 0; v < N; ++v) CHECK(a.param_apply(ctx, a.index(v, 3, "t_release"), Value::Trigger) == KNH_OK);
    if (blk == 2) CHECK(a.param_apply(ctx, a.index(7, 2, "cutoff_freq"), Value::Float, 1234.5) == KNH_OK);
    // ... then the task loop (:196-200)
    CHECK(a.process_block(ctx, flags, nullptr, &blocks_a[size_t(blk) * 2 * B]) == KNH_OK);
    ctx.frame_clock += B;
  }
  std::vector<uint32_t> voices(N), st(N, 3), pr(N), kinds(N, KNH_VALUE_TRIGGER);
  for (int v = 0; v < N; ++v) voices[v] = v;
  std::fill(pr.begin(), pr.end(), 3u);
  CHECK(knh_bank_param_apply_many_at(b.raw(), 0, N, voices.data(), st.data(), pr.data(), kinds.data(), nullptr, nullptr, nullptr) == KNH_OK);
  std::fill(pr.begin(), pr.end(), 2u);
  CHECK(knh_bank_param_apply_many_at(b.raw(), 4, N, voices.data(), st.data(), pr.data(), kinds.data(), nullptr, nullptr, nullptr) == KNH_OK);
  const uint32_t v7 = 7, s2 = 2, p0 = 0, kf = KNH_VALUE_FLOAT;
  const double f = 1234.5;
  CHECK(knh_bank_param_apply_many_at(b.raw(), 2, 1, &v7, &s2, &p0, &kf, &f, nullptr, nullptr) == KNH_OK);
  uint32_t fl = 0;
  CHECK(knh_bank_process_blocks(b.raw(), BLOCKS, 0, blocks_b.data(), &fl) == KNH_OK);
  CHECK(std::memcmp(blocks_a.data(), blocks_b.data(), blocks_a.size() * sizeof(float)) == 0);
  float peak = 0;
  for (float x : blocks_a) peak = std::max(peak, std::fabs(x));
  CHECK(peak > 1e-4f);
}

// The bank under a splitting wrapper, called the way WrPreciseTiming::process_block calls the UGen it wraps
// (knaster_core_dsp/src/wrappers_core/precise_timing.rs:65-114): for a change due at frame 40 of a 64-frame block,
//   output.partial_mut(0, 40) + org_block.make_partial(0, 40), then output.partial_mut(40, 24) + make_partial(40, 24)
// -- PartialBlockMut views whose slices START at the offset (knaster_primitives/src/block.rs:307-339) beside a ctx that
// carries the same offset (ugen.rs:87-93).  Same samples as the whole block, nothing written outside the views.
// `legacy` runs the shim's logic of rounds 1-3 (the start of channel 0's slice taken for the base of the block AND
// block_start_offset handed on) against the same views: it writes frames [2 off, 2 off + n) and runs past the end of channel
// 1 -- the test must be able to tell (the output buffer has a guard zone behind it for exactly that).
static int32_t legacy_process_block(knh_bank* h, const AudioCtx& ctx, float* channel0_slice_start) {
  uint32_t f = 0;
  return knh_bank_process_block(h, ctx.frames_to_process, ctx.block_start_offset, ctx.frame_clock, channel0_slice_start, &f);
}
static void gpu_partial_blocks() {
  const int N = 70, B = 64, GUARD = 64;
  GpuVoiceBank<float> a(c3_chain(), N, c3_ctor(N)), b(c3_chain(), N, c3_ctor(N)), old(c3_chain(), N, c3_ctor(N));
  a.init(48000, B);
  b.init(48000, B);
  old.init(48000, B);
  AudioCtx ctx;
  UGenFlags flags;
  for (int v = 0; v < N; ++v) {
    a.param_apply(ctx, a.index(v, 3, "t_restart"), Value::Trigger);
    b.param_apply(ctx, b.index(v, 3, "t_restart"), Value::Trigger);
    old.param_apply(ctx, old.index(v, 3, "t_restart"), Value::Trigger);
  }
  std::vector<float> whole(2 * B), parts(2 * B + GUARD), legacy(2 * B + GUARD);
  bool legacy_differs = false, legacy_overruns = false;
  AggregateBlockRead<float> no_input{nullptr, 0, size_t(B)};
  for (int blk = 0; blk < 3; ++blk) {
    ctx.block_start_offset = 0; ctx.frames_to_process = B;
    CHECK(a.process_block(ctx, flags, nullptr, whole.data()) == KNH_OK);
    std::fill(parts.begin(), parts.end(), -7.f);
    std::fill(legacy.begin(), legacy.end(), -7.f);
    ContiguousBlock<float> out{parts.data(), 2, size_t(B)}, out_old{legacy.data(), 2, size_t(B)};
    const size_t cuts[3][2] = {{0, 40}, {40, 24}, {0, 0}};
    for (int part = 0; part < 2; ++part) {
      const size_t off = cuts[part][0], len = cuts[part][1];
      auto view = out.partial_mut(off, len);                                  // precise_timing.rs:103
      AudioCtx pctx = make_partial(ctx, off, len);                            // precise_timing.rs:104-105
      PartialBlock<float, AggregateBlockRead<float>> in_view{&no_input, off, len};  // precise_timing.rs:102
      CHECK(b.process_block(pctx, flags, in_view, view) == KNH_OK);
      auto view_old = out_old.partial_mut(off, len);
      CHECK(legacy_process_block(old.raw(), pctx, view_old.channel_as_slice_mut(0).ptr) == KNH_OK);
    }
    CHECK(std::memcmp(whole.data(), parts.data(), whole.size() * sizeof(float)) == 0);
    if (std::memcmp(whole.data(), parts.data(), whole.size() * sizeof(float)) != 0) {
      int first = -1, count = 0;
      for (int i = 0; i < 2 * B; ++i) if (std::memcmp(&whole[i], &parts[i], 4) != 0) { if (first < 0) first = i; ++count; }
      std::printf("  block %d: %d of %d samples differ, first at %d: %.9g vs %.9g\n", blk, count, 2 * B, first, whole[first], parts[first]);
    }
    for (int g = 0; g < GUARD; ++g) CHECK(parts[2 * B + g] == -7.f);  // nothing behind channel 1
    legacy_differs = legacy_differs || std::memcmp(whole.data(), legacy.data(), whole.size() * sizeof(float)) != 0;
    for (int g = 0; g < GUARD; ++g) legacy_overruns = legacy_overruns || legacy[2 * B + g] != -7.f;
    ctx.frame_clock += B;
  }
  CHECK(legacy_differs);   // the old logic puts the second part at frames [80, 104) of a 64-frame channel ...
  CHECK(legacy_overruns);  // ... i.e. past the end of channel 1
  // a partial view of a partial view (a splitting wrapper inside another): offsets add up on both sides
  {
    ctx.block_start_offset = 0; ctx.frames_to_process = B;
    CHECK(a.process_block(ctx, flags, nullptr, whole.data()) == KNH_OK);
    std::fill(parts.begin(), parts.end(), -7.f);
    ContiguousBlock<float> out{parts.data(), 2, size_t(B)};
    auto first = out.partial_mut(0, 16);
    AudioCtx c0 = make_partial(ctx, 0, 16);
    PartialBlock<float, AggregateBlockRead<float>> in0{&no_input, 0, 16};
    CHECK(b.process_block(c0, flags, in0, first) == KNH_OK);
    auto rest = out.partial_mut(16, 48);
    AudioCtx c1 = make_partial(ctx, 16, 48);
    auto rest_a = rest.partial_mut(0, 20);
    auto rest_b = rest.partial_mut(20, 28);
    AudioCtx c1a = make_partial(c1, 0, 20), c1b = make_partial(c1, 20, 28);
    CHECK(c1b.block_start_offset == 36 && c1b.frame_clock == ctx.frame_clock + 36);
    PartialBlock<float, AggregateBlockRead<float>> in1{&no_input, 16, 20}, in2{&no_input, 36, 28};
    CHECK(b.process_block(c1a, flags, in1, rest_a) == KNH_OK);
    CHECK(b.process_block(c1b, flags, in2, rest_b) == KNH_OK);
    CHECK(std::memcmp(whole.data(), parts.data(), whole.size() * sizeof(float)) == 0);
    for (int g = 0; g < GUARD; ++g) CHECK(parts[2 * B + g] == -7.f);
  }
  // channels that do not follow each other in memory (the Block trait promises one slice per channel, nothing more)
  {
    ctx.frame_clock += B;
    CHECK(a.process_block(ctx, flags, nullptr, whole.data()) == KNH_OK);
    std::vector<float> left(B, -7.f), right(B, -7.f);
    struct TwoBuffers {
      float* ch[2]; size_t bs;
      Slice<float> channel_as_slice_mut(size_t c) { return {ch[c], bs}; }
    } apart{{right.data(), left.data()}, size_t(B)};  // channel 0 BEHIND channel 1
    CHECK(b.process_block(ctx, flags, no_input, apart) == KNH_OK);
    CHECK(std::memcmp(whole.data(), right.data(), B * sizeof(float)) == 0);
    CHECK(std::memcmp(whole.data() + B, left.data(), B * sizeof(float)) == 0);
  }
}

// UGen::Inputs = 1 and one audio-rate parameter slot: per voice  out = SinWt.ar_params(freq <- slot * depth + f0) * input0
// (the modulator's block handed over by set_ar_param_buffer, the audio input by process_block's `input`).
static void gpu_inputs_and_audio_rate_buffer() {
  const int N = 2, B = 64;
  // stages: 0 INPUT(ch 1 = slot 0)  1 * depth  2 + f0  3 SinWt (audio-rate freq)  4 INPUT(ch 0)  5 sine * input
  std::vector<knh_stage_desc> st = {stage(KNH_STAGE_INPUT), stage(KNH_STAGE_MUL_CONST), stage(KNH_STAGE_ADD_CONST), stage(KNH_STAGE_SIN_WT),
                                    stage(KNH_STAGE_INPUT), stage(KNH_STAGE_MATH_MUL)};
  st[3].flags = KNH_STAGE_FLAG_AR_FREQ;
  st[5].input = 4;   // the SinWt (stage 3)
  st[5].input2 = 5;  // the audio input (stage 4)
  const double depth[N] = {100.0, 250.0}, f0[N] = {440.0, 660.0};
  GpuVoiceBank<float, 1> bank(st, N, {{1.0, 1.0}, {depth[0], depth[1]}, {f0[0], f0[1]}, {0.0, 0.0}, {0.0, 0.0}, {}}, 0, 1);
  bank.init(48000, B);
  CHECK(bank.init_error().empty());
  if (!bank.init_error().empty()) { std::printf("  %s\n", bank.init_error().c_str()); return; }
  AudioCtx ctx;
  UGenFlags flags;
  std::vector<float> mod(B), in0(B), out(2 * B);
  bank.set_ar_param_buffer(ctx, 0, mod.data());  // task.rs:113-120: once, when the schedule is taken
  const auto& table = kno::sine_wavetable_f32();
  const double f2pi = 16384.0 * 65536.0 * (1.0 / 48000.0);
  uint32_t phase[N] = {0, 0};
  for (int blk = 0; blk < 4; ++blk) {
    for (int i = 0; i < B; ++i) {  // the modulator node and the input node rendered this block
      mod[i] = std::sin(0.05f * float(blk * B + i));
      in0[i] = 0.5f + 0.001f * float(i);
    }
    const float* inputs[1] = {in0.data()};
    CHECK(bank.process_block(ctx, flags, inputs, out.data()) == KNH_OK);
    ctx.frame_clock += B;
    for (int i = 0; i < B; ++i) {
      float v[N];
      for (int k = 0; k < N; ++k) {
        const float freq = mod[i] * float(depth[k]) + float(f0[k]);
        const uint32_t inc = kno::sat_u32(double(freq) * f2pi);  // WrArParams::process -> SinWt::freq, audio_rate.rs:42-57, osc.rs:127-130
        const float s = table[(phase[k] >> 16) & 16383];
        phase[k] += inc;
        v[k] = s * in0[i];
      }
      const float want = v[0] + v[1];
      CHECK(out[i] == want && out[B + i] == want);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// batched: a block's triggers go through ONE knh_bank_param_apply_many call (what a host that hands its events over in
// batches does) instead of one knh_bank_param_apply per voice (what GraphGen::apply_parameter_change does per SchedulingEvent)
// ktime: also measure the voice kernel's device time per call (a timed bank launches a kernel per call: no resident kernel)
static int bench(const char* name, int blocks, bool batched, bool ktime, bool many) {
  const bool c1 = !std::strcmp(name, "C1");
  const char* nv_env = std::getenv("TWIN_VOICES");  // (diagnostics: the C3 voice at another bank size)
  const int N = c1 ? 1 : (nv_env ? std::atoi(nv_env) : 16384), B = c1 ? 64 : 512, UGENS = c1 ? 3 : 4;
  std::vector<knh_stage_desc> chain = c1 ? std::vector<knh_stage_desc>{stage(KNH_STAGE_SIN_WT), stage(KNH_STAGE_MUL_CONST)} : c3_chain();
  std::vector<std::vector<double>> ctor = c1 ? std::vector<std::vector<double>>{{440.0}, {0.2}} : c3_ctor(N);
  GpuVoiceBank<float> bank(chain, N, ctor);
  bank.init(48000, B);
  if (!bank.init_error().empty()) { std::fprintf(stderr, "%s\n", bank.init_error().c_str()); return 2; }
  AudioCtx ctx;
  ctx.block_size = ctx.frames_to_process = B;
  UGenFlags flags;
  std::vector<float> out(2 * B);
  std::vector<size_t> i_restart, i_release;
  if (!c1) for (int v = 0; v < N; ++v) { i_restart.push_back(bank.index(v, 3, "t_restart")); i_release.push_back(bank.index(v, 3, "t_release")); }
  std::vector<double> us;
  us.reserve(blocks);
  double peak = 0;
  double trace_us[4] = {0, 0, 0, 0};
  long trace_n = 0;
  // the blocks that carry a bank's worth of triggers, apart: the parameter calls, the process call, the device's side of it
  double trig_apply_us = 0, trig_process_us = 0, trig_trace_us[4] = {0, 0, 0, 0};
  long trig_n = 0;
  auto run = [&](int n, bool timed) {
    for (int blk = 0; blk < n; ++blk) {
      const auto t0 = std::chrono::steady_clock::now();
      if (!c1 && batched) {
        if (many) {
          if (blk % 64 == 0) bank.param_apply_many(i_restart, Value::Trigger);   // GpuVoiceBank::param_apply_many, lib.rs
          if (blk % 64 == 32) bank.param_apply_many(i_release, Value::Trigger);
        } else {
          if (blk % 64 == 0) bank.param_apply_range(0, uint32_t(N), 3, 3, Value::Trigger);   // GpuVoiceBank::param_apply_range: t_restart of the EnvAsr stage
          if (blk % 64 == 32) bank.param_apply_range(0, uint32_t(N), 3, 2, Value::Trigger);  // t_release
        }
      } else if (!c1) {
        if (blk % 64 == 0) for (size_t i : i_restart) bank.param_apply(ctx, i, Value::Trigger);   // one SchedulingEvent per voice
        if (blk % 64 == 32) for (size_t i : i_release) bank.param_apply(ctx, i, Value::Trigger);
      }
      const auto tm = std::chrono::steady_clock::now();
      bank.process_block(ctx, flags, nullptr, out.data());
      ctx.frame_clock += B;
      const auto t1 = std::chrono::steady_clock::now();
      if (timed) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
      const bool trig = !c1 && blk % 32 == 0;
      if (timed && trig) {
        trig_apply_us += std::chrono::duration<double, std::micro>(tm - t0).count();
        trig_process_us += std::chrono::duration<double, std::micro>(t1 - tm).count();
        trig_n += 1;
      }
      if (timed) {
        uint64_t tk[5];
        if (knh_bank_resident_trace(bank.raw(), tk) == KNH_OK && tk[0] && tk[4] >= tk[0]) {
          for (int k = 0; k < 4; ++k) trace_us[k] += double(int64_t(tk[k + 1] - tk[0])) * 0.01;
          trace_n += 1;
          if (trig) for (int k = 0; k < 4; ++k) trig_trace_us[k] += double(int64_t(tk[k + 1] - tk[0])) * 0.01;
        }
      }
      for (float x : out) peak = std::max(peak, double(std::fabs(x)));
    }
  };
  run(128, false);  // warm-up: clocks, first-use allocations
  if (ktime) knh_bank_timing_reset(bank.raw(), 1);
  const auto t0 = std::chrono::steady_clock::now();
  run(blocks, true);
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  double kms = 0;
  uint64_t launches = 0;
  if (ktime) knh_bank_timing_read(bank.raw(), &kms, &launches);
  std::sort(us.begin(), us.end());
  const double rate = double(N) * B * UGENS * blocks / secs;
  std::printf("{\"config\": \"%s\", \"driver\": \"C++ twin of the Rust shim: one knh_bank_process_block_channels per block, %s\", \"kernel_timed\": %s, "
              "\"voices\": %d, \"block_size\": %d, \"blocks\": %d, \"ugen_samples_per_s\": %.6g, \"us_per_block_mean\": %.3f, \"us_per_block_p50\": %.3f, "
              "\"us_per_block_p99\": %.3f, \"us_per_block_min\": %.3f, \"voice_kernel_us_per_block\": %.3f, \"output_peak\": %.4g, "
              "\"resident_calls\": %ld, \"device_us_after_the_voice_kernel_saw_the_command\": {\"fold_server_saw_it\": %.2f, \"first_tile_complete\": %.2f, \"last_tile_complete\": %.2f, \"block_and_flags_written\": %.2f}, "
              "\"trigger_blocks\": {\"n\": %ld, \"parameter_calls_us\": %.2f, \"process_call_us\": %.2f, \"device_first_tile_complete\": %.2f, \"device_block_and_flags_written\": %.2f}}\n",
              name, batched ? (many ? "a block's triggers in one knh_bank_param_apply_many" : "a block's triggers in one knh_bank_param_apply_range") : "single-call param_apply per event", ktime ? "true" : "false", N, B, blocks, rate, secs * 1e6 / blocks, us[us.size() / 2], us[size_t(us.size() * 0.99)], us.front(),
              launches ? kms * 1e3 / double(launches) : 0.0, peak, trace_n, trace_n ? trace_us[0] / trace_n : 0.0, trace_n ? trace_us[1] / trace_n : 0.0,
              trace_n ? trace_us[2] / trace_n : 0.0, trace_n ? trace_us[3] / trace_n : 0.0,
              trig_n, trig_n ? trig_apply_us / trig_n : 0.0, trig_n ? trig_process_us / trig_n : 0.0, trig_n ? trig_trace_us[1] / trig_n : 0.0, trig_n ? trig_trace_us[3] / trig_n : 0.0);
  return 0;
}

int main(int argc, char** argv) {
  bool cpu = false, gpu = false;
  for (int i = 1; i < argc; ++i) {
    cpu = cpu || !std::strcmp(argv[i], "--cpu");
    gpu = gpu || !std::strcmp(argv[i], "--gpu");
    if (!std::strcmp(argv[i], "--bench")) {
      if (knh_device_count() < 1) { std::printf("no gfx950 device\n"); return 2; }
      bool batched = false, ktime = false, many = false;  // "batched": one range call per block's triggers; "many": one array call
      for (int k = i + 1; k < argc; ++k) {
        batched = batched || !std::strcmp(argv[k], "batched") || !std::strcmp(argv[k], "many");
        many = many || !std::strcmp(argv[k], "many");
        ktime = ktime || !std::strcmp(argv[k], "ktime");
      }
      return bench(i + 1 < argc ? argv[i + 1] : "C3", i + 2 < argc && std::atoi(argv[i + 2]) > 0 ? std::atoi(argv[i + 2]) : 1024, batched, ktime, many);
    }
  }
  if (!cpu && !gpu) cpu = true;
  if (cpu) RUN(cpu_no_exception_crosses_the_abi);
  if (gpu) {
    if (knh_device_count() < 1) { std::printf("no gfx950 device\n"); return 2; }
    RUN(gpu_readme_example_block_by_block);
    RUN(gpu_c3_block_by_block_equals_one_launch);
    RUN(gpu_partial_blocks);
    RUN(gpu_inputs_and_audio_rate_buffer);
  }
  std::printf("%s (%d failures)\n", g_fail ? "SHIM TWIN FAILED" : "SHIM TWIN PASSED", g_fail);
  return g_fail ? 1 : 0;
}
