// oracle_capi.cpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
// ctypes-friendly entry points over oracle_bank.hpp.  Used only by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg.
#include <chrono>
#include <cstdio>
#include <cstring>

#include "oracle_bank.hpp"

using namespace kno;

namespace {
struct BankBase {
  virtual ~BankBase() = default;
  virtual int set_ctor(uint32_t stage, uint32_t first, uint32_t count, const double* args, uint32_t n_args) = 0;
  virtual int init(uint32_t sr, size_t bs) = 0;
  virtual int set_buffer(const void* samples, size_t n_frames, double sample_rate) = 0;
  virtual int param_apply(uint32_t voice, uint32_t stage, uint32_t param, ParameterValue v) = 0;
  virtual int set_delay(uint32_t voice, uint32_t stage, uint32_t param, uint16_t delay) = 0;
  virtual int schedule(uint32_t voice, uint32_t stage, uint32_t param, ParameterValue v, int mode, uint32_t s, uint32_t t) = 0;
  virtual int process(void* out, void* voices, uint32_t* flags, uint32_t* done, const void* in) = 0;
  virtual void set_in_channels(uint32_t n) = 0;
  virtual size_t mix_tasks() const = 0;
  virtual size_t mix_buffer_len() const = 0;
  std::string err;
};
template <typename F>
struct BankImpl : BankBase {
  OracleBank<F> b;
  BankImpl(const knh_stage_desc* st, uint32_t ns, uint32_t nv, uint32_t oc, bool mix, bool voices)
      : b(st, ns, nv, oc, mix, voices) {}
  int set_ctor(uint32_t stage, uint32_t first, uint32_t count, const double* args, uint32_t n_args) override {
    if (stage >= b.stages.size() || first + count > b.n_voices) return KNH_ERR_OUT_OF_RANGE;
    const int want = stage_n_ctor_args(b.stages[stage].kind);
    if (want >= 0 && static_cast<int>(n_args) != want) return KNH_ERR_INVALID_ARGUMENT;
    if (want < 0 && (n_args < 6 || (n_args - 4) % 2 != 0)) return KNH_ERR_INVALID_ARGUMENT;
    for (uint32_t i = 0; i < count; ++i) {
      b.ctor[first + i][stage].assign(n_args, 0.0);
      for (uint32_t a = 0; a < n_args; ++a) b.ctor[first + i][stage][a] = args[static_cast<size_t>(i) * n_args + a];
    }
    return KNH_OK;
  }
  int set_buffer(const void* samples, size_t n_frames, double sample_rate) override {
    auto buf = std::make_shared<Buffer<F>>();
    buf->buffer.assign(static_cast<const F*>(samples), static_cast<const F*>(samples) + n_frames);
    buf->sample_rate = sample_rate;
    b.buffer = buf;
    return KNH_OK;
  }
  int init(uint32_t sr, size_t bs) override {
    try {
      b.init(sr, bs);
    } catch (const std::exception& e) {
      err = e.what();
      return KNH_ERR_UNSUPPORTED_CHAIN;
    }
    return KNH_OK;
  }
  int param_apply(uint32_t voice, uint32_t stage, uint32_t param, ParameterValue v) override {
    try {
      return b.param_apply(voice, stage, param, v);
    } catch (const std::exception& e) {
      err = e.what();
      return KNH_ERR_WRONG_VALUE_KIND;
    }
  }
  int set_delay(uint32_t voice, uint32_t stage, uint32_t param, uint16_t delay) override {
    return b.set_delay(voice, stage, param, delay);
  }
  int schedule(uint32_t voice, uint32_t stage, uint32_t param, ParameterValue v, int mode, uint32_t s, uint32_t t) override {
    Time tm = mode == 2 ? Time::at(Seconds{s, t}) : Time::after(Seconds{s, t});
    return b.schedule(voice, stage, param, v, mode != 0, tm);
  }
  void set_in_channels(uint32_t n) override { b.in_channels = n; }
  int process(void* out, void* voices, uint32_t* flags, uint32_t* done, const void* in) override {
    try {
      uint32_t f = b.process_block(static_cast<F*>(out), static_cast<const F*>(in));
      if (flags) *flags = f;
      if (voices && b.want_voices) std::memcpy(voices, b.voice_block.data(), b.voice_block.size() * sizeof(F));
      if (done && b.want_voices) std::memcpy(done, b.done_frames.data(), b.done_frames.size() * sizeof(uint32_t));
    } catch (const std::exception& e) {
      err = e.what();
      return KNH_ERR_WRONG_VALUE_KIND;
    }
    return KNH_OK;
  }
  size_t mix_tasks() const override { return b.mix_graph ? b.mix_graph->num_tasks() : 0; }
  size_t mix_buffer_len() const override { return b.mix_graph ? b.mix_graph->buffer_len() : 0; }
};
ParameterValue make_value(uint32_t kind, double f, int64_t i) {
  switch (kind) {
    case KNH_VALUE_FLOAT: return ParameterValue::Flt(f);
    case KNH_VALUE_TRIGGER: return ParameterValue::Trig();
    case KNH_VALUE_INTEGER: return ParameterValue::Int(static_cast<uint64_t>(i));
    case KNH_VALUE_SMOOTHING: {
      ParameterSmoothing sm;
      sm.linear = i != 0;
      sm.seconds = static_cast<float>(f);
      return ParameterValue::Smooth(sm, i == 2 ? Rate::AudioRate : Rate::BlockRate);
    }
    default: {
      ParameterValue p;
      p.kind = ParameterValue::Bool;
      p.b = i != 0;
      return p;
    }
  }
}
}  // namespace

extern "C" {

void* kno_bank_create(const knh_stage_desc* stages, uint32_t n_stages, uint32_t n_voices, uint32_t sample_type,
                      uint32_t out_channels, int want_mix, int want_voices) {
  if (sample_type == KNH_F64)
    return new BankImpl<double>(stages, n_stages, n_voices, out_channels, want_mix != 0, want_voices != 0);
  return new BankImpl<float>(stages, n_stages, n_voices, out_channels, want_mix != 0, want_voices != 0);
}
void kno_bank_destroy(void* h) { delete static_cast<BankBase*>(h); }
const char* kno_bank_last_error(void* h) { return static_cast<BankBase*>(h)->err.c_str(); }
int kno_bank_set_ctor_args(void* h, uint32_t stage, uint32_t first, uint32_t count, const double* args, uint32_t n_args) {
  return static_cast<BankBase*>(h)->set_ctor(stage, first, count, args, n_args);
}
int kno_bank_set_buffer(void* h, const void* samples, size_t n_frames, double sample_rate) {
  return static_cast<BankBase*>(h)->set_buffer(samples, n_frames, sample_rate);
}
int kno_bank_init(void* h, uint32_t sr, size_t bs) { return static_cast<BankBase*>(h)->init(sr, bs); }
int kno_bank_param_apply(void* h, uint32_t voice, uint32_t stage, uint32_t param, uint32_t kind, double f, int64_t i) {
  return static_cast<BankBase*>(h)->param_apply(voice, stage, param, make_value(kind, f, i));
}
int kno_bank_set_delay_within_block_for_param(void* h, uint32_t voice, uint32_t stage, uint32_t param, uint16_t delay) {
  return static_cast<BankBase*>(h)->set_delay(voice, stage, param, delay);
}
int kno_bank_param_apply_many(void* h, size_t count, const uint32_t* voices, const uint32_t* stages, const uint32_t* params,
                              const uint32_t* kinds, const double* fvalues, const int64_t* ivalues, const uint16_t* delays) {
  BankBase* b = static_cast<BankBase*>(h);
  int rc = KNH_OK;
  for (size_t k = 0; k < count; ++k) {
    if (delays && delays[k] > 0) {
      int r = b->set_delay(voices[k], stages[k], params[k], delays[k]);
      if (r != KNH_OK) { rc = r; continue; }
    }
    int r = b->param_apply(voices[k], stages[k], params[k],
                           make_value(kinds[k], fvalues ? fvalues[k] : 0.0, ivalues ? ivalues[k] : 0));
    if (r != KNH_OK) rc = r;
  }
  return rc;
}
// time_mode: 0 = no time (asap), 1 = Time::after, 2 = Time::at
int kno_bank_schedule(void* h, uint32_t voice, uint32_t stage, uint32_t param, uint32_t kind, double f, int64_t i,
                      int time_mode, uint32_t seconds, uint32_t tesimals) {
  return static_cast<BankBase*>(h)->schedule(voice, stage, param, make_value(kind, f, i), time_mode, seconds, tesimals);
}
int kno_bank_process_block(void* h, void* out, void* voices_out, uint32_t* flags, uint32_t* done_frames) {
  return static_cast<BankBase*>(h)->process(out, voices_out, flags, done_frames, nullptr);
}
// the bank node's input channels (UGen::Inputs): declared before init, one [in_channels][block_size] block per process call
void kno_bank_set_in_channels(void* h, uint32_t n) { static_cast<BankBase*>(h)->set_in_channels(n); }
int kno_bank_process_block_in(void* h, const void* in, void* out, void* voices_out, uint32_t* flags, uint32_t* done_frames) {
  return static_cast<BankBase*>(h)->process(out, voices_out, flags, done_frames, in);
}
size_t kno_bank_mix_tasks(void* h) { return static_cast<BankBase*>(h)->mix_tasks(); }
size_t kno_bank_mix_buffer_len(void* h) { return static_cast<BankBase*>(h)->mix_buffer_len(); }

// --- small utilities for tests -------------------------------------------------
void kno_sine_table(float* out) { std::memcpy(out, sine_wavetable_f32().data(), TABLE_SIZE * sizeof(float)); }
uint32_t kno_xorshift32_next(uint32_t* state) {
  XOrShift32Rng r(*state);
  uint32_t v = r.gen_u32();
  *state = r.fpd;
  return v;
}
float kno_xorshift32_f32(uint32_t* state) {
  XOrShift32Rng r(*state);
  float v = r.gen_f32();
  *state = r.fpd;
  return v;
}
void kno_svf_coeffs_f32(uint32_t ty, float cutoff, float q, float gain_db, float sr, float* out6) {
  auto c = svf_set_coeffs<float>(svf_type_from_pinteger(ty), cutoff, q, gain_db, sr);
  out6[0] = c.a1; out6[1] = c.a2; out6[2] = c.a3; out6[3] = c.m0; out6[4] = c.m1; out6[5] = c.m2;
}
void kno_svf_coeffs_f64(uint32_t ty, double cutoff, double q, double gain_db, double sr, double* out6) {
  auto c = svf_set_coeffs<double>(svf_type_from_pinteger(ty), cutoff, q, gain_db, sr);
  out6[0] = c.a1; out6[1] = c.a2; out6[2] = c.a3; out6[3] = c.m0; out6[4] = c.m1; out6[5] = c.m2;
}
uint64_t kno_seconds_roundtrip(uint64_t samples, uint64_t from_rate, uint64_t to_rate) {
  return Seconds::from_samples(samples, from_rate).to_samples(to_rate);
}
uint64_t kno_time_to_samples_until_due(int absolute, uint32_t* seconds, uint32_t* tesimals, uint64_t block_size,
                                       uint64_t sample_rate, uint64_t frame_clock) {
  Time t{Seconds{*seconds, *tesimals}, absolute != 0};
  uint64_t r = t.to_samples_until_due(block_size, sample_rate, frame_clock);
  *seconds = t.seconds.seconds;
  *tesimals = t.seconds.subsecond_tesimals;
  return r;
}

// --- CPU baseline --------------------------------------------------------------
// Runs `blocks` blocks of a bank split into `threads` contiguous voice shards,
// each shard a reference-shaped sequential graph on its own thread; partial
// mixes are summed per block.  Returns wall seconds of the block loop only
// (graph construction excluded, as knaster_benchmarks/benches/graph_dsp_performance.rs:27-35).
// Events: restart_stage >= 0 fires param `restart_param` (trigger) on every voice
// before block 0; release_stage/param likewise before block `release_block`.
double kno_baseline_run(const knh_stage_desc* stages, uint32_t n_stages, uint32_t n_voices, uint32_t sample_type,
                        uint32_t out_channels, const double* const* ctor_args /* [stage] -> [n_voices][n_args] */,
                        uint32_t sample_rate, size_t block_size, uint32_t warmup_blocks, uint32_t blocks,
                        uint32_t threads, int restart_stage, uint32_t restart_param, int release_stage,
                        uint32_t release_param, uint32_t release_block, void* last_out) {
  if (threads == 0) threads = 1;
  if (threads > n_voices) threads = n_voices;
  std::vector<std::unique_ptr<BankBase>> shards;
  std::vector<uint32_t> first(threads + 1);
  for (uint32_t t = 0; t <= threads; ++t) first[t] = static_cast<uint32_t>(static_cast<uint64_t>(n_voices) * t / threads);
  for (uint32_t t = 0; t < threads; ++t) {
    uint32_t nv = first[t + 1] - first[t];
    std::unique_ptr<BankBase> b;
    if (sample_type == KNH_F64) b.reset(new BankImpl<double>(stages, n_stages, nv, out_channels, true, false));
    else b.reset(new BankImpl<float>(stages, n_stages, nv, out_channels, true, false));
    for (uint32_t s = 0; s < n_stages; ++s) {
      int na_i = stage_n_ctor_args(stages[s].kind);
      uint32_t na = na_i > 0 ? static_cast<uint32_t>(na_i) : 0;
      if (na) b->set_ctor(s, 0, nv, ctor_args[s] + static_cast<size_t>(first[t]) * na, na);
    }
    if (b->init(sample_rate, block_size) != KNH_OK) return -1.0;
    shards.push_back(std::move(b));
  }
  const size_t esz = sample_type == KNH_F64 ? 8 : 4;
  const size_t out_len = out_channels * block_size;
  // Every shard runs all its blocks on its own thread (an independent sequential scheduler, as a
  // user could do with independent AudioProcessors); per-block partial mixes are summed afterwards.
  const uint32_t total_blocks = warmup_blocks + blocks;
  std::vector<std::vector<unsigned char>> outs(threads, std::vector<unsigned char>(out_len * esz * blocks));
  auto shard_run = [&](uint32_t t, uint32_t from, uint32_t to) {
    const uint32_t nv = first[t + 1] - first[t];
    std::vector<unsigned char> scratch(out_len * esz);
    for (uint32_t i = from; i < to; ++i) {
      if (restart_stage >= 0 && i == 0)
        for (uint32_t v = 0; v < nv; ++v) shards[t]->param_apply(v, static_cast<uint32_t>(restart_stage), restart_param, ParameterValue::Trig());
      if (release_stage >= 0 && i == warmup_blocks + release_block)
        for (uint32_t v = 0; v < nv; ++v) shards[t]->param_apply(v, static_cast<uint32_t>(release_stage), release_param, ParameterValue::Trig());
      void* dst = i < warmup_blocks ? scratch.data() : outs[t].data() + static_cast<size_t>(i - warmup_blocks) * out_len * esz;
      shards[t]->process(dst, nullptr, nullptr, nullptr, nullptr);
    }
  };
  auto run_range = [&](uint32_t from, uint32_t to) {
    if (threads == 1) {
      shard_run(0, from, to);
    } else {
      std::vector<std::thread> th;
      for (uint32_t t = 0; t < threads; ++t) th.emplace_back(shard_run, t, from, to);
      for (auto& x : th) x.join();
    }
  };
  run_range(0, warmup_blocks);
  auto t0 = std::chrono::steady_clock::now();
  run_range(warmup_blocks, total_blocks);
  std::vector<unsigned char> total(out_len * esz);
  for (uint32_t i = 0; i < blocks; ++i) {
    std::memcpy(total.data(), outs[0].data() + static_cast<size_t>(i) * out_len * esz, total.size());
    for (uint32_t t = 1; t < threads; ++t) {
      const unsigned char* src = outs[t].data() + static_cast<size_t>(i) * out_len * esz;
      if (esz == 4) {
        float* a = reinterpret_cast<float*>(total.data());
        const float* b = reinterpret_cast<const float*>(src);
        for (size_t k = 0; k < out_len; ++k) a[k] += b[k];
      } else {
        double* a = reinterpret_cast<double*>(total.data());
        const double* b = reinterpret_cast<const double*>(src);
        for (size_t k = 0; k < out_len; ++k) a[k] += b[k];
      }
    }
  }
  auto t1 = std::chrono::steady_clock::now();
  if (last_out) std::memcpy(last_out, total.data(), total.size());
  return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
