// host_shards.hpp -- a bank whose HOST work (parameter changes -> per-voice queues -> device event lists,
// precise_timing.rs:65-135 restated in bank.hip) runs on several threads.
//
// Why: with sample-accurate changes on every voice (BASELINE config C5) the host side of a launch costs more than
// its kernel -- ~35 ns per change on one core against 13 ns of kernel time.  The voices of a bank are independent
// (no state is shared between them on the host either), so the bank is cut into K contiguous voice ranges, each a
// complete bank of its own with its own queues, shadows, event lists and stream; a worker thread per range does
// that range's share of knh_bank_param_apply_many[_at] and of the block assembly in process, and launches the
// range's kernel on the range's stream.  The K partial mixes are then summed on the caller's stream in range order
// (sum_shards_kernel).  Included by bank.hip only; it is not a second engine, every range is a Bank<F>.
//
// The ranges may also live on DIFFERENT GPUs of the node (knh_bank_create_multi_device): range k is then a bank on
// devices[k] with its stream and its worker thread there, its mix is copied peer-to-peer (xGMI) into slot k of the
// staging buffer on devices[0], and the sum on devices[0] runs as before, in range order -- the one-shot direct sum of
// deterministic, one 4..8 KiB-per-block copy per GPU per launch, no ring.
//
// Only for KNH_MIX_TREE banks: the tree mix is already "deterministic, within 1e-5 of the left fold", and a sum of K
// range mixes is one more such grouping; KNH_MIX_LEFT_FOLD (the reference's exact order) keeps one range.
#pragma once
#include "shard_workers.hpp"

namespace {

template <typename F>
struct ShardedBank final : knh_bank {
  std::vector<std::unique_ptr<knh_bank>> shard;
  std::vector<uint32_t> base;    // first voice of each shard (+ n_voices at the end)
  uint32_t per_shard = 0;        // voices per shard (a multiple of 64; the last one may hold fewer)
  uint32_t nv = 0;
  std::unique_ptr<ShardWorkers> workers;
  std::vector<int> shard_device;  // device of each range; empty: all on the bank's device
  std::vector<F*> d_local;        // multi-device: a range's mix on its own device, before the peer copy
  std::vector<hipStream_t> streams;
  std::vector<hipEvent_t> shard_done;
  hipEvent_t sum_done = nullptr;
  bool sum_pending = false;
  hipStream_t own_stream = nullptr;
  F* d_parts = nullptr;          // [K][cap_blocks][channels][block_size]: each shard's mix
  F* d_out = nullptr;            // used when the caller gives no device buffer
  F* h_out = nullptr;            // pinned
  uint32_t cap_blocks = 0;
  std::vector<int> rcs;
  std::vector<uint32_t> shard_flags;

  ~ShardedBank() override {
    workers.reset();
    if (device >= 0 && initialised) (void)hipSetDevice(device);
    if (initialised) (void)hipDeviceSynchronize();
    if (initialised && multi_device())
      for (int k = 0; k < n(); ++k) { (void)hipSetDevice(dev_of(k)); (void)hipDeviceSynchronize(); }
    shard.clear();
    for (int k = 0; k < static_cast<int>(streams.size()); ++k) {
      if (multi_device()) (void)hipSetDevice(dev_of(k));
      if (streams[k]) (void)hipStreamDestroy(streams[k]);
      if (k < static_cast<int>(shard_done.size()) && shard_done[k]) (void)hipEventDestroy(shard_done[k]);
      if (k < static_cast<int>(d_local.size()) && d_local[k]) (void)hipFree(d_local[k]);
    }
    if (device >= 0 && initialised) (void)hipSetDevice(device);
    if (sum_done) (void)hipEventDestroy(sum_done);
    if (own_stream) (void)hipStreamDestroy(own_stream);
    if (d_parts) (void)hipFree(d_parts);
    if (d_out) (void)hipFree(d_out);
    if (h_out) (void)hipHostFree(h_out);
  }
  int n() const { return static_cast<int>(shard.size()); }
  uint32_t ranks() const override { return static_cast<uint32_t>(shard.size()); }
  bool multi_device() const { return !shard_device.empty(); }
  int dev_of(int k) const { return multi_device() ? shard_device[static_cast<size_t>(k)] : device; }
  int adopt(int k, int rc) {  // a shard's error becomes the bank's
    if (rc != KNH_OK) err = shard[k]->err;
    return rc;
  }
  int of_voice(uint32_t voice) const { return static_cast<int>(std::min<uint32_t>(voice / per_shard, static_cast<uint32_t>(n() - 1))); }

  int set_ctor(uint32_t stage, uint32_t first, uint32_t count, const double* args, uint32_t n_args) override {
    if (initialised) return fail(KNH_ERR_INVALID_ARGUMENT, "constructor arguments must be set before init");
    if (stage >= stages.size()) return fail(KNH_ERR_OUT_OF_RANGE, "stage out of range");
    if (static_cast<uint64_t>(first) + count > nv) return fail(KNH_ERR_OUT_OF_RANGE, "voice range out of range");
    if (count == 0) return adopt(0, shard[0]->set_ctor(stage, 0, 0, args, n_args));
    for (int k = 0; k < n(); ++k) {
      const uint32_t lo = std::max(first, base[k]), hi = std::min(first + count, base[k + 1]);
      if (lo >= hi) continue;
      int rc = shard[k]->set_ctor(stage, lo - base[k], hi - lo, args ? args + static_cast<size_t>(lo - first) * n_args : nullptr, n_args);
      if (rc != KNH_OK) return adopt(k, rc);
    }
    return KNH_OK;
  }
  int set_buffer(uint32_t stage, const void* samples, size_t n_frames, double sr) override {
    for (int k = 0; k < n(); ++k) {
      int rc = shard[k]->set_buffer(stage, samples, n_frames, sr);
      if (rc != KNH_OK) return adopt(k, rc);
    }
    return KNH_OK;
  }
  int set_input(uint32_t n_blocks, const void* host, const void* dev) override {  // every range reads the same input block(s)
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    if (dev && multi_device()) return fail(KNH_ERR_INVALID_ARGUMENT, "a bank on several GPUs takes its input from host memory");
    for (int k = 0; k < n(); ++k) {
      int rc = shard[k]->set_input(n_blocks, host, dev);
      if (rc != KNH_OK) return adopt(k, rc);
    }
    KNH_HIP(hipSetDevice(device));
    return KNH_OK;
  }
  int init(uint32_t sr, size_t bs) override {
    if (initialised) return fail(KNH_ERR_INVALID_ARGUMENT, "already initialised");
    for (int k = 0; k < n(); ++k) {  // one after the other: a run-time fused kernel is compiled once and found in the cache by the rest
      int rc = shard[k]->init(sr, bs);
      if (rc != KNH_OK) return adopt(k, rc);
    }
    device = shard[0]->device;
    sample_rate = sr;
    block_size = bs;
    streams.assign(n(), nullptr);
    shard_done.assign(n(), nullptr);
    d_local.assign(n(), nullptr);
    for (int k = 0; k < n(); ++k) {
      KNH_HIP(hipSetDevice(dev_of(k)));
      KNH_HIP(hipStreamCreateWithFlags(&streams[k], hipStreamNonBlocking));
      KNH_HIP(hipEventCreateWithFlags(&shard_done[k], hipEventDisableTiming));
    }
    KNH_HIP(hipSetDevice(device));
    KNH_HIP(hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking));
    KNH_HIP(hipEventCreateWithFlags(&sum_done, hipEventDisableTiming));
    workers.reset(new ShardWorkers(n()));
    scratch.assign(static_cast<size_t>(n()), Scratch{});
    rcs.assign(n(), KNH_OK);
    shard_flags.assign(n(), 0u);
    initialised = true;
    return KNH_OK;
  }
  // A batched call is checked at once (same codes and messages as the one-range bank) but applied later, by the
  // range workers: at the start of the next process, or ahead of any single call (order per voice is kept).
  struct Batch {
    uint32_t block_offset = 0;
    size_t count = 0;
    bool has_f = false, has_i = false, has_d = false;
    std::vector<uint32_t> voices, stgs, params, kinds;
    std::vector<double> f;
    std::vector<int64_t> i;
    std::vector<uint16_t> d;
  };
  std::vector<Batch> deferred;
  size_t n_deferred = 0;  // batches in use (the vectors keep their capacity)

  struct Scratch {  // a batch's calls for one range, with the range's own voice indices
    std::vector<uint32_t> voices, stgs, params, kinds;
    std::vector<double> f;
    std::vector<int64_t> i;
    std::vector<uint16_t> d;
  };
  std::vector<Scratch> scratch;
  void apply_deferred(int k) {  // on worker k
    knh_bank* b = shard[k].get();
    const uint32_t lo = base[k], hi = base[k + 1];
    Scratch& S = scratch[static_cast<size_t>(k)];
    for (size_t q = 0; q < n_deferred; ++q) {
      const Batch& B = deferred[q];
      S.voices.clear(); S.stgs.clear(); S.params.clear(); S.kinds.clear(); S.f.clear(); S.i.clear(); S.d.clear();
      for (size_t i = 0; i < B.count; ++i) {
        const uint32_t v = B.voices[i];
        if (v < lo || v >= hi) continue;
        S.voices.push_back(v - lo); S.stgs.push_back(B.stgs[i]); S.params.push_back(B.params[i]); S.kinds.push_back(B.kinds[i]);
        if (B.has_f) S.f.push_back(B.f[i]);
        if (B.has_i) S.i.push_back(B.i[i]);
        if (B.has_d) S.d.push_back(B.d[i]);
      }
      if (!S.voices.empty())  // the range's bank turns runs of one (stage, parameter, kind) into patches in one pass
        (void)b->apply_many(B.block_offset, S.voices.size(), S.voices.data(), S.stgs.data(), S.params.data(), S.kinds.data(),
                            B.has_f ? S.f.data() : nullptr, B.has_i ? S.i.data() : nullptr, B.has_d ? S.d.data() : nullptr);
    }
  }
  void flush_deferred() {
    if (n_deferred == 0) return;
    workers->run([&](int k) { apply_deferred(k); });
    n_deferred = 0;
  }
  bool call_is_valid(uint32_t block_offset, uint32_t voice, uint32_t stage, uint32_t param, uint32_t kind) const {
    if (voice >= nv || stage >= stages.size() || param >= static_cast<uint32_t>(stages[stage].n_params) || block_offset >= 65536) return false;
    const int want = expected_value_kind(stages[stage].kind, param);
    return static_cast<int>(kind) == want ||
           (kind == KNH_VALUE_SMOOTHING && (stages[stage].flags & KNH_STAGE_FLAG_SMOOTH_PARAMS) && want == KNH_VALUE_FLOAT);
  }

  int param_apply(uint32_t voice, uint32_t stage, uint32_t param, uint32_t kind, double f, int64_t i) override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    if (voice >= nv) return fail(KNH_ERR_OUT_OF_RANGE, "voice out of range");
    flush_deferred();
    const int k = of_voice(voice);
    return adopt(k, shard[k]->param_apply(voice - base[k], stage, param, kind, f, i));
  }
  int set_delay(uint32_t voice, uint32_t stage, uint32_t param, uint16_t delay) override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    if (voice >= nv) return fail(KNH_ERR_OUT_OF_RANGE, "voice out of range");
    flush_deferred();
    const int k = of_voice(voice);
    return adopt(k, shard[k]->set_delay(voice - base[k], stage, param, delay));
  }
  int call_at(uint32_t block_offset, bool is_delay, uint32_t voice, uint32_t stage, uint32_t param, uint32_t kind, double f, int64_t i,
              uint16_t delay) override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    if (voice >= nv) return fail(KNH_ERR_OUT_OF_RANGE, "voice out of range");
    flush_deferred();
    const int k = of_voice(voice);
    return adopt(k, shard[k]->call_at(block_offset, is_delay, voice - base[k], stage, param, kind, f, i, delay));
  }
  int apply_many(uint32_t block_offset, size_t count, const uint32_t* voices, const uint32_t* stgs, const uint32_t* params,
                 const uint32_t* kinds, const double* fvalues, const int64_t* ivalues, const uint16_t* delays) override {
    bool ok = initialised && count >= 256;
    {  // a batch usually addresses few (stage, parameter, kind) triples: each is checked once, the voices all
      uint32_t ls = ~0u, lp = ~0u, lk = ~0u;
      for (size_t i = 0; ok && i < count; ++i) {
        if (stgs[i] != ls || params[i] != lp || kinds[i] != lk) {
          ok = call_is_valid(block_offset, voices[i], stgs[i], params[i], kinds[i]);
          ls = stgs[i]; lp = params[i]; lk = kinds[i];
        } else {
          ok = voices[i] < nv;
        }
      }
    }
    if (!ok)  // small batches, and batches with a call that will be refused: one by one, exactly as the one-range bank does
      return knh_bank::apply_many(block_offset, count, voices, stgs, params, kinds, fvalues, ivalues, delays);
    if (n_deferred == deferred.size()) deferred.emplace_back();
    Batch& B = deferred[n_deferred++];
    B.block_offset = block_offset;
    B.count = count;
    B.voices.assign(voices, voices + count);
    B.stgs.assign(stgs, stgs + count);
    B.params.assign(params, params + count);
    B.kinds.assign(kinds, kinds + count);
    B.has_f = fvalues != nullptr;
    B.has_i = ivalues != nullptr;
    B.has_d = delays != nullptr;
    if (B.has_f) B.f.assign(fvalues, fvalues + count);
    if (B.has_i) B.i.assign(ivalues, ivalues + count);
    if (B.has_d) B.d.assign(delays, delays + count);
    return KNH_OK;
  }

  int ensure_capacity(uint32_t n_blocks, bool need_own_out, hipStream_t s) {
    if (n_blocks <= cap_blocks && (!need_own_out || d_out)) return KNH_OK;
    KNH_HIP(hipDeviceSynchronize());
    const uint32_t cap = std::max(n_blocks, cap_blocks);
    const size_t elems = static_cast<size_t>(cap) * desc.out_channels * block_size;
    if (cap > cap_blocks || !d_parts) {
      if (d_parts) KNH_HIP(hipFree(d_parts));
      d_parts = nullptr;
      KNH_HIP(hipMalloc(&d_parts, elems * n() * sizeof(F)));
      KNH_HIP(hipMemsetAsync(d_parts, 0, elems * n() * sizeof(F), s));
      for (int k = 0; k < n() && multi_device(); ++k) {  // a range on another GPU renders into memory of its own device
        if (dev_of(k) == device) continue;
        KNH_HIP(hipSetDevice(dev_of(k)));
        KNH_HIP(hipDeviceSynchronize());
        if (d_local[k]) KNH_HIP(hipFree(d_local[k]));
        d_local[k] = nullptr;
        KNH_HIP(hipMalloc(&d_local[k], elems * sizeof(F)));
        KNH_HIP(hipMemset(d_local[k], 0, elems * sizeof(F)));
      }
      KNH_HIP(hipSetDevice(device));
      if (d_out) { KNH_HIP(hipFree(d_out)); d_out = nullptr; }
      if (h_out) { KNH_HIP(hipHostFree(h_out)); h_out = nullptr; }
      cap_blocks = cap;
    }
    if (need_own_out && !d_out) {
      KNH_HIP(hipMalloc(&d_out, elems * sizeof(F)));
      KNH_HIP(hipMemsetAsync(d_out, 0, elems * sizeof(F), s));
      KNH_HIP(hipHostMalloc(&h_out, elems * sizeof(F)));
    }
    KNH_HIP(hipStreamSynchronize(s));
    return KNH_OK;
  }

  int process(uint32_t n_blocks, size_t ftp, size_t offset, uint64_t clock, void* out_host, void* out_device, void* voices_host,
              uint32_t* out_flags, void* stream, bool sync, bool accumulate) override {
    if (accumulate && !out_device) return fail(KNH_ERR_INVALID_ARGUMENT, "accumulation needs a device output buffer");
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    if (offset + ftp > block_size) return fail(KNH_ERR_INVALID_ARGUMENT, "block_start_offset + frames_to_process exceeds block_size");
    if (n_blocks == 0 || n_blocks > 4096) return fail(KNH_ERR_INVALID_ARGUMENT, "n_blocks must be in 1..4096");
    if (n_blocks > 1 && (offset != 0 || ftp != block_size)) return fail(KNH_ERR_INVALID_ARGUMENT, "multi-block launches process whole blocks");
    if (n_blocks > 1 && voices_host) return fail(KNH_ERR_INVALID_ARGUMENT, "per-voice output is only available for single blocks");
    KNH_HIP(hipSetDevice(device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : own_stream;
    int rc = ensure_capacity(n_blocks, !out_device, s);
    if (rc != KNH_OK) return rc;
    const size_t elems = static_cast<size_t>(cap_blocks) * desc.out_channels * block_size;  // per shard
    const bool pending = sum_pending;
    const size_t n_out_elems = static_cast<size_t>(n_blocks) * desc.out_channels * block_size;
    workers->run([&](int k) {
      if (hipSetDevice(dev_of(k)) != hipSuccess) { rcs[k] = KNH_ERR_DEVICE; return; }
      apply_deferred(k);
      // the previous launch's sum must be done with this shard's mix before the shard overwrites it
      if (pending && hipStreamWaitEvent(streams[k], sum_done, 0) != hipSuccess) { rcs[k] = KNH_ERR_DEVICE; return; }
      void* vh = voices_host ? static_cast<void*>(static_cast<F*>(voices_host) + static_cast<size_t>(base[k]) * block_size) : nullptr;
      // with a host destination the shard also waits for its stream (per-voice rows, flags); its mix stays on the device
      shard_flags[k] = 0;
      F* slot = d_parts + k * elems;                     // on the bank's (first) device
      F* mix = d_local[k] ? d_local[k] : slot;           // where this range's kernels write
      rcs[k] = shard[k]->process(n_blocks, ftp, offset, clock, nullptr, mix, vh, &shard_flags[k], streams[k], sync, false);
      // a range on another GPU: its blocks cross to the first device in one peer-to-peer copy (xGMI), in the range's stream
      if (rcs[k] == KNH_OK && mix != slot &&
          hipMemcpyPeerAsync(slot, device, mix, dev_of(k), n_out_elems * sizeof(F), streams[k]) != hipSuccess) rcs[k] = KNH_ERR_DEVICE;
      if (rcs[k] == KNH_OK && hipEventRecord(shard_done[k], streams[k]) != hipSuccess) rcs[k] = KNH_ERR_DEVICE;
    });
    KNH_HIP(hipSetDevice(device));
    n_deferred = 0;
    for (int k = 0; k < n(); ++k) {
      if (rcs[k] == KNH_ERR_DEVICE && shard[k]->err.empty()) return fail(KNH_ERR_DEVICE, "HIP error in a shard worker");
      if (rcs[k] != KNH_OK) return adopt(k, rcs[k]);
    }
    for (int k = 0; k < n(); ++k) KNH_HIP(hipStreamWaitEvent(s, shard_done[k], 0));
    F* dst = out_device ? static_cast<F*>(out_device) : d_out;
    const size_t n_out = static_cast<size_t>(n_blocks) * desc.out_channels * block_size;
    KNH_HIP(launch_sum(d_parts, static_cast<unsigned>(n()), elems, n_out, static_cast<unsigned>(block_size), static_cast<unsigned>(offset),
                       static_cast<unsigned>(offset + ftp), dst, accumulate, s));
    KNH_HIP(hipEventRecord(sum_done, s));
    sum_pending = true;
    if (!sync) return KNH_OK;
    if (out_host) KNH_HIP(hipMemcpyAsync(h_out, dst, n_out * sizeof(F), hipMemcpyDeviceToHost, s));
    KNH_HIP(hipStreamSynchronize(s));
    if (out_host) {
      if (n_blocks > 1) {
        std::memcpy(out_host, h_out, n_out * sizeof(F));
      } else {
        for (uint32_t c = 0; c < desc.out_channels; ++c)
          std::memcpy(static_cast<F*>(out_host) + c * block_size + offset, h_out + c * block_size + offset, ftp * sizeof(F));
      }
    }
    if (out_flags) {
      uint32_t any = 0, all = KNH_FLAG_ALL_DONE;
      for (int k = 0; k < n(); ++k) { any |= shard_flags[k] & KNH_FLAG_ANY_DONE; all &= shard_flags[k]; }
      *out_flags = any | (all & KNH_FLAG_ALL_DONE);
    }
    return KNH_OK;
  }
  static hipError_t launch_sum(const float* p, unsigned k, size_t stride, size_t n, unsigned bs, unsigned fb, unsigned fe, float* out, bool acc, hipStream_t s) {
    return knh::launch_sum_shards_f32(p, k, stride, n, bs, fb, fe, out, acc, s);
  }
  static hipError_t launch_sum(const double* p, unsigned k, size_t stride, size_t n, unsigned bs, unsigned fb, unsigned fe, double* out, bool acc, hipStream_t s) {
    return knh::launch_sum_shards_f64(p, k, stride, n, bs, fb, fe, out, acc, s);
  }

  int read_done_frames(uint32_t* out) override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    if (!out) return fail(KNH_ERR_INVALID_ARGUMENT, "null output");
    int rcs_ = synchronize();
    if (rcs_ != KNH_OK) return rcs_;
    for (int k = 0; k < n(); ++k) {
      int rc = shard[k]->read_done_frames(out + base[k]);
      if (rc != KNH_OK) return adopt(k, rc);
    }
    KNH_HIP(hipSetDevice(device));
    return KNH_OK;
  }
  int synchronize() override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    for (int k = 0; k < n() && multi_device(); ++k) {
      if (dev_of(k) == device) continue;
      KNH_HIP(hipSetDevice(dev_of(k)));
      KNH_HIP(hipDeviceSynchronize());
    }
    KNH_HIP(hipSetDevice(device));
    KNH_HIP(hipDeviceSynchronize());
    return KNH_OK;
  }
  int debug_read(uint32_t* out16) override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    return adopt(0, shard[0]->debug_read(out16));
  }
  int timing_reset(int enable) override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    for (int k = 0; k < n(); ++k) {
      int rc = shard[k]->timing_reset(enable);
      if (rc != KNH_OK) return adopt(k, rc);
    }
    return KNH_OK;
  }
  // the shards' kernels run side by side on their own streams: the launch time is that of the slowest
  int timing_read(double* ms, uint64_t* launches) override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    double worst = 0.0;
    uint64_t count = 0;
    for (int k = 0; k < n(); ++k) {
      double m = 0.0;
      uint64_t l = 0;
      int rc = shard[k]->timing_read(&m, &l);
      if (rc != KNH_OK) return adopt(k, rc);
      worst = std::max(worst, m);
      count = std::max(count, l);
    }
    if (ms) *ms = worst;
    if (launches) *launches = count;
    return KNH_OK;
  }
};

}  // namespace
