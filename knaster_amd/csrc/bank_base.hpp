// bank_base.hpp -- the handle behind the C ABI: what every kind of bank (one range, host-sharded / multi-device, one rank's
// share) implements.  Included by bank.hip only.
#pragma once

// ---------------------------------------------------------------------------
// The bank
// ---------------------------------------------------------------------------
struct knh_bank {
  virtual ~knh_bank() {
    // the pipelined host output (knh_bank_process_blocks_begin / _end); the derived bank has already waited for the device
    if (pipe_stream || pipe[0].dev || pipe[1].dev) (void)hipSetDevice(device);
    for (PipeSlot& p : pipe) {
      if (p.done) (void)hipEventDestroy(p.done);
      if (p.dev) (void)hipFree(p.dev);
      if (p.host) (void)hipHostFree(p.host);
    }
    if (pipe_stream) (void)hipStreamDestroy(pipe_stream);
  }
  struct PipeSlot { void* dev = nullptr; void* host = nullptr; hipEvent_t done = nullptr; size_t cap = 0, bytes = 0; };
  PipeSlot pipe[2];
  hipStream_t pipe_stream = nullptr;
  unsigned pipe_head = 0, pipe_count = 0;
  // a stream that is to read the mix of the launch just enqueued waits for whatever sums it across GPUs (rank banks)
  virtual int order_after_collective(void* /*stream*/) { return KNH_OK; }
  // knh_bank_process_block_channels: the block the bank renders into before its channels go to the caller's slices
  // ([out_channels][block_size] of F; sized by knh_bank_init, so that the per-block call never allocates)
  std::vector<unsigned char> channel_block;
  std::string err;
  std::vector<std::string> warnings;
  knh_bank_desc desc{};
  std::vector<StageInfo> stages;
  int n_slots = 0, n_params_total = 0;
  bool initialised = false;
  uint32_t sample_rate = 0;
  size_t block_size = 0;
  int device = 0;

  virtual int set_ctor(uint32_t stage, uint32_t first, uint32_t count, const double* args, uint32_t n_args) = 0;
  virtual int set_buffer(uint32_t stage, const void* samples, size_t n_frames, double buffer_sample_rate) = 0;
  virtual int init(uint32_t sr, size_t bs) = 0;
  virtual int param_apply(uint32_t voice, uint32_t stage, uint32_t param, uint32_t kind, double f, int64_t i) = 0;
  virtual int set_delay(uint32_t voice, uint32_t stage, uint32_t param, uint16_t delay) = 0;
  virtual int call_at(uint32_t block_offset, bool is_delay, uint32_t voice, uint32_t stage, uint32_t param, uint32_t kind, double f,
                      int64_t i, uint16_t delay) = 0;
  // would a value of this kind for this parameter of this voice be accepted?  (no side effect)
  virtual int check_call(uint32_t /*voice*/, uint32_t /*stage*/, uint32_t /*param*/, uint32_t /*kind*/) { return KNH_OK; }
  virtual int process(uint32_t n_blocks, size_t ftp, size_t offset, uint64_t clock, void* out_host, void* out_device,
                      void* voices_host, uint32_t* out_flags, void* stream, bool sync, bool accumulate = false) = 0;
  // knh_bank_param_apply_range: one (stage, parameter, value) for the voices [v0, v1) in rising order -- spelled out as a batch
  // here; Bank<F> (voice_bank.hpp) keeps an envelope trigger as one range event
  std::vector<uint32_t> range_scratch[4];
  std::vector<double> range_f;
  std::vector<int64_t> range_i;
  virtual int apply_range(uint32_t v0, uint32_t v1, uint32_t stage, uint32_t param, uint32_t kind, double f, int64_t iv) {
    const size_t n = v1 - v0;
    if (n == 0) return KNH_OK;
    range_scratch[0].resize(n);
    for (size_t k = 0; k < n; ++k) range_scratch[0][k] = v0 + static_cast<uint32_t>(k);
    range_scratch[1].assign(n, stage);
    range_scratch[2].assign(n, param);
    range_scratch[3].assign(n, kind);
    range_f.assign(n, f);
    range_i.assign(n, iv);
    return apply_many(0, n, range_scratch[0].data(), range_scratch[1].data(), range_scratch[2].data(), range_scratch[3].data(), range_f.data(), range_i.data(), nullptr);
  }
  // knh_bank_param_apply_many[_at]: the calls in array order (block_offset 0 = now); a host-sharded bank spreads them
  // over its threads (host_shards.hpp)
  virtual int apply_many(uint32_t block_offset, size_t count, const uint32_t* voices, const uint32_t* stgs, const uint32_t* params,
                         const uint32_t* kinds, const double* fvalues, const int64_t* ivalues, const uint16_t* delays) {
    int rc = KNH_OK;
    for (size_t k = 0; k < count; ++k) {
      if (delays && delays[k] > 0) {
        // (a call that is going to be refused -- a value of the wrong kind -- must not leave its delay armed for the next one)
        int r = check_call(voices[k], stgs[k], params[k], kinds[k]);
        if (r == KNH_OK) r = call_at(block_offset, true, voices[k], stgs[k], params[k], 0, 0.0, 0, delays[k]);
        if (r != KNH_OK) { rc = r; continue; }
      }
      int r = call_at(block_offset, false, voices[k], stgs[k], params[k], kinds[k], fvalues ? fvalues[k] : 0.0, ivalues ? ivalues[k] : 0, 0);
      if (r != KNH_OK) rc = r;
    }
    return rc;
  }
  // the bank node's input block(s) for the next process call (host memory: copied; or device memory)
  virtual int set_input(uint32_t n_blocks, const void* host, const void* dev) = 0;
  virtual int read_done_frames(uint32_t* out) = 0;
  virtual int synchronize() = 0;
  virtual int debug_read(uint32_t* out16) = 0;
  virtual const char* debug_signature() const { return ""; }  // (a bank cut into ranges: its parts have one each)
  virtual int timing_reset(int enable) = 0;
  virtual int timing_read(double* ms, uint64_t* launches) = 0;
  virtual int collective_timing_read(double* ms, uint64_t* reduces) {
    if (ms) *ms = 0.0;
    if (reduces) *reduces = 0;
    return KNH_OK;
  }
  virtual uint32_t ranks() const { return 1; }
  virtual void resident_stats(uint64_t* calls, uint64_t* launches) { if (calls) *calls = 0; if (launches) *launches = 0; }
  virtual void resident_trace(uint64_t* five) { for (int k = 0; k < 5; ++k) five[k] = 0; }

  int fail(int code, const std::string& msg) {
    err = msg;
    return code;
  }
  void warn(const std::string& msg) {
    if (warnings.size() < 32) warnings.push_back(msg);
  }
};
