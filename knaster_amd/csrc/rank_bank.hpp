// rank_bank.hpp -- one rank's share of a voice bank that is sharded over the GPUs of a node, ONE PROCESS PER GPU
// (SURVEY.md 8(e)).  Included by bank.hip only.
//
// Rank r of R owns the contiguous voices [first(r), first(r + 1)) of the graph's N voices (whole 64-voice groups, as
// even as they go).  Every entry point keeps its meaning and takes GLOBAL voice indices: a call for a voice of another
// rank is that rank's business and returns KNH_OK without effect (every rank can be handed the same event stream, which
// is how the reference's single GraphGen sees it).  The sine table, the chain and the kernels are replicated; per-voice
// state, parameters and event lists exist only where the voice lives.  The one exchange is the sum of the ranks' mixed
// blocks to rank 0 after each launch: RCCL's ncclReduce on the communicator's own stream (comm.hip), or a reduce
// function the host supplies (tests run two ranks on one GPU that way, where RCCL refuses to put two ranks).
// Reduction order across ranks is RCCL's, not the reference's left fold: covered by the mix tolerance, like the tree mix.
#pragma once
// comm.hip: the minimum over all ranks of one status word (an internal helper beside the C ABI's knh_comm_* functions)
extern "C" int32_t knh_comm_all_min(knh_comm* comm, int32_t value, int32_t* out);


namespace {

// Voice range of `rank`: the 64-voice groups are dealt out as evenly as they go, lower ranks first.
inline void shard_voice_range(uint32_t n_voices, uint32_t rank, uint32_t world, uint32_t* first, uint32_t* count) {
  const uint64_t groups = (static_cast<uint64_t>(n_voices) + 63u) / 64u;
  const uint64_t g0 = groups * rank / world, g1 = groups * (static_cast<uint64_t>(rank) + 1) / world;
  const uint64_t lo = std::min<uint64_t>(g0 * 64u, n_voices), hi = std::min<uint64_t>(g1 * 64u, n_voices);
  *first = static_cast<uint32_t>(lo);
  *count = static_cast<uint32_t>(hi - lo);
}

template <typename F>
struct RankBank final : knh_bank {
  std::unique_ptr<knh_bank> local;  // this rank's voices (null when the rank owns none)
  uint32_t total = 0, rank = 0, world = 1, lo = 0, hi = 0;
  uint8_t comm_id[KNH_COMM_ID_BYTES] = {0};
  knh_comm* comm = nullptr;         // RCCL (null: custom reduce, or world == 1)
  knh_reduce_fn custom = nullptr;
  void* custom_user = nullptr;
  hipStream_t own_stream = nullptr;
  F* d_out = nullptr;               // used when the caller gives no device buffer
  F* h_out = nullptr;
  uint32_t cap_blocks = 0;
  // scratch for routing a batch
  std::vector<uint32_t> r_voices, r_stages, r_params, r_kinds;
  std::vector<double> r_f;
  std::vector<int64_t> r_i;
  std::vector<uint16_t> r_d;

  ~RankBank() override {
    if (initialised) (void)hipSetDevice(device);
    if (comm) knh_comm_destroy(comm);
    if (initialised) (void)hipDeviceSynchronize();
    local.reset();
    if (own_stream) (void)hipStreamDestroy(own_stream);
    if (d_out) (void)hipFree(d_out);
    if (h_out) (void)hipHostFree(h_out);
  }
  int adopt(int rc) {
    if (rc != KNH_OK && local) err = local->err;
    return rc;
  }
  bool mine(uint32_t v) const { return v >= lo && v < hi; }
  int order_after_collective(void* stream) override {
    if (!comm) return KNH_OK;
    int rc = knh_comm_wait(comm, stream);
    return rc == KNH_OK ? KNH_OK : fail(rc, knh_comm_last_error(comm));
  }
  uint32_t ranks() const override { return comm ? knh_comm_world(comm) : world; }

  int set_ctor(uint32_t stage, uint32_t first, uint32_t count, const double* args, uint32_t n_args) override {
    if (initialised) return fail(KNH_ERR_INVALID_ARGUMENT, "constructor arguments must be set before init");
    if (stage >= stages.size()) return fail(KNH_ERR_OUT_OF_RANGE, "stage out of range");
    if (static_cast<uint64_t>(first) + count > total) return fail(KNH_ERR_OUT_OF_RANGE, "voice range out of range");
    const uint32_t a = std::max(first, lo), b = std::min(first + count, hi);
    if (!local || a >= b) return KNH_OK;
    return adopt(local->set_ctor(stage, a - lo, b - a, args ? args + static_cast<size_t>(a - first) * n_args : nullptr, n_args));
  }
  int set_buffer(uint32_t stage, const void* samples, size_t n_frames, double sr) override {
    return local ? adopt(local->set_buffer(stage, samples, n_frames, sr)) : KNH_OK;
  }
  int set_input(uint32_t n_blocks, const void* host, const void* dev) override {  // every rank is handed the same input block(s)
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    return local ? adopt(local->set_input(n_blocks, host, dev)) : KNH_OK;
  }
  int init(uint32_t sr, size_t bs) override {
    if (initialised) return fail(KNH_ERR_INVALID_ARGUMENT, "already initialised");
    // A failed init leaves nothing half made that a second call could trip over: the communicator and the stream are
    // taken down again, and a local bank that did come up is not initialised twice -- the handle then refuses every
    // further init (the host creates a new bank, as bench.py's fallback does).
    if (init_failed) return fail(KNH_ERR_INVALID_ARGUMENT, "an earlier knh_bank_init of this bank failed (" + init_error + "): create a new bank");
    if (knh_device_count() <= 0) return fail(KNH_ERR_NO_DEVICE, "no gfx950 device visible; this engine has no CPU path");
    if (desc.device >= 0) device = desc.device;
    else KNH_HIP(hipGetDevice(&device));
    KNH_HIP(hipSetDevice(device));
    auto undo = [&](int rc) {
      init_failed = true;
      init_error = err;
      if (comm) { knh_comm_destroy(comm); comm = nullptr; }
      if (own_stream) { (void)hipStreamDestroy(own_stream); own_stream = nullptr; }
      return rc;
    };
    // the communicator first: creating it is a collective of all ranks; the ranks then agree, over it, that every one of
    // them brought its own voices up (below) before any of them calls the bank initialised
    if (world > 1 && !custom) {
      int rc = knh_comm_create(rank, world, comm_id, device, &comm);
      if (rc != KNH_OK) return undo(fail(rc, std::string("knh_comm_create: ") + knh_comm_last_error(nullptr)));
      if (knh_comm_world(comm) != world) return undo(fail(KNH_ERR_DEVICE, "RCCL reports a different number of ranks than the host asked for"));
    }
    int local_rc = KNH_OK;
    if (local) local_rc = local->init(sr, bs);
    if (local_rc != KNH_OK) adopt(local_rc);
    if (comm) {
      // every rank says whether its own voices came up, and every rank learns whether all did: a rank that failed and left
      // would leave the others blocked in their first ncclReduce.  (A host-supplied reduce function has no such channel: there
      // the host takes every rank down when one rank's init fails.)
      int32_t all_ok = 0;
      const std::string mine_err = err;
      const int rc = knh_comm_all_min(comm, local_rc == KNH_OK ? 1 : 0, &all_ok);
      if (rc != KNH_OK) return undo(fail(rc, std::string("agreeing on knh_bank_init across the ranks: ") + knh_comm_last_error(comm)));
      if (local_rc != KNH_OK) { err = mine_err; return undo(local_rc); }
      if (!all_ok) return undo(fail(KNH_ERR_DEVICE, "knh_bank_init failed on another rank of this bank: no rank keeps it"));
    } else if (local_rc != KNH_OK) {
      return undo(local_rc);
    }
    sample_rate = sr;
    block_size = bs;
    {
      hipError_t e = hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking);
      if (e != hipSuccess) return undo(fail(KNH_ERR_DEVICE, std::string("hipStreamCreateWithFlags: ") + hipGetErrorString(e)));
    }
    initialised = true;
    return KNH_OK;
  }
  bool init_failed = false;
  std::string init_error;
  bool kind_ok(uint32_t stage, uint32_t param, uint32_t kind) const {
    const int want = expected_value_kind(stages[stage].kind, param);
    return static_cast<int>(kind) == want || (kind == KNH_VALUE_SMOOTHING && (stages[stage].flags & KNH_STAGE_FLAG_SMOOTH_PARAMS) && want == KNH_VALUE_FLOAT);
  }
  int check_global(uint32_t voice, uint32_t stage, uint32_t param) {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    if (voice >= total) return fail(KNH_ERR_OUT_OF_RANGE, "voice out of range");
    if (stage >= stages.size()) return fail(KNH_ERR_OUT_OF_RANGE, "stage out of range");
    if (param >= static_cast<uint32_t>(stages[stage].n_params)) return fail(KNH_ERR_OUT_OF_RANGE, "parameter index out of range");
    return KNH_OK;
  }
  int param_apply(uint32_t voice, uint32_t stage, uint32_t param, uint32_t kind, double f, int64_t i) override {
    int rc = check_global(voice, stage, param);
    if (rc != KNH_OK) return rc;
    if (!kind_ok(stage, param, kind)) return fail(KNH_ERR_WRONG_VALUE_KIND, "parameter value kind does not match the parameter type");
    if (!mine(voice)) return KNH_OK;
    return adopt(local->param_apply(voice - lo, stage, param, kind, f, i));
  }
  int set_delay(uint32_t voice, uint32_t stage, uint32_t param, uint16_t delay) override {
    int rc = check_global(voice, stage, param);
    if (rc != KNH_OK) return rc;
    if (!mine(voice)) return KNH_OK;
    return adopt(local->set_delay(voice - lo, stage, param, delay));
  }
  int call_at(uint32_t block_offset, bool is_delay, uint32_t voice, uint32_t stage, uint32_t param, uint32_t kind, double f, int64_t i,
              uint16_t delay) override {
    int rc = check_global(voice, stage, param);
    if (rc != KNH_OK) return rc;
    if (block_offset >= 65536) return fail(KNH_ERR_OUT_OF_RANGE, "block_offset too large");
    if (!is_delay && !kind_ok(stage, param, kind)) return fail(KNH_ERR_WRONG_VALUE_KIND, "parameter value kind does not match the parameter type");
    if (!mine(voice)) return KNH_OK;
    return adopt(local->call_at(block_offset, is_delay, voice - lo, stage, param, kind, f, i, delay));
  }
  // a batch: the calls for this rank's voices, in array order, with local indices; the rest is checked -- voice, stage,
  // parameter and value kind, exactly as the owning rank checks them, so that every rank returns the same code for the
  // same batch -- and dropped
  int apply_many(uint32_t block_offset, size_t count, const uint32_t* voices, const uint32_t* stgs, const uint32_t* params,
                 const uint32_t* kinds, const double* fvalues, const int64_t* ivalues, const uint16_t* delays) override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    r_voices.clear(); r_stages.clear(); r_params.clear(); r_kinds.clear(); r_f.clear(); r_i.clear(); r_d.clear();
    int rc = KNH_OK;  // the code of the last refused call, in array order: the same on every rank
    for (size_t k = 0; k < count; ++k) {
      const uint32_t v = voices[k];
      if (v >= total) { rc = fail(KNH_ERR_OUT_OF_RANGE, "voice out of range"); continue; }
      if (stgs[k] >= stages.size()) { rc = fail(KNH_ERR_OUT_OF_RANGE, "stage out of range"); continue; }
      if (params[k] >= static_cast<uint32_t>(stages[stgs[k]].n_params)) { rc = fail(KNH_ERR_OUT_OF_RANGE, "parameter index out of range"); continue; }
      if (block_offset >= 65536) { rc = fail(KNH_ERR_OUT_OF_RANGE, "block_offset too large"); continue; }
      if (!kind_ok(stgs[k], params[k], kinds[k])) { rc = fail(KNH_ERR_WRONG_VALUE_KIND, "parameter value kind does not match the parameter type"); continue; }
      if (!mine(v)) continue;
      r_voices.push_back(v - lo); r_stages.push_back(stgs[k]); r_params.push_back(params[k]); r_kinds.push_back(kinds[k]);
      if (fvalues) r_f.push_back(fvalues[k]);
      if (ivalues) r_i.push_back(ivalues[k]);
      if (delays) r_d.push_back(delays[k]);
    }
    if (!r_voices.empty()) {
      int r2 = adopt(local->apply_many(block_offset, r_voices.size(), r_voices.data(), r_stages.data(), r_params.data(), r_kinds.data(),
                                       fvalues ? r_f.data() : nullptr, ivalues ? r_i.data() : nullptr, delays ? r_d.data() : nullptr));
      if (r2 != KNH_OK && rc == KNH_OK) rc = r2;  // (what is forwarded has passed the checks: only a device-side failure is left)
    }
    return rc;
  }

  int ensure_out(uint32_t n_blocks, hipStream_t s) {
    if (n_blocks <= cap_blocks) return KNH_OK;
    KNH_HIP(hipStreamSynchronize(s));
    if (comm) { int rc = knh_comm_synchronize(comm); if (rc != KNH_OK) return fail(rc, knh_comm_last_error(comm)); }
    if (d_out) KNH_HIP(hipFree(d_out));
    if (h_out) KNH_HIP(hipHostFree(h_out));
    d_out = nullptr; h_out = nullptr;
    const size_t elems = static_cast<size_t>(n_blocks) * desc.out_channels * block_size;
    KNH_HIP(hipMalloc(&d_out, elems * sizeof(F)));
    KNH_HIP(hipMemset(d_out, 0, elems * sizeof(F)));
    KNH_HIP(hipHostMalloc(&h_out, elems * sizeof(F)));
    cap_blocks = n_blocks;
    return KNH_OK;
  }

  int process(uint32_t n_blocks, size_t ftp, size_t offset, uint64_t clock, void* out_host, void* out_device, void* voices_host,
              uint32_t* out_flags, void* stream, bool sync, bool accumulate) override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    if (accumulate) return fail(KNH_ERR_INVALID_ARGUMENT, "a rank's share of a sharded bank cannot accumulate into an existing mix");
    if (voices_host) return fail(KNH_ERR_INVALID_ARGUMENT, "per-voice output is not available from a sharded bank");
    if (offset + ftp > block_size) return fail(KNH_ERR_INVALID_ARGUMENT, "block_start_offset + frames_to_process exceeds block_size");
    if (n_blocks == 0 || n_blocks > 4096) return fail(KNH_ERR_INVALID_ARGUMENT, "n_blocks must be in 1..4096");
    if (n_blocks > 1 && (offset != 0 || ftp != block_size)) return fail(KNH_ERR_INVALID_ARGUMENT, "multi-block launches process whole blocks");
    KNH_HIP(hipSetDevice(device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : own_stream;
    if (!out_device) { int rc = ensure_out(n_blocks, s); if (rc != KNH_OK) return rc; }
    F* dst = out_device ? static_cast<F*>(out_device) : d_out;
    const size_t n_out = static_cast<size_t>(n_blocks) * desc.out_channels * block_size;
    // an earlier reduce may still be reading (rank > 0) or writing (rank 0) the buffer this launch fills: wait for the last
    // one that used it (a host that alternates two buffers overlaps each launch with the reduce of the one before)
    if (comm) { int rc = knh_comm_wait_buffer(comm, dst, s); if (rc != KNH_OK) return fail(rc, knh_comm_last_error(comm)); }
    uint32_t lflags = KNH_FLAG_ALL_DONE;  // a rank without voices: nothing is running
    if (local) {
      // a blocking call that wants the flag summary lets the local bank wait for its kernels and read its own flags
      const bool want_flags = sync && out_flags;
      int rc = local->process(n_blocks, ftp, offset, clock, nullptr, dst, nullptr, want_flags ? &lflags : nullptr, s, want_flags, false);
      if (rc != KNH_OK) return adopt(rc);
    } else {
      KNH_HIP(hipMemsetAsync(dst, 0, n_out * sizeof(F), s));  // a rank without voices contributes silence
    }
    if (world > 1) {
      if (custom) {
        const auto t0 = std::chrono::steady_clock::now();
        int rc = custom(custom_user, dst, n_out, sizeof(F) == 8 ? KNH_F64 : KNH_F32, 0, s);
        if (reduce_timing) { custom_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); custom_count += 1; }
        if (rc != KNH_OK) return fail(rc, "the host's reduce function failed");
      } else {
        int rc = knh_comm_reduce_sum(comm, dst, n_out, sizeof(F) == 8 ? KNH_F64 : KNH_F32, 0, s);
        if (rc != KNH_OK) return fail(rc, knh_comm_last_error(comm));
      }
    }
    if (!sync) return KNH_OK;
    if (comm) { int rc = knh_comm_wait(comm, s); if (rc != KNH_OK) return fail(rc, knh_comm_last_error(comm)); }
    // rank 0 holds the sum; the other ranks see their own share of it
    if (out_host) KNH_HIP(hipMemcpyAsync(h_out_for(n_blocks, s), dst, n_out * sizeof(F), hipMemcpyDeviceToHost, s));
    KNH_HIP(hipStreamSynchronize(s));
    if (out_host) {
      if (n_blocks > 1) {
        std::memcpy(out_host, h_out, n_out * sizeof(F));
      } else {
        for (uint32_t c = 0; c < desc.out_channels; ++c)
          std::memcpy(static_cast<F*>(out_host) + c * block_size + offset, h_out + c * block_size + offset, ftp * sizeof(F));
      }
    }
    if (out_flags) *out_flags = lflags;  // this rank's voices only: the host combines the ranks (any = OR, all = AND)
    return KNH_OK;
  }
  F* h_out_for(uint32_t n_blocks, hipStream_t s) {
    if (n_blocks > cap_blocks) (void)ensure_out(n_blocks, s);
    return h_out;
  }

  int read_done_frames(uint32_t* out) override {  // [total]: this rank's voices filled in, the others UINT32_MAX
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    if (!out) return fail(KNH_ERR_INVALID_ARGUMENT, "null output");
    std::fill(out, out + total, 0xFFFFFFFFu);
    int rc = synchronize();
    if (rc != KNH_OK) return rc;
    return local ? adopt(local->read_done_frames(out + lo)) : KNH_OK;
  }
  int synchronize() override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    KNH_HIP(hipSetDevice(device));
    if (comm) { int rc = knh_comm_synchronize(comm); if (rc != KNH_OK) return fail(rc, knh_comm_last_error(comm)); }
    KNH_HIP(hipDeviceSynchronize());
    return KNH_OK;
  }
  int debug_read(uint32_t* out16) override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    if (!local) { std::memset(out16, 0, 16 * sizeof(uint32_t)); return KNH_OK; }
    return adopt(local->debug_read(out16));
  }
  bool reduce_timing = false;
  double custom_ms = 0.0;
  uint64_t custom_count = 0;
  int timing_reset(int enable) override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    reduce_timing = enable != 0;
    custom_ms = 0.0;
    custom_count = 0;
    if (comm) { int rc = knh_comm_timing_reset(comm, enable); if (rc != KNH_OK) return fail(rc, knh_comm_last_error(comm)); }
    return local ? adopt(local->timing_reset(enable)) : KNH_OK;
  }
  int collective_timing_read(double* ms, uint64_t* reduces) override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    if (comm) { int rc = knh_comm_timing_read(comm, ms, reduces); return rc == KNH_OK ? KNH_OK : fail(rc, knh_comm_last_error(comm)); }
    if (ms) *ms = custom_ms;
    if (reduces) *reduces = custom_count;
    return KNH_OK;
  }
  int timing_read(double* ms, uint64_t* launches) override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    if (!local) { if (ms) *ms = 0.0; if (launches) *launches = 0; return KNH_OK; }
    return adopt(local->timing_read(ms, launches));
  }
};

}  // namespace
