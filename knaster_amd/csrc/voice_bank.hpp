// voice_bank.hpp -- Bank<F>: one voice range on one GPU.  The host keeps a shadow of every *parameter-derived* quantity
// (never of audio-evolving state) and turns each UGen::param_apply into device state patches: all transcendental work (tan /
// pow / sqrt / exp for filter coefficients, the f64 phase-increment product) happens here with the same libm the reference's
// std-backed num-traits would call; WrPreciseTiming's queues, WrSmoothParams' ramps, the per-launch event lists, the kernel
// choice and the launch itself.  Included by bank.hip only.
#pragma once

namespace {

#define KNH_HIP(expr)                                                                                      \
  do {                                                                                                     \
    hipError_t e_ = (expr);                                                                                \
    if (e_ != hipSuccess)                                                                                  \
      return fail(KNH_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));                      \
  } while (0)

// ---- resident launches (voice_chain.hpp, Resident) -----------------------------------------------------------------------
// A resident kernel keeps every CU's LDS: another bank's launch on the same device could not start beside it.  So there is at
// most one per device, and whoever is about to launch anything there asks it to leave first.  One mutex guards every
// transition (a bank is single-caller, but two banks may belong to two threads).
struct ResidentSlot {
  std::mutex mu;
  knh_bank* owner[64] = {};
  void (*leave[64])(knh_bank*) = {};
};
inline ResidentSlot& resident_slot() {
  static ResidentSlot s;
  return s;
}

template <typename F>
struct Bank final : knh_bank {
  typedef typename knh_dev::WordOf<F>::type W;
  const knh::KernelEntry* entry = nullptr;
  const knh::PipeEntry* pipe = nullptr;  // wave-specialised variant, used when built for this chain
  bool pipe_pair = false;                // ... in its two-groups-per-workgroup form (banks of more groups than CUs)
  const knh::DagEntry* dag = nullptr;    // five-role variant (f32, source -> SVF -> x*env -> post chains)
  const knh::WideEntry* wide = nullptr;  // 4/8 voice groups per workgroup, for banks larger than the chip's SIMD count
  int wide_waves = 0;                    // 0 = not used, else 4 or 8
  // a graph-shaped voice of SinWt oscillators and arithmetic run by the frame-parallel interpreter (kernels_interp.hip)
  bool interp = false;
  std::vector<knh_dev::InterpOp> h_prog;
  knh_dev::InterpOp* d_prog = nullptr;
  unsigned interp_sigs = 0, interp_out = 0;
  // ... or, by default, as one frame-parallel kernel hiprtc builds from the stages written out as straight-line code
  // (voice_frame.hpp, jit_frame_kernel); KNH_FRAME_JIT=0 keeps the interpreter
  const knh::JitKernel* frame_jit = nullptr;
  uint32_t* d_sin_slots = nullptr;
  unsigned frame_vpw = 1, n_sin = 0;
  uint64_t env_ranks = 0;  // VoiceKernelArgs::env_ranks (graph-shaped voices with several envelope stages)
  const knh::JitKernel* jit = nullptr;   // run-time fused kernel (hiprtc) for chains without a pre-built one
  int pipeline_level = 1;                // KNH_PIPELINE
  std::string signature;
  const char* debug_signature() const override { return signature.c_str(); }
  uint32_t nv = 0;
  long stride = 0;

  // construction-time values
  std::vector<std::vector<double>> ctor;  // [stage][voice * n_ctor + a]
  // parameter shadows (only what a later setter needs to read back)
  struct Shadow {
    std::vector<F> a, b, c;        // SinWt: a=freq | Svf: a=cutoff b=q c=gain_db | Env: a=attack_s b=release_s
    std::vector<uint8_t> ty;       // Svf filter type
  };
  std::vector<Shadow> shadow;
  double f2pi = 0.0;
  // WrPreciseTiming state: next_delay per (param, voice) for wrapped stages; queues keyed by voice*n_stages+stage
  std::vector<uint16_t> next_delay;  // [n_params_total][nv], allocated only if some stage is wrapped
  // WrPreciseTiming::waiting_changes of every wrapped node, flattened: (key = voice * n_stages + stage, change)
  // in arrival order; grouped per node when the block is assembled.
  std::vector<std::pair<uint64_t, QueuedChange>> queued;
  // WrSmoothParams (smooth_params.rs:12-311) for stages flagged KNH_STAGE_FLAG_SMOOTH_PARAMS: the ramp state
  // lives on the host, exactly as it lives on the reference's audio thread; once per (partial) block every
  // ramp in flight hands its interpolated value to the wrapped node's setter.
  struct SmoothState {
    bool linear = false;
    double current_value = 0, start_value = 0, end_value = 0;
    size_t duration_frames = 0, frames_elapsed = 0;
    uint8_t audio_rate = 0;
    bool done = true;
    double interpolated() const {
      double mix = static_cast<double>(frames_elapsed) / static_cast<double>(duration_frames);
      return (end_value - start_value) * mix + start_value;
    }
  };
  std::vector<std::vector<SmoothState>> smooth;      // [stage][voice * n_params + param], flagged stages only
  std::vector<uint64_t> smooth_active;               // nodes (voice * n_stages + stage) with a ramp possibly in flight
  std::vector<std::vector<uint32_t>> smooth_mark;    // [stage][voice]: bit 0 = listed in smooth_active; rest = last ticked epoch
  uint32_t smooth_epoch = 0;
  // Device state patches of the next launch, in application order: block 0's immediate changes as they
  // arrive, then (at process time) block 0's queued changes, block 1's immediate ones, ...
  std::vector<HostEvent> pending;
  bool pending_needs_sort = false;   // some voice may have events out of frame order
  // The same change for a run of neighbouring voices (a whole bank's note-on or note-off in one knh_bank_param_apply_many):
  // one record instead of an event per voice.  `at` = how many events `pending` held when it arrived (its place in the
  // application order).  A resident call whose events are nothing but a few of these passes them on as they are (ResCall,
  // voice_chain.hpp: 32 bytes each over PCIe instead of 16 per voice); anything else turns them into per-voice events first.
  struct RangeEvent { uint32_t v0, v1, frame, op, slot; uint64_t bits; size_t at; };
  std::vector<RangeEvent> pending_ranges, sent_ranges;
  uint32_t res_n_ranges = 0;         // the list upload_events has just made holds this many range events (0: per-voice lists)
  void expand_ranges() {
    if (pending_ranges.empty()) return;
    size_t extra = 0;
    for (const RangeEvent& r : pending_ranges) extra += r.v1 - r.v0;
    std::vector<HostEvent> out;
    out.reserve(pending.size() + extra);
    size_t k = 0;
    auto emit = [&](const RangeEvent& r) { for (uint32_t v = r.v0; v < r.v1; ++v) out.push_back(HostEvent{v, r.frame, r.op, r.slot, r.bits}); };
    for (size_t i = 0; i < pending.size(); ++i) {
      while (k < pending_ranges.size() && pending_ranges[k].at <= i) emit(pending_ranges[k++]);
      out.push_back(pending[i]);
    }
    while (k < pending_ranges.size()) emit(pending_ranges[k++]);
    pending.swap(out);
    pending_ranges.clear();
  }
  // the bank's resident kernel form walks range events: the pipelined kernels (voice_pipe.hpp), not the one-wavefront kernel
  bool res_ranges_ok() const { return jit ? jit_pipe : pipe != nullptr; }
  uint32_t frame_base = 0;           // absolute frame of frame 0 of the block being assembled
  // param_apply / set_delay calls addressed to later blocks of the next multi-block launch
  struct Call { uint8_t is_delay; uint16_t delay; uint32_t voice, stage, param, kind; double f; int64_t i; };
  std::vector<std::vector<Call>> future;  // [block_offset]
  // Calls to a node wrapped in WrPreciseTiming (and in nothing that keeps host state of its own): one compact record per
  // call, per block in arrival order.  When the block is assembled a single pass replays them against the armed delays
  // and each node's queue state (precise_timing.rs:65-135) and writes the device events; no queue is ever materialised.
  struct QRec {         // (the same bytes as knh_dev::DevRec: records of device-resolved stages are read by the resolver kernels as they are)
    uint32_t voice;
    uint16_t delay;     // set_delay_within_block_for_param value, when `arm` is set
    uint16_t stage;     // (graph-shaped voices hold up to 512 stages, frame-parallel ones 4 096)
    uint8_t param;
    uint8_t kb;         // bits 0-3 ParameterValue kind, bit 4 arm, bit 5 has a value
    uint16_t block;     // device-resolved stages: the block of the launch the call is addressed to
    uint32_t pad;
    union { double f; int64_t i; } v;
    uint32_t kind() const { return kb & 15u; }
    bool arm() const { return (kb & 0x10u) != 0; }
    bool has_value() const { return (kb & 0x20u) != 0; }
  };
  static_assert(sizeof(QRec) == sizeof(knh_dev::DevRec) && offsetof(QRec, v) == offsetof(knh_dev::DevRec, value) &&
                offsetof(QRec, block) == offsetof(knh_dev::DevRec, block) && offsetof(QRec, kb) == offsetof(knh_dev::DevRec, kb), "QRec is DevRec");
  std::vector<std::vector<QRec>> qfuture;  // [block_offset]
  struct NodeQ { uint32_t epoch; uint16_t at; uint16_t taken : 15, blocked : 1; };  // a node's queue during the block `epoch`
  std::vector<NodeQ> node_q;               // [voice * n_wrapped + wrapped index of the stage]
  std::vector<int> wrapped_index;          // [stage] -> index among the stages of this kind, or -1
  uint32_t n_wrapped = 0, q_epoch = 0;
  bool fastq(const StageInfo& S) const { return S.dcpb > 0 && !(S.flags & KNH_STAGE_FLAG_SMOOTH_PARAMS); }
  std::vector<QRec>& qblock(uint32_t block_offset) {
    if (qfuture.size() <= block_offset) qfuture.resize(block_offset + 1);
    return qfuture[block_offset];
  }
  // ---- change queues resolved on the device (kernels_events.hip) ---------------------------------------------------
  // Stages wrapped in WrPreciseTiming whose setters need no host library call (SinWt, SinNumeric, constants and wr_mul,
  // the EnvAsr / EnvAr times and triggers): the host appends the call's record to pinned memory and that is all; armed
  // delays, queue order, capacity, the patches and the per-voice lists are the resolver kernels' (KNH_DEV_EVENTS=0: the host's,
  // as in round 2).  Stages that do need the host (SvfFilter: tan, one-pole: exp, ...) keep the host path; a node's queue
  // lives in exactly one of the two places.
  std::vector<uint8_t> stage_dev;             // [stage]
  std::vector<uint8_t> dev_class;             // [stage * 8 + param]: 1 + the value kind a device-resolved node's parameter takes, 0: not one
  bool dev_events = false;
  QRec* h_recs2[2] = {nullptr, nullptr};      // pinned; two alternate: the resolver of a launch reads one while the host fills the other
  size_t h_recs_cap2[2] = {0, 0};
  hipEvent_t recs_done[2] = {nullptr, nullptr};
  bool recs_busy[2] = {false, false};
  unsigned recs_parity = 0;
  QRec* h_recs = nullptr;                     // = h_recs2[recs_parity]
  size_t n_recs = 0;
  uint32_t recs_max_block = 0;
  knh_dev::DevStage* d_stages = nullptr;
  uint16_t* d_armed = nullptr;
  uint32_t *d_ev_cnt = nullptr, *d_rec_start = nullptr;
  knh_dev::u64* d_keys = nullptr;
  knh_dev::DevRec* d_recs = nullptr;          // the launch's records, copied by the counting kernel (one pass over PCIe)
  size_t d_keys_cap = 0;
  // The resolver runs on a stream of its own, so that it works on launch k + 1 while the voice kernel of launch k runs; the
  // lists it makes therefore come in two sets, used alternately: a set is rewritten only after the voice kernel that read it
  // has finished (lists_free), and a voice kernel starts only when its set is complete (recs_done of that launch).
  hipStream_t ev_stream = nullptr;
  hipEvent_t lists_free[2] = {nullptr, nullptr};
  bool lists_busy[2] = {false, false};
  uint32_t* d_out_start2[2] = {nullptr, nullptr};
  Event* d_out_events2[2] = {nullptr, nullptr};
  size_t d_out_cap2[2] = {0, 0};
  unsigned out_parity = 0;
  int out_in_use = -1;                        // the set the voice kernel being launched reads
  uint32_t* h_ev_overflow = nullptr;          // mapped pinned: a resolver kernel found a change queue full (looked at by the next process call)
  static bool dev_resolvable_kind(uint16_t kind) {
    switch (kind) {
      case KNH_STAGE_SIN_WT: case KNH_STAGE_SIN_NUMERIC: case KNH_STAGE_MUL_ENV_ASR: case KNH_STAGE_MUL_ENV_AR:
      case KNH_STAGE_MUL_CONST: case KNH_STAGE_ADD_CONST: case KNH_STAGE_SUB_CONST: case KNH_STAGE_DIV_CONST: case KNH_STAGE_POW_CONST:
      case KNH_STAGE_WR_MUL: return true;
      default: return false;
    }
  }
  int dev_reserve(size_t more) {  // room for `more` records in the buffer being filled
    const unsigned b = recs_parity;
    if (n_recs + more <= h_recs_cap2[b]) return KNH_OK;
    const size_t cap = std::max<size_t>((n_recs + more) * 2, 16384);
    QRec* fresh = nullptr;
    KNH_HIP(hipSetDevice(device));
    KNH_HIP(hipHostMalloc(&fresh, cap * sizeof(QRec)));
    if (n_recs) std::memcpy(fresh, h_recs2[b], n_recs * sizeof(QRec));
    if (h_recs2[b]) KNH_HIP(hipHostFree(h_recs2[b]));  // (the buffer being filled is not one a kernel reads)
    h_recs2[b] = fresh;
    h_recs_cap2[b] = cap;
    h_recs = fresh;
    return KNH_OK;
  }
  int push_rec(uint32_t block_offset, QRec r) {
    if (!stage_dev[r.stage]) { qblock(block_offset).push_back(r); return KNH_OK; }
    int rc = dev_reserve(1);
    if (rc != KNH_OK) return rc;
    r.block = static_cast<uint16_t>(block_offset);
    h_recs[n_recs++] = r;
    recs_max_block = std::max(recs_max_block, block_offset);
    return KNH_OK;
  }
  // The launch's records -> the per-voice event lists in device memory, merged with the host-made list (ev_start / events,
  // pinned, or null).  Enqueued on `s` in front of the voice kernel.
  int resolve_on_device(hipStream_t s, uint32_t n_blocks, uint32_t fb, uint32_t fe, bool have_host, size_t host_total) {
    const unsigned b = recs_parity;
    size_t n_now = n_recs;
    if (recs_max_block >= n_blocks) {  // calls addressed beyond this launch: they wait, in the other buffer, for the next one
      const unsigned o = b ^ 1u;
      if (recs_busy[o]) { KNH_HIP(hipEventSynchronize(recs_done[o])); recs_busy[o] = false; }
      size_t keep = 0, later = 0;
      for (size_t i = 0; i < n_recs; ++i) later += h_recs[i].block >= n_blocks;
      if (later > h_recs_cap2[o]) {
        if (h_recs2[o]) KNH_HIP(hipHostFree(h_recs2[o]));
        h_recs2[o] = nullptr;
        h_recs_cap2[o] = std::max<size_t>(later * 2, 16384);
        KNH_HIP(hipHostMalloc(&h_recs2[o], h_recs_cap2[o] * sizeof(QRec)));
      }
      later = 0;
      uint32_t mx = 0;
      for (size_t i = 0; i < n_recs; ++i) {
        if (h_recs[i].block >= n_blocks) {
          QRec r = h_recs[i];
          r.block = static_cast<uint16_t>(r.block - n_blocks);
          mx = std::max<uint32_t>(mx, r.block);
          h_recs2[o][later++] = r;
        } else {
          h_recs[keep++] = h_recs[i];
        }
      }
      n_now = keep;
      n_recs = later;  // what the next launch starts with
      recs_max_block = mx;
    } else {
      n_recs = 0;
      recs_max_block = 0;
    }
    const unsigned set = out_parity;
    out_parity ^= 1u;
    if (n_now > d_keys_cap) {
      KNH_HIP(hipStreamSynchronize(ev_stream));
      if (d_keys) KNH_HIP(hipFree(d_keys));
      if (d_recs) KNH_HIP(hipFree(d_recs));
      d_keys = nullptr; d_recs = nullptr;
      d_keys_cap = std::max<size_t>(n_now * 2, 16384);
      KNH_HIP(hipMalloc(&d_keys, d_keys_cap * sizeof(knh_dev::u64)));
      KNH_HIP(hipMalloc(&d_recs, d_keys_cap * sizeof(knh_dev::DevRec)));
    }
    if (host_total + n_now > d_out_cap2[set]) {
      if (lists_busy[set]) { KNH_HIP(hipEventSynchronize(lists_free[set])); lists_busy[set] = false; }
      KNH_HIP(hipStreamSynchronize(ev_stream));
      if (d_out_events2[set]) KNH_HIP(hipFree(d_out_events2[set]));
      d_out_events2[set] = nullptr;
      d_out_cap2[set] = std::max<size_t>((host_total + n_now) * 2, 16384);
      KNH_HIP(hipMalloc(&d_out_events2[set], d_out_cap2[set] * sizeof(Event)));
    }
    if (lists_busy[set]) { KNH_HIP(hipStreamWaitEvent(ev_stream, lists_free[set], 0)); lists_busy[set] = false; }
    knh_dev::EventResolveArgs ra{};
    ra.recs = reinterpret_cast<const knh_dev::DevRec*>(h_recs2[b]);
    ra.n_recs = static_cast<uint32_t>(n_now);
    ra.stages = d_stages;
    ra.n_voices = nv;
    ra.block_size = static_cast<uint32_t>(block_size);
    ra.frame_begin = fb;
    ra.frame_end = fe;
    ra.n_blocks = n_blocks;
    ra.sample_rate = sample_rate;
    ra.f64 = sizeof(F) == 8 ? 1u : 0u;
    ra.f2pi = f2pi;
    ra.armed = d_armed;
    ra.host_start = have_host ? h_ev_start : nullptr;
    ra.host_events = h_events;
    ra.cnt = d_ev_cnt;
    ra.val_cnt = d_ev_cnt + nv;
    ra.cursor = d_ev_cnt + 2 * static_cast<size_t>(nv);
    ra.rec_start = d_rec_start;
    ra.keys = d_keys;
    ra.dev_recs = d_recs;
    ra.out_start = d_out_start2[set];
    ra.out_events = d_out_events2[set];
    ra.overflow = h_ev_overflow;
    KNH_HIP(knh::launch_resolve_events(ra, ev_stream));
    KNH_HIP(hipEventRecord(recs_done[b], ev_stream));  // the records are read, and the lists complete: one event says both
    recs_busy[b] = true;
    KNH_HIP(hipStreamWaitEvent(s, recs_done[b], 0));  // the voice kernel reads this set
    out_in_use = static_cast<int>(set);
    // the host goes on filling the other buffer
    recs_parity = b ^ 1u;
    if (recs_busy[recs_parity]) { KNH_HIP(hipEventSynchronize(recs_done[recs_parity])); recs_busy[recs_parity] = false; }
    h_recs = h_recs2[recs_parity];
    return KNH_OK;
  }

  // device
  hipStream_t own_stream = nullptr;
  W* d_state = nullptr;
  float* d_sine = nullptr;
  double* d_seg_table = nullptr;  // segment Envelope: [voice][seg_max][3]
  uint32_t seg_max = 0;
  F* d_buffer = nullptr;          // BufferReader's shared Buffer (device copy), staged in h_buffer until init
  std::vector<F> h_buffer;
  double buffer_sr = 0.0;
  std::vector<double> buf_start, buf_dur, buf_rate;  // BufferReader shadows per voice: start_frame, dur_frame, rate
  double buf_base_rate = 0.0;
  void* d_delay = nullptr;        // SampleDelay rings: [voice][delay_stride] of F
  uint32_t delay_stride = 0;
  std::vector<uint32_t> delay_len;  // ring length per voice (samples)
  std::vector<double> env_start;  // Envelope::start_value per voice (t_restart restores it)
  std::vector<uint32_t> env_nseg;
  F* d_partials = nullptr;
  F* d_out = nullptr;
  F* d_voices = nullptr;
  uint32_t* d_done = nullptr;
  uint32_t* d_flags = nullptr;
  uint32_t flags_parity = 0;        // which of the two flag sets the next launch uses
  uint32_t* flags_last = nullptr;   // the set of the last launch (knh_bank_debug_words)
  hipStream_t flags_stream = nullptr;  // the stream of the last launch, whose fold kernel cleared the set of this one
  bool flags_stream_set = false;
  // pinned host staging
  // Event lists are read by the kernel straight from pinned host memory (each is read once, a few hundred KB per
  // launch): no copy in the stream, the kernel's first waves pull them over PCIe while the others start.  Two
  // buffers alternate; one is rewritten only after the kernel that read it has finished (list_done).
  uint32_t* h_ev_start2[2] = {nullptr, nullptr};
  Event* h_events2[2] = {nullptr, nullptr};
  size_t h_events_cap2[2] = {0, 0};
  hipEvent_t list_done[2] = {nullptr, nullptr};
  bool list_busy[2] = {false, false};
  unsigned list_parity = 0;
  int list_in_use = -1;            // buffer the kernel being launched reads
  uint32_t* h_ev_start = nullptr;  // the buffer of the launch being assembled
  Event* h_events = nullptr;
  F* h_out = nullptr;  // [channels][block] then 2 x u32 flags
  // Blocking calls with a host destination (knh_bank_process_block: the call the reference makes once per block) hand the
  // mixed block over without a copy command: the fold kernel writes it into h_out -- mapped pinned host memory -- and then
  // an epoch number into h_done, which the host polls (knh_dev::HostDone).  KNH_MAPPED_OUT=0: the copies and the stream
  // wait of round 2 (A/B runs).
  uint32_t* h_done = nullptr;        // pinned: [0] epoch, [1] flags[0], [2] flags[1]
  uint32_t* d_fold_count = nullptr;  // device: workgroups of the fold kernel that are through
  uint32_t done_epoch = 0;
  bool mapped_out = true;
  // ---- the per-block call on a resident launch (voice_chain.hpp, Resident) ------------------------------------------------
  // knh_bank_process_block -- the call the reference makes once per block (Task::run, knaster_graph/src/task.rs:25-31) -- of
  // a bank on the pipelined kernel form with a mixer wavefront: the first such call launches the kernel, and it stays until
  // something else needs the device state (any other entry point that reads it or launches), another bank launches on the
  // device, or the host stays away for KNH_RESIDENT_IDLE_US (default 5 000).  KNH_RESIDENT=0: never (a launch per call).
  int res_policy = -1;                 // -1 not decided, 0 never, 1 where possible
  bool res_on = false;                 // a resident kernel is running (or has ended by itself) on own_stream
  uint32_t res_epoch = 0;              // the last epoch handed out (24 bits)
  uint64_t* res_bell = nullptr;        // the command word as the host writes it ...
  uint64_t* res_bell_dev = nullptr;    // ... and as the kernel reads it (the same fine-grained device word behind a large BAR; else mapped pinned memory)
  bool res_bell_is_device = false;
  uint64_t* d_res_relay = nullptr;
  uint64_t *d_res_rows = nullptr, *d_res_wg_flags = nullptr, *d_res_group_rows = nullptr, *d_res_group_flags = nullptr;  // granules (voice_chain.hpp)
  uint32_t *d_res_arrivals = nullptr, *d_res_group_arrivals = nullptr;
  hipStream_t res_stream = nullptr;    // the fold server runs beside the voice kernel
  uint64_t* h_res_out = nullptr;       // mapped pinned: the block as the fold server's root leaves it, granules {sample bits, tag}: [plane][block_size][W], then the flags granule
  uint32_t* h_res_done = nullptr;      // mapped pinned: epoch, done count, running count
  uint32_t res_max_tiles = 0;
  uint32_t res_cooldown = 0;           // calls to sit out after another bank asked this one to leave
  uint64_t res_idle_ticks = 500000;    // 5 ms of the 100 MHz clock
  uint64_t res_calls = 0, res_launches = 0;
  void resident_stats(uint64_t* calls, uint64_t* launches) override { if (calls) *calls = res_calls; if (launches) *launches = res_launches; }
  // diagnostics: the last call's milestones on the device clock (ticks of 10 ns): the voice kernel saw the command, the fold
  // server's root did, its first tile was complete, its last tile was, it had written everything
  void resident_trace(uint64_t* five) override {
    for (int k = 0; k < 5; ++k) five[k] = 0;
    if (!h_res_done) return;
    for (int k = 0; k < 5; ++k) std::memcpy(&five[k], h_res_done + 8 + 2 * k, 8);
  }
  bool jit_pipe = false;  // the run-time fused kernel is a pipeline (mixer form, short tiles), not a one-wavefront kernel
  bool res_possible() const {
    // Kernel forms in which every workgroup can be resident at once and the fold server can take the rows: the pipelined forms
    // with a mixer wavefront and one voice group per workgroup, and the one-wavefront kernel (pre-built or fused at run time);
    // up to 256 voice groups (the server folds 8 x 32 rows), tree mix, no bank inputs (their upload rides in a stream).
    if (interp || dag || uses_input || desc.mix_mode != KNH_MIX_TREE) return false;
    if ((nv + 63u) / 64u > 256u || block_size > 4096) return false;
    if (wide_waves != 0) return false;
    if (jit) return true;
    if (pipe) return !pipe_pair && pipe->form != 1 /* PIPE_FOLD: no mixer wavefront */ && pipe->gpw == 1;
    return entry != nullptr;
  }
  // frames per tile of the bank's pipelined kernel form (voice_pipe.hpp, PipeTile: the forms with the long tiles, and the
  // Fan pipelines, 64 -- f64: 32; the mixer form 32 / 16)
  uint32_t res_tile_frames() const {
    if (jit && jit_pipe) return sizeof(F) == 8 ? 16u : 32u;  // (jit.hip fuses pipelines in the mixer form with the short tiles)
    if (jit || !pipe) return 64u;                             // the one-wavefront kernel hands its rows over in 64-frame tiles
    const bool big = pipe->long_tiles != 0;
    return sizeof(F) == 8 ? (big ? 32u : 16u) : (big ? 64u : 32u);
  }
  static hipError_t launch_res_server(const knh_dev::ResServerArgs<float>& a, hipStream_t s) { return knh::launch_res_server_f32(a, s); }
  static hipError_t launch_res_server(const knh_dev::ResServerArgs<double>& a, hipStream_t s) { return knh::launch_res_server_f64(a, s); }
  static void res_leave_thunk(knh_bank* b) { static_cast<Bank<F>*>(b)->res_leave_locked(true); }
  // the caller holds resident_slot().mu
  int res_leave_locked(bool evicted) {
    if (!res_on) return KNH_OK;
    KNH_HIP(hipSetDevice(device));
    res_epoch = (res_epoch + 1u) & 0xFFFFFFu;
    const uint64_t cmd = static_cast<uint64_t>(res_epoch) | (1ull << 58);
    __atomic_store_n(res_bell, cmd, __ATOMIC_RELEASE);
#if defined(__x86_64__)
    if (res_bell_is_device) __builtin_ia32_sfence();
#endif
    hipError_t e = hipStreamSynchronize(own_stream);  // (bounded on the device side: every wait of the kernels is)
    const hipError_t e2 = hipStreamSynchronize(res_stream);
    if (e == hipSuccess) e = e2;
    res_on = false;
    ResidentSlot& rs = resident_slot();
    if (device >= 0 && device < 64 && rs.owner[device] == this) { rs.owner[device] = nullptr; rs.leave[device] = nullptr; }
    if (evicted) res_cooldown = 256;
    if (e != hipSuccess) return fail(KNH_ERR_DEVICE, std::string("the resident kernel ended with an error: ") + hipGetErrorString(e));
    return KNH_OK;
  }
  int res_leave() {
    if (!res_on) return KNH_OK;
    std::lock_guard<std::mutex> lock(resident_slot().mu);
    return res_leave_locked(false);
  }
  // before anything is launched on this device by this bank: no other bank's resident kernel is in the way
  void res_make_room() {
    ResidentSlot& rs = resident_slot();
    if (device < 0 || device >= 64) return;
    std::lock_guard<std::mutex> lock(rs.mu);
    if (rs.owner[device] && rs.owner[device] != this) rs.leave[device](rs.owner[device]);
  }
  int res_alloc() {
    if (h_res_done) return KNH_OK;
    KNH_HIP(hipSetDevice(device));
    int large_bar = 0;
    (void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device);
    const char* be = std::getenv("KNH_RESIDENT_BELL");  // "host": the command word in pinned host memory whatever the BAR
    if (large_bar && !(be && be[0] == 'h')) {
      void* p = nullptr;
      if (hipExtMallocWithFlags(&p, 64, hipDeviceMallocFinegrained) == hipSuccess && p) {
        KNH_HIP(hipMemset(p, 0xFF, 64));
        KNH_HIP(hipDeviceSynchronize());
        res_bell = res_bell_dev = static_cast<uint64_t*>(p);
        res_bell_is_device = true;
      }
    }
    if (!res_bell) {
      KNH_HIP(hipHostMalloc(&res_bell, 64, hipHostMallocMapped | hipHostMallocCoherent));
      res_bell_dev = res_bell;
      *res_bell = ~0ull;
    }
    res_max_tiles = static_cast<uint32_t>((block_size + res_tile_frames() - 1) / res_tile_frames());
    {
      // granules: every one starts with a tag no call will ever carry (all ones)
      const size_t w = sizeof(F) == 8 ? 2 : 1, rows = (nv + 63) / 64;
      const size_t n_rows = static_cast<size_t>(res_max_tiles) * 2 * rows * 64 * w, n_group = static_cast<size_t>(res_max_tiles) * 2 * 8 * 64 * w;
      KNH_HIP(hipMalloc(&d_res_rows, n_rows * 8));
      KNH_HIP(hipMemset(d_res_rows, 0xFF, n_rows * 8));
      KNH_HIP(hipMalloc(&d_res_group_rows, n_group * 8));
      KNH_HIP(hipMemset(d_res_group_rows, 0xFF, n_group * 8));
      KNH_HIP(hipMalloc(&d_res_wg_flags, rows * 8));
      KNH_HIP(hipMemset(d_res_wg_flags, 0xFF, rows * 8));
      KNH_HIP(hipMalloc(&d_res_group_flags, 64));
      KNH_HIP(hipMemset(d_res_group_flags, 0xFF, 64));
      KNH_HIP(hipMalloc(&d_res_arrivals, (static_cast<size_t>(res_max_tiles) + 1) * 8 * sizeof(uint32_t)));
      KNH_HIP(hipMemset(d_res_arrivals, 0, (static_cast<size_t>(res_max_tiles) + 1) * 8 * sizeof(uint32_t)));
      KNH_HIP(hipMalloc(&d_res_group_arrivals, (static_cast<size_t>(res_max_tiles) + 1) * sizeof(uint32_t)));
      KNH_HIP(hipMemset(d_res_group_arrivals, 0, (static_cast<size_t>(res_max_tiles) + 1) * sizeof(uint32_t)));
      // The fold server must run BESIDE the voice kernel, so it must not sit behind it in one hardware queue (the runtime
      // multiplexes streams onto a few).  A stream of another priority gets a queue of its own; res_launch checks that both
      // kernels have started before anything relies on it.
      int prio_low = 0, prio_high = 0;
      (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
      if (hipStreamCreateWithPriority(&res_stream, hipStreamNonBlocking, prio_high) != hipSuccess) KNH_HIP(hipStreamCreateWithFlags(&res_stream, hipStreamNonBlocking));
    }
    KNH_HIP(hipMalloc(&d_res_relay, 1024));  // the command word, and (words 16 ..) a call's range events (voice_chain.hpp RES_RELAY_RANGES)
    KNH_HIP(hipMemset(d_res_relay, 0xFF, 1024));
    {
      const size_t n_gran = 2 * static_cast<size_t>(block_size) * (sizeof(F) == 8 ? 2 : 1) + 8;
      KNH_HIP(hipHostMalloc(&h_res_out, n_gran * sizeof(uint64_t), hipHostMallocMapped | hipHostMallocCoherent));
      std::memset(h_res_out, 0xFF, n_gran * sizeof(uint64_t));  // (a tag no call carries)
    }
    KNH_HIP(hipHostMalloc(&h_res_done, 256, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(h_res_done, 0, 256);
    h_res_done[0] = 0xFFFFFFFFu; h_res_done[1] = 0u; h_res_done[2] = 0u; h_res_done[4] = 0xFFFFFFFFu; h_res_done[5] = 0xFFFFFFFFu;
    KNH_HIP(hipDeviceSynchronize());
    if (const char* e = std::getenv("KNH_RESIDENT_IDLE_US")) { const long us = std::atol(e); if (us >= 50 && us <= 2000000) res_idle_ticks = static_cast<uint64_t>(us) * 100u; }
    return KNH_OK;
  }
  void fill_launch_args(VoiceKernelArgs<F>& a, uint32_t n_blocks, uint32_t fb, uint32_t fe) {
    a.state = d_state;
    a.stride = stride;
    a.n_voices = nv;
    a.env_ranks = env_ranks;
    a.block_size = static_cast<uint32_t>(block_size);
    a.n_blocks = n_blocks;
    a.frame_begin = fb;
    a.frame_end = fe;
    a.sine_table = d_sine;
    a.f2pi = f2pi;
    a.sample_rate = sample_rate;
    a.seg_table = d_seg_table;
    a.seg_max = seg_max;
    a.delay_ring = d_delay;
    a.delay_stride = delay_stride;
    a.buffer = d_buffer;
    a.buffer_frames = static_cast<uint32_t>(h_buffer.size());
    a.input = nullptr;
    a.in_channels = desc.in_channels;
    a.ev_start = nullptr;
    a.events = nullptr;
    a.partials = d_partials;
    a.voices_out = nullptr;
    a.done_frames = d_done;
    a.flags = d_flags;
    a.res = knh_dev::Resident{};
  }
  // the caller holds resident_slot().mu; the command word already carries `first_epoch`'s command or will
  static bool res_debug() { static const bool on = std::getenv("KNH_DEBUG_RES") != nullptr; return on; }
  int res_launch(uint32_t first_epoch) {
    if (res_debug()) std::fprintf(stderr, "[knh resident] launch, first epoch %u, %u voices, tile %u frames\n", first_epoch, nv, res_tile_frames());
    VoiceKernelArgs<F> a;
    fill_launch_args(a, 1, 0, static_cast<uint32_t>(block_size));
    a.flags = d_flags + 32;  // (a third set: the two the ordinary launches alternate stay as their fold kernels left them)
    a.res.bell = reinterpret_cast<const knh_dev::u64*>(res_bell_dev);
    a.res.relay = reinterpret_cast<knh_dev::u64*>(d_res_relay);
    a.res.rows = reinterpret_cast<knh_dev::u64*>(d_res_rows);
    a.res.wg_flags = reinterpret_cast<knh_dev::u64*>(d_res_wg_flags);
    for (int k = 0; k < 2; ++k) { a.res.ev_start[k] = h_ev_start2[k]; a.res.events[k] = h_events2[k]; }
    a.res.idle_ticks = res_idle_ticks;
    a.res.host_started = h_res_done + 4;
    a.res.first_epoch = first_epoch;
    a.res.max_tiles = res_max_tiles;
    // (0: only workgroup 0 reads the host's word, also when it lives in device memory.  With every workgroup reading it, a
    // command that arrives just as workgroup 0 gives up waiting would be taken by some workgroups and not by it: the relay makes
    // workgroup 0 the one place where "this command" or "leave" is decided.  Costs 0.5 us per call.)
    a.res.bell_is_device = 0u;
    {  // the fold server first: a handful of wavefronts that will sit beside the voice kernel's workgroups
      knh_dev::ResServerArgs<F> sa{};
      sa.relay = reinterpret_cast<const knh_dev::u64*>(d_res_relay);
      sa.bell = nullptr;  // (as for the voice kernel: the relay decides)
      sa.rows = reinterpret_cast<const knh_dev::u64*>(d_res_rows);
      sa.wg_flags = reinterpret_cast<const knh_dev::u64*>(d_res_wg_flags);
      sa.group_rows = reinterpret_cast<knh_dev::u64*>(d_res_group_rows);
      sa.group_flags = reinterpret_cast<knh_dev::u64*>(d_res_group_flags);
      sa.host_out = reinterpret_cast<knh_dev::u64*>(h_res_out);
      sa.host_done = h_res_done;
      sa.idle_ticks = res_idle_ticks;
      sa.first_epoch = first_epoch;
      sa.n_rows = (nv + 63) / 64;
      sa.planes = pan ? 2u : 1u;
      sa.out_channels = desc.out_channels;
      sa.block_size = static_cast<uint32_t>(block_size);
      sa.tile_frames = res_tile_frames();
      KNH_HIP(launch_res_server(sa, res_stream));
    }
    KNH_HIP(launch_voice(a, (nv + 63) / 64, own_stream));
    res_on = true;
    res_launches += 1;
    ResidentSlot& rs = resident_slot();
    if (device >= 0 && device < 64) { rs.owner[device] = this; rs.leave[device] = &Bank<F>::res_leave_thunk; }
    // Both kernels are running?  (If the two streams share a hardware queue, the second kernel waits for the first to END --
    // which, for kernels that wait for each other's work, is never in time.  Then this bank keeps to a launch per call.)
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spin = 0;; ++spin) {
      if (__atomic_load_n(&h_res_done[4], __ATOMIC_ACQUIRE) == first_epoch && __atomic_load_n(&h_res_done[5], __ATOMIC_ACQUIRE) == first_epoch) break;
      if ((spin & 0xFFu) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.05) {
        if (res_debug()) std::fprintf(stderr, "[knh resident] handshake failed: voice %u server %u (want %u)\n", h_res_done[4], h_res_done[5], first_epoch);
        res_policy = 0;
        warn("the resident voice kernel and its fold server did not start side by side (a shared hardware queue?): this bank launches per call");
        int rc = res_leave_locked(false);
        return rc != KNH_OK ? rc : KNH_ERR_UNSUPPORTED_CHAIN;  // (res_call: fall back to an ordinary launch)
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    return KNH_OK;
  }
  // One block through the resident kernel: frames [fb, fe) of the block into out_host ([channels][block_size], written at
  // their place).  The event list of the call (if any) is the pinned list upload_events has just made.
  int res_call(uint32_t fb, uint32_t fe, bool have_events, void* out_host, uint32_t* out_flags) {
    ResidentSlot& rs = resident_slot();
    std::lock_guard<std::mutex> lock(rs.mu);
    KNH_HIP(hipSetDevice(device));
    if (device >= 0 && device < 64 && rs.owner[device] && rs.owner[device] != this) rs.leave[device](rs.owner[device]);
    const uint64_t payload = (static_cast<uint64_t>(fb) << 24) | (static_cast<uint64_t>(fe) << 40) | (have_events ? 1ull << 56 : 0ull) |
                             (have_events && list_in_use == 1 ? 1ull << 57 : 0ull) | (have_events ? static_cast<uint64_t>(res_n_ranges & 15u) << 59 : 0ull);
    if (have_events && list_in_use >= 0) { list_busy[list_in_use] = false; list_in_use = -1; }  // (no stream order to keep: the call is over when this returns)
    uint32_t epoch = 0;
    auto ring = [&]() -> int {  // the next epoch's command; a kernel to take it if there is none
      res_epoch = (res_epoch + 1u) & 0xFFFFFFu;
      epoch = res_epoch;
      if (!res_on) { int rc = res_launch(epoch); if (rc != KNH_OK) return rc; }  // (KNH_ERR_UNSUPPORTED_CHAIN: no resident launch for this bank after all)
#if defined(__x86_64__)
      __builtin_ia32_sfence();  // the event list is in memory before the word that announces it
#endif
      __atomic_store_n(res_bell, static_cast<uint64_t>(epoch) | payload, __ATOMIC_RELEASE);
#if defined(__x86_64__)
      if (res_bell_is_device) __builtin_ia32_sfence();
#endif
      return KNH_OK;
    };
    { int rc = ring(); if (rc != KNH_OK) return rc; }
    res_calls += 1;
    const auto t0 = std::chrono::steady_clock::now();
    // Waits for something the device stores (`ready`): 0 = there; kRestart = the kernel had ended by itself and the command has
    // gone out again under a new epoch (whatever was read so far belongs to no call: start over); anything else = an error.
    constexpr int kRestart = -12345;
    auto wait_for = [&](auto&& ready) -> int {
      for (uint64_t spin = 1;; ++spin) {
        if (ready()) return 0;
        if ((spin & 0x3FFFu) == 0) {
          const hipError_t q = hipStreamQuery(own_stream);
          if (q == hipSuccess) {
            // The kernel has ended by itself (the host was away for longer than its patience) just as this command was written.
            // Its workgroup 0 left "leave" in the relay under THIS epoch, so the command goes out again under the next one, to a
            // new launch.
            if (ready()) return 0;
            if (res_debug()) std::fprintf(stderr, "[knh resident] the kernel ended without answering epoch %u (done word %u, %.3f ms into the call, server stream %s; root wavefronts at %x %x %x %x)\n", epoch, h_res_done[0],
                                          std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), hipStreamQuery(res_stream) == hipSuccess ? "idle" : "busy",
                                          h_res_done[20], h_res_done[21], h_res_done[22], h_res_done[23]);
            // (its fold server has then heard "leave" over the relay too.  A server that is still busy was in the middle of a
            // call: the voice kernel took the command, and taking it again would render the block twice.)
            hipError_t qs = hipStreamQuery(res_stream);
            for (int k = 0; k < 200 && qs == hipErrorNotReady; ++k) { std::this_thread::sleep_for(std::chrono::microseconds(500)); qs = hipStreamQuery(res_stream); }
            res_on = false;
            if (qs != hipSuccess) {
              res_policy = 0;
              (void)res_leave_locked(false);
              return fail(KNH_ERR_DEVICE, "the resident voice kernel ended in the middle of a call (its mix never arrived)");
            }
            int rc = ring();
            if (rc != KNH_OK) return rc;
            return kRestart;
          } else if (q != hipErrorNotReady) {
            res_on = false;
            return fail(KNH_ERR_DEVICE, std::string("hipStreamQuery: ") + hipGetErrorString(q));
          }
          if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 10.0) {
            return fail(KNH_ERR_DEVICE, "the resident kernel did not answer within 10 s");
          }
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
    };
    // The block arrives tile by tile as granules {sample bits, tag = epoch << 8 | tile}: every frame is taken the moment its
    // tag is there (the early tiles while the kernel is still at the later ones), a mono mix copied to every channel; the
    // call's done / running counts come as one more granule behind the last tile.
    constexpr size_t W = sizeof(F) == 8 ? 2 : 1;
    const uint32_t n = fe - fb, tf = res_tile_frames(), planes = fold_planes;
    F* const out = static_cast<F*>(out_host);
    uint64_t flag_word = 0;
    for (bool again = true; again;) {
      again = false;
      for (uint32_t rel = 0; rel < n && !again; ++rel) {
        const uint32_t tag = (epoch << 8) | ((rel / tf) & 0xFFu);
        for (uint32_t p = 0; p < planes && !again; ++p) {
          volatile uint64_t* g = h_res_out + (static_cast<size_t>(p) * block_size + fb + rel) * W;
          // (the device's stores took these lines out of the CPU's caches: a tile that has landed is eight cache misses in a row
          // unless they are asked for together -- 0.7 us at the end of every call)
          if ((reinterpret_cast<uintptr_t>(g) & 63u) == 0) {
            __builtin_prefetch(const_cast<const uint64_t*>(g) + 8, 0, 3);
            __builtin_prefetch(const_cast<const uint64_t*>(g) + 16, 0, 3);
            __builtin_prefetch(const_cast<const uint64_t*>(g) + 24, 0, 3);
            __builtin_prefetch(const_cast<const uint64_t*>(g) + 32, 0, 3);
          }
          uint64_t w0 = 0, w1 = 0;
          auto ready = [&]() -> bool {
            w0 = __atomic_load_n(g, __ATOMIC_RELAXED);
            if (static_cast<uint32_t>(w0 >> 32) != tag) return false;
            if (W == 2) { w1 = __atomic_load_n(g + 1, __ATOMIC_RELAXED); if (static_cast<uint32_t>(w1 >> 32) != tag) return false; }
            return true;
          };
          if (!ready()) {
            const int rc = wait_for(ready);
            if (rc == kRestart) { again = true; break; }
            if (rc != 0) return rc;
          }
          F v;
          if (W == 1) { const uint32_t bits = static_cast<uint32_t>(w0); std::memcpy(&v, &bits, sizeof(F) < 4 ? sizeof(F) : 4); }
          else { const uint64_t bits = (w0 & 0xFFFFFFFFull) | (w1 << 32); std::memcpy(&v, &bits, sizeof(F)); }
          if (planes == 2) out[static_cast<size_t>(p) * block_size + fb + rel] = v;
          else for (uint32_t c = 0; c < desc.out_channels; ++c) out[static_cast<size_t>(c) * block_size + fb + rel] = v;
        }
      }
      if (again) continue;
      volatile uint64_t* gf = h_res_out + static_cast<size_t>(planes) * block_size * W;
      const uint32_t ftag = (epoch << 8) | 255u;
      auto fready = [&]() -> bool { flag_word = __atomic_load_n(gf, __ATOMIC_RELAXED); return static_cast<uint32_t>(flag_word >> 32) == ftag; };
      if (!fready()) {
        const int rc = wait_for(fready);
        if (rc == kRestart) { again = true; continue; }
        if (rc != 0) return rc;
      }
    }
    if (out_flags) {
      uint32_t fl = 0;
      const uint32_t n_done = static_cast<uint32_t>(flag_word) & 0xFFFFu, n_run = (static_cast<uint32_t>(flag_word) >> 16) & 0xFFFFu;
      if (n_done) fl |= KNH_FLAG_ANY_DONE;
      bool has_env = false;
      for (const StageInfo& st : stages) has_env = has_env || st.kind == KNH_STAGE_MUL_ENV_ASR || st.kind == KNH_STAGE_MUL_ENV_AR || st.kind == KNH_STAGE_MUL_ENVELOPE ||
                          st.kind == KNH_STAGE_BUFFER_READER;
      if (has_env && n_run == 0) fl |= KNH_FLAG_ALL_DONE;
      *out_flags = fl;
    }
    return KNH_OK;
  }
  // timing
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> timing_pool;
  size_t timing_used = 0;
  double timing_ms = 0.0;
  uint64_t timing_launches = 0;

  ~Bank() override {
    if (device >= 0) (void)hipSetDevice(device);
    (void)res_leave();
    {
      void* rdev[] = {res_bell_is_device ? static_cast<void*>(res_bell) : nullptr, d_res_relay, d_res_rows, d_res_wg_flags, d_res_group_rows, d_res_group_flags, d_res_arrivals, d_res_group_arrivals};
      if (res_stream) { (void)hipStreamSynchronize(res_stream); (void)hipStreamDestroy(res_stream); }
      void* rhost[] = {res_bell_is_device ? nullptr : static_cast<void*>(res_bell), h_res_out, h_res_done};
      if (own_stream) (void)hipStreamSynchronize(own_stream);
      for (void* p : rdev) if (p) (void)hipFree(p);
      for (void* p : rhost) if (p) (void)hipHostFree(p);
    }
    // everything that may still read or write this bank's memory has finished before any of it is freed: the bank's own
    // stream, the resolver's (its kernels read the pinned record and event lists), and the stream the last launch was given
    if (own_stream) (void)hipStreamSynchronize(own_stream);
    if (ev_stream) (void)hipStreamSynchronize(ev_stream);
    if (flags_stream_set && flags_stream != own_stream) (void)hipStreamSynchronize(flags_stream);
    void* dev_ptrs[] = {d_state, d_sine, d_seg_table, d_delay, d_buffer, d_partials, d_out, d_voices, d_done, d_flags, d_input, d_prog, d_sin_slots, d_fold_count};
    for (void* p : dev_ptrs)
      if (p) (void)hipFree(p);
    void* host_ptrs[] = {h_ev_start2[0], h_ev_start2[1], h_events2[0], h_events2[1], h_out, h_input, h_done, h_ev_overflow};
    for (void* p : host_ptrs)
      if (p) (void)hipHostFree(p);
    for (hipEvent_t e : list_done)
      if (e) (void)hipEventDestroy(e);
    if (in_copied) (void)hipEventDestroy(in_copied);
    for (hipEvent_t e : recs_done)
      if (e) (void)hipEventDestroy(e);
    if (ev_stream) (void)hipStreamSynchronize(ev_stream);
    void* ev_dev[] = {d_stages, d_armed, d_ev_cnt, d_rec_start, d_out_start2[0], d_out_start2[1], d_keys, d_recs, d_out_events2[0], d_out_events2[1]};
    for (void* p : ev_dev)
      if (p) (void)hipFree(p);
    for (hipEvent_t e : lists_free)
      if (e) (void)hipEventDestroy(e);
    if (ev_stream) (void)hipStreamDestroy(ev_stream);
    for (QRec* p : h_recs2)
      if (p) (void)hipHostFree(p);
    for (auto& p : timing_pool) {
      (void)hipEventDestroy(p.first);
      (void)hipEventDestroy(p.second);
    }
    if (own_stream) (void)hipStreamDestroy(own_stream);
  }

  int set_ctor(uint32_t stage, uint32_t first, uint32_t count, const double* args, uint32_t n_args) override {
    if (initialised) return fail(KNH_ERR_INVALID_ARGUMENT, "constructor arguments must be set before init");
    if (stage >= stages.size()) return fail(KNH_ERR_OUT_OF_RANGE, "stage out of range");
    if (static_cast<uint64_t>(first) + count > nv) return fail(KNH_ERR_OUT_OF_RANGE, "voice range out of range");
    if (stages[stage].kind == KNH_STAGE_MUL_ENVELOPE && stages[stage].n_ctor < 0) {
      // Envelope::new(start, segments): the first call fixes the bank-wide segment capacity
      if (n_args < 6 || (n_args - 4) % 2 != 0) return fail(KNH_ERR_INVALID_ARGUMENT, "Envelope takes 4 + 2 * n_max constructor arguments");
      stages[stage].n_ctor = static_cast<int>(n_args);
      ctor[stage].assign(static_cast<size_t>(nv) * n_args, 0.0);
    }
    if (static_cast<int>(n_args) != stages[stage].n_ctor) return fail(KNH_ERR_INVALID_ARGUMENT, "wrong number of constructor arguments");
    if (n_args && !args) return fail(KNH_ERR_INVALID_ARGUMENT, "null args");
    std::copy(args, args + static_cast<size_t>(count) * n_args, ctor[stage].begin() + static_cast<size_t>(first) * n_args);
    return KNH_OK;
  }

  int set_buffer(uint32_t stage, const void* samples, size_t n_frames, double sr) override {
    if (initialised) return fail(KNH_ERR_INVALID_ARGUMENT, "knh_bank_set_buffer comes before knh_bank_init");
    if (stage >= stages.size() || stages[stage].kind != KNH_STAGE_BUFFER_READER) return fail(KNH_ERR_INVALID_ARGUMENT, "not a BufferReader stage");
    if (!samples || n_frames == 0 || n_frames >= (1ull << 31) || !(sr > 0.0)) return fail(KNH_ERR_INVALID_ARGUMENT, "empty buffer or bad sample rate");
    h_buffer.assign(static_cast<const F*>(samples), static_cast<const F*>(samples) + n_frames);
    buffer_sr = sr;
    return KNH_OK;
  }

  // the bank node's input channels for the next launch
  F* d_input = nullptr;           // [in_blocks_cap][in_channels][block_size]
  F* h_input = nullptr;           // pinned staging of the same
  uint32_t in_blocks_cap = 0, in_blocks_set = 0;
  const void* in_device = nullptr;  // set_input_device: read where it is
  bool in_host_pending = false;
  hipEvent_t in_copied = nullptr;   // recorded behind the upload of h_input, on the stream of that launch
  bool in_copy_pending = false;
  bool uses_input = false;
  int set_input(uint32_t n_blocks, const void* host, const void* dev) override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    if (desc.in_channels == 0) return fail(KNH_ERR_INVALID_ARGUMENT, "the bank has no input channels (knh_bank_desc.in_channels)");
    if (n_blocks == 0 || n_blocks > 4096 || (!host && !dev)) return fail(KNH_ERR_INVALID_ARGUMENT, "knh_bank_set_input: n_blocks in 1..4096 and a buffer");
    KNH_HIP(hipSetDevice(device));
    in_blocks_set = n_blocks;
    in_device = dev;
    in_host_pending = false;
    if (dev) return KNH_OK;
    const size_t elems = static_cast<size_t>(n_blocks) * desc.in_channels * block_size;
    if (n_blocks > in_blocks_cap) {
      KNH_HIP(hipDeviceSynchronize());
      if (d_input) KNH_HIP(hipFree(d_input));
      if (h_input) KNH_HIP(hipHostFree(h_input));
      d_input = nullptr; h_input = nullptr;
      KNH_HIP(hipMalloc(&d_input, elems * sizeof(F)));
      KNH_HIP(hipHostMalloc(&h_input, elems * sizeof(F)));
      in_blocks_cap = n_blocks;
    } else if (in_copy_pending) {
      // the upload of the launch before may still be reading the staging buffer -- on whatever stream that launch was
      // given (the caller's, the pipelined host output's, a rank bank's), hence an event and not a stream to wait for
      KNH_HIP(hipEventSynchronize(in_copied));
      in_copy_pending = false;
    }
    std::memcpy(h_input, host, elems * sizeof(F));
    in_host_pending = true;
    return KNH_OK;
  }

  // ---- UGen::init for every node of every voice --------------------------------------
  int init(uint32_t sr, size_t bs) override {
    if (initialised) return fail(KNH_ERR_INVALID_ARGUMENT, "already initialised");
    if (sr == 0 || bs == 0 || bs > 65535) return fail(KNH_ERR_INVALID_ARGUMENT, "sample_rate/block_size out of range (block_size <= 65535)");
    int ndev = knh_device_count();
    if (ndev <= 0) return fail(KNH_ERR_NO_DEVICE, "no gfx950 device visible; this engine has no CPU path");
    if (desc.device >= 0) device = desc.device;
    else KNH_HIP(hipGetDevice(&device));
    KNH_HIP(hipSetDevice(device));
    // Chains without a pre-built pipelined kernel are fused now (hiprtc).  Up to two voice groups per CU the
    // pipelined form wins by a wide margin, so the chain is cut into at most three stage groups of similar cost
    // (estimated instructions per sample) and instantiated as voice_pipe_kernel; KNH_JIT_PIPE=0 keeps the
    // single-wave form.
    // The 64-sample-tile pipeline only where the block is made of whole tiles: a partial tile runs sample by sample,
    // and a 32- or 96-frame block would be half partial tiles (its 32-sample form has none).
    if (pipe && !pipe_pair && pipe->form != 0 && bs % (sizeof(F) == 4 ? 64u : 32u) != 0) pipe = knh::find_pipe(signature.c_str(), 1u);
    const unsigned n_groups = (nv + 63u) / 64u;
    {  // Voices made of SinWt oscillators and arithmetic alone (graphs, or chains without a pre-built kernel) are not run a
       // lane per voice at all: every stage is a pure function of the frame index, so a lane per FRAME it is (voice_frame.hpp)
       // -- measured 10-27 times the lane-per-voice form from one voice to 65 536 (tools/bench_fm_cascade.py), and the only
       // form that takes the reference's 1 531-stage cascade.  KNH_INTERP=0: never (the lane-per-voice form, A/B runs).
      const char* ie = std::getenv("KNH_INTERP");
      bool can = !entry && bs <= 1024 && stages.size() <= 4096;
      for (const StageInfo& S : stages) can = can && S.flags == 0 && S.dcpb == 0 && S.ar_param == 0 && std::strchr("Wmasdvq*+-/", kKinds[S.kind].sig) != nullptr;
      if (can && !(ie && ie[0] == '0')) {
        h_prog.clear();
        size_t si = 0;
        const char* p = signature.c_str();
        while (*p && *p != '#') {
          knh_dev::InterpOp op{};
          const char c = *p++;
          int v[3] = {-1, -1, -1};
          if (*p == '@') {
            ++p;
            for (int k = 0; k < 3; ++k) {
              if (*p == '_') { ++p; } else { v[k] = 0; while (*p >= '0' && *p <= '9') v[k] = v[k] * 10 + (*p++ - '0'); }
              if (*p == ',') ++p;
            }
          }
          switch (c) {
            case 'W': op.kind = knh_dev::INTERP_SIN_WT; break;
            case 'm': op.kind = knh_dev::INTERP_VAL_MUL; break;
            case 'a': op.kind = knh_dev::INTERP_VAL_ADD; break;
            case 's': op.kind = knh_dev::INTERP_VAL_SUB; break;
            case 'd': op.kind = knh_dev::INTERP_VAL_DIV; break;
            case 'v': op.kind = knh_dev::INTERP_VAL_VSUB; break;
            case 'q': op.kind = knh_dev::INTERP_VAL_VDIV; break;
            case '*': op.kind = knh_dev::INTERP_MATH_MUL; break;
            case '+': op.kind = knh_dev::INTERP_MATH_ADD; break;
            case '-': op.kind = knh_dev::INTERP_MATH_SUB; break;
            default: op.kind = knh_dev::INTERP_MATH_DIV; break;
          }
          if (!signature_is_dag(signature)) {  // a plain chain: one signal, every stage works on it in place
            v[0] = c == 'W' ? -1 : 0;
            v[2] = 0;
          }
          if (si >= stages.size() || v[2] < 0 || (c != 'W' && v[0] < 0)) return fail(KNH_ERR_UNSUPPORTED_CHAIN, "interpreter: malformed graph signature");
          op.a = static_cast<unsigned short>(v[0] < 0 ? 0 : v[0]);
          op.b = static_cast<unsigned short>(v[1] < 0 ? 0 : v[1]);
          op.o = static_cast<unsigned short>(v[2]);
          op.slot = static_cast<uint32_t>(stages[si].slot_base);
          h_prog.push_back(op);
          ++si;
        }
        interp_sigs = *p == '#' ? static_cast<unsigned>(std::atoi(p + 1)) : (signature_is_dag(signature) ? 0u : 1u);
        interp_out = h_prog.empty() ? 0u : h_prog.back().o;
        if (si != stages.size() || interp_sigs == 0) return fail(KNH_ERR_UNSUPPORTED_CHAIN, "interpreter: malformed graph signature");
        if (knh::interp_lds_bytes(static_cast<unsigned>(h_prog.size()), static_cast<unsigned>(n_slots), interp_sigs, static_cast<unsigned>(bs), sizeof(F) == 8) <= 158u * 1024u)
          interp = true;
        const char* fj = std::getenv("KNH_FRAME_JIT");
        if (interp && !(fj && fj[0] == '0')) {
          const unsigned tpv = ((static_cast<unsigned>(bs) + 63u) / 64u) * 64u;
          const size_t nwp = (static_cast<size_t>(n_slots) + 3u) & ~size_t(3);
          // voices per workgroup: as many as keep every CU busy, fit 1 024 threads and (beside the 64 KiB table) the LDS
          unsigned vpw = std::max(1u, nv / 256u);
          vpw = std::min(vpw, 1024u / tpv);
          while (vpw > 1u && 65536u + vpw * nwp * sizeof(W) > 156u * 1024u) --vpw;
          if (65536u + vpw * nwp * sizeof(W) <= 156u * 1024u) {
            std::vector<knh::FrameOp> fops(h_prog.size());
            for (size_t k = 0; k < h_prog.size(); ++k) fops[k] = knh::FrameOp{h_prog[k].kind, h_prog[k].a, h_prog[k].b, h_prog[k].o, h_prog[k].slot};
            std::string why;
            frame_jit = knh::jit_frame_kernel(fops.data(), static_cast<unsigned>(fops.size()), interp_sigs, interp_out, static_cast<unsigned>(n_slots), vpw, tpv,
                                              sizeof(F) == 8, &why);
            if (frame_jit) frame_vpw = vpw;
            else warnings.push_back("frame-parallel kernel not built (" + why.substr(0, 300) + "): the interpreter runs the voice");
          }
        }
      }
    }
    env_ranks = 0;
    if (signature_is_dag(signature)) {
      // Which envelope's mark_done names the voice's done frame when several finish in one block: the last one in the
      // reference's TASK order (graph_gen.rs:196-200), which for a graph is the order Graph::calculate_node_order sorts the
      // nodes into (graph.rs:1938-2067): depth first from the output, a node's inputs in channel order, each node after
      // everything it reads; nodes the output does not depend on come last, in the order they were pushed.
      const int n = static_cast<int>(stages.size());
      auto is_src = [&](int i) { return std::strchr("WNPUKOGBFI", kKinds[stages[i].kind].sig) != nullptr && !(stages[i].flags & KNH_STAGE_FLAG_AR_FREQ); };
      auto node_output = [&](int k) { while (k + 1 < n && is_wrapper_kind(stages[k + 1].kind)) ++k; return k; };
      std::vector<int> a(n, -1), b(n, -1);
      for (int i = 0; i < n; ++i) {
        if (is_math2_kind(stages[i].kind)) { a[i] = node_output(stages[i].input - 1); b[i] = node_output(stages[i].input2 - 1); }
        else if (i > 0 && !is_src(i)) a[i] = stages[i].input ? node_output(stages[i].input - 1) : i - 1;
        // an audio-rate parameter edge: followed after the node's input edges (graph.rs:1938-1980)
        if (stages[i].ar_param && !is_math2_kind(stages[i].kind)) b[i] = node_output(stages[i].input2 - 1);
      }
      std::vector<int> order, state(n, 0), stack{n - 1};
      while (!stack.empty()) {  // post-order, first operand first
        const int k = stack.back();
        if (state[k] == 0) { state[k] = 1; if (a[k] >= 0 && state[a[k]] == 0) { stack.push_back(a[k]); continue; } }
        if (state[k] == 1) { state[k] = 2; if (b[k] >= 0 && state[b[k]] == 0) { stack.push_back(b[k]); continue; } }
        if (state[k] == 2) { state[k] = 3; order.push_back(k); }
        stack.pop_back();
      }
      for (int i = 0; i < n; ++i) if (state[i] == 0) order.push_back(i);
      std::vector<int> rank(n, 0);
      for (size_t r = 0; r < order.size(); ++r) rank[order[r]] = static_cast<int>(r);
      std::vector<int> envs;
      for (int i = 0; i < n; ++i)
        if (stages[i].kind == KNH_STAGE_MUL_ENV_ASR || stages[i].kind == KNH_STAGE_MUL_ENV_AR || stages[i].kind == KNH_STAGE_MUL_ENVELOPE) envs.push_back(i);
      bool in_list_order = true;
      for (size_t j = 1; j < envs.size(); ++j) in_list_order = in_list_order && rank[envs[j - 1]] < rank[envs[j]];
      if (!in_list_order && envs.size() <= 15) {
        std::vector<int> by_rank(envs);
        std::sort(by_rank.begin(), by_rank.end(), [&](int x, int y) { return rank[x] < rank[y]; });
        for (size_t j = 0; j < envs.size(); ++j) {
          const uint64_t place = 1 + static_cast<uint64_t>(std::find(by_rank.begin(), by_rank.end(), envs[j]) - by_rank.begin());
          env_ranks |= place << (4 * j);
        }
      }
    }
    const char* jp = std::getenv("KNH_JIT_PIPE");
    // (a single voice group with a pre-built kernel stays on it: nothing to gain, and no compile at init)
    // (a voice that is a graph, not a chain, runs in the single-wave form: the pipeline's edges carry one signal)
    // (the pipeline covers two rounds of 256 voice groups in f32; one in f64 and for chains with a delay, whose rings want as
    // many wavefronts as there are to keep requests in flight: the thresholds of the pre-built chains, bank.hip make_bank)
    // (a chain with a pre-built one-wavefront kernel but no pre-built wide form keeps the fused pipeline up to 512 groups)
    const unsigned jit_pipe_max = !entry && (sizeof(F) == 8 || signature.find_first_of("DYZ") != std::string::npos) ? 256u : 512u;
    const bool pipe_jit = !interp && !pipe && !dag && wide_waves == 0 && pipeline_level >= 1 && n_groups <= jit_pipe_max && !(jp && jp[0] == '0') &&
                          !(entry && n_groups == 1) && !signature_is_dag(signature);
    if (pipe_jit) {
      std::string why;
      unsigned cuts[2];
      const unsigned n_cuts = partition_chain(signature, cuts);
      jit = knh::jit_pipe_kernel(signature.c_str(), cuts, n_cuts, sizeof(F) == 8, desc.allow_fma != 0, &why);
      jit_pipe = jit != nullptr;
      if (!jit) return fail(why.rfind("JIT_CRASH: ", 0) == 0 ? KNH_ERR_INTERNAL : KNH_ERR_UNSUPPORTED_CHAIN, "run-time fusion of chain '" + signature + "' (pipelined) failed: " + why);
    } else if (!entry && !interp) {  // no pre-built kernel at all
      std::string why;
      // Beyond the pipeline's reach (more than 512 voice groups) a fused chain runs as whole-chain wavefronts sharing a staged
      // sine table, four to a workgroup (one per SIMD) up to 1 024 groups, eight beyond -- the forms of the pre-built
      // chains (make_bank); round 3 gave every fused voice group a workgroup, and a 64 KiB table staging, of its own.
      // A voice that is a graph keeps the one-wavefront form (its signals' registers leave no room to share a SIMD).
      // KNH_JIT_WAVES=1|4|8|16 overrides (tests: the filter's steps at one, two and four wavefronts per SIMD).
      unsigned jw = !signature_is_dag(signature) && n_groups > jit_pipe_max ? (n_groups <= 1024 ? 4u : 8u) : 1u;
      if (const char* e = std::getenv("KNH_JIT_WAVES")) { const int v = std::atoi(e); if ((v == 1 || v == 4 || v == 8 || v == 16) && !signature_is_dag(signature)) jw = static_cast<unsigned>(v); }
      if (sizeof(F) == 8 && jw == 16) jw = 8;  // (sixteen f64 tiles do not fit beside the table)
      jit = knh::jit_voice_kernel(signature.c_str(), sizeof(F) == 8, desc.allow_fma != 0, &why, jw);
      if (!jit) return fail(why.rfind("JIT_CRASH: ", 0) == 0 ? KNH_ERR_INTERNAL : KNH_ERR_UNSUPPORTED_CHAIN, "run-time fusion of chain '" + signature + "' failed: " + why);
    }
    sample_rate = sr;
    block_size = bs;
    stride = (static_cast<long>(nv) + 63) / 64 * 64;
    // osc.rs:144-145
    f2pi = 16384.0 * 65536.0 * (1.0 / static_cast<double>(sr));
    const F sr_as_f32 = static_cast<F>(static_cast<float>(sr));  // F::new(sample_rate as f32)

    std::vector<W> st(static_cast<size_t>(n_slots) * stride, W(0));
    std::vector<double> seg_rows;
    auto slot = [&](int s, uint32_t v) -> W& { return st[static_cast<size_t>(s) * stride + v]; };
    auto fw = [](F x) { return static_cast<W>(to_bits(x)); };
    shadow.assign(stages.size(), Shadow{});
    for (size_t si = 0; si < stages.size(); ++si) {
      const StageInfo& S = stages[si];
      const double* ca = ctor[si].data();
      Shadow& sh = shadow[si];
      for (uint32_t v = 0; v < nv; ++v) {
        const double* a = ca + static_cast<size_t>(v) * S.n_ctor;
        switch (S.kind) {
          case KNH_STAGE_SIN_WT: {  // osc.rs:110-123,142-147
            if (v == 0) sh.a.resize(nv);
            F freq = static_cast<F>(a[0]);
            sh.a[v] = freq;
            slot(S.slot_base + 0, v) = 0;
            slot(S.slot_base + 1, v) = 0;
            slot(S.slot_base + 2, v) = sat_u32(static_cast<double>(freq) * f2pi);
          } break;
          case KNH_STAGE_SIN_NUMERIC: {  // osc.rs:231-236,253-261
            F freq = static_cast<F>(a[0]);
            slot(S.slot_base + 0, v) = fw(F(0));
            slot(S.slot_base + 1, v) = fw(F(0));
            slot(S.slot_base + 2, v) = fw(freq / sr_as_f32);
          } break;
          case KNH_STAGE_SVF: {  // svf.rs:64-79,134-141
            if (v == 0) { sh.a.resize(nv); sh.b.resize(nv); sh.c.resize(nv); sh.ty.resize(nv); }
            double tyd = a[0];
            uint32_t ty = (tyd >= 0 && tyd <= 8) ? static_cast<uint32_t>(tyd) : 0u;
            sh.ty[v] = static_cast<uint8_t>(ty);
            sh.a[v] = static_cast<F>(a[1]); sh.b[v] = static_cast<F>(a[2]); sh.c[v] = static_cast<F>(a[3]);
            F co[6];
            svf_coeffs<F>(ty, sh.a[v], sh.b[v], sh.c[v], sr_as_f32, co);
            slot(S.slot_base + 0, v) = fw(F(0));
            slot(S.slot_base + 1, v) = fw(F(0));
            for (int k = 0; k < 6; ++k) slot(S.slot_base + 2 + k, v) = fw(co[k]);
            if (S.n_slots == 12) {  // a parameter driven at audio rate: the setter runs on the device and needs the other values
              slot(S.slot_base + 8, v) = fw(sh.a[v]); slot(S.slot_base + 9, v) = fw(sh.b[v]); slot(S.slot_base + 10, v) = fw(sh.c[v]);
              slot(S.slot_base + 11, v) = ty;
            }
          } break;
          case KNH_STAGE_ONEPOLE_LPF:    // onepole.rs:118-129
          case KNH_STAGE_ONEPOLE_HPF: {  // onepole.rs:157-167 (b1 = 0 -> exp(0) = 1)
            F freq = S.kind == KNH_STAGE_ONEPOLE_LPF ? static_cast<F>(a[0]) : F(0);
            F f = freq / sr_as_f32;
            F b1 = std::exp(F(-2.0) * Consts<F>::PI * f);
            F a0 = F(1.0) - b1;
            slot(S.slot_base + 0, v) = fw(F(0));
            slot(S.slot_base + 1, v) = fw(a0);
            slot(S.slot_base + 2, v) = fw(b1);
          } break;
          case KNH_STAGE_MUL_ENV_ASR:
          case KNH_STAGE_MUL_ENV_AR: {  // envelopes.rs:33-43,135-151 / :187-197,268-284
            if (v == 0) { sh.a.resize(nv); sh.b.resize(nv); }
            F atk = static_cast<F>(a[0]), rel = static_cast<F>(a[1]);
            sh.a[v] = atk; sh.b[v] = rel;
            F ar = atk == F(0) ? F(1) : F(1) / (atk * static_cast<F>(sr));
            F rr = rel == F(0) ? F(1) : F(1) / (rel * static_cast<F>(sr));
            slot(S.slot_base + 0, v) = 0;          // Stopped
            slot(S.slot_base + 1, v) = fw(F(0));   // t
            slot(S.slot_base + 2, v) = fw(ar);
            slot(S.slot_base + 3, v) = fw(rr);
            slot(S.slot_base + 4, v) = fw(F(1));   // release_scale
          } break;
          case KNH_STAGE_MUL_ENVELOPE: {  // envelopes.rs:373-400 (+ builder methods), init :404-406
            if (S.n_ctor < 0) return fail(KNH_ERR_INVALID_ARGUMENT, "Envelope stage without constructor arguments");
            const uint32_t n_max = static_cast<uint32_t>((S.n_ctor - 4) / 2);
            if (v == 0) { env_start.assign(nv, 0.0); env_nseg.assign(nv, 0); seg_max = n_max; seg_rows.assign(static_cast<size_t>(nv) * n_max * 3, 0.0); }
            uint32_t n_seg = a[3] >= 1 ? static_cast<uint32_t>(a[3]) : 1u;
            if (n_seg > n_max) n_seg = n_max;
            env_start[v] = a[0];
            env_nseg[v] = n_seg;
            const double dt = a[1] * (1.0 / static_cast<double>(sr));  // time_scale * base_scale
            auto put2 = [&](int rel, double d) {
              uint64_t b = to_bits(d);
              slot(S.slot_base + rel, v) = static_cast<W>(static_cast<uint32_t>(b));
              slot(S.slot_base + rel + 1, v) = static_cast<W>(static_cast<uint32_t>(b >> 32));
            };
            slot(S.slot_base + 0, v) = 0;  // Stopped
            slot(S.slot_base + 1, v) = 0;
            put2(2, 0.0);
            put2(4, a[0]);  // from_value = start_value
            put2(6, dt);
            slot(S.slot_base + 8, v) = n_seg;
            slot(S.slot_base + 9, v) = a[2] != 0.0 ? 1u : 0u;
            slot(S.slot_base + 10, v) = v;
            for (uint32_t k = 0; k < n_max; ++k) {
              const double dur = a[4 + 2 * k], val = a[5 + 2 * k];
              double* row = &seg_rows[(static_cast<size_t>(v) * n_max + k) * 3];
              row[0] = dur; row[1] = 1.0 / dur; row[2] = val;  // EnvelopeSegment::new, envelopes.rs:327-333
            }
          } break;
          case KNH_STAGE_BUFFER_READER: {  // buffer.rs:40-57 (new, start_at), :106-115 (init)
            if (h_buffer.empty()) return fail(KNH_ERR_INVALID_ARGUMENT, "BufferReader stage without knh_bank_set_buffer");
            if (S.dcpb > 0) return fail(KNH_ERR_INVALID_ARGUMENT, "BufferReader cannot be wrapped in WrPreciseTiming here");
            if (v == 0) { buf_start.assign(nv, 0.0); buf_dur.assign(nv, 0.0); buf_rate.assign(nv, 0.0); }
            buf_base_rate = buffer_sr / static_cast<double>(sr);  // Buffer::buf_rate_scale
            const double length_seconds = static_cast<double>(h_buffer.size()) / buffer_sr;
            auto secs_to_frames = [&](double secs) {  // Seconds::from_secs_f64(secs).to_samples_f64(buffer_sr), time.rs:59-64,92-96
              const double whole = std::floor(secs);
              const uint32_t tes = sat_u32((secs - std::trunc(secs)) * 282240000.0);
              return static_cast<double>(sat_u32(whole)) * buffer_sr + (static_cast<double>(tes) * buffer_sr) / 282240000.0;
            };
            const double start = secs_to_frames(a[2]), dur = secs_to_frames(length_seconds);
            buf_start[v] = start; buf_dur[v] = dur; buf_rate[v] = a[0];
            auto put2 = [&](int rel, double d) {
              const uint64_t b = to_bits(d);
              slot(S.slot_base + rel, v) = static_cast<W>(static_cast<uint32_t>(b));
              slot(S.slot_base + rel + 1, v) = static_cast<W>(static_cast<uint32_t>(b >> 32));
            };
            put2(0, start);                   // jump_to(start_frame)
            put2(2, buf_base_rate * a[0]);    // base_rate * rate, the per-sample step
            put2(4, start);
            put2(6, start + dur);
            slot(S.slot_base + 8, v) = 0;
            slot(S.slot_base + 9, v) = a[1] != 0.0 ? 1u : 0u;
          } break;
          case KNH_STAGE_PHASOR: {  // osc.rs:181-188 (new), :197-200 (init: step = freq * (1 / sample_rate))
            const double step = a[0] * (1.0 / static_cast<double>(sr));
            const uint64_t sb = to_bits(step);
            slot(S.slot_base + 0, v) = 0;
            slot(S.slot_base + 1, v) = 0;
            slot(S.slot_base + 2, v) = static_cast<W>(static_cast<uint32_t>(sb));
            slot(S.slot_base + 3, v) = static_cast<W>(static_cast<uint32_t>(sb >> 32));
          } break;
          case KNH_STAGE_SAFETY_LIMITER: break;
          case KNH_STAGE_INPUT: {
            uses_input = true;
            if (!(a[0] >= 0.0) || a[0] >= static_cast<double>(desc.in_channels)) return fail(KNH_ERR_INVALID_ARGUMENT, "KNH_STAGE_INPUT: channel is not below knh_bank_desc.in_channels");
            slot(S.slot_base, v) = static_cast<W>(static_cast<uint32_t>(a[0]));
          } break;
          case KNH_STAGE_MATH_ADD: case KNH_STAGE_MATH_SUB: case KNH_STAGE_MATH_MUL: case KNH_STAGE_MATH_DIV: case KNH_STAGE_MATH_POW: break;  // no state
          case KNH_STAGE_WHITE_NOISE: case KNH_STAGE_PINK_NOISE: case KNH_STAGE_BROWN_NOISE: {
            // fastrand::Rng::with_seed(next_randomness_seed()) (noise.rs:34,66,134): the state is the seed
            const uint64_t seed = a[0] >= 0.0 ? static_cast<uint64_t>(a[0]) : 0u;
            slot(S.slot_base + 0, v) = static_cast<W>(static_cast<uint32_t>(seed));
            slot(S.slot_base + 1, v) = static_cast<W>(static_cast<uint32_t>(seed >> 32));
            if (S.kind == KNH_STAGE_BROWN_NOISE) slot(S.slot_base + 2, v) = to_bits(F(0));
            if (S.kind == KNH_STAGE_PINK_NOISE) {  // noise.rs:64-75: counter 1, everything else zero
              slot(S.slot_base + 2, v) = 1u;
              for (int k = 3; k < 14; ++k) slot(S.slot_base + k, v) = to_bits(F(0));
            }
          } break;
          case KNH_STAGE_RANDOM_LIN: {  // noise.rs:172-200: new() draws the first value, init() turns freq into a step and draws the second
            uint64_t rng = (a[0] >= 0.0 ? static_cast<uint64_t>(a[0]) : 0u) * 94u + 53u;
            auto draw = [&rng]() {  // fastrand 2.3.0 Rng::f32 (wyrand), restated: voice_stages.hpp NoiseRng
              rng += 0x2d358dccaa6c78a5ull;
              const unsigned __int128 t = static_cast<unsigned __int128>(rng) * static_cast<unsigned __int128>(rng ^ 0x8bb84b93962eacc9ull);
              const uint32_t r = static_cast<uint32_t>(static_cast<uint64_t>(t) ^ static_cast<uint64_t>(t >> 64));
              const uint32_t bits = 0x3F800000u + (r >> 9);
              float f;
              std::memcpy(&f, &bits, 4);
              return f - 1.0f;
            };
            const F first = static_cast<F>(draw());              // current_value: F::new(rng.f32())
            const F inc = F(1) / static_cast<F>(sr);             // freq_to_phase_inc = F::ONE / F::from(sample_rate)
            const F step = static_cast<F>(a[1]) * inc;           // phase_step *= freq_to_phase_inc
            const F old_target = first + F(0);                   // new_value(): current_value + current_change_width
            const F second = static_cast<F>(draw());
            slot(S.slot_base + 0, v) = static_cast<W>(static_cast<uint32_t>(rng));
            slot(S.slot_base + 1, v) = static_cast<W>(static_cast<uint32_t>(rng >> 32));
            slot(S.slot_base + 2, v) = to_bits(old_target);
            slot(S.slot_base + 3, v) = to_bits(static_cast<F>(second - old_target));
            slot(S.slot_base + 4, v) = to_bits(F(0));
            slot(S.slot_base + 5, v) = to_bits(step);
          } break;
          case KNH_STAGE_POLYBLEP: {  // polyblep.rs:136-153: new(waveform, freq), init -> set_freq: dt = freq / sample_rate
            const F srf = static_cast<F>(sr);  // F::from(sample_rate)
            const F freq = static_cast<F>(a[1]);
            const F dt = freq != F(0) ? freq / srf : F(0);
            const uint64_t wf = a[0] >= 0.0 && a[0] < 14.0 ? static_cast<uint64_t>(a[0]) : 0u;
            slot(S.slot_base + 0, v) = fw(F(0));
            slot(S.slot_base + 1, v) = fw(dt);
            slot(S.slot_base + 2, v) = fw(F(0.5));
            slot(S.slot_base + 3, v) = static_cast<W>(wf);
            slot(S.slot_base + 4, v) = (dt * srf >= srf / F(4)) ? 1u : 0u;  // get_freq_in_hz() >= sample_rate / 4, :210
          } break;
          case KNH_STAGE_ALLPASS_FB_DELAY:  // delay.rs:221-229: an AllpassDelay and feedback = 0
          case KNH_STAGE_ALLPASS_DELAY: {  // delay.rs:107-123: buffer = max_delay_seconds.to_samples(sample_rate) zeros
            if (v == 0) delay_len.assign(nv, 0u);
            const double secs_in = a[0];
            if (!(secs_in >= 0.0) || secs_in >= 4294967296.0) return fail(KNH_ERR_INVALID_ARGUMENT, "AllpassDelay: max delay out of range");
            const uint64_t whole = static_cast<uint64_t>(std::floor(secs_in));
            const uint64_t tes = sat_u32((secs_in - std::floor(secs_in)) * 282240000.0);
            const uint64_t nsamp = whole * sr + tes * static_cast<uint64_t>(sr) / 282240000ull;  // Seconds::to_samples, time.rs:86-90
            if (nsamp == 0) return fail(KNH_ERR_INVALID_ARGUMENT, "AllpassDelay: the ring would be empty (the reference takes a remainder by zero)");
            if (nsamp >= (1ull << 30)) return fail(KNH_ERR_INVALID_ARGUMENT, "AllpassDelay: max delay too long");
            delay_len[v] = static_cast<uint32_t>(nsamp);
            slot(S.slot_base + 0, v) = 0;  // write_frame
            slot(S.slot_base + 1, v) = 0;  // read_frame
            slot(S.slot_base + 2, v) = static_cast<W>(nsamp);
            slot(S.slot_base + 3, v) = v;
            slot(S.slot_base + 4, v) = fw(F(1));  // AllpassInterpolator::new: coeff, prev_input, prev_output all ONE (:61-67)
            slot(S.slot_base + 5, v) = fw(F(1));
            slot(S.slot_base + 6, v) = fw(F(1));
            if (S.kind == KNH_STAGE_ALLPASS_FB_DELAY) slot(S.slot_base + 7, v) = fw(F(0));
          } break;
          case KNH_STAGE_SAMPLE_DELAY: {  // delay.rs:24-31 (new), :45-49 (init)
            if (v == 0) delay_len.assign(nv, 0u);
            // Seconds::from_secs_f64 / to_secs_f64 (knaster_primitives/src/time.rs:59-74), then `as usize`
            const double secs_in = a[0];
            if (!(secs_in >= 0.0) || secs_in >= 4294967296.0) return fail(KNH_ERR_INVALID_ARGUMENT, "SampleDelay: max delay out of range");
            const uint32_t whole = static_cast<uint32_t>(std::floor(secs_in));
            const uint32_t tes = sat_u32((secs_in - std::floor(secs_in)) * 282240000.0);
            const double secs = static_cast<double>(whole) + static_cast<double>(tes) / 282240000.0;
            const double nf = secs * static_cast<double>(sr);
            if (!(nf >= 1.0)) return fail(KNH_ERR_INVALID_ARGUMENT, "SampleDelay: the ring would be empty (the reference divides by zero)");
            if (nf >= 1073741824.0) return fail(KNH_ERR_INVALID_ARGUMENT, "SampleDelay: max delay too long");
            const uint32_t len = static_cast<uint32_t>(nf);
            delay_len[v] = len;
            slot(S.slot_base + 0, v) = 0;    // write_position
            slot(S.slot_base + 1, v) = len;  // len - delay_samples, delay_samples = 0
            slot(S.slot_base + 2, v) = len;
            slot(S.slot_base + 3, v) = v;
          } break;
          case KNH_STAGE_WR_POWI:  // WrPowi::new(ugen, value: i32), wrappers_core/math.rs:591-595
            slot(S.slot_base, v) = static_cast<W>(static_cast<uint32_t>(static_cast<int32_t>(a[0])));
            break;
          case KNH_STAGE_PAN2: {  // Pan2::new(pan: f32), pan.rs:18-23; the gains of process(), :33-35, as F::new(..)
            float gl, gr;
            pan2_gains(static_cast<float>(a[0]), &gl, &gr);
            slot(S.slot_base + 0, v) = fw(static_cast<F>(gl));
            slot(S.slot_base + 1, v) = fw(static_cast<F>(gr));
          } break;
          default:  // Constant / wrapper value: util.rs:43-45, wrappers_core/math.rs:21-23
            slot(S.slot_base, v) = fw(static_cast<F>(a[0]));
            break;
        }
      }
    }
    // device allocations
    KNH_HIP(hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking));
    KNH_HIP(hipMalloc(&d_state, st.size() * sizeof(W)));
    KNH_HIP(hipMemcpy(d_state, st.data(), st.size() * sizeof(W), hipMemcpyHostToDevice));
    {  // NonAaWavetable::sine(), wavetable.rs:130-139: f64 sin, rounded to f32
      std::vector<float> table(16384);
      const double PI = 3.14159265358979323846;
      for (int i = 0; i < 16384; ++i) table[i] = static_cast<float>(std::sin((static_cast<double>(i) / 16384.0) * PI * 2.0));
      KNH_HIP(hipMalloc(&d_sine, 16384 * sizeof(float)));
      KNH_HIP(hipMemcpy(d_sine, table.data(), 16384 * sizeof(float), hipMemcpyHostToDevice));
    }
    if (!h_buffer.empty()) {
      KNH_HIP(hipMalloc(&d_buffer, h_buffer.size() * sizeof(F)));
      KNH_HIP(hipMemcpy(d_buffer, h_buffer.data(), h_buffer.size() * sizeof(F), hipMemcpyHostToDevice));
    }
    if (!delay_len.empty()) {
      uint32_t mx = 0;
      for (uint32_t l : delay_len) mx = std::max(mx, l);
      delay_stride = (mx + 3u) & ~3u;
      // (+ a spare ring behind the last voice's: where lanes without a voice move their lines, voice_stages.hpp RingLines)
      const size_t bytes = (static_cast<size_t>(nv) + 1) * delay_stride * sizeof(F) + 4096;
      size_t free_b = 0, total_b = 0;
      KNH_HIP(hipMemGetInfo(&free_b, &total_b));
      if (bytes > free_b) return fail(KNH_ERR_DEVICE, "SampleDelay: the delay rings do not fit in device memory");
      KNH_HIP(hipMalloc(&d_delay, bytes));
      KNH_HIP(hipMemset(d_delay, 0, bytes));  // vec![F::ZERO; len]
    }
    if (!seg_rows.empty()) {
      KNH_HIP(hipMalloc(&d_seg_table, seg_rows.size() * sizeof(double)));
      KNH_HIP(hipMemcpy(d_seg_table, seg_rows.data(), seg_rows.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    const size_t n_waves = interp ? nv : (nv + 63) / 64;  // rows the fold kernels take: one per wavefront, or (interpreter) one per voice
    if (interp) {
      KNH_HIP(hipMalloc(&d_prog, h_prog.size() * sizeof(knh_dev::InterpOp)));
      KNH_HIP(hipMemcpy(d_prog, h_prog.data(), h_prog.size() * sizeof(knh_dev::InterpOp), hipMemcpyHostToDevice));
      std::vector<uint32_t> sins;
      for (const knh_dev::InterpOp& op : h_prog)
        if (op.kind == knh_dev::INTERP_SIN_WT) sins.push_back(op.slot);
      n_sin = static_cast<unsigned>(sins.size());
      KNH_HIP(hipMalloc(&d_sin_slots, std::max<size_t>(1, sins.size()) * sizeof(uint32_t)));
      if (!sins.empty()) KNH_HIP(hipMemcpy(d_sin_slots, sins.data(), sins.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    pan = !signature.empty() && stages.back().kind == KNH_STAGE_PAN2;  // a Pan2 ends the chain: every voice has a left and a right signal
    fold_planes = pan ? 2u : 1u;
    KNH_HIP(hipMalloc(&d_partials, fold_planes * n_waves * bs * sizeof(F)));
    KNH_HIP(hipMalloc(&d_out, desc.out_channels * bs * sizeof(F)));
    KNH_HIP(hipMemset(d_out, 0, desc.out_channels * bs * sizeof(F)));
    KNH_HIP(hipMalloc(&d_done, static_cast<size_t>(nv) * sizeof(uint32_t)));
    KNH_HIP(hipMemset(d_done, 0xFF, static_cast<size_t>(nv) * sizeof(uint32_t)));
    // two sets of 16 words, used by alternate launches: the fold kernel of a launch clears the other set's counters
    KNH_HIP(hipMalloc(&d_flags, 48 * sizeof(uint32_t)));  // (+ a third set: a resident launch's diagnostics)
    KNH_HIP(hipMemset(d_flags, 0, 48 * sizeof(uint32_t)));
    for (int b = 0; b < 2; ++b) {
      KNH_HIP(hipHostMalloc(&h_ev_start2[b], (static_cast<size_t>(nv) + 2) * sizeof(uint32_t)));
      // room for two events per voice from the start: a list that has to grow later costs a resident kernel its place (it knows
      // the lists by their addresses)
      h_events_cap2[b] = std::max<size_t>(2048, 2 * static_cast<size_t>(nv));
      KNH_HIP(hipHostMalloc(&h_events2[b], h_events_cap2[b] * sizeof(Event)));
      KNH_HIP(hipEventCreateWithFlags(&list_done[b], hipEventDisableTiming));
    }
    KNH_HIP(hipHostMalloc(&h_out, desc.out_channels * bs * sizeof(F) + 2 * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
    KNH_HIP(hipHostMalloc(&h_done, 64, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(h_done, 0, 64);
    KNH_HIP(hipMalloc(&d_fold_count, sizeof(uint32_t)));
    KNH_HIP(hipMemset(d_fold_count, 0, sizeof(uint32_t)));
    {
      const char* me = std::getenv("KNH_MAPPED_OUT");
      mapped_out = !(me && me[0] == '0');
    }
    if (desc.mix_mode == KNH_MIX_LEFT_FOLD) KNH_HIP(ensure_voices());
    bool any_wrapped = false;
    for (auto& S : stages) any_wrapped = any_wrapped || S.dcpb > 0;
    if (any_wrapped) next_delay.assign(static_cast<size_t>(n_params_total) * nv, 0);
    wrapped_index.assign(stages.size(), -1);
    n_wrapped = 0;
    for (size_t si = 0; si < stages.size(); ++si)
      if (fastq(stages[si])) wrapped_index[si] = static_cast<int>(n_wrapped++);
    node_q.assign(static_cast<size_t>(nv) * n_wrapped, NodeQ{0u, 0, 0, 0});
    {  // which of the wrapped nodes have their queues resolved on the device (kernels_events.hip): up to eight per voice
      const char* de = std::getenv("KNH_DEV_EVENTS");
      stage_dev.assign(stages.size(), 0);
      std::vector<knh_dev::DevStage> ds(stages.size());
      int n_dev = 0;
      for (size_t si = 0; si < stages.size(); ++si) {
        const StageInfo& S = stages[si];
        const bool on = fastq(S) && dev_resolvable_kind(S.kind) && n_dev < 8 && !(de && de[0] == '0') && S.slot_base < 65536 && n_params_total < 65536;
        ds[si] = knh_dev::DevStage{S.kind, S.dcpb, static_cast<unsigned short>(S.slot_base), static_cast<unsigned short>(S.param_base), S.flags, S.ar_param,
                                   static_cast<short>(on ? n_dev : -1), 0};
        if (on) { stage_dev[si] = 1; ++n_dev; }
      }
      dev_events = n_dev > 0;
      dev_class.assign(stages.size() * 8u, 0);
      for (size_t si = 0; si < stages.size(); ++si)
        if (stage_dev[si])
          for (int pp = 0; pp < stages[si].n_params && pp < 8; ++pp) dev_class[si * 8u + pp] = static_cast<uint8_t>(1 + expected_value_kind(stages[si].kind, pp));
      if (dev_events) {
        KNH_HIP(hipMalloc(&d_stages, ds.size() * sizeof(knh_dev::DevStage)));
        KNH_HIP(hipMemcpy(d_stages, ds.data(), ds.size() * sizeof(knh_dev::DevStage), hipMemcpyHostToDevice));
        KNH_HIP(hipHostMalloc(&h_ev_overflow, 64, hipHostMallocMapped | hipHostMallocCoherent));
        *h_ev_overflow = 0u;
        KNH_HIP(hipMalloc(&d_armed, static_cast<size_t>(n_params_total) * nv * sizeof(uint16_t)));
        KNH_HIP(hipMemset(d_armed, 0, static_cast<size_t>(n_params_total) * nv * sizeof(uint16_t)));
        KNH_HIP(hipMalloc(&d_ev_cnt, static_cast<size_t>(nv) * 3 * sizeof(uint32_t)));
        KNH_HIP(hipMemset(d_ev_cnt, 0, static_cast<size_t>(nv) * 3 * sizeof(uint32_t)));  // (kept zero by the resolver's last pass)
        KNH_HIP(hipMalloc(&d_rec_start, (static_cast<size_t>(nv) + 1) * sizeof(uint32_t)));
        KNH_HIP(hipStreamCreateWithFlags(&ev_stream, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
          KNH_HIP(hipMalloc(&d_out_start2[b], (static_cast<size_t>(nv) + 1) * sizeof(uint32_t)));
          KNH_HIP(hipEventCreateWithFlags(&recs_done[b], hipEventDisableTiming));
          KNH_HIP(hipEventCreateWithFlags(&lists_free[b], hipEventDisableTiming));
        }
      }
    }
    smooth.assign(stages.size(), {});
    smooth_mark.assign(stages.size(), {});
    for (size_t si = 0; si < stages.size(); ++si)
      if (stages[si].flags & KNH_STAGE_FLAG_SMOOTH_PARAMS) {
        smooth[si].assign(static_cast<size_t>(nv) * stages[si].n_params, SmoothState{});
        smooth_mark[si].assign(nv, 0u);
      }
    initialised = true;
    return KNH_OK;
  }
  bool pan = false;          // the chain ends in a Pan2
  unsigned fold_planes = 1;  // channel planes of the partial rows and of the per-voice output: 2 for a Pan2 chain
  hipError_t ensure_voices() {
    if (d_voices) return hipSuccess;
    return hipMalloc(&d_voices, static_cast<size_t>(fold_planes) * nv * block_size * sizeof(F));
  }

  // ---- parameter changes ----------------------------------------------------------------
  int check_target(uint32_t voice, uint32_t stage, uint32_t param) {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    if (voice >= nv) return fail(KNH_ERR_OUT_OF_RANGE, "voice out of range");
    if (stage >= stages.size()) return fail(KNH_ERR_OUT_OF_RANGE, "stage out of range");
    if (param >= static_cast<uint32_t>(stages[stage].n_params)) return fail(KNH_ERR_OUT_OF_RANGE, "parameter index out of range");
    return KNH_OK;
  }
  int set_delay(uint32_t voice, uint32_t stage, uint32_t param, uint16_t delay) override {
    int rc = check_target(voice, stage, param);
    if (rc != KNH_OK) return rc;
    const StageInfo& S = stages[stage];
    if (S.dcpb == 0) {  // ugen.rs:339-341
      warn("Parameter delay set, but the stage is not wrapped in WrPreciseTiming; no effect");
      return KNH_OK;
    }
    if (fastq(S)) {
      QRec r{};
      r.voice = voice; r.delay = delay; r.stage = static_cast<uint16_t>(stage); r.param = static_cast<uint8_t>(param); r.kb = 0x10u;
      return push_rec(0, r);
    }
    next_delay[static_cast<size_t>(S.param_base + param) * nv + voice] = delay;  // precise_timing.rs:146-148
    return KNH_OK;
  }
  int param_apply(uint32_t voice, uint32_t stage, uint32_t param, uint32_t kind, double f, int64_t i) override {
    int rc = check_target(voice, stage, param);
    if (rc != KNH_OK) return rc;
    const StageInfo& S = stages[stage];
    if (!kind_ok(S, param, kind)) return fail(KNH_ERR_WRONG_VALUE_KIND, "parameter value kind does not match the parameter type");
    if (fastq(S) && frame_base == 0) {  // (frame_base != 0: a replay inside process, which has its own order)
      return push_rec(0, make_qrec(voice, stage, param, kind, f, i, 0, false));
    }
    if (S.dcpb > 0) {  // WrPreciseTiming::param_apply, precise_timing.rs:126-135
      uint16_t d = next_delay[static_cast<size_t>(S.param_base + param) * nv + voice];
      if (d != 0) {  // capacity (DELAYED_CHANGES_PER_BLOCK) is enforced per node when the block is assembled
        queued.emplace_back(static_cast<uint64_t>(voice) * stages.size() + stage, QueuedChange{d, param, kind, f, i});
        return KNH_OK;
      }
    }
    deliver(voice, stage, param, kind, f, i, frame_base);
    return KNH_OK;
  }
  static QRec make_qrec(uint32_t voice, uint32_t stage, uint32_t param, uint32_t kind, double f, int64_t i, uint16_t delay, bool arm) {
    QRec r{};
    r.voice = voice; r.delay = delay; r.stage = static_cast<uint16_t>(stage); r.param = static_cast<uint8_t>(param);
    r.kb = static_cast<uint8_t>((kind & 15u) | (arm ? 0x10u : 0u) | 0x20u);
    if (kind == KNH_VALUE_FLOAT) r.v.f = f; else r.v.i = i;
    return r;
  }
  static bool kind_ok(const StageInfo& S, uint32_t param, uint32_t kind) {
    const int want = expected_value_kind(S.kind, param);
    if (static_cast<int>(kind) == want) return true;
    // ParameterValue::Smoothing is accepted by a WrSmoothParams-wrapped node for its Float parameters; without
    // the wrapper the reference's generated param_apply panics on it (knaster_macros/src/lib.rs:601-606,752-757)
    return kind == KNH_VALUE_SMOOTHING && (S.flags & KNH_STAGE_FLAG_SMOOTH_PARAMS) && want == KNH_VALUE_FLOAT;
  }
  // WrSmoothParams::param_apply (smooth_params.rs:210-259) in front of the node's own setters.
  void deliver(uint32_t voice, uint32_t stage, uint32_t param, uint32_t kind, double f, int64_t iv, uint32_t frame) {
    const StageInfo& S = stages[stage];
    if (!(S.flags & KNH_STAGE_FLAG_SMOOTH_PARAMS) || (kind != KNH_VALUE_FLOAT && kind != KNH_VALUE_SMOOTHING)) {
      apply_now(voice, stage, param, f, iv, frame, pending);
      return;
    }
    SmoothState& st = smooth[stage][static_cast<size_t>(voice) * S.n_params + param];
    if (kind == KNH_VALUE_SMOOTHING) {  // set_smoothing, :30-102
      if (iv == 0) {
        if (st.linear) {
          double cv = st.interpolated();
          st = SmoothState{};
          st.current_value = cv;
        }
        return;
      }
      const size_t dur = static_cast<size_t>(static_cast<double>(static_cast<float>(f)) * static_cast<double>(sample_rate));
      if (!st.linear) {
        double cv = st.current_value;
        st.linear = true;
        st.start_value = cv;
        st.end_value = cv;
        st.frames_elapsed = 0;
      } else if (st.done) {
        st.start_value = st.end_value;
        st.frames_elapsed = 0;
      } else {
        st.start_value = st.interpolated();
      }
      st.duration_frames = dur;
      st.audio_rate = iv == 2;
      st.done = true;
      return;
    }
    if (!st.linear) {  // no smoothing selected for this parameter: straight through
      apply_now(voice, stage, param, f, iv, frame, pending);
      return;
    }
    st.start_value = st.done ? st.end_value : st.interpolated();
    st.end_value = f;
    st.done = false;
    st.frames_elapsed = 0;
    uint32_t& mark = smooth_mark[stage][voice];
    if (!(mark & 1u)) {
      mark |= 1u;
      smooth_active.push_back(static_cast<uint64_t>(voice) * stages.size() + stage);
    }
  }
  // WrSmoothParams::process_block's block-rate step (:188-197) for one node, at the start of a (partial) block.
  void smooth_tick(uint32_t voice, uint32_t stage, uint32_t frame) {
    const StageInfo& S = stages[stage];
    SmoothState* st = &smooth[stage][static_cast<size_t>(voice) * S.n_params];
    for (int p = 0; p < S.n_params; ++p) {
      SmoothState& x = st[p];
      if (!x.linear || x.done) continue;  // next_value, :263-300 (frame_in_block is 0 on this path)
      const double v = x.interpolated();
      if (x.frames_elapsed == x.duration_frames) x.done = true;
      else if (x.audio_rate) x.frames_elapsed += 1;
      else x.frames_elapsed = std::min(x.frames_elapsed + block_size, x.duration_frames);
      apply_now(voice, stage, static_cast<uint32_t>(p), v, 0, frame_base + frame, pending);
    }
    smooth_mark[stage][voice] = (smooth_mark[stage][voice] & 1u) | (smooth_epoch << 1);
  }
  // The same two calls addressed to block `block_offset` of the next multi-block launch: validated now,
  // replayed in order when that block is assembled.
  int check_call(uint32_t voice, uint32_t stage, uint32_t param, uint32_t kind) override {
    int rc = check_target(voice, stage, param);
    if (rc != KNH_OK) return rc;
    if (!kind_ok(stages[stage], param, kind)) return fail(KNH_ERR_WRONG_VALUE_KIND, "parameter value kind does not match the parameter type");
    return KNH_OK;
  }
  int call_at(uint32_t block_offset, bool is_delay, uint32_t voice, uint32_t stage, uint32_t param, uint32_t kind, double f,
              int64_t i, uint16_t delay) override {
    if (block_offset == 0) return is_delay ? set_delay(voice, stage, param, delay) : param_apply(voice, stage, param, kind, f, i);
    int rc = check_target(voice, stage, param);
    if (rc != KNH_OK) return rc;
    if (block_offset >= 65536) return fail(KNH_ERR_OUT_OF_RANGE, "block_offset too large");
    if (!is_delay && !kind_ok(stages[stage], param, kind))
      return fail(KNH_ERR_WRONG_VALUE_KIND, "parameter value kind does not match the parameter type");
    if (fastq(stages[stage])) {
      if (is_delay) {
        QRec r{};
        r.voice = voice; r.delay = delay; r.stage = static_cast<uint16_t>(stage); r.param = static_cast<uint8_t>(param); r.kb = 0x10u;
        return push_rec(block_offset, r);
      }
      return push_rec(block_offset, make_qrec(voice, stage, param, kind, f, i, 0, false));
    }
    if (future.size() <= block_offset) future.resize(block_offset + 1);
    future[block_offset].push_back(Call{static_cast<uint8_t>(is_delay), delay, voice, stage, param, kind, f, i});
    return KNH_OK;
  }

  // A parameter whose device patches depend on the new value alone (no shadow of an earlier value is read, and no other
  // parameter's patches touch the same words): a call for a later block of the launch can be turned into its patches
  // at once, instead of being kept and replayed when that block is assembled.  Stages wrapped in WrPreciseTiming or
  // WrSmoothParams keep state on the host per call and always take the general path.
  bool direct_ok(const StageInfo& S, uint32_t param) const {
    if (S.dcpb > 0 || (S.flags & KNH_STAGE_FLAG_SMOOTH_PARAMS)) return false;
    switch (S.kind) {
      case KNH_STAGE_SVF: return false;             // every setter recomputes from the three shadows
      case KNH_STAGE_BUFFER_READER: return false;   // start / duration / rate shadows
      case KNH_STAGE_MUL_ENV_ASR: case KNH_STAGE_MUL_ENV_AR: return param >= 2;  // the times skip when unchanged (shadow); the triggers do not
      default: return true;
    }
  }
  // knh_bank_param_apply_range: an envelope trigger for the voices [v0, v1) is one range event, in O(1); anything else is the
  // batch it stands for (bank_base.hpp).
  int apply_range(uint32_t v0, uint32_t v1, uint32_t stage, uint32_t param, uint32_t kind, double f, int64_t iv) override {
    if (v1 > v0 && v1 <= nv && v1 - v0 >= 16 && initialised && stage < stages.size() && kind == KNH_VALUE_TRIGGER &&
        param < static_cast<uint32_t>(stages[stage].n_params) && kind_ok(stages[stage], param, kind) && direct_ok(stages[stage], param) &&
        (stages[stage].kind == KNH_STAGE_MUL_ENV_ASR || stages[stage].kind == KNH_STAGE_MUL_ENV_AR) &&
        !(stages[stage].ar_param != 0 && param + 1u == stages[stage].ar_param) && pending_ranges.size() < 64) {
      const StageInfo& S = stages[stage];
      note_frame(0u);
      const bool release = S.kind == KNH_STAGE_MUL_ENV_ASR && param == 2;
      pending_ranges.push_back(RangeEvent{v0, v1, 0u, release ? static_cast<uint32_t>(knh_dev::EV_ENV_ASR_RELEASE) : static_cast<uint32_t>(knh_dev::EV_SET),
                                          static_cast<uint32_t>(S.slot_base), release ? 0ull : 1ull, pending.size()});
      return KNH_OK;
    }
    return knh_bank::apply_range(v0, v1, stage, param, kind, f, iv);
  }
  // knh_bank_param_apply_many[_at]: runs of calls to the same (stage, parameter, kind) -- how a host sends "this parameter
  // of these voices" -- are checked once and turned into patches in one pass; anything else goes call by call.
  int apply_many(uint32_t block_offset, size_t count, const uint32_t* voices, const uint32_t* stgs, const uint32_t* params,
                 const uint32_t* kinds, const double* fvalues, const int64_t* ivalues, const uint16_t* delays) override {
    if (!initialised || count < 16 || block_offset >= 65536) return knh_bank::apply_many(block_offset, count, voices, stgs, params, kinds, fvalues, ivalues, delays);
    int rc = KNH_OK;
    size_t k = 0;
    while (k < count) {
      if (dev_events) {
        // Calls to nodes whose queues the DEVICE resolves: one table look-up and one 24-byte record in pinned memory per call,
        // whatever the order of stages and parameters in the batch (a host that addresses two parameters of alternate voices
        // -- BASELINE config C5 -- sends runs of one call).  dev_class[stage][param] = 1 + the ParameterValue kind it takes.
        const size_t ns = stages.size();
        if (stgs[k] < ns && params[k] < 8u && kinds[k] < 8u && dev_class[stgs[k] * 8u + params[k]] == kinds[k] + 1u) {
          int r2 = dev_reserve(count - k);
          if (r2 != KNH_OK) return r2;
          uint64_t* out = reinterpret_cast<uint64_t*>(h_recs + n_recs);  // three 8-byte words per record (QRec's layout)
          const uint64_t blk = static_cast<uint64_t>(block_offset & 0xFFFFu) << 16;
          size_t p = k, w = 0;
          for (; p < count; ++p) {
            const uint32_t st = stgs[p], pr = params[p], kd = kinds[p];
            if (!(st < ns && pr < 8u && kd < 8u && dev_class[st * 8u + pr] == kd + 1u)) break;
            const uint32_t v = voices[p];
            if (v >= nv) { rc = fail(KNH_ERR_OUT_OF_RANGE, "voice out of range"); continue; }
            const uint64_t d = delays ? delays[p] : 0u;
            uint64_t val;
            if (kd == KNH_VALUE_FLOAT) { const double f = fvalues ? fvalues[p] : 0.0; std::memcpy(&val, &f, 8); }
            else { const int64_t iv = ivalues ? ivalues[p] : 0; std::memcpy(&val, &iv, 8); }
            out[3 * w + 0] = static_cast<uint64_t>(v) | (d << 32) | (static_cast<uint64_t>(st) << 48);
            out[3 * w + 1] = static_cast<uint64_t>(pr) | (static_cast<uint64_t>(kd | (d ? 0x10u : 0u) | 0x20u) << 8) | blk;
            out[3 * w + 2] = val;
            ++w;
          }
          n_recs += w;
          recs_max_block = std::max(recs_max_block, block_offset);
          k = p;
          continue;
        }
      }
      size_t e = k + 1;
      {  // (64 entries at a time without a branch in between -- the compiler vectorises that -- then the ragged end one by one:
         // a bank's 16 384 triggers are 200 KB of arrays to look through)
        const uint32_t s0 = stgs[k], p0 = params[k], k0 = kinds[k];
        while (e + 64 <= count) {
          uint32_t d = 0;
          for (size_t q = e; q < e + 64; ++q) d |= (stgs[q] ^ s0) | (params[q] ^ p0) | (kinds[q] ^ k0);
          if (d) break;
          e += 64;
        }
      }
      while (e < count && stgs[e] == stgs[k] && params[e] == params[k] && kinds[e] == kinds[k]) ++e;
      bool direct = e - k >= 16 && stgs[k] < stages.size() && params[k] < static_cast<uint32_t>(stages[stgs[k]].n_params) &&
                    kind_ok(stages[stgs[k]], params[k], kinds[k]) && direct_ok(stages[stgs[k]], params[k]);
      if (direct && delays)
        for (size_t q = k; q < e && direct; ++q) direct = delays[q] == 0;  // an armed delay on an unwrapped stage: the warning path
      // Calls already kept for that block (they are replayed when the block is assembled) come first: a later call must not
      // overtake them by being turned into its patches now (two changes of one parameter in one block: the last one holds).
      if (direct && block_offset > 0 && block_offset < future.size() && !future[block_offset].empty()) direct = false;
      // (runs of any length: a host that addresses two parameters of alternate voices sends runs of one)
      const bool queued_run = !direct && stgs[k] < stages.size() && params[k] < static_cast<uint32_t>(stages[stgs[k]].n_params) &&
                              fastq(stages[stgs[k]]) && kind_ok(stages[stgs[k]], params[k], kinds[k]) && kinds[k] != KNH_VALUE_SMOOTHING;
      if (queued_run) {  // calls to a WrPreciseTiming-wrapped node: one record each (arm the delay, then the value)
        std::vector<QRec>& q = qblock(block_offset);
        if (q.capacity() < q.size() + (e - k)) q.reserve(std::max(q.size() + (count - k), q.capacity() * 2));
        const uint32_t stage = stgs[k], param = params[k], kind = kinds[k];
        for (size_t p = k; p < e; ++p) {
          const uint32_t v = voices[p];
          if (v >= nv) { rc = fail(KNH_ERR_OUT_OF_RANGE, "voice out of range"); continue; }
          const uint16_t d = delays ? delays[p] : 0;
          q.push_back(make_qrec(v, stage, param, kind, fvalues ? fvalues[p] : 0.0, ivalues ? ivalues[p] : 0, d, d > 0));
        }
      } else if (direct && kinds[k] == KNH_VALUE_TRIGGER && (stages[stgs[k]].kind == KNH_STAGE_MUL_ENV_ASR || stages[stgs[k]].kind == KNH_STAGE_MUL_ENV_AR) &&
                 !(stages[stgs[k]].ar_param != 0 && params[k] + 1u == stages[stgs[k]].ar_param)) {
        // An envelope trigger for a run of voices (the note-on / note-off of a whole bank): the patch is the same for every voice
        // (apply_now: EV_SET state = Attacking, or the release op) -- one event template, stamped with each voice's index.
        const StageInfo& S = stages[stgs[k]];
        const uint32_t frame = block_offset * static_cast<uint32_t>(block_size);
        note_frame(frame);
        const bool release = S.kind == KNH_STAGE_MUL_ENV_ASR && params[k] == 2;
        HostEvent t{0u, frame, release ? static_cast<uint32_t>(knh_dev::EV_ENV_ASR_RELEASE) : static_cast<uint32_t>(knh_dev::EV_SET), static_cast<uint32_t>(S.slot_base), release ? 0ull : 1ull};
        {  // neighbouring voices in rising order (the usual way to address a bank): one range event
          bool run = voices[e - 1] < nv && pending_ranges.size() < 64;
          size_t q = k + 1;
          for (; q + 64 <= e && run; q += 64) {  // (branch-free runs of 64, as above)
            uint32_t d = 0;
            for (size_t j = q; j < q + 64; ++j) d |= voices[j] - voices[j - 1] - 1u;
            run = d == 0;
          }
          for (; q < e && run; ++q) run = voices[q] == voices[q - 1] + 1u;
          if (run) {
            pending_ranges.push_back(RangeEvent{voices[k], voices[e - 1] + 1u, frame, t.op, t.slot, t.bits, pending.size()});
            k = e;
            continue;
          }
        }
        const size_t at = pending.size();
        pending.resize(at + (e - k));
        HostEvent* out = pending.data() + at;
        size_t w = 0;
        for (size_t q = k; q < e; ++q) {
          const uint32_t v = voices[q];
          if (v >= nv) { rc = fail(KNH_ERR_OUT_OF_RANGE, "voice out of range"); continue; }
          t.voice = v;
          out[w++] = t;
        }
        pending.resize(at + w);
      } else if (direct) {
        const uint32_t stage = stgs[k], param = params[k];
        const uint32_t frame = block_offset * static_cast<uint32_t>(block_size);
        pending.reserve(pending.size() + (e - k) * 2);
        for (size_t q = k; q < e; ++q) {
          const uint32_t v = voices[q];
          if (v >= nv) { rc = fail(KNH_ERR_OUT_OF_RANGE, "voice out of range"); continue; }
          apply_now(v, stage, param, fvalues ? fvalues[q] : 0.0, ivalues ? ivalues[q] : 0, frame, pending);
        }
      } else {
        int r = knh_bank::apply_many(block_offset, e - k, voices + k, stgs + k, params + k, kinds + k, fvalues ? fvalues + k : nullptr,
                                     ivalues ? ivalues + k : nullptr, delays ? delays + k : nullptr);
        if (r != KNH_OK) rc = r;
      }
      k = e;
    }
    return rc;
  }

  // The parameter setters of each UGen, restated as "new shadow value -> device patches".
  // Events reach `pending` in application order; a voice's list must also be in frame order.  As long as every new event's
  // frame is at least the largest one seen, both hold by construction and the per-voice sort of upload_events is skipped.
  uint32_t pending_max_frame = 0;
  void note_frame(uint32_t frame) {
    if (frame < pending_max_frame) pending_needs_sort = true;
    else pending_max_frame = frame;
  }
  void apply_now(uint32_t v, uint32_t stage, uint32_t param, double f, int64_t iv, uint32_t frame, std::vector<HostEvent>& out) {
    const StageInfo& S = stages[stage];
    Shadow& sh = shadow[stage];
    // a parameter a signal drives at audio rate ignores ordinary changes while the link stands (audio_rate.rs:70-74)
    if (S.ar_param != 0 && param + 1u == S.ar_param) return;
    note_frame(frame);
    auto set = [&](int rel, uint64_t bits) { out.push_back(HostEvent{v, frame, knh_dev::EV_SET, static_cast<uint32_t>(S.slot_base + rel), bits}); };
    const F sr_as_f32 = static_cast<F>(static_cast<float>(sample_rate));
    switch (S.kind) {
      case KNH_STAGE_SIN_WT:
        if (param == 0) {  // osc.rs:127-130; ignored while an audio-rate buffer drives it (audio_rate.rs:70-74)
          if (S.flags & KNH_STAGE_FLAG_AR_FREQ) return;
          F freq = static_cast<F>(f);
          sh.a[v] = freq;
          set(2, sat_u32(static_cast<double>(freq) * f2pi));
        } else if (param == 1) {  // osc.rs:133-135
          set(1, sat_u32(f * 65536.0));
        } else {
          set(0, 0);  // reset_phase
        }
        break;
      case KNH_STAGE_SIN_NUMERIC:
        if (param == 0) set(2, to_bits(static_cast<F>(f) / sr_as_f32));  // osc.rs:240-242
        else if (param == 1) set(1, to_bits(static_cast<F>(f)));
        else set(0, to_bits(F(0)));
        break;
      case KNH_STAGE_SVF: {  // svf.rs:81-133: every setter recomputes the coefficients
        if (param == 0) sh.a[v] = static_cast<F>(f);
        else if (param == 1) sh.b[v] = static_cast<F>(f);
        else if (param == 2) sh.c[v] = static_cast<F>(f);
        else if (param == 3) sh.ty[v] = (iv >= 0 && iv <= 8) ? static_cast<uint8_t>(iv) : 0;  // knaster_macros/src/lib.rs:44-47
        F co[6];
        svf_coeffs<F>(sh.ty[v], sh.a[v], sh.b[v], sh.c[v], sr_as_f32, co);
        for (int k = 0; k < 6; ++k) set(2 + k, to_bits(co[k]));
        if (S.n_slots == 12) {  // the values the device-side setter reads (another parameter of this filter is driven at audio rate)
          if (param == 0) set(8, to_bits(sh.a[v]));
          else if (param == 1) set(9, to_bits(sh.b[v]));
          else if (param == 2) set(10, to_bits(sh.c[v]));
          else if (param == 3) set(11, sh.ty[v]);
        }
      } break;
      case KNH_STAGE_ONEPOLE_LPF:
      case KNH_STAGE_ONEPOLE_HPF: {  // onepole.rs:135-139,172-176 -> :35-46
        F fr = static_cast<F>(f) / static_cast<F>(sample_rate);
        F b1 = std::exp(F(-2.0) * Consts<F>::PI * fr);
        set(2, to_bits(b1));
        set(1, to_bits(F(1.0) - b1));
      } break;
      case KNH_STAGE_MUL_ENV_ASR:
      case KNH_STAGE_MUL_ENV_AR:
        if (param == 0 || param == 1) {  // envelopes.rs:85-110 / :236-261 (skip when unchanged)
          std::vector<F>& secs = param == 0 ? sh.a : sh.b;
          F s = static_cast<F>(f);
          if (secs[v] != s) {
            secs[v] = s;
            F rate = s == F(0) ? F(1) : F(1) / (s * static_cast<F>(sample_rate));
            set(param == 0 ? 2 : 3, to_bits(rate));
          }
        } else if (S.kind == KNH_STAGE_MUL_ENV_ASR && param == 2) {  // t_release needs the live state: device op
          out.push_back(HostEvent{v, frame, knh_dev::EV_ENV_ASR_RELEASE, static_cast<uint32_t>(S.slot_base), 0});
        } else {
          set(0, 1);  // t_restart: state = Attacking, t untouched (envelopes.rs:47-49,131-133)
        }
        break;
      case KNH_STAGE_BUFFER_READER: {  // buffer.rs:62-103
        auto set2 = [&](int rel, double d) {
          const uint64_t b = to_bits(d);
          set(rel, static_cast<uint32_t>(b));
          set(rel + 1, static_cast<uint32_t>(b >> 32));
        };
        auto secs_to_frames = [&](double secs) {
          const double whole = std::floor(secs);
          const uint32_t tes = sat_u32((secs - std::trunc(secs)) * 282240000.0);
          return static_cast<double>(sat_u32(whole)) * buffer_sr + (static_cast<double>(tes) * buffer_sr) / 282240000.0;
        };
        switch (param) {
          case 0: buf_rate[v] = f; set2(2, buf_base_rate * f); break;
          case 1: set(9, iv != 0 ? 1u : 0u); break;
          case 2: buf_start[v] = secs_to_frames(f); set2(4, buf_start[v]); set2(6, buf_start[v] + buf_dur[v]); break;
          case 3: buf_dur[v] = secs_to_frames(f); set2(6, buf_start[v] + buf_dur[v]); break;
          case 4: set2(6, secs_to_frames(f)); break;
          default: set2(0, buf_start[v]); set(8, 0); break;  // t_restart -> reset -> jump_to(start_frame)
        }
      } break;
      case KNH_STAGE_POLYBLEP: {  // polyblep.rs:158-182
        const F srf = static_cast<F>(sample_rate);
        if (param == 0) {
          const F dt = static_cast<F>(f) / srf;
          set(1, to_bits(dt));
          set(4, (dt * srf >= srf / F(4)) ? 1u : 0u);
        } else if (param == 1) {
          set(2, to_bits(static_cast<F>(f)));
        } else {  // Waveform::from(PInteger): out of range -> default (Sawtooth)
          set(3, iv >= 0 && iv < 14 ? static_cast<uint64_t>(iv) : 0u);
        }
      } break;
      case KNH_STAGE_PAN2: {  // Pan2::pan(pan: f32), pan.rs:26-29 (the macro hands the PFloat over `as f32`)
        float gl, gr;
        pan2_gains(static_cast<float>(f), &gl, &gr);
        set(0, to_bits(static_cast<F>(gl)));
        set(1, to_bits(static_cast<F>(gr)));
      } break;
      case KNH_STAGE_RANDOM_LIN: {  // noise.rs:213-221: phase_step = F::new(value) * freq_to_phase_inc
        const F inc = F(1) / static_cast<F>(sample_rate);
        set(5, to_bits(static_cast<F>(static_cast<F>(f) * inc)));
      } break;
      case KNH_STAGE_PHASOR: {  // osc.rs:189-196
        const uint64_t sb = to_bits(f * (1.0 / static_cast<double>(sample_rate)));
        set(2, static_cast<uint32_t>(sb));
        set(3, static_cast<uint32_t>(sb >> 32));
      } break;
      case KNH_STAGE_ALLPASS_FB_DELAY:
        if (param == 1) {  // feedback, :237-240
          set(7, to_bits(static_cast<F>(f)));
          break;
        }
        [[fallthrough]];  // delay_time, :231-236: set_delay_in_frames without the length check of AllpassDelay's
      case KNH_STAGE_ALLPASS_DELAY: {  // delay_time, :136-143 -> set_delay_in_frames, :160-174
        const double delay_frames = f * static_cast<double>(sample_rate);
        const uint32_t len = delay_len[v];
        if (!(delay_frames < static_cast<double>(len))) {  // `(delay_frames as usize) < buffer.len()` fails: ignored
          if (S.kind == KNH_STAGE_ALLPASS_FB_DELAY) warn("AllpassFeedbackDelay: delay_time longer than the ring, change ignored");
          break;
        }
        F num = static_cast<F>(delay_frames);  // F::new; a negative or NaN value casts to 0 frames above, and goes on as it is
        const F fl = std::floor(num);
        uint32_t whole = fl > F(0) ? static_cast<uint32_t>(fl) : 0u;  // to_usize().unwrap() on a negative value panics in the reference
        F delta = num - fl;
        if (num > F(0.5) && delta < F(0.5)) {
          delta += F(1);
          whole -= 1u;
        }
        out.push_back(HostEvent{v, frame, knh_dev::EV_ALLPASS_DELAY, static_cast<uint32_t>(S.slot_base), whole});
        set(4, to_bits((F(1) - delta) / (F(1) + delta)));  // AllpassInterpolator::set_delta, :74-76
      } break;
      case KNH_STAGE_SAMPLE_DELAY: {  // delay.rs:33-36: delay_samples = (seconds * sample_rate) as usize
        const double ds = f * static_cast<double>(sample_rate);
        const uint32_t len = delay_len[v];
        if (!(ds < static_cast<double>(len) + 1.0)) {  // the reference would read outside its buffer
          warn("SampleDelay: delay_time longer than the ring, change ignored");
          break;
        }
        const uint32_t d = ds > 0.0 ? static_cast<uint32_t>(ds) : 0u;  // NaN and negatives -> 0, as `as usize` does
        set(1, len - d);
      } break;
      case KNH_STAGE_MUL_ENVELOPE: {  // envelopes.rs:478-524
        auto set2 = [&](int rel, double d) {
          uint64_t b = to_bits(d);
          set(rel, static_cast<uint32_t>(b));
          set(rel + 1, static_cast<uint32_t>(b >> 32));
        };
        if (param == 0) {  // time_scale = F::new(value).to_f64()
          set2(6, static_cast<double>(static_cast<F>(f)) * (1.0 / static_cast<double>(sample_rate)));
        } else if (param == 1) {  // jump_to_segment (clamped), state = Running { segment, 0.0 }
          uint64_t j = iv < 0 ? 0 : static_cast<uint64_t>(iv);
          if (j >= env_nseg[v]) j = env_nseg[v] - 1;
          set(0, 1);
          set2(2, 0.0);
          set(1, j);
        } else if (param == 2) {  // t_restart
          set(0, 1);
          set2(2, 0.0);
          set2(4, env_start[v]);
          set(1, 0);
        } else {  // t_stop needs the live time: device op
          out.push_back(HostEvent{v, frame, knh_dev::EV_SEGENV_STOP, static_cast<uint32_t>(S.slot_base), 0});
        }
      } break;
      default:  // Constant::value (util.rs:47-50) / WrMul "wr_mul" (wrappers_core/math.rs:92-98)
        set(0, to_bits(static_cast<F>(f)));
        break;
    }
  }

  // The records of one block, in arrival order: set_delay_within_block_for_param arms (precise_timing.rs:146-148),
  // param_apply queues when a delay is armed (:126-135, capacity DELAYED_CHANGES_PER_BLOCK) and applies at once otherwise,
  // and process_block's change loop (:65-114) applies a node's queued changes first in, first out, each at
  // max(its delay, where the node's block has got to) -- a change behind one that is not due inside the processed range
  // is never reached.  A node's events come out in frame order by construction.
  struct DueRec { uint32_t rec; uint32_t due; };
  std::vector<DueRec> due_scratch;
  void resolve_qrecs(std::vector<QRec>& recs, uint32_t frame_begin, uint32_t frame_end) {
    if (recs.empty()) return;
    q_epoch += 1;
    if (q_epoch == 0) {  // wrapped around: no stale state may look current
      std::fill(node_q.begin(), node_q.end(), NodeQ{0u, 0, 0, 0});
      q_epoch = 1;
    }
    pending.reserve(pending.size() + recs.size());
    // Two passes, as the reference's time runs: a call without an armed delay goes straight through to the node when it is
    // made -- between two blocks, so BEFORE every queued change of the block is applied (those are applied inside
    // process_block) -- whatever the order the calls arrived in.  That order matters for setters that compute from what
    // the other parameters are at that moment (SvfFilter's cutoff / q / gain, the envelope times): a cutoff set at once
    // after a q change was queued is computed with the old q, and the q change, when due, with the new cutoff.
    due_scratch.clear();
    for (size_t ri = 0; ri < recs.size(); ++ri) {
      const QRec& r = recs[ri];
      const StageInfo& S = stages[r.stage];
      uint16_t& armed = next_delay[static_cast<size_t>(S.param_base + r.param) * nv + r.voice];
      if (r.arm()) armed = r.delay;
      if (!r.has_value()) continue;
      const double f = r.kind() == KNH_VALUE_FLOAT ? r.v.f : 0.0;
      const int64_t iv = r.kind() == KNH_VALUE_FLOAT ? 0 : r.v.i;
      if (armed == 0) {  // no delay armed: straight through, before the block
        apply_now(r.voice, r.stage, r.param, f, iv, frame_base, pending);
        continue;
      }
      NodeQ& q = node_q[static_cast<size_t>(r.voice) * n_wrapped + static_cast<uint32_t>(wrapped_index[r.stage])];
      if (q.epoch != q_epoch) q = NodeQ{q_epoch, static_cast<uint16_t>(frame_begin), 0, 0};
      if (q.taken >= S.dcpb) {  // precise_timing.rs:129-134: the queue was full when this change arrived
        if (q.taken == S.dcpb) { warn("Not enough space for scheduled changes in WrPreciseTiming, change ignored"); q.taken = static_cast<uint16_t>(std::min<uint32_t>(S.dcpb + 1u, 32767u)); }
        continue;
      }
      q.taken = static_cast<uint16_t>(q.taken + 1);
      if (q.blocked) continue;  // behind a change that is not due in this block: never reached
      const uint32_t due = std::max<uint32_t>(armed, q.at);
      if (due > frame_end) { q.blocked = 1; continue; }
      q.at = static_cast<uint16_t>(due);
      due_scratch.push_back(DueRec{static_cast<uint32_t>(ri), due});
    }
    for (const auto& d : due_scratch) {  // (a node's queued changes: first in, first out, their due frames never decrease)
      const QRec& r = recs[d.rec];
      const StageInfo& S = stages[r.stage];
      const double f = r.kind() == KNH_VALUE_FLOAT ? r.v.f : 0.0;
      const int64_t iv = r.kind() == KNH_VALUE_FLOAT ? 0 : r.v.i;
      const uint32_t due = d.due;
      const size_t first_ev = pending.size();
      apply_now(r.voice, r.stage, r.param, f, iv, frame_base + due, pending);
      if (due > frame_begin) {  // a split point: the node's block restarts here (precise_timing.rs:104-110)
        if (pending.size() == first_ev) {
          note_frame(frame_base + due);
          pending.push_back(HostEvent{r.voice, frame_base + due, knh_dev::EV_NOP, static_cast<uint32_t>(S.slot_base), 0});
        }
        for (size_t e = first_ev; e < pending.size(); ++e) pending[e].op |= knh_dev::EV_SPLIT;
      }
    }
    recs.clear();
  }

  // WrPreciseTiming::process_block's change loop (precise_timing.rs:65-114) for every wrapped node
  // with queued changes: FIFO with head-of-line blocking, changes past the processed range are lost.
  void resolve_queues(uint32_t frame_begin, uint32_t frame_end) {  // block-relative range; events get frame_base added
    smooth_epoch = (smooth_epoch + 1) & 0x7FFFFFFFu;
    if (!queued.empty()) {
      // group by node, keeping arrival order inside each node's queue
      // (callers that walk the voices in order, as the batched entry points are normally used, arrive sorted)
      auto by_node = [](const auto& a, const auto& b) { return a.first < b.first; };
      if (!std::is_sorted(queued.begin(), queued.end(), by_node)) std::stable_sort(queued.begin(), queued.end(), by_node);
      size_t i = 0;
      std::vector<std::pair<uint32_t, const QueuedChange*>> due_list;
      while (i < queued.size()) {
        const uint64_t key = queued[i].first;
        const uint32_t voice = static_cast<uint32_t>(key / stages.size());
        const uint32_t stage = static_cast<uint32_t>(key % stages.size());
        const uint32_t cap = stages[stage].dcpb;
        const bool smoothed = (stages[stage].flags & KNH_STAGE_FLAG_SMOOTH_PARAMS) != 0;
        uint32_t at = frame_begin, taken = 0;
        bool blocked = false;
        due_list.clear();
        for (; i < queued.size() && queued[i].first == key; ++i) {
          if (taken >= cap) {  // precise_timing.rs:129-134: the queue was full when this change arrived
            if (taken == cap) warn("Not enough space for scheduled changes in WrPreciseTiming, change ignored");
            ++taken;
            continue;
          }
          ++taken;
          if (blocked) continue;  // behind a change that is not due in this block: never reached
          const QueuedChange& c = queued[i].second;
          uint32_t due = std::max<uint32_t>(c.delay, at);
          if (due > frame_end) { blocked = true; continue; }
          at = due;
          due_list.emplace_back(due, &c);
        }
        // WrPreciseTiming::process_block (precise_timing.rs:65-114): apply what is due, run the inner
        // (partial) block from there, repeat.  An inner WrSmoothParams steps its ramps at the start of
        // every one of those partial blocks.
        size_t k = 0;
        uint32_t seg = frame_begin;
        bool first = true;
        while (true) {
          const size_t first_ev = pending.size();
          while (k < due_list.size() && due_list[k].first <= seg) {
            const QueuedChange& c = *due_list[k].second;
            deliver(voice, stage, c.param, c.kind, c.f, c.i, frame_base + seg);
            ++k;
          }
          if (seg < frame_end && smoothed) smooth_tick(voice, stage, seg);
          if (!first) {  // a split point: the node's block restarts here (precise_timing.rs:104-110)
            if (pending.size() == first_ev)
              pending.push_back(HostEvent{voice, frame_base + seg, knh_dev::EV_NOP, static_cast<uint32_t>(stages[stage].slot_base), 0});
            for (size_t e = first_ev; e < pending.size(); ++e) pending[e].op |= knh_dev::EV_SPLIT;
            pending_needs_sort = true;
          }
          if (k >= due_list.size()) break;
          seg = due_list[k].first;
          first = false;
        }
      }
      queued.clear();
    }
    // nodes with a ramp in flight that were not stepped above: one step at the start of the block
    if (!smooth_active.empty()) {
      size_t w = 0;
      for (size_t r = 0; r < smooth_active.size(); ++r) {
        const uint64_t key = smooth_active[r];
        const uint32_t voice = static_cast<uint32_t>(key / stages.size());
        const uint32_t stage = static_cast<uint32_t>(key % stages.size());
        if ((smooth_mark[stage][voice] >> 1) != smooth_epoch) smooth_tick(voice, stage, frame_begin);
        bool alive = false;
        const SmoothState* st = &smooth[stage][static_cast<size_t>(voice) * stages[stage].n_params];
        for (int p = 0; p < stages[stage].n_params; ++p) alive = alive || (st[p].linear && !st[p].done);
        if (alive) smooth_active[w++] = key;
        else smooth_mark[stage][voice] &= ~1u;
      }
      smooth_active.resize(w);
      pending_needs_sort = true;
    }
  }

  // ---- processing ---------------------------------------------------------------------------
  // Events addressed past the blocks of this launch (a call for block k of a later launch, turned into patches at once):
  // they stay in `pending`, k launches' worth of blocks earlier, for the next launch.
  std::vector<HostEvent> later;
  int upload_events(hipStream_t s, bool* have_events, uint32_t n_blocks, bool allow_ranges = false) {
    *have_events = false;
    later.clear();
    res_n_ranges = 0;
    const uint64_t horizon = static_cast<uint64_t>(n_blocks) * block_size;
    if (!pending_ranges.empty()) {
      bool as_ranges = allow_ranges && pending.empty() && pending_ranges.size() <= 15;
      for (const RangeEvent& r : pending_ranges) as_ranges = as_ranges && r.frame < horizon;
      if (!as_ranges) expand_ranges();
    }
    if (pending_max_frame >= horizon && !pending.empty()) {
      size_t w = 0;
      for (const HostEvent& e : pending) {
        if (e.frame >= horizon) { later.push_back(e); later.back().frame -= static_cast<uint32_t>(horizon); }
        else pending[w++] = e;
      }
      pending.resize(w);
    }
    // A resident call's per-voice events that are the same few changes for a run of neighbouring voices -- a bank's note-on
    // sent as one param_apply per voice, "this cutoff for all of them" -- are range events too: 32 bytes per change instead of
    // 16 per voice and change fetched over PCIe (a bank's 16 384 single triggers: 14 us of the call).
    if (allow_ranges && pending_ranges.empty() && pending.size() >= 32 && later.empty() && !pending_needs_sort) {
      const uint32_t v_first = pending[0].voice;
      size_t k = 1;
      while (k < pending.size() && pending[k].voice == v_first) ++k;  // the first voice's events
      bool same = k <= 15 && pending.size() % k == 0;
      const size_t n_v = same ? pending.size() / k : 0;
      same = same && v_first + n_v <= nv;
      for (size_t i = k; same && i < pending.size(); ++i) {
        const HostEvent& e = pending[i];
        const HostEvent& f = pending[i % k];
        same = e.voice == v_first + i / k && e.frame == f.frame && e.op == f.op && e.slot == f.slot && e.bits == f.bits;
      }
      if (same) {
        for (size_t j = 0; j < k; ++j) {
          const HostEvent& f = pending[j];
          pending_ranges.push_back(RangeEvent{v_first, v_first + static_cast<uint32_t>(n_v), f.frame, f.op, f.slot, f.bits, 0});
        }
        pending.clear();
      }
    }
    const size_t total = pending.empty() ? 2 * pending_ranges.size() : pending.size();
    auto finish = [&] {
      pending.clear();
      pending_needs_sort = false;
      pending_max_frame = 0;
      if (!later.empty()) {
        pending.swap(later);
        pending_needs_sort = true;  // conservatively: their order among what arrives next is not tracked
        for (const HostEvent& e : pending) pending_max_frame = std::max(pending_max_frame, e.frame);
      }
    };
    if (total == 0) { finish(); return KNH_OK; }
    const unsigned lb = list_parity;
    list_parity ^= 1u;
    if (list_busy[lb]) {  // the kernel that read this buffer two launches ago must be done with it
      KNH_HIP(hipEventSynchronize(list_done[lb]));
      list_busy[lb] = false;
    }
    if (total > h_events_cap2[lb]) {
      { int rl = res_leave(); if (rl != KNH_OK) return rl; }  // (a resident kernel knows the list by its address)
      size_t cap = std::max<size_t>(total, 1024) * 2;
      if (h_events2[lb]) KNH_HIP(hipHostFree(h_events2[lb]));
      h_events2[lb] = nullptr;
      KNH_HIP(hipHostMalloc(&h_events2[lb], cap * sizeof(Event)));
      h_events_cap2[lb] = cap;
    }
    h_ev_start = h_ev_start2[lb];
    h_events = h_events2[lb];
    list_in_use = static_cast<int>(lb);
    if (!pending_ranges.empty()) {  // (pending is empty: see above) the call's events as range events, two records each
      for (size_t j = 0; j < pending_ranges.size(); ++j) {
        const RangeEvent& r = pending_ranges[j];
        Event& d = h_events[2 * j];
        d.frame = r.frame;
        d.slot_op = (r.slot & 0xFFFFFFu) | (r.op << 24);
        d.bits = r.bits;
        Event& w = h_events[2 * j + 1];
        w.frame = r.v0;
        w.slot_op = r.v1;
        w.bits = 0;
      }
      res_n_ranges = static_cast<uint32_t>(pending_ranges.size());
      sent_ranges.swap(pending_ranges);
      pending_ranges.clear();
      finish();
      *have_events = true;
      return KNH_OK;
    }
    uint32_t* start = h_ev_start;  // nv + 2 words
    // Already in voice order (a batch of triggers for voices 0 .. N - 1, the common big list): one pass, no sort.
    bool in_voice_order = true;
    {
      uint32_t prev = 0;
      for (const HostEvent& e : pending) { if (e.voice < prev) { in_voice_order = false; break; } prev = e.voice; }
    }
    if (in_voice_order) {
      uint32_t v_next = 0, i = 0;
      for (const HostEvent& e : pending) {
        while (v_next <= e.voice) start[v_next++] = i;
        Event& d = h_events[i++];
        d.frame = e.frame;
        d.slot_op = (e.slot & 0xFFFFFFu) | (e.op << 24);
        d.bits = e.bits;
      }
      while (v_next <= nv) start[v_next++] = i;
    } else {
    // counting sort by voice (stable: keeps application order): counts two places up, so that after the prefix sum
    // start[v + 1] is where voice v's events begin, and after the scatter (which advances it) where voice v + 1's do
    std::fill(start, start + nv + 2, 0u);
    for (const HostEvent& e : pending) start[e.voice + 2]++;
    for (uint32_t v = 0; v < nv; ++v) start[v + 2] += start[v + 1];
    for (const HostEvent& e : pending) {
      Event& d = h_events[start[e.voice + 1]++];
      d.frame = e.frame;
      d.slot_op = (e.slot & 0xFFFFFFu) | (e.op << 24);
      d.bits = e.bits;
    }
    }
    if (pending_needs_sort) {  // then each voice's few events by frame (stable)
      for (uint32_t v = 0; v < nv; ++v) {
        Event* b = h_events + h_ev_start[v];
        Event* e = h_events + h_ev_start[v + 1];
        auto by_frame = [](const Event& x, const Event& y) { return x.frame < y.frame; };
        if (e - b > 1 && !std::is_sorted(b, e, by_frame)) std::stable_sort(b, e, by_frame);
      }
    }
    (void)s;
    finish();
    *have_events = true;
    return KNH_OK;
  }

  int process(uint32_t n_blocks, size_t ftp, size_t offset, uint64_t /*clock*/, void* out_host, void* out_device, void* voices_host,
              uint32_t* out_flags, void* stream, bool sync, bool accumulate) override {
    if (accumulate && !out_device) return fail(KNH_ERR_INVALID_ARGUMENT, "accumulation needs a device output buffer");
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    if (offset + ftp > block_size) return fail(KNH_ERR_INVALID_ARGUMENT, "block_start_offset + frames_to_process exceeds block_size");
    if (n_blocks == 0 || n_blocks > 4096) return fail(KNH_ERR_INVALID_ARGUMENT, "n_blocks must be in 1..4096");
    if (n_blocks > 1 && (offset != 0 || ftp != block_size)) return fail(KNH_ERR_INVALID_ARGUMENT, "multi-block launches process whole blocks");
    if (n_blocks > 1 && voices_host) return fail(KNH_ERR_INVALID_ARGUMENT, "per-voice output is only available for single blocks");
    KNH_HIP(hipSetDevice(device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : own_stream;
    if (h_ev_overflow && __atomic_load_n(h_ev_overflow, __ATOMIC_RELAXED) != 0u) {  // a resolver kernel of an earlier launch dropped a change
      __atomic_store_n(h_ev_overflow, 0u, __ATOMIC_RELAXED);
      warn("Not enough space for scheduled changes in WrPreciseTiming, change ignored");
    }
    const uint32_t fb = static_cast<uint32_t>(offset), fe = static_cast<uint32_t>(offset + ftp);
    // Assemble the launch's state patches block by block, in the order the reference would apply them.
    for (uint32_t b = 0; b < n_blocks; ++b) {
      frame_base = b * static_cast<uint32_t>(block_size);
      if (b > 0 && b < future.size()) {
        for (const Call& c : future[b]) {
          if (c.is_delay) set_delay(c.voice, c.stage, c.param, c.delay);
          else param_apply(c.voice, c.stage, c.param, c.kind, c.f, c.i);
        }
      }
      if (b < qfuture.size()) resolve_qrecs(qfuture[b], fb, fe);
      resolve_queues(fb, fe);
    }
    frame_base = 0;
    if (!qfuture.empty()) {  // records addressed beyond this launch move up (their vectors keep their capacity)
      if (qfuture.size() <= n_blocks) {
        for (auto& q : qfuture) q.clear();
      } else {
        std::rotate(qfuture.begin(), qfuture.begin() + n_blocks, qfuture.end());
        for (size_t k = qfuture.size() - n_blocks; k < qfuture.size(); ++k) qfuture[k].clear();
      }
    }
    // (whether this call goes to a resident launch -- below -- decides what its events may look like)
    if (res_policy < 0) {
      const char* e = std::getenv("KNH_RESIDENT");
      res_policy = e && e[0] == '0' ? 0 : 1;
    }
    const bool res_ok = res_policy == 1 && sync && out_host && !out_device && !voices_host && !stream && n_blocks == 1 && fe > fb && mapped_out &&
                        !accumulate && !timing && !(dev_events && n_recs > 0) && res_possible();
    bool have_events = false;
    int rc = upload_events(s, &have_events, n_blocks, res_ok && res_cooldown == 0 && res_ranges_ok());
    if (rc != KNH_OK) return rc;
    if (!future.empty()) {  // calls addressed beyond this launch move up; those now due for the next
                            // block are applied right away, ahead of anything that arrives later
      if (future.size() <= n_blocks) future.clear();
      else future.erase(future.begin(), future.begin() + n_blocks);
      if (!future.empty()) {
        std::vector<Call> due;
        due.swap(future[0]);
        for (const Call& c : due) {
          if (c.is_delay) set_delay(c.voice, c.stage, c.param, c.delay);
          else param_apply(c.voice, c.stage, c.param, c.kind, c.f, c.i);
        }
      }
    }
    // The call the reference makes -- one whole or partial block into host memory, blocking -- on a resident launch where the
    // bank's kernel form has one (res_possible); anything else first asks a resident kernel (this bank's, or another bank's
    // on this device) to leave: it would be in the way of the launch, and it holds the voices' state in its registers.
    {
      if (res_ok && res_cooldown == 0) {
        rc = res_alloc();
        if (rc != KNH_OK) return rc;
        const int held_list = list_in_use;
        rc = res_call(fb, fe, have_events, out_host, out_flags);
        if (rc != KNH_ERR_UNSUPPORTED_CHAIN) return rc;
        list_in_use = held_list;  // the kernels would not run side by side: this call, and the bank from now on, takes the launch per call
        if (held_list >= 0) list_busy[held_list] = false;
        if (res_n_ranges) {  // the launch per call reads per-voice lists: the call's range events once more, spelled out
          pending_ranges.swap(sent_ranges);
          for (RangeEvent& r : pending_ranges) r.at = 0;
          rc = upload_events(s, &have_events, n_blocks, false);
          if (rc != KNH_OK) return rc;
        }
      }
      if (res_cooldown) --res_cooldown;
      rc = res_leave();
      if (rc != KNH_OK) return rc;
      res_make_room();
    }
    const bool want_voices = voices_host != nullptr || (desc.mix_mode == KNH_MIX_LEFT_FOLD);
    if (desc.mix_mode == KNH_MIX_LEFT_FOLD && n_blocks > 1)
      return fail(KNH_ERR_INVALID_ARGUMENT, "KNH_MIX_LEFT_FOLD processes one block per call");
    if (want_voices) KNH_HIP(ensure_voices());
    const unsigned n_waves = interp ? nv : (nv + 63) / 64;
    if (n_blocks > partials_blocks) {
      KNH_HIP(hipStreamSynchronize(s));
      KNH_HIP(hipFree(d_partials));
      d_partials = nullptr;
      KNH_HIP(hipMalloc(&d_partials, static_cast<size_t>(n_blocks) * fold_planes * n_waves * block_size * sizeof(F)));
      partials_blocks = n_blocks;
    }
    if (!out_device && n_blocks > out_blocks) {
      KNH_HIP(hipStreamSynchronize(s));
      KNH_HIP(hipFree(d_out));
      d_out = nullptr;
      KNH_HIP(hipMalloc(&d_out, static_cast<size_t>(n_blocks) * desc.out_channels * block_size * sizeof(F)));
      KNH_HIP(hipMemsetAsync(d_out, 0, static_cast<size_t>(n_blocks) * desc.out_channels * block_size * sizeof(F), s));
      if (h_out) KNH_HIP(hipHostFree(h_out));
      h_out = nullptr;
      KNH_HIP(hipHostMalloc(&h_out, static_cast<size_t>(n_blocks) * desc.out_channels * block_size * sizeof(F) + 2 * sizeof(uint32_t),
                            hipHostMallocMapped | hipHostMallocCoherent));
      out_blocks = n_blocks;
    }

    if (flags_stream_set && flags_stream != s) KNH_HIP(hipStreamSynchronize(flags_stream));  // that set was cleared in the other stream's order
    flags_stream = s;
    flags_stream_set = true;
    uint32_t* const flags_now = d_flags + 16 * flags_parity;
    uint32_t* const flags_next = d_flags + 16 * (flags_parity ^ 1u);
    flags_parity ^= 1u;
    flags_last = flags_now;
    VoiceKernelArgs<F> a;
    fill_launch_args(a, n_blocks, fb, fe);
    a.input = nullptr;
    a.in_channels = desc.in_channels;
    if (uses_input) {
      if (in_blocks_set != n_blocks) return fail(KNH_ERR_INVALID_ARGUMENT, "a bank with KNH_STAGE_INPUT stages needs knh_bank_set_input for exactly the blocks of this call");
      if (in_device) {
        a.input = in_device;
      } else {
        if (in_host_pending) {
          KNH_HIP(hipMemcpyAsync(d_input, h_input, static_cast<size_t>(n_blocks) * desc.in_channels * block_size * sizeof(F), hipMemcpyHostToDevice, s));
          if (!in_copied) KNH_HIP(hipEventCreateWithFlags(&in_copied, hipEventDisableTiming));
          KNH_HIP(hipEventRecord(in_copied, s));
          in_copy_pending = true;
        }
        in_host_pending = false;
        a.input = d_input;
      }
      in_blocks_set = 0;  // one set_input per process call
    }
    a.ev_start = have_events ? h_ev_start : nullptr;  // pinned host memory, device-visible
    a.events = h_events;
    if (dev_events && n_recs > 0) {  // calls to nodes whose queues the device resolves: the lists are made there, the host's merged in
      int r2 = resolve_on_device(s, n_blocks, fb, fe, have_events, have_events ? h_ev_start[nv] : 0u);
      if (r2 != KNH_OK) return r2;
      a.ev_start = d_out_start2[out_in_use];
      a.events = d_out_events2[out_in_use];
    }
    a.partials = d_partials;
    a.voices_out = want_voices ? d_voices : nullptr;
    a.done_frames = d_done;
    a.flags = flags_now;
    std::pair<hipEvent_t, hipEvent_t>* tp = nullptr;
    if (timing) {
      if (timing_used == timing_pool.size()) {
        if (timing_pool.size() >= 8192) {
          int r = timing_collect();
          if (r != KNH_OK) return r;
        } else {
          hipEvent_t e0, e1;
          KNH_HIP(hipEventCreate(&e0));
          KNH_HIP(hipEventCreate(&e1));
          timing_pool.emplace_back(e0, e1);
        }
      }
      tp = &timing_pool[timing_used++];
      KNH_HIP(hipEventRecord(tp->first, s));
    }
    if (interp) KNH_HIP(launch_interp(a, s));
    else KNH_HIP(launch_voice(a, n_waves, s));
    if (tp) KNH_HIP(hipEventRecord(tp->second, s));
    if (out_in_use >= 0) {  // the resolver may rewrite this set of lists once this kernel has read it
      KNH_HIP(hipEventRecord(lists_free[out_in_use], s));
      lists_busy[out_in_use] = true;
      out_in_use = -1;
    }
    if (have_events && list_in_use >= 0) {
      KNH_HIP(hipEventRecord(list_done[list_in_use], s));
      list_busy[list_in_use] = true;
      list_in_use = -1;
    }

    // A blocking call for host memory: the fold kernel writes into the pinned block itself and tells the host when it is through
    const bool hand_over = sync && out_host && !out_device && !voices_host && mapped_out && fe > fb;
    F* dst = out_device ? static_cast<F*>(out_device) : (hand_over ? h_out : d_out);
    knh_dev::HostDone hd{nullptr, nullptr, nullptr, 0u};
    if (hand_over) {
      done_epoch += 1;
      if (done_epoch == 0) done_epoch = 1;
      hd = knh_dev::HostDone{h_done, d_fold_count, flags_now, done_epoch};
    }
    // A Pan2 chain's row sets are [block][channel][rows][frame] and its output [block][channel][frame]: the fold sees
    // twice as many "blocks" of one channel each.
    const unsigned fold_channels = pan ? 1u : desc.out_channels;
    // (the interpreter's rows ARE the voices' signals: one buffer serves both mix orders and the per-voice debug output)
    if (desc.mix_mode == KNH_MIX_LEFT_FOLD)
      KNH_HIP(launch_fold(false, interp ? d_partials : d_voices, nv, static_cast<unsigned>(block_size), fb, fe, dst, fold_channels, static_cast<unsigned>(block_size), fold_planes, accumulate, flags_next, s, hand_over ? &hd : nullptr));
    else
      KNH_HIP(launch_fold(true, d_partials, n_waves, static_cast<unsigned>(block_size), fb, fe, dst, fold_channels, static_cast<unsigned>(block_size), n_blocks * fold_planes, accumulate, flags_next, s, hand_over ? &hd : nullptr));

    if (!sync) return KNH_OK;
    const size_t blk_elems = desc.out_channels * block_size;
    const size_t out_bytes = static_cast<size_t>(n_blocks) * blk_elems * sizeof(F);
    uint32_t* h_flags = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(h_out) + static_cast<size_t>(out_blocks) * blk_elems * sizeof(F));
    if (hand_over) {
      // Poll the epoch word (the kernel's last store, a system-scope release).  A stream that has finished without the
      // word having moved means the kernel failed: hipStreamQuery says how; it is asked rarely (it is a driver call).
      volatile uint32_t* ep = h_done;
      for (uint64_t spin = 1;; ++spin) {
        if (__atomic_load_n(ep, __ATOMIC_ACQUIRE) == done_epoch) break;
        if ((spin & 0x3FFFu) == 0) {
          const hipError_t q = hipStreamQuery(s);
          if (q == hipSuccess) {
            if (__atomic_load_n(ep, __ATOMIC_ACQUIRE) == done_epoch) break;
            KNH_HIP(hipStreamSynchronize(s));
            if (__atomic_load_n(ep, __ATOMIC_ACQUIRE) != done_epoch) return fail(KNH_ERR_DEVICE, "the fold kernel finished without handing its block over");
            break;
          }
          if (q != hipErrorNotReady) return fail(KNH_ERR_DEVICE, std::string("hipStreamQuery: ") + hipGetErrorString(q));
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
      h_flags[0] = h_done[1];
      h_flags[1] = h_done[2];
    } else {
      if (out_host) KNH_HIP(hipMemcpyAsync(h_out, dst, out_bytes, hipMemcpyDeviceToHost, s));
      KNH_HIP(hipMemcpyAsync(h_flags, flags_now, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      if (voices_host)
        KNH_HIP(hipMemcpyAsync(voices_host, interp ? d_partials : d_voices, static_cast<size_t>(fold_planes) * nv * block_size * sizeof(F), hipMemcpyDeviceToHost, s));
      KNH_HIP(hipStreamSynchronize(s));
    }
    if (out_host) {
      if (n_blocks > 1) {
        std::memcpy(out_host, h_out, out_bytes);
      } else {
        for (uint32_t c = 0; c < desc.out_channels; ++c)
          std::memcpy(static_cast<F*>(out_host) + c * block_size + offset, h_out + c * block_size + offset, ftp * sizeof(F));
      }
    }
    if (out_flags) {
      uint32_t fl = 0;
      if (h_flags[0]) fl |= KNH_FLAG_ANY_DONE;
      bool has_env = false;
      for (const StageInfo& st : stages) has_env = has_env || st.kind == KNH_STAGE_MUL_ENV_ASR || st.kind == KNH_STAGE_MUL_ENV_AR || st.kind == KNH_STAGE_MUL_ENVELOPE ||
                          st.kind == KNH_STAGE_BUFFER_READER;
      if (has_env && h_flags[1] == 0) fl |= KNH_FLAG_ALL_DONE;  // a chain without an envelope never finishes
      *out_flags = fl;
    }
    return KNH_OK;
  }
  uint32_t partials_blocks = 1, out_blocks = 1;
  template <typename FF> hipError_t launch_frame(const VoiceKernelArgs<FF>& a, hipStream_t s) {
    struct { VoiceKernelArgs<FF> a; FF* rows; const uint32_t* sin_slots; uint32_t n_sin; } args{a, reinterpret_cast<FF*>(d_partials), d_sin_slots, n_sin};
    return knh::jit_frame_launch(frame_jit, &args, sizeof(args), (a.n_voices + frame_vpw - 1u) / frame_vpw, s);
  }
  hipError_t launch_interp(const VoiceKernelArgs<float>& a, hipStream_t s) {
    if (frame_jit) return launch_frame<float>(a, s);
    return knh::launch_interp_f32(a, d_prog, static_cast<unsigned>(h_prog.size()), static_cast<unsigned>(n_slots), interp_sigs, interp_out, reinterpret_cast<float*>(d_partials), s);
  }
  hipError_t launch_interp(const VoiceKernelArgs<double>& a, hipStream_t s) {
    if (frame_jit) return launch_frame<double>(a, s);
    return knh::launch_interp_f64(a, d_prog, static_cast<unsigned>(h_prog.size()), static_cast<unsigned>(n_slots), interp_sigs, interp_out, reinterpret_cast<double*>(d_partials), s);
  }
  hipError_t launch_voice(const VoiceKernelArgs<float>& a, unsigned n_waves, hipStream_t s) {
    if (jit) return knh::jit_launch(jit, &a, sizeof(a), n_waves, s);
    if (wide_waves == 4) return wide->f32_w4[desc.allow_fma ? 1 : 0](a, n_waves, s);
    if (wide_waves == 8) return wide->f32_w8[desc.allow_fma ? 1 : 0](a, n_waves, s);
    if (wide_waves == 16) return wide->f32_w16[desc.allow_fma ? 1 : 0](a, n_waves, s);
    if (dag) return dag->f32[desc.allow_fma ? 1 : 0](a, n_waves, s);
    if (pipe) return pipe->f32[desc.allow_fma ? 1 : 0](a, n_waves, s);
    return entry->f32[desc.allow_fma ? 1 : 0](a, n_waves, s);
  }
  hipError_t launch_voice(const VoiceKernelArgs<double>& a, unsigned n_waves, hipStream_t s) {
    if (jit) return knh::jit_launch(jit, &a, sizeof(a), n_waves, s);
    if (wide_waves == 4) return wide->f64_w4[desc.allow_fma ? 1 : 0](a, n_waves, s);
    if (wide_waves == 8) return wide->f64_w8[desc.allow_fma ? 1 : 0](a, n_waves, s);
    if (wide_waves == 16) return wide->f64_w16[desc.allow_fma ? 1 : 0](a, n_waves, s);
    if (pipe) return pipe->f64[desc.allow_fma ? 1 : 0](a, n_waves, s);
    return entry->f64[desc.allow_fma ? 1 : 0](a, n_waves, s);
  }
  static hipError_t launch_fold(bool tree, const float* rows, unsigned n, unsigned len, unsigned fb, unsigned fe, float* out, unsigned ch, unsigned os, unsigned nb, bool acc, uint32_t* zf, hipStream_t s, const knh_dev::HostDone* hd) {
    return knh::launch_fold_f32(tree, rows, n, len, fb, fe, out, ch, os, nb, acc, zf, s, hd);
  }
  static hipError_t launch_fold(bool tree, const double* rows, unsigned n, unsigned len, unsigned fb, unsigned fe, double* out, unsigned ch, unsigned os, unsigned nb, bool acc, uint32_t* zf, hipStream_t s, const knh_dev::HostDone* hd) {
    return knh::launch_fold_f64(tree, rows, n, len, fb, fe, out, ch, os, nb, acc, zf, s, hd);
  }

  int read_done_frames(uint32_t* out) override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    if (!out) return fail(KNH_ERR_INVALID_ARGUMENT, "null output");
    { int rl = res_leave(); if (rl != KNH_OK) return rl; }  // (a resident kernel's stores reach the copy engine when it ends)
    KNH_HIP(hipSetDevice(device));
    KNH_HIP(hipStreamSynchronize(own_stream));
    KNH_HIP(hipMemcpy(out, d_done, static_cast<size_t>(nv) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return KNH_OK;
  }
  int debug_read(uint32_t* out16) override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    { int rl = res_leave(); if (rl != KNH_OK) return rl; }
    KNH_HIP(hipSetDevice(device));
    KNH_HIP(hipDeviceSynchronize());
    KNH_HIP(hipMemcpy(out16, flags_last ? flags_last : d_flags, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return KNH_OK;
  }
  int synchronize() override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    { int rl = res_leave(); if (rl != KNH_OK) return rl; }
    KNH_HIP(hipSetDevice(device));
    KNH_HIP(hipDeviceSynchronize());
    return KNH_OK;
  }
  int timing_collect() {
    KNH_HIP(hipDeviceSynchronize());
    for (size_t k = 0; k < timing_used; ++k) {
      float ms = 0.f;
      KNH_HIP(hipEventElapsedTime(&ms, timing_pool[k].first, timing_pool[k].second));
      timing_ms += ms;
      timing_launches += 1;
    }
    timing_used = 0;
    return KNH_OK;
  }
  int timing_reset(int enable) override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    { int rl = res_leave(); if (rl != KNH_OK) return rl; }  // (kernel time is measured launch by launch: a timed bank launches per call)
    KNH_HIP(hipSetDevice(device));
    int rc = timing_collect();
    if (rc != KNH_OK) return rc;
    timing_ms = 0.0;
    timing_launches = 0;
    timing = enable != 0;
    return KNH_OK;
  }
  int timing_read(double* ms, uint64_t* launches) override {
    if (!initialised) return fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    KNH_HIP(hipSetDevice(device));
    int rc = timing_collect();
    if (rc != KNH_OK) return rc;
    if (ms) *ms = timing_ms;
    if (launches) *launches = timing_launches;
    return KNH_OK;
  }
};

}  // namespace
