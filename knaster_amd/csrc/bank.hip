// bank.hip -- host side of the voice-bank UGen and the extern "C" boundary (include/knaster_hip.h).
//
// The host keeps a shadow of every *parameter-derived* quantity (never of
// audio-evolving state) and turns each UGen::param_apply into device state
// patches.  All transcendental work (tan/pow/sqrt/exp for filter coefficients,
// the f64 phase-increment product) happens here, with the same libm the
// reference's std-backed num-traits would call, so the device only does + - * and
// one table gather per sample.  Citations are file:line in the knaster repo.
//
// There is no CPU processing path: without a gfx950 device every compute entry
// point fails with KNH_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/knaster_hip.h"
#include "jit.hpp"
#include "kernel_registry.hpp"

using knh_dev::Event;
using knh_dev::VoiceKernelArgs;

#include "stage_table.hpp"
#include "bank_base.hpp"
#include "voice_bank.hpp"
#include "chain_signature.hpp"

namespace {

template <typename F>
knh_bank* make_bank(const knh_bank_desc& d, const knh::KernelEntry* entry, const std::string& sig) {
  std::unique_ptr<Bank<F>> b(new Bank<F>());
  b->desc = d;
  b->entry = entry;
  b->signature = sig;
  {  // KNH_PIPELINE=0 forces the single-wave kernel (A/B measurements); KNH_JIT=1 forces run-time fusion
    const char* jit_env = std::getenv("KNH_JIT");
    if (jit_env && jit_env[0] == '1') b->entry = nullptr;
    // KNH_PIPELINE: 0 = single-wave kernel, 1 (default) = linear wave pipeline, 2 = five-role pipeline
    // (voice_dag.hpp; bit-identical, measured 5-8 % slower than level 1 on MI355X, kept for experiments)
    const char* env = std::getenv("KNH_PIPELINE");
    const int level = env && env[0] >= '0' && env[0] <= '2' ? env[0] - '0' : 1;
    b->pipeline_level = level;
    // The 64-sample-tile form with the fold in the last stage group is used where one is built (kernels.hip): measured
    // 3 % faster than the 32-sample form over C3's note cycle, 12 % with every envelope at rest.  KNH_PIPE_BIG=0 keeps the
    // 32-sample tiles and the mixer wavefront (A/B runs).
    // KNH_PIPE_BIG=1: 64-sample tiles with the fold in the last stage group (round 1's form); default: 64-sample tiles, the
    // last group in place and a mixer wavefront on the fourth SIMD.
    const char* big_env = std::getenv("KNH_PIPE_BIG");
    const unsigned forms = big_env && big_env[0] == '0' ? 1u : (big_env && big_env[0] == '1' ? 3u : 7u);
    if (b->entry && level >= 1) b->pipe = knh::find_pipe(sig.c_str(), forms);
    if (b->entry && level >= 2 && d.sample_type == KNH_F32) b->dag = knh::find_dag(sig.c_str());
    // Occupancy regime: the wave pipeline minimises latency when every 64-voice group can have a CU to
    // itself (<= ~1.5 groups per CU); beyond that throughput wins and the groups are packed 4 or 8 to a
    // workgroup (one or two wavefronts per SIMD) sharing one staged sine table.  KNH_WIDE=0/4/8 overrides.
    // Banks of more 64-voice groups than the chip has CUs: two groups per workgroup, each with its own pipeline, sharing the
    // staged sine table -- two wavefronts per SIMD (voice_pipe.hpp, GPW).  One round of it renders 512 groups; KNH_PAIR=0
    // keeps the forms of round 2 (A/B runs), KNH_PAIR=1 uses it for every bank it is built for.
    {
      const unsigned groups = (d.n_voices + 63u) / 64u;
      const char* penv = std::getenv("KNH_PAIR");
      const knh::PipeEntry* pair = b->entry && level >= 1 ? knh::find_pipe(sig.c_str(), 1u << 2 /* PIPE_INPLACE */, 2) : nullptr;
      // Measured (us per 512-frame block, profiles/r03_form_sweep.txt): C3 f32 at 320 / 512 groups 17.8 / 18.5 against 23.5 / 23.8
      // for one group per workgroup and 25.5 / 25.8 for four whole-chain wavefronts per workgroup; from 640 groups on the latter
      // win (26.1-27.1 up to 1 024 groups against 34.7-35.1).  f64 gains nothing from it (38.6-39.1 against 37.0-37.2: an f64
      // wavefront alone already keeps its SIMD's f64 pipe busy).  So: f32 banks of 257-512 groups.
      const bool want = penv ? penv[0] == '1' : (groups > 256u && groups <= 512u && d.sample_type != KNH_F64);
      if (pair && want && !(penv && penv[0] == '0')) { b->pipe = pair; b->pipe_pair = true; }
    }
    b->wide = b->entry && !b->pipe_pair ? knh::find_wide(sig.c_str()) : nullptr;
    if (b->wide) {
      const unsigned groups = (d.n_voices + 63u) / 64u;
      // The pipeline takes ceil(groups / 256 CUs) rounds of its one-group-per-CU time, the 4-group kernel one round of the
      // whole chain's single-wavefront time up to 1 024 groups, the 8-group kernel one of two wavefronts per SIMD up to 2 048.
      // Measured on C3 / C4 (us per 512-frame block, profiles/r03_form_sweep.txt): f32 pipeline 23.5 at 257-512 groups, 35 at
      // 513-768, against 25.5-27.1 flat for four groups per workgroup and 37-40 for eight (79.6 at 4 096 groups, where sixteen
      // take 85.6 -- their tiles are too short for the 32-sample visits the others make); f64 pipeline 39.6-40.1 at 257-512
      // groups against 37.0-38.7 flat for four per workgroup, 68-72 for eight up to 2 048.  So: the pipeline for two rounds in
      // f32, one in f64; four groups per workgroup up to 1 024 groups, eight beyond.
      const unsigned pipe_max = d.sample_type == KNH_F64 ? 256u : 512u;
      int ww = groups <= pipe_max && b->pipe ? 0 : (groups <= 1024 ? 4 : 8);
      if (!b->pipe && groups <= 256) ww = 0;
      // A chain with a delay (rings in HBM, moved as whole lines: voice_stages.hpp RingLines) is bound by memory, and the
      // pipeline has one wavefront per voice group to keep requests in flight: beyond one round of it the whole-chain forms
      // win (D3, us per block, profiles/r04_delay_forms.txt: 20 480 voices 57.8 / 51.8, 65 536 117 / 67.9 for four per
      // workgroup, 131 072 233 / 126 / 118 for eight).  (Rounds 1-3 kept every delay chain on the pipeline: a lane per
      // voice's 16-byte pieces ran at 2.0 TB/s in any form.)
      if (b->pipe && sig.find_first_of("DYZ") != std::string::npos) ww = groups <= 256 ? 0 : (groups <= 1024 ? 4 : 8);
      const char* wenv = std::getenv("KNH_WIDE");
      if (wenv) ww = std::atoi(wenv);
      if (ww == 4 || ww == 8 || ww == 16) b->wide_waves = ww;
    }
  }
  b->nv = d.n_voices;
  int slot = 0, pbase = 0;
  for (uint32_t i = 0; i < d.n_stages; ++i) {
    const KindInfo& k = kKinds[d.stages[i].kind];
    StageInfo s{d.stages[i].kind, d.stages[i].flags, d.stages[i].delayed_changes_per_block, slot, k.n_slots, k.n_params, k.n_ctor, pbase,
                d.stages[i].input, d.stages[i].input2, d.stages[i].ar_param};
    // an SvfFilter with a parameter at audio rate keeps cutoff, q, gain and type on the device too (knh_dev::SvfP)
    if (s.kind == KNH_STAGE_SVF && s.ar_param != 0) s.n_slots = 12;
    b->stages.push_back(s);
    b->ctor.emplace_back(static_cast<size_t>(d.n_voices) * (k.n_ctor > 0 ? k.n_ctor : 0), 0.0);
    slot += s.n_slots;
    pbase += k.n_params;
  }
  b->n_slots = slot;
  b->n_params_total = pbase;
  b->desc.stages = nullptr;  // the caller's array is not retained
  return b.release();
}

}  // namespace

#include "host_shards.hpp"
#include "rank_bank.hpp"

namespace {
// K host threads: the bank is cut into K voice ranges of whole 64-voice groups (fewer when there are fewer groups).
// devices (or null): range k lives on devices[k] (knh_bank_create_multi_device); host_threads == the number of devices then
template <typename F>
knh_bank* make_sharded(const knh_bank_desc& d, const knh::KernelEntry* entry, const std::string& sig, uint32_t host_threads,
                       const int32_t* devices = nullptr) {
  const uint32_t groups = (d.n_voices + 63u) / 64u;
  const uint32_t k = std::min(host_threads, groups);
  const uint32_t per = ((groups + k - 1) / k) * 64u;
  std::unique_ptr<ShardedBank<F>> b(new ShardedBank<F>());
  b->desc = d;
  b->nv = d.n_voices;
  b->per_shard = per;
  for (uint32_t v = 0; v < d.n_voices; v += per) {
    knh_bank_desc dk = d;
    dk.n_voices = std::min(per, d.n_voices - v);
    if (devices) {
      dk.device = devices[b->shard.size()];
      b->shard_device.push_back(dk.device);
    }
    b->base.push_back(v);
    b->shard.emplace_back(make_bank<F>(dk, entry, sig));
  }
  b->base.push_back(d.n_voices);
  b->stages = b->shard[0]->stages;
  b->n_slots = b->shard[0]->n_slots;
  b->n_params_total = b->shard[0]->n_params_total;
  b->desc.stages = nullptr;
  return b.release();
}
}  // namespace

namespace {
template <typename F>
knh_bank* make_rank_bank(const knh_bank_desc& d, const knh::KernelEntry* entry, const std::string& sig, uint32_t rank, uint32_t world,
                                const uint8_t* comm_id, knh_reduce_fn reduce, void* user) {
  std::unique_ptr<RankBank<F>> b(new RankBank<F>());
  b->desc = d;
  b->total = d.n_voices;
  b->rank = rank;
  b->world = world;
  uint32_t first = 0, count = 0;
  shard_voice_range(d.n_voices, rank, world, &first, &count);
  b->lo = first;
  b->hi = first + count;
  if (comm_id) std::memcpy(b->comm_id, comm_id, KNH_COMM_ID_BYTES);
  b->custom = reduce;
  b->custom_user = user;
  // the rank's own voices: an ordinary bank (host-sharded when KNH_HOST_THREADS asks for it) on this rank's device
  knh_bank_desc dl = d;
  dl.n_voices = count ? count : 1;
  const char* env = std::getenv("KNH_HOST_THREADS");
  const long k = env ? std::strtol(env, nullptr, 10) : 0;
  std::unique_ptr<knh_bank> proto(k >= 2 && k <= 64 && dl.n_voices > 64 ? make_sharded<F>(dl, entry, sig, static_cast<uint32_t>(k)) : make_bank<F>(dl, entry, sig));
  b->stages = proto->stages;
  b->n_slots = proto->n_slots;
  b->n_params_total = proto->n_params_total;
  if (count) b->local = std::move(proto);
  b->desc.stages = nullptr;
  return b.release();
}
}  // namespace

// ---------------------------------------------------------------------------
// No C++ exception crosses the C ABI (a Rust caller unwinding through foreign frames is undefined behaviour): every
// entry point that can allocate runs inside guarded(), which turns std::bad_alloc and anything else into a status and a
// message on the handle.  (Setting the message may itself fail for want of memory: then only the status is returned.)
// ---------------------------------------------------------------------------
namespace {
template <typename Fn>
int32_t guarded(knh_bank* bank, Fn&& fn) noexcept {
  const char* what = nullptr;
  int32_t code = KNH_ERR_INTERNAL;
  try {
    return fn();
  } catch (const std::bad_alloc&) {
    what = "out of host memory (std::bad_alloc)";
    code = KNH_ERR_OUT_OF_MEMORY;
  } catch (const std::exception& e) {
    try {
      (bank ? bank->err : g_create_error) = std::string("internal error: ") + e.what();
      return KNH_ERR_INTERNAL;
    } catch (...) {
      what = "internal error";
    }
  } catch (...) {
    what = "internal error (unknown exception)";
  }
  try {
    (bank ? bank->err : g_create_error) = what;
  } catch (...) {
  }
  return code;
}
}  // namespace

// ---------------------------------------------------------------------------
// extern "C" boundary
// ---------------------------------------------------------------------------
extern "C" {

uint32_t knh_abi_version(void) { return KNH_ABI_VERSION; }

int32_t knh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int usable = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ++usable;
  }
  return usable;
}

const char* knh_status_string(int32_t status) {
  switch (status) {
    case KNH_OK: return "ok";
    case KNH_ERR_INVALID_ARGUMENT: return "invalid argument";
    case KNH_ERR_OUT_OF_RANGE: return "index out of range";
    case KNH_ERR_UNSUPPORTED_CHAIN: return "no fused kernel for this chain";
    case KNH_ERR_DEVICE: return "HIP error";
    case KNH_ERR_NOT_INITIALISED: return "bank not initialised";
    case KNH_ERR_NO_DEVICE: return "no gfx950 device";
    case KNH_ERR_WRONG_VALUE_KIND: return "wrong parameter value kind";
    case KNH_ERR_OUT_OF_MEMORY: return "out of host memory";
    case KNH_ERR_INTERNAL: return "internal error";
    default: return "unknown status";
  }
}

const char* knh_last_error(const knh_bank* bank) { return bank ? bank->err.c_str() : g_create_error.c_str(); }

int32_t knh_chain_ugen_count(const knh_stage_desc* stages, uint32_t n_stages) {
  return guarded(nullptr, [&]() -> int32_t {
    if (!stages) return 0;
    int n = 0;
    for (uint32_t i = 0; i < n_stages; ++i)
      if (stages[i].kind < KNH_STAGE_KIND_COUNT) n += kKinds[stages[i].kind].n_nodes;
    return n;
  });
}

static int32_t create_bank(const knh_bank_desc* desc, uint32_t host_threads, knh_bank** out_bank);

// The checks every way of creating a bank shares; on success *sig is the chain's device signature.
static int32_t check_desc(const knh_bank_desc* desc, knh_bank** out_bank, std::string* sig) {
  return guarded(nullptr, [&]() -> int32_t {
    if (out_bank) *out_bank = nullptr;
    if (!desc || !out_bank) { g_create_error = "null argument"; return KNH_ERR_INVALID_ARGUMENT; }
    if (desc->abi_version != KNH_ABI_VERSION) { g_create_error = "ABI version mismatch"; return KNH_ERR_INVALID_ARGUMENT; }
    if (desc->n_voices == 0 || !desc->stages) { g_create_error = "n_voices must be > 0 and stages non-null"; return KNH_ERR_INVALID_ARGUMENT; }
    if (desc->sample_type > KNH_F64) { g_create_error = "unknown sample type"; return KNH_ERR_INVALID_ARGUMENT; }
    if (desc->out_channels < 1 || desc->out_channels > 2) { g_create_error = "out_channels must be 1 or 2"; return KNH_ERR_INVALID_ARGUMENT; }
    if (desc->mix_mode > KNH_MIX_LEFT_FOLD) { g_create_error = "unknown mix mode"; return KNH_ERR_INVALID_ARGUMENT; }
    if (desc->in_channels > 16) { g_create_error = "in_channels must be at most 16"; return KNH_ERR_INVALID_ARGUMENT; }
    std::string why;
    int rc = build_signature(desc->stages, desc->n_stages, sig, &why);
    if (rc != KNH_OK) { g_create_error = why; return rc; }
    if (desc->stages[desc->n_stages - 1].kind == KNH_STAGE_PAN2 && desc->out_channels != 2) { g_create_error = "a chain ending in Pan2 has two output channels (out_channels = 2)"; return KNH_ERR_INVALID_ARGUMENT; }
    return KNH_OK;
  });
}

int32_t knh_bank_create_multi_device(const knh_bank_desc* desc, const int32_t* devices, uint32_t n_devices, knh_bank** out_bank) {
  return guarded(nullptr, [&]() -> int32_t {
    std::string sig;
    int rc = check_desc(desc, out_bank, &sig);
    if (rc != KNH_OK) return rc;
    if (!devices || n_devices == 0 || n_devices > 64) { g_create_error = "devices: 1 to 64 device ordinals"; return KNH_ERR_INVALID_ARGUMENT; }
    if (desc->mix_mode != KNH_MIX_TREE) { g_create_error = "a bank sharded over several GPUs mixes with KNH_MIX_TREE (the sum over GPUs re-associates)"; return KNH_ERR_INVALID_ARGUMENT; }
    int visible = 0;
    (void)hipGetDeviceCount(&visible);
    for (uint32_t k = 0; k < n_devices; ++k)
      if (devices[k] < 0 || devices[k] >= std::max(visible, 1)) { g_create_error = "devices: ordinal out of range"; return KNH_ERR_INVALID_ARGUMENT; }
    const knh::KernelEntry* entry = knh::find_kernel(sig.c_str());
    *out_bank = desc->sample_type == KNH_F64 ? make_sharded<double>(*desc, entry, sig, n_devices, devices) : make_sharded<float>(*desc, entry, sig, n_devices, devices);
    return KNH_OK;
  });
}

int32_t knh_shard_voice_range(uint32_t n_voices, uint32_t rank, uint32_t world, uint32_t* first, uint32_t* count) {
  return guarded(nullptr, [&]() -> int32_t {
    if (world == 0 || rank >= world || !first || !count) return KNH_ERR_INVALID_ARGUMENT;
    shard_voice_range(n_voices, rank, world, first, count);
    return KNH_OK;
  });
}

static int32_t create_rank_bank(const knh_bank_desc* desc, uint32_t rank, uint32_t world, const uint8_t* comm_id, knh_reduce_fn reduce, void* user,
                                knh_bank** out_bank) {
  return guarded(nullptr, [&]() -> int32_t {
    std::string sig;
    int rc = check_desc(desc, out_bank, &sig);
    if (rc != KNH_OK) return rc;
    if (world == 0 || rank >= world) { g_create_error = "rank must be below world"; return KNH_ERR_INVALID_ARGUMENT; }
    if (world > 1 && !comm_id && !reduce) { g_create_error = "more than one rank needs a communicator id (knh_comm_unique_id) or a reduce function"; return KNH_ERR_INVALID_ARGUMENT; }
    if (desc->mix_mode != KNH_MIX_TREE) { g_create_error = "a bank sharded over several GPUs mixes with KNH_MIX_TREE (the sum over GPUs re-associates)"; return KNH_ERR_INVALID_ARGUMENT; }
    const knh::KernelEntry* entry = knh::find_kernel(sig.c_str());
    *out_bank = desc->sample_type == KNH_F64 ? make_rank_bank<double>(*desc, entry, sig, rank, world, comm_id, reduce, user)
                                            : make_rank_bank<float>(*desc, entry, sig, rank, world, comm_id, reduce, user);
    return KNH_OK;
  });
}
int32_t knh_bank_create_rank(const knh_bank_desc* desc, uint32_t rank, uint32_t world, const uint8_t* comm_id, knh_bank** out_bank) {
  return guarded(nullptr, [&]() -> int32_t {
    return create_rank_bank(desc, rank, world, comm_id, nullptr, nullptr, out_bank);
  });
}
int32_t knh_bank_create_rank_custom(const knh_bank_desc* desc, uint32_t rank, uint32_t world, knh_reduce_fn reduce, void* user, knh_bank** out_bank) {
  return guarded(nullptr, [&]() -> int32_t {
    if (world > 1 && !reduce) { g_create_error = "null reduce function"; if (out_bank) *out_bank = nullptr; return KNH_ERR_INVALID_ARGUMENT; }
    return create_rank_bank(desc, rank, world, nullptr, reduce, user, out_bank);
  });
}
uint32_t knh_bank_ranks(const knh_bank* bank) { return bank ? bank->ranks() : 0; }

static int32_t create_bank(const knh_bank_desc* desc, uint32_t host_threads, knh_bank** out_bank) {
  return guarded(nullptr, [&]() -> int32_t {
    std::string sig;
    int rc = check_desc(desc, out_bank, &sig);
    if (rc != KNH_OK) return rc;
    if (host_threads > 64) { g_create_error = "host_threads must be at most 64"; return KNH_ERR_INVALID_ARGUMENT; }
    // a chain without a pre-built kernel is fused at knh_bank_init time (hiprtc); entry == nullptr marks it
    const knh::KernelEntry* entry = knh::find_kernel(sig.c_str());
    // the reference's exact mix order (KNH_MIX_LEFT_FOLD) and banks of a single voice group keep one range
    if (host_threads >= 2 && desc->mix_mode == KNH_MIX_TREE && desc->n_voices > 64)
      *out_bank = desc->sample_type == KNH_F64 ? make_sharded<double>(*desc, entry, sig, host_threads) : make_sharded<float>(*desc, entry, sig, host_threads);
    else
      *out_bank = desc->sample_type == KNH_F64 ? make_bank<double>(*desc, entry, sig) : make_bank<float>(*desc, entry, sig);
    return KNH_OK;
  });
}

int32_t knh_bank_create(const knh_bank_desc* desc, knh_bank** out_bank) {
  return guarded(nullptr, [&]() -> int32_t {
    // KNH_HOST_THREADS=K: every bank created through this entry point gets K host threads (A/B runs of existing programs)
    const char* env = std::getenv("KNH_HOST_THREADS");
    const long k = env ? std::strtol(env, nullptr, 10) : 0;
    return create_bank(desc, k >= 2 && k <= 64 ? static_cast<uint32_t>(k) : 0u, out_bank);
  });
}

int32_t knh_bank_create_sharded(const knh_bank_desc* desc, uint32_t host_threads, knh_bank** out_bank) {
  return guarded(nullptr, [&]() -> int32_t {
    return create_bank(desc, host_threads, out_bank);
  });
}

void knh_bank_destroy(knh_bank* bank) {
  try {
    if (bank && bank->pipe_stream) {  // launches begun and never fetched still hold the bank's buffers
      (void)hipSetDevice(bank->device);
      (void)hipStreamSynchronize(bank->pipe_stream);
    }
    delete bank;
  } catch (...) {
  }
}

int32_t knh_bank_set_ctor_args(knh_bank* bank, uint32_t stage, uint32_t first_voice, uint32_t count, const double* args, uint32_t n_args) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    return bank->set_ctor(stage, first_voice, count, args, n_args);
  });
}
int32_t knh_bank_set_buffer(knh_bank* bank, uint32_t stage, const void* samples, size_t n_frames, double buffer_sample_rate) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    return bank->set_buffer(stage, samples, n_frames, buffer_sample_rate);
  });
}
int32_t knh_bank_init(knh_bank* bank, uint32_t sample_rate, size_t block_size) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    const int rc = bank->init(sample_rate, block_size);
    if (rc == KNH_OK) bank->channel_block.assign(static_cast<size_t>(bank->desc.out_channels) * block_size * (bank->desc.sample_type == KNH_F64 ? 8 : 4), 0);
    return rc;
  });
}
uint16_t knh_bank_inputs(const knh_bank* bank) { return bank ? static_cast<uint16_t>(bank->desc.in_channels) : 0; }
int32_t knh_bank_set_input(knh_bank* bank, uint32_t n_blocks, const void* in) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    if (!in) return bank->fail(KNH_ERR_INVALID_ARGUMENT, "null input");
    return bank->set_input(n_blocks, in, nullptr);
  });
}
int32_t knh_bank_set_input_device(knh_bank* bank, uint32_t n_blocks, const void* in_device) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    if (!in_device) return bank->fail(KNH_ERR_INVALID_ARGUMENT, "null input");
    return bank->set_input(n_blocks, nullptr, in_device);
  });
}
uint16_t knh_bank_outputs(const knh_bank* bank) { return bank ? static_cast<uint16_t>(bank->desc.out_channels) : 0; }
uint16_t knh_bank_stage_parameters(const knh_bank* bank, uint32_t stage) {
  if (!bank || stage >= bank->stages.size()) return 0;
  return static_cast<uint16_t>(bank->stages[stage].n_params);
}
const char* knh_bank_stage_param_description(const knh_bank* bank, uint32_t stage, uint32_t param) {
  if (!bank || stage >= bank->stages.size() || param >= static_cast<uint32_t>(bank->stages[stage].n_params)) return nullptr;
  return kKinds[bank->stages[stage].kind].params[param];
}
int32_t knh_bank_param_apply(knh_bank* bank, uint32_t voice, uint32_t stage, uint32_t param, uint32_t kind, double fvalue, int64_t ivalue) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    return bank->param_apply(voice, stage, param, kind, fvalue, ivalue);
  });
}
int32_t knh_bank_set_delay_within_block_for_param(knh_bank* bank, uint32_t voice, uint32_t stage, uint32_t param, uint16_t delay) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    return bank->set_delay(voice, stage, param, delay);
  });
}
int32_t knh_bank_param_apply_many(knh_bank* bank, size_t count, const uint32_t* voices, const uint32_t* stages, const uint32_t* params,
                                  const uint32_t* kinds, const double* fvalues, const int64_t* ivalues, const uint16_t* delays) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    if (count && (!voices || !stages || !params || !kinds)) return bank->fail(KNH_ERR_INVALID_ARGUMENT, "null array");
    return bank->apply_many(0, count, voices, stages, params, kinds, fvalues, ivalues, delays);
  });
}
int32_t knh_bank_param_apply_range(knh_bank* bank, uint32_t voice_begin, uint32_t voice_end, uint32_t stage, uint32_t param, uint32_t kind,
                                   double fvalue, int64_t ivalue) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    if (voice_end < voice_begin) return bank->fail(KNH_ERR_INVALID_ARGUMENT, "voice_end < voice_begin");
    return bank->apply_range(voice_begin, voice_end, stage, param, kind, fvalue, ivalue);
  });
}
int32_t knh_bank_process_block(knh_bank* bank, size_t frames_to_process, size_t block_start_offset, uint64_t frame_clock, void* out, uint32_t* out_flags) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    if (!out) return bank->fail(KNH_ERR_INVALID_ARGUMENT, "null output block");
    return bank->process(1, frames_to_process, block_start_offset, frame_clock, out, nullptr, nullptr, out_flags, nullptr, true);
  });
}
int32_t knh_bank_process_block_channels(knh_bank* bank, size_t frames_to_process, size_t block_start_offset, uint64_t frame_clock, void* const* out_channels, uint32_t* out_flags) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    if (!out_channels) return bank->fail(KNH_ERR_INVALID_ARGUMENT, "null output channels");
    for (uint32_t c = 0; c < bank->desc.out_channels; ++c)
      if (!out_channels[c]) return bank->fail(KNH_ERR_INVALID_ARGUMENT, "null output channel");
    if (!bank->initialised) return bank->fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    // the bank renders frames [offset, offset + n) of its own [channels][block_size] block; each channel's piece then goes
    // to the slice the caller gave for it, which starts at the first frame of this call
    unsigned char* blk = bank->channel_block.data();
    const int rc = bank->process(1, frames_to_process, block_start_offset, frame_clock, blk, nullptr, nullptr, out_flags, nullptr, true);
    if (rc != KNH_OK) return rc;
    const size_t word = bank->desc.sample_type == KNH_F64 ? 8 : 4;
    for (uint32_t c = 0; c < bank->desc.out_channels; ++c)
      std::memcpy(out_channels[c], blk + (static_cast<size_t>(c) * bank->block_size + block_start_offset) * word, frames_to_process * word);
    return KNH_OK;
  });
}
int32_t knh_bank_resident_stats(knh_bank* bank, uint64_t* calls, uint64_t* launches) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    bank->resident_stats(calls, launches);
    return KNH_OK;
  });
}
int32_t knh_bank_resident_trace(knh_bank* bank, uint64_t* ticks5) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank || !ticks5) return KNH_ERR_INVALID_ARGUMENT;
    bank->resident_trace(ticks5);
    return KNH_OK;
  });
}
int32_t knh_bank_process_block_device(knh_bank* bank, size_t frames_to_process, size_t block_start_offset, uint64_t frame_clock, void* out_device, void* hip_stream) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    return bank->process(1, frames_to_process, block_start_offset, frame_clock, nullptr, out_device, nullptr, nullptr, hip_stream, false);
  });
}
int32_t knh_bank_process_block_voices(knh_bank* bank, size_t frames_to_process, size_t block_start_offset, uint64_t frame_clock, void* out, void* voices_out, uint32_t* out_flags) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    if (!voices_out) return bank->fail(KNH_ERR_INVALID_ARGUMENT, "null voices_out");
    return bank->process(1, frames_to_process, block_start_offset, frame_clock, out, nullptr, voices_out, out_flags, nullptr, true);
  });
}
int32_t knh_bank_process_blocks(knh_bank* bank, uint32_t n_blocks, uint64_t frame_clock, void* out, uint32_t* out_flags) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    if (!out) return bank->fail(KNH_ERR_INVALID_ARGUMENT, "null output");
    return bank->process(n_blocks, bank->block_size, 0, frame_clock, out, nullptr, nullptr, out_flags, nullptr, true);
  });
}
int32_t knh_bank_process_blocks_device(knh_bank* bank, uint32_t n_blocks, uint64_t frame_clock, void* out_device, void* hip_stream) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    return bank->process(n_blocks, bank->block_size, 0, frame_clock, nullptr, out_device, nullptr, nullptr, hip_stream, false);
  });
}
int32_t knh_bank_process_blocks_device_add(knh_bank* bank, uint32_t n_blocks, uint64_t frame_clock, void* out_device, void* hip_stream) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    return bank->process(n_blocks, bank->block_size, 0, frame_clock, nullptr, out_device, nullptr, nullptr, hip_stream, false, true);
  });
}
int32_t knh_bank_process_blocks_begin(knh_bank* bank, uint32_t n_blocks, uint64_t frame_clock) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    if (!bank->initialised) return bank->fail(KNH_ERR_NOT_INITIALISED, "bank not initialised");
    if (bank->pipe_count == 2) return bank->fail(KNH_ERR_INVALID_ARGUMENT, "two launches are outstanding: fetch one with knh_bank_process_blocks_end first");
    if (n_blocks == 0 || n_blocks > 4096) return bank->fail(KNH_ERR_INVALID_ARGUMENT, "n_blocks must be in 1..4096");
    auto hip = [&](hipError_t e, const char* what) { return e == hipSuccess ? KNH_OK : bank->fail(KNH_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); };
    int rc = hip(hipSetDevice(bank->device), "hipSetDevice");
    if (rc != KNH_OK) return rc;
    if (!bank->pipe_stream && (rc = hip(hipStreamCreateWithFlags(&bank->pipe_stream, hipStreamNonBlocking), "hipStreamCreate")) != KNH_OK) return rc;
    knh_bank::PipeSlot& p = bank->pipe[(bank->pipe_head + bank->pipe_count) % 2];
    const size_t bytes = static_cast<size_t>(n_blocks) * bank->desc.out_channels * bank->block_size * (bank->desc.sample_type == KNH_F64 ? 8 : 4);
    if (bytes > p.cap) {
      if (p.dev) (void)hipFree(p.dev);
      if (p.host) (void)hipHostFree(p.host);
      p.dev = p.host = nullptr;
      p.cap = 0;
      if ((rc = hip(hipMalloc(&p.dev, bytes), "hipMalloc")) != KNH_OK) return rc;
      if ((rc = hip(hipMemset(p.dev, 0, bytes), "hipMemset")) != KNH_OK) return rc;
      if ((rc = hip(hipHostMalloc(&p.host, bytes), "hipHostMalloc")) != KNH_OK) return rc;
      p.cap = bytes;
    }
    if (!p.done && (rc = hip(hipEventCreateWithFlags(&p.done, hipEventDisableTiming), "hipEventCreate")) != KNH_OK) return rc;
    p.bytes = bytes;
    rc = bank->process(n_blocks, bank->block_size, 0, frame_clock, nullptr, p.dev, nullptr, nullptr, bank->pipe_stream, false);
    if (rc != KNH_OK) return rc;
    if ((rc = bank->order_after_collective(bank->pipe_stream)) != KNH_OK) return rc;
    if ((rc = hip(hipSetDevice(bank->device), "hipSetDevice")) != KNH_OK) return rc;
    if ((rc = hip(hipMemcpyAsync(p.host, p.dev, bytes, hipMemcpyDeviceToHost, bank->pipe_stream), "hipMemcpyAsync")) != KNH_OK) return rc;
    if ((rc = hip(hipEventRecord(p.done, bank->pipe_stream), "hipEventRecord")) != KNH_OK) return rc;
    bank->pipe_count += 1;
    return KNH_OK;
  });
}
int32_t knh_bank_process_blocks_end(knh_bank* bank, void* out) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    if (!out) return bank->fail(KNH_ERR_INVALID_ARGUMENT, "null output");
    if (bank->pipe_count == 0) return bank->fail(KNH_ERR_INVALID_ARGUMENT, "no launch is outstanding (knh_bank_process_blocks_begin)");
    knh_bank::PipeSlot& p = bank->pipe[bank->pipe_head];
    if (hipSetDevice(bank->device) != hipSuccess || hipEventSynchronize(p.done) != hipSuccess) return bank->fail(KNH_ERR_DEVICE, "waiting for the launch failed");
    std::memcpy(out, p.host, p.bytes);
    bank->pipe_head = (bank->pipe_head + 1) % 2;
    bank->pipe_count -= 1;
    return KNH_OK;
  });
}
void* knh_device_malloc(size_t bytes, int32_t device) {
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return nullptr;
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  if (hipMemset(p, 0, bytes) != hipSuccess) { (void)hipFree(p); return nullptr; }
  return p;
}
void knh_device_free(void* p) {
  if (p) (void)hipFree(p);
}
int32_t knh_device_read(void* dst_host, const void* src_device, size_t bytes, void* hip_stream) {
  return guarded(nullptr, [&]() -> int32_t {
    // the banks enqueue on their own non-blocking streams unless told otherwise: wait for the device first
    if (hipDeviceSynchronize() != hipSuccess) return KNH_ERR_DEVICE;
    (void)hip_stream;
    return hipMemcpy(dst_host, src_device, bytes, hipMemcpyDeviceToHost) == hipSuccess ? KNH_OK : KNH_ERR_DEVICE;
  });
}
int32_t knh_bank_param_apply_many_at(knh_bank* bank, uint32_t block_offset, size_t count, const uint32_t* voices, const uint32_t* stages,
                                     const uint32_t* params, const uint32_t* kinds, const double* fvalues, const int64_t* ivalues,
                                     const uint16_t* delays) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    if (count && (!voices || !stages || !params || !kinds)) return bank->fail(KNH_ERR_INVALID_ARGUMENT, "null array");
    return bank->apply_many(block_offset, count, voices, stages, params, kinds, fvalues, ivalues, delays);
  });
}
int32_t knh_bank_read_done_frames(knh_bank* bank, uint32_t* done_frames) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    return bank->read_done_frames(done_frames);
  });
}
void knh_jit_stats(uint64_t* memory_hits, uint64_t* disk_hits, uint64_t* helper_compiles, uint64_t* in_process_compiles) {
  const knh::JitStats st = knh::jit_stats();
  if (memory_hits) *memory_hits = st.memory_hits;
  if (disk_hits) *disk_hits = st.disk_hits;
  if (helper_compiles) *helper_compiles = st.helper_runs;
  if (in_process_compiles) *in_process_compiles = st.in_process;
}
const char* knh_bank_debug_signature(const knh_bank* bank) { return bank ? bank->debug_signature() : ""; }
int32_t knh_bank_debug_words(knh_bank* bank, uint32_t* out16) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank || !out16) return KNH_ERR_INVALID_ARGUMENT;
    return bank->debug_read(out16);
  });
}
int32_t knh_bank_synchronize(knh_bank* bank) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    return bank->synchronize();
  });
}
int32_t knh_bank_timing_reset(knh_bank* bank, int32_t enable) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    return bank->timing_reset(enable);
  });
}
int32_t knh_bank_timing_read(knh_bank* bank, double* kernel_ms, uint64_t* launches) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    return bank->timing_read(kernel_ms, launches);
  });
}
int32_t knh_bank_collective_timing_read(knh_bank* bank, double* reduce_ms, uint64_t* reduces) {
  return guarded(bank, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    return bank->collective_timing_read(reduce_ms, reduces);
  });
}
int32_t knh_bank_algorithmic_bytes_per_voice_block(const knh_bank* bank, uint32_t* read_bytes, uint32_t* write_bytes) {
  return guarded(nullptr, [&]() -> int32_t {
    if (!bank) return KNH_ERR_INVALID_ARGUMENT;
    // every slot is read once; mutable slots are written once (masks mirror voice_stages.hpp kMutableMask)
    uint32_t r = 0, w = 0;
    const uint32_t word = bank->desc.sample_type == KNH_F64 ? 8 : 4;
    for (const StageInfo& s : bank->stages) {
      r += word * s.n_slots;
      switch (s.kind) {
        case KNH_STAGE_SIN_WT: w += word * ((s.flags & KNH_STAGE_FLAG_AR_FREQ) ? 2 : 1); break;
        case KNH_STAGE_SIN_NUMERIC: w += word; break;
        case KNH_STAGE_SVF: w += word * 2; break;
        case KNH_STAGE_ONEPOLE_LPF: case KNH_STAGE_ONEPOLE_HPF: w += word; break;
        case KNH_STAGE_MUL_ENV_ASR: case KNH_STAGE_MUL_ENV_AR: w += word * 3; break;
        case KNH_STAGE_MUL_ENVELOPE: w += word * 6; break;
        case KNH_STAGE_SAMPLE_DELAY: w += word; break;
        case KNH_STAGE_PHASOR: w += word * 2; break;
        case KNH_STAGE_WHITE_NOISE: w += word * 2; break;
        case KNH_STAGE_PINK_NOISE: w += word * 14; break;
        case KNH_STAGE_BROWN_NOISE: w += word * 3; break;
        case KNH_STAGE_RANDOM_LIN: w += word * 5; break;
        case KNH_STAGE_POLYBLEP: w += word; break;
        case KNH_STAGE_BUFFER_READER: w += word * 3; break;  // + two Buffer samples read per frame
        case KNH_STAGE_ALLPASS_DELAY: case KNH_STAGE_ALLPASS_FB_DELAY: w += word * 4; break;  // + one sample read and one written per frame (ring in HBM)  // + one sample read and one written per frame (ring in HBM)
        default: break;
      }
    }
    if (read_bytes) *read_bytes = r;
    if (write_bytes) *write_bytes = w;
    return KNH_OK;
  });
}

}  // extern "C"
