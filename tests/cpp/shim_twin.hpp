// shim_twin.hpp -- the C++ twin of bindings/rust/knaster_hip/src/lib.rs.
//
// The Rust shim (`impl UGen for GpuVoiceBank<F, I>`) cannot be compiled in this image (no rustc).  This class makes the
// SAME C-ABI calls in the SAME order for every method of the shim, method by method and under the same names, so that the
// call pattern the reference drives a bank with -- Task::run once per block (knaster_graph/src/task.rs:25-31) under
// AudioProcessor::run_without_inputs (processor.rs:142-179), parameter events before the task loop
// (graph_gen.rs:110-166 then :196-200), set_ar_param_buffer when a schedule is taken (task.rs:113-120) -- is exercised
// and timed from compiled code.  tests/test_abi.py::test_shim_twin_makes_the_shims_calls holds the two files to the same
// list of entry points per method.  Test infrastructure: nothing in the product includes it.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/knaster_hip.h"

namespace shim_twin {

constexpr size_t MAX_PARAMS = 6;  // lib.rs: MAX_PARAMS

struct AudioCtx {  // knaster_core/src/ugen.rs:8-49 (what the shim reads of it)
  uint32_t sample_rate = 48000;
  size_t block_size = 64;
  size_t block_start_offset = 0, frames_to_process = 64;
  uint64_t frame_clock = 0;
};
struct UGenFlags {  // ugen.rs:121-219: done + frame
  bool done = false;
  uint32_t done_frame = 0;
  void mark_done(uint32_t frame) { done = true; done_frame = frame; }
};
enum class Value { Float, Trigger, Integer, Bool };  // ParameterValue, parameters/types.rs:25-36

inline knh_stage_desc stage(uint16_t kind) { return knh_stage_desc{kind, 0, 0, 0, 0, 0}; }  // lib.rs: stage()

template <typename F, unsigned INPUTS = 0>
class GpuVoiceBank {
 public:
  // lib.rs: GpuVoiceBank::with_options -> knh_bank_create_sharded, knh_bank_set_ctor_args per stage
  GpuVoiceBank(const std::vector<knh_stage_desc>& stages, uint32_t n_voices, const std::vector<std::vector<double>>& ctor,
               uint32_t host_threads = 0, size_t ar_slots = 0)
      : n_stages_(stages.size()), n_voices_(n_voices), n_ar_(ar_slots), ar_bufs_(ar_slots, nullptr) {
    if (INPUTS + ar_slots > 16) throw std::runtime_error("a bank node has at most 16 input channels");
    knh_bank_desc desc{};
    desc.abi_version = KNH_ABI_VERSION;
    desc.n_voices = n_voices;
    desc.sample_type = sizeof(F) == 8 ? KNH_F64 : KNH_F32;
    desc.n_stages = static_cast<uint32_t>(stages.size());
    desc.stages = stages.data();
    desc.out_channels = 2;
    desc.mix_mode = KNH_MIX_TREE;
    desc.device = -1;
    desc.allow_fma = 0;
    desc.in_channels = static_cast<uint32_t>(INPUTS + ar_slots);
    if (knh_bank_create_sharded(&desc, host_threads, &h_) != KNH_OK) throw std::runtime_error(knh_last_error(nullptr));
    for (size_t s = 0; s < ctor.size(); ++s) {
      const uint32_t n_args = static_cast<uint32_t>(ctor[s].size() / n_voices);
      if (n_args > 0 && knh_bank_set_ctor_args(h_, static_cast<uint32_t>(s), 0, n_voices, ctor[s].data(), n_args) != KNH_OK) {
        std::string e = knh_last_error(h_);
        knh_bank_destroy(h_);
        throw std::runtime_error(e);
      }
    }
  }
  ~GpuVoiceBank() { knh_bank_destroy(h_); }  // lib.rs: Drop
  GpuVoiceBank(const GpuVoiceBank&) = delete;
  GpuVoiceBank& operator=(const GpuVoiceBank&) = delete;

  // lib.rs: GpuVoiceBank::index
  size_t index(uint32_t voice, size_t stage_i, const char* name) const {
    const size_t n = knh_bank_stage_parameters(h_, static_cast<uint32_t>(stage_i));
    for (size_t p = 0; p < n; ++p) {
      const char* d = knh_bank_stage_param_description(h_, static_cast<uint32_t>(stage_i), static_cast<uint32_t>(p));
      if (d && std::strcmp(d, name) == 0) return (voice * n_stages_ + stage_i) * MAX_PARAMS + p;
    }
    throw std::runtime_error(std::string("DescriptionNotFound(") + name + ")");
  }
  const std::string& init_error() const { return init_error_; }
  knh_bank* raw() const { return h_; }

  // ---- impl UGen ---------------------------------------------------------------------------------------
  // lib.rs: UGen::init -> knh_bank_init; the input pack is allocated here (control thread)
  void init(uint32_t sample_rate, size_t block_size) {
    if (knh_bank_init(h_, sample_rate, block_size) != KNH_OK) init_error_ = knh_last_error(h_);
    block_size_ = block_size;
    in_pack_.assign((INPUTS + n_ar_) * block_size, F(0));
  }
  // lib.rs: UGen::process_block -> [knh_bank_set_input] knh_bank_process_block
  //   input: INPUTS channel pointers (the reference's RawAggregateBlockRead: one pointer per channel, block.rs:158-197),
  //   output: contiguous channel-major [2][block_size] (RawContiguousBlock, block.rs:19-78)
  int32_t process_block(AudioCtx& ctx, UGenFlags& flags, const F* const* input, F* output) {
    if (!init_error_.empty()) {
      std::memset(output, 0, 2 * block_size_ * sizeof(F));
      return KNH_ERR_NOT_INITIALISED;
    }
    const size_t n_in = INPUTS + n_ar_;
    if (n_in > 0) {
      const size_t bs = block_size_, off = ctx.block_start_offset, ftp = ctx.frames_to_process;
      for (size_t ch = 0; ch < INPUTS; ++ch) std::memcpy(&in_pack_[ch * bs + off], input[ch], std::min(ftp, bs - off) * sizeof(F));
      for (size_t k = 0; k < n_ar_; ++k) {
        F* dst = &in_pack_[(INPUTS + k) * bs];
        if (!ar_bufs_[k]) std::memset(dst, 0, bs * sizeof(F));
        else std::memcpy(dst, ar_bufs_[k], bs * sizeof(F));
      }
      (void)knh_bank_set_input(h_, 1, in_pack_.data());
    }
    uint32_t f = 0;
    const int32_t rc = knh_bank_process_block(h_, ctx.frames_to_process, ctx.block_start_offset, ctx.frame_clock, output, &f);
    if (f & KNH_FLAG_ALL_DONE) flags.mark_done(0);
    return rc;
  }
  // lib.rs: UGen::param_apply -> knh_bank_param_apply
  int32_t param_apply(AudioCtx&, size_t index, Value kind, double f = 0.0, int64_t i = 0) {
    const size_t param = index % MAX_PARAMS, rest = index / MAX_PARAMS;
    const uint32_t k = kind == Value::Float ? KNH_VALUE_FLOAT : kind == Value::Trigger ? KNH_VALUE_TRIGGER : kind == Value::Integer ? KNH_VALUE_INTEGER : KNH_VALUE_BOOL;
    return knh_bank_param_apply(h_, static_cast<uint32_t>(rest / n_stages_), static_cast<uint32_t>(rest % n_stages_), static_cast<uint32_t>(param), k, f, i);
  }
  // lib.rs: UGen::set_ar_param_buffer: the slot's pointer is kept; process_block packs its samples
  void set_ar_param_buffer(AudioCtx&, size_t index, const F* buffer) {
    if (index < n_ar_) ar_bufs_[index] = buffer;
  }
  // lib.rs: UGen::set_delay_within_block_for_param -> knh_bank_set_delay_within_block_for_param
  int32_t set_delay_within_block_for_param(AudioCtx&, size_t index, uint16_t delay) {
    const size_t param = index % MAX_PARAMS, rest = index / MAX_PARAMS;
    return knh_bank_set_delay_within_block_for_param(h_, static_cast<uint32_t>(rest / n_stages_), static_cast<uint32_t>(rest % n_stages_),
                                                     static_cast<uint32_t>(param), delay);
  }

 private:
  knh_bank* h_ = nullptr;
  size_t n_stages_;
  uint32_t n_voices_;
  size_t n_ar_;
  std::vector<const F*> ar_bufs_;
  std::vector<F> in_pack_;
  size_t block_size_ = 0;
  std::string init_error_;
};

}  // namespace shim_twin
