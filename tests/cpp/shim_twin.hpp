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

// ---- the reference's block views, as the shim sees them -------------------------------------------------------------
// What `process_block` is handed is a `Block` / `BlockRead` (knaster_primitives/src/block.rs:33-263): per channel a slice,
// `channel_as_slice[_mut](ch)`, and nothing else -- no base pointer, no promise that channel c + 1 follows channel c.
template <typename F> struct Slice { F* ptr; size_t len; };
// RawContiguousBlock (knaster_graph/src/block.rs:19-78): the graph's output buffer of a node, channel-major and contiguous
template <typename F> struct ContiguousBlock {
  F* base; size_t n_channels, bs;
  Slice<F> channel_as_slice_mut(size_t ch) { return {base + ch * bs, bs}; }
  size_t channels() const { return n_channels; }
  size_t block_size() const { return bs; }
  // Block::partial_mut (block.rs:190-196)
  template <typename Self = ContiguousBlock> auto partial_mut(size_t start_offset, size_t length);
};
// PartialBlockMut (block.rs:307-339): `&mut block.channel_as_slice_mut(ch)[start_offset .. start_offset + length]` -- its
// slices START AT THE OFFSET, and a partial view of a partial view adds the offsets up
template <typename F, typename B> struct PartialBlockMut {
  B* block; size_t start_offset, length;
  Slice<F> channel_as_slice_mut(size_t ch) { Slice<F> s = block->channel_as_slice_mut(ch); return {s.ptr + start_offset, length}; }
  size_t channels() const { return block->channels(); }
  size_t block_size() const { return length; }
  PartialBlockMut<F, PartialBlockMut> partial_mut(size_t off, size_t len) { return {this, off, len}; }
};
template <typename F> template <typename Self> auto ContiguousBlock<F>::partial_mut(size_t start_offset, size_t length) {
  return PartialBlockMut<F, ContiguousBlock<F>>{this, start_offset, length};
}
// RawAggregateBlockRead (knaster_graph/src/block.rs:158-197): one pointer per input channel
template <typename F> struct AggregateBlockRead {
  const F* const* ch; size_t n_channels, bs;
  Slice<const F> channel_as_slice(size_t c) const { return {ch[c], bs}; }
};
// PartialBlock (block.rs:269-302)
template <typename F, typename B> struct PartialBlock {
  const B* block; size_t start_offset, length;
  Slice<const F> channel_as_slice(size_t c) const { Slice<const F> s = block->channel_as_slice(c); return {s.ptr + start_offset, length}; }
};
// BlockMetadata::make_partial (knaster_core/src/ugen.rs:87-93): offset and clock move on, the frame count is the partial one
inline AudioCtx make_partial(const AudioCtx& c, size_t start_offset, size_t length) {
  AudioCtx p = c;
  p.block_start_offset = c.block_start_offset + start_offset;
  p.frames_to_process = length;
  p.frame_clock = c.frame_clock + start_offset;
  return p;
}

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
  // lib.rs: UGen::process_block -> [knh_bank_set_input] knh_bank_process_block_channels
  //   input / output: the reference's block views above -- whole blocks from Task::run (task.rs:25-31), partial ones from a
  //   splitting wrapper (precise_timing.rs:98-110).
  template <typename InBlock, typename OutBlock>
  int32_t process_block(AudioCtx& ctx, UGenFlags& flags, const InBlock& input, OutBlock& output) {
    if (!init_error_.empty()) {
      for (size_t ch = 0; ch < 2; ++ch) {
        Slice<F> s = output.channel_as_slice_mut(ch);
        std::memset(s.ptr, 0, s.len * sizeof(F));
      }
      return KNH_ERR_NOT_INITIALISED;
    }
    const size_t n_in = INPUTS + n_ar_;
    if (n_in > 0) {
      const size_t bs = block_size_, off = ctx.block_start_offset, ftp = ctx.frames_to_process;
      for (size_t ch = 0; ch < INPUTS; ++ch) {
        Slice<const F> src = input.channel_as_slice(ch);
        std::memcpy(&in_pack_[ch * bs + off], src.ptr, std::min(std::min(src.len, ftp), bs - off) * sizeof(F));
      }
      for (size_t k = 0; k < n_ar_; ++k) {
        F* dst = &in_pack_[(INPUTS + k) * bs];
        if (!ar_bufs_[k]) std::memset(dst, 0, bs * sizeof(F));
        else std::memcpy(dst, ar_bufs_[k], bs * sizeof(F));
      }
      (void)knh_bank_set_input(h_, 1, in_pack_.data());
    }
    uint32_t f = 0;
    void* out[2] = {output.channel_as_slice_mut(0).ptr, output.channel_as_slice_mut(1).ptr};
    const int32_t rc = knh_bank_process_block_channels(h_, ctx.frames_to_process, ctx.block_start_offset, ctx.frame_clock, out, &f);
    if (f & KNH_FLAG_ALL_DONE) flags.mark_done(0);
    return rc;
  }
  // whole blocks as Task::run builds them: INPUTS channel pointers, one contiguous [2][block_size] output
  int32_t process_block(AudioCtx& ctx, UGenFlags& flags, const F* const* input, F* output) {
    AggregateBlockRead<F> in{input, INPUTS, block_size_};
    ContiguousBlock<F> out{output, 2, block_size_};
    return process_block(ctx, flags, in, out);
  }
  // lib.rs: UGen::param_apply -> knh_bank_param_apply
  int32_t param_apply(AudioCtx&, size_t index, Value kind, double f = 0.0, int64_t i = 0) {
    const size_t param = index % MAX_PARAMS, rest = index / MAX_PARAMS;
    const uint32_t k = kind == Value::Float ? KNH_VALUE_FLOAT : kind == Value::Trigger ? KNH_VALUE_TRIGGER : kind == Value::Integer ? KNH_VALUE_INTEGER : KNH_VALUE_BOOL;
    return knh_bank_param_apply(h_, static_cast<uint32_t>(rest / n_stages_), static_cast<uint32_t>(rest % n_stages_), static_cast<uint32_t>(param), k, f, i);
  }
  // lib.rs: GpuVoiceBank::param_apply_range -> knh_bank_param_apply_range: one parameter of the voices [voice_begin, voice_end)
  int32_t param_apply_range(uint32_t voice_begin, uint32_t voice_end, size_t stage, size_t param, Value kind, double f = 0.0, int64_t i = 0) {
    const uint32_t k = kind == Value::Float ? KNH_VALUE_FLOAT : kind == Value::Trigger ? KNH_VALUE_TRIGGER : kind == Value::Integer ? KNH_VALUE_INTEGER : KNH_VALUE_BOOL;
    return knh_bank_param_apply_range(h_, voice_begin, voice_end, static_cast<uint32_t>(stage), static_cast<uint32_t>(param), k, f, i);
  }
  // lib.rs: GpuVoiceBank::param_apply_many -> knh_bank_param_apply_many: one parameter of many voices in one call
  int32_t param_apply_many(const std::vector<size_t>& indices, Value kind, double f = 0.0, int64_t i = 0) {
    const uint32_t k = kind == Value::Float ? KNH_VALUE_FLOAT : kind == Value::Trigger ? KNH_VALUE_TRIGGER : kind == Value::Integer ? KNH_VALUE_INTEGER : KNH_VALUE_BOOL;
    const size_t n = indices.size();
    b_voices_.clear(); b_stages_.clear(); b_params_.clear();  // (the arrays are kept between calls, as the shim's `batch`)
    for (size_t q = 0; q < n; ++q) {
      const size_t param = indices[q] % MAX_PARAMS, rest = indices[q] / MAX_PARAMS;
      b_voices_.push_back(static_cast<uint32_t>(rest / n_stages_));
      b_stages_.push_back(static_cast<uint32_t>(rest % n_stages_));
      b_params_.push_back(static_cast<uint32_t>(param));
    }
    b_kinds_.assign(n, k);
    b_f_.assign(n, f);
    b_i_.assign(n, i);
    return knh_bank_param_apply_many(h_, n, b_voices_.data(), b_stages_.data(), b_params_.data(), b_kinds_.data(), b_f_.data(), b_i_.data(), nullptr);
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
  std::vector<uint32_t> b_voices_, b_stages_, b_params_, b_kinds_;
  std::vector<double> b_f_;
  std::vector<int64_t> b_i_;
};

}  // namespace shim_twin
