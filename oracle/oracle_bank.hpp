// oracle_bank.hpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
//
// Builds, for a voice-chain descriptor of include/knaster_hip.h, the UNFUSED
// node graph the reference would run for it: one node per UGen, operator nodes
// (Constant + MathUGen) exactly as graph_edit.rs:1036-1066 creates them, and a
// linear chain of MathUGen<Add> for the additive graph output
// (knaster_graph/src/graph.rs:827-872).  Parity status: see knaster_oracle.hpp.
#pragma once
#include <stdexcept>
#include <thread>

#include "../include/knaster_hip.h"
#include "knaster_oracle.hpp"

namespace kno {

inline int stage_n_ctor_args(uint16_t kind) {
  switch (kind) {
    case KNH_STAGE_SIN_WT: case KNH_STAGE_SIN_NUMERIC: return 1;
    case KNH_STAGE_SVF: return 4;
    case KNH_STAGE_ONEPOLE_LPF: return 1;
    case KNH_STAGE_ONEPOLE_HPF: return 0;
    case KNH_STAGE_MUL_ENV_ASR: case KNH_STAGE_MUL_ENV_AR: return 2;
    case KNH_STAGE_MUL_CONST: case KNH_STAGE_ADD_CONST: case KNH_STAGE_SUB_CONST: case KNH_STAGE_DIV_CONST: return 1;
    case KNH_STAGE_WR_MUL: case KNH_STAGE_WR_ADD: case KNH_STAGE_WR_SUB: return 1;
    case KNH_STAGE_WR_VSUB: case KNH_STAGE_WR_DIV: case KNH_STAGE_WR_VDIV: case KNH_STAGE_WR_POWF: case KNH_STAGE_WR_POWI:
    case KNH_STAGE_POW_CONST: case KNH_STAGE_SAMPLE_DELAY: case KNH_STAGE_PHASOR: case KNH_STAGE_ALLPASS_DELAY: case KNH_STAGE_ALLPASS_FB_DELAY: return 1;
    case KNH_STAGE_SAFETY_LIMITER: return 0;
    case KNH_STAGE_POLYBLEP: return 2;
    case KNH_STAGE_BUFFER_READER: return 3;
    case KNH_STAGE_WHITE_NOISE: case KNH_STAGE_PINK_NOISE: case KNH_STAGE_BROWN_NOISE: return 1;
    case KNH_STAGE_RANDOM_LIN: return 2;
    case KNH_STAGE_PAN2: return 1;
    case KNH_STAGE_INPUT: return 1;
    case KNH_STAGE_MATH_ADD: case KNH_STAGE_MATH_SUB: case KNH_STAGE_MATH_MUL: case KNH_STAGE_MATH_DIV: case KNH_STAGE_MATH_POW: return 0;
    case KNH_STAGE_MUL_ENVELOPE: return -1;  // 4 + 2 * n_max
  }
  return 0;
}
inline bool stage_is_wrapper(uint16_t kind) {
  return kind == KNH_STAGE_WR_MUL || kind == KNH_STAGE_WR_ADD || kind == KNH_STAGE_WR_SUB ||
         (kind >= KNH_STAGE_WR_VSUB && kind <= KNH_STAGE_WR_POWI);
}

// Where a (voice, stage) parameter lives in a built graph.
struct ParamTarget {
  NodeKey node = 0;
  size_t index_offset = 0;  // added to the stage-local parameter index
  size_t n_params = 0;
};

template <typename F>
struct VoiceChainBuilder {
  const std::vector<knh_stage_desc>& stages;
  std::shared_ptr<const Buffer<F>> buffer;  // the bank's shared Buffer, for a BufferReader stage
  explicit VoiceChainBuilder(const std::vector<knh_stage_desc>& s) : stages(s) {}

  // Builds one voice in `g`, returns the node producing the voice's signal and
  // fills targets[stage].  args[stage] = ctor args of this voice.
  NodeKey build(Graph<F>& g, const std::vector<std::vector<double>>& args, std::vector<ParamTarget>& targets) {
    targets.assign(stages.size(), ParamTarget{});
    bool have_x = false;
    NodeKey x = 0;
    std::vector<NodeKey> out_of(stages.size(), 0);  // the node whose output is stage s's signal (GRAPH_KEY: a graph input) ...
    std::vector<uint16_t> ch_of(stages.size(), 0);  // ... and which of its channels (a graph input's number)
    uint16_t x_ch = 0;
    for (size_t s = 0; s < stages.size(); ++s) {
      const knh_stage_desc& st = stages[s];
      if (stage_is_wrapper(st.kind)) throw std::runtime_error("wrapper stage without a node to wrap");
      const std::vector<double>& a = args[s];
      // which signal the stage reads: the stage before it, or the one it names (knh_stage_desc.input)
      if (st.input > s || st.input2 > s) throw std::runtime_error("a stage reads the output of an earlier stage");
      if (st.input != 0) { x = out_of[st.input - 1]; x_ch = ch_of[st.input - 1]; }
      if (st.kind == KNH_STAGE_INPUT) {  // the bank node's input channel: the voice graph's own input of that number
        out_of[s] = GRAPH_KEY;
        ch_of[s] = static_cast<uint16_t>(a[0]);
        x = GRAPH_KEY;
        x_ch = ch_of[s];
        have_x = true;
        continue;
      }
      if (st.kind >= KNH_STAGE_MATH_ADD && st.kind <= KNH_STAGE_MATH_POW) {  // MathUGen<_, U1, Op> of two signals (graph_edit.rs:936-971)
        if (st.input == 0 || st.input2 == 0) throw std::runtime_error("a MATH stage names both operands");
        const MathOp op = st.kind == KNH_STAGE_MATH_ADD ? MathOp::Add : st.kind == KNH_STAGE_MATH_SUB ? MathOp::Sub
                        : st.kind == KNH_STAGE_MATH_MUL ? MathOp::Mul : st.kind == KNH_STAGE_MATH_DIV ? MathOp::Div : MathOp::Pow;
        UGenPtr<F> math2 = std::make_unique<MathUGen<F>>(1, op);
        // wrapper stages that follow wrap this node
        size_t s2 = s + 1;
        std::vector<std::pair<size_t, size_t>> wr_targets;
        while (s2 < stages.size() && stage_is_wrapper(stages[s2].kind)) {
          size_t off = math2->parameters();
          WrOp wop = WrOp::Mul;
          switch (stages[s2].kind) {
            case KNH_STAGE_WR_ADD: wop = WrOp::Add; break;
            case KNH_STAGE_WR_SUB: wop = WrOp::Sub; break;
            case KNH_STAGE_WR_VSUB: wop = WrOp::VSub; break;
            case KNH_STAGE_WR_DIV: wop = WrOp::Div; break;
            case KNH_STAGE_WR_VDIV: wop = WrOp::VDiv; break;
            case KNH_STAGE_WR_POWF: wop = WrOp::Powf; break;
            case KNH_STAGE_WR_POWI: wop = WrOp::Powi; break;
            default: break;
          }
          if (wop == WrOp::Powi) math2 = std::make_unique<WrMath<F>>(std::move(math2), static_cast<int32_t>(args[s2][0]));
          else math2 = std::make_unique<WrMath<F>>(std::move(math2), wop, fnew<F>(args[s2][0]));
          wr_targets.emplace_back(s2, off);
          ++s2;
        }
        NodeKey m = g.push(std::move(math2));
        g.connect_to_node(out_of[st.input - 1], ch_of[st.input - 1], 0, m, false);
        g.connect_to_node(out_of[st.input2 - 1], ch_of[st.input2 - 1], 1, m, false);
        targets[s].node = m;
        for (auto& [ws, off] : wr_targets) {
          targets[ws].node = m;
          targets[ws].index_offset = off;
          targets[ws].n_params = stages[ws].kind == KNH_STAGE_WR_MUL ? 1 : 0;
        }
        for (size_t k = s; k < s2; ++k) { out_of[k] = m; ch_of[k] = 0; }
        x = m;
        x_ch = 0;
        have_x = true;
        s = s2 - 1;
        continue;
      }
      // Core UGen of the stage (the parameterised node).
      UGenPtr<F> core;
      switch (st.kind) {
        case KNH_STAGE_SIN_WT: core = std::make_unique<SinWt<F>>(fnew<F>(a[0])); break;
        case KNH_STAGE_SIN_NUMERIC: core = std::make_unique<SinNumeric<F>>(fnew<F>(a[0])); break;
        case KNH_STAGE_SVF:
          core = std::make_unique<SvfFilter<F>>(svf_type_from_pinteger(static_cast<uint64_t>(a[0])), fnew<F>(a[1]),
                                                fnew<F>(a[2]), fnew<F>(a[3]));
          break;
        case KNH_STAGE_ONEPOLE_LPF: core = std::make_unique<OnePoleLpf<F>>(fnew<F>(a[0])); break;
        case KNH_STAGE_ONEPOLE_HPF: core = std::make_unique<OnePoleHpf<F>>(); break;
        case KNH_STAGE_PHASOR: core = std::make_unique<Phasor<F>>(a[0]); break;
        case KNH_STAGE_WHITE_NOISE: core = std::make_unique<WhiteNoise<F>>(a[0] >= 0.0 ? static_cast<uint64_t>(a[0]) : 0u); break;
        case KNH_STAGE_PINK_NOISE: core = std::make_unique<PinkNoise<F>>(a[0] >= 0.0 ? static_cast<uint64_t>(a[0]) : 0u); break;
        case KNH_STAGE_BROWN_NOISE: core = std::make_unique<BrownNoise<F>>(a[0] >= 0.0 ? static_cast<uint64_t>(a[0]) : 0u); break;
        case KNH_STAGE_RANDOM_LIN: core = std::make_unique<RandomLin<F>>(a[0] >= 0.0 ? static_cast<uint64_t>(a[0]) : 0u, fnew<F>(a[1])); break;
        case KNH_STAGE_BUFFER_READER:
          if (!buffer) throw std::runtime_error("BufferReader stage without a buffer");
          core = std::make_unique<BufferReader<F>>(buffer, a[0], a[1] != 0.0, a[2]);
          break;
        case KNH_STAGE_POLYBLEP:
          core = std::make_unique<PolyBlep<F>>(waveform_from_pinteger(a[0] >= 0.0 ? static_cast<uint64_t>(a[0]) : 0u), fnew<F>(a[1]));
          break;
        case KNH_STAGE_SAFETY_LIMITER: core = std::make_unique<SafetyLimiter<F>>(); break;
        case KNH_STAGE_PAN2:
          if (s + 1 != stages.size()) throw std::runtime_error("Pan2 must be the last stage");
          core = std::make_unique<Pan2<F>>(static_cast<float>(a[0]));
          break;
        case KNH_STAGE_ALLPASS_FB_DELAY: core = std::make_unique<AllpassFeedbackDelay<F>>(Seconds::from_secs_f64(a[0])); break;
        case KNH_STAGE_ALLPASS_DELAY: core = std::make_unique<AllpassDelay<F>>(Seconds::from_secs_f64(a[0])); break;
        case KNH_STAGE_SAMPLE_DELAY: core = std::make_unique<SampleDelay<F>>(Seconds::from_secs_f64(a[0])); break;
        case KNH_STAGE_MUL_ENV_ASR: core = std::make_unique<EnvAsr<F>>(fnew<F>(a[0]), fnew<F>(a[1])); break;
        case KNH_STAGE_MUL_ENV_AR: core = std::make_unique<EnvAr<F>>(fnew<F>(a[0]), fnew<F>(a[1])); break;
        case KNH_STAGE_MUL_ENVELOPE: {
          std::vector<EnvelopeSegment> segs;
          size_t n_max = (a.size() - 4) / 2, n = a[3] >= 1 ? static_cast<size_t>(a[3]) : 1;
          if (n > n_max) n = n_max;
          for (size_t k = 0; k < n; ++k) segs.emplace_back(a[4 + 2 * k], a[5 + 2 * k]);
          auto e = std::make_unique<Envelope<F>>(a[0], std::move(segs));
          e->time_scale = a[1];
          e->looping = a[2] != 0.0;
          core = std::move(e);
        } break;
        case KNH_STAGE_MUL_CONST: case KNH_STAGE_ADD_CONST: case KNH_STAGE_SUB_CONST: case KNH_STAGE_DIV_CONST:
        case KNH_STAGE_POW_CONST:
          core = std::make_unique<Constant<F>>(fnew<F>(a[0]));
          break;
        default: throw std::runtime_error("unknown stage kind");
      }
      const bool two_node = (st.kind >= KNH_STAGE_MUL_ENV_ASR && st.kind <= KNH_STAGE_DIV_CONST) || st.kind == KNH_STAGE_MUL_ENVELOPE ||
                            st.kind == KNH_STAGE_POW_CONST;
      targets[s].n_params = core->parameters();
      if (st.kind == KNH_STAGE_SIN_WT && (st.flags & KNH_STAGE_FLAG_AR_FREQ)) core = std::make_unique<WrArParams<F>>(std::move(core));
      // knh_stage_desc.ar_param: `.ar_params()` on the parameterised node, its parameter linked below (a wrapper stage's
      // own parameter -- WrMul's "wr_mul" -- is the wrapped node's, so the wrapper goes outside the Wr* stages: further down)
      if (st.ar_param != 0 && !(st.flags & KNH_STAGE_FLAG_AR_FREQ)) core = std::make_unique<WrArParams<F>>(std::move(core));
      if (st.flags & KNH_STAGE_FLAG_SMOOTH_PARAMS) {
        if (st.flags & KNH_STAGE_FLAG_AR_FREQ) throw std::runtime_error("SMOOTH_PARAMS and AR_FREQ cannot be combined");
        core = std::make_unique<WrSmoothParams<F>>(std::move(core));
      }
      // Wrapper stages that follow wrap the node producing x.  For single-node
      // stages that is `core`; for two-node stages it is the MathUGen.
      UGenPtr<F> math;
      if (two_node) {
        MathOp op = MathOp::Mul;
        if (st.kind == KNH_STAGE_ADD_CONST) op = MathOp::Add;
        if (st.kind == KNH_STAGE_SUB_CONST) op = MathOp::Sub;
        if (st.kind == KNH_STAGE_DIV_CONST) op = MathOp::Div;
        if (st.kind == KNH_STAGE_POW_CONST) op = MathOp::Pow;
        math = std::make_unique<MathUGen<F>>(1, op);
      }
      UGenPtr<F>& wrapped = two_node ? math : core;
      size_t s2 = s + 1;
      std::vector<std::pair<size_t, size_t>> wr_targets;  // (stage, param offset)
      while (s2 < stages.size() && stage_is_wrapper(stages[s2].kind)) {
        size_t off = wrapped->parameters();
        WrOp op = WrOp::Mul;
        switch (stages[s2].kind) {
          case KNH_STAGE_WR_ADD: op = WrOp::Add; break;
          case KNH_STAGE_WR_SUB: op = WrOp::Sub; break;
          case KNH_STAGE_WR_VSUB: op = WrOp::VSub; break;
          case KNH_STAGE_WR_DIV: op = WrOp::Div; break;
          case KNH_STAGE_WR_VDIV: op = WrOp::VDiv; break;
          case KNH_STAGE_WR_POWF: op = WrOp::Powf; break;
          case KNH_STAGE_WR_POWI: op = WrOp::Powi; break;
          default: break;
        }
        if (op == WrOp::Powi) wrapped = std::make_unique<WrMath<F>>(std::move(wrapped), static_cast<int32_t>(args[s2][0]));
        else wrapped = std::make_unique<WrMath<F>>(std::move(wrapped), op, fnew<F>(args[s2][0]));
        wr_targets.emplace_back(s2, off);
        ++s2;
      }
      // a wrapper stage whose own parameter (WrMul's "wr_mul") is driven at audio rate: `.wr_mul(v).ar_params()`
      size_t wr_ar_stage = 0, wr_ar_index = 0;
      for (auto& [ws, off] : wr_targets)
        if (stages[ws].ar_param != 0) {
          if (wr_ar_stage != 0 || st.ar_param != 0) throw std::runtime_error("one audio-rate parameter per node and its wrappers");
          wr_ar_stage = ws;
          wr_ar_index = off + (stages[ws].ar_param - 1);
        }
      if (wr_ar_stage != 0) wrapped = std::make_unique<WrArParams<F>>(std::move(wrapped));
      // WrPreciseTiming is outermost (precise_timing.rs:13).
      if (st.delayed_changes_per_block > 0) core = std::make_unique<WrPreciseTiming<F>>(st.delayed_changes_per_block, std::move(core));
      if (two_node && !wr_targets.empty() && stages[s2 - 1].delayed_changes_per_block > 0)
        math = std::make_unique<WrPreciseTiming<F>>(stages[s2 - 1].delayed_changes_per_block, std::move(math));

      const bool is_source = st.kind == KNH_STAGE_SIN_WT || st.kind == KNH_STAGE_SIN_NUMERIC || st.kind == KNH_STAGE_PHASOR ||
                             st.kind == KNH_STAGE_POLYBLEP || st.kind == KNH_STAGE_BUFFER_READER ||
                             st.kind == KNH_STAGE_WHITE_NOISE || st.kind == KNH_STAGE_PINK_NOISE || st.kind == KNH_STAGE_BROWN_NOISE ||
                             st.kind == KNH_STAGE_RANDOM_LIN;
      const bool ar = st.kind == KNH_STAGE_SIN_WT && (st.flags & KNH_STAGE_FLAG_AR_FREQ);
      NodeKey core_key = g.push(std::move(core));
      targets[s].node = core_key;
      NodeKey out_key = core_key;
      if (is_source) {
        if (ar) {
          if (!have_x) throw std::runtime_error("AR_FREQ stage needs a preceding signal");
          if (x == GRAPH_KEY) throw std::runtime_error("an audio-rate parameter edge starts at a node, not at a graph input");
          g.connect_to_parameter(x, 0, 0, core_key);
        }  // (a source after the first stage starts a new signal of the voice)
      } else if (!two_node) {
        if (!have_x) throw std::runtime_error("processor stage needs a preceding signal");
        g.connect_to_node(x, x_ch, 0, core_key, false);
      } else {
        if (!have_x) throw std::runtime_error("math stage needs a preceding signal");
        NodeKey m = g.push(std::move(math));
        g.connect_to_node(x, x_ch, 0, m, false);
        g.connect_to_node(core_key, 0, 1, m, false);
        out_key = m;
      }
      for (auto& [ws, off] : wr_targets) {
        targets[ws].node = out_key;
        targets[ws].index_offset = off;
        targets[ws].n_params = stages[ws].kind == KNH_STAGE_WR_MUL ? 1 : 0;
      }
      // audio-rate parameter edges: node.link(param, signal) (graph_edit.rs:735-754 -> connect_replace_to_parameter)
      auto link = [&](const knh_stage_desc& d, size_t param, NodeKey sink) {
        if (d.input2 == 0) throw std::runtime_error("an audio-rate parameter names the signal that drives it (input2)");
        if (out_of[d.input2 - 1] == GRAPH_KEY) throw std::runtime_error("an audio-rate parameter edge starts at a node, not at a graph input");
        g.connect_to_parameter(out_of[d.input2 - 1], ch_of[d.input2 - 1], param, sink);
      };
      if (st.ar_param != 0 && !(st.flags & KNH_STAGE_FLAG_AR_FREQ)) link(st, st.ar_param - 1u, core_key);
      if (wr_ar_stage != 0) link(stages[wr_ar_stage], wr_ar_index, out_key);
      for (size_t k = s; k < s2; ++k) { out_of[k] = out_key; ch_of[k] = 0; }
      x = out_key;
      x_ch = 0;
      have_x = true;
      s = s2 - 1;
    }
    if (!have_x) throw std::runtime_error("empty chain");
    if (x == GRAPH_KEY) throw std::runtime_error("a voice that is only a bank input has no node to connect to the output");
    return x;
  }
};

// An oracle voice bank: the reference-shaped graph for the mix, and (optionally)
// one single-voice graph per voice to observe each voice's own signal.
template <typename F>
struct OracleBank {
  std::vector<knh_stage_desc> stages;
  uint32_t n_voices, out_channels;
  std::vector<std::vector<std::vector<double>>> ctor;  // [voice][stage][arg]
  bool want_mix, want_voices;
  uint32_t sample_rate = 0;
  size_t block_size = 0;
  std::unique_ptr<Graph<F>> mix_graph;
  std::vector<std::vector<ParamTarget>> mix_targets;
  std::vector<std::unique_ptr<Graph<F>>> voice_graphs;
  std::vector<std::vector<ParamTarget>> voice_targets;
  std::vector<F> voice_block;  // [n_voices][block_size], last processed block
  std::vector<uint32_t> done_frames;

  OracleBank(const knh_stage_desc* st, uint32_t n_stages, uint32_t nv, uint32_t oc, bool mix, bool voices)
      : stages(st, st + n_stages), n_voices(nv), out_channels(oc), want_mix(mix), want_voices(voices) {
    ctor.assign(nv, std::vector<std::vector<double>>(n_stages));
    for (uint32_t v = 0; v < nv; ++v)
      for (uint32_t s = 0; s < n_stages; ++s) ctor[v][s].assign(static_cast<size_t>(std::max(0, stage_n_ctor_args(stages[s].kind))), 0.0);
  }
  std::shared_ptr<Buffer<F>> buffer;
  uint32_t in_channels = 0;  // UGen::Inputs of the bank node: every voice graph has that many graph inputs
  // A chain that ends in Pan2: `(voice >> pan).to_graph_out()` (many_sines.rs:59-60) -- the node's outputs 0 and 1
  // go to graph outputs 0 and 1, and every voice has two signals: voice_block is [2][n_voices][block_size].
  bool pan() const { return !stages.empty() && stages.back().kind == KNH_STAGE_PAN2; }
  void init(uint32_t sr, size_t bs) {
    sample_rate = sr;
    block_size = bs;
    VoiceChainBuilder<F> b(stages);
    b.buffer = buffer;
    if (pan() && out_channels != 2) throw std::runtime_error("a chain ending in Pan2 has two output channels");
    if (want_mix) {
      mix_graph = std::make_unique<Graph<F>>(in_channels, out_channels, bs, sr);
      mix_targets.resize(n_voices);
      for (uint32_t v = 0; v < n_voices; ++v) {
        NodeKey x = b.build(*mix_graph, ctor[v], mix_targets[v]);
        for (uint32_t c = 0; c < out_channels; ++c) mix_graph->connect_to_output(x, pan() ? static_cast<uint16_t>(c) : 0, static_cast<uint16_t>(c), true);
      }
      mix_graph->commit_changes();
    }
    if (want_voices) {
      voice_targets.resize(n_voices);
      const uint16_t planes = pan() ? 2 : 1;
      for (uint32_t v = 0; v < n_voices; ++v) {
        voice_graphs.push_back(std::make_unique<Graph<F>>(in_channels, planes, bs, sr));
        NodeKey x = b.build(*voice_graphs[v], ctor[v], voice_targets[v]);
        for (uint16_t c = 0; c < planes; ++c) voice_graphs[v]->connect_to_output(x, c, c, true);
        voice_graphs[v]->commit_changes();
      }
      voice_block.assign(static_cast<size_t>(planes) * n_voices * bs, F(0));
      done_frames.assign(n_voices, 0xFFFFFFFFu);
    }
  }
  template <typename Fn>
  void for_targets(uint32_t voice, uint32_t stage, Fn&& fn) {
    if (want_mix) fn(*mix_graph, mix_targets[voice][stage]);
    if (want_voices) fn(*voice_graphs[voice], voice_targets[voice][stage]);
  }
  int param_apply(uint32_t voice, uint32_t stage, uint32_t param, ParameterValue v) {
    if (voice >= n_voices || stage >= stages.size()) return KNH_ERR_OUT_OF_RANGE;
    int rc = KNH_OK;
    for_targets(voice, stage, [&](Graph<F>& g, const ParamTarget& t) {
      if (param >= t.n_params) { rc = KNH_ERR_OUT_OF_RANGE; return; }
      g.ugen(t.node)->param_apply(g.audio_ctx(), t.index_offset + param, v);
    });
    return rc;
  }
  int set_delay(uint32_t voice, uint32_t stage, uint32_t param, uint16_t delay) {
    if (voice >= n_voices || stage >= stages.size()) return KNH_ERR_OUT_OF_RANGE;
    int rc = KNH_OK;
    for_targets(voice, stage, [&](Graph<F>& g, const ParamTarget& t) {
      if (param >= t.n_params) { rc = KNH_ERR_OUT_OF_RANGE; return; }
      g.ugen(t.node)->set_delay_within_block_for_param(g.audio_ctx(), t.index_offset + param, delay);
    });
    return rc;
  }
  // Through GraphGen's event path with a Time (graph_gen.rs:269-305).
  int schedule(uint32_t voice, uint32_t stage, uint32_t param, ParameterValue v, bool has_time, Time t) {
    if (voice >= n_voices || stage >= stages.size()) return KNH_ERR_OUT_OF_RANGE;
    int rc = KNH_OK;
    for_targets(voice, stage, [&](Graph<F>& g, const ParamTarget& tg) {
      if (param >= tg.n_params) { rc = KNH_ERR_OUT_OF_RANGE; return; }
      if (has_time) g.set_at(tg.node, tg.index_offset + param, v, t);
      else g.set(tg.node, tg.index_offset + param, v);
    });
    return rc;
  }
  // out: [out_channels][block_size] (may be null); in: [in_channels][block_size] (the bank node's input block).  Returns KNH_FLAG_*.
  uint32_t process_block(F* out, const F* in = nullptr) {
    uint32_t flags = 0;
    std::vector<const F*> ins;
    for (uint32_t c = 0; c < in_channels; ++c) ins.push_back(in ? in + static_cast<size_t>(c) * block_size : nullptr);
    if (in_channels && !in) throw std::runtime_error("a bank with input channels needs an input block");
    if (want_mix) {
      std::vector<F> tmp;
      if (!out) { tmp.resize(out_channels * block_size); out = tmp.data(); }
      mix_graph->run(ins, out);
      if (mix_graph->last_flags.done_) flags |= KNH_FLAG_ANY_DONE;
    }
    if (want_voices) {
      std::vector<F> lr;
      if (pan()) lr.resize(2 * block_size);
      for (uint32_t v = 0; v < n_voices; ++v) {
        if (pan()) {  // [left][right] of this voice -> the two planes
          voice_graphs[v]->run(ins, lr.data());
          std::copy(lr.begin(), lr.begin() + block_size, voice_block.begin() + static_cast<size_t>(v) * block_size);
          std::copy(lr.begin() + block_size, lr.end(), voice_block.begin() + (static_cast<size_t>(n_voices) + v) * block_size);
        } else {
          voice_graphs[v]->run(ins, voice_block.data() + static_cast<size_t>(v) * block_size);
        }
        uint32_t f = 0xFFFFFFFFu;
        done_frames[v] = voice_graphs[v]->last_flags.done(&f) ? f : 0xFFFFFFFFu;
        if (done_frames[v] != 0xFFFFFFFFu) flags |= KNH_FLAG_ANY_DONE;
      }
    }
    return flags;
  }
};

}  // namespace kno
