// chain_signature.hpp -- a chain descriptor (knh_stage_desc[]) checked and turned into the device signature
// (kernel_registry.hpp: one character per stage, operands and signal slots for graph-shaped voices).  Included by bank.hip only.
#pragma once

namespace {

// Chain descriptor -> device signature (kernel_registry.hpp) with structural validation.
int build_signature(const knh_stage_desc* st, uint32_t n, std::string* sig, std::string* why) {
  if (n == 0) { *why = "empty chain"; return KNH_ERR_INVALID_ARGUMENT; }
  sig->clear();
  bool have_x = false;
  for (uint32_t i = 0; i < n; ++i) {
    if (st[i].kind >= KNH_STAGE_KIND_COUNT) { *why = "unknown stage kind"; return KNH_ERR_INVALID_ARGUMENT; }
    const bool source = st[i].kind == KNH_STAGE_SIN_WT || st[i].kind == KNH_STAGE_SIN_NUMERIC || st[i].kind == KNH_STAGE_PHASOR ||
                        st[i].kind == KNH_STAGE_WHITE_NOISE || st[i].kind == KNH_STAGE_PINK_NOISE || st[i].kind == KNH_STAGE_BROWN_NOISE ||
                        st[i].kind == KNH_STAGE_RANDOM_LIN ||
                        st[i].kind == KNH_STAGE_POLYBLEP || st[i].kind == KNH_STAGE_BUFFER_READER || st[i].kind == KNH_STAGE_INPUT;
    const bool ar = st[i].kind == KNH_STAGE_SIN_WT && (st[i].flags & KNH_STAGE_FLAG_AR_FREQ);
    const bool math2 = is_math2_kind(st[i].kind);
    if (st[i].flags & ~(KNH_STAGE_FLAG_AR_FREQ | KNH_STAGE_FLAG_SMOOTH_PARAMS)) { *why = "unknown stage flag"; return KNH_ERR_INVALID_ARGUMENT; }
    if ((st[i].flags & KNH_STAGE_FLAG_SMOOTH_PARAMS) && (st[i].flags & KNH_STAGE_FLAG_AR_FREQ)) { *why = "SMOOTH_PARAMS and AR_FREQ cannot be combined"; return KNH_ERR_INVALID_ARGUMENT; }
    if ((st[i].flags & KNH_STAGE_FLAG_SMOOTH_PARAMS) && is_wrapper_kind(st[i].kind)) { *why = "SMOOTH_PARAMS applies to a node, not to a wrapper stage"; return KNH_ERR_INVALID_ARGUMENT; }
    if ((st[i].flags & KNH_STAGE_FLAG_AR_FREQ) && st[i].kind != KNH_STAGE_SIN_WT) { *why = "AR_FREQ is only defined for SIN_WT"; return KNH_ERR_INVALID_ARGUMENT; }
    // operands: `input` / `input2` name the stage whose output is read (1 + its index), 0 = the stage before this one
    if (st[i].input > i || st[i].input2 > i) { *why = "a stage reads the output of an earlier stage"; return KNH_ERR_INVALID_ARGUMENT; }
    if (math2 && (st[i].input == 0 || st[i].input2 == 0)) { *why = "a KNH_STAGE_MATH_* stage names both of its operands (input, input2)"; return KNH_ERR_INVALID_ARGUMENT; }
    if (!math2 && st[i].input2 != 0 && st[i].ar_param == 0) { *why = "input2 is the second operand of the KNH_STAGE_MATH_* stages and the driver of an audio-rate parameter (ar_param)"; return KNH_ERR_INVALID_ARGUMENT; }
    if (math2 && st[i].ar_param != 0) { *why = "a KNH_STAGE_MATH_* stage has no parameters"; return KNH_ERR_INVALID_ARGUMENT; }
    if (is_wrapper_kind(st[i].kind) && st[i].input != 0) { *why = "a wrapper stage wraps the stage before it (input = 0)"; return KNH_ERR_INVALID_ARGUMENT; }
    if (source && !ar && st[i].input != 0) { *why = "a source stage reads no signal"; return KNH_ERR_INVALID_ARGUMENT; }
    if ((!source || ar) && !have_x) { *why = "stage needs a preceding signal"; return KNH_ERR_INVALID_ARGUMENT; }
    if ((st[i].kind == KNH_STAGE_SAMPLE_DELAY || st[i].kind == KNH_STAGE_ALLPASS_DELAY || st[i].kind == KNH_STAGE_ALLPASS_FB_DELAY) &&
        sig->find_first_of("DYZ") != std::string::npos) {
      *why = "at most one delay stage per chain";
      return KNH_ERR_INVALID_ARGUMENT;
    }
    if (st[i].kind == KNH_STAGE_MUL_ENVELOPE && sig->find('V') != std::string::npos) { *why = "at most one Envelope stage per chain"; return KNH_ERR_INVALID_ARGUMENT; }
    if (st[i].kind == KNH_STAGE_PAN2 && i + 1 != n) { *why = "Pan2 ends the chain: it must be the last stage"; return KNH_ERR_INVALID_ARGUMENT; }
    if (st[i].kind == KNH_STAGE_PAN2 && st[i].delayed_changes_per_block > 0) { *why = "Pan2 cannot be wrapped in WrPreciseTiming here (its gains change at block boundaries)"; return KNH_ERR_INVALID_ARGUMENT; }
    if (st[i].ar_param != 0) {  // an audio-rate parameter: the node's float parameter ar_param - 1 is driven by the signal input2 names
      const uint32_t p = st[i].ar_param - 1u;
      if (p >= static_cast<uint32_t>(kKinds[st[i].kind].n_params) || expected_value_kind(st[i].kind, p) != KNH_VALUE_FLOAT) { *why = "ar_param names a float parameter of the stage (1 + its index)"; return KNH_ERR_INVALID_ARGUMENT; }
      if (!ar_param_supported(st[i].kind, p)) { *why = "this parameter cannot be driven at audio rate here (knh_stage_desc.ar_param lists what can)"; return KNH_ERR_UNSUPPORTED_CHAIN; }
      if (st[i].input2 == 0) { *why = "an audio-rate parameter names the signal that drives it (input2)"; return KNH_ERR_INVALID_ARGUMENT; }
      if (st[i].flags & (KNH_STAGE_FLAG_AR_FREQ | KNH_STAGE_FLAG_SMOOTH_PARAMS)) { *why = "ar_param cannot be combined with AR_FREQ or SMOOTH_PARAMS on one stage"; return KNH_ERR_INVALID_ARGUMENT; }
      if (kKinds[st[i].kind].sig == 'I' || st[st[i].input2 - 1].kind == KNH_STAGE_INPUT) { *why = "an audio-rate parameter edge starts at a node, not at a bank input (put the input through `* 1.0`)"; return KNH_ERR_INVALID_ARGUMENT; }
    }
    sig->push_back(ar ? 'R' : kKinds[st[i].kind].sig);
    have_x = true;
  }
  // A voice that is a graph (a stage that names its operands, a MathUGen of two signals, a second source): every stage is
  // annotated "@a,b,o" with the SIGNAL SLOTS it reads and writes -- they are part of the kernel's type (knh_dev::At).  Slots
  // are handed out like registers, a signal's slot free again after its last reader, so that a voice of a thousand stages
  // (the reference's 256-oscillator FM cascade) keeps a handful of signals alive, not a thousand.
  bool dag = false;
  {
    uint32_t sources = 0;
    for (uint32_t i = 0; i < n; ++i) {
      const bool ar = st[i].kind == KNH_STAGE_SIN_WT && (st[i].flags & KNH_STAGE_FLAG_AR_FREQ);
      sources += std::strchr("WNPUKOGBFI", kKinds[st[i].kind].sig) != nullptr && !ar;
      dag = dag || is_math2_kind(st[i].kind) || (st[i].input != 0 && st[i].input != i) || st[i].ar_param != 0;
    }
    dag = dag || sources > 1;
  }
  if (dag && n > 512 && !(interp_can_run(st, n) && n <= 4096)) {
    // every stage unrolls into the one kernel the voice is fused into: 91 stages build in 2 s, 379 in a minute, and the
    // time grows faster than the count (the reference's 256-oscillator FM cascade, 1 531 stages, does not finish)
    // (graphs of SinWt oscillators and arithmetic alone are not fused at all: up to 4 096 stages run in kernels_interp.hip)
    *why = "a voice that is a graph may hold at most 512 stages (4 096 if it is made of SinWt oscillators and arithmetic only)";
    return KNH_ERR_UNSUPPORTED_CHAIN;
  }
  if (dag) {
    std::vector<int> a(n, -1), b(n, -1), last_use(n, -1);
    // "the output of stage k" is the output of the NODE stage k stands for: if wrapper stages follow it (they wrap it: the
    // reference's wr_mul() etc. are part of the UGen), what a reader gets is the last wrapper's output
    auto node_output = [&](int k) {
      while (k + 1 < static_cast<int>(n) && is_wrapper_kind(st[k + 1].kind)) ++k;
      return k;
    };
    for (uint32_t i = 0; i < n; ++i) {
      const bool reads = i > 0 && !(std::strchr("WNPUKOGBFI", (*sig)[i]) != nullptr);  // 'R' reads, the plain sources do not
      if (is_math2_kind(st[i].kind)) { a[i] = node_output(st[i].input - 1); b[i] = node_output(st[i].input2 - 1); }
      else if (reads) a[i] = st[i].input ? node_output(st[i].input - 1) : static_cast<int>(i) - 1;
      if (st[i].ar_param != 0) b[i] = node_output(st[i].input2 - 1);  // the signal that drives the parameter
      if (a[i] >= 0) last_use[a[i]] = static_cast<int>(i);
      if (b[i] >= 0) last_use[b[i]] = static_cast<int>(i);
    }
    last_use[n - 1] = static_cast<int>(n);  // the voice's output
    std::vector<int> slot(n, -1);
    std::vector<char> busy;
    std::string out;
    for (uint32_t i = 0; i < n; ++i) {
      const int sa = a[i] >= 0 ? slot[a[i]] : -1, sb = b[i] >= 0 ? slot[b[i]] : -1;
      // operands whose last reader this is give their slot back first: the stage may then write where it read
      if (a[i] >= 0 && last_use[a[i]] == static_cast<int>(i)) busy[sa] = 0;
      if (b[i] >= 0 && last_use[b[i]] == static_cast<int>(i) && sb >= 0) busy[sb] = 0;
      int o = -1;
      if (sa >= 0 && !busy[sa] && !is_math2_kind(st[i].kind)) o = sa;  // in place, like a chain
      for (size_t k = 0; o < 0 && k < busy.size(); ++k)
        if (!busy[k]) o = static_cast<int>(k);
      if (o < 0) { o = static_cast<int>(busy.size()); busy.push_back(0); }
      if (last_use[i] >= 0) busy[o] = 1;  // (a signal nobody reads holds its slot only while it is written)
      slot[i] = o;
      out.push_back((*sig)[i]);
      if (st[i].ar_param != 0) out += "%" + std::to_string(st[i].ar_param - 1);  // "%P": parameter P at audio rate (knh_dev::ArP)
      auto num = [](int v) { return v < 0 ? std::string("_") : std::to_string(v); };  // "_": none
      out += "@" + num(sa) + "," + num(sb) + "," + num(o);
    }
    *sig = out + "#" + std::to_string(busy.size());  // "#R": the number of slots
  }
  return KNH_OK;
}

}  // namespace
