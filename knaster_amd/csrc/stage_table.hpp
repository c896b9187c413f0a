// stage_table.hpp -- what the host knows about each stage kind: slots, parameters, constructor arguments, value kinds, and
// the reference's coefficient arithmetic restated for the host (SvfFilter::set_coeffs, Pan2's fastapprox gains).  Included
// by bank.hip only.  Citations are file:line in the knaster repo.
#pragma once

namespace {

// ---------------------------------------------------------------------------
// Static description of the stage kinds
// ---------------------------------------------------------------------------
struct KindInfo {
  int n_slots, n_params, n_ctor, n_nodes;
  char sig;
  const char* params[6];
};
const KindInfo kKinds[KNH_STAGE_KIND_COUNT] = {
    /* SIN_WT      */ {3, 3, 1, 1, 'W', {"freq", "phase_offset", "reset_phase"}},
    /* SIN_NUMERIC */ {3, 3, 1, 1, 'N', {"freq", "phase_offset", "reset_phase"}},
    /* SVF         */ {8, 5, 4, 1, 'S', {"cutoff_freq", "q", "gain", "filter", "t_calculate_coefficients"}},
    /* ONEPOLE_LPF */ {3, 1, 1, 1, 'L', {"cutoff_freq"}},
    /* ONEPOLE_HPF */ {3, 1, 0, 1, 'H', {"cutoff_freq"}},
    /* MUL_ENV_ASR */ {5, 4, 2, 2, 'A', {"attack_time", "release_time", "t_release", "t_restart"}},
    /* MUL_ENV_AR  */ {5, 3, 2, 2, 'E', {"attack_time", "release_time", "t_restart"}},
    /* MUL_CONST   */ {1, 1, 1, 2, 'm', {"value"}},
    /* ADD_CONST   */ {1, 1, 1, 2, 'a', {"value"}},
    /* SUB_CONST   */ {1, 1, 1, 2, 's', {"value"}},
    /* DIV_CONST   */ {1, 1, 1, 2, 'd', {"value"}},
    /* WR_MUL      */ {1, 1, 1, 0, 'm', {"wr_mul"}},
    /* WR_ADD      */ {1, 0, 1, 0, 'a', {nullptr}},
    /* WR_SUB      */ {1, 0, 1, 0, 's', {nullptr}},
    /* MUL_ENVELOPE*/ {11, 4, -1, 2, 'V', {"time_scale", "jump_to_segment", "t_restart", "t_stop"}},  // n_ctor: 4 + 2 * n_max
    /* WR_VSUB     */ {1, 0, 1, 0, 'v', {nullptr}},
    /* WR_DIV      */ {1, 0, 1, 0, 'd', {nullptr}},
    /* WR_VDIV     */ {1, 0, 1, 0, 'q', {nullptr}},
    /* WR_POWF     */ {1, 0, 1, 0, 'p', {nullptr}},
    /* WR_POWI     */ {1, 0, 1, 0, 'i', {nullptr}},
    /* POW_CONST   */ {1, 1, 1, 2, 'p', {"value"}},
    /* SAMPLE_DELAY*/ {4, 1, 1, 1, 'D', {"delay_time"}},
    /* PHASOR      */ {4, 1, 1, 1, 'P', {"freq"}},
    /* SAFETY_LIM  */ {0, 0, 0, 1, 'X', {nullptr}},
    /* POLYBLEP    */ {5, 3, 2, 1, 'B', {"freq", "pulse_width", "waveform"}},
    /* ALLPASS_DLY */ {7, 1, 1, 1, 'Y', {"delay_time"}},
    /* ALLPASS_FB  */ {8, 2, 1, 1, 'Z', {"delay_time", "feedback"}},
    /* BUFFER_READ */ {10, 6, 3, 1, 'F', {"rate", "looping", "start_s", "duration_s", "end_s", "t_restart"}},
    /* WHITE_NOISE */ {2, 0, 1, 1, 'U', {nullptr}},
    /* PINK_NOISE  */ {14, 0, 1, 1, 'K', {nullptr}},
    /* BROWN_NOISE */ {3, 0, 1, 1, 'O', {nullptr}},
    /* RANDOM_LIN  */ {6, 1, 2, 1, 'G', {"freq"}},
    /* PAN2        */ {2, 1, 1, 1, 'J', {"pan"}},
    /* MATH_ADD    */ {0, 0, 0, 1, '+', {nullptr}},
    /* MATH_SUB    */ {0, 0, 0, 1, '-', {nullptr}},
    /* MATH_MUL    */ {0, 0, 0, 1, '*', {nullptr}},
    /* MATH_DIV    */ {0, 0, 0, 1, '/', {nullptr}},
    /* MATH_POW    */ {0, 0, 0, 1, '^', {nullptr}},
    /* INPUT       */ {1, 0, 1, 0, 'I', {nullptr}},
};
inline bool is_math2_kind(uint16_t kind) { return kind >= KNH_STAGE_MATH_ADD && kind <= KNH_STAGE_MATH_POW; }
// A voice that is a graph rather than a chain: explicit operands, a MathUGen of two signals, or a second source.
inline bool signature_is_dag(const std::string& sig) { return sig.find('@') != std::string::npos; }

inline bool is_wrapper_kind(uint16_t kind) {
  return kind == KNH_STAGE_WR_MUL || kind == KNH_STAGE_WR_ADD || kind == KNH_STAGE_WR_SUB || (kind >= KNH_STAGE_WR_VSUB && kind <= KNH_STAGE_WR_POWI);
}
// A graph-shaped voice the frame-parallel interpreter can run (kernels_interp.hip): free-running SinWt oscillators and
// arithmetic only, nothing wrapped in WrPreciseTiming or WrSmoothParams.
bool interp_can_run(const knh_stage_desc* st, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i) {
    if (st[i].flags != 0 || st[i].delayed_changes_per_block != 0 || st[i].ar_param != 0) return false;
    if (std::strchr("Wmasdvq*+-/", kKinds[st[i].kind].sig) == nullptr) return false;
  }
  return true;
}

// Rough instructions per sample of each stage (kernel_registry.hpp signature characters), used only to balance the
// stage groups of a run-time-built pipeline.
inline int stage_cost(char c) {
  switch (c) {
    case 'W': return 7;  case 'R': return 10; case 'N': return 25; case 'S': return 10; case 'L': return 3;
    case 'H': return 4;  case 'A': return 5;  case 'E': return 5;  case 'V': return 14; case 'D': return 12;
    case 'd': case 'q': return 8;  case 'p': return 40; case 'i': return 10; case 'P': return 6; case 'U': return 14; case 'O': return 19; case 'K': return 50; case 'G': return 10; case 'X': return 4; case 'B': return 45; case 'Y': return 14; case 'Z': return 18; case 'F': return 16;
    default: return 1;
  }
}
// Cuts `sig` into 1..3 contiguous groups minimising the heaviest group (each group pays ~4 for its tile I/O);
// returns the number of cuts and the first stage of every group after the first.
inline unsigned partition_chain(const std::string& sig, unsigned cuts[2]) {
  const unsigned n = static_cast<unsigned>(sig.size());
  std::vector<int> pre(n + 1, 0);
  for (unsigned i = 0; i < n; ++i) pre[i + 1] = pre[i] + stage_cost(sig[i]);
  auto cost = [&](unsigned a, unsigned b) { return pre[b] - pre[a] + 4; };
  int best = cost(0, n);
  unsigned n_cuts = 0;
  for (unsigned i = 1; i < n; ++i) {
    const int c2 = std::max(cost(0, i), cost(i, n));
    if (c2 < best) { best = c2; n_cuts = 1; cuts[0] = i; }
  }
  for (unsigned i = 1; i < n; ++i)
    for (unsigned j = i + 1; j < n; ++j) {
      const int c3 = std::max(cost(0, i), std::max(cost(i, j), cost(j, n)));
      if (c3 < best) { best = c3; n_cuts = 2; cuts[0] = i; cuts[1] = j; }
    }
  return n_cuts;
}

// 0 float, 1 trigger, 2 integer : expected ParameterValue kind per (stage kind, param)
int expected_value_kind(uint16_t kind, uint32_t param) {
  switch (kind) {
    case KNH_STAGE_SIN_WT: case KNH_STAGE_SIN_NUMERIC: return param == 2 ? KNH_VALUE_TRIGGER : KNH_VALUE_FLOAT;
    case KNH_STAGE_SVF: return param == 3 ? KNH_VALUE_INTEGER : param == 4 ? KNH_VALUE_TRIGGER : KNH_VALUE_FLOAT;
    case KNH_STAGE_MUL_ENV_ASR: return param >= 2 ? KNH_VALUE_TRIGGER : KNH_VALUE_FLOAT;
    case KNH_STAGE_MUL_ENV_AR: return param == 2 ? KNH_VALUE_TRIGGER : KNH_VALUE_FLOAT;
    case KNH_STAGE_MUL_ENVELOPE: return param == 0 ? KNH_VALUE_FLOAT : param == 1 ? KNH_VALUE_INTEGER : KNH_VALUE_TRIGGER;
    case KNH_STAGE_POLYBLEP: return param == 2 ? KNH_VALUE_INTEGER : KNH_VALUE_FLOAT;
    case KNH_STAGE_BUFFER_READER: return param == 1 ? KNH_VALUE_BOOL : param == 5 ? KNH_VALUE_TRIGGER : KNH_VALUE_FLOAT;
    default: return KNH_VALUE_FLOAT;
  }
}

std::string g_create_error;  // last failed knh_bank_create

// Rust `as u32` from f64 (saturating; NaN -> 0).  osc.rs:129,134
inline uint32_t sat_u32(double v) {
  if (!(v > 0.0)) return 0u;
  if (v >= 4294967295.0) return 0xFFFFFFFFu;
  return static_cast<uint32_t>(v);
}

// fastapprox::fast::{sin, cos} (crate fastapprox 0.3.1, Cargo.lock:931 -- a crates.io dependency that is not in the
// reference tree), called by Pan2::process (pan.rs:34-35).  Restated from the published algorithm (Paul Mineiro's
// fastapprox, fasttrig.h `fastsin` / `fastcos`, of which the crate is a port): PARITY UNPINNED, DESIGN.md section 2.
// Every operation is an f32 operation in source order (Rust never contracts a*b+c).
inline float fastapprox_fast_sin(float x) {
  const float FOUROVERPI = 1.2732395447351627f, FOUROVERPISQ = 0.40528473456935109f, Q = 0.78444488374548933f;
  uint32_t p, r, s, vx;
  const float P = 0.20363937680730309f, R = 0.015124940802184233f, S = -0.0032225901625579573f;
  std::memcpy(&p, &P, 4); std::memcpy(&r, &R, 4); std::memcpy(&s, &S, 4); std::memcpy(&vx, &x, 4);
  const uint32_t sign = vx & 0x80000000u;
  vx &= 0x7FFFFFFFu;
  float ax;
  std::memcpy(&ax, &vx, 4);
  const float qpprox = FOUROVERPI * x - FOUROVERPISQ * x * ax;
  const float qpproxsq = qpprox * qpprox;
  p |= sign; r |= sign; s ^= sign;
  float pf, rf, sf;
  std::memcpy(&pf, &p, 4); std::memcpy(&rf, &r, 4); std::memcpy(&sf, &s, 4);
  return Q * qpprox + qpproxsq * (pf + qpproxsq * (rf + qpproxsq * sf));
}
inline float fastapprox_fast_cos(float x) {
  const float HALFPI = 1.5707963267948966f, HALFPIMINUSTWOPI = -4.7123889803846899f;
  const float offset = x > HALFPI ? HALFPIMINUSTWOPI : HALFPI;
  return fastapprox_fast_sin(x + offset);
}
// Pan2's two gains for a `pan` parameter value (pan.rs:18-23 / :26-29, then :33-35).
inline void pan2_gains(float pan_param, float* left, float* right) {
  const float pan = pan_param * 0.5f + 0.5f;
  const float rad = pan * 1.57079632679489661923132169163975144f;  // core::f32::consts::FRAC_PI_2
  *left = fastapprox_fast_cos(rad);
  *right = fastapprox_fast_sin(rad);
}

template <typename F> struct Consts;
template <> struct Consts<float> { static constexpr float PI = 3.14159265358979323846f; };
template <> struct Consts<double> { static constexpr double PI = 3.14159265358979323846; };

// SvfFilter::set_coeffs -- knaster_core_dsp/src/ugens/svf.rs:146-242.  F-precision libm calls.
template <typename F>
void svf_coeffs(uint32_t ty, F cutoff, F q, F gain_db, F sr, F out[6]) {
  const F one = 1;
  F g = std::tan((Consts<F>::PI * cutoff) / sr);
  F k = one / q;
  F m0 = 0, m1 = 0, m2 = 0;
  F amp = 0;
  if (ty >= KNH_SVF_BELL && ty <= KNH_SVF_HIGH_SHELF) amp = std::pow(F(10), gain_db / F(40));
  switch (ty) {
    default:
    case KNH_SVF_LOW: m0 = 0; m1 = 0; m2 = one; break;
    case KNH_SVF_BAND: m0 = 0; m1 = one; m2 = 0; break;
    case KNH_SVF_HIGH: m0 = one; m1 = -k; m2 = -one; break;
    case KNH_SVF_NOTCH: m0 = one; m1 = -k; m2 = 0; break;
    case KNH_SVF_PEAK: m0 = one; m1 = -k; m2 = -F(2); break;
    case KNH_SVF_ALL: m0 = one; m1 = -F(2) * k; m2 = 0; break;
    case KNH_SVF_BELL:
      g = g / std::sqrt(amp);
      k = one / (q * amp);
      m0 = one; m1 = k * (amp * amp - one); m2 = 0;
      break;
    case KNH_SVF_LOW_SHELF:
      g = g / std::sqrt(amp);
      m0 = one; m1 = k * (amp - one); m2 = amp * amp - one;
      break;
    case KNH_SVF_HIGH_SHELF:
      g = g * std::sqrt(amp);
      m0 = amp * amp; m1 = k * (one - amp) * amp; m2 = one - amp * amp;
      break;
  }
  const F a1 = one / (one + g * (g + k));
  const F a2 = g * a1;
  const F a3 = g * a2;
  out[0] = a1; out[1] = a2; out[2] = a3; out[3] = m0; out[4] = m1; out[5] = m2;
}

inline uint64_t to_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline uint64_t to_bits(double f) { uint64_t u; std::memcpy(&u, &f, 8); return u; }

struct StageInfo {
  uint16_t kind, flags, dcpb;
  int slot_base, n_slots, n_params, n_ctor;
  int param_base;  // index of this stage's first parameter in the flat per-voice parameter table
  uint16_t input = 0, input2 = 0;  // knh_stage_desc: the stage(s) whose output this one reads (0: the one before it)
  uint16_t ar_param = 0;           // knh_stage_desc: 1 + the float parameter a second signal (input2) drives at audio rate, 0: none
};

// Which (stage kind, float parameter) pairs can be driven at audio rate (knh_stage_desc.ar_param): the setters restated on
// the device (voice_stages.hpp, ar_set).
inline bool ar_param_supported(uint16_t kind, uint32_t param) {
  switch (kind) {
    case KNH_STAGE_SIN_WT: case KNH_STAGE_SIN_NUMERIC: return param <= 1;
    case KNH_STAGE_MUL_CONST: case KNH_STAGE_ADD_CONST: case KNH_STAGE_SUB_CONST: case KNH_STAGE_DIV_CONST: case KNH_STAGE_POW_CONST:
    case KNH_STAGE_WR_MUL: return param == 0;
    case KNH_STAGE_MUL_ENV_ASR: case KNH_STAGE_MUL_ENV_AR: return param <= 1;
    case KNH_STAGE_SVF: return param <= 2;
    case KNH_STAGE_ONEPOLE_LPF: case KNH_STAGE_ONEPOLE_HPF: return param == 0;
    default: return false;
  }
}

struct HostEvent {
  uint32_t voice;
  uint32_t frame;
  uint32_t op;
  uint32_t slot;
  uint64_t bits;
};
struct QueuedChange {  // WrPreciseTiming::waiting_changes entry, precise_timing.rs:17
  uint16_t delay;
  uint32_t param, kind;
  double f;
  int64_t i;
};

}  // namespace
