// voice_dag.hpp -- five-role wave pipeline for chains of the shape
//     [source stages...] -> SvfFilter -> x * EnvAsr|EnvAr -> [post stages...]        (f32 banks)
//
// The linear pipeline of voice_pipe.hpp is bound by its heaviest wavefront, the SVF one (17 VALU + 2
// LDS per sample).  Two things in that wave are not part of the filter's recurrence: the output mix
// m0*v0 + m1*v1 + m2*v2 (5 of the 17 operations) and nothing else -- and the envelope, which does not
// depend on the signal at all, sits in series behind it.  Here the work is cut along the data
// dependences instead of along the chain:
//
//     wave 0  source   x            = source stages                    -> X  (3-deep ring)
//     wave 1  svf      v1, v2       = SVF recurrence on x (state only) -> V1, V2 (2-deep)
//     wave 2  env      e            = envelope generator               -> E  (2-deep)
//     wave 3  out      y            = (m0*x + m1*v1 + m2*v2) * e, post stages -> transposed mix tile
//     wave 4  mixer    per-frame left fold over the 64 voices          -> partials
//
// Tile t is produced by wave 0 in step t, by waves 1 and 2 in step t+1, consumed by wave 3 in step t+2
// and folded by wave 4 in step t+3; one workgroup barrier per step.  Waves 0 and 4 (the two lightest)
// share a SIMD.  Every voice executes exactly the operations of the single-wave kernel, in the same
// order, so the results are bit-identical to it (tests/test_gpu_properties.py).
//
// STATUS (round 1, MI355X): opt-in with KNH_PIPELINE=2.  Measured with the -DKNH_DAG_STAMPS build
// (tools/dag_stamps.py), busy cycles per 32-sample tile: source 1590, svf 2620, env 870-2130, out 2400,
// mixer 1690 -- the SVF wave drops from ~15 to 10 VALU per sample but pays for two 4-byte LDS stores per
// sample (a lone wave's ds_write_b32 costs ~8 cycles), and the out/env waves end up as heavy as it is.
// Net 21.3 us per C3 block against 20.2 us for the three-group pipeline, so that one stays the default.
#pragma once
#include "voice_pipe.hpp"

namespace knh_dev {

// SVF split along its data dependences (svf.rs:272-278).
struct SvfRec {  // slots 0..4 of the Svf stage: ic1eq, ic2eq, a1, a2, a3
  template <typename F> struct Regs { F ic1, ic2, a1, a2, a3; };
  template <typename F, typename W>
  static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long st) {
    r.ic1 = word_to_f<F>(s[0]); r.ic2 = word_to_f<F>(s[st]); r.a1 = word_to_f<F>(s[2 * st]);
    r.a2 = word_to_f<F>(s[3 * st]); r.a3 = word_to_f<F>(s[4 * st]);
  }
  template <typename F, typename W>
  static __device__ __forceinline__ void store(const Regs<F>& r, W* s, long st) {
    s[0] = f_to_word(r.ic1); s[st] = f_to_word(r.ic2);
  }
  // f32, exact arithmetic: the ten instructions of one tick in a hand-chosen order.  A wavefront that
  // is alone on its SIMD issues an instruction every 4 cycles but a *dependent* one only every ~8
  // (measured: the compiler's order, which puts each consumer right behind its producer, ran at 78
  // cycles per sample); here no instruction reads the result of the one directly before it, and the
  // only exposed dependence is the recurrence itself (sub -> mul -> add -> fma -> next sub).
  static __device__ __forceinline__ void tick_scheduled(Regs<float>& r, float v0, float& v1, float& v2) {
    float v3, t1, t2, t3, t4, u;
    asm volatile(
        "v_sub_f32 %[v3], %[v0], %[ic2]\n\t"
        "v_mul_f32 %[t1], %[a1], %[ic1]\n\t"
        "v_mul_f32 %[t2], %[a2], %[ic1]\n\t"
        "v_mul_f32 %[t4], %[a3], %[v3]\n\t"
        "v_add_f32 %[u], %[ic2], %[t2]\n\t"
        "v_mul_f32 %[t3], %[a2], %[v3]\n\t"
        "v_add_f32 %[v2], %[u], %[t4]\n\t"
        "v_add_f32 %[v1], %[t1], %[t3]\n\t"
        "v_fma_f32 %[ic2], %[v2], 2.0, -%[ic2]\n\t"
        "v_fma_f32 %[ic1], %[v1], 2.0, -%[ic1]"
        : [v3] "=&v"(v3), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [u] "=&v"(u), [v1] "=&v"(v1),
          [v2] "=&v"(v2), [ic1] "+v"(r.ic1), [ic2] "+v"(r.ic2)
        : [v0] "v"(v0), [a1] "v"(r.a1), [a2] "v"(r.a2), [a3] "v"(r.a3));
  }
  template <typename F, bool FMA>
  static __device__ __forceinline__ void tick(Regs<F>& r, F v0, F& v1, F& v2) {
    if constexpr (!FMA && sizeof(F) == 4) {
      tick_scheduled(r, v0, v1, v2);
      return;
    }
    const F v3 = v0 - r.ic2;
    if constexpr (FMA) {
      v1 = mad<true>(r.a2, v3, r.a1 * r.ic1);
      v2 = mad<true>(r.a3, v3, mad<true>(r.a2, r.ic1, r.ic2));
    } else {
      v1 = r.a1 * r.ic1 + r.a2 * v3;
      v2 = r.ic2 + r.a2 * r.ic1 + r.a3 * v3;
    }
    r.ic1 = mad<true>((F)2, v1, -r.ic1);  // exact: see Svf::tick
    r.ic2 = mad<true>((F)2, v2, -r.ic2);
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits) {
    if ((op & 0x7Fu) != EV_SET) return;
    F v = word_to_f<F>((typename WordOf<F>::type)bits);
    switch (rel) {
      case 0: r.ic1 = v; break; case 1: r.ic2 = v; break; case 2: r.a1 = v; break; case 3: r.a2 = v; break;
      default: r.a3 = v; break;
    }
  }
};
struct SvfOut {  // slots 5..7 of the Svf stage: m0, m1, m2
  template <typename F> struct Regs { F m0, m1, m2; };
  template <typename F, typename W>
  static __device__ __forceinline__ void load(Regs<F>& r, const W* s, long st) {
    r.m0 = word_to_f<F>(s[5 * st]); r.m1 = word_to_f<F>(s[6 * st]); r.m2 = word_to_f<F>(s[7 * st]);
  }
  template <typename F, bool FMA>
  static __device__ __forceinline__ F combine(const Regs<F>& r, F v0, F v1, F v2) {
    if constexpr (FMA) return mad<true>(r.m2, v2, mad<true>(r.m1, v1, r.m0 * v0));
    else return r.m0 * v0 + r.m1 * v1 + r.m2 * v2;
  }
  template <typename F>
  static __device__ __forceinline__ void on_event(Regs<F>& r, u32 op, u32 rel, u64 bits) {
    if ((op & 0x7Fu) != EV_SET) return;
    F v = word_to_f<F>((typename WordOf<F>::type)bits);
    if (rel == 5) r.m0 = v; else if (rel == 6) r.m1 = v; else r.m2 = v;
  }
};

template <typename F> struct DagLayout {
  static constexpr int T = 32;                   // samples per step
  static constexpr int TN = 32;                  // frames per mix tile
  static constexpr int TS = 68;
  static constexpr int tile = 64 * T;            // edge tile [T][64 lanes]: conflict-free 4-byte accesses, no padding
};
template <typename F, int T>
__device__ __forceinline__ void edge_read(const F* base, int lane, F (&x)[T]) {
#pragma unroll
  for (int j = 0; j < T; ++j) x[j] = base[j * 64 + lane];
}
template <typename F, int T>
__device__ __forceinline__ void edge_write(F* base, int lane, const F (&x)[T]) {
#pragma unroll
  for (int j = 0; j < T; ++j) base[j * 64 + lane] = x[j];
}

// Shared by every role: the voice's event cursor.  `apply(op, slot, bits, rel_frame)` is called for the
// events whose slot lies in [lo, hi); the owner also persists patched words.
template <typename F>
struct EventCursor {
  typedef typename WordOf<F>::type W;
  const VoiceKernelArgs<F>& a;
  u32 voice;
  bool live;
  u32 ev_i = 0, ev_end = 0, next_frame = 0xFFFFFFFFu;
  __device__ __forceinline__ EventCursor(const VoiceKernelArgs<F>& a_, u32 voice_, bool live_) : a(a_), voice(voice_), live(live_) {
    if (a.ev_start) { ev_i = a.ev_start[voice]; ev_end = a.ev_start[voice + 1]; }
    if (ev_i < ev_end) next_frame = a.events[ev_i].frame;
  }
  template <typename Fn>
  __device__ __forceinline__ void upto(u32 n_abs, u32 base, u32 lo, u32 hi, Fn&& apply) {
    while (next_frame <= n_abs) {
      Event e = a.events[ev_i];
      const u32 op = e.slot_op >> 24, slot = e.slot_op & 0xFFFFFFu;
      if (slot >= lo && slot < hi) {
        apply(op, slot, e.bits, e.frame - base);
        if (live && (op & 0x7Fu) == EV_SET) a.state[(long)slot * a.stride + voice] = (W)e.bits;
      }
      ++ev_i;
      next_frame = ev_i < ev_end ? a.events[ev_i].frame : 0xFFFFFFFFu;
    }
  }
};

// SRC / POST are Group<...> stage lists (POST may be Group<>); AR selects EnvAr instead of EnvAsr.
template <typename F, bool FMA, bool AR, typename SRC, typename POST>
__global__ void __launch_bounds__(320) voice_dag_kernel(VoiceKernelArgs<F> a) {
  typedef DagLayout<F> LY;
  typedef typename WordOf<F>::type W;
  constexpr int T = LY::T, TN = LY::TN, TS = LY::TS;
  constexpr int SRC_SLOTS = GroupInfo<SRC>::slots;
  constexpr int SVF_BASE = SRC_SLOTS, ENV_BASE = SVF_BASE + 8, POST_BASE = ENV_BASE + 5;
  constexpr bool kSine = GroupInfo<SRC>::uses_sine;
  typedef MulEnvT<AR> Env;
  __shared__ float sine[kSine ? 16384 : 1];
  __shared__ __attribute__((aligned(16))) F ring_x[3 * LY::tile];
  __shared__ __attribute__((aligned(16))) F ring_v1[2 * LY::tile];
  __shared__ __attribute__((aligned(16))) F ring_v2[2 * LY::tile];
  __shared__ __attribute__((aligned(16))) F ring_e[2 * LY::tile];
  __shared__ __attribute__((aligned(16))) F mix[2 * TN * TS];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  if (kSine) {
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
#pragma unroll 4
    for (int k = wave; k < 64; k += 5) {
      const float* g = a.sine_table + (k * 64 + lane) * 4;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(sine + k * 256), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();

  const u32 wave_global = blockIdx.x;
  const u32 v0 = wave_global * 64u;
  const u32 nv = a.n_voices - v0 < 64u ? a.n_voices - v0 : 64u;
  const bool live = (u32)lane < nv;
  const u32 voice = live ? v0 + lane : v0 + nv - 1;
  const u32 n_frames = a.frame_end - a.frame_begin;
  const int tpb = (int)((n_frames + T - 1) / T);
  const int qpb = (int)((n_frames + TN - 1) / TN);
  const int n_tiles = tpb * (int)a.n_blocks;
  const int n_steps = n_tiles + 3;
  const u32 n_waves_total = (a.n_voices + 63u) / 64u;
  Ctx ctx;
  ctx.ring_tile = nullptr;  // (RingLines: the whole-chain kernels)
  ctx.sine = sine;
  ctx.f2pi = a.f2pi;
  ctx.seg_table = a.seg_table;
  ctx.seg_max = a.seg_max;
  ctx.delay_ring = a.delay_ring;
  ctx.delay_stride = a.delay_stride;
  ctx.ring_sink_row = a.n_voices;
  ctx.buffer = a.buffer;
  ctx.buffer_frames = a.buffer_frames;

  // Each role walks the same (block, tile) sequence, `lag` steps behind wave 0.
  auto role_loop = [&](int lag, auto&& body, auto&& block_end) {
    int blk = 0, ti = 0;
    u32 base = 0;
#ifdef KNH_DAG_STAMPS  // diagnostic build only: cycles each role spends between barriers
    u64 busy = 0;
#endif
    for (int s = 0; s < n_steps; ++s) {
      const int g = s - lag;
      if (g >= 0 && g < n_tiles) {
#ifdef KNH_DAG_STAMPS
        const u64 t0 = __builtin_amdgcn_s_memtime();
#endif
        const u32 n = a.frame_begin + (u32)ti * T;
        const u32 m = a.frame_end - n < (u32)T ? a.frame_end - n : (u32)T;
        body(g, blk, ti, base, n, m);
        if (++ti == tpb) {
          block_end(base);
          ti = 0;
          ++blk;
          base += a.block_size;
        }
#ifdef KNH_DAG_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        busy += __builtin_amdgcn_s_memtime() - t0;
#endif
      }
      __syncthreads();
    }
#ifdef KNH_DAG_STAMPS
    if (blockIdx.x == 0 && lane == 0) a.flags[4 + wave] = (u32)(busy / (u64)(n_tiles > 0 ? n_tiles : 1));
#endif
  };

  if (wave == 0) {  // ---- source ------------------------------------------------------------------
    typedef typename GroupChain<F, FMA, 0, SRC>::type ChainT;
    ChainT chain;
    chain.load(a.state + voice, a.stride);
    EventCursor<F> ev(a, voice, live);
    auto on = [&](u32 op, u32 slot, u64 bits, u32 rel) { chain.on_event(op, slot, bits, rel); };
    role_loop(0,
              [&](int g, int, int ti, u32 base, u32 n, u32 m) {
                if (ti == 0) chain.begin_block(a.frame_begin, ctx);
                ev.upto(base + n, base, 0, SRC_SLOTS, on);
                F* xt = ring_x + (long)(g % 3) * LY::tile;
                if (m == (u32)T && !__builtin_amdgcn_ballot_w64(ev.next_frame < base + n + T)) {
                  F x[T];
#pragma unroll
                  for (int j = 0; j < T; ++j) x[j] = (F)0;
                  chain.template tick_tile<T>(x, ctx, n);
                  edge_write<F, T>(xt, lane, x);
                } else {  // sample by sample, straight through LDS (no register tile: keeps it out of scratch)
                  for (u32 j = 0; j < m; ++j) {
                    ev.upto(base + n + j, base, 0, SRC_SLOTS, on);
                    xt[j * 64 + lane] = chain.tick((F)0, ctx, n + j);
                  }
                }
              },
              [&](u32 base) { ev.upto(base + a.frame_end, base, 0, SRC_SLOTS, on); });
    if (live) chain.store(a.state + voice, a.stride);
  } else if (wave == 1) {  // ---- SVF recurrence ---------------------------------------------------
    SvfRec::Regs<F> r;
    SvfRec::load<F, W>(r, a.state + (long)SVF_BASE * a.stride + voice, a.stride);
    EventCursor<F> ev(a, voice, live);
    auto on = [&](u32 op, u32 slot, u64 bits, u32) { SvfRec::on_event<F>(r, op, slot - SVF_BASE, bits); };
    role_loop(1,
              [&](int g, int, int, u32 base, u32 n, u32 m) {
                ev.upto(base + n, base, SVF_BASE, SVF_BASE + 5, on);
                const F* xt = ring_x + (long)(g % 3) * LY::tile;
                F* v1t = ring_v1 + (long)(g & 1) * LY::tile;
                F* v2t = ring_v2 + (long)(g & 1) * LY::tile;
                if (m == (u32)T && !__builtin_amdgcn_ballot_w64(ev.next_frame < base + n + T)) {
                  F x[T], v1[T], v2[T];
                  edge_read<F, T>(xt, lane, x);
#pragma unroll
                  for (int j = 0; j < T; ++j) SvfRec::tick<F, FMA>(r, x[j], v1[j], v2[j]);
                  edge_write<F, T>(v1t, lane, v1);
                  edge_write<F, T>(v2t, lane, v2);
                } else {
                  for (u32 j = 0; j < m; ++j) {
                    ev.upto(base + n + j, base, SVF_BASE, SVF_BASE + 5, on);
                    F v1, v2;
                    SvfRec::tick<F, FMA>(r, xt[j * 64 + lane], v1, v2);
                    v1t[j * 64 + lane] = v1;
                    v2t[j * 64 + lane] = v2;
                  }
                }
              },
              [&](u32 base) { ev.upto(base + a.frame_end, base, SVF_BASE, SVF_BASE + 5, on); });
    if (live) SvfRec::store<F, W>(r, a.state + (long)SVF_BASE * a.stride + voice, a.stride);
  } else if (wave == 2) {  // ---- envelope generator -------------------------------------------------
    typename Env::template Regs<F> r;
    Env::template load<F, W>(r, a.state + (long)ENV_BASE * a.stride + voice, a.stride);
    EventCursor<F> ev(a, voice, live);
    u32 done_frame = 0xFFFFFFFFu;
    auto on = [&](u32 op, u32 slot, u64 bits, u32 rel) { Env::template on_event<F>(r, op, slot - ENV_BASE, bits, rel); };
    role_loop(1,
              [&](int g, int, int ti, u32 base, u32 n, u32 m) {
                if (ti == 0) r.seg = a.frame_begin;
                ev.upto(base + n, base, ENV_BASE, ENV_BASE + 5, on);
                F* et = ring_e + (long)(g & 1) * LY::tile;
                if (m == (u32)T && !__builtin_amdgcn_ballot_w64(ev.next_frame < base + n + T)) {
                  F e[T];
                  Env::template env_tile<F, T>(r, e, n, done_frame);
                  edge_write<F, T>(et, lane, e);
                } else {
                  for (u32 j = 0; j < m; ++j) {
                    ev.upto(base + n + j, base, ENV_BASE, ENV_BASE + 5, on);
                    et[j * 64 + lane] = Env::template env_next<F>(r, n + j, done_frame);
                  }
                }
              },
              [&](u32 base) { ev.upto(base + a.frame_end, base, ENV_BASE, ENV_BASE + 5, on); });
    if (live) {
      Env::template store<F, W>(r, a.state + (long)ENV_BASE * a.stride + voice, a.stride);
      a.done_frames[voice] = done_frame;
    }
    const u64 bd = __builtin_amdgcn_ballot_w64(live && done_frame != 0xFFFFFFFFu);
    const u64 br = __builtin_amdgcn_ballot_w64(live && r.state != 0u);
    if (lane == 0) {
      if (bd) atomicOr(&a.flags[0], 1u);
      if (br) atomicAdd(&a.flags[1], (u32)__builtin_popcountll(br));
    }
  } else if (wave == 3) {  // ---- output: filter mix, * envelope, post stages -------------------------
    typedef typename GroupChain<F, FMA, POST_BASE, POST>::type PostT;
    SvfOut::Regs<F> mo;
    SvfOut::load<F, W>(mo, a.state + (long)SVF_BASE * a.stride + voice, a.stride);
    PostT post;
    post.load(a.state + voice, a.stride);
    EventCursor<F> ev(a, voice, live);
    auto on = [&](u32 op, u32 slot, u64 bits, u32 rel) {
      if (slot < (u32)ENV_BASE) SvfOut::on_event<F>(mo, op, slot - SVF_BASE, bits);
      else post.on_event(op, slot, bits, rel);
    };
    // this wave owns two slot ranges: [SVF_BASE+5, SVF_BASE+8) and [POST_BASE, ...)
    auto events_upto = [&](u32 n_abs, u32 base) {
      while (ev.next_frame <= n_abs) {
        Event e = a.events[ev.ev_i];
        const u32 op = e.slot_op >> 24, slot = e.slot_op & 0xFFFFFFu;
        if ((slot >= (u32)SVF_BASE + 5 && slot < (u32)ENV_BASE) || slot >= (u32)POST_BASE) {
          on(op, slot, e.bits, e.frame - base);
          if (live && (op & 0x7Fu) == EV_SET) a.state[(long)slot * a.stride + voice] = (W)e.bits;
        }
        ++ev.ev_i;
        ev.next_frame = ev.ev_i < ev.ev_end ? a.events[ev.ev_i].frame : 0xFFFFFFFFu;
      }
    };
    role_loop(2,
              [&](int g, int blk, int ti, u32 base, u32 n, u32 m) {
                if (ti == 0) post.begin_block(a.frame_begin, ctx);
                events_upto(base + n, base);
                const F* xt = ring_x + (long)(g % 3) * LY::tile;
                const F* v1t = ring_v1 + (long)(g & 1) * LY::tile;
                const F* v2t = ring_v2 + (long)(g & 1) * LY::tile;
                const F* et = ring_e + (long)(g & 1) * LY::tile;
                const u32 rel = (u32)ti * T;
                const u32 qg = (u32)blk * (u32)qpb + rel / TN;
                F* out = mix + ((long)(qg & 1u) * TN + (rel % TN)) * TS + lane;
                if (m == (u32)T && !__builtin_amdgcn_ballot_w64(ev.next_frame < base + n + T)) {
                  F x[T], v1[T], v2[T], e[T];
                  edge_read<F, T>(xt, lane, x);
                  edge_read<F, T>(v1t, lane, v1);
                  edge_read<F, T>(v2t, lane, v2);
                  edge_read<F, T>(et, lane, e);
#pragma unroll
                  for (int j = 0; j < T; ++j) x[j] = SvfOut::combine<F, FMA>(mo, x[j], v1[j], v2[j]) * e[j];
                  post.template tick_tile<T>(x, ctx, n);
#pragma unroll
                  for (int j = 0; j < T; ++j) out[j * TS] = x[j];
                } else {
                  for (u32 j = 0; j < m; ++j) {
                    events_upto(base + n + j, base);
                    const F y = SvfOut::combine<F, FMA>(mo, xt[j * 64 + lane], v1t[j * 64 + lane], v2t[j * 64 + lane]) * et[j * 64 + lane];
                    out[j * TS] = post.tick(y, ctx, n + j);
                  }
                }
              },
              [&](u32 base) { events_upto(base + a.frame_end, base); });
    // post stages have no mutable state worth writing back except what store() decides
    if (live) post.store(a.state + voice, a.stride);
  } else {  // ---- mixer -------------------------------------------------------------------------------
    role_loop(3,
              [&](int, int blk, int ti, u32, u32, u32) {
                const u32 rel_end = (u32)(ti + 1) * T < n_frames ? (u32)(ti + 1) * T : n_frames;
                if (rel_end % TN == 0 || ti == tpb - 1) {
                  const u32 q = ((u32)ti * T) / TN;
                  const u32 qg = (u32)blk * (u32)qpb + q;
                  const u32 len = rel_end - q * TN;
                  const F* my = mix + (long)(qg & 1u) * TN * TS;
                  const u32 n0 = a.frame_begin + q * TN;
                  if ((u32)lane < len) {
                    const F* row = my + (long)lane * TS;
                    const F acc = fold_group<F, 16>(row, 1, nv);
                    a.partials[((long)blk * n_waves_total + wave_global) * a.block_size + n0 + lane] = acc;
                  }
                  if (a.voices_out) {
                    for (u32 v = 0; v < nv; ++v)
                      if ((u32)lane < len) a.voices_out[(long)(v0 + v) * a.block_size + n0 + lane] = my[(long)lane * TS + v];
                  }
                }
              },
              [&](u32) {});
  }
}

}  // namespace knh_dev
