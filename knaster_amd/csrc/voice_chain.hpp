// voice_chain.hpp -- the chain of stages of a voice and the fused voice-bank kernel (gfx950).
//
// One lane = one voice.  A chain is a compile-time list of stages (voice_stages.hpp) evaluated in
// order on one running sample x; all per-voice state lives in registers for the
// whole block and is read from / written back to a struct-of-arrays in HBM once
// per launch (coalesced: lane i touches word i of each slot row).
//
// Arithmetic contract: with FMA == false every a*b+c is a separate
// multiply and add in source order (the translation unit is also built with
// -ffp-contract=off), which makes each voice's signal bit-identical to the
// reference's scalar Rust.  Citations are file:line in the knaster repo.
//
// Self-contained like voice_stages.hpp (no libc/libstdc++ includes): the same text
// is handed to hiprtc for chains that are not pre-instantiated.
#pragma once
#include "voice_stages.hpp"

namespace knh_dev {

// ---------------------------------------------------------------------------
// Chain = compile-time stage list with all registers inline.
// ---------------------------------------------------------------------------
template <typename F, bool FMA, int BASE, typename... S> struct Chain;
template <typename F, bool FMA, int BASE>
struct Chain<F, FMA, BASE> {
  static constexpr int kSlots = BASE;
  static constexpr bool kUsesSine = false;
  static constexpr bool kPan = false;
  static constexpr u64 kParamBits = 0ull, kNopOkBits = 0ull;
  static constexpr bool kBinds = false;
  static constexpr bool kUsesRing = false;
  template <int T> __device__ __forceinline__ void tick_tile_sw(const Chain&, u32, u64, F (&)[T], const Ctx&, u32) {}
  __device__ __forceinline__ void pan_gains(F&, F&) const {}
  template <typename W> __device__ __forceinline__ void load(const W*, long) {}
  template <typename W> __device__ __forceinline__ void store(W*, long) const {}
  __device__ __forceinline__ F tick(F x, const Ctx&, u32) { return x; }
  template <int T> __device__ __forceinline__ void tick_tile(F (&)[T], const Ctx&, u32) {}
#ifdef KNH_DAG_STAMPS
  template <int T> __device__ __forceinline__ void tick_tile_stamped(F (&)[T], const Ctx&, u32, u64*, u64&) {}
#endif
  __device__ __forceinline__ void on_event(u32, u32, u64, u32) {}
  __device__ __forceinline__ bool last_env_stopped(bool dflt) const { return dflt; }
  __device__ __forceinline__ u32 collect_done(u32 acc) const { return acc; }
  __device__ __forceinline__ void reset_marks() {}
  __device__ __forceinline__ void begin_block(u32, const Ctx&) {}
};
template <typename F, bool FMA, int BASE, typename S0, typename... Rest>
struct Chain<F, FMA, BASE, S0, Rest...> {
  typedef Chain<F, FMA, BASE + S0::kSlots, Rest...> RestT;
  static constexpr int kSlots = RestT::kSlots;
  static constexpr bool kUsesSine = S0::kUsesSine || RestT::kUsesSine;
  static constexpr bool kPan = IsPan<S0>::value || RestT::kPan;  // the chain ends in a Pan2: two output channels per voice
  // Absolute slots (bit = slot index, chains of up to 64 slots) that are parameters in the sense of
  // StageDefaults::kParamMask, and stage base slots to which a split mark without a change (EV_NOP) may be addressed
  // in the middle of a tile (the stage either carries its partial-block origin along in take_params or has none).
  static constexpr u64 kStageBits = BASE + S0::kSlots <= 64 ? (((u64)1 << S0::kSlots) - 1) << (BASE < 64 ? BASE : 0) : 0ull;
  static constexpr u64 kParamBits = (BASE + S0::kSlots <= 64 ? (u64)S0::kParamMask << (BASE < 64 ? BASE : 0) : 0ull) | RestT::kParamBits;
  static constexpr u64 kNopOkBits = ((S0::kParamMask != 0u || !S0::kHasSeg) && BASE < 64 ? (u64)1 << (BASE < 64 ? BASE : 0) : 0ull) | RestT::kNopOkBits;
  static constexpr bool kBinds = S0::kNeedsBind || RestT::kBinds;  // a stage with memory behind it (delay ring, segment table, buffer)
  static constexpr bool kUsesRing = S0::kUsesRing || RestT::kUsesRing;
  typename S0::template Regs<F> r;
  // The frame this stage last passed to mark_done (UGenFlags::mark_done, ugen.rs:199-202), 0xFFFFFFFF: never.  The
  // reference hands one UGenFlags to every task of a block in node order (graph_gen.rs:196-200), so the mark a voice
  // ends up with is that of the last node in order that set one: collect_done.
  u32 mark = 0xFFFFFFFFu;
  RestT rest;
  template <typename W> __device__ __forceinline__ void load(const W* s, long stride) {
    S0::template load<F, W>(r, s + (long)BASE * stride, stride);
    rest.load(s, stride);
  }
  template <typename W> __device__ __forceinline__ void store(W* s, long stride) const {
    S0::template store<F, W>(r, s + (long)BASE * stride, stride);
    rest.store(s, stride);
  }
  __device__ __forceinline__ F tick(F x, const Ctx& c, u32 frame) {
    x = S0::template tick<F, FMA>(r, x, c, frame, mark);
    return rest.tick(x, c, frame);
  }
  template <int T> __device__ __forceinline__ void tick_tile(F (&x)[T], const Ctx& c, u32 frame0) {
    S0::template tick_tile<F, FMA, T>(r, x, c, frame0, mark);
    rest.template tick_tile<T>(x, c, frame0);
  }
#ifdef KNH_DAG_STAMPS  // diagnostic build only: the same walk with the clock read behind every stage (tools/wide_stamps.py)
  template <int T> __device__ __forceinline__ void tick_tile_stamped(F (&x)[T], const Ctx& c, u32 frame0, u64* acc, u64& t_prev) {
    S0::template tick_tile<F, FMA, T>(r, x, c, frame0, mark);
    asm volatile("" : "+v"(x[T - 1]) : : "memory");  // the stage's last sample exists before the clock is read
    const u64 t = __builtin_amdgcn_s_memtime();
    acc[0] += t - t_prev;
    t_prev = t;
    rest.template tick_tile_stamped<T>(x, c, frame0, acc + 1, t_prev);
  }
#endif
  // A tile in which voices change parameters at frames of their own: `cn` holds each voice's registers with its changes
  // applied, `sw` the tile-relative frame at which the voice takes them over (T: never), `touched` the slots its changes
  // address.  A stage none of whose slots is touched in the whole wave runs its ordinary tile code.
  template <int T> __device__ __forceinline__ void tick_tile_sw(const Chain& cn, u32 sw, u64 touched, F (&x)[T], const Ctx& c, u32 frame0) {
    // (sample by sample in every stage, also those no change addresses: this path is rare, and the kernels are better off
    // without a second, eight-sample copy of every stage's tile code -- the wavefronts of a pipeline share an instruction cache)
    bool hit = false;
    if constexpr (S0::kParamMask != 0u) hit = __builtin_amdgcn_ballot_w64((touched & kStageBits) != 0ull) != 0;
    if (hit) {
#pragma unroll
      for (int j = 0; j < T; ++j) {
        S0::take_params(r, cn.r, (u32)j == sw);
        x[j] = S0::template tick<F, FMA>(r, x[j], c, frame0 + j, mark);
      }
    } else {
#pragma unroll
      for (int j = 0; j < T; ++j) x[j] = S0::template tick<F, FMA>(r, x[j], c, frame0 + j, mark);
    }
    rest.template tick_tile_sw<T>(cn.rest, sw, touched, x, c, frame0);
  }
  __device__ __forceinline__ u32 collect_done(u32 acc) const {
    if constexpr (S0::kIsEnv) acc = mark != 0xFFFFFFFFu ? mark : acc;
    return rest.collect_done(acc);
  }
  // a resident launch reports the marks of each CALL (an ordinary launch: of the launch)
  __device__ __forceinline__ void reset_marks() { mark = 0xFFFFFFFFu; rest.reset_marks(); }
  __device__ __forceinline__ void pan_gains(F& l, F& rg) const {
    if constexpr (IsPan<S0>::value) { l = r.l; rg = r.r; }
    else rest.pan_gains(l, rg);
  }
  __device__ __forceinline__ void on_event(u32 op, u32 slot, u64 bits, u32 frame) {
    if (slot >= (u32)BASE && slot < (u32)(BASE + S0::kSlots)) S0::template on_event<F>(r, op, slot - BASE, bits, frame);
    else rest.on_event(op, slot, bits, frame);
  }
  __device__ __forceinline__ bool last_env_stopped(bool dflt) const {
    if constexpr (S0::kIsEnv) return rest.last_env_stopped(S0::template is_stopped<F>(r));
    else return rest.last_env_stopped(dflt);
  }
  __device__ __forceinline__ void begin_block(u32 frame_begin, const Ctx& c) {
    if constexpr (S0::kNeedsBind) S0::template bind<F>(r, c);
    if constexpr (S0::kIsEnv && S0::kHasSeg) r.seg = frame_begin;
    rest.begin_block(frame_begin, c);
  }
};

// ---------------------------------------------------------------------------
// Voices that are not a chain on one running signal but a small graph (DAG) of the same stages: `sine_a * sine_b`,
// `l * 440 + s * l` (knaster_benchmarks/benches/graph_dsp_performance.rs:37-72), one signal feeding two consumers.
// Every stage's output is a signal of its own, numbered by the stage's position; a stage reads the output of the stage
// before it unless it says otherwise (At<S, A>), and MathUGen<_, U1, Op> of two signals (math.rs:94-165,
// graph_edit.rs:936-971) is a stage with two operands (At<Math2<OP>, A, B>).  The voice's output is the last stage's.
// All of it unrolls into straight-line code on registers, like a chain; it runs in the single-wave kernel.
// ---------------------------------------------------------------------------
template <int OP>  // 0 mul, 1 add, 2 sub, 3 div, 6 pow (ValT's numbering)
struct Math2 : StageDefaults {
  static constexpr int kSlots = 0;
  static constexpr u32 kMutableMask = 0u;
  static constexpr bool kUsesSine = false;
  static constexpr bool kIsEnv = false;
  static constexpr bool kNeedsBind = false;
  static constexpr bool kHasSeg = false;
  static constexpr bool kBinary = true;
  template <typename F> struct Regs {};
  template <typename F, typename W> static __device__ __forceinline__ void load(Regs<F>&, const W*, long) {}
  template <typename F, typename W> static __device__ __forceinline__ void store(const Regs<F>&, W*, long) {}
  template <typename F> static __device__ __forceinline__ F apply(F a, F b) {  // math.rs:22-85: a op b
    if (OP == 0) return a * b;
    if (OP == 1) return a + b;
    if (OP == 2) return a - b;
    if (OP == 3) return a / b;
    return dev_pow(a, b);
  }
  template <typename F, bool FMA> static __device__ __forceinline__ F tick(Regs<F>&, F x, const Ctx&, u32, u32&) { return x; }
  template <typename F, bool FMA, int T> static __device__ __forceinline__ void tick_tile(Regs<F>&, F (&)[T], const Ctx&, u32, u32&) {}
  template <typename F> static __device__ __forceinline__ void on_event(Regs<F>&, u32, u32, u64, u32) {}
};
// A stage of a graph-shaped voice: the signal slots it reads (A, and B for Math2; -1: none -- a source) and writes (O).
// The host hands slots out like registers (bank.hip, build_signature): a slot is free again after its signal's last
// reader, and a stage whose operand dies with it writes in place.  Slots<R> at the end of the list: how many there are.
template <typename S, int A, int B, int O> struct At {};
template <int R> struct Slots {};
template <typename X> struct NodeOf { static constexpr bool dag = false; };
template <typename S, int A, int B, int O> struct NodeOf<At<S, A, B, O>> { typedef S stage; static constexpr int a = A, b = B, o = O; static constexpr bool dag = true; };
template <int R> struct NodeOf<Slots<R>> { static constexpr bool dag = true; };

template <typename F, bool FMA, int BASE, int R, int LAST, typename... Ns> struct DagChain;
// end of the list: Slots<R> (LAST = the slot of the last stage's output: the voice's signal)
template <typename F, bool FMA, int BASE, int R, int LAST, int R2>
struct DagChain<F, FMA, BASE, R, LAST, Slots<R2>> {
  static_assert(R == R2, "the slot count is the list's last entry");
  static constexpr int kSlots = BASE;
  static constexpr bool kUsesSine = false;
  static constexpr bool kPan = false;
  static constexpr int kOut = LAST;
  static constexpr bool kUsesRing = false;
  __device__ __forceinline__ void pan_gains(F&, F&) const {}
  template <typename W> __device__ __forceinline__ void load(const W*, long) {}
  template <typename W> __device__ __forceinline__ void store(W*, long) const {}
  template <int T> __device__ __forceinline__ void run_tile(F (&)[R][T], const Ctx&, u32) {}
  __device__ __forceinline__ void run_one(F (&)[R], const Ctx&, u32) {}
  __device__ __forceinline__ void on_event(u32, u32, u64, u32) {}
  __device__ __forceinline__ bool last_env_stopped(bool dflt) const { return dflt; }
  __device__ __forceinline__ u32 collect_done(u32 acc) const { return acc; }
  __device__ __forceinline__ void reset_marks() {}
  template <int J> __device__ __forceinline__ void collect_done_ranked(u64, u32&, u32&) const {}
  __device__ __forceinline__ void begin_block(u32, const Ctx&) {}
};
template <typename F, bool FMA, int BASE, int R, int LAST, typename N0, typename... Rest>
struct DagChain<F, FMA, BASE, R, LAST, N0, Rest...> {
  typedef typename NodeOf<N0>::stage S0;
  static constexpr int A = NodeOf<N0>::a, B = NodeOf<N0>::b, O = NodeOf<N0>::o;
  static_assert(A < R && B < R && O >= 0 && O < R, "signal slots are 0 .. R-1");
  static_assert(!S0::kBinary || (A >= 0 && B >= 0), "a MathUGen of two signals reads both");
  static_assert(S0::kArParam < 0 || B >= 0, "an audio-rate parameter names the signal that drives it");
  typedef DagChain<F, FMA, BASE + S0::kSlots, R, O, Rest...> RestT;
  static constexpr int kSlots = RestT::kSlots;
  static constexpr bool kUsesSine = S0::kUsesSine || RestT::kUsesSine;
  static constexpr bool kPan = IsPan<S0>::value || RestT::kPan;
  static constexpr int kOut = RestT::kOut;
  static constexpr bool kUsesRing = S0::kUsesRing || RestT::kUsesRing;
  typename S0::template Regs<F> r;
  u32 mark = 0xFFFFFFFFu;
  RestT rest;
  template <typename W> __device__ __forceinline__ void load(const W* s, long stride) {
    S0::template load<F, W>(r, s + (long)BASE * stride, stride);
    rest.load(s, stride);
  }
  template <typename W> __device__ __forceinline__ void store(W* s, long stride) const {
    S0::template store<F, W>(r, s + (long)BASE * stride, stride);
    rest.store(s, stride);
  }
  template <int T> __device__ __forceinline__ void run_tile(F (&sig)[R][T], const Ctx& c, u32 frame0) {
    if constexpr (S0::kBinary) {
#pragma unroll
      for (int j = 0; j < T; ++j) sig[O][j] = S0::template apply<F>(sig[A][j], sig[B][j]);
    } else if constexpr (S0::kArParam >= 0) {
      // WrArParams::process (audio_rate.rs:42-57): param_apply(p, buffer[i]), then one sample of the node -- frame by frame
#pragma unroll
      for (int j = 0; j < T; ++j) sig[O][j] = ar_one(sig[A >= 0 ? A : 0][j], sig[B >= 0 ? B : 0][j], c, frame0 + (u32)j);
    } else {
      if constexpr (A != O) {
#pragma unroll
        for (int j = 0; j < T; ++j) sig[O][j] = A >= 0 ? sig[A >= 0 ? A : 0][j] : (F)0;
      }
      S0::template tick_tile<F, FMA, T>(r, sig[O], c, frame0, mark);
    }
    rest.template run_tile<T>(sig, c, frame0);
  }
  // one sample of a node whose parameter kArParam a second signal drives (b = that signal's sample)
  __device__ __forceinline__ F ar_one(F a, F b, const Ctx& c, u32 frame) {
    S0::template ar_set<F, (S0::kArParam >= 0 ? S0::kArParam : 0)>(r, b, c);
    // the wrapper runs the node sample by sample through UGen::process, where an envelope marks done at frame 0 of its
    // one-sample "block" (envelopes.rs:153-156 / :285-288: next_sample(flags, 0)): the mark carries no in-block frame
    if constexpr (S0::kIsEnv && S0::kHasSeg) frame = r.seg;
    return S0::template tick<F, FMA>(r, A >= 0 ? a : (F)0, c, frame, mark);
  }
  __device__ __forceinline__ void run_one(F (&sig)[R], const Ctx& c, u32 frame) {
    if constexpr (S0::kBinary) sig[O] = S0::template apply<F>(sig[A], sig[B]);
    else if constexpr (S0::kArParam >= 0) sig[O] = ar_one(sig[A >= 0 ? A : 0], sig[B >= 0 ? B : 0], c, frame);
    else sig[O] = S0::template tick<F, FMA>(r, A >= 0 ? sig[A >= 0 ? A : 0] : (F)0, c, frame, mark);
    rest.run_one(sig, c, frame);
  }
  // the kernel's view: a voice's tile / sample is the last stage's signal (called on the head of the list)
  template <int T> __device__ __forceinline__ void tick_tile(F (&x)[T], const Ctx& c, u32 frame0) {
    F sig[R][T];
    run_tile<T>(sig, c, frame0);
#pragma unroll
    for (int j = 0; j < T; ++j) x[j] = sig[kOut][j];
  }
  __device__ __forceinline__ F tick(F, const Ctx& c, u32 frame) {
    F sig[R];
    run_one(sig, c, frame);
    return sig[kOut];
  }
  __device__ __forceinline__ u32 collect_done(u32 acc) const {
    if constexpr (S0::kIsEnv) acc = mark != 0xFFFFFFFFu ? mark : acc;
    return rest.collect_done(acc);
  }
  __device__ __forceinline__ void reset_marks() { mark = 0xFFFFFFFFu; rest.reset_marks(); }
  // The same with the envelopes taken in the reference's task order (one UGenFlags for all tasks of a graph: the last
  // mark_done wins, graph_gen.rs:196-200): J = this stage's number among the envelope stages, ranks = VoiceKernelArgs::env_ranks.
  template <int J> __device__ __forceinline__ void collect_done_ranked(u64 ranks, u32& best_rank, u32& best) const {
    if constexpr (S0::kIsEnv) {
      const u32 rk = J < 16 ? (u32)((ranks >> (4 * (J < 16 ? J : 0))) & 15ull) : 0u;
      if (mark != 0xFFFFFFFFu && (best == 0xFFFFFFFFu || rk > best_rank)) { best = mark; best_rank = rk; }
      rest.template collect_done_ranked<J + 1>(ranks, best_rank, best);
    } else {
      rest.template collect_done_ranked<J>(ranks, best_rank, best);
    }
  }
  __device__ __forceinline__ void pan_gains(F& l, F& rg) const {
    if constexpr (IsPan<S0>::value) { l = r.l; rg = r.r; }
    else rest.pan_gains(l, rg);
  }
  __device__ __forceinline__ void on_event(u32 op, u32 slot, u64 bits, u32 frame) {
    if (slot >= (u32)BASE && slot < (u32)(BASE + S0::kSlots)) S0::template on_event<F>(r, op, slot - BASE, bits, frame);
    else rest.on_event(op, slot, bits, frame);
  }
  __device__ __forceinline__ bool last_env_stopped(bool dflt) const {
    if constexpr (S0::kIsEnv) return rest.last_env_stopped(S0::template is_stopped<F>(r));
    else return rest.last_env_stopped(dflt);
  }
  __device__ __forceinline__ void begin_block(u32 frame_begin, const Ctx& c) {
    if constexpr (S0::kNeedsBind) S0::template bind<F>(r, c);
    if constexpr (S0::kIsEnv && S0::kHasSeg) r.seg = frame_begin;
    rest.begin_block(frame_begin, c);
  }
};
// the slot count of a graph-shaped stage list (its last entry), 0 for a plain chain
template <typename... S> struct SlotCount { static constexpr int value = 0; };
template <typename S0, typename... S> struct SlotCount<S0, S...> { static constexpr int value = SlotCount<S...>::value; };
template <int R> struct SlotCount<Slots<R>> { static constexpr int value = R; };
// the kernel's chain type: a plain chain unless the list is a graph's
template <bool DAG, typename F, bool FMA, typename... S> struct ChainSelect { typedef Chain<F, FMA, 0, S...> type; };
template <typename F, bool FMA, typename... S> struct ChainSelect<true, F, FMA, S...> { typedef DagChain<F, FMA, 0, SlotCount<S...>::value, 0, S...> type; };

// ---------------------------------------------------------------------------
// Kernel arguments
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// A RESIDENT launch (round 4): the call the reference makes is UGen::process_block once per block (Task::run,
// knaster_graph/src/task.rs:25-31).  As one launch per call that costs a kernel launch, the sine table's staging, the state's
// trip from HBM and back and the pipeline's fill -- 22 us of kernel for 11 us of work, plus a second launch for the fold
// (profiles/r03_per_block_twin_c3.json).  A resident launch stays on its CUs between calls: the state stays in registers, the
// table in LDS, and a call is a COMMAND WORD the host stores and the kernel picks up:
//   bits 0-23 epoch (the kernel expects them in order), 24-39 frame_begin, 40-55 frame_end, 56 the block has host-made events,
//   57 which of the two event lists, 58 leave (store the state and end).
// Workgroup 0 polls the word the host writes and republishes it in device memory for the others (one reader across PCIe / the
// fabric instead of 256: profiles/r04_micro_doorbell.txt).  Every wait is bounded by the constant-rate clock: a kernel whose
// host has gone quiet for `idle_ticks` ends by itself -- the host, finding the stream idle, launches again.
// The mix leaves the device without a second launch too: every workgroup stores its partial row of a 64-frame tile as granules
// (below) and goes on -- it never waits for anybody; a second, small resident kernel (res_fold_server, one wavefront per 32
// rows and one for the root) watches the granules arrive, continues the bank's pairwise sum over them (32 rows, then up to 8
// of those: the same binary tree as fold_tree_kernel, same bits) and writes each tile into mapped pinned host memory, the
// call's flags and its epoch behind the last one.
// ---------------------------------------------------------------------------
struct Resident {
  const u64* bell;          // the command word the host stores (device memory it reaches through the BAR, or mapped pinned host memory); null: an ordinary launch
  u64* relay;               // device: the command as workgroup 0 republishes it
  u64* rows;                // device: the workgroups' partial rows as granules, [tile][plane][workgroup][64 frames][W] (W = 1, f64: 2)
  u64* wg_flags;            // device: [workgroup] one granule: voices that marked done | voices still running << 8
  const u32* ev_start[2];   // the two alternating host-made event lists (pinned host memory)
  const Event* events[2];
  u64 idle_ticks;           // workgroup 0's patience without a command (s_memrealtime ticks, 10 ns)
  u32* host_started;        // mapped pinned host word: workgroup 0 stores first_epoch there when it starts (the host checks that the
                            // voice kernel and the fold server really run side by side before it relies on them)
  u32 first_epoch;          // the epoch of the first command this launch takes
  u32 max_tiles;
  u32 bell_is_device;       // the command word lives in device memory (the host writes it through the BAR): every workgroup reads it
                            // there, a fabric read each; 0: it is in host memory, and only workgroup 0 reads it (256 readers across
                            // PCIe take 47 us to see a word: profiles/r04_micro_doorbell.txt) and republishes it in `relay`
};
// A GRANULE is one naturally aligned 8-byte word {32 bits of data, 32-bit tag} written by ONE store and read by ONE load
// (8-byte device-scope atomics on both sides): whoever reads the tag it is waiting for has the data that was stored with it --
// no flag, no counter, no fence, one trip through the memory system (MI355X_MICROARCH.md, hand-off price list: "handoff-1to1").
// tag = epoch << 8 | tile (255: the call's flag granules).  An f32 sample is one granule, an f64 sample two (low and high word).
__device__ __forceinline__ u32 res_tag(u32 epoch, u32 tile) { return (epoch << 8) | (tile & 0xFFu); }
__device__ __forceinline__ void res_put(u64* g, u32 data, u32 tag) { __hip_atomic_store(g, (u64)data | ((u64)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 res_get(const u64* g) { return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ... and the same word stored to mapped host memory (the root of the fold server: the host reads a frame's sum the moment its tag is there)
__device__ __forceinline__ void res_put_host(u64* g, u32 data, u32 tag) { __hip_atomic_store(g, (u64)data | ((u64)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void res_put_host_sample(u64* g, float v, u32 tag) { res_put_host(g, __builtin_bit_cast(u32, v), tag); }
__device__ __forceinline__ void res_put_host_sample(u64* g, double v, u32 tag) {
  const u64 b = __builtin_bit_cast(u64, v);
  res_put_host(g, (u32)b, tag);
  res_put_host(g + 1, (u32)(b >> 32), tag);
}
template <typename F> struct ResWords { static constexpr int value = sizeof(F) == 8 ? 2 : 1; };
__device__ __forceinline__ void res_put_sample(u64* g, float v, u32 tag) { res_put(g, __builtin_bit_cast(u32, v), tag); }
__device__ __forceinline__ void res_put_sample(u64* g, double v, u32 tag) {
  const u64 b = __builtin_bit_cast(u64, v);
  res_put(g, (u32)b, tag);
  res_put(g + 1, (u32)(b >> 32), tag);
}
enum : u32 { RES_RELAY_RANGES = 16 };  // the relay buffer: word 0 the command, words 16 .. 16 + 4 * 15 a call's range events (below)
enum : u64 { RES_EPOCH_MASK = 0xFFFFFFull, RES_HAS_EVENTS = 1ull << 56, RES_LIST = 1ull << 57, RES_LEAVE = 1ull << 58, RES_RANGES_SHIFT = 59, RES_RANGES_MASK = 0xFull };
// A call's events come as per-voice lists (ev_start + events, sorted by voice) or -- bits 59..62 = n > 0 -- as n RANGE events:
// the same change for every voice of [v_begin, v_end), a whole bank's note-on in 32 bytes instead of 16 bytes per voice
// fetched over PCIe.  A range event is two Event-sized records at events[2 j], events[2 j + 1] of the call's list:
// {frame, slot_op, bits} and {v_begin, v_end, -}; every voice walks the n of them in order and takes those that cover it.
struct ResCall {  // one command, unpacked
  u32 epoch, frame_begin, frame_end;
  bool has_events, leave;
  u32 list;
  u32 n_ranges;
};
__device__ __forceinline__ ResCall res_unpack(u64 c) {
  ResCall r;
  r.epoch = (u32)(c & RES_EPOCH_MASK);
  r.frame_begin = (u32)((c >> 24) & 0xFFFFu);
  r.frame_end = (u32)((c >> 40) & 0xFFFFu);
  r.has_events = (c & RES_HAS_EVENTS) != 0ull;
  r.list = (c & RES_LIST) ? 1u : 0u;
  r.leave = (c & RES_LEAVE) != 0ull;
  r.n_ranges = (u32)((c >> RES_RANGES_SHIFT) & RES_RANGES_MASK);
  return r;
}
// A call's events into LDS: the workgroup's piece of the call's list (its voices are neighbours, the list is sorted by voice),
// fetched from pinned host memory by all threads together, 16 bytes each -- a lane that fetched its own events one by one paid
// three PCIe round trips per event, and 16 384 lanes doing so at once took 100 us over it.  Every wavefront of the workgroup
// calls it (it ends in a barrier); `count` 0: nothing staged (no events, or more than fit), the list is read where it is.
struct ResStaged { u32 first, count; };
__device__ __forceinline__ ResStaged res_stage_events(const Resident& r, const ResCall& call, Event* stage, u32 cap, u32 gv0, u32 gnv, u32 tid, u32 n_threads) {
  ResStaged st{0u, 0u};
  if (!call.has_events || cap == 0u) return st;  // uniform
  if (call.n_ranges) {  // range events: the same 2 n records for every workgroup -- workgroup 0 has put them beside the relay
    st.count = 2u * call.n_ranges;
    if (st.count > cap) { st.count = 0u; return st; }  // (the voices then read them where the host left them)
    const u64* src = r.relay + RES_RELAY_RANGES;
    u64* dst = reinterpret_cast<u64*>(stage);
    for (u32 i = tid; i < 2u * st.count; i += n_threads) dst[i] = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return st;
  } else {
    const u32* evs = r.ev_start[call.list];
    st.first = __hip_atomic_load(&evs[gv0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    st.count = __hip_atomic_load(&evs[gv0 + gnv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - st.first;
  }
  if (st.count > cap) st.count = 0u;
  // The list is in host memory the host has rewritten since this kernel started, so the loads must not be served by this
  // CU's L1: VOLATILE 16-byte loads (system-scope cache bypass in the instruction itself, neighbouring lanes neighbouring
  // events).  As 8-byte atomic loads, which nothing can merge, a bank's worth of single events was 65 536 reads over PCIe,
  // 40 us of a call; a system-scope acquire fence in front of plain loads cost every call with events 7 us.
  typedef u32 U4 __attribute__((ext_vector_type(4)));
  const volatile U4* src = reinterpret_cast<const volatile U4*>(r.events[call.list] + st.first);
  U4* dst = reinterpret_cast<U4*>(stage);
  for (u32 i = tid; i < st.count; i += n_threads) { const U4 e = src[i]; dst[i] = e; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return st;
}
// Every wavefront of the workgroup calls this between two calls; `slot` = two words of LDS.  Wavefront 0's lane 0 waits for
// the command with epoch `expect` (workgroup 0: from the host's word, and passes it on; the others: from the relay), bounded.
__device__ __forceinline__ ResCall res_wait(const Resident& r, u32 expect, u32* slot, int wave_all, int lane) {
  if (wave_all == 0) {
    const bool leader = blockIdx.x == 0u;
    u64 c = 0ull;
    if (lane == 0) {
      if (leader && expect == r.first_epoch) __hip_atomic_store(r.host_started, r.first_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const u64 t0 = __builtin_amdgcn_s_memrealtime();
      const u64 patience = leader ? r.idle_ticks : 2ull * r.idle_ticks + 5000000ull;  // (the others outwait workgroup 0: they hear of its leaving through the relay)
      const bool from_bell = leader || r.bell_is_device != 0u;
      for (;;) {
        c = from_bell ? __hip_atomic_load(r.bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(r.relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((u32)(c & RES_EPOCH_MASK) == expect) break;
        if (!leader && from_bell) {  // workgroup 0 gave up waiting and said so (the only word of its leaving the others get)
          const u64 l = __hip_atomic_load(r.relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((l & RES_LEAVE) && (u32)(l & RES_EPOCH_MASK) == expect) { c = l; break; }
        }
        if (__builtin_amdgcn_s_memrealtime() - t0 > patience) { c = RES_LEAVE | (u64)expect; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    if (leader) {
      // A call's range events travel with the command: the whole wavefront fetches them from the host's list (a few 8-byte
      // reads over PCIe, side by side) and leaves them beside the relay, BEFORE the command appears there -- 256 workgroups
      // fetching the same 32 bytes from host memory themselves took 47 us over it (the doorbell's lesson once more).
      const u32 lo = __builtin_amdgcn_readfirstlane((u32)c), hi = __builtin_amdgcn_readfirstlane((u32)(c >> 32));
      const u64 cw = (u64)lo | ((u64)hi << 32);
      const ResCall call = res_unpack(cw);
      if (!call.leave && call.has_events && call.n_ranges) {
        const u64* src = reinterpret_cast<const u64*>(r.events[call.list]);
        if ((u32)lane < 4u * call.n_ranges)
          __hip_atomic_store(r.relay + RES_RELAY_RANGES + lane, __hip_atomic_load(&src[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      }
      if (lane == 0) {
        __hip_atomic_store(r.relay, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (diagnostics, knh_bank_resident_trace: when the command was seen, on the device's 100 MHz clock)
        __hip_atomic_store(reinterpret_cast<u64*>(r.host_started + 4), __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (lane == 0) {
      slot[0] = (u32)c;
      slot[1] = (u32)(c >> 32);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const u64 c = (u64)slot[0] | ((u64)slot[1] << 32);
  return res_unpack(c);
}
template <typename F>
struct VoiceKernelArgs {
  typename WordOf<F>::type* state;  // [n_slots][stride]
  long stride;                      // words per slot row (>= n_voices, multiple of 64)
  u32 n_voices;
  u32 block_size;                   // row length of partials / voices_out
  u32 n_blocks;                     // consecutive blocks processed by this launch (state stays in registers)
  u32 frame_begin, frame_end;       // frames [begin, end) of each block are processed ([0, block_size) unless n_blocks == 1)
  const float* sine_table;          // 16384 floats in HBM (staged to LDS)
  double f2pi;
  u32 sample_rate;                  // for the setters that run on the device (audio-rate parameters, ArP)
  const double* seg_table;          // segment Envelope table [n_voices][seg_max][3], or null
  u32 seg_max;
  void* delay_ring;                 // SampleDelay rings [n_voices][delay_stride] of F, or null
  u32 delay_stride;
  const void* buffer;               // BufferReader's shared Buffer (single channel, F), or null
  u32 buffer_frames;
  const void* input;                // the bank node's input channels, [n_blocks][in_channels][block_size] of F, or null
  u32 in_channels;
  const u32* ev_start;              // [n_voices + 1] or null when the block has no events
  const Event* events;
  F* partials;                      // [n_blocks][n_waves][block_size]: per-wavefront left-fold of its voices
                                    // (chains ending in Pan2: [n_blocks][2][n_waves][block_size], left and right)
  F* voices_out;                    // [n_voices][block_size] or null (n_blocks == 1 only); Pan2 chains: [2][n_voices][block_size]
  u64 env_ranks;                    // a graph-shaped voice with several envelope stages: nibble j = the place of the j-th of them (list
                                    // order) in the reference's task order (graph.rs calculate_node_order); 0: list order is task order
  u32* done_frames;                 // [n_voices]
  u32* flags;                       // [0] |= any-done, [1] += voices whose last envelope is not Stopped
  Resident res;                     // res.bell != null: a resident launch (one block per command; frame_begin / frame_end / ev_start / events come with each command)
};

constexpr int kWave = 64;
constexpr int kTile = 8;  // samples evaluated stage-by-stage in registers

// ---------------------------------------------------------------------------
// KNH_MIX_TREE: the voices of a bank are summed pairwise, in voice order -- a binary tree over the voice index:
//   node(0, v) = voice v;  node(L, i) = node(L-1, 2i) + node(L-1, 2i+1), or node(L-1, 2i) alone where 2i+1 does not exist.
// The order of the additions depends on the number of voices only, not on the kernel form.  A wavefront folds the subtree
// of its own 64 voices and fold_tree_kernel continues from those nodes upwards.
// ---------------------------------------------------------------------------
// t[0..N): nodes of one level, each standing for `unit` voices, the first of them voice 0 of the subtree; nv = voices that
// exist in it.  FULL: all of them do.
template <typename F, int N, bool FULL>
__device__ __forceinline__ F tree_reduce(F (&t)[N], u32 unit, u32 nv) {
  static_assert((N & (N - 1)) == 0, "a power of two");
#pragma unroll
  for (int w = N, step = 1; w > 1; w >>= 1, step <<= 1) {
#pragma unroll
    for (int i = 0; i < w / 2; ++i) {
      const F sum = t[2 * i] + t[2 * i + 1];
      t[i] = (FULL || (u32)((2 * i + 1) * step) * unit < nv) ? sum : t[2 * i];  // the right child exists: its first voice does
    }
  }
  return t[0];
}
// the subtree over col[0], col[st], .. col[(N - 1) st]; the values are read B at a time (B = N: all reads ahead of all adds)
template <typename F, int N, bool FULL, int B>
__device__ __forceinline__ F tree_cols(const F* col, int st, u32 nv) {
  constexpr int BB = B < N ? B : N;
  F node[N / BB];
#pragma unroll
  for (int c = 0; c < N / BB; ++c) {
    F t[BB];
#pragma unroll
    for (int k = 0; k < BB; ++k) t[k] = col[(c * BB + k) * st];
    node[c] = tree_reduce<F, BB, FULL>(t, 1u, nv > (u32)(c * BB) ? nv - (u32)(c * BB) : 0u);
  }
  return tree_reduce<F, N / BB, FULL>(node, (u32)BB, nv);
}
// Pan2: the subtrees over col[k st] * gl[k gst] and over col[k st] * gr[k gst] (pan.rs:36: the product is rounded, then summed)
template <typename F, int N, bool FULL, int B>
__device__ __forceinline__ void tree_cols_pan(const F* col, int st, const F* gl, const F* gr, int gst, u32 nv, F& left, F& right) {
  constexpr int BB = B < N ? B : N;
  F nl[N / BB], nr[N / BB];
#pragma unroll
  for (int c = 0; c < N / BB; ++c) {
    F t[BB], p[BB];
#pragma unroll
    for (int k = 0; k < BB; ++k) t[k] = col[(c * BB + k) * st];
    const u32 sub = nv > (u32)(c * BB) ? nv - (u32)(c * BB) : 0u;
#pragma unroll
    for (int k = 0; k < BB; ++k) p[k] = t[k] * gl[(c * BB + k) * gst];
    nl[c] = tree_reduce<F, BB, FULL>(p, 1u, sub);
#pragma unroll
    for (int k = 0; k < BB; ++k) p[k] = t[k] * gr[(c * BB + k) * gst];
    nr[c] = tree_reduce<F, BB, FULL>(p, 1u, sub);
  }
  left = tree_reduce<F, N / BB, FULL>(nl, (u32)BB, nv);
  right = tree_reduce<F, N / BB, FULL>(nr, (u32)BB, nv);
}
// a wavefront's 64 voices, nv of them live
template <typename F, int B>
__device__ __forceinline__ F fold_group(const F* col, int st, u32 nv) {
  return nv == 64u ? tree_cols<F, 64, true, B>(col, st, nv) : tree_cols<F, 64, false, B>(col, st, nv);
}
template <typename F, int B>
__device__ __forceinline__ void fold_group_pan(const F* col, int st, const F* gl, const F* gr, int gst, u32 nv, F& l, F& r) {
  if (nv == 64u) tree_cols_pan<F, 64, true, B>(col, st, gl, gr, gst, nv, l, r);
  else tree_cols_pan<F, 64, false, B>(col, st, gl, gr, gst, nv, l, r);
}

// ---------------------------------------------------------------------------
// A resident launch's mix (see Resident): the fold server.  Workgroup k < n_groups folds rows [32 k, 32 k + 32) of every tile
// (lane = frame of the tile): it reads the 32 granules of its frame until all carry the tag of (call, tile), sums them as
// tree_reduce does, and passes the node on as a granule of its own -- or, alone (a bank of up to 32 voice groups), writes the
// host's block itself.  Workgroup n_groups is the root: up to 8 group granules per frame -> the host's block, and behind a
// call's last tile the flags and the epoch.  Waits are polls with a clock-bounded patience like the voice kernel's; the
// command comes from the same relay word.
// ---------------------------------------------------------------------------
template <typename F>
struct ResServerArgs {
  const u64* relay;
  const u64* rows;       // [tile][plane][row][64][W]
  const u64* wg_flags;   // [row]
  u64* group_rows;       // [tile][plane][8][64][W]
  u64* group_flags;      // [8]
  u64* host_out;         // mapped pinned host memory: the block as granules, [plane][block_size][W], then one granule for the call's
                         // done / running counts: every frame's sum carries its own tag, so the host needs no "everything is there"
                         // word behind them -- and the device no wait for its stores between the last tile and such a word (1.5 us of a call)
  const u64* bell;       // the command word itself when it lives in device memory (Resident::bell_is_device), else null: the relay is read
  u32* host_done;        // mapped pinned: [0] epoch of the last finished call, [1] voices that marked done, [2] voices still running,
                         // [5] first_epoch once the server's first workgroup runs; [8..9] the voice kernel saw the call's command,
                         // [10..11] the root did, [12..13] tile 0's rows (root: nodes) had all arrived, [14..15] the last tile's,
                         // [16..17] the root had written the last tile and the flags (device clock, 10 ns)
  u64 idle_ticks;
  u32 first_epoch, n_rows, planes, out_channels, block_size, tile_frames;
};
template <typename F> __device__ __forceinline__ bool res_read_sample(const u64* g, u32 tag, F& out);
template <> __device__ __forceinline__ bool res_read_sample<float>(const u64* g, u32 tag, float& out) {
  const u64 w = res_get(g);
  out = __builtin_bit_cast(float, (u32)w);
  return (u32)(w >> 32) == tag;
}
template <> __device__ __forceinline__ bool res_read_sample<double>(const u64* g, u32 tag, double& out) {
  const u64 lo = res_get(g), hi = res_get(g + 1);
  out = __builtin_bit_cast(double, (u64)(u32)lo | ((u64)(u32)hi << 32));
  return (u32)(lo >> 32) == tag && (u32)(hi >> 32) == tag;
}
// Four wavefronts per workgroup, wavefront w taking tiles w, w + 4, ..: a tile costs its folder two trips through the memory
// system (the probe that finds it complete, the read of its 32 granules per frame, all in flight together) -- about 3 us --
// and the voice kernel turns one out every 1.4.  At most 168 registers each, so that a server wavefront fits on a SIMD BESIDE a
// voice wavefront, which takes 300 of the 512 (with the compiler's free choice a first version took 247, and the CUs it sat
// on had no room for their voice workgroup).
template <typename F>
__global__ void __launch_bounds__(256, 3) res_fold_server(ResServerArgs<F> a) {
  constexpr int W = ResWords<F>::value;
  constexpr u32 NW = 4u;
  const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const u32 n_g = (a.n_rows + 31u) / 32u;
  const bool root = blockIdx.x == n_g;
  const bool alone = n_g == 1u;  // one group: its folder is the root
  if (root && alone) return;
  const bool writes_host = root || alone;
  const u32 g = blockIdx.x;
  const u32 in_g = root ? n_g : (a.n_rows - g * 32u < 32u ? a.n_rows - g * 32u : 32u);
  u32 expect = a.first_epoch;
  const u64 patience = 2ull * a.idle_ticks + 5000000ull;
  if (blockIdx.x == 0u && threadIdx.x == 0u) __hip_atomic_store(&a.host_done[5], a.first_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __builtin_amdgcn_s_setprio(0);  // (the voice wavefronts raise theirs: a server wavefront shares its SIMD with one of them and, launched first, is the older)
  for (;;) {
    // the call
    u64 c;
    {
      const u64 t0 = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        c = a.bell ? __hip_atomic_load(a.bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(a.relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((u32)(c & RES_EPOCH_MASK) == expect) break;
        if (a.bell) {  // (the voice kernel's workgroup 0 gave up waiting: it says so in the relay)
          const u64 l = __hip_atomic_load(a.relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((l & RES_LEAVE) && (u32)(l & RES_EPOCH_MASK) == expect) { c = l; break; }
        }
        if (__builtin_amdgcn_s_memrealtime() - t0 > patience) { c = RES_LEAVE; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    const ResCall call = res_unpack(c);
    if (call.leave) return;
    expect = (expect + 1u) & (u32)RES_EPOCH_MASK;
    const u32 n_frames = call.frame_end - call.frame_begin;
    const u32 n_tiles = (n_frames + a.tile_frames - 1u) / a.tile_frames;
    const bool tracer = writes_host && lane == 0u;  // diagnostics: the device clock at the call's milestones (host_done[10..])
    if (tracer && wv == 0u) __hip_atomic_store(reinterpret_cast<u64*>(a.host_done + 10), __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // Tiles wv, wv + NW, ..; the call's flag granules -- summed the same way -- are "tile" n_tiles in that dealing, so that the
    // wavefront they fall to works on them beside the one that has the last tile.  Then a barrier of the workgroup, behind
    // which everything this workgroup writes to the host is written, and the epoch.
    for (u32 t = wv; t <= n_tiles; t += NW) {
      const bool flags_pass = t == n_tiles;
      const u32 tag = res_tag(call.epoch, flags_pass ? 255u : t);
      const u32 rel = t * a.tile_frames;
      const u32 len = flags_pass ? 1u : (n_frames - rel < a.tile_frames ? n_frames - rel : a.tile_frames);
      const bool mine = lane < len;
      bool gave_up = false;
      for (u32 p = 0; p < (flags_pass ? 1u : a.planes); ++p) {
        // where this wavefront's granules come from: row k's granule of frame 0 is base[k * stride]
        const u64* base;
        long stride;  // granules between one row's granule of a frame and the next row's
        if (flags_pass) { base = root ? a.group_flags : a.wg_flags + g * 32u; stride = 1; }
        else if (root) { base = a.group_rows + (((long)t * a.planes + p) * 8) * 64 * W; stride = 64 * W; }
        else { base = a.rows + (((long)t * a.planes + p) * a.n_rows + g * 32u) * 64 * W; stride = 64 * W; }
        const u64* const src = flags_pass ? base : base + (long)lane * W;  // this lane's frame
        // First a PROBE: one granule of every row (frame 0's; lane k watches row k), asleep in between -- a wavefront that read
        // all 32 x 64 granules in a loop took the issue slots of the voice wavefront it shares a SIMD with, and counters the
        // workgroups added to became hot lines in the memory system.  Each probed line is written by one workgroup and read
        // by one lane.  A row's other frames may land a moment later: the tags decide, below.
        if (writes_host && lane == 0u) __hip_atomic_store(&a.host_done[20 + wv], 0x100u | (t << 16) | (p << 12), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (diagnostics: where each wavefront is)
        // (the root has at most 8 granules per frame to read: it reads them all, napping in between, and needs no probe)
        if (p == 0u && !root) {
          const u64 tc = __builtin_amdgcn_s_memrealtime();
          for (;;) {
            // (until the FIRST row is there: the workgroups run in step, the others are then a fraction of a microsecond
            // behind, and the loads of the whole tile that follow are in flight while they land -- waiting for the last row
            // here cost every tile one more trip through the memory system, 0.7 us of each call)
            bool there = false;
            if (lane < in_g) there = (u32)(res_get(base + (long)lane * stride) >> 32) == tag;
            if (__builtin_amdgcn_ballot_w64(there) != 0ull) break;
            if (__builtin_amdgcn_s_memrealtime() - tc > patience) { gave_up = true; break; }
            __builtin_amdgcn_s_sleep(4);
          }
          if (gave_up) break;
          if (tracer && t == 0u) __hip_atomic_store(reinterpret_cast<u64*>(a.host_done + 12), __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (tracer && t + 1u == n_tiles) __hip_atomic_store(reinterpret_cast<u64*>(a.host_done + 14), __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (writes_host && lane == 0u) __hip_atomic_store(&a.host_done[20 + wv], 0x200u | (t << 16) | (p << 12), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // then every row's granule(s) of this lane's frame, all requested before any is looked at
        u64 w[32 * W];
        bool all = !mine;
        const u64 t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_ballot_w64(!all) != 0ull) {
          if (!all) {
#pragma unroll
            for (int k = 0; k < 32; ++k) {
#pragma unroll
              for (int j = 0; j < W; ++j)  // (a flag granule is one word whatever the sample type)
                w[k * W + j] = (u32)k < in_g && !(flags_pass && j > 0) ? res_get(src + (long)k * stride + j) : ((u64)tag << 32);
            }
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 32 * W; ++k) ok = ok && (u32)(w[k] >> 32) == tag;
            all = ok;
          }
          if (__builtin_amdgcn_s_memrealtime() - t0 > patience) { gave_up = true; break; }
          if (__builtin_amdgcn_ballot_w64(!all) != 0ull) { if (root) __builtin_amdgcn_s_sleep(3); else __builtin_amdgcn_s_sleep(1); }
        }
        if (root && p == 0u && !gave_up) {
          if (tracer && t == 0u) __hip_atomic_store(reinterpret_cast<u64*>(a.host_done + 12), __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (tracer && t + 1u == n_tiles) __hip_atomic_store(reinterpret_cast<u64*>(a.host_done + 14), __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (gave_up) break;
        if (!mine) continue;
        if (flags_pass) {
          u32 n_done = 0, n_run = 0;
          // (workgroup granules: counts in bits 0-7 and 8-15; group granules: bits 0-15 and 16-31)
#pragma unroll
          for (int k = 0; k < 32; ++k) {
            const u32 d = (u32)k < in_g ? (u32)w[k * W] : 0u;
            n_done += root ? (d & 0xFFFFu) : (d & 0xFFu);
            n_run += root ? (d >> 16) : ((d >> 8) & 0xFFu);
          }
          if (writes_host) {
            res_put_host(a.host_out + (long)a.planes * a.block_size * W, n_done | (n_run << 16), tag);
          } else {
            res_put(a.group_flags + g, n_done | (n_run << 16), tag);
          }
        } else {
          F v[32];
#pragma unroll
          for (int k = 0; k < 32; ++k) {
            if constexpr (W == 1) v[k] = (F)__builtin_bit_cast(float, (u32)w[k]);
            else v[k] = (F)__builtin_bit_cast(double, (u64)(u32)w[2 * k] | ((u64)(u32)w[2 * k + (W - 1)] << 32));
          }
          const F node = in_g == 32u ? tree_reduce<F, 32, true>(v, 1u, 32u) : tree_reduce<F, 32, false>(v, 1u, in_g);
          if (writes_host) {
            // the root, straight into the host's staging block (the host copies a mono mix to every channel; a Pan2 chain's planes are the channels)
            res_put_host_sample(a.host_out + ((long)p * a.block_size + call.frame_begin + rel + lane) * W, node, tag);
          } else {
            res_put_sample(a.group_rows + ((((long)t * a.planes + p) * 8 + g) * 64 + lane) * W, node, tag);
          }
        }
      }
      if (gave_up) return;  // (the voice kernel went away in mid-call: nothing to wait for; a wavefront that has ended no longer counts at the barrier)
    }
    if (writes_host && lane == 0u) {  // (diagnostics only: nothing waits for these)
      __hip_atomic_store(&a.host_done[20 + wv], 0x300u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((n_tiles % NW) == wv) {  // the wavefront that had the flags pass, the call's last piece of work
        __hip_atomic_store(reinterpret_cast<u64*>(a.host_done + 16), __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&a.host_done[0], call.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// The wavefront's subtree of KNH_MIX_TREE WITHOUT a trip through LDS (round 4; the whole-chain kernels): a butterfly over the
// lanes.  Every lane holds N consecutive frames of its voice in registers.  At level k the lane meets the lane that holds the
// sibling subtree -- the voice whose index differs in bit k -- keeps one half of its frames, hands the other half over, and
// adds what it receives: after level k it holds level-k + 1 nodes for half as many frames.  Once a lane is down to one frame
// the remaining levels are plain exchanges (both lanes form the same sum).  The additions are those of tree_reduce, operands
// possibly swapped (IEEE addition commutes): the same bits.
// Which lanes meet is a matter of what the hardware exchanges in one instruction, so the voices of a wavefront are dealt to
// its lanes accordingly (fold_voice_of_lane): voice bit 0 <-> lanes l and l ^ 32 (v_permlane32_swap), bit 1 <-> l ^ 16
// (v_permlane16_swap), bit 2 <-> l ^ 15 (DPP row_mirror), bit 3 <-> l ^ 7 (row_half_mirror), bit 4 <-> l ^ 2, bit 5 <-> l ^ 1
// (quad_perm).  The state rows are read and written with that permutation: still one coalesced 256-byte access per row.
// Measured before (profiles/r04_wide_stamps_*): the tile's LDS stores and the column-wise fold were 3 650 of the 13 000 cycles an
// f64 wavefront spends per 64 samples (1 800 of 8 500 in f32), on half-empty wavefronts (a 32-frame tile has 32 columns).
// ---------------------------------------------------------------------------
__device__ __forceinline__ u32 fold_voice_of_lane(u32 l) {
  const u32 b0 = (l >> 5) & 1u, b1 = (l >> 4) & 1u, b2 = (l >> 3) & 1u, b3 = ((l >> 2) ^ (l >> 3)) & 1u, b4 = ((l >> 1) ^ (l >> 2)) & 1u, b5 = (l ^ (l >> 2)) & 1u;
  return b0 | (b1 << 1) | (b2 << 2) | (b3 << 3) | (b4 << 4) | (b5 << 5);
}
// what the lane's partner of level K holds in `v` (K = 0, 1 are done with the swap instructions below: not here)
template <int K, typename F> __device__ __forceinline__ F fold_from_partner(F v) {
  static_assert(K >= 2 && K <= 5, "levels 2..5 are DPP exchanges");
  constexpr int ctrl = K == 2 ? 0x140 /* row_mirror */ : K == 3 ? 0x141 /* row_half_mirror */ : K == 4 ? 0x4E /* quad_perm [2,3,0,1] */ : 0xB1 /* quad_perm [1,0,3,2] */;
  return __builtin_amdgcn_update_dpp((F)0, v, ctrl, 0xF, 0xF, true);
}
// lanes whose voice has bit 0 (bit 1) set hold X = the partner's X... : a's upper rows swapped with b's lower rows.  After it,
// in EVERY lane, a and b are the two values of one frame -- the lane's own and its partner's -- for the half the lane keeps.
template <int K> __device__ __forceinline__ void fold_swap(float& a, float& b) {
  static_assert(K == 0 || K == 1, "levels 0 and 1 are the swap instructions");
  typedef u32 v2 __attribute__((ext_vector_type(2)));
  const v2 r = K == 0 ? __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(u32, a), __builtin_bit_cast(u32, b), false, false)
                      : __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(u32, a), __builtin_bit_cast(u32, b), false, false);
  // (the elements go through scalars of their own: __builtin_bit_cast applied to r[1] directly reads r[0] with this compiler)
  const u32 r0 = r[0], r1 = r[1];
  a = __builtin_bit_cast(float, r0);
  b = __builtin_bit_cast(float, r1);
}
template <int K> __device__ __forceinline__ void fold_swap(double& a, double& b) {
  typedef u32 v2 __attribute__((ext_vector_type(2)));
  const u64 ua = __builtin_bit_cast(u64, a), ub = __builtin_bit_cast(u64, b);
  const v2 lo = K == 0 ? __builtin_amdgcn_permlane32_swap((u32)ua, (u32)ub, false, false) : __builtin_amdgcn_permlane16_swap((u32)ua, (u32)ub, false, false);
  const v2 hi = K == 0 ? __builtin_amdgcn_permlane32_swap((u32)(ua >> 32), (u32)(ub >> 32), false, false)
                       : __builtin_amdgcn_permlane16_swap((u32)(ua >> 32), (u32)(ub >> 32), false, false);
  a = __builtin_bit_cast(double, (u64)lo[0] | ((u64)hi[0] << 32));
  b = __builtin_bit_cast(double, (u64)lo[1] | ((u64)hi[1] << 32));
}
// One lane's view of the fold: its voice index inside the wavefront, and per level whether its own subtree and its sibling's
// hold any live voice (a node without a right neighbour passes through; a dead right-hand lane takes over what its live
// partner hands it, so that every lane ends with the sum for the frame it owns).
struct FoldLane {
  u32 v;   // voice index in the wavefront (0..63) of this lane
  u32 nv;  // live voices of the wavefront
  template <int K> __device__ __forceinline__ bool bit() const { return ((v >> K) & 1u) != 0u; }
  template <int K> __device__ __forceinline__ bool own() const { return ((v >> K) << K) < nv; }
  template <int K> __device__ __forceinline__ bool sibling() const { return (((v >> K) ^ 1u) << K) < nv; }
};
template <int K, bool FULL, typename F> __device__ __forceinline__ F fold_join(const FoldLane& fl, F keep, F recv) {
  const F sum = keep + recv;
  if (FULL) return sum;
  return fl.own<K>() ? (fl.sibling<K>() ? sum : keep) : recv;
}
template <int K, int N, bool FULL, typename F> struct FoldLevels {
  // x[0..N): N frames of level-K nodes.  Returns the lane's sum over all 64 voices for the frame it ends up owning.
  static __device__ __forceinline__ F run(const FoldLane& fl, F (&x)[N]) {
    if constexpr (K == 6) {
      static_assert(N == 1, "six levels fold 64 lanes");
      return x[0];
    } else if constexpr (N == 1) {  // down to one frame: both lanes of a pair form the same sum
      F recv;
      if constexpr (K <= 1) {
        // (the swap hands the upper lanes' `a` to the lower lanes' `b` and back: with a = b = x both lanes see both values)
        F a = x[0], b = x[0];
        fold_swap<K>(a, b);
        recv = fl.bit<K>() ? a : b;
      } else {
        recv = fold_from_partner<K>(x[0]);
      }
      F y[1] = {fold_join<K, FULL>(fl, x[0], recv)};
      return FoldLevels<K + 1, 1, FULL, F>::run(fl, y);
    } else {
      F y[N / 2];
      if constexpr (K <= 1) {
#pragma unroll
        for (int k = 0; k < N / 2; ++k) {
          F a = x[k], b = x[k + N / 2];
          fold_swap<K>(a, b);  // lanes without the bit: a = own lower-half frame, b = the partner's; with it: b = own upper-half frame, a = the partner's
          if (FULL) y[k] = a + b;
          else y[k] = fold_join<K, false>(fl, fl.bit<K>() ? b : a, fl.bit<K>() ? a : b);
        }
      } else {
        const bool up = fl.bit<K>();
#pragma unroll
        for (int k = 0; k < N / 2; ++k) {
          const F keep = up ? x[k + N / 2] : x[k];
          const F give = up ? x[k] : x[k + N / 2];
          y[k] = fold_join<K, FULL>(fl, keep, fold_from_partner<K>(give));
        }
      }
      return FoldLevels<K + 1, N / 2, FULL, F>::run(fl, y);
    }
  }
};
// the frame of x[0..N) a lane owns after the fold (N a power of two <= 64), and whether it is the lane that writes it (of the
// lanes that end with the same frame -- N < 64 -- the one whose remaining voice bits are zero)
template <int N> __device__ __forceinline__ u32 fold_frame_of(u32 v, bool& writer) {
  u32 f = 0;
  int n = N, k = 0;
#pragma unroll
  for (; n > 1; n >>= 1, ++k) f += ((v >> k) & 1u) * (u32)(n >> 1);
  writer = (v >> k) == 0u;
  return f;
}
template <int N, typename F> __device__ __forceinline__ F wave_tree_fold(const FoldLane& fl, F (&x)[N]) {
  if (__builtin_expect(fl.nv == 64u, 1)) return FoldLevels<0, N, true, F>::run(fl, x);
  // the last wavefront of a bank whose voice count is not a multiple of 64: its eighteen lane masks (own / sibling / bit per
  // level) are formed here, behind a barrier the optimiser cannot hoist them across -- kept live through the kernel's main
  // loop they were spilled, and read back lane by lane, on the full wavefronts' path too
  FoldLane cold = fl;
  asm volatile("" : "+v"(cold.v), "+v"(cold.nv));
  return FoldLevels<0, N, false, F>::run(cold, x);
}

// An envelope stage in a kernel whose wavefronts have 256 registers or fewer (eight and more wavefronts per workgroup): a
// 64-sample visit is two tiles of 32 to it.  The envelope's rarer paths hold a second array of T values, and with 64 of each
// the eight-wavefront f32 kernel spilled 264 bytes per lane -- in this stage, whose cycles grew fourfold from one wavefront per
// SIMD to two where the filter's grew 1.8-fold (tools/wide_stamps.py): 262 144 voices 75.8 -> 70.9 us per block on one box.
// Two tiles in a row are the same samples (every path of the stage is the per-sample sequence, value for value).  The
// four-wavefront kernels (512 registers) keep the single 64-sample tile: split, they lose 3 %.
template <typename S, int BYTES> struct TightEnv : S {  // BYTES: the longest tile the stage is given, in bytes of samples
  template <typename F, bool FMA, int T>
  static __device__ __forceinline__ void tick_tile(typename S::template Regs<F>& r, F (&x)[T], const Ctx& c, u32 frame0, u32& done_frame) {
    constexpr int M = BYTES / (int)sizeof(F);
    if constexpr (T > M && T % M == 0 && M >= 8) {
#pragma unroll
      for (int h = 0; h < T / M; ++h) S::template tick_tile<F, FMA, M>(r, *reinterpret_cast<F(*)[M]>(&x[h * M]), c, frame0 + (u32)(h * M), done_frame);
    } else {
      S::template tick_tile<F, FMA, T>(r, x, c, frame0, done_frame);
    }
  }
};
template <int WAVES, typename S> struct ForWaves { typedef S type; };
template <bool AR> struct ForWaves<8, MulEnvT<AR>> { typedef TightEnv<MulEnvT<AR>, 128> type; };
template <bool AR> struct ForWaves<16, MulEnvT<AR>> { typedef TightEnv<MulEnvT<AR>, 32> type; };  // (128 registers: eight f32 samples at a time)

// LDS: the sine table (64 KiB, only if a stage uses it) + eight rows of 64 samples per wavefront for the sample-by-sample path.
// WAVES = wavefronts (64-voice groups) per workgroup sharing the table: 1 for small banks, 4, 8 or 16 when
// the bank has more 64-voice groups than the chip has SIMDs to give each its own (throughput regime).
// The voices' samples never leave the registers on the fast path: each visit's frames are folded over the wavefront's
// voices by wave_tree_fold (above) and the sums go straight to the wavefront's partial row.
template <typename F, bool FMA, int WAVES, typename... S>
__global__ void __launch_bounds__(WAVES * 64) voice_kernel(VoiceKernelArgs<F> a) {
  typedef typename ChainSelect<(SlotCount<S...>::value > 0), F, FMA, typename ForWaves<WAVES, S>::type...>::type ChainT;
  typedef typename WordOf<F>::type W;
  // samples evaluated stage by stage in registers per visit: 32 (the per-visit costs -- event test, the filter's choice of step,
  // register set-up around its fixed-register code -- are paid a quarter as often as with the eight of rounds 1-2:
  // 53.7 -> 39.9 us per block at 131 072 voices, C4's 65 536 f64 voices 50.5 -> 38.0, C1's one voice 2.43 -> 1.94); sixteen
  // wavefronts per workgroup have 128 registers each: 16 samples in f32, 8 in f64
  // (a voice of more than sixteen stages -- a graph, as a rule -- keeps the eight-sample visits: every stage's tile code is
  // unrolled per visit length, and hiprtc needs minutes for a 200-stage voice at 32 + 8 samples where it needs seconds at 8)
  // Up to four wavefronts per workgroup (a SIMD's registers to themselves) visit 64 samples at a time in f32: 65 536 voices
  // 26.1 -> 22.7 us per block (profiles/r04_wide_visit_length.txt); f64 gains nothing from it (36.6 -> 36.4) and keeps 32.
#ifndef KNH_WIDE_KT8
#define KNH_WIDE_KT8 64  // (eight wavefronts per workgroup, two per SIMD: 262 144 voices 76.2 -> 73.9 us per block; 32 for an A/B build)
#endif
  constexpr int KT = sizeof...(S) > 16 ? kTile : (WAVES >= 16 ? (sizeof(F) == 4 ? 16 : 8) : (WAVES <= 4 ? (sizeof(F) == 4 ? 64 : 32) : (sizeof(F) == 4 && !ChainT::kUsesRing ? KNH_WIDE_KT8 : 32)));  // (a delay's tile, its lines and the lines read ahead: 64 samples of each spill at 256 registers)
  // (one LDS object with the table first: the table at LDS address 0, a table read's address is the masked phase itself)
  constexpr bool kRingTile = ChainT::kUsesRing && WAVES <= 8 && KT * (int)sizeof(F) >= RingLines<F>::kLine;  // (sixteen tiles do not fit beside the table)
  constexpr int kSlowBytes = kTile * 64 * (int)sizeof(F);
  constexpr int kScratch = kRingTile && RingLines<F>::kTileBytes > kSlowBytes ? RingLines<F>::kTileBytes : kSlowBytes;
  struct Lds {
    float sine[ChainT::kUsesSine ? 16384 : 4];
    // per wavefront: the sample-by-sample path's samples, [frame][lane] -- and, in the same bytes, the tile through which a
    // delay moves its ring as whole lines (RingLines; visits of a line's samples or more, so never while the other is in use)
    __attribute__((aligned(16))) char scratch[WAVES][kScratch];
    u32 res_slot[4];                                        // a resident launch's command word
  };
  __shared__ Lds lds;
  auto& sine = lds.sine;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  if (ChainT::kUsesSine) {
    // Stage the 64 KiB table with LDS-DMA: 1 KiB per wave-instruction (16 B per lane, linear in LDS),
    // all issued back to back, one wait at the end.  The LDS base of each piece is wave-uniform.
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
#pragma unroll 8
    for (int k = wave; k < 64; k += WAVES) {
      const float* g = a.sine_table + (k * 64 + lane) * 4;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(sine + k * 256), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  Ctx ctx;
  ctx.sine = sine;
  ctx.f2pi = a.f2pi;
  ctx.seg_table = a.seg_table;
  ctx.seg_max = a.seg_max;
  ctx.delay_ring = a.delay_ring;
  ctx.delay_stride = a.delay_stride;
  ctx.ring_sink_row = a.n_voices;
  ctx.buffer = a.buffer;
  ctx.buffer_frames = a.buffer_frames;
  ctx.input_block = a.input;
  ctx.in_stride = a.block_size;
  ctx.sample_rate = a.sample_rate;
  ctx.ring_tile = nullptr;
  if constexpr (kRingTile) ctx.ring_tile = (__attribute__((address_space(3))) char*)lds.scratch[wave];

  const u32 wave_global = blockIdx.x * WAVES + wave;
  const u32 v0 = wave_global * 64u;
  if (v0 >= a.n_voices) return;
  const u32 nv = a.n_voices - v0 < 64u ? a.n_voices - v0 : 64u;  // live voices in this wave
  // the wavefront's voices are dealt to its lanes the way the fold exchanges them (fold_voice_of_lane)
  FoldLane fl;
  fl.v = fold_voice_of_lane((u32)lane);
  fl.nv = nv;
  const bool live = fl.v < nv;
  const u32 voice = live ? v0 + fl.v : v0 + nv - 1;  // idle lanes shadow the last live voice, never store

  ChainT chain;
  chain.load(a.state + voice, a.stride);

  // A resident launch (Resident, above): one pass of the loop below per command -- the frame range and the event list are the
  // call's, the sums leave as granules for the fold server (64-frame tiles from frame_begin on), done frames and flags are
  // reported per call.  An ordinary launch makes one pass.
  const bool resident = a.res.bell != nullptr;
  u32 res_expect = a.res.first_epoch, res_epoch = 0u;
  u32 fbeg = a.frame_begin, fend = a.frame_end;
  const u32* evs = a.ev_start;
  const Event* evl = a.events;
  // the voice's next event waits in registers, whole: one 16-byte read per event (voice_pipe.hpp)
  u32 ev_i = 0, ev_end = 0;
  Event nxt;
  nxt.frame = 0xFFFFFFFFu; nxt.slot_op = 0u; nxt.bits = 0ull;
  auto fetch = [&](u32 i) -> Event {
    Event e;
    if (resident) {  // (lists the host has rewritten since the kernel started: past this CU's L1)
      e.frame = __hip_atomic_load(&evl[i].frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      e.slot_op = __hip_atomic_load(&evl[i].slot_op, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      e.bits = __hip_atomic_load(&evl[i].bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      e = evl[i];
    }
    return e;
  };
  u32 base = 0;  // absolute frame of the current block's frame 0

  auto apply_events_upto = [&](u32 n_abs) {
    while (nxt.frame <= n_abs) {
      const u32 op = nxt.slot_op >> 24, slot = nxt.slot_op & 0xFFFFFFu;
      chain.on_event(op, slot, nxt.bits, nxt.frame - base);
      // a patched coefficient slot is not part of the end-of-launch write-back: persist it now
      // (mutable slots are overwritten by their evolved value at the end)
      if (live && (op & 0x7Fu) == EV_SET) a.state[(long)slot * a.stride + voice] = (W)nxt.bits;
      ++ev_i;
      if (ev_i < ev_end) nxt = fetch(ev_i);
      else nxt.frame = 0xFFFFFFFFu;
    }
  };

  const u32 n_waves_total = (a.n_voices + 63u) / 64u;
#ifdef KNH_DAG_STAMPS  // diagnostic build only: cycles of this wavefront per stage and per fold (tools/wide_stamps.py)
  u64 st_stage[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_fold = 0, st_visits = 0;
  const u64 st_begin = __builtin_amdgcn_s_memtime();
#endif
  F pan_l = (F)0, pan_r = (F)0;
  for (;;) {  // calls
  if (resident) {
    const ResCall call = res_wait(a.res, res_expect, lds.res_slot, wave, lane);
    if (call.leave) break;
    res_expect = (res_expect + 1u) & (u32)RES_EPOCH_MASK;
    res_epoch = call.epoch;
    fbeg = call.frame_begin;
    fend = call.frame_end;
    evs = call.has_events ? a.res.ev_start[call.list] : nullptr;
    evl = a.res.events[call.list];
    chain.reset_marks();
  }
  ev_i = 0; ev_end = 0;
  nxt.frame = 0xFFFFFFFFu;
  if (evs) {
    if (resident) { ev_i = __hip_atomic_load(&evs[voice], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); ev_end = __hip_atomic_load(&evs[voice + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    else { ev_i = evs[voice]; ev_end = evs[voice + 1]; }
  }
  if (ev_i < ev_end) nxt = fetch(ev_i);
  base = 0;
  for (u32 b = 0; b < a.n_blocks; ++b, base += a.block_size) {
    ctx.input_block = reinterpret_cast<const F*>(a.input) + (long)b * a.in_channels * a.block_size;
    chain.begin_block(fbeg, ctx);
    // a chain that ends in Pan2 has a left and a right partial row per wavefront: [block][channel][wavefront][frame]
    F* const row0 = a.partials + ((long)b * (ChainT::kPan ? 2 : 1) * n_waves_total + wave_global) * a.block_size;
    F* const row1 = row0 + (long)n_waves_total * a.block_size;
    // m of the V frames x[0..V) from frame n on are real: fold them over the wavefront's voices, one sum per frame
    auto emit = [&](auto& x, u32 n, u32 m) {
      constexpr int V = (int)(sizeof(x) / sizeof(x[0]));
#ifdef KNH_DAG_STAMPS
      const u64 st_f0 = __builtin_amdgcn_s_memtime();
#endif
      bool writer;
      const u32 f = fold_frame_of<V>(fl.v, writer);
      if constexpr (!ChainT::kPan) {
        if (a.voices_out && live) {  // (parity / debug output, single blocks only: every lane its own voice's row)
#pragma unroll
          for (int j = 0; j < V; ++j)
            if ((u32)j < m) a.voices_out[(long)voice * a.block_size + n + j] = x[j];
        }
        const F total = wave_tree_fold<V>(fl, x);
        if (writer && f < m) {
          if (resident) {
            const u32 rel = n + f - fbeg, tile = rel >> 6;
            res_put_sample(a.res.rows + (((long)tile * n_waves_total + wave_global) * 64 + (rel & 63u)) * ResWords<F>::value, total, res_tag(res_epoch, tile));
          } else {
            row0[n + f] = total;
          }
        }
      } else {
        // Pan2 (pan.rs:31-36): each voice's sample times its two gains (the product is rounded), then one sum per channel
        chain.pan_gains(pan_l, pan_r);
        F xl[V], xr[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { xl[j] = x[j] * pan_l; xr[j] = x[j] * pan_r; }
        if (a.voices_out && live) {
#pragma unroll
          for (int j = 0; j < V; ++j)
            if ((u32)j < m) {
              a.voices_out[(long)voice * a.block_size + n + j] = xl[j];
              a.voices_out[((long)a.n_voices + voice) * a.block_size + n + j] = xr[j];
            }
        }
        const F tl = wave_tree_fold<V>(fl, xl);
        const F tr = wave_tree_fold<V>(fl, xr);
        if (writer && f < m) {
          if (resident) {
            const u32 rel = n + f - fbeg, tile = rel >> 6;
            res_put_sample(a.res.rows + ((((long)tile * 2 + 0) * n_waves_total + wave_global) * 64 + (rel & 63u)) * ResWords<F>::value, tl, res_tag(res_epoch, tile));
            res_put_sample(a.res.rows + ((((long)tile * 2 + 1) * n_waves_total + wave_global) * 64 + (rel & 63u)) * ResWords<F>::value, tr, res_tag(res_epoch, tile));
          } else {
            row0[n + f] = tl; row1[n + f] = tr;
          }
        }
      }
#ifdef KNH_DAG_STAMPS
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      st_fold += __builtin_amdgcn_s_memtime() - st_f0;
#endif
    };
    // visits of KT samples where a whole one fits and no voice of the wavefront has a change inside it, of eight samples
    // where that holds for eight (the rest of a block that is not a multiple of KT long, the neighbourhood of a change),
    // sample by sample (changes applied in front of their frame) for what is left
    for (u32 n = fbeg; n < fend;) {
      const u32 left = fend - n;
      apply_events_upto(base + n);
      if (left >= (u32)KT && !__builtin_amdgcn_ballot_w64(nxt.frame < base + n + KT)) {
        F x[KT];
#pragma unroll
        for (int j = 0; j < KT; ++j) x[j] = (F)0;
#ifdef KNH_DAG_STAMPS
        if constexpr (SlotCount<S...>::value == 0 && sizeof...(S) <= 8) {
          u64 t_prev = __builtin_amdgcn_s_memtime();
          chain.template tick_tile_stamped<KT>(x, ctx, n, st_stage, t_prev);
          st_visits += 1;
        } else
#endif
        chain.template tick_tile<KT>(x, ctx, n);
        emit(x, n, (u32)KT);
        n += KT;
      } else if (KT > kTile && left >= (u32)kTile && !__builtin_amdgcn_ballot_w64(nxt.frame < base + n + kTile)) {
        F x[kTile];
#pragma unroll
        for (int j = 0; j < kTile; ++j) x[j] = (F)0;
        chain.template tick_tile<kTile>(x, ctx, n);
        emit(x, n, (u32)kTile);
        n += kTile;
      } else {
        // (through eight rows of LDS, written with a run-time index: the register tile is never indexed dynamically, which
        // would put it -- the fast path's too -- in scratch memory; same-wave LDS traffic, program order is enough)
        const u32 m = left < (u32)kTile ? left : (u32)kTile;
        F(*rows)[64] = reinterpret_cast<F(*)[64]>(lds.scratch[wave]);
        for (u32 j = 0; j < m; ++j) {
          apply_events_upto(base + n + j);
          rows[j][lane] = chain.tick((F)0, ctx, n + j);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        F x[kTile];
#pragma unroll
        for (int j = 0; j < kTile; ++j) x[j] = (u32)j < m ? rows[j][lane] : (F)0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        emit(x, n, m);
        n += m;
      }
    }
    // changes due exactly at the end of the processed range (precise_timing.rs:85-103 runs the
    // change loop once more before breaking out)
    apply_events_upto(base + fend);
  }
  if (!resident) break;
  {  // the call's done frames and flags (an ordinary launch: below, once)
    u32 d;
    if constexpr (SlotCount<S...>::value > 0) {
      d = 0xFFFFFFFFu;
      if (a.env_ranks != 0ull) { u32 best_rank = 0u; chain.template collect_done_ranked<0>(a.env_ranks, best_rank, d); }
      else d = chain.collect_done(0xFFFFFFFFu);
    } else {
      d = chain.collect_done(0xFFFFFFFFu);
    }
    if (live) a.done_frames[voice] = d;
    const u32 n_done = (u32)__builtin_popcountll(__builtin_amdgcn_ballot_w64(live && d != 0xFFFFFFFFu));
    const u32 n_run = (u32)__builtin_popcountll(__builtin_amdgcn_ballot_w64(live && !chain.last_env_stopped(false)));
    if (lane == 0) res_put(a.res.wg_flags + wave_global, n_done | (n_run << 8), res_tag(res_epoch, 255u));
  }
  }  // calls
  if (resident) {
    if (live) chain.store(a.state + voice, a.stride);
    return;
  }
#ifdef KNH_DAG_STAMPS
  if (wave_global == 0u && lane == 0) {
    // cycles per 64 samples of one voice group: [4 .. 11] the stages in chain order, [13] the folds (partial-row stores
    // included), [14] everything
    const u64 samples = (u64)a.n_blocks * (fend - fbeg);
    const u64 d = samples > 0 ? samples : 1;
    for (int k = 0; k < 8; ++k) a.flags[4 + k] = (u32)(st_stage[k] * 64 / d);
    a.flags[12] = 0u;
    a.flags[13] = (u32)(st_fold * 64 / d);
    a.flags[14] = (u32)((__builtin_amdgcn_s_memtime() - st_begin) * 64 / d);
    a.flags[15] = (u32)(st_visits * KT * 64 / d);  // share of the samples that took the stamped (whole-visit) path, x 64
  }
#endif

  u32 done_frame = 0xFFFFFFFFu;
  if constexpr (SlotCount<S...>::value > 0) {  // a graph-shaped voice: its envelopes in the reference's task order, when that is not list order
    if (a.env_ranks != 0ull) {
      u32 best_rank = 0u;
      chain.template collect_done_ranked<0>(a.env_ranks, best_rank, done_frame);
    } else {
      done_frame = chain.collect_done(0xFFFFFFFFu);
    }
  } else {
    done_frame = chain.collect_done(0xFFFFFFFFu);
  }
  if (live) {
    chain.store(a.state + voice, a.stride);
    a.done_frames[voice] = done_frame;
  }
  const bool any_done = live && done_frame != 0xFFFFFFFFu;
  const bool running = live && !chain.last_env_stopped(false);
  const u64 bd = __builtin_amdgcn_ballot_w64(any_done);
  const u64 br = __builtin_amdgcn_ballot_w64(running);
  if (lane == 0) {
    if (bd) atomicOr(&a.flags[0], 1u);
    if (br) atomicAdd(&a.flags[1], (u32)__builtin_popcountll(br));
  }
}

// Hand-over of a launch's mixed block(s) to the HOST with no copy command and no stream synchronisation behind it: the fold
// kernel writes its output straight into mapped pinned host memory (`out` is then a host pointer) and, when the last of its
// workgroups is through, the launch's two flag words and an epoch number the host polls.  What UGen::process_block costs per
// call at the C ABI (knh_bank_process_block, one call per block as Task::run makes them, knaster_graph/src/task.rs:25-31) is
// then two kernel launches and one PCIe write, not two launches, two copies and a stream wait.
// Order: every thread that stored to host memory makes its stores visible system-wide (__threadfence_system: write-back and
// a wait for the acknowledgements) before its workgroup counts itself in; the workgroup that counts last has therefore seen
// every other one's fence, and its own epoch store is a system-scope release behind the flag words.
struct HostDone {
  u32* words;        // pinned host memory: [0] epoch of the last finished launch, [1] flags[0], [2] flags[1]; null: no hand-over
  u32* counter;      // device word, zero between launches: workgroups of this fold kernel that are through
  const u32* flags;  // the launch's flag words (written by the voice kernel, which has finished)
  u32 epoch;
};
__device__ __forceinline__ void fold_signal_host(const HostDone& h) {
  if (!h.words) return;  // uniform
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const u32 total = gridDim.x * gridDim.y;
    if (atomicAdd(h.counter, 1u) == total - 1u) {
      *h.counter = 0u;  // for the next launch (stream order)
      __hip_atomic_store(&h.words[1], h.flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&h.words[2], h.flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&h.words[0], h.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Exact left fold of `n_rows` rows, row order = voice order: out[n] = ((r0+r1)+r2)+... in sample
// precision (knaster_graph/src/graph.rs:827-872).  Serial in the row axis by definition; loads run
// 16 rows ahead of the adds.  out: [channels][out_stride]; frames [frame_begin, frame_end) are written.
template <typename F>
__global__ void __launch_bounds__(64) fold_rows_kernel(const F* rows, u32 n_rows, u32 row_len, u32 frame_begin,
                                                        u32 frame_end, F* out, u32 channels, u32 out_stride, u32 accumulate, u32* zero_flags,
                                                        HostDone host) {
  // the flag words the NEXT launch's voice kernel accumulates into (two sets alternate; this spares a memset node)
  if (zero_flags && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { zero_flags[0] = 0u; zero_flags[1] = 0u; }
  const u32 n = frame_begin + blockIdx.x * 64u + threadIdx.x;
  if (n < frame_end) {
    rows += (long)blockIdx.y * n_rows * row_len;   // blockIdx.y = block of the launch
    out += (long)blockIdx.y * channels * out_stride;
    const F* p = rows + n;
    F acc = p[0];
    u32 r = 1;
    for (; r + 16 <= n_rows; r += 16) {
      F v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = p[(long)(r + k) * row_len];
#pragma unroll
      for (int k = 0; k < 16; ++k) acc = acc + v[k];
    }
    for (; r < n_rows; ++r) acc = acc + p[(long)r * row_len];
    // accumulate: this bank is a further additive source on the same output (existing + new, graph.rs:850-864)
    for (u32 c = 0; c < channels; ++c) {
      F* o = out + (long)c * out_stride + n;
      *o = accumulate ? *o + acc : acc;
    }
  }
  fold_signal_host(host);
}

// KNH_MIX_TREE, from the per-wavefront rows upwards: the rows are consecutive nodes of one level of the tree described at
// tree_reduce, and the same rule goes on -- pairs of neighbours, a node without a right neighbour passes through.
// One 256-thread workgroup handles 16 frames: thread (g, f) folds the 16 rows [256 p + 16 g, + 16) of pass p (four levels,
// in registers), thread (0, f) the 16 results of the pass (four more), and the passes' results meet on a small stack in
// LDS the way a binary counter carries (pass k merges with as many finished neighbours as k has trailing one bits).
template <typename F>
__global__ void __launch_bounds__(256) fold_tree_kernel(const F* rows, u32 n_rows, u32 row_len, u32 frame_begin,
                                                         u32 frame_end, F* out, u32 channels, u32 out_stride, u32 accumulate, u32* zero_flags,
                                                         HostDone host) {
  __shared__ F part[16][17];
  __shared__ F stack[32][16];
  if (zero_flags && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { zero_flags[0] = 0u; zero_flags[1] = 0u; }  // see fold_rows_kernel
  rows += (long)blockIdx.y * n_rows * row_len;   // blockIdx.y = block of the launch
  out += (long)blockIdx.y * channels * out_stride;
  const u32 f = threadIdx.x & 15u, g = threadIdx.x >> 4;
  const u32 n = frame_begin + blockIdx.x * 16u + f;
  const bool in_range = n < frame_end;
  u32 depth = 0;
  const u32 n_pass = (n_rows + 255u) / 256u;
  for (u32 p = 0; p < n_pass; ++p) {
    const u32 r0 = p * 256u + g * 16u;
    F acc = (F)0;
    if (in_range && r0 < n_rows) {
      const u32 cnt = n_rows - r0 < 16u ? n_rows - r0 : 16u;
      const F* q = rows + (long)r0 * row_len + n;
      F v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = (u32)k < cnt ? q[(long)k * row_len] : (F)0;
      acc = cnt == 16u ? tree_reduce<F, 16, true>(v, 1u, 16u) : tree_reduce<F, 16, false>(v, 1u, cnt);
    }
    part[g][f] = acc;
    __syncthreads();
    if (g == 0 && in_range) {
      const u32 left = n_rows - p * 256u;                      // rows of this pass and later ones
      const u32 chunks = left >= 256u ? 16u : (left + 15u) / 16u;
      F c[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) c[k] = part[k][f];
      const F node = chunks == 16u ? tree_reduce<F, 16, true>(c, 1u, 16u) : tree_reduce<F, 16, false>(c, 1u, chunks);
      stack[depth++][f] = node;
      for (u32 m = p + 1u; (m & 1u) == 0u; m >>= 1) {  // p has a trailing one bit: its left neighbour of that level is on the stack
        stack[depth - 2][f] = stack[depth - 2][f] + stack[depth - 1][f];
        --depth;
      }
    }
    __syncthreads();
  }
  if (g == 0 && in_range) {
    // what is left are nodes of falling level, left to right: each passes through until it meets its left neighbour
    F total = stack[depth - 1][f];
    for (u32 d = depth - 1; d > 0; --d) total = stack[d - 1][f] + total;
    for (u32 c = 0; c < channels; ++c) {
      F* o = out + (long)c * out_stride + n;
      *o = accumulate ? *o + total : total;
    }
  }
  fold_signal_host(host);
}

}  // namespace knh_dev
