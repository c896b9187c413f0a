// voice_pipe.hpp -- wave-specialised (software-pipelined) form of the fused voice-bank kernel.
//
// Why: at the headline size (16 384 voices = 256 wavefronts) every wavefront is alone on its
// SIMD, where one wave issues at most one VALU instruction per 4 cycles.  A block's run time is
// then (instructions per sample in that one wave) x 4 cycles x block_size -- the serial SVF /
// envelope recurrences forbid splitting a voice over time.  So the chain is cut into stage groups
// and each group runs in its own wavefront of the same workgroup (one per SIMD of the CU), handing
// 16-sample tiles to the next group through double-buffered LDS; a final "mixer" wavefront does the
// transposed per-frame fold.  The critical path per sample drops from the whole chain to the
// heaviest group.  Every voice still executes exactly the same arithmetic in the same order, so
// results are bit-identical to the single-wave kernel.
#pragma once
#include "voice_chain.hpp"

namespace knh_dev {

template <typename... S> struct Group {};
// A stage group whose stages keep no state from one sample to the next (nothing but parameters: sin(p * TAU), x * c,
// x.powi(n) ..): K wavefronts run it side by side, wavefront k taking samples [k * T / K, (k + 1) * T / K) of every tile.
// That is how the expensive, per-sample-independent part of a chain (SinNumeric's sin) is spread over the SIMDs of the
// CU while the serial part (its phase accumulator) stays in one wavefront.  Every wavefront of the group holds the
// group's parameters and applies the parameter changes addressed to them.
template <int K, typename... S> struct Fan {};

template <typename G> struct GroupInfo;
template <typename... S> struct GroupInfo<Group<S...>> {
  static constexpr int slots = (0 + ... + S::kSlots);
  static constexpr bool uses_sine = (false || ... || S::kUsesSine);
  static constexpr bool uses_ring = (false || ... || S::kUsesRing);  // a delay: the group's wavefront moves ring tiles as lines (RingLines)
  static constexpr bool has_env = (false || ... || S::kIsEnv);
  static constexpr bool pan = (false || ... || IsPan<S>::value);  // the group ends the chain with a Pan2
  static constexpr int fan_for(int) { return 1; }
};
template <int K, typename... S> struct GroupInfo<Fan<K, S...>> {
  static constexpr int slots = (0 + ... + S::kSlots);
  static constexpr bool uses_sine = (false || ... || S::kUsesSine);
  static constexpr bool uses_ring = false;
  static constexpr bool has_env = false;
  static constexpr bool pan = false;
  // wavefronts for tiles of T samples: K, or as many as leave every one a window of eight samples (f64 tiles are shorter)
  static constexpr int fan_for(int T) { return K * 8 <= T ? K : (T >= 8 ? T / 8 : 1); }
  static_assert(((S::kMutableMask == 0u) && ...), "a Fan group's stages keep no state");
  static_assert((!S::kIsEnv && ...) && (!S::kNeedsBind && ...), "a Fan group's stages keep no state");
};
template <typename F, bool FMA, int BASE, typename G> struct GroupChain;
template <typename F, bool FMA, int BASE, typename... S> struct GroupChain<F, FMA, BASE, Group<S...>> {
  typedef Chain<F, FMA, BASE, S...> type;
};
template <typename F, bool FMA, int BASE, int K, typename... S> struct GroupChain<F, FMA, BASE, Fan<K, S...>> {
  typedef Chain<F, FMA, BASE, S...> type;
};

// Samples per pipeline step.  `value`: three double-buffered edges + the mixer's beside the 64 KiB sine table.
// `big`: twice that, for the forms with one tile fewer (PIPE_FOLD, PIPE_INPLACE below), so that the doubled tiles still
// fit the 160 KiB; the per-tile costs of the busiest wave (LDS hand-over, block/event bookkeeping, the barrier) are
// then paid half as often.
template <typename F> struct PipeTile {
  static constexpr int value = sizeof(F) == 4 ? 32 : 16;
  static constexpr int big = 2 * value;
};

// Edge tiles between stage groups: [edge][2 buffers][64 lanes][kEdgeStride] -- each lane's T samples are
// contiguous and moved with 16-byte LDS accesses; the row padding (T + 16 B) keeps both ds_write_b128
// (8-lane groups, 32 banks) and ds_read_b128 (16-lane groups, 64 banks) conflict-free.
template <typename F, int TILE> struct EdgeLayout {
  static constexpr int T = TILE;
  static constexpr int VW = 16 / (int)sizeof(F);          // elements per 16-byte access
  static constexpr int stride = T + VW;                   // elements per lane row
  static constexpr int tile = 64 * stride;                // elements per buffer
  typedef F Vec __attribute__((ext_vector_type(16 / sizeof(F))));
};
// How a pipeline's tiles reach the fold over the voices:
//   PIPE_MIXER    a mixer wavefront reads the last group's own double-buffered edge (short tiles: four edges beside the sine table)
//   PIPE_FOLD     the last stage group folds its tile itself (one private buffer, no mixer wavefront)
//   PIPE_INPLACE  the last stage group writes its tile over the one it read (it holds the whole tile in registers by then),
//                 and a mixer wavefront folds it one step later; the edge into the last group is triple-buffered for that.
//                 As many tiles as PIPE_FOLD, so the long tiles fit, and the fold runs on the CU's fourth SIMD instead of
//                 in the busiest wavefront.
enum { PIPE_MIXER = 0, PIPE_FOLD = 1, PIPE_INPLACE = 2 };
template <int MODE, int NG> struct EdgeMap {
  static_assert(MODE != PIPE_INPLACE || NG >= 2, "an in-place last group reads an edge");
  static constexpr int tiles = MODE == PIPE_MIXER ? NG * 2 : NG * 2 - 1;
  // the buffer (in tiles) group I writes its global tile g to / reads it from
  static __device__ __forceinline__ int out_tile(int I, int g) {
    if (MODE == PIPE_FOLD && I == NG - 1) return I * 2;
    if (MODE == PIPE_INPLACE && I >= NG - 2) return (NG - 2) * 2 + g % 3;
    return I * 2 + (g & 1);
  }
  static __device__ __forceinline__ int in_tile(int I, int g) { return I > 0 ? out_tile(I - 1, g) : 0; }
  static __device__ __forceinline__ int mixer_tile(int g) { return out_tile(NG - 1, g); }
};
// The pipeline's step barrier.  Its wavefronts run different code (one role each) and reach the barrier from different call
// sites; what is relied on is the HARDWARE barrier of gfx950 -- s_barrier counts the wavefronts of the workgroup that have
// arrived, wherever in the program each one is -- not HIP's __syncthreads(), whose contract speaks of one call site reached
// by all threads.  So it is spelled as the instruction, between the two workgroup-scope fences that make each wavefront's LDS
// writes of the step visible to the readers of the next (what __syncthreads() expands to on this target).
__device__ __forceinline__ void pipe_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <typename F> struct PipeShared {
  float* sine;
  F* edge;  // tiles of [64][stride]; edge i carries group i's output: EdgeMap says which tile holds what
  // The workgroup's piece of the launch's event list (contiguous: the list is sorted by voice), copied into spare LDS at
  // kernel start when it fits: applying an event is then an LDS read, not a round trip to pinned host memory (~1 us,
  // during which the wave -- and at the next barrier its whole pipeline -- stands still).  ev_lds_n == 0: read in place.
  const Event* ev_lds;
  u32 ev_lds_first, ev_lds_n;
  // a resident launch (voice_chain.hpp, Resident): two words for the command, and where the stage groups leave what the
  // mixer wavefront reports per call: [group wavefront][lane] the group's done mark, then [lane] "the voice's last envelope is running"
  u32* res_slot;
  u32* res_marks;
  Event* ev_stage;       // ... and the LDS its events are staged in, call by call (ev_cap of them; the workgroup has n_threads threads)
  u32 ev_cap, n_threads;
  __attribute__((address_space(3))) char* ring_tile_of[4];  // per stage group: its RingLines tile (voice_stages.hpp), null for a group without a delay
};

template <typename F, bool FMA, int T, bool PAN>
__device__ __forceinline__ void pipe_fold_tile(const F* tile, const VoiceKernelArgs<F>& a, int lane, u32 wave_global, u32 n_waves_total,
                                               int blk, u32 n0, u32 len, u32 v0, u32 nv, u32 res_tile = 0u, u32 res_epoch = 0u);

// One stage group = one wavefront.  I: group index, NG: number of chain groups (mixer excluded),
// LAST_ENV: index of the group holding the chain's last envelope stage (-1: none).
// MODE PIPE_FOLD: the last group also folds its tile over the voices (what pipe_run_mixer does in a wavefront of its own
// otherwise): it stores the tile in a buffer no other wavefront touches and reads it back column-wise.
template <typename F, bool FMA, int T, int MODE, int NG, int I, int BASE, int LAST_ENV, typename G>
__device__ __forceinline__ u32 pipe_run_group(const PipeShared<F>& sh, const VoiceKernelArgs<F>& a, int lane, u32 wave_global, u32 v0, u32 nv, int fan_i, int wave_all) {
  // (a resident launch: the frame range and the event list are each call's)
  const bool resident = a.res.bell != nullptr;
  u32 res_expect = a.res.first_epoch;
  u32 fbeg = a.frame_begin, fend = a.frame_end;
  const u32* evs = a.ev_start;
  typedef typename GroupChain<F, FMA, BASE, G>::type ChainT;
  typedef typename WordOf<F>::type W;
  constexpr bool FOLDS = MODE == PIPE_FOLD && I == NG - 1;
  typedef EdgeMap<MODE, NG> Map;
  // A Fan group's wavefront works on a WINDOW of every tile: TW samples starting fo samples into it (an ordinary group's
  // window is the tile).  Everything below -- the register tile, the event paths, the LDS rows -- is per window.
  constexpr int KF = GroupInfo<G>::fan_for(T);
  constexpr int TW = T / KF;
  static_assert(T % KF == 0 && TW % 8 == 0, "a window is whole runs of eight samples");
  static_assert(!(FOLDS && KF > 1), "the folding group is not a Fan group");
  const u32 fo = KF > 1 ? (u32)fan_i * (u32)TW : 0u;
  constexpr u32 SLOT_LO = (u32)BASE, SLOT_HI = (u32)(BASE + GroupInfo<G>::slots);

  Ctx ctx;
  ctx.ring_tile = nullptr;
  if constexpr (GroupInfo<G>::uses_ring && KF == 1 && T * (int)sizeof(F) >= RingLines<F>::kLine && I < 4) ctx.ring_tile = sh.ring_tile_of[I];
  ctx.sine = sh.sine;
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
  const bool live = (u32)lane < nv;
  const u32 voice = live ? v0 + lane : v0 + nv - 1;
  ChainT chain;
  chain.load(a.state + voice, a.stride);

  // The voice's next event waits in registers, whole: one 16-byte read per event (the list may sit in pinned host
  // memory, a PCIe round trip away), issued as soon as the event before it has been applied.
  u32 ev_i = 0, ev_end = 0;
  Event nxt;
  nxt.frame = 0xFFFFFFFFu; nxt.slot_op = 0u; nxt.bits = 0ull;
  bool ev_staged = sh.ev_lds_n != 0u;  // uniform: two plain loads (LDS / global, spelled with their address spaces), never a flat one
  u32 ev_lds_first = sh.ev_lds_first;
  typedef __attribute__((address_space(3))) const Event* lds_ev_t;
  typedef __attribute__((address_space(1))) const Event* glb_ev_t;
  const lds_ev_t ev_l = (lds_ev_t)sh.ev_lds;
  glb_ev_t ev_g = (glb_ev_t)a.events;
  auto fetch = [&](u32 i) -> Event {
    Event e;
    if (ev_staged) { e.frame = ev_l[i - ev_lds_first].frame; e.slot_op = ev_l[i - ev_lds_first].slot_op; e.bits = ev_l[i - ev_lds_first].bits; }
    else if (resident) {
      // a resident launch reads lists the host has rewritten since the kernel started: no kernel boundary has emptied this
      // CU's L1, so the loads say "system scope" themselves
      const Event* q = (const Event*)ev_g + i;
      e.frame = __hip_atomic_load(&q->frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      e.slot_op = __hip_atomic_load(&q->slot_op, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      e.bits = __hip_atomic_load(&q->bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    else { e.frame = ev_g[i].frame; e.slot_op = ev_g[i].slot_op; e.bits = ev_g[i].bits; }
    return e;
  };
  // The frame of the next event in a register of its own, written by an instruction and not by a load: the per-tile
  // checks below then never wait for memory (the event may have been read from LDS, whose counter also covers the
  // tile reads that were just issued; the wait belongs where the event is fetched, not where its frame is looked at).
  u32 next_frame = 0xFFFFFFFFu;
  auto note_next = [&]() { asm volatile("v_mov_b32 %0, %1" : "=v"(next_frame) : "v"(nxt.frame)); };
  // a resident call whose events are range events (ResCall): event i is record 2 i of the list, record 2 i + 1 says which
  // voices it is for; ev_i walks all of them and stops at those that cover this lane's voice
  bool ev_ranged = false;
  auto take_next = [&]() {  // ev_i: the next candidate
    if (ev_ranged) {
      while (ev_i < ev_end) {
        const Event who = fetch(2u * ev_i + 1u);
        if (who.frame <= voice && voice < who.slot_op) break;
        ++ev_i;
      }
    }
    if (ev_i < ev_end) { nxt = fetch(ev_ranged ? 2u * ev_i : ev_i); note_next(); }
    else { nxt.frame = 0xFFFFFFFFu; next_frame = 0xFFFFFFFFu; }
  };
  u32 base = 0;  // absolute frame of the current block's frame 0
  auto apply_events_upto = [&](u32 n_abs) {
    while (__builtin_expect(next_frame <= n_abs, 0)) {
      const u32 op = nxt.slot_op >> 24, slot = nxt.slot_op & 0xFFFFFFu;
      if (slot >= SLOT_LO && slot < SLOT_HI) {  // every group scans the list, the owner applies
        chain.on_event(op, slot, nxt.bits, nxt.frame - base);
        if (live && (op & 0x7Fu) == EV_SET) a.state[(long)slot * a.stride + voice] = (W)nxt.bits;
      }
      ++ev_i;
      take_next();
    }
  };
  u32 done_frame = 0xFFFFFFFFu;
  for (;;) {  // one pass per call of a resident launch; an ordinary launch makes one
  if (resident) {
    const ResCall call = res_wait(a.res, res_expect, sh.res_slot, wave_all, lane);
    if (call.leave) break;
    res_expect = (res_expect + 1u) & (u32)RES_EPOCH_MASK;
    fbeg = call.frame_begin;
    fend = call.frame_end;
    evs = call.has_events ? a.res.ev_start[call.list] : nullptr;
    ev_g = (glb_ev_t)a.res.events[call.list];
    chain.reset_marks();
    const ResStaged st = res_stage_events(a.res, call, sh.ev_stage, sh.ev_cap, v0, nv, (u32)(wave_all * 64 + lane), sh.n_threads);
    ev_staged = st.count != 0u;
    ev_lds_first = st.first;
    ev_ranged = call.has_events && call.n_ranges != 0u;
    if (ev_ranged) { evs = nullptr; ev_i = 0; ev_end = call.n_ranges; }
  }
  if (!ev_ranged) {
    ev_i = 0; ev_end = 0;
    if (evs) {
      // (a resident call reads lists the host has rewritten since the kernel started: volatile = past this CU's L1, and
      // unlike atomic loads neighbouring lanes' words travel together)
      if (resident) { ev_i = *reinterpret_cast<const volatile u32*>(evs + voice); ev_end = *reinterpret_cast<const volatile u32*>(evs + voice + 1); }
      else { ev_i = evs[voice]; ev_end = evs[voice + 1]; }
    }
  }
  nxt.frame = 0xFFFFFFFFu; next_frame = 0xFFFFFFFFu;
  take_next();
  base = 0;

  // The pipeline runs continuously over all blocks of the launch: global tile g = (block, tile in block).
  const u32 n_frames = fend - fbeg;
  const int tpb = (int)((n_frames + T - 1) / T);           // tiles per block
  const int n_tiles = tpb * (int)a.n_blocks;
  const int n_steps = n_tiles + NG - (MODE == PIPE_FOLD ? 1 : 0);
  const u32 n_waves_total = (a.n_voices + 63u) / 64u;
  int blk = 0, ti = 0;                                      // position of this group's next tile
#ifdef KNH_DAG_STAMPS  // diagnostic build only: cycles this wavefront is busy per tile (tools/pipe_stamps.py)
  u64 busy = 0, busy_in = 0, busy_out = 0, busy_tick = 0, busy_fold = 0;
#endif
  for (int s = 0; s < n_steps; ++s) {
    const int g = s - I;
    if (g >= 0 && g < n_tiles) {
#ifdef KNH_DAG_STAMPS
      const u64 t0 = __builtin_amdgcn_s_memtime();
#endif
      // the tile's LDS reads go out first, so that their latency runs under the block/event bookkeeping below
      F x[TW];
      if (I > 0) {
        typedef typename EdgeLayout<F, T>::Vec Vec;
        constexpr int VW = EdgeLayout<F, T>::VW;
        const Vec* in = reinterpret_cast<const Vec*>(sh.edge + (long)Map::in_tile(I, g) * EdgeLayout<F, T>::tile +
                                                     (long)lane * EdgeLayout<F, T>::stride + fo);
#pragma unroll
        for (int j = 0; j < TW / VW; ++j) {
          const Vec v = in[j];
#pragma unroll
          for (int k = 0; k < VW; ++k) x[j * VW + k] = v[k];
        }
      } else {
#pragma unroll
        for (int j = 0; j < TW; ++j) x[j] = (F)0;
      }
      if (ti == 0) {
        ctx.input_block = reinterpret_cast<const F*>(a.input) + (long)blk * a.in_channels * a.block_size;
        chain.begin_block(fbeg, ctx);
      }
      const u32 n_tile = fbeg + (u32)ti * T;
      const u32 m_tile = fend - n_tile < (u32)T ? fend - n_tile : (u32)T;
      const u32 n = n_tile + fo;                                                            // the window's first frame
      const u32 m = m_tile > fo ? (m_tile - fo < (u32)TW ? m_tile - fo : (u32)TW) : 0u;     // frames of it inside the block
      apply_events_upto(base + n);  // (a Fan wavefront: also the changes inside the part of the tile before its window)
      const bool ev_inside = next_frame < base + n + TW;
#if defined(KNH_DAG_STAMPS) || defined(KNH_TILE_FENCES)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#ifdef KNH_DAG_STAMPS
      const u64 t1 = __builtin_amdgcn_s_memtime();
      busy_in += t1 - t0;
#endif
      {  // every group, the last one included, hands its tile on as 64 rows of T samples
        typedef typename EdgeLayout<F, T>::Vec Vec;
        constexpr int VW = EdgeLayout<F, T>::VW;
        F* out_tile = sh.edge + (long)Map::out_tile(I, g) * EdgeLayout<F, T>::tile;
        F* out_row = out_tile + (long)lane * EdgeLayout<F, T>::stride + fo;
        if (__builtin_expect(m == (u32)TW && !__builtin_amdgcn_ballot_w64(ev_inside), 1)) {  // (the cold paths out of line: one instruction cache for all roles)
          chain.template tick_tile<TW>(x, ctx, n);
#ifdef KNH_DAG_STAMPS
          asm volatile("" ::: "memory");
          busy_tick += __builtin_amdgcn_s_memtime() - t1;
#endif
          Vec* out = reinterpret_cast<Vec*>(out_row);  // 16-byte LDS stores
#pragma unroll
          for (int j = 0; j < TW / VW; ++j) {
            Vec v;
#pragma unroll
            for (int k = 0; k < VW; ++k) v[k] = x[j * VW + k];
            out[j] = v;
          }
#ifdef KNH_AB_NO_SW
        } else if (false && [&]() -> bool {
#else
        } else if (m == (u32)TW && ChainT::kParamBits != 0ull && !ChainT::kBinds && [&]() -> bool {
#endif
          // (groups that hold a delay line, a segment table or a buffer reader keep to the general path: their registers --
          // a prefetched tile of the ring among them -- are not worth copying for this)
          // Voices of the wave change PARAMETERS inside this tile (sample-accurate changes out of a WrPreciseTiming queue:
          // a new frequency, gain, filter coefficient set ..), each at one frame of its own.  The changes are applied to a
          // copy of the voice's registers in one pass, and the tile then runs stage by stage as usual, every sample taking
          // over its voice's new parameter values at that voice's frame (Chain::tick_tile_sw).  Anything else inside the tile
          // -- a trigger, a phase reset, two changes of one voice at different frames -- leaves everything as it was and
          // takes the general path below.
          ChainT cn = chain;
          const u32 sv_i = ev_i;
          const Event sv_nxt = nxt;
          const u32 sv_next_frame = next_frame;
          const u32 tile_end = base + n + (u32)TW;
          u32 sw = (u32)TW;
          u64 touched = 0ull;
          bool bad = false;
          while (next_frame < tile_end) {
            const u32 op = nxt.slot_op >> 24, slot = nxt.slot_op & 0xFFFFFFu, code = op & 0x7Fu;
            if (slot >= SLOT_LO && slot < SLOT_HI) {
              const u32 rel = nxt.frame - (base + n);
              if (sw == (u32)TW) sw = rel;
              bad = bad || sw != rel || slot >= 64u ||
                    !((code == EV_SET && ((ChainT::kParamBits >> (slot & 63u)) & 1ull)) || (code == EV_NOP && ((ChainT::kNopOkBits >> (slot & 63u)) & 1ull)));
              if (!bad) {
                cn.on_event(op, slot, nxt.bits, nxt.frame - base);
                touched |= 1ull << (slot & 63u);
                if (live && code == EV_SET) a.state[(long)slot * a.stride + voice] = (W)nxt.bits;
              }
            }
            ++ev_i;
            take_next();
          }
          if (__builtin_amdgcn_ballot_w64(bad) != 0) {
            ev_i = sv_i;
            nxt = sv_nxt;
            next_frame = sv_next_frame;
            return false;
          }
          // eight samples at a time, row to row in LDS (a run-time loop, like the general path: keeps this rare path small
          // and the register tile of the fast path out of it)
          const F* in_row = sh.edge + (long)Map::in_tile(I, g) * EdgeLayout<F, T>::tile + (long)lane * EdgeLayout<F, T>::stride + fo;
          for (u32 j0 = 0; j0 < (u32)TW; j0 += 8u) {
            F sub[8];
            if (I > 0) {
#pragma unroll
              for (int q = 0; q < 8 / VW; ++q) {
                const Vec v = reinterpret_cast<const Vec*>(in_row + j0)[q];
#pragma unroll
                for (int k = 0; k < VW; ++k) sub[q * VW + k] = v[k];
              }
            } else {
#pragma unroll
              for (int k = 0; k < 8; ++k) sub[k] = (F)0;
            }
            chain.template tick_tile_sw<8>(cn, sw - j0, touched, sub, ctx, n + j0);  // sw - j0 outside 0..7: no switch in this run
#pragma unroll
            for (int q = 0; q < 8 / VW; ++q) {
              Vec v;
#pragma unroll
              for (int k = 0; k < VW; ++k) v[k] = sub[q * VW + k];
              reinterpret_cast<Vec*>(out_row + j0)[q] = v;
            }
          }
          return true;
        }()) {
        } else {
          // Some voice of the wave has a change inside this tile (sample-accurate parameter changes, WrPreciseTiming), or
          // the tile is a partial one at the end of a block: it is walked eight samples at a time, row to row in LDS (a
          // run-time loop: the register tile above is never indexed by a run-time value, which would put all of it, the
          // fast path's too, in scratch memory), sample by sample with the changes applied in front of their frame.
          const F* in_row = sh.edge + (long)Map::in_tile(I, g) * EdgeLayout<F, T>::tile + (long)lane * EdgeLayout<F, T>::stride + fo;
          for (u32 j0 = 0; j0 < m; j0 += 8u) {
            const u32 cnt = m - j0 < 8u ? m - j0 : 8u;
            F sub[8];
            if (I > 0) {
#pragma unroll
              for (int q = 0; q < 8 / VW; ++q) {
                const Vec v = reinterpret_cast<const Vec*>(in_row + j0)[q];
#pragma unroll
                for (int k = 0; k < VW; ++k) sub[q * VW + k] = v[k];
              }
            } else {
#pragma unroll
              for (int k = 0; k < 8; ++k) sub[k] = (F)0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              if ((u32)k < cnt) {
                apply_events_upto(base + n + j0 + (u32)k);
                sub[k] = chain.tick(sub[k], ctx, n + j0 + (u32)k);
              }
            }
#pragma unroll
            for (int q = 0; q < 8 / VW; ++q) {
              Vec v;
#pragma unroll
              for (int k = 0; k < VW; ++k) v[k] = sub[q * VW + k];
              reinterpret_cast<Vec*>(out_row + j0)[q] = v;  // a partial run stores its unused tail too: the row is T (+ padding) long
            }
          }
        }
        if constexpr (GroupInfo<G>::pan) {
          // Pan2: the voice's two gains ride in the padding of its row (elements T, T + 1), beside the samples they scale
          F gl = (F)0, gr = (F)0;
          chain.pan_gains(gl, gr);
          out_row[T] = gl;
          out_row[T + 1] = gr;
        }
#ifdef KNH_DAG_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const u64 tf0 = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (FOLDS) pipe_fold_tile<F, FMA, T, GroupInfo<G>::pan>(out_tile, a, lane, wave_global, n_waves_total, blk, n_tile, m_tile, v0, nv);
#ifdef KNH_DAG_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        busy_fold += __builtin_amdgcn_s_memtime() - tf0;
#endif
      }
#if defined(KNH_DAG_STAMPS) || defined(KNH_TILE_FENCES)
      asm volatile("" ::: "memory");
#endif
#ifdef KNH_DAG_STAMPS
      const u64 t2 = __builtin_amdgcn_s_memtime();
#endif
      if (++ti == tpb) {  // block finished for this group
        apply_events_upto(base + fend);  // changes due exactly at the end (precise_timing.rs:85-103)
        ti = 0;
        ++blk;
        base += a.block_size;
        if (resident) {  // what the mixer wavefront reports for this call (it folds this tile one or more steps from now)
          sh.res_marks[wave_all * 64 + lane] = chain.collect_done(0xFFFFFFFFu);
          if (GroupInfo<G>::has_env && I == LAST_ENV) sh.res_marks[15 * 64 + lane] = live && !chain.last_env_stopped(false) ? 1u : 0u;
        }
      }
#if defined(KNH_DAG_STAMPS) || defined(KNH_TILE_FENCES)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#ifdef KNH_DAG_STAMPS
      const u64 t3 = __builtin_amdgcn_s_memtime();
      busy += t3 - t0;
      busy_out += t3 - t2;
#endif
    }
    pipe_barrier();
  }
#ifdef KNH_DAG_STAMPS
  if (wave_global == 0u && lane == 0) {
    const u64 d = (u64)(n_tiles > 0 ? n_tiles : 1);
    a.flags[4 + I] = (u32)(busy / d);
    if (8 + 2 * I + 1 < 16) { a.flags[8 + 2 * I] = (u32)(busy_in / d); a.flags[9 + 2 * I] = (u32)(busy_out / d); }
    if (FOLDS) { a.flags[14] = (u32)(busy_tick / d); a.flags[15] = (u32)(busy_fold / d); }  // the folding group's stage arithmetic and fold
    else if (MODE != PIPE_FOLD && I == 1) { a.flags[14] = (u32)(busy_tick / d); a.flags[15] = (u32)((busy - busy_in - busy_out - busy_tick) / d); }  // group 1: its arithmetic / its tile stores
  }
#endif
  done_frame = chain.collect_done(0xFFFFFFFFu);  // this group's envelopes; the kernel (a resident launch: the mixer wavefront) combines the groups in order
  if (!resident) break;
  }  // calls
  if (live) chain.store(a.state + voice, a.stride);
  if (!resident && GroupInfo<G>::has_env && I == LAST_ENV) {
    const bool running = live && !chain.last_env_stopped(false);
    const u64 br = __builtin_amdgcn_ballot_w64(running);
    if (lane == 0 && br) atomicAdd(&a.flags[1], (u32)__builtin_popcountll(br));
  }
  return done_frame;
}

// Lane j folds frame j of a finished tile over the wave's voices: the wavefront's subtree of the bank's pairwise sum
// (tree_reduce in voice_chain.hpp).  The tile is stored voice-major ([voice][T], the common edge format), so a read of one
// voice's row by lanes 0..T-1 is conflict-free and the transposition costs nothing; all reads go out before the first add.
// (a resident launch: the sums leave as granules for the fold server, tile `res_tile` of call `res_epoch`: voice_chain.hpp)
template <typename F, bool FMA, int T, bool PAN>
__device__ __forceinline__ void pipe_fold_tile(const F* tile, const VoiceKernelArgs<F>& a, int lane, u32 wave_global, u32 n_waves_total,
                                               int blk, u32 n0, u32 len, u32 v0, u32 nv, u32 res_tile, u32 res_epoch) {
  constexpr int ST = EdgeLayout<F, T>::stride;
  if ((u32)lane < len) {
    const F* col = tile + lane;
    if constexpr (!PAN) {
      F* const row = a.partials + ((long)blk * n_waves_total + wave_global) * a.block_size + n0 + lane;
      const F sum = fold_group<F, 64>(col, ST, nv);
      if (a.res.bell) res_put_sample(a.res.rows + (((long)res_tile * n_waves_total + wave_global) * 64 + lane) * ResWords<F>::value, sum, res_tag(res_epoch, res_tile));
      else *row = sum;
      if (a.voices_out) {
        for (u32 v = 0; v < nv; ++v) a.voices_out[(long)(v0 + v) * a.block_size + n0 + lane] = col[v * ST];
      }
    } else {
      // Pan2 (pan.rs:31-36): voice v contributes x * left_gain to channel 0 and x * right_gain to channel 1.
      // The gains sit in the padding of the voice's row: every lane reads the same word (a broadcast).
      const F* gain = tile + T;
      F accl, accr;
      fold_group_pan<F, 64>(col, ST, gain, gain + 1, ST, nv, accl, accr);
      F* const rl = a.partials + (((long)blk * 2 + 0) * n_waves_total + wave_global) * a.block_size + n0 + lane;
      F* const rr = a.partials + (((long)blk * 2 + 1) * n_waves_total + wave_global) * a.block_size + n0 + lane;
      if (a.res.bell) {
        res_put_sample(a.res.rows + ((((long)res_tile * 2 + 0) * n_waves_total + wave_global) * 64 + lane) * ResWords<F>::value, accl, res_tag(res_epoch, res_tile));
        res_put_sample(a.res.rows + ((((long)res_tile * 2 + 1) * n_waves_total + wave_global) * 64 + lane) * ResWords<F>::value, accr, res_tag(res_epoch, res_tile));
      } else { *rl = accl; *rr = accr; }
      if (a.voices_out) {
        for (u32 v = 0; v < nv; ++v) {
          const F t = col[v * ST];
          a.voices_out[(long)(v0 + v) * a.block_size + n0 + lane] = t * gain[v * ST];
          a.voices_out[((long)a.n_voices + v0 + v) * a.block_size + n0 + lane] = t * gain[v * ST + 1];
        }
      }
    }
  }
}

// The mixer wavefront: folds the tile the last chain group finished in the previous step.  In a resident launch it also takes
// the workgroup's row to the fold server as granules and, with a call's last tile, reports the call's done marks
// and flags.  CHAINW = wavefronts that run stage groups; HAS_ENV: some group holds an envelope.
template <typename F, bool FMA, int T, int MODE, int NG, bool PAN, int CHAINW, bool HAS_ENV>
__device__ __forceinline__ void pipe_run_mixer(const PipeShared<F>& sh, const VoiceKernelArgs<F>& a, int lane, u32 wave_global,
                                               u32 v0, u32 nv, int wave_all) {
  u32 fbeg = a.frame_begin, fend = a.frame_end;
  const bool resident = a.res.bell != nullptr;
  u32 res_expect = a.res.first_epoch;
  const u32 n_waves_total = (a.n_voices + 63u) / 64u;
  for (;;) {  // one pass per call of a resident launch
  ResCall call;
  call.epoch = 0u;
  if (resident) {
    call = res_wait(a.res, res_expect, sh.res_slot, wave_all, lane);
    if (call.leave) break;
    res_expect = (res_expect + 1u) & (u32)RES_EPOCH_MASK;
    fbeg = call.frame_begin;
    fend = call.frame_end;
    (void)res_stage_events(a.res, call, sh.ev_stage, sh.ev_cap, v0, nv, (u32)(wave_all * 64 + lane), sh.n_threads);  // (its share of the copy, and the barrier)
  }
  const u32 n_frames = fend - fbeg;
  const int tpb = (int)((n_frames + T - 1) / T);
  const int n_tiles = tpb * (int)a.n_blocks;
  const int n_steps = n_tiles + NG;
  int blk = 0, ti = 0;
#ifdef KNH_DAG_STAMPS
  u64 busy = 0;
#endif
  for (int s = 0; s < n_steps; ++s) {
    const int g = s - NG;  // the tile the last chain group finished in the previous step
    if (g >= 0 && g < n_tiles) {
#ifdef KNH_DAG_STAMPS
      const u64 t0 = __builtin_amdgcn_s_memtime();
#endif
      const u32 rel = (u32)ti * T;
      const u32 len = n_frames - rel < (u32)T ? n_frames - rel : (u32)T;
      const F* tile = sh.edge + (long)EdgeMap<MODE, NG>::mixer_tile(g) * EdgeLayout<F, T>::tile;
      pipe_fold_tile<F, FMA, T, PAN>(tile, a, lane, wave_global, n_waves_total, blk, fbeg + rel, len, v0, nv, (u32)g, call.epoch);
      if (resident && g == n_tiles - 1) {
        // the call's last tile: every stage group is through with the call (its marks are in LDS since its last step's
        // barrier).  mark_done of a voice = that of the last node in task order that set one: the groups in chain order.
        u32 n_done = 0u, n_run = 0u;
        if (HAS_ENV) {
          u32 d = 0xFFFFFFFFu;
#pragma unroll
          for (int w = 0; w < CHAINW; ++w) d = sh.res_marks[w * 64 + lane] != 0xFFFFFFFFu ? sh.res_marks[w * 64 + lane] : d;
          const bool live = (u32)lane < nv;
          if (live) a.done_frames[v0 + lane] = d;
          n_done = (u32)__builtin_popcountll(__builtin_amdgcn_ballot_w64(live && d != 0xFFFFFFFFu));
          n_run = (u32)__builtin_popcountll(__builtin_amdgcn_ballot_w64(live && sh.res_marks[15 * 64 + lane] != 0u));
        }
        if (lane == 0) res_put(a.res.wg_flags + wave_global, n_done | (n_run << 8), res_tag(call.epoch, 255u));
      }
      if (++ti == tpb) { ti = 0; ++blk; }
#ifdef KNH_DAG_STAMPS
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      busy += __builtin_amdgcn_s_memtime() - t0;
#endif
    }
    // (a resident launch: the mixer's granules are write-through stores nobody in the workgroup waits for; the step barrier only
    // has to order its LDS reads -- the workgroup-scope release of pipe_barrier would also wait for those stores to be acknowledged)
    if (resident) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else pipe_barrier();
  }
#ifdef KNH_DAG_STAMPS
  if (wave_global == 0u && lane == 0) a.flags[4 + NG] = (u32)(busy / (u64)(n_tiles > 0 ? n_tiles : 1));
#endif
  if (!resident) break;
  }  // calls
}

// W0: the first wavefront of group I (a Fan group takes several)
template <typename F, bool FMA, int T, int MODE, int NG, int I, int BASE, int W0, int LAST_ENV, typename G, typename... Rest>
__device__ __forceinline__ u32 pipe_dispatch(int wave, const PipeShared<F>& sh, const VoiceKernelArgs<F>& a, int lane, u32 wave_global, u32 v0, u32 nv, int wave_all) {
  constexpr int KF = GroupInfo<G>::fan_for(T);
  if (wave >= W0 && wave < W0 + KF) return pipe_run_group<F, FMA, T, MODE, NG, I, BASE, LAST_ENV, G>(sh, a, lane, wave_global, v0, nv, wave - W0, wave_all);
  if constexpr (sizeof...(Rest) > 0)
    return pipe_dispatch<F, FMA, T, MODE, NG, I + 1, BASE + GroupInfo<G>::slots, W0 + KF, LAST_ENV, Rest...>(wave, sh, a, lane, wave_global, v0, nv, wave_all);
  return 0xFFFFFFFFu;
}
// wavefronts of a pipeline: one per group (K per Fan group), plus the mixer unless the last group folds
template <int T, int MODE, typename... Gs> struct PipeWaves {
  static constexpr int chain = (0 + ... + GroupInfo<Gs>::fan_for(T));
  static constexpr int value = chain + (MODE == PIPE_FOLD ? 0 : 1);
};

template <int I, typename... Gs> struct LastEnv;
template <int I> struct LastEnv<I> { static constexpr int value = -1; };
template <int I, typename G, typename... Rest> struct LastEnv<I, G, Rest...> {
  static constexpr int later = LastEnv<I + 1, Rest...>::value;
  static constexpr int value = later >= 0 ? later : (GroupInfo<G>::has_env ? I : -1);
};

// One workgroup = GPW x 64 voices; per 64-voice group one wavefront per stage group (K for a Fan group) + the mixer, or
// without it with PIPE_FOLD.
// GPW = 2 (banks of more voice groups than the chip has CUs): two 64-voice groups share the workgroup's one copy of the sine
// table, each with its own pipeline of wavefronts and its own edge tiles (the short tiles: 64 + 2 x 46 KiB of the 160).  A
// workgroup's wavefronts are dealt round the CU's four SIMDs, so wavefront k of the second group sits beside wavefront k of
// the first: two wavefronts per SIMD, the regime in which a SIMD issues an instruction every two cycles instead of one
// every four (MI355X_MICROARCH.md, Wave scheduling; tools/micro/valu_issue.hip: a second wave on a SIMD runs at the first
// one's rate).  The per-voice arithmetic and the partial rows are those of the one-group form: same bits.
template <typename F, bool FMA, int T, int MODE, int GPW, typename... Gs>
__global__ void __launch_bounds__((GPW * PipeWaves<T, MODE, Gs...>::value * 64)) voice_pipe_kernel(VoiceKernelArgs<F> a) {
  constexpr int NG = (int)sizeof...(Gs);
  constexpr int CHAINW = PipeWaves<T, MODE, Gs...>::chain;  // wavefronts that run stage groups
  constexpr int WAVES = PipeWaves<T, MODE, Gs...>::value;   // wavefronts of ONE 64-voice group
  static_assert(GPW == 1 || GPW == 2, "one or two voice groups per workgroup");
  static_assert(T <= 64 && T % 8 == 0, "a tile column per lane of the folding wavefront");
  constexpr bool kSine = (false || ... || GroupInfo<Gs>::uses_sine);
  // the last edge feeds the mixer; with PIPE_FOLD it is one buffer private to the last group, with PIPE_INPLACE there is none
  // (the edge before it has three buffers instead)
  constexpr long kEdgeElems = (long)EdgeMap<MODE, NG>::tiles * EdgeLayout<F, T>::tile;  // per voice group
  // what is left of the CU's 160 KiB holds the workgroup's events (16 bytes each), up to 2 048 of them
  // a stage group with a delay moves its ring tiles as whole lines through a tile of its own (RingLines)
  constexpr int kRingTiles = T * (int)sizeof(F) >= RingLines<F>::kLine ? (0 + ... + (GroupInfo<Gs>::uses_ring ? 1 : 0)) : 0;
  constexpr long kLdsFree = 160 * 1024 - 1024 - 4096 - 64 - (long)sizeof(float) * (kSine ? 16384 : 4) - (long)sizeof(F) * GPW * kEdgeElems
                            - (long)GPW * kRingTiles * RingLines<F>::kTileBytes;
#ifndef KNH_EVCAP_MAX
#define KNH_EVCAP_MAX 2048
#endif
  constexpr int kEvCap = kLdsFree < 16 ? 0 : (kLdsFree / 16 > KNH_EVCAP_MAX ? KNH_EVCAP_MAX : (int)(kLdsFree / 16));
  // One LDS object, the sine table first: it then sits at LDS address 0, and a table read's address is the masked phase
  // itself (the 16-bit offset field of ds_read_b32 cannot hold the table's address behind 85 KiB of tiles; an add per sample could).
  struct Lds {
    float sine[kSine ? 16384 : 4];
    __attribute__((aligned(16))) F edge[GPW * kEdgeElems];
    __attribute__((aligned(16))) Event ev_stage[kEvCap > 0 ? kEvCap : 1];
    __attribute__((aligned(16))) char ring_tiles[GPW * kRingTiles > 0 ? GPW * kRingTiles * RingLines<F>::kTileBytes : 16];
    u32 res_marks[16 * 64];  // a resident launch: the groups' done marks and running flags of the call (PipeShared)
    u32 res_slot[16];        // ... and its command word
  };
  __shared__ Lds lds;
  float* const sine = lds.sine;
  F* const edge = lds.edge;
  Event* const ev_stage = lds.ev_stage;

  const int lane = threadIdx.x & 63;
  const int wave_all = threadIdx.x >> 6;               // wavefront of the workgroup
  const int grp = GPW > 1 ? wave_all / WAVES : 0;      // its voice group ...
  // ... and its role in that group's pipeline.  Wavefronts k and k + 4 of a workgroup share a SIMD (they are dealt round the
  // four SIMDs in turn), so the second group takes its roles rotated by half a turn: its filter wavefront sits beside the
  // first group's mixer, not beside its filter -- two filter wavefronts on one SIMD would halve each other's packed
  // instructions (tools/micro/valu_issue.hip: with two waves on a SIMD scalar ops keep their rate, packed f32 ops do not).
  const int wave = GPW > 1 ? (wave_all + (grp ? WAVES / 2 : 0)) % WAVES : wave_all;
  u32 ev_first = 0, ev_count = 0;
  if (kEvCap > 0 && a.ev_start && !a.res.bell) {  // (the groups' voices are neighbours: one contiguous piece of the list, which is sorted by voice)
    const u32 gv0 = blockIdx.x * (u32)(GPW * 64);
    const u32 gnv = a.n_voices - gv0 < (u32)(GPW * 64) ? a.n_voices - gv0 : (u32)(GPW * 64);
    ev_first = a.ev_start[gv0];
    ev_count = a.ev_start[gv0 + gnv] - ev_first;
    if (ev_count > (u32)kEvCap) ev_count = 0u;  // does not fit: the groups read the list where it is
    for (u32 i = threadIdx.x; i < ev_count; i += GPW * WAVES * 64) ev_stage[i] = a.events[ev_first + i];
  }
  if (kSine) {
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
#pragma unroll 4
    for (int k = wave_all; k < 64; k += GPW * WAVES) {
      const float* g = a.sine_table + (k * 64 + lane) * 4;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(sine + k * 256), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // a resident launch shares its SIMDs with the fold server's wavefronts (older, and so the winners of every issue slot both
  // want): the voice wavefronts take the higher priority
  if (a.res.bell) __builtin_amdgcn_s_setprio(3);
  PipeShared<F> sh;
  sh.sine = sine;
  sh.edge = edge + (long)grp * kEdgeElems;
  sh.ev_lds = ev_stage;
  sh.ev_lds_first = ev_first;
  sh.ev_lds_n = ev_count;
  sh.res_slot = lds.res_slot;
  sh.res_marks = lds.res_marks;
  sh.ev_stage = ev_stage;
  sh.ev_cap = (u32)(kEvCap > 0 ? kEvCap : 0);
  sh.n_threads = (u32)(GPW * WAVES * 64);
  {
    constexpr bool ring_groups[] = {GroupInfo<Gs>::uses_ring...};
    int k = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool u = i < NG && kRingTiles > 0 && ring_groups[i < NG ? i : 0];
      sh.ring_tile_of[i] = u ? (__attribute__((address_space(3))) char*)(lds.ring_tiles + ((long)grp * kRingTiles + k) * RingLines<F>::kTileBytes) : nullptr;
      k += u ? 1 : 0;
    }
  }
  const u32 wave_global = blockIdx.x * (u32)GPW + (u32)grp;  // the 64-voice group of the bank
  const u32 v0 = wave_global * 64u;
  const bool dead = GPW > 1 && v0 >= a.n_voices;  // the last workgroup of a bank with an odd number of voice groups
  const u32 nv = dead ? 0u : (a.n_voices - v0 < 64u ? a.n_voices - v0 : 64u);
  u32 done_frame = 0xFFFFFFFFu;
  constexpr bool kPan = (false || ... || GroupInfo<Gs>::pan);
  if (dead) {
    // no voices: this group only keeps the workgroup's barriers company, step for step
    const u32 n_frames = a.frame_end - a.frame_begin;
    const int n_steps = (int)((n_frames + T - 1) / T) * (int)a.n_blocks + NG - (MODE == PIPE_FOLD ? 1 : 0);
    for (int s = 0; s < n_steps; ++s) pipe_barrier();
  } else if (wave == CHAINW) {
    constexpr bool kAnyEnvM = (false || ... || GroupInfo<Gs>::has_env);
    pipe_run_mixer<F, FMA, T, MODE, NG, kPan, CHAINW, kAnyEnvM>(sh, a, lane, wave_global, v0, nv, wave_all);
  } else {
    done_frame = pipe_dispatch<F, FMA, T, MODE, NG, 0, 0, 0, LastEnv<0, Gs...>::value, Gs...>(wave, sh, a, lane, wave_global, v0, nv, wave_all);
  }
  if (a.res.bell) return;  // a resident launch has reported call by call (pipe_run_mixer)
  // mark_done of a voice = that of the last node in task order that set one: combine the groups in chain order
  constexpr bool kAnyEnv = (false || ... || GroupInfo<Gs>::has_env);
  if constexpr (kAnyEnv) {
    u32* marks = reinterpret_cast<u32*>(edge) + (long)grp * (CHAINW * 64);  // the tiles are dead: every wavefront is past its last barrier-separated read
    pipe_barrier();
    if (wave < CHAINW) marks[wave * 64 + lane] = done_frame;
    pipe_barrier();
    if (wave == 0 && !dead) {
      u32 d = 0xFFFFFFFFu;
#pragma unroll
      for (int g = 0; g < CHAINW; ++g) d = marks[g * 64 + lane] != 0xFFFFFFFFu ? marks[g * 64 + lane] : d;
      const bool live = (u32)lane < nv;
      if (live) a.done_frames[v0 + lane] = d;
      const u64 bd = __builtin_amdgcn_ballot_w64(live && d != 0xFFFFFFFFu);
      if (lane == 0 && bd) atomicOr(&a.flags[0], 1u);
    }
  }
}

}  // namespace knh_dev
