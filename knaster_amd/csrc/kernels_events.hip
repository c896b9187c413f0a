// kernels_events.hip -- WrPreciseTiming's change queues resolved ON THE DEVICE (knaster_core_dsp/src/wrappers_core/
// precise_timing.rs:14-149), for nodes whose parameter setters need no host library call.
//
// Why: BASELINE config C5 sends every voice a sample-accurate change every second block -- 2 048 delayed changes per
// 128-frame block against a 6.7 us kernel.  Turning each call into device events on the host (armed delay, queue position,
// capacity, head-of-line blocking, the patch itself, then a counting sort by voice into the per-voice lists) cost ~19 ns per
// change on one core: the bench ran at a sixth of its kernel's rate.  Here the host only appends one 24-byte record per call
// to a pinned buffer; four small kernels on a side stream, overlapping the previous launch's voice kernel, do the rest:
//   count    records per voice (atomics); the records are copied to device memory on the way (their one trip over PCIe)
//   scan     exclusive sums: where each voice's records and its output events start
//   scatter  record keys (block, arrival index) into the voice's segment
//   resolve  one thread per voice: sorts its keys, replays the records in arrival order against the armed delays (device
//            state, persistent) and each wrapped node's queue of the block -- first in first out, a change behind one that is
//            not due inside the processed range is never reached, changes beyond the capacity are dropped -- computes the
//            patches (integer / f64 arithmetic the reference's setters do: osc.rs:127-135,240-247, util.rs:47-50,
//            envelopes.rs:85-133), and merges them with the events the host made for the voice's other nodes.
// The result is the same CSR list, in device memory, that the host path builds: bit-identical voices (tests/
// test_gpu_event_fuzz.py runs both).  Warnings (a full queue) are not reported from here.
#include <hip/hip_runtime.h>

#include "../../include/knaster_hip.h"
#include "kernel_registry.hpp"

namespace knh {
using namespace knh_dev;

namespace {

__global__ void __launch_bounds__(256) ev_count_kernel(EventResolveArgs a) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= a.n_recs) return;
  const DevRec r = a.recs[i];
  a.dev_recs[i] = r;  // the later passes read device memory
  if (r.voice >= a.n_voices || r.block >= a.n_blocks) return;  // (the host has checked; a record for a later launch is not in this buffer)
  atomicAdd(&a.cnt[r.voice], 1u);
  if (r.kb & 0x20u) atomicAdd(&a.val_cnt[r.voice], 1u);
}

// one workgroup: exclusive sums over the voices
__global__ void __launch_bounds__(1024) ev_scan_kernel(EventResolveArgs a) {
  __shared__ u32 part_r[1024], part_o[1024];
  const u32 t = threadIdx.x, chunk = (a.n_voices + 1023u) / 1024u;
  const u32 lo = t * chunk, hi = lo + chunk < a.n_voices ? lo + chunk : a.n_voices;
  u32 sr = 0, so = 0;
  for (u32 v = lo; v < hi; ++v) {
    sr += a.cnt[v];
    so += a.val_cnt[v] + (a.host_start ? a.host_start[v + 1] - a.host_start[v] : 0u);
  }
  part_r[t] = sr;
  part_o[t] = so;
  __syncthreads();
  // inclusive scan of the 1 024 partial sums (Hillis-Steele in LDS), then shifted to exclusive
  for (u32 d = 1; d < 1024u; d <<= 1) {
    const u32 xr = t >= d ? part_r[t - d] : 0u, xo = t >= d ? part_o[t - d] : 0u;
    __syncthreads();
    part_r[t] += xr;
    part_o[t] += xo;
    __syncthreads();
  }
  if (t == 1023u) {
    a.rec_start[a.n_voices] = part_r[t];
    a.out_start[a.n_voices] = part_o[t];
  }
  const u32 excl_r = part_r[t] - sr, excl_o = part_o[t] - so;
  __syncthreads();
  part_r[t] = excl_r;
  part_o[t] = excl_o;
  u32 ar = part_r[t], ao = part_o[t];
  for (u32 v = lo; v < hi; ++v) {
    a.rec_start[v] = ar;
    a.out_start[v] = ao;
    ar += a.cnt[v];
    ao += a.val_cnt[v] + (a.host_start ? a.host_start[v + 1] - a.host_start[v] : 0u);
  }
}

__global__ void __launch_bounds__(256) ev_scatter_kernel(EventResolveArgs a) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= a.n_recs) return;
  const DevRec r = a.dev_recs[i];
  if (r.voice >= a.n_voices || r.block >= a.n_blocks) return;
  const u32 pos = a.rec_start[r.voice] + atomicAdd(&a.cursor[r.voice], 1u);
  a.keys[pos] = ((u64)r.block << 32) | (u64)i;
}

// Rust `as u32` from f64 (saturating; NaN -> 0): v_cvt_u32_f64 does exactly that
__device__ __forceinline__ u32 ev_sat_u32(double v) {
  u32 r;
  asm("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ u64 ev_fbits(double f, bool f64) {  // F::new(f) as the slot word
  return f64 ? __builtin_bit_cast(u64, f) : (u64)__builtin_bit_cast(u32, (float)f);
}
// The setter of (kind, param) for `value`, restated as one device event; false: the call patches nothing (a parameter that
// an audio-rate signal drives ignores ordinary changes, audio_rate.rs:70-74).  Mirrors Bank::apply_now (bank.hip).
__device__ __forceinline__ bool ev_patch(const DevStage& S, u32 param, u64 value, const EventResolveArgs& a, u32& op, u32& slot, u64& bits) {
  const double f = __builtin_bit_cast(double, value);
  const bool f64 = a.f64 != 0u;
  if (S.ar_param != 0 && param + 1u == S.ar_param) return false;
  op = EV_SET;
  switch (S.kind) {
    case KNH_STAGE_SIN_WT:
      if (param == 0) {  // osc.rs:127-130
        if (S.flags & KNH_STAGE_FLAG_AR_FREQ) return false;
        const double fq = f64 ? f : (double)(float)f;
        slot = S.slot_base + 2u; bits = ev_sat_u32(fq * a.f2pi);
      } else if (param == 1) {  // osc.rs:133-135
        slot = S.slot_base + 1u; bits = ev_sat_u32(f * 65536.0);
      } else {
        slot = S.slot_base; bits = 0;  // reset_phase
      }
      return true;
    case KNH_STAGE_SIN_NUMERIC:
      if (param == 0) {  // osc.rs:240-242: F::new(freq) / F::new(sample_rate as f32)
        slot = S.slot_base + 2u;
        bits = f64 ? __builtin_bit_cast(u64, f / (double)(float)a.sample_rate) : (u64)__builtin_bit_cast(u32, (float)f / (float)a.sample_rate);
      } else if (param == 1) {
        slot = S.slot_base + 1u; bits = ev_fbits(f, f64);
      } else {
        slot = S.slot_base; bits = ev_fbits(0.0, f64);
      }
      return true;
    case KNH_STAGE_MUL_ENV_ASR:
    case KNH_STAGE_MUL_ENV_AR:
      if (param <= 1) {  // envelopes.rs:85-110: rate = 1 / (seconds * sample_rate), 1 for zero seconds (the "unchanged" skip recomputes the same number)
        slot = S.slot_base + 2u + param;
        if (f64) { const double s = f; bits = __builtin_bit_cast(u64, s == 0.0 ? 1.0 : 1.0 / (s * (double)a.sample_rate)); }
        else { const float s = (float)f; bits = (u64)__builtin_bit_cast(u32, s == 0.0f ? 1.0f : 1.0f / (s * (float)a.sample_rate)); }
      } else if (S.kind == KNH_STAGE_MUL_ENV_ASR && param == 2) {
        op = EV_ENV_ASR_RELEASE; slot = S.slot_base; bits = 0;  // t_release needs the live state: a device op
      } else {
        slot = S.slot_base; bits = 1;  // t_restart: state = Attacking, t untouched
      }
      return true;
    default:  // Constant::value / WrMul "wr_mul": F::new(value)
      slot = S.slot_base; bits = ev_fbits(f, f64);
      return true;
  }
}

constexpr int kMaxWrapped = 8;  // device-resolved WrPreciseTiming nodes per voice

__global__ void __launch_bounds__(64) ev_resolve_kernel(EventResolveArgs a) {
  const u32 v = blockIdx.x * 64u + threadIdx.x;
  if (v >= a.n_voices) return;
  // the counting passes are through with this voice's counters: zero them for the next launch (no clearing command per launch)
  a.cnt[v] = 0u;
  a.val_cnt[v] = 0u;
  a.cursor[v] = 0u;
  u64* keys = a.keys + a.rec_start[v];
  const u32 n = a.rec_start[v + 1] - a.rec_start[v];
  // arrival order: (block, index) ascending.  A voice's records of one launch are few (BASELINE config C5: sixteen): up to
  // kSortLds of them are sorted in LDS (a column per thread: no bank conflicts), longer segments in place with a Shell sort.
  constexpr u32 kSortLds = 32;
  __shared__ u64 lk[kSortLds * 64];
  const u32 tid = threadIdx.x;
  const bool in_lds = n <= kSortLds;
  if (in_lds) {
    for (u32 i = 0; i < n; ++i) {
      const u64 k = keys[i];
      u32 j = i;
      for (; j > 0u && lk[(j - 1u) * 64u + tid] > k; --j) lk[j * 64u + tid] = lk[(j - 1u) * 64u + tid];
      lk[j * 64u + tid] = k;
    }
  } else {
    for (u32 gap = n / 2u; gap > 0u; gap /= 2u)
      for (u32 i = gap; i < n; ++i) {
        const u64 k = keys[i];
        u32 j = i;
        for (; j >= gap && keys[j - gap] > k; j -= gap) keys[j] = keys[j - gap];
        keys[j] = k;
      }
  }
  const u32 h0 = a.host_start ? a.host_start[v] : 0u, h1 = a.host_start ? a.host_start[v + 1] : 0u;
  const u32 n_host = h1 - h0;
  Event* out = a.out_events + a.out_start[v];
  const u32 cap = a.out_start[v + 1] - a.out_start[v];
  Event* dev = out + n_host;  // the device-made events first go behind a gap the size of the host's list, then merge forward
  u32 nd = 0;
  struct Q { u32 at, taken, blocked; } q[kMaxWrapped];
  u32 cur_block = 0xFFFFFFFFu, block_first = 0;
  auto sort_block = [&](u32 from, u32 to) {  // stable, by frame: the block's immediate changes (its first frame), then the queued ones as due
    for (u32 i = from + 1u; i < to; ++i) {
      const Event e = dev[i];
      u32 j = i;
      for (; j > from && dev[j - 1u].frame > e.frame; --j) dev[j] = dev[j - 1u];
      dev[j] = e;
    }
  };
  for (u32 k = 0; k < n; ++k) {
    const DevRec r = a.dev_recs[(u32)(in_lds ? lk[k * 64u + tid] : keys[k])];
    if (r.block != cur_block) {
      if (cur_block != 0xFFFFFFFFu) sort_block(block_first, nd);
      cur_block = r.block;
      block_first = nd;
      for (int w = 0; w < kMaxWrapped; ++w) q[w] = Q{a.frame_begin, 0u, 0u};
    }
    const DevStage S = a.stages[r.stage];
    unsigned short* armed_p = a.armed + (size_t)(S.param_base + r.param) * a.n_voices + v;
    u32 armed = *armed_p;
    if (r.kb & 0x10u) { armed = r.delay; *armed_p = r.delay; }  // set_delay_within_block_for_param, precise_timing.rs:146-148
    if (!(r.kb & 0x20u)) continue;
    const u32 frame_base = r.block * a.block_size;
    u32 op = 0, slot = 0;
    u64 bits = 0;
    const bool patched = ev_patch(S, r.param, r.value, a, op, slot, bits);
    if (armed == 0u) {  // no delay armed: straight through, before the block (precise_timing.rs:126-135)
      if (patched && nd < cap - n_host) { dev[nd].frame = frame_base; dev[nd].slot_op = (slot & 0xFFFFFFu) | (op << 24); dev[nd].bits = bits; ++nd; }
      continue;
    }
    Q& qq = q[S.widx < kMaxWrapped ? S.widx : 0];
    if (qq.taken >= S.dcpb) {  // the queue was full when this change arrived (:129-134): dropped, and the host is told (once per launch is enough)
      if (qq.taken == S.dcpb && a.overflow) *a.overflow = 1u;
      qq.taken = S.dcpb + 1u;
      continue;
    }
    qq.taken += 1u;
    if (qq.blocked) continue;  // behind a change that is not due in this block: never reached
    const u32 due = armed > qq.at ? armed : qq.at;
    if (due > a.frame_end) { qq.blocked = 1u; continue; }
    qq.at = due;
    const bool split = due > a.frame_begin;  // the node's block restarts here (:104-110)
    if (!patched) {
      if (!split) continue;
      op = EV_NOP; slot = S.slot_base; bits = 0;
    }
    if (split) op |= EV_SPLIT;
    if (nd < cap - n_host) { dev[nd].frame = frame_base + due; dev[nd].slot_op = (slot & 0xFFFFFFu) | (op << 24); dev[nd].bits = bits; ++nd; }
  }
  if (cur_block != 0xFFFFFFFFu) sort_block(block_first, nd);
  // merge with the host's events for this voice (other nodes: the order among equal frames does not matter)
  u32 i = 0, j = 0, w = 0;
  const Event* host = a.host_events + h0;
  while (i < n_host || j < nd) {
    if (j >= nd || (i < n_host && host[i].frame <= dev[j].frame)) out[w++] = host[i++];
    else out[w++] = dev[j++];
  }
  for (; w < cap; ++w) { out[w].frame = 0xFFFFFFFEu; out[w].slot_op = (u32)EV_NOP << 24; out[w].bits = 0; }  // never reached
}

}  // namespace

hipError_t launch_resolve_events(const EventResolveArgs& a, hipStream_t s) {
  if (a.n_voices == 0) return hipSuccess;
  // (cnt, val_cnt, cursor are zero: cleared when they were allocated, and by the last pass of the launch before)
  if (a.n_recs) hipLaunchKernelGGL(ev_count_kernel, dim3((a.n_recs + 255u) / 256u), dim3(256), 0, s, a);
  hipLaunchKernelGGL(ev_scan_kernel, dim3(1), dim3(1024), 0, s, a);
  if (a.n_recs) hipLaunchKernelGGL(ev_scatter_kernel, dim3((a.n_recs + 255u) / 256u), dim3(256), 0, s, a);
  hipLaunchKernelGGL(ev_resolve_kernel, dim3((a.n_voices + 63u) / 64u), dim3(64), 0, s, a);
  return hipGetLastError();
}

}  // namespace knh
