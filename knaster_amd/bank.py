"""VoiceBank: a numpy-friendly handle on one `knh_bank` (include/knaster_hip.h).

Mirrors the reference's UGen surface for the bank node -- init / param_apply /
set_delay_within_block_for_param / process_block (knaster_core/src/ugen.rs:232-369) --
and adds nothing of its own: every method is one C-ABI call.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib as L

TRIGGER = object()  # PTrigger


@dataclass(frozen=True)
class Stage:
    kind: int
    flags: int = 0
    delayed_changes_per_block: int = 0  # > 0: wrapped in WrPreciseTiming<N, _>
    input: int = 0   # 0: reads the stage before it; k > 0: reads the output of stage k - 1 (knh_stage_desc.input)
    input2: int = 0  # second operand of the STAGE_MATH_* stages, or the driver of an audio-rate parameter; same numbering
    ar_param: int = 0  # 1 + the float parameter that the signal `input2` drives at audio rate (.ar_params() + link), 0: none


def _stage_array(stages: Sequence[Stage]):
    arr = (L.StageDesc * len(stages))()
    for i, s in enumerate(stages):
        arr[i].kind = s.kind
        arr[i].flags = s.flags
        arr[i].delayed_changes_per_block = s.delayed_changes_per_block
        arr[i].input = s.input
        arr[i].input2 = s.input2
        arr[i].ar_param = s.ar_param
    return arr


def chain_ugen_count(stages: Sequence[Stage]) -> int:
    return int(L.load().knh_chain_ugen_count(_stage_array(stages), len(stages)))


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the library: rank 0 makes it, the host hands it to the other ranks."""
    lib = L.load()
    buf = C.create_string_buffer(L.COMM_ID_BYTES)
    rc = lib.knh_comm_unique_id(buf)
    if rc != L.OK:
        raise L.KnasterHipError(rc, (lib.knh_comm_last_error(None) or b"").decode())
    return buf.raw


def shard_voice_range(n_voices: int, rank: int, world: int):
    """(first, count) of `rank`'s voices: knh_shard_voice_range."""
    lib = L.load()
    first, count = C.c_uint32(0), C.c_uint32(0)
    if lib.knh_shard_voice_range(n_voices, rank, world, C.byref(first), C.byref(count)) != L.OK:
        raise ValueError("rank out of range")
    return int(first.value), int(count.value)


class VoiceBank:
    def __init__(self, stages: Sequence[Stage], n_voices: int, sample_type: int = L.F32, out_channels: int = 2,
                 mix_mode: int = L.MIX_TREE, device: int = -1, allow_fma: bool = False, host_threads: int = 0,
                 devices: Optional[Sequence[int]] = None, rank: Optional[int] = None, world: int = 1,
                 comm_id: Optional[bytes] = None, reduce_fn=None, in_channels: int = 0):
        """devices=[..]: one process, voice ranges on several GPUs (knh_bank_create_multi_device).
        rank/world (+ comm_id from comm_unique_id(), or reduce_fn): one process per GPU, n_voices is the TOTAL
        (knh_bank_create_rank / _custom).  Voice indices are global in both."""
        self._lib = L.load()
        self.stages = list(stages)
        self.n_voices = int(n_voices)
        self.sample_type = sample_type
        self.dtype = np.float64 if sample_type == L.F64 else np.float32
        self.out_channels = out_channels
        self._stage_arr = _stage_array(self.stages)
        self.in_channels = int(in_channels)
        desc = L.BankDesc(L.KNH_ABI_VERSION, self.n_voices, sample_type, len(self.stages), self._stage_arr,
                          out_channels, mix_mode, device, 1 if allow_fma else 0, self.in_channels)
        h = C.c_void_p()
        self._keep = None
        if devices is not None:
            arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            rc = self._lib.knh_bank_create_multi_device(C.byref(desc), arr, len(devices), C.byref(h))
        elif rank is not None and reduce_fn is not None:
            self._keep = L.REDUCE_FN(reduce_fn)  # the callback must outlive the bank
            rc = self._lib.knh_bank_create_rank_custom(C.byref(desc), int(rank), int(world), self._keep, None, C.byref(h))
        elif rank is not None:
            self._keep = C.create_string_buffer(bytes(comm_id), L.COMM_ID_BYTES) if comm_id is not None else None
            rc = self._lib.knh_bank_create_rank(C.byref(desc), int(rank), int(world), self._keep, C.byref(h))
        elif host_threads >= 2:  # host work (change queues, event lists) on several threads: knh_bank_create_sharded
            rc = self._lib.knh_bank_create_sharded(C.byref(desc), int(host_threads), C.byref(h))
        else:
            rc = self._lib.knh_bank_create(C.byref(desc), C.byref(h))
        if rc != L.OK:
            raise L.KnasterHipError(rc, (self._lib.knh_last_error(None) or b"").decode())
        self._h = h
        self.block_size = 0
        self.sample_rate = 0

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.knh_bank_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int):
        if rc != L.OK:
            raise L.KnasterHipError(rc, (self._lib.knh_last_error(self._h) or b"").decode())

    # -- construction -----------------------------------------------------------------------
    def set_ctor_args(self, stage: int, args, first_voice: int = 0):
        """args: [count, n_args] constructor arguments (e.g. SinWt::new(freq)) for consecutive voices."""
        a = np.ascontiguousarray(np.asarray(args, dtype=np.float64))
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        self._check(self._lib.knh_bank_set_ctor_args(self._h, stage, first_voice, a.shape[0],
                                                     a.ctypes.data_as(C.c_void_p), a.shape[1]))

    def set_buffer(self, stage: int, samples, buffer_sample_rate: float):
        """Buffer::from_vec(samples, sample_rate) for the chain's BufferReader stage (single channel)."""
        a = np.ascontiguousarray(np.asarray(samples, dtype=self.dtype))
        self._check(self._lib.knh_bank_set_buffer(self._h, stage, a.ctypes.data_as(C.c_void_p), a.shape[0], float(buffer_sample_rate)))

    def init(self, sample_rate: int, block_size: int):
        self._check(self._lib.knh_bank_init(self._h, sample_rate, block_size))
        self.sample_rate, self.block_size = sample_rate, block_size

    # -- UGen surface -----------------------------------------------------------------------
    def inputs(self) -> int:
        return int(self._lib.knh_bank_inputs(self._h))

    def outputs(self) -> int:
        return int(self._lib.knh_bank_outputs(self._h))

    def stage_parameters(self, stage: int) -> int:
        return int(self._lib.knh_bank_stage_parameters(self._h, stage))

    def stage_param_descriptions(self, stage: int):
        out = []
        for p in range(self.stage_parameters(stage)):
            d = self._lib.knh_bank_stage_param_description(self._h, stage, p)
            out.append(d.decode() if d else None)
        return out

    @staticmethod
    def _value(value):
        if value is TRIGGER:
            return L.VALUE_TRIGGER, 0.0, 0
        if isinstance(value, (bool, np.bool_)):
            return L.VALUE_BOOL, 0.0, int(value)
        if isinstance(value, (int, np.integer)):
            return L.VALUE_INTEGER, 0.0, int(value)
        return L.VALUE_FLOAT, float(value), 0

    def param_apply(self, voice: int, stage: int, param: int, value):
        kind, f, i = self._value(value)
        self._check(self._lib.knh_bank_param_apply(self._h, voice, stage, param, kind, f, i))

    def param(self, voice: int, stage: int, param, value):
        """UGen::param: index or description -> param_apply (ugen.rs:344-368)."""
        if isinstance(param, str):
            descs = self.stage_param_descriptions(stage)
            if param not in descs:
                raise KeyError(f"DescriptionNotFound({param!r})")
            param = descs.index(param)
        self.param_apply(voice, stage, param, value)

    def set_delay_within_block_for_param(self, voice: int, stage: int, param: int, delay: int):
        self._check(self._lib.knh_bank_set_delay_within_block_for_param(self._h, voice, stage, param, delay))

    def param_apply_many(self, voices, stages, params, kinds, fvalues=None, ivalues=None, delays=None, block_offset=0):
        self.param_apply_prepared(self.prepare_many(voices, stages, params, kinds, fvalues, ivalues, delays), block_offset)

    def param_apply_range(self, voice_begin, voice_end, stage, param, kind, fvalue=0.0, ivalue=0):
        """One parameter of the voices [voice_begin, voice_end), in rising order: knh_bank_param_apply_range."""
        self._check(self._lib.knh_bank_param_apply_range(self._h, voice_begin, voice_end, stage, param, kind, float(fvalue), int(ivalue)))

    @staticmethod
    def prepare_many(voices, stages, params, kinds, fvalues=None, ivalues=None, delays=None):
        """The argument arrays of knh_bank_param_apply_many[_at] in the C layout, made once (a caller that sends the same
        kind of batch every block keeps the result instead of converting numpy arrays on every call)."""
        v = np.ascontiguousarray(voices, dtype=np.uint32)
        n = v.shape[0]
        s = np.ascontiguousarray(np.broadcast_to(np.asarray(stages, dtype=np.uint32), (n,)))
        p = np.ascontiguousarray(np.broadcast_to(np.asarray(params, dtype=np.uint32), (n,)))
        k = np.ascontiguousarray(np.broadcast_to(np.asarray(kinds, dtype=np.uint32), (n,)))
        f = None if fvalues is None else np.ascontiguousarray(np.broadcast_to(np.asarray(fvalues, dtype=np.float64), (n,)))
        i = None if ivalues is None else np.ascontiguousarray(np.broadcast_to(np.asarray(ivalues, dtype=np.int64), (n,)))
        d = None if delays is None else np.ascontiguousarray(np.broadcast_to(np.asarray(delays, dtype=np.uint16), (n,)))
        arrays = (v, s, p, k, f, i, d)
        return n, arrays, tuple(None if a is None else a.ctypes.data_as(C.c_void_p) for a in arrays)

    def param_apply_prepared(self, batch, block_offset=0):
        n, _arrays, ptrs = batch
        if block_offset:
            self._check(self._lib.knh_bank_param_apply_many_at(self._h, block_offset, n, *ptrs))
        else:
            self._check(self._lib.knh_bank_param_apply_many(self._h, n, *ptrs))

    def set_input(self, blocks):
        """The bank node's input block(s) for the next process call: [n_blocks, in_channels, block_size] (or one block
        [in_channels, block_size]) -- knh_bank_set_input."""
        a = np.ascontiguousarray(np.asarray(blocks, dtype=self.dtype))
        if a.ndim == 2:
            a = a.reshape(1, *a.shape)
        assert a.shape[1:] == (self.in_channels, self.block_size), a.shape
        self._check(self._lib.knh_bank_set_input(self._h, a.shape[0], a.ctypes.data_as(C.c_void_p)))

    def process_block(self, frames_to_process: Optional[int] = None, block_start_offset: int = 0, frame_clock: int = 0,
                      out: Optional[np.ndarray] = None):
        ftp = self.block_size if frames_to_process is None else frames_to_process
        if out is None:
            out = np.zeros((self.out_channels, self.block_size), dtype=self.dtype)
        flags = C.c_uint32(0)
        self._check(self._lib.knh_bank_process_block(self._h, ftp, block_start_offset, frame_clock,
                                                     out.ctypes.data_as(C.c_void_p), C.byref(flags)))
        return out, int(flags.value)

    def process_block_voices(self, frames_to_process: Optional[int] = None, block_start_offset: int = 0, frame_clock: int = 0):
        ftp = self.block_size if frames_to_process is None else frames_to_process
        out = np.zeros((self.out_channels, self.block_size), dtype=self.dtype)
        # a chain that ends in Pan2 has a left and a right signal per voice: [2][n_voices][block_size]
        pan = bool(self.stages) and self.stages[-1].kind == L.STAGE_PAN2
        voices = np.zeros((2, self.n_voices, self.block_size) if pan else (self.n_voices, self.block_size), dtype=self.dtype)
        flags = C.c_uint32(0)
        self._check(self._lib.knh_bank_process_block_voices(self._h, ftp, block_start_offset, frame_clock,
                                                            out.ctypes.data_as(C.c_void_p),
                                                            voices.ctypes.data_as(C.c_void_p), C.byref(flags)))
        return out, voices, int(flags.value)

    def process_block_device(self, out_device_ptr: int = 0, hip_stream: int = 0, frames_to_process: Optional[int] = None,
                             block_start_offset: int = 0, frame_clock: int = 0):
        ftp = self.block_size if frames_to_process is None else frames_to_process
        self._check(self._lib.knh_bank_process_block_device(self._h, ftp, block_start_offset, frame_clock,
                                                            C.c_void_p(out_device_ptr or None), C.c_void_p(hip_stream or None)))

    def process_blocks(self, n_blocks: int, frame_clock: int = 0):
        """n_blocks whole blocks in one launch -> (out [n_blocks, ch, B], flags)."""
        out = np.zeros((n_blocks, self.out_channels, self.block_size), dtype=self.dtype)
        flags = C.c_uint32(0)
        self._check(self._lib.knh_bank_process_blocks(self._h, n_blocks, frame_clock, out.ctypes.data_as(C.c_void_p), C.byref(flags)))
        return out, int(flags.value)

    def process_blocks_begin(self, n_blocks: int, frame_clock: int = 0):
        """Enqueue a launch and the copy of its blocks to pinned host memory; fetch them with process_blocks_end (up to two
        may be outstanding)."""
        self._check(self._lib.knh_bank_process_blocks_begin(self._h, n_blocks, frame_clock))
        self._begun = getattr(self, "_begun", []) + [n_blocks]

    def process_blocks_end(self):
        """-> out [n_blocks, ch, B] of the oldest outstanding launch."""
        n_blocks = self._begun.pop(0)
        out = np.zeros((n_blocks, self.out_channels, self.block_size), dtype=self.dtype)
        self._check(self._lib.knh_bank_process_blocks_end(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def process_blocks_device(self, n_blocks: int, out_device_ptr: int = 0, hip_stream: int = 0, frame_clock: int = 0):
        self._check(self._lib.knh_bank_process_blocks_device(self._h, n_blocks, frame_clock, C.c_void_p(out_device_ptr or None),
                                                             C.c_void_p(hip_stream or None)))

    def read_done_frames(self) -> np.ndarray:
        d = np.zeros(self.n_voices, dtype=np.uint32)
        self._check(self._lib.knh_bank_read_done_frames(self._h, d.ctypes.data_as(C.c_void_p)))
        return d

    def debug_signature(self) -> str:
        """the chain as the device code names it (knh_bank_debug_signature)"""
        return self._lib.knh_bank_debug_signature(self._h).decode()

    def resident_stats(self):
        """(calls served by a resident kernel, times one was launched): knh_bank_resident_stats"""
        c, n = C.c_uint64(0), C.c_uint64(0)
        self._check(self._lib.knh_bank_resident_stats(self._h, C.byref(c), C.byref(n)))
        return int(c.value), int(n.value)

    def debug_words(self) -> np.ndarray:
        d = np.zeros(16, dtype=np.uint32)
        self._check(self._lib.knh_bank_debug_words(self._h, d.ctypes.data_as(C.c_void_p)))
        return d

    def synchronize(self):
        self._check(self._lib.knh_bank_synchronize(self._h))

    def ranks(self) -> int:
        """Ranks RCCL reports (rank banks), voice ranges (multi-device / host-sharded banks), else 1."""
        return int(self._lib.knh_bank_ranks(self._h))

    def timing_reset(self, enable: bool = True):
        self._check(self._lib.knh_bank_timing_reset(self._h, 1 if enable else 0))

    def timing_read(self):
        ms, n = C.c_double(0), C.c_uint64(0)
        self._check(self._lib.knh_bank_timing_read(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def collective_timing_read(self):
        """(ms, count) of the sums across GPUs since the last timing_reset (rank banks; 0, 0 otherwise)."""
        ms, n = C.c_double(0), C.c_uint64(0)
        self._check(self._lib.knh_bank_collective_timing_read(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def algorithmic_bytes_per_voice_block(self):
        r, w = C.c_uint32(0), C.c_uint32(0)
        self._check(self._lib.knh_bank_algorithmic_bytes_per_voice_block(self._h, C.byref(r), C.byref(w)))
        return int(r.value), int(w.value)
