"""ctypes wrapper of oracle/liboracle.so -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
`OracleBank` has the same call surface as knaster_amd.VoiceBank so a parity test
drives both with identical code.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
KAT_PATH = os.path.join(HERE, "oracle_kat")

TRIGGER = object()
F32, F64 = 0, 1
VALUE_FLOAT, VALUE_TRIGGER, VALUE_INTEGER, VALUE_BOOL = 0, 1, 2, 3


class StageDesc(C.Structure):  # layout of knh_stage_desc
    _fields_ = [("kind", C.c_uint16), ("flags", C.c_uint16), ("delayed_changes_per_block", C.c_uint16),
                ("ar_param", C.c_uint16), ("input", C.c_uint16), ("input2", C.c_uint16)]


def build(force: bool = False) -> None:
    """make -C oracle (gcc, -ffp-contract=off)."""
    srcs = ["knaster_oracle.hpp", "oracle_bank.hpp", "oracle_capi.cpp", "oracle_kat.cpp", "Makefile",
            os.path.join("..", "include", "knaster_hip.h")]
    newest = max(os.path.getmtime(os.path.join(HERE, s)) for s in srcs)
    fresh = all(os.path.exists(p) and os.path.getmtime(p) >= newest for p in (LIB_PATH, KAT_PATH))
    if fresh and not force:
        return
    subprocess.run(["make", "-C", HERE, "-j4", "all"], check=True, capture_output=True)


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        lib = C.CDLL(LIB_PATH)
        vp, u32, i32, f64, i64, u16, sz = C.c_void_p, C.c_uint32, C.c_int, C.c_double, C.c_int64, C.c_uint16, C.c_size_t
        lib.kno_bank_create.restype = vp
        lib.kno_bank_create.argtypes = [C.POINTER(StageDesc), u32, u32, u32, u32, i32, i32]
        lib.kno_bank_destroy.argtypes = [vp]
        lib.kno_bank_last_error.restype = C.c_char_p
        lib.kno_bank_last_error.argtypes = [vp]
        lib.kno_bank_set_ctor_args.argtypes = [vp, u32, u32, u32, vp, u32]
        lib.kno_bank_set_buffer.argtypes = [vp, vp, C.c_size_t, C.c_double]
        lib.kno_bank_set_buffer.restype = C.c_int
        lib.kno_bank_init.argtypes = [vp, u32, sz]
        lib.kno_bank_param_apply.argtypes = [vp, u32, u32, u32, u32, f64, i64]
        lib.kno_bank_set_delay_within_block_for_param.argtypes = [vp, u32, u32, u32, u16]
        lib.kno_bank_param_apply_many.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp]
        lib.kno_bank_schedule.argtypes = [vp, u32, u32, u32, u32, f64, i64, i32, u32, u32]
        lib.kno_bank_process_block.argtypes = [vp, vp, vp, C.POINTER(u32), vp]
        lib.kno_bank_set_in_channels.argtypes = [vp, u32]
        lib.kno_bank_set_in_channels.restype = None
        lib.kno_bank_process_block_in.argtypes = [vp, vp, vp, vp, C.POINTER(u32), vp]
        lib.kno_bank_mix_tasks.restype = sz
        lib.kno_bank_mix_tasks.argtypes = [vp]
        lib.kno_bank_mix_buffer_len.restype = sz
        lib.kno_bank_mix_buffer_len.argtypes = [vp]
        lib.kno_sine_table.argtypes = [vp]
        lib.kno_xorshift32_next.restype = u32
        lib.kno_xorshift32_next.argtypes = [C.POINTER(u32)]
        lib.kno_xorshift32_f32.restype = C.c_float
        lib.kno_xorshift32_f32.argtypes = [C.POINTER(u32)]
        lib.kno_svf_coeffs_f32.argtypes = [u32, C.c_float, C.c_float, C.c_float, C.c_float, vp]
        lib.kno_svf_coeffs_f64.argtypes = [u32, f64, f64, f64, f64, vp]
        lib.kno_seconds_roundtrip.restype = C.c_uint64
        lib.kno_seconds_roundtrip.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        lib.kno_baseline_run.restype = f64
        lib.kno_baseline_run.argtypes = [C.POINTER(StageDesc), u32, u32, u32, u32, C.POINTER(C.c_void_p), u32, sz, u32, u32,
                                         u32, i32, u32, i32, u32, u32, vp]
        _lib = lib
    return _lib


def _stage_array(stages):
    arr = (StageDesc * len(stages))()
    for i, s in enumerate(stages):
        arr[i].kind, arr[i].flags, arr[i].delayed_changes_per_block = s.kind, s.flags, s.delayed_changes_per_block
        arr[i].input, arr[i].input2 = getattr(s, "input", 0), getattr(s, "input2", 0)
        arr[i].ar_param = getattr(s, "ar_param", 0)
    return arr


class OracleBank:
    """Reference-shaped (unfused) graph for the mix + one graph per voice for the per-voice signals."""

    def __init__(self, stages: Sequence, n_voices: int, sample_type: int = F32, out_channels: int = 2,
                 want_mix: bool = True, want_voices: bool = True):
        self._lib = load()
        self.stages = list(stages)
        self.n_voices, self.sample_type, self.out_channels = int(n_voices), sample_type, out_channels
        self.dtype = np.float64 if sample_type == F64 else np.float32
        self.want_mix, self.want_voices = want_mix, want_voices
        self._arr = _stage_array(self.stages)
        self._h = C.c_void_p(self._lib.kno_bank_create(self._arr, len(self.stages), self.n_voices, sample_type,
                                                       out_channels, int(want_mix), int(want_voices)))
        self.block_size = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.kno_bank_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"oracle status {rc}: {(self._lib.kno_bank_last_error(self._h) or b'').decode()}")

    def set_ctor_args(self, stage, args, first_voice=0):
        a = np.ascontiguousarray(np.asarray(args, dtype=np.float64))
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        self._check(self._lib.kno_bank_set_ctor_args(self._h, stage, first_voice, a.shape[0], a.ctypes.data_as(C.c_void_p), a.shape[1]))

    def set_buffer(self, stage, samples, buffer_sample_rate):
        a = np.ascontiguousarray(np.asarray(samples, dtype=self.dtype))
        self._check(self._lib.kno_bank_set_buffer(self._h, a.ctypes.data_as(C.c_void_p), a.shape[0], float(buffer_sample_rate)))

    def init(self, sample_rate, block_size):
        self._check(self._lib.kno_bank_init(self._h, sample_rate, block_size))
        self.sample_rate, self.block_size = sample_rate, block_size

    @staticmethod
    def _value(value):
        if value is TRIGGER or type(value).__name__ == "object":
            return VALUE_TRIGGER, 0.0, 0
        if isinstance(value, (bool, np.bool_)):
            return VALUE_BOOL, 0.0, int(value)
        if isinstance(value, (int, np.integer)):
            return VALUE_INTEGER, 0.0, int(value)
        return VALUE_FLOAT, float(value), 0

    def param_apply(self, voice, stage, param, value):
        k, f, i = self._value(value)
        self._check(self._lib.kno_bank_param_apply(self._h, voice, stage, param, k, f, i))

    def set_delay_within_block_for_param(self, voice, stage, param, delay):
        self._check(self._lib.kno_bank_set_delay_within_block_for_param(self._h, voice, stage, param, delay))

    def param_apply_many(self, voices, stages, params, kinds, fvalues=None, ivalues=None, delays=None):
        v = np.ascontiguousarray(voices, dtype=np.uint32)
        n = v.shape[0]
        bc = lambda a, dt: None if a is None else np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=dt), (n,)))
        s, p, k = bc(stages, np.uint32), bc(params, np.uint32), bc(kinds, np.uint32)
        f, i, d = bc(fvalues, np.float64), bc(ivalues, np.int64), bc(delays, np.uint16)
        ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        self._check(self._lib.kno_bank_param_apply_many(self._h, n, ptr(v), ptr(s), ptr(p), ptr(k), ptr(f), ptr(i), ptr(d)))

    def schedule(self, voice, stage, param, value, time_mode=0, seconds=0, tesimals=0):
        """Through GraphGen's SchedulingEvent path; time_mode 0 none, 1 Time::after, 2 Time::at."""
        k, f, i = self._value(value)
        self._check(self._lib.kno_bank_schedule(self._h, voice, stage, param, k, f, i, time_mode, seconds, tesimals))

    def set_in_channels(self, n: int):
        """UGen::Inputs of the bank node; before init."""
        self._lib.kno_bank_set_in_channels(self._h, int(n))
        self.in_channels = int(n)

    def set_input(self, block):
        """The input block of the next process_block call: [in_channels, B]."""
        self._input = np.ascontiguousarray(np.asarray(block, dtype=self.dtype)).reshape(self.in_channels, self.block_size)

    def process_block(self):
        """-> (out [ch, B], voices [N, B] ([2, N, B] for a chain ending in Pan2) or None, flags, done_frames or None)"""
        out = np.zeros((self.out_channels, self.block_size), dtype=self.dtype)
        pan = bool(self.stages) and self.stages[-1].kind == 32  # KNH_STAGE_PAN2
        shape = (2, self.n_voices, self.block_size) if pan else (self.n_voices, self.block_size)
        voices = np.zeros(shape, dtype=self.dtype) if self.want_voices else None
        done = np.zeros(self.n_voices, dtype=np.uint32) if self.want_voices else None
        flags = C.c_uint32(0)
        if getattr(self, "in_channels", 0):
            self._check(self._lib.kno_bank_process_block_in(
                self._h, self._input.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                None if voices is None else voices.ctypes.data_as(C.c_void_p), C.byref(flags), None if done is None else done.ctypes.data_as(C.c_void_p)))
        else:
            self._check(self._lib.kno_bank_process_block(
                self._h, out.ctypes.data_as(C.c_void_p), None if voices is None else voices.ctypes.data_as(C.c_void_p),
                C.byref(flags), None if done is None else done.ctypes.data_as(C.c_void_p)))
        return out, voices, int(flags.value), done

    def mix_tasks(self):
        return int(self._lib.kno_bank_mix_tasks(self._h))

    def mix_buffer_len(self):
        return int(self._lib.kno_bank_mix_buffer_len(self._h))


def sine_table() -> np.ndarray:
    t = np.zeros(16384, dtype=np.float32)
    load().kno_sine_table(t.ctypes.data_as(C.c_void_p))
    return t


def svf_coeffs(ty, cutoff, q, gain_db, sr, dtype=np.float32) -> np.ndarray:
    out = np.zeros(6, dtype=dtype)
    if dtype == np.float32:
        load().kno_svf_coeffs_f32(ty, cutoff, q, gain_db, sr, out.ctypes.data_as(C.c_void_p))
    else:
        load().kno_svf_coeffs_f64(ty, cutoff, q, gain_db, sr, out.ctypes.data_as(C.c_void_p))
    return out


def baseline_run(stages, n_voices, sample_type, out_channels, ctor: dict, sample_rate, block_size, warmup, blocks, threads,
                 restart=(), release=()):
    """CPU baseline: the unfused reference-shaped graph, `threads` contiguous voice shards.
    Returns (seconds of the block loop, last mixed block)."""
    lib = load()
    arr = _stage_array(stages)
    keep = []
    ptrs = (C.c_void_p * len(stages))()
    for s in range(len(stages)):
        a = ctor.get(s)
        if a is None:
            ptrs[s] = None
        else:
            a = np.ascontiguousarray(a, dtype=np.float64)
            keep.append(a)
            ptrs[s] = a.ctypes.data
    dtype = np.float64 if sample_type == F64 else np.float32
    last = np.zeros((out_channels, block_size), dtype=dtype)
    rs, rp = (restart[0], restart[1]) if restart else (-1, 0)
    ls, lp, lb = (release[0], release[1], release[2]) if release else (-1, 0, 0)
    secs = lib.kno_baseline_run(arr, len(stages), n_voices, sample_type, out_channels, ptrs, sample_rate, block_size, warmup,
                                blocks, threads, rs, rp, ls, lp, lb, last.ctypes.data_as(C.c_void_p))
    return float(secs), last
