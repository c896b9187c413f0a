"""Shared test plumbing: drive the HIP bank and the CPU oracle with identical calls."""
from __future__ import annotations

import numpy as np

from knaster_amd import _lib as L
from knaster_amd import configs


def make_oracle(oracle, w: configs.Workload, want_mix=True, want_voices=True):
    b = oracle.OracleBank(w.stages, w.n_voices, w.sample_type, w.out_channels, want_mix, want_voices)
    if getattr(w, "in_channels", 0):
        b.set_in_channels(w.in_channels)
    for s, a in w.ctor.items():
        b.set_ctor_args(s, a)
    if w.buffer is not None:
        b.set_buffer(*w.buffer)
    b.init(configs.SAMPLE_RATE, w.block_size)
    return b


def make_gpu(knh, w: configs.Workload, mix_mode=L.MIX_TREE, allow_fma=False, host_threads=0, **kw):
    """kw: devices=[..] (voice ranges on several GPUs) or rank=, world= (one rank's share), as VoiceBank takes them."""
    b = knh.VoiceBank(w.stages, w.n_voices, w.sample_type, w.out_channels, mix_mode, -1, allow_fma, host_threads,
                      in_channels=getattr(w, "in_channels", 0), **kw)
    for s, a in w.ctor.items():
        b.set_ctor_args(s, a)
    if w.buffer is not None:
        b.set_buffer(*w.buffer)
    b.init(configs.SAMPLE_RATE, w.block_size)
    return b


def fire_all(bank, n_voices, stage, param):
    """One trigger per voice, in voice order (param_apply_many on both backends)."""
    bank.param_apply_many(np.arange(n_voices, dtype=np.uint32), stage, param, L.VALUE_TRIGGER)


def bits(a: np.ndarray) -> np.ndarray:
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


def assert_bit_equal(a: np.ndarray, b: np.ndarray, what="", strict_zero=False):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, f"{what}: shape/dtype {a.shape}{a.dtype} vs {b.shape}{b.dtype}"
    # -0.0 and +0.0 are the same sample; everything else must match to the bit.  strict_zero: the sign of a zero too (the
    # tests of code that argues about it: the low-pass filter's shortened step, voice_stages.hpp Svf::tick_tile_low)
    same = (bits(a) == bits(b)) | ((a == 0) & (b == 0) & (not strict_zero))
    if not same.all():
        idx = np.argwhere(~same)[0]
        raise AssertionError(f"{what}: {np.count_nonzero(~same)} of {a.size} samples differ; first at {tuple(idx)}: "
                             f"{a[tuple(idx)]!r} vs {b[tuple(idx)]!r}")


def f64_mix(voices: np.ndarray) -> np.ndarray:
    """The mix accumulated in f64 (the tolerance reference of SURVEY.md hard part 3)."""
    return voices.astype(np.float64).sum(axis=0)


def pairwise_sum(rows: np.ndarray) -> np.ndarray:
    """KNH_MIX_TREE as the header defines it: neighbours are added in pairs, level by level, a row without a right neighbour
    passes through -- in the rows' own precision (numpy adds elementwise with IEEE rounding, like the device)."""
    level = np.ascontiguousarray(rows)
    while level.shape[0] > 1:
        even = level.shape[0] // 2 * 2
        nxt = level[0:even:2] + level[1:even:2]
        level = np.concatenate([nxt, level[even:]], axis=0) if even < level.shape[0] else nxt
    return level[0]
