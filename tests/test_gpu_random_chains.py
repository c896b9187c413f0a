"""Randomised chains: every feedback-free chain of supported stages is fused at run time (hiprtc) when no pre-built
kernel exists, so the space of kernels is open-ended.  Seeded random chains with random constructor arguments and a
few random parameter changes, each compared with the oracle bit for bit per voice (exact stages only)."""
import os

import numpy as np
import pytest

from helpers import assert_bit_equal, make_gpu, make_oracle
from knaster_amd import _lib as L
from knaster_amd import configs
from knaster_amd.bank import Stage

pytestmark = pytest.mark.gpu

# (kind, wrapper?, number of float parameters that can be changed at run time: (param index, low, high))
PROCESSORS = [
    (L.STAGE_SVF, [(0, 200.0, 6000.0), (1, 0.5, 4.0)]),
    (L.STAGE_ONEPOLE_LPF, [(0, 100.0, 8000.0)]),
    (L.STAGE_ONEPOLE_HPF, [(0, 100.0, 8000.0)]),
    (L.STAGE_MUL_CONST, [(0, -1.0, 1.0)]),
    (L.STAGE_ADD_CONST, [(0, -0.5, 0.5)]),
    (L.STAGE_SUB_CONST, [(0, -0.5, 0.5)]),
    (L.STAGE_DIV_CONST, [(0, 1.0, 3.0)]),
    (L.STAGE_MUL_ENV_ASR, []),
    (L.STAGE_MUL_ENV_AR, []),
    (L.STAGE_SAMPLE_DELAY, [(0, 0.0, 0.004)]),
    (L.STAGE_MUL_ENVELOPE, [(0, 0.5, 2.0)]),
    (L.STAGE_SAFETY_LIMITER, []),
    (L.STAGE_ALLPASS_DELAY, [(0, 0.0001, 0.004)]),
    (L.STAGE_ALLPASS_FB_DELAY, [(0, 0.0001, 0.004), (1, -0.8, 0.8)]),
]
WRAPPERS = [L.STAGE_WR_MUL, L.STAGE_WR_ADD, L.STAGE_WR_SUB, L.STAGE_WR_VSUB, L.STAGE_WR_DIV, L.STAGE_WR_POWI]


def ctor_for(kind, n, rng, p):
    if kind == L.STAGE_SIN_WT:
        return p["freq"].reshape(n, 1)
    if kind == L.STAGE_SVF:
        ty = rng.integers(0, 9, n).astype(np.float64)
        return np.stack([ty, p["cutoff"], p["q"], rng.uniform(-6.0, 6.0, n)], axis=1)
    if kind == L.STAGE_ONEPOLE_LPF:
        return p["cutoff"].reshape(n, 1)
    if kind in (L.STAGE_ONEPOLE_HPF, L.STAGE_SAFETY_LIMITER):
        return None
    if kind in (L.STAGE_MUL_ENV_ASR, L.STAGE_MUL_ENV_AR):
        return np.stack([p["attack"] * 0.2, p["release"] * 0.02], axis=1)
    if kind in (L.STAGE_SAMPLE_DELAY, L.STAGE_ALLPASS_DELAY, L.STAGE_ALLPASS_FB_DELAY):
        return rng.uniform(0.0045, 0.006, (n, 1))
    if kind == L.STAGE_MUL_ENVELOPE:  # [start, time_scale, looping, n_segments, (duration, value) * 4]
        a = np.zeros((n, 12))
        a[:, 0] = rng.uniform(-0.5, 0.5, n)
        a[:, 1] = rng.uniform(0.5, 2.0, n)
        a[:, 2] = rng.integers(0, 2, n)
        a[:, 3] = rng.integers(1, 5, n)
        a[:, 4::2] = rng.uniform(0.0005, 0.003, (n, 4))
        a[:, 5::2] = rng.uniform(-1.0, 1.0, (n, 4))
        return a
    if kind == L.STAGE_WR_POWI:
        return rng.integers(0, 4, (n, 1)).astype(np.float64)  # negative exponents turn the oscillator's zeros into inf and then NaN, whose bit pattern is platform-specific
    if kind in (L.STAGE_DIV_CONST, L.STAGE_WR_DIV):
        return rng.uniform(1.0, 3.0, (n, 1))
    return rng.uniform(-1.0, 1.0, (n, 1))


def random_chain(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(65, 200))
    bs = int(rng.choice([32, 64, 100, 128]))
    st = L.F64 if seed % 4 == 3 else L.F32
    p = configs.voice_parameters(n)
    src = L.STAGE_PHASOR if seed % 5 == 4 else (L.STAGE_POLYBLEP if seed % 5 == 3 else L.STAGE_SIN_WT)
    if seed % 10 == 2:
        src = [L.STAGE_WHITE_NOISE, L.STAGE_PINK_NOISE, L.STAGE_BROWN_NOISE][(seed // 10) % 3]
    src_args = ctor_for(L.STAGE_SIN_WT, n, rng, p)  # a frequency per voice
    if src in (L.STAGE_WHITE_NOISE, L.STAGE_PINK_NOISE, L.STAGE_BROWN_NOISE):
        src_args = (1000.0 * seed + np.arange(n)).reshape(n, 1)  # a seed per voice
    if src == L.STAGE_POLYBLEP:  # waveforms without sin: those are bit-exact
        wf = rng.choice([0, 3, 4, 5, 6, 7, 8, 11, 12, 13], n).astype(np.float64)
        src_args = np.stack([wf, src_args[:, 0]], axis=1)
    stages, ctor, changes, triggers = [Stage(src)], {0: src_args}, [], []
    have_delay = have_segenv = False
    for _ in range(int(rng.integers(2, 6))):
        if rng.random() < 0.3:
            kind = int(rng.choice(WRAPPERS))
            stages.append(Stage(kind))
            ctor[len(stages) - 1] = ctor_for(kind, n, rng, p)
            continue
        kind, params = PROCESSORS[int(rng.integers(0, len(PROCESSORS)))]
        if kind in (L.STAGE_SAMPLE_DELAY, L.STAGE_ALLPASS_DELAY, L.STAGE_ALLPASS_FB_DELAY):
            if have_delay:
                continue
            have_delay = True
        if kind == L.STAGE_MUL_ENVELOPE:
            if have_segenv:
                continue
            have_segenv = True
        precise = int(rng.integers(0, 3)) if kind != L.STAGE_MUL_CONST else 0
        stages.append(Stage(kind, delayed_changes_per_block=precise))
        s = len(stages) - 1
        c = ctor_for(kind, n, rng, p)
        if c is not None:
            ctor[s] = c
        for (pi, lo, hi) in params:
            changes.append((s, pi, lo, hi, precise > 0))
        if kind == L.STAGE_MUL_ENV_ASR:
            triggers.append((s, 3, 2))
        if kind == L.STAGE_MUL_ENV_AR:
            triggers.append((s, 2, None))
        if kind == L.STAGE_MUL_ENVELOPE:
            triggers.append((s, 2, 3))  # t_restart, t_stop
    w = configs.Workload(f"random{seed}", stages, n, bs, st, 2)
    w.ctor = ctor
    return w, rng, changes, triggers


# (KNH_TEST_SEEDS=400 python -m pytest tests/test_gpu_random_chains.py -m gpu: a soak run over more seeds)
@pytest.mark.parametrize("seed", range(int(os.environ.get("KNH_TEST_SEEDS", "40"))))
def test_random_chain_matches_oracle(knh, oracle, seed):
    w, rng, changes, triggers = random_chain(seed)
    n = w.n_voices
    v = np.arange(n, dtype=np.uint32)
    g, o = make_gpu(knh, w, L.MIX_LEFT_FOLD), make_oracle(oracle, w)
    script = {}
    for block in range(6):
        ev = []
        if block == 0:
            for (s, restart, _rel) in triggers:
                ev.append((v, s, restart, L.VALUE_TRIGGER, None, None))
        if block == 3:
            for (s, _restart, rel) in triggers:
                if rel is not None:
                    ev.append((v[::2], s, rel, L.VALUE_TRIGGER, None, None))
        if block in (1, 2, 4) and changes:
            s, pi, lo, hi, precise = changes[int(rng.integers(0, len(changes)))]
            sel = v[rng.random(n) < 0.5]
            if len(sel):
                delays = (rng.integers(0, w.block_size, len(sel)).astype(np.uint16)) if precise else None
                ev.append((sel, s, pi, L.VALUE_FLOAT, rng.uniform(lo, hi, len(sel)), delays))
        script[block] = ev
    for block in range(6):
        for (sel, s, pi, kind, f, d) in script[block]:
            for bank in (g, o):
                bank.param_apply_many(sel, s, pi, kind, f, None, d)
        g_out, g_voices, g_flags = g.process_block_voices()
        o_out, o_voices, o_flags, o_done = o.process_block()
        kinds = [s.kind for s in w.stages]
        assert_bit_equal(g_voices, o_voices, f"seed {seed} chain {kinds} block {block} per-voice")
        for c in range(2):
            assert_bit_equal(g_out[c], o_out[c], f"seed {seed} block {block} left-fold mix")
        np.testing.assert_array_equal(g.read_done_frames(), o_done)
    g.close()
    o.close()
