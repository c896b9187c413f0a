"""The delay rings moved as whole lines (RingLines, voice_stages.hpp): the whole-chain kernels -- one wavefront per
workgroup, four, eight -- hand a tile's ring traffic to groups of eight lanes per voice through an LDS tile, read the next
tile's lines ahead, and let either pointer cross the end of its ring inside a tile.  Per voice bit for bit against the
oracle (delay.rs:14-50, 93-306), in f32 (64-sample visits, two lines per voice) and f64 (32-sample visits, two lines)."""
import numpy as np
import pytest

from helpers import assert_bit_equal, make_gpu, make_oracle
from knaster_amd import _lib as L
from knaster_amd import configs
from knaster_amd.bank import Stage

pytestmark = pytest.mark.gpu

FORMS = {"one": {"KNH_PIPELINE": "0", "KNH_WIDE": "0"}, "four": {"KNH_PIPELINE": "0", "KNH_WIDE": "4"},
         "eight": {"KNH_PIPELINE": "0", "KNH_WIDE": "8"}, "pipeline": {}}


def set_form(monkeypatch, form):
    for k in ("KNH_PIPELINE", "KNH_WIDE"):
        monkeypatch.delenv(k, raising=False)
    for k, v in FORMS[form].items():
        monkeypatch.setenv(k, v)


def compare(knh, oracle, w, blocks, events):
    g = make_gpu(knh, w, L.MIX_LEFT_FOLD)
    o = make_oracle(oracle, w)
    for b in range(blocks):
        events(b, g)
        events(b, o)
        g_out, g_voices, _ = g.process_block_voices()
        o_out, o_voices, _, _ = o.process_block()
        assert_bit_equal(g_voices, o_voices, f"{w.name} block {b} per voice")
        assert_bit_equal(g_out[0], o_out[0], f"{w.name} block {b} left-fold mix")
    g.close()
    o.close()


def ring_seconds(n):
    """Ring lengths per 64-voice group: a multiple of the visit length, one that is not (the write pointer crosses the end
    inside a tile), an odd one (a 16-byte chunk straddles the end), the first again."""
    per_group = [0.02, 0.0205, 0.020125 + 1e-7, 0.02]  # 960, 984, 966 samples at 48 kHz
    return np.array([per_group[(v // 64) % len(per_group)] for v in range(n)])


@pytest.mark.parametrize("sample_type", [L.F32, L.F64])
@pytest.mark.parametrize("form", list(FORMS))
def test_sample_delay_tiles_as_whole_lines(knh, oracle, monkeypatch, form, sample_type):
    """The D3 chain.  Delays of two tiles and more (lines read ahead), between one and two tiles (no read-ahead), shorter
    than a tile in one voice group from block 5 on (that wavefront goes sample by sample, the others do not), a
    sample-accurate change of every delay inside block 3, rings that wrap two to three times."""
    set_form(monkeypatch, form)
    n, bs = 250, 256
    w = configs.config("D3", n_voices=n, block_size=bs, sample_type=sample_type)
    rs = ring_seconds(n)
    w.ctor[3] = rs.reshape(n, 1)
    v = np.arange(n, dtype=np.uint32)
    ring = np.floor(rs * 48000.0 + 1e-9).astype(np.int64)
    long_delays = (130 + (v * 37) % 700).astype(np.float64)          # >= two 64-sample tiles
    mid_delays = (66 + (v * 5) % 60).astype(np.float64)               # one to two tiles

    def ev(block, bank):
        if block == 0:
            bank.param_apply_many(v, 4, 3, L.VALUE_TRIGGER)
            bank.param_apply_many(v, 3, 0, L.VALUE_FLOAT, (long_delays + 0.25) / 48000.0)
        if block == 2:  # voices 64..127: between one and two tiles
            sel = v[64:128]
            bank.param_apply_many(sel, 3, 0, L.VALUE_FLOAT, (mid_delays[64:128] + 0.25) / 48000.0)
        if block == 3:  # sample-accurate, every voice at a frame of its own
            bank.param_apply_many(v, 3, 0, L.VALUE_FLOAT, (np.minimum(long_delays + 64.0, ring - 70.0) + 0.25) / 48000.0,
                                  delays=(v * 3 % bs).astype(np.uint16))
        if block == 5:  # voices 128..191: shorter than a tile (and 0, 1, the whole ring among them)
            sel = v[128:192]
            d = np.array([[0, 1, 5, 17, 31, 40, 63][i % 7] for i in range(64)], dtype=np.float64)
            d[10] = ring[138]
            bank.param_apply_many(sel, 3, 0, L.VALUE_FLOAT, (d + 0.25) / 48000.0)
        if block == 6:
            bank.param_apply_many(v, 4, 2, L.VALUE_TRIGGER)
        if block == 8:  # ... and long again
            sel = v[128:192]
            bank.param_apply_many(sel, 3, 0, L.VALUE_FLOAT, (long_delays[128:192] + 0.25) / 48000.0)
    compare(knh, oracle, w, 11, ev)


@pytest.mark.parametrize("kind", [L.STAGE_ALLPASS_DELAY, L.STAGE_ALLPASS_FB_DELAY])
@pytest.mark.parametrize("sample_type", [L.F32, L.F64])
@pytest.mark.parametrize("form", ["one", "four"])
def test_allpass_delay_tiles_as_whole_lines(knh, oracle, monkeypatch, form, sample_type, kind):
    """AllpassDelay / AllpassFeedbackDelay with delays of a tile and more (whole lines), both pointers crossing the end of
    rings of three lengths, a change mid-block, one voice group with delays shorter than a tile."""
    set_form(monkeypatch, form)
    monkeypatch.setenv("KNH_JIT_WAVES", "1" if form == "one" else "4")
    n, bs = 200, 128
    p = configs.voice_parameters(n)
    v = np.arange(n, dtype=np.uint32)
    w = configs.Workload("allpass_lines", [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_WR_MUL), Stage(kind, delayed_changes_per_block=2),
                                           Stage(L.STAGE_MUL_CONST)], n, bs, sample_type, 2)
    rs = ring_seconds(n)
    w.ctor = {0: p["freq"].reshape(n, 1), 1: np.full((n, 1), 0.5), 2: rs.reshape(n, 1), 3: np.full((n, 1), 1.0 / n)}
    frames = 70.0 + (v * 13 % 800) + 0.37

    def ev(block, bank):
        if kind == L.STAGE_ALLPASS_FB_DELAY and block in (1, 7):
            bank.param_apply_many(v, 2, 1, L.VALUE_FLOAT, (0.3 + 0.002 * v) * (1.0 if block == 1 else -0.9))
        if block == 2:
            bank.param_apply_many(v, 2, 0, L.VALUE_FLOAT, frames / 48000.0)
        if block == 5:
            bank.param_apply_many(v, 2, 0, L.VALUE_FLOAT, (frames * 0.5 + 40.0) / 48000.0, delays=(v % bs).astype(np.uint16))
        if block == 8:
            sel = v[64:128]
            bank.param_apply_many(sel, 2, 0, L.VALUE_FLOAT, (3.3 + (sel % 50)) / 48000.0)
    compare(knh, oracle, w, 22, ev)


def test_wide_delay_bank_equals_the_pipeline(knh, monkeypatch):
    """A bank big enough for the eight-wavefront form by itself (2 100 voice groups would be; here the form is forced at
    600 voices), several blocks per launch: the same bits as the pipeline form's launch."""
    n, bs, blocks = 600, 512, 4
    w = configs.config("D3", n_voices=n, block_size=bs)
    w.ctor[3] = np.full((n, 1), 0.03)
    v = np.arange(n, dtype=np.uint32)
    outs = []
    for form in ("pipeline", "eight", "four"):
        set_form(monkeypatch, form)
        b = make_gpu(knh, w)
        b.param_apply_many(v, 4, 3, L.VALUE_TRIGGER)
        b.param_apply_many(v, 3, 0, L.VALUE_FLOAT, w.delay_times * 0.1 + 0.003)
        out, _ = b.process_blocks(blocks)
        out2, _ = b.process_blocks(blocks)
        outs.append(np.concatenate([np.asarray(out), np.asarray(out2)]))
        b.close()
    assert np.abs(outs[0]).max() > 1e-3
    assert_bit_equal(outs[1], outs[0], "eight wavefronts per workgroup against the pipeline")
    assert_bit_equal(outs[2], outs[0], "four wavefronts per workgroup against the pipeline")


def test_a_fused_delay_chain_beyond_one_pipeline_round(knh, monkeypatch):
    """A chain with a delay and no pre-built kernel, 266 voice groups: fused at init as four whole-chain wavefronts per
    workgroup (one round of the pipeline covers 256 groups; rings want more wavefronts in flight, not a second round) -- the same
    bits as the one-wavefront form."""
    n, bs, blocks = 17000, 128, 3
    p = configs.voice_parameters(n)
    v = np.arange(n, dtype=np.uint32)
    w = configs.Workload("allpass_big", [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_WR_MUL), Stage(L.STAGE_ALLPASS_FB_DELAY), Stage(L.STAGE_MUL_CONST)],
                         n, bs, L.F32, 2)
    w.ctor = {0: p["freq"].reshape(n, 1), 1: np.full((n, 1), 0.5), 2: np.full((n, 1), 0.0101), 3: np.full((n, 1), 1.0 / n)}
    outs = []
    for env in ({}, {"KNH_PIPELINE": "0", "KNH_JIT_WAVES": "1"}):
        for k in ("KNH_PIPELINE", "KNH_WIDE", "KNH_JIT_WAVES"):
            monkeypatch.delenv(k, raising=False)
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        b = make_gpu(knh, w)
        b.param_apply_many(v, 2, 1, L.VALUE_FLOAT, 0.3 + 0.00001 * v)
        b.param_apply_many(v, 2, 0, L.VALUE_FLOAT, (70.0 + (v * 13 % 380) + 0.37) / 48000.0)
        out, _ = b.process_blocks(blocks)
        out2, _ = b.process_blocks(blocks)
        outs.append(np.concatenate([np.asarray(out), np.asarray(out2)]))
        b.close()
    assert np.abs(outs[0]).max() > 1e-4
    assert_bit_equal(outs[0], outs[1], "four whole-chain wavefronts per workgroup against one")
