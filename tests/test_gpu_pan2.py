"""Pan2 (pan.rs:12-37) on the device: a chain that ends in KNH_STAGE_PAN2 has a left and a right signal per
voice and one left fold per output channel.  Per-voice signals and the KNH_MIX_LEFT_FOLD mix are bit-identical to the
oracle in every kernel form; the tree mix is within the north-star tolerance."""
import numpy as np
import pytest

from helpers import assert_bit_equal, fire_all, make_gpu, make_oracle
from knaster_amd import _lib as L
from knaster_amd import configs
from knaster_amd.bank import Stage

pytestmark = pytest.mark.gpu

FORMS = {  # environment switches read by knh_bank_create / knh_bank_init
    "pipe64": {"KNH_PIPELINE": "1", "KNH_PIPE_BIG": "1"},
    "pipe64inplace": {"KNH_PIPELINE": "1", "KNH_PIPE_BIG": "2"},
    "pipe32": {"KNH_PIPELINE": "1", "KNH_PIPE_BIG": "0"},
    "single": {"KNH_PIPELINE": "0"},
    "wide4": {"KNH_PIPELINE": "0", "KNH_WIDE": "4"},
    "jit": {"KNH_JIT": "1", "KNH_JIT_PIPE": "0"},
    "jitpipe": {"KNH_JIT": "1", "KNH_JIT_PIPE": "1"},
}


def pan_events(w, n_blocks):
    nv = w.n_voices
    v = np.arange(nv, dtype=np.uint32)
    env_stage = w.restart[0]
    pan_stage = len(w.stages) - 1

    def ev(block, bank):
        if block == 0:
            fire_all(bank, nv, *w.restart)
        if block == 1:  # new pans for every third voice, new frequencies for some
            bank.param_apply_many(v[::3], pan_stage, 0, L.VALUE_FLOAT, np.linspace(1.0, -1.0, len(v[::3])))
            bank.param_apply_many(v[1::4], 0, 0, L.VALUE_FLOAT, 220.0 + 3.0 * v[1::4])
        if block == 2 and w.release:
            bank.param_apply_many(v, w.release[0], w.release[1], L.VALUE_TRIGGER)
        if block == 3:
            bank.param_apply_many(v, env_stage, w.restart[1], L.VALUE_TRIGGER)
            bank.param_apply_many(v[::2], pan_stage, 0, L.VALUE_FLOAT, -0.25)
    return ev


@pytest.mark.parametrize("form", sorted(FORMS))
@pytest.mark.parametrize("name,n_voices,block_size,sample_type", [
    ("M1", 600, 64, L.F32), ("M1", 70, 48, L.F32), ("P3", 256, 128, L.F32), ("P3", 130, 96, L.F64), ("M1", 64, 33, L.F64)])
def test_pan2_chain_bit_exact_in_every_kernel_form(knh, oracle, monkeypatch, form, name, n_voices, block_size, sample_type):
    for k, val in FORMS[form].items():
        monkeypatch.setenv(k, val)
    w = configs.config(name, n_voices=n_voices, block_size=block_size, sample_type=sample_type)
    g = make_gpu(knh, w, L.MIX_LEFT_FOLD)
    t = make_gpu(knh, w, L.MIX_TREE)
    o = make_oracle(oracle, w)
    ev = pan_events(w, 5)
    for b in range(5):
        for bank in (g, t, o):
            ev(b, bank)
        g_out, g_voices, g_flags = g.process_block_voices()
        t_out, _ = t.process_block()
        o_out, o_voices, o_flags, o_done = o.process_block()
        assert g_voices.shape == (2, n_voices, block_size)
        assert_bit_equal(g_voices, o_voices, f"{name} {form} block {b}: per-voice left/right")
        assert_bit_equal(g_out, o_out, f"{name} {form} block {b}: left-fold mix per channel")
        ref = o_voices.astype(np.float64).sum(axis=1)
        tol = 1e-5 * max(1.0, float(np.abs(o_voices).max(axis=2).sum(axis=1).max()))
        assert np.max(np.abs(t_out.astype(np.float64) - ref)) <= tol
        np.testing.assert_array_equal(g.read_done_frames(), o_done)
        assert (g_flags & L.FLAG_ANY_DONE) == (o_flags & L.FLAG_ANY_DONE)
        if b == 0:
            assert np.abs(g_out[0] - g_out[1]).max() > 0  # not the mono mix twice
    for bank in (g, t, o):
        bank.close()


@pytest.mark.parametrize("form", ["pipe64", "pipe32", "single"])
def test_pan2_many_blocks_per_launch_equal_block_by_block(knh, monkeypatch, form):
    for k, val in FORMS[form].items():
        monkeypatch.setenv(k, val)
    w = configs.config("M1", n_voices=300, block_size=64)
    a, b = make_gpu(knh, w), make_gpu(knh, w)
    v = np.arange(w.n_voices, dtype=np.uint32)
    pans = np.linspace(-1, 1, len(v[::2]))
    for bank in (a, b):
        fire_all(bank, w.n_voices, *w.restart)
    # block 3 of the launch: new pans; block 5: restart
    a.param_apply_many(v[::2], 3, 0, L.VALUE_FLOAT, pans, block_offset=3)
    a.param_apply_many(v, 2, 2, L.VALUE_TRIGGER, block_offset=5)
    many, _ = a.process_blocks(8)
    for k in range(8):
        if k == 3:
            b.param_apply_many(v[::2], 3, 0, L.VALUE_FLOAT, pans)
        if k == 5:
            b.param_apply_many(v, 2, 2, L.VALUE_TRIGGER)
        one, _ = b.process_block()
        assert_bit_equal(many[k], one, f"block {k}")
    a.close()
    b.close()


def test_pan2_host_sharded_bank(knh):
    w = configs.config("M1", n_voices=600, block_size=64)
    a, b = make_gpu(knh, w), make_gpu(knh, w, host_threads=3)
    for bank in (a, b):
        fire_all(bank, w.n_voices, *w.restart)
    for _ in range(3):
        x, _ = a.process_block()
        y, _ = b.process_block()
        assert np.max(np.abs(x.astype(np.float64) - y)) <= 1e-5
    a.close()
    b.close()


def test_pan2_descriptor_rules(knh):
    pan_first = [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_PAN2), Stage(L.STAGE_MUL_CONST)]
    with pytest.raises(L.KnasterHipError):
        knh.VoiceBank(pan_first, 4, L.F32, 2)
    with pytest.raises(L.KnasterHipError):  # two channels come out of a Pan2
        knh.VoiceBank([Stage(L.STAGE_SIN_WT), Stage(L.STAGE_PAN2)], 4, L.F32, 1)
    with pytest.raises(L.KnasterHipError):
        knh.VoiceBank([Stage(L.STAGE_SIN_WT), Stage(L.STAGE_PAN2, delayed_changes_per_block=2)], 4, L.F32, 2)
    b = knh.VoiceBank([Stage(L.STAGE_SIN_WT), Stage(L.STAGE_PAN2)], 4, L.F32, 2)
    assert b.outputs() == 2
    b.set_ctor_args(0, np.full((4, 1), 440.0))
    b.set_ctor_args(1, np.array([[-1.0], [0.0], [0.5], [1.0]]))
    b.init(48000, 16)
    assert b.stage_param_descriptions(1) == ["pan"]
    _, voices, _ = b.process_block_voices()
    # hard left: nothing on the right to within the approximation's error, and the other way round
    assert np.abs(voices[1, 0]).max() < 2e-4 * np.abs(voices[0, 0]).max() + 1e-7
    assert np.abs(voices[0, 3]).max() < 2e-4 * np.abs(voices[1, 3]).max() + 1e-7
    b.close()
