"""A bank node with input channels (UGen::Inputs > 0, ugen.rs:232-284; what set_ar_param_buffer hands over, :309-329): the
host graph's signal on an input is a source every voice reads (KNH_STAGE_INPUT).  Against the oracle, whose voice graphs get
the same block as graph inputs (knaster_graph/src/task.rs:17-32)."""
import numpy as np
import pytest

from helpers import assert_bit_equal, fire_all, make_gpu, make_oracle
from knaster_amd import _lib as L
from knaster_amd import configs
from knaster_amd.bank import Stage

pytestmark = pytest.mark.gpu


def input_blocks(rng, n_blocks, channels, bs, dtype):
    t = np.arange(n_blocks * bs).reshape(n_blocks, 1, bs) / 48000.0
    base = np.concatenate([np.sin(2 * np.pi * (110.0 * (c + 1)) * t) for c in range(channels)], axis=1)
    return (0.5 * base + 0.05 * rng.standard_normal(base.shape)).astype(dtype)


@pytest.mark.parametrize("form", ["single", "pipe"])
@pytest.mark.parametrize("sample_type", [L.F32, L.F64])
def test_every_voice_filters_the_node_input(knh, oracle, monkeypatch, form, sample_type):
    """input 0 -> per-voice SvfFilter -> * EnvAsr, plus input 1 added dry behind a gain: a filter bank on an external signal."""
    monkeypatch.setenv("KNH_PIPELINE", "0" if form == "single" else "1")
    n, bs = 150, 64
    p = configs.voice_parameters(n)
    st = [Stage(L.STAGE_INPUT), Stage(L.STAGE_SVF), Stage(L.STAGE_MUL_ENV_ASR), Stage(L.STAGE_MUL_CONST)]
    w = configs.Workload("in", st, n, bs, sample_type, 2, in_channels=2)
    w.ctor = {0: np.zeros((n, 1)), 1: np.stack([np.full(n, 2.0), p["cutoff"], p["q"], np.zeros(n)], axis=1),
              2: np.tile([0.002, 0.01], (n, 1)), 3: np.full((n, 1), 1.0 / n)}
    g = make_gpu(knh, w, L.MIX_LEFT_FOLD)
    o = make_oracle(oracle, w)
    rng = np.random.default_rng(3)
    dtype = np.float64 if sample_type == L.F64 else np.float32
    ins = input_blocks(rng, 4, 2, bs, dtype)
    for b in range(4):
        for bank in (g, o):
            if b == 0:
                fire_all(bank, n, 2, 3)
            if b == 2:
                fire_all(bank, n, 2, 2)
            bank.set_input(ins[b])
        g_out, g_voices, _ = g.process_block_voices()
        o_out, o_voices, _, o_done = o.process_block()
        assert_bit_equal(g_voices, o_voices, f"block {b} per-voice")
        assert_bit_equal(g_out, o_out, f"block {b} mix")
        np.testing.assert_array_equal(g.read_done_frames(), o_done)
    assert np.abs(o_voices).max() > 1e-4
    # several blocks in one launch read their own block of the input
    g2 = make_gpu(knh, w, L.MIX_TREE)
    g3 = make_gpu(knh, w, L.MIX_TREE)
    for bank in (g2, g3):
        fire_all(bank, n, 2, 3)
    g2.set_input(ins)
    many, _ = g2.process_blocks(4)
    for b in range(4):
        g3.set_input(ins[b])
        one, _ = g3.process_block()
        assert_bit_equal(many[b], one, f"launch block {b}")
    for bank in (g, o, g2, g3):
        bank.close()


def assert_same_class_or_bits(a, b, what, strict_zero=False):
    """bit for bit where the reference's sample is a number; NaN where it is NaN (payloads are the FPU's, not the algorithm's)"""
    nan = np.isnan(b)
    np.testing.assert_array_equal(np.isnan(a), nan, err_msg=f"{what}: NaN positions")
    assert_bit_equal(np.where(nan, 0, a).astype(a.dtype), np.where(nan, 0, b).astype(b.dtype), what, strict_zero=strict_zero)


@pytest.mark.parametrize("poison", ["overflow", "inf", "nan"])
@pytest.mark.parametrize("form", ["single", "pipe"])
@pytest.mark.parametrize("sample_type", [L.F32, L.F64])
def test_low_pass_tiles_on_zeros_denormals_overflow_and_nan(knh, oracle, monkeypatch, form, sample_type, poison):
    """The low-pass filter's tiles leave out the output mix's multiplies by m0 = 0, m1 = 0, m2 = 1 (svf.rs:148-157, :262-279;
    Svf::tick_tile_low).  That is only the same value if the signs of zero, the subnormals, an overflow inside the filter and a
    NaN / infinity on the input come out as the reference's: wavefronts of low-pass voices only (the shortened step) beside
    wavefronts with every filter type (the general step), all against the oracle -- the SIGN of every zero included (round 4:
    `(0*x + 0*v1) + v2` gives a zero of the other sign than `fma(0, v1, v2)` only for v2 = -0, which needs an ic2 that was SET
    to -0: the shortened step's entry test; nothing here relies on -0 == +0)."""
    monkeypatch.setenv("KNH_PIPELINE", "0" if form == "single" else "1")
    n, bs, nb = 192, 64, 6
    p = configs.voice_parameters(n)
    ty = np.full(n, float(L.SVF_LOW))
    ty[64:128] = np.arange(64) % 9  # the nine types side by side in the middle wavefront
    st = [Stage(L.STAGE_INPUT), Stage(L.STAGE_SVF), Stage(L.STAGE_MUL_CONST)]
    w = configs.Workload("lowedge", st, n, bs, sample_type, 1, in_channels=1)
    w.ctor = {0: np.zeros((n, 1)), 1: np.stack([ty, p["cutoff"], p["q"], np.full(n, 3.0)], axis=1), 2: np.full((n, 1), 0.5)}
    g = make_gpu(knh, w, L.MIX_LEFT_FOLD)
    o = make_oracle(oracle, w)
    dtype = np.float64 if sample_type == L.F64 else np.float32
    tiny = np.finfo(dtype).smallest_subnormal
    big = np.finfo(dtype).max
    rng = np.random.default_rng(11)
    ins = (0.5 * rng.standard_normal((nb, 1, bs))).astype(dtype)
    ins[1, 0, :] = 0.0
    ins[1, 0, 5:9] = -0.0
    ins[1, 0, 20:40] = (rng.integers(-3, 4, 20) * tiny).astype(dtype)          # subnormal steps around zero
    ins[2, 0, :] = (rng.standard_normal(bs) * 1e-3).astype(dtype) * tiny * 1e3   # a signal that lives among the subnormals
    ins[3, 0, :] = 0.0                                                          # the state decays through them
    if poison == "overflow":
        ins[4, 0, 17] = big
        ins[4, 0, 18] = -big                                                    # overflows inside the filter
    else:
        ins[4, 0, 21] = np.inf if poison == "inf" else np.nan                   # (block 5: the filter never recovers)
    neg_zero = pos_zero = False
    with np.errstate(all="ignore"):
        for b in range(nb):
            g.set_input(ins[b])
            o.set_input(ins[b])
            g_out, g_voices, _ = g.process_block_voices()
            o_out, o_voices, _, _ = o.process_block()
            assert_same_class_or_bits(g_voices, o_voices, f"block {b} per-voice", strict_zero=True)
            if b < 4:
                assert_bit_equal(g_out, o_out, f"block {b} mix", strict_zero=True)
            neg_zero = neg_zero or bool(np.any((o_voices == 0) & np.signbit(o_voices)))
            pos_zero = pos_zero or bool(np.any((o_voices == 0) & ~np.signbit(o_voices)))
    if sample_type == L.F32:  # (f64 subnormals are out of these signals' reach: the f64 runs compare fewer zeros)
        assert neg_zero and pos_zero, "the case is meant to produce zeros of both signs"
    assert np.isnan(o_voices).any()
    g.close()
    o.close()


def test_ring_modulating_the_node_input_and_driving_a_frequency_with_it(knh, oracle):
    """Graph-shaped voices on an input: input 0 times a per-voice oscillator; and (input 1 * depth + f0) as the audio-rate
    frequency of a SinWt (.ar_params() + link, audio_rate.rs:42-57) -- the set_ar_param_buffer case."""
    n, bs = 70, 128
    p = configs.voice_parameters(n)
    st = [Stage(L.STAGE_INPUT),                                       # 1: input 0
          Stage(L.STAGE_SIN_WT),                                      # 2
          Stage(L.STAGE_MATH_MUL, input=1, input2=2),                 # 3: ring modulation
          Stage(L.STAGE_INPUT),                                       # 4: input 1
          Stage(L.STAGE_MUL_CONST), Stage(L.STAGE_ADD_CONST),         # 5, 6: * depth + f0
          Stage(L.STAGE_SIN_WT, flags=L.STAGE_FLAG_AR_FREQ),          # 7: FM carrier driven by it
          Stage(L.STAGE_MATH_ADD, input=3, input2=7),                 # 8
          Stage(L.STAGE_MUL_CONST)]                                   # 9
    w = configs.Workload("inring", st, n, bs, L.F32, 1, in_channels=2)
    w.ctor = {0: np.zeros((n, 1)), 1: p["freq"].reshape(n, 1), 3: np.ones((n, 1)), 4: p["fm_index"].reshape(n, 1), 5: p["freq"].reshape(n, 1),
              6: p["freq"].reshape(n, 1), 8: np.full((n, 1), 0.5 / n)}
    g = make_gpu(knh, w, L.MIX_LEFT_FOLD)
    o = make_oracle(oracle, w)
    ins = input_blocks(np.random.default_rng(5), 3, 2, bs, np.float32)
    for b in range(3):
        g.set_input(ins[b])
        o.set_input(ins[b])
        g_out, g_voices, _ = g.process_block_voices()
        o_out, o_voices, _, _ = o.process_block()
        assert_bit_equal(g_voices, o_voices, f"block {b} per-voice")
        assert_bit_equal(g_out, o_out, f"block {b} mix")
    assert np.abs(o_voices).max() > 1e-4
    g.close()
    o.close()


def test_input_rules(knh):
    st = [Stage(L.STAGE_INPUT), Stage(L.STAGE_MUL_CONST)]
    with pytest.raises(L.KnasterHipError):  # more than 16 input channels
        knh.VoiceBank(st, 4, in_channels=17)
    b = knh.VoiceBank(st, 4, in_channels=1)
    assert b.inputs() == 1
    b.set_ctor_args(0, np.full((4, 1), 3.0))  # channel 3 of 1
    b.set_ctor_args(1, np.ones((4, 1)))
    with pytest.raises(L.KnasterHipError):
        b.init(48000, 32)
    b.close()
    b = knh.VoiceBank(st, 4, in_channels=1)
    b.set_ctor_args(0, np.zeros((4, 1)))
    b.set_ctor_args(1, np.full((4, 1), 0.25))
    b.init(48000, 32)
    with pytest.raises(L.KnasterHipError):  # no input block given for this call
        b.process_block()
    x = np.linspace(-1, 1, 32, dtype=np.float32).reshape(1, 32)
    b.set_input(x)
    out, _ = b.process_block()
    assert np.array_equal(out[0], ((x[0] * np.float32(0.25)) * 4).astype(np.float32)) or np.allclose(out[0], x[0])
    with pytest.raises(L.KnasterHipError):  # one set_input per process call
        b.process_block()
    b.set_input(np.stack([x, x]))
    with pytest.raises(L.KnasterHipError):  # two blocks set, one processed
        b.process_block()
    b.close()


@pytest.mark.parametrize("kind", ["plain", "rank"])
def test_inputs_through_asynchronous_launches(knh, kind):
    """Two launches in flight (knh_bank_process_blocks_begin / _end), each with its own input blocks handed over right behind
    the other: the staging buffer of a launch's input must not be rewritten while that launch's upload is still queued on
    the stream it was given (an event, not the bank's own stream, says when).  Same samples as blocking calls; also for a
    rank bank (whose launches run on a stream of its own)."""
    n, bs, k = 130, 64, 16
    p = configs.voice_parameters(n)
    st = [Stage(L.STAGE_INPUT), Stage(L.STAGE_SVF), Stage(L.STAGE_MUL_CONST)]
    w = configs.Workload("in_async", st, n, bs, L.F32, 2, in_channels=1)
    w.ctor = {0: np.zeros((n, 1)), 1: np.stack([np.full(n, 2.0), p["cutoff"], p["q"], np.zeros(n)], axis=1), 2: np.full((n, 1), 1.0 / n)}
    kw = dict(rank=0, world=1) if kind == "rank" else {}
    a = make_gpu(knh, w, L.MIX_TREE, **kw)
    b = make_gpu(knh, w, L.MIX_TREE, **kw)
    rng = np.random.default_rng(11)
    ins = [input_blocks(rng, k, 1, bs, np.float32) * (1.0 + i) for i in range(6)]
    want = []
    for x in ins:  # blocking calls
        b.set_input(x)
        want.append(b.process_blocks(k)[0])
    got = []
    a.set_input(ins[0])
    a.process_blocks_begin(k)
    for i in range(len(ins)):
        if i + 1 < len(ins):
            a.set_input(ins[i + 1])  # while launch i may still be waiting for its upload
            a.process_blocks_begin(k)
        got.append(a.process_blocks_end())
    for i in range(len(ins)):
        assert_bit_equal(got[i], want[i], f"launch {i}")
    assert np.abs(want[-1]).max() > 1e-5
    a.close()
    b.close()
