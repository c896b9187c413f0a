"""WrArParams in general (knaster_core_dsp/src/wrappers_core/audio_rate.rs:11-85, `link`
graph_edit.rs:735-754): any float parameter of a node pushed as `.ar_params()` driven, sample by sample, by another signal
of the voice (knh_stage_desc.ar_param / .input2).  Against the oracle, whose voices hold the reference's own wrapper and
parameter edge.  Bit for bit where the parameter's setter is + - x / only; within a stated tolerance where it calls
tan / pow / sqrt / exp (SvfFilter, one-pole filters) or where the node itself does (SinNumeric's sin)."""
import numpy as np
import pytest

from helpers import assert_bit_equal, fire_all, make_gpu, make_oracle
from knaster_amd import _lib as L
from knaster_amd import configs
from knaster_amd.bank import Stage

pytestmark = pytest.mark.gpu


def lfo(n, p, depth, offset, rate_scale=0.01):
    """three stages: SinWt(slow) * depth + offset -- the driving signal; returns (stages, ctor entries by relative index)"""
    st = [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_MUL_CONST), Stage(L.STAGE_ADD_CONST)]
    ctor = {0: (p["freq"] * rate_scale).reshape(n, 1), 1: np.asarray(depth, dtype=np.float64).reshape(n, 1),
            2: np.asarray(offset, dtype=np.float64).reshape(n, 1)}
    return st, ctor


def run(knh, oracle, w, blocks, events=None, tol=None, check_done=True):
    g = make_gpu(knh, w, L.MIX_LEFT_FOLD)
    o = make_oracle(oracle, w)
    peak = 0.0
    for b in range(blocks):
        if events:
            events(b, g)
            events(b, o)
        g_out, g_voices, _ = g.process_block_voices()
        o_out, o_voices, _, o_done = o.process_block()
        if tol is None:
            assert_bit_equal(g_voices, o_voices, f"{w.name} block {b} per-voice")
            assert_bit_equal(g_out, o_out, f"{w.name} block {b} left-fold mix")
        else:
            err = np.max(np.abs(g_voices.astype(np.float64) - o_voices.astype(np.float64)))
            assert err <= tol, f"{w.name} block {b}: {err} > {tol}"
        if check_done:
            np.testing.assert_array_equal(g.read_done_frames(), o_done)
        peak = max(peak, float(np.abs(o_voices).max()))
    assert peak > 1e-4
    g.close()
    o.close()


@pytest.mark.parametrize("sample_type", [L.F32, L.F64])
@pytest.mark.parametrize("case", ["sinwt_phase_offset", "sinwt_freq", "const_value", "wr_mul", "add_const_value"])
def test_bit_exact_audio_rate_parameters(knh, oracle, case, sample_type):
    n, bs = 100, 96
    p = configs.voice_parameters(n)
    if case == "sinwt_phase_offset":  # carrier.link("phase_offset", lfo): (v * 65536.0) as u32 every sample
        drv, c = lfo(n, p, 4000.0 + 100.0 * np.arange(n), np.full(n, 8192.0))
        st = drv + [Stage(L.STAGE_SIN_WT, ar_param=2, input2=3), Stage(L.STAGE_MUL_CONST)]
        c.update({3: p["freq"].reshape(n, 1), 4: np.full((n, 1), 1.0 / n)})
    elif case == "sinwt_freq":  # the general spelling of audio-rate FM (negative and huge frequencies saturate, osc.rs:129)
        drv, c = lfo(n, p, p["fm_index"] * 40.0, p["freq"], rate_scale=1.7)
        st = drv + [Stage(L.STAGE_SIN_WT, ar_param=1, input2=3), Stage(L.STAGE_MUL_CONST)]
        c.update({3: p["freq"].reshape(n, 1), 4: np.full((n, 1), 1.0 / n)})
    elif case == "const_value":  # sine * Constant with Constant.value linked to the lfo: a tremolo
        drv, c = lfo(n, p, np.full(n, 0.4), np.full(n, 0.5))
        st = drv + [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_MUL_CONST, ar_param=1, input2=3), Stage(L.STAGE_MUL_CONST)]
        c.update({3: p["freq"].reshape(n, 1), 4: np.full((n, 1), 0.3), 5: np.full((n, 1), 1.0 / n)})
    elif case == "add_const_value":
        drv, c = lfo(n, p, np.full(n, 0.25), np.full(n, -0.1))
        st = drv + [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_ADD_CONST, ar_param=1, input2=3), Stage(L.STAGE_MUL_CONST)]
        c.update({3: p["freq"].reshape(n, 1), 4: np.full((n, 1), 0.0), 5: np.full((n, 1), 1.0 / n)})
    else:  # SinWt(f).wr_mul(g).ar_params() with "wr_mul" linked
        drv, c = lfo(n, p, np.full(n, 0.4), np.full(n, 0.5))
        st = drv + [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_WR_MUL, ar_param=1, input2=3), Stage(L.STAGE_MUL_CONST)]
        c.update({3: p["freq"].reshape(n, 1), 4: np.full((n, 1), 0.3), 5: np.full((n, 1), 1.0 / n)})
    w = configs.Workload("ar_" + case, st, n, bs, sample_type, 1)
    w.ctor = c

    def ev(block, bank):  # an ordinary change of the linked parameter is ignored while the link stands (audio_rate.rs:70-74) ...
        if block == 2:
            ar_stage = next(i for i, s in enumerate(st) if s.ar_param)
            bank.param_apply(5, ar_stage, st[ar_stage].ar_param - 1, 0.123)
            bank.param_apply(7, 1, 0, 17.0)  # ... and the driver's own parameters change as ever
    run(knh, oracle, w, 5, ev)


@pytest.mark.parametrize("kind", [L.STAGE_MUL_ENV_ASR, L.STAGE_MUL_ENV_AR])
@pytest.mark.parametrize("param", [0, 1])
def test_envelope_times_at_audio_rate(knh, oracle, kind, param):
    """attack_time / release_time linked to a signal: the rate follows it sample by sample.  Under the wrapper the envelope
    runs through UGen::process, so mark_done carries frame 0 (envelopes.rs:153-156): the done frames say so."""
    n, bs = 80, 64
    p = configs.voice_parameters(n)
    drv, c = lfo(n, p, np.full(n, 0.0015), np.full(n, 0.002), rate_scale=0.5)  # 0.5 .. 3.5 ms
    st = drv + [Stage(L.STAGE_SIN_WT), Stage(kind, ar_param=param + 1, input2=3), Stage(L.STAGE_MUL_CONST)]
    c.update({3: p["freq"].reshape(n, 1), 4: np.tile([0.001, 0.002], (n, 1)), 5: np.full((n, 1), 1.0 / n)})
    w = configs.Workload("ar_env", st, n, bs, L.F32, 1)
    w.ctor = c
    restart = 3 if kind == L.STAGE_MUL_ENV_ASR else 2

    def ev(block, bank):
        if block in (0, 5):
            fire_all(bank, n, 4, restart)
        if block == 2 and kind == L.STAGE_MUL_ENV_ASR:
            fire_all(bank, n, 4, 2)
    run(knh, oracle, w, 8, ev)


@pytest.mark.parametrize("sample_type", [L.F32, L.F64])
@pytest.mark.parametrize("param", [0, 1, 2])
def test_svf_parameters_at_audio_rate_within_tolerance(knh, oracle, param, sample_type):
    """filter.link("cutoff_freq" | "q" | "gain", lfo): set_coeffs runs every sample with the device's tan / pow / sqrt.
    Tolerance: 2e-4 of a unit-range signal (coefficients a few ulp off, carried by the recurrence), stated here and in DESIGN.md."""
    n, bs = 70, 128
    p = configs.voice_parameters(n)
    if param == 0:
        drv, c = lfo(n, p, p["cutoff"] * 0.4, p["cutoff"], rate_scale=0.05)
    elif param == 1:
        drv, c = lfo(n, p, np.full(n, 0.3), p["q"], rate_scale=0.05)
    else:
        drv, c = lfo(n, p, np.full(n, 5.0), np.zeros(n), rate_scale=0.05)
    ty = L.SVF_LOW if param < 2 else L.SVF_BELL  # the gain matters for Bell and the shelves only
    st = drv + [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_SVF, ar_param=param + 1, input2=3), Stage(L.STAGE_MUL_CONST)]
    c.update({3: p["freq"].reshape(n, 1), 4: np.stack([np.full(n, float(ty)), p["cutoff"], p["q"], np.zeros(n)], axis=1), 5: np.full((n, 1), 0.5)})
    w = configs.Workload("ar_svf", st, n, bs, sample_type, 1)
    w.ctor = c

    def ev(block, bank):  # another parameter of the same filter set the ordinary way: the device-side setter must see it
        if block == 2:
            bank.param_apply_many(np.arange(n, dtype=np.uint32), 4, 1 if param != 1 else 0, L.VALUE_FLOAT,
                                  (p["q"] * 1.3) if param != 1 else (p["cutoff"] * 0.8))
    run(knh, oracle, w, 4, ev, tol=2e-4)


@pytest.mark.parametrize("kind", [L.STAGE_ONEPOLE_LPF, L.STAGE_ONEPOLE_HPF])
def test_onepole_cutoff_at_audio_rate_within_tolerance(knh, oracle, kind):
    n, bs = 70, 128
    p = configs.voice_parameters(n)
    drv, c = lfo(n, p, p["cutoff"] * 0.5, p["cutoff"], rate_scale=0.05)
    st = drv + [Stage(L.STAGE_SIN_WT), Stage(kind, ar_param=1, input2=3), Stage(L.STAGE_MUL_CONST)]
    c.update({3: p["freq"].reshape(n, 1), 5: np.full((n, 1), 0.5)})
    if kind == L.STAGE_ONEPOLE_LPF:
        c[4] = p["cutoff"].reshape(n, 1)
    w = configs.Workload("ar_onepole", st, n, bs, L.F32, 1)
    w.ctor = c
    run(knh, oracle, w, 4, tol=2e-5)  # device exp against glibc's expf


def test_sin_numeric_parameters_at_audio_rate(knh, oracle):
    """SinNumeric's freq and phase_offset: the setters are exact (a division, a copy); the node's sin is the device's, as for
    every SinNumeric voice (4e-5, tests/test_gpu_parity.py)."""
    n, bs = 64, 64
    p = configs.voice_parameters(n)
    for param in (0, 1):
        drv, c = (lfo(n, p, p["fm_index"], p["freq"], rate_scale=0.3) if param == 0 else lfo(n, p, np.full(n, 0.2), np.full(n, 0.25)))
        st = drv + [Stage(L.STAGE_SIN_NUMERIC, ar_param=param + 1, input2=3), Stage(L.STAGE_MUL_CONST)]
        c.update({3: p["freq"].reshape(n, 1), 4: np.full((n, 1), 0.5)})
        w = configs.Workload("ar_sinnum", st, n, bs, L.F32, 1)
        w.ctor = c
        run(knh, oracle, w, 3, tol=4e-5)


def test_audio_rate_parameter_rules(knh):
    sw, mc = Stage(L.STAGE_SIN_WT), Stage(L.STAGE_MUL_CONST)
    with pytest.raises(L.KnasterHipError):  # no driver named
        knh.VoiceBank([sw, mc, Stage(L.STAGE_SIN_WT, ar_param=2)], 4, out_channels=1)
    with pytest.raises(L.KnasterHipError):  # reset_phase is a trigger, not a float parameter
        knh.VoiceBank([sw, mc, Stage(L.STAGE_SIN_WT, ar_param=3, input2=2)], 4, out_channels=1)
    with pytest.raises(L.KnasterHipError):  # not a parameter the device-side setters cover (Phasor.freq)
        knh.VoiceBank([sw, mc, Stage(L.STAGE_PHASOR, ar_param=1, input2=2)], 4, out_channels=1)
    with pytest.raises(L.KnasterHipError):  # together with the older spelling
        knh.VoiceBank([sw, mc, Stage(L.STAGE_SIN_WT, flags=L.STAGE_FLAG_AR_FREQ, ar_param=2, input2=2)], 4, out_channels=1)
    b = knh.VoiceBank([sw, mc, Stage(L.STAGE_SIN_WT, ar_param=2, input2=2)], 4, out_channels=1)
    b.close()
