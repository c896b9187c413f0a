"""GPU parity: the HIP voice bank, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Bar: every voice's own signal bit-exact (integer phase, + - * only); SinNumeric
within 1e-5 (device sin); the mixed block within 1e-5 of the f64-accumulated mix, and bit-exact
in KNH_MIX_LEFT_FOLD mode."""
import numpy as np
import pytest

from helpers import assert_bit_equal, f64_mix, fire_all, make_gpu, make_oracle
from knaster_amd import _lib as L
from knaster_amd import configs
from knaster_amd.bank import Stage

pytestmark = pytest.mark.gpu

MIX_TOL = 1e-5  # north_star: "f32 output within 1e-5 of CPU reference"


def run_pair(knh, oracle, w, blocks, events=None, mix_mode=L.MIX_TREE, allow_fma=False, voice_tol=None):
    """events(block, bank) is called on both banks before each block."""
    g = make_gpu(knh, w, mix_mode, allow_fma)
    o = make_oracle(oracle, w)
    for b in range(blocks):
        if events:
            events(b, g)
            events(b, o)
        g_out, g_voices, g_flags = g.process_block_voices()
        o_out, o_voices, o_flags, o_done = o.process_block()
        if voice_tol is None:
            assert_bit_equal(g_voices, o_voices, f"{w.name} block {b} per-voice")
        else:
            assert np.max(np.abs(g_voices.astype(np.float64) - o_voices)) <= voice_tol, f"{w.name} block {b}"
        ref = f64_mix(o_voices)
        # 1e-5 is stated for a normalised mix (sum of |gain| <= 1); scale it for unnormalised test chains
        tol = MIX_TOL * max(1.0, float(np.abs(o_voices).max(axis=1).sum()))
        for c in range(w.out_channels):
            assert np.max(np.abs(g_out[c].astype(np.float64) - ref)) <= tol, f"{w.name} block {b} mix ch{c}"
            if mix_mode == L.MIX_LEFT_FOLD and voice_tol is None:
                assert_bit_equal(g_out[c], o_out[c], f"{w.name} block {b} left-fold mix ch{c}")
        if w.out_channels == 2:
            assert_bit_equal(g_out[0], g_out[1], "L == R")
        np.testing.assert_array_equal(g.read_done_frames(), o_done)
        assert (g_flags & L.FLAG_ANY_DONE) == (o_flags & L.FLAG_ANY_DONE)
    g.close()
    o.close()


def test_c1_readme_example(knh, oracle):
    w = configs.config("C1")
    run_pair(knh, oracle, w, 4, mix_mode=L.MIX_LEFT_FOLD)


def c3_events(w):
    def ev(block, bank):
        if block == 0:
            fire_all(bank, w.n_voices, *w.restart)
        if block == 2:
            fire_all(bank, w.n_voices, w.release[0], w.release[1])
    return ev


@pytest.mark.parametrize("mix_mode", [L.MIX_TREE, L.MIX_LEFT_FOLD])
def test_c3_chain_bit_exact(knh, oracle, mix_mode):
    w = configs.config("C3", n_voices=320, block_size=512)
    run_pair(knh, oracle, w, 5, c3_events(w), mix_mode)


def test_c4_f64_chain_bit_exact(knh, oracle):
    w = configs.config("C4", n_voices=192, block_size=512)
    run_pair(knh, oracle, w, 4, c3_events(w), L.MIX_LEFT_FOLD)


def test_c2_sin_numeric_within_tolerance(knh, oracle):
    w = configs.config("C2", n_voices=256, block_size=256)
    # the device's sine (f32: v_sin_f32 on the phase in revolutions, within 8.7e-7 of glibc's sinf of the rounded product over
    # every phase of [0, 2): tools/micro/hw_sin.hip) is not bit-exact by design; the phase accumulation is, so errors do not
    # grow.  The tolerance is the north star's 1e-5 on a full-scale sine (the voices' gain is 1/256).
    run_pair(knh, oracle, w, 6, voice_tol=1e-5 / 256)
    w64 = configs.config("C2", n_voices=64, block_size=256, sample_type=L.F64)
    run_pair(knh, oracle, w64, 3, voice_tol=1e-12)


@pytest.mark.parametrize("form", ["single", "pipe"])
def test_sin_numeric_phases_of_several_turns(knh, oracle, monkeypatch, form):
    """A phase offset of several turns, a negative frequency running away from zero: the reference's f32 rounding of
    (phase + offset) * TAU grows with the phase and is part of its value; the device then takes the library sine of the same
    rounded product (SinNum::sin_turns), still within 1e-5 of the reference -- at 40 turns the exact sine is 1e-5 away."""
    monkeypatch.setenv("KNH_PIPELINE", "0" if form == "single" else "1")
    n = 192
    w = configs.config("C2", n_voices=n, block_size=256)
    w.ctor[0][64:128, 0] *= -1.0      # negative frequencies: the phase only wraps above one (osc.rs:266-268)
    off = np.zeros(n)
    off[:64] = np.linspace(1.5, 40.0, 64)   # some below two turns (the hardware sine), most above
    off[128:] = np.linspace(-40.0, -1.5, 64)

    def ev(block, bank):
        if block == 0:
            bank.param_apply_many(np.arange(n, dtype=np.uint32), 0, 1, L.VALUE_FLOAT, off)
    run_pair(knh, oracle, w, 8, ev, voice_tol=1e-5 / n)


def test_sin_numeric_value_does_not_depend_on_the_wavefront_mates(knh, monkeypatch):
    """Which sine a sample takes (hardware below two turns, library beyond) is decided per voice and sample, so a voice's signal
    is the same whoever shares its wavefront: every voice alone in a bank against all of them in one, in both kernel forms."""
    n = 70
    w = configs.config("C2", n_voices=n, block_size=64)
    off = np.where(np.arange(n) % 3 == 0, 17.25, 0.125)   # every third voice far out, its neighbours not
    many = {}
    for form in ("0", "1"):
        monkeypatch.setenv("KNH_PIPELINE", form)
        g = make_gpu(knh, w, L.MIX_LEFT_FOLD)
        g.param_apply_many(np.arange(n, dtype=np.uint32), 0, 1, L.VALUE_FLOAT, off)
        _, many[form], _ = g.process_block_voices()
        g.close()
    assert_bit_equal(many["0"], many["1"], "single-wave against pipelined")
    for v in (0, 1, 3, 64, 69):
        w1 = configs.config("C2", n_voices=1, block_size=64)
        w1.ctor = {0: w.ctor[0][v:v + 1], 1: w.ctor[1][v:v + 1]}
        g = make_gpu(knh, w1, L.MIX_LEFT_FOLD)
        g.param_apply_many(np.zeros(1, dtype=np.uint32), 0, 1, L.VALUE_FLOAT, off[v:v + 1])
        _, alone, _ = g.process_block_voices()
        g.close()
        assert_bit_equal(alone[0], many["1"][v], f"voice {v} alone")


def test_c5_audio_rate_fm_with_sample_accurate_changes(knh, oracle):
    w = configs.config("C5", n_voices=192, block_size=128)

    def ev(block, bank):
        e = configs.c5_events(w, block)
        if e is not None:
            voices, stages, params, kinds, fvalues, delays = e
            bank.param_apply_many(voices, stages, params, kinds, fvalues, None, delays)
    run_pair(knh, oracle, w, 8, ev, L.MIX_LEFT_FOLD)


def test_fm_saturating_frequency_cast(knh, oracle):
    """Negative / huge instantaneous frequencies: Rust `as u32` saturates (osc.rs:129)."""
    w = configs.config("C5", n_voices=64, block_size=128)
    w.ctor[1] = np.full((64, 1), 3.0e9)   # index: +-3e9 Hz swings -> negative and > u32::MAX increments
    w.ctor[2] = np.linspace(-1000.0, 1000.0, 64).reshape(64, 1)
    run_pair(knh, oracle, w, 3, None, L.MIX_TREE)


@pytest.mark.parametrize("ty", range(9))
def test_every_svf_type_and_its_setters(knh, oracle, ty):
    n = 64
    w = configs.Workload("svf", [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_SVF)], n, 256, L.F32, 1)
    p = configs.voice_parameters(n)
    w.ctor = {0: p["freq"].reshape(n, 1),
              1: np.stack([np.full(n, float(ty)), p["cutoff"], p["q"], np.linspace(-12, 12, n)], axis=1)}
    v = np.arange(n, dtype=np.uint32)

    def ev(block, bank):
        if block == 1:
            bank.param_apply_many(v, 1, 0, L.VALUE_FLOAT, p["cutoff"] * 0.5)
        if block == 2:
            bank.param_apply_many(v, 1, 1, L.VALUE_FLOAT, p["q"] + 1.0)
            bank.param_apply_many(v, 1, 2, L.VALUE_FLOAT, np.full(n, 3.0))
        if block == 3:
            bank.param_apply_many(v, 1, 3, L.VALUE_INTEGER, None, (v.astype(np.int64) + ty) % 11)  # 9,10 -> Low
            bank.param_apply_many(v, 1, 4, L.VALUE_TRIGGER)
    run_pair(knh, oracle, w, 5, ev)


@pytest.mark.parametrize("sample_type", [L.F32, L.F64])
def test_onepole_and_envelope_chains(knh, oracle, sample_type):
    n = 96
    p = configs.voice_parameters(n)
    gain = np.full((n, 1), 1.0 / n)
    lp = configs.Workload("lp", [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_ONEPOLE_LPF), Stage(L.STAGE_MUL_ENV_ASR),
                                 Stage(L.STAGE_MUL_CONST)], n, 128, sample_type, 2)
    lp.ctor = {0: p["freq"].reshape(n, 1), 1: p["cutoff"].reshape(n, 1),
               2: np.stack([p["attack"] * 0.1, p["release"] * 0.02], axis=1), 3: gain}
    hp = configs.Workload("hp", [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_ONEPOLE_HPF), Stage(L.STAGE_MUL_ENV_AR),
                                 Stage(L.STAGE_MUL_CONST)], n, 128, sample_type, 2)
    hp.ctor = {0: p["freq"].reshape(n, 1), 2: np.stack([p["attack"] * 0.1, p["release"] * 0.02], axis=1), 3: gain}
    v = np.arange(n, dtype=np.uint32)

    def ev_lp(block, bank):
        if block == 0:
            fire_all(bank, n, 2, 3)
        if block == 1:  # early release for half the voices while still attacking (release_scale = t)
            bank.param_apply_many(v[::2], 2, 2, L.VALUE_TRIGGER)
            bank.param_apply_many(v, 1, 0, L.VALUE_FLOAT, p["cutoff"] * 2.0)
        if block == 4:
            fire_all(bank, n, 2, 2)
        if block == 6:  # restart from wherever t is (envelopes.rs:47-49)
            fire_all(bank, n, 2, 3)
            bank.param_apply_many(v, 2, 0, L.VALUE_FLOAT, np.zeros(n))  # attack 0 -> rate 1

    def ev_hp(block, bank):
        if block in (0, 5):
            fire_all(bank, n, 2, 2)
        if block == 2:
            bank.param_apply_many(v, 1, 0, L.VALUE_FLOAT, p["cutoff"])
            bank.param_apply_many(v, 2, 1, L.VALUE_FLOAT, p["release"] * 0.01)
    run_pair(knh, oracle, lp, 9, ev_lp, L.MIX_LEFT_FOLD)
    run_pair(knh, oracle, hp, 8, ev_hp, L.MIX_LEFT_FOLD)


def test_many_sines_shape_and_math_stages(knh, oracle):
    n = 130
    p = configs.voice_parameters(n)
    ms = configs.Workload("many_sines", [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_WR_MUL), Stage(L.STAGE_MUL_ENV_AR)], n, 64,
                          L.F32, 2)
    ms.ctor = {0: p["freq"].reshape(n, 1), 1: np.full((n, 1), 0.0125), 2: np.tile([0.01, 0.1], (n, 1))}
    v = np.arange(n, dtype=np.uint32)

    def ev(block, bank):
        if block % 3 == 0:
            bank.param_apply_many(v, 2, 2, L.VALUE_TRIGGER)
            bank.param_apply_many(v, 0, 0, L.VALUE_FLOAT, p["freq"] * (1 + block))
        if block == 4:
            bank.param_apply_many(v, 1, 0, L.VALUE_FLOAT, np.full(n, 0.02))  # "wr_mul"
            bank.param_apply_many(v, 0, 1, L.VALUE_FLOAT, np.linspace(0, 16383, n))  # phase_offset
            bank.param_apply_many(v[::3], 0, 2, L.VALUE_TRIGGER)  # reset_phase
    run_pair(knh, oracle, ms, 8, ev, L.MIX_LEFT_FOLD)
    mt = configs.Workload("math", [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_ADD_CONST), Stage(L.STAGE_SUB_CONST),
                                   Stage(L.STAGE_DIV_CONST)], n, 64, L.F32, 1)
    mt.ctor = {0: p["freq"].reshape(n, 1), 1: p["q"].reshape(n, 1), 2: p["attack"].reshape(n, 1), 3: p["cutoff"].reshape(n, 1)}
    run_pair(knh, oracle, mt, 3, None, L.MIX_LEFT_FOLD)


@pytest.mark.parametrize("n_voices,block_size", [(1, 1), (63, 7), (65, 100), (130, 64), (64, 513)])
def test_ragged_sizes(knh, oracle, n_voices, block_size):
    w = configs.config("C3", n_voices=n_voices, block_size=block_size)
    run_pair(knh, oracle, w, 4, c3_events(w), L.MIX_LEFT_FOLD)


def test_precise_timing_semantics(knh, oracle):
    """WrPreciseTiming quirks (precise_timing.rs): FIFO head-of-line blocking, armed delays persisting,
    capacity overflow dropping changes, changes due past the block being lost."""
    n = 64
    w = configs.Workload("pt", [Stage(L.STAGE_SIN_WT, delayed_changes_per_block=3), Stage(L.STAGE_SVF, delayed_changes_per_block=2),
                                Stage(L.STAGE_MUL_ENV_ASR, delayed_changes_per_block=4)], n, 64, L.F32, 1)
    p = configs.voice_parameters(n)
    w.ctor = {0: p["freq"].reshape(n, 1), 1: np.stack([np.zeros(n), p["cutoff"], p["q"], np.zeros(n)], axis=1),
              2: np.tile([0.0005, 0.001], (n, 1))}
    rng = np.random.default_rng(7)
    script = []
    for block in range(10):
        ops = []
        for _ in range(rng.integers(0, 40)):
            voice = int(rng.integers(0, n))
            stage = int(rng.integers(0, 3))
            if stage == 0:
                param = int(rng.integers(0, 3))
                value = [float(rng.uniform(50, 5000)), float(rng.uniform(0, 16000)), None][param]
            elif stage == 1:
                param = int(rng.integers(0, 3))
                value = [float(rng.uniform(100, 9000)), float(rng.uniform(0.5, 4)), float(rng.uniform(-6, 6))][param]
            else:
                param = int(rng.integers(0, 4))
                value = [float(rng.uniform(0.0001, 0.002)), float(rng.uniform(0.0001, 0.002)), None, None][param]
            delay = int(rng.choice([0, 0, 1, 5, 17, 63, 64, 70]))
            arm = bool(rng.integers(0, 2))
            ops.append((voice, stage, param, value, delay, arm))
        script.append(ops)

    def ev(block, bank):
        from knaster_amd.bank import TRIGGER
        for voice, stage, param, value, delay, arm in script[block]:
            if arm:
                bank.set_delay_within_block_for_param(voice, stage, param, delay)
            bank.param_apply(voice, stage, param, TRIGGER if value is None else value)
    run_pair(knh, oracle, w, 10, ev)


def test_precise_timing_on_a_stage_beyond_index_255(knh, oracle):
    """A WrPreciseTiming-wrapped node far down a long voice (stage index 258): the host's compact record of a delayed call
    carries the stage index in 16 bits (an 8-bit field once sent such a change to stage 258 & 0xFF = 2).  Single calls and
    batched ones, against the oracle."""
    n, n_pad = 96, 256
    p = configs.voice_parameters(n)
    st = [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_WR_MUL)] + [Stage(L.STAGE_ADD_CONST) for _ in range(n_pad)] + \
         [Stage(L.STAGE_SVF, delayed_changes_per_block=4), Stage(L.STAGE_MUL_CONST, delayed_changes_per_block=2)]
    svf, gain = 2 + n_pad, 3 + n_pad
    assert svf == 258
    w = configs.Workload("deep", st, n, 64, L.F32, 1)
    w.ctor = {0: p["freq"].reshape(n, 1), 1: np.full((n, 1), 0.5), svf: np.stack([np.zeros(n), p["cutoff"], p["q"], np.zeros(n)], axis=1),
              gain: np.full((n, 1), 1.0 / n)}
    for k in range(n_pad):
        w.ctor[2 + k] = np.full((n, 1), 0.0 if k % 2 else 1e-3 * (k % 5))
    rng = np.random.default_rng(258)

    def ev(block, bank):
        if block % 2 == 0:  # single calls: arm, then the value
            for voice in rng_voices[block]:
                bank.set_delay_within_block_for_param(int(voice), svf, 0, int(5 + voice % 50))
                bank.param_apply(int(voice), svf, 0, float(300.0 + 10.0 * voice + 100.0 * block))
                bank.set_delay_within_block_for_param(int(voice), gain, 0, int(9 + voice % 40))
                bank.param_apply(int(voice), gain, 0, float(0.5 / n + 1e-4 * block))
        else:  # the batched entry point: every voice, delays from 1 to 63
            v = np.arange(n, dtype=np.uint32)
            bank.param_apply_many(v, svf, 1, L.VALUE_FLOAT, 0.6 + 0.01 * v + 0.1 * block, None, (1 + (7 * v) % 63).astype(np.uint16))
    rng_voices = {b: rng.choice(n, size=20, replace=False) for b in range(0, 6, 2)}
    run_pair(knh, oracle, w, 6, ev)


def test_fma_variant_within_tolerance(knh, oracle):
    w = configs.config("C3", n_voices=256, block_size=512)
    run_pair(knh, oracle, w, 4, c3_events(w), allow_fma=True, voice_tol=1e-5 / 256 * 8)


def test_done_flags_and_all_done(knh, oracle):
    n = 70
    w = configs.Workload("done", [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_MUL_ENV_ASR)], n, 16, L.F32, 2)
    w.ctor = {0: np.full((n, 1), 440.0), 1: np.tile([0.0, 0.0], (n, 1))}  # graph.rs:2483-2513

    def ev(block, bank):
        if block == 1:
            fire_all(bank, n, 1, 3)
        if block == 2:
            fire_all(bank, n, 1, 2)
    run_pair(knh, oracle, w, 5, ev, L.MIX_LEFT_FOLD)
    g = make_gpu(knh, w)
    _, flags = g.process_block()
    assert flags & L.FLAG_ALL_DONE  # never started: every envelope Stopped
    fire_all(g, n, 1, 3)
    _, flags = g.process_block()
    assert not (flags & L.FLAG_ALL_DONE)
    fire_all(g, n, 1, 2)
    _, flags = g.process_block()
    assert flags & L.FLAG_ANY_DONE
    _, flags = g.process_block()
    assert flags & L.FLAG_ALL_DONE and not (flags & L.FLAG_ANY_DONE)
    g.close()


def test_error_behaviour_on_device(knh):
    w = configs.config("C3", n_voices=8, block_size=32)
    g = make_gpu(knh, w)
    with pytest.raises(L.KnasterHipError) as e:
        g.param_apply(8, 0, 0, 1.0)
    assert e.value.status == L.ERR_OUT_OF_RANGE
    with pytest.raises(L.KnasterHipError) as e:
        g.param_apply(0, 0, 3, 1.0)  # SinWt has 3 params
    assert e.value.status == L.ERR_OUT_OF_RANGE
    with pytest.raises(L.KnasterHipError) as e:
        g.param_apply(0, 2, 0, knh.TRIGGER)  # cutoff_freq wants a float
    assert e.value.status == L.ERR_WRONG_VALUE_KIND
    with pytest.raises(L.KnasterHipError) as e:
        g.process_block(frames_to_process=33)
    assert e.value.status == L.ERR_INVALID_ARGUMENT
    out, _ = g.process_block()  # still usable after errors ("log and continue")
    assert out.shape == (2, 32)
    g.close()


def test_smooth_params_block_rate_ramps(knh, oracle):
    """WrSmoothParams (smooth_params.rs): linear block-rate ramps on SinWt freq, SVF cutoff/q and a constant,
    retargeted mid-ramp, switched off mid-ramp, and combined with sample-accurate changes (precise timing
    outside: every partial block steps the ramp, as in the reference)."""
    n = 70
    p = configs.voice_parameters(n)
    SM = L.STAGE_FLAG_SMOOTH_PARAMS
    w = configs.Workload("smooth", [Stage(L.STAGE_SIN_WT, flags=SM), Stage(L.STAGE_SVF, flags=SM, delayed_changes_per_block=4),
                                    Stage(L.STAGE_MUL_CONST, flags=SM)], n, 64, L.F32, 2)
    w.ctor = {0: p["freq"].reshape(n, 1), 1: np.stack([np.zeros(n), p["cutoff"], p["q"], np.zeros(n)], axis=1),
              2: np.full((n, 1), 1.0 / n)}
    v = np.arange(n, dtype=np.uint32)
    KS = L.VALUE_SMOOTHING

    def ev(block, bank):
        if block == 0:
            bank.param_apply_many(v, 0, 0, KS, np.full(n, 0.01), np.ones(n, dtype=np.int64))     # freq: 10 ms linear
            bank.param_apply_many(v, 1, 0, KS, np.full(n, 0.004), np.ones(n, dtype=np.int64))    # cutoff: 4 ms
            bank.param_apply_many(v, 2, 0, KS, np.full(n, 0.02), np.ones(n, dtype=np.int64))     # gain
            # the wrapper starts from current_value 0 (smooth_params.rs:26 "TODO: Initialise state to default"):
            bank.param_apply_many(v, 0, 0, L.VALUE_FLOAT, p["freq"] * 2)
            bank.param_apply_many(v, 1, 0, L.VALUE_FLOAT, p["cutoff"] * 0.25)
            bank.param_apply_many(v, 2, 0, L.VALUE_FLOAT, np.full(n, 2.0 / n))
        if block == 2:   # retarget mid-ramp
            bank.param_apply_many(v[::2], 0, 0, L.VALUE_FLOAT, p["freq"][::2] * 0.5)
        if block == 3:   # q is not smoothed (no Smoothing value sent): passes straight through; delayed, so it splits the block
            bank.param_apply_many(v, 1, 1, L.VALUE_FLOAT, p["q"] + 0.5, None, np.full(n, 17, dtype=np.uint16))
            bank.param_apply_many(v[1::2], 1, 0, L.VALUE_FLOAT, p["cutoff"][1::2], None, np.full(len(v[1::2]), 40, dtype=np.uint16))
        if block == 5:   # smoothing off mid-ramp on a third of the voices, new duration on another third
            bank.param_apply_many(v[::3], 2, 0, KS, np.zeros(len(v[::3])), np.zeros(len(v[::3]), dtype=np.int64))
            bank.param_apply_many(v[1::3], 2, 0, KS, np.full(len(v[1::3]), 0.002), np.full(len(v[1::3]), 2, dtype=np.int64))
            bank.param_apply_many(v, 2, 0, L.VALUE_FLOAT, np.full(n, 0.5 / n))
    run_pair(knh, oracle, w, 12, ev, L.MIX_LEFT_FOLD)


def _segment_envelope_args(n, n_max, rng, looping=False):
    """[start, time_scale, looping, n_segments, (duration, value) * n_max] per voice; ragged segment counts."""
    a = np.zeros((n, 4 + 2 * n_max))
    a[:, 0] = rng.uniform(-0.5, 0.5, n)
    a[:, 1] = rng.uniform(0.5, 2.0, n)
    a[:, 2] = 1.0 if looping else 0.0
    a[:, 3] = rng.integers(1, n_max + 1, n)
    a[:, 4::2] = rng.uniform(0.0005, 0.004, (n, n_max))  # 24 .. 192 samples at 48 kHz
    a[:, 5::2] = rng.uniform(-1.0, 1.0, (n, n_max))
    return a


@pytest.mark.parametrize("sample_type", [L.F32, L.F64])
@pytest.mark.parametrize("looping", [False, True])
def test_segment_envelope(knh, oracle, sample_type, looping):
    """Envelope (envelopes.rs:359-527): f64 segment state, ragged segment counts, every parameter, done at frame 0."""
    n, n_max = 100, 5
    rng = np.random.default_rng(5)
    p = configs.voice_parameters(n)
    w = configs.Workload("segenv", [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_WR_MUL), Stage(L.STAGE_MUL_ENVELOPE, delayed_changes_per_block=2)],
                         n, 64, sample_type, 2)
    w.ctor = {0: p["freq"].reshape(n, 1), 1: np.full((n, 1), 1.0 / n), 2: _segment_envelope_args(n, n_max, rng, looping)}
    v = np.arange(n, dtype=np.uint32)

    def ev(block, bank):
        if block == 1:
            fire_all(bank, n, 2, 2)  # t_restart
        if block == 3:  # t_stop mid-segment for a third of the voices, at a sample-accurate offset
            sel = v[::3]
            bank.param_apply_many(sel, 2, 3, L.VALUE_TRIGGER, delays=(sel % 64).astype(np.uint16))
        if block == 4:  # jump (clamped to the last segment for most voices) and a new time_scale
            bank.param_apply_many(v, 2, 1, L.VALUE_INTEGER, ivalues=(v % 7).astype(np.int64))
            bank.param_apply_many(v, 2, 0, L.VALUE_FLOAT, 0.25 + (v % 5) * 0.6)
        if block == 9:
            bank.param_apply_many(v[1::2], 2, 2, L.VALUE_TRIGGER, delays=np.full(n // 2, 33, dtype=np.uint16))
    run_pair(knh, oracle, w, 14, ev, L.MIX_LEFT_FOLD)


def test_segment_envelope_in_a_jit_fused_chain(knh, oracle):
    n, n_max = 70, 3
    rng = np.random.default_rng(11)
    p = configs.voice_parameters(n)
    w = configs.Workload("segenv_jit", [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_SVF), Stage(L.STAGE_MUL_ENVELOPE), Stage(L.STAGE_MUL_CONST)],
                         n, 128, L.F32, 2)
    svf = np.stack([np.full(n, float(L.SVF_LOW)), p["cutoff"], p["q"], np.zeros(n)], axis=1)
    w.ctor = {0: p["freq"].reshape(n, 1), 1: svf, 2: _segment_envelope_args(n, n_max, rng), 3: np.full((n, 1), 1.0 / n)}

    def ev(block, bank):
        if block in (0, 5):
            fire_all(bank, n, 2, 2)
    run_pair(knh, oracle, w, 8, ev, L.MIX_LEFT_FOLD)


@pytest.mark.parametrize("sample_type", [L.F32, L.F64])
def test_remaining_math_wrappers_exact(knh, oracle, sample_type):
    """WrVSub / WrDiv / WrVDiv / WrPowi (wrappers_core/math.rs:272-505, 587-661): + - * / only, so bit-exact.
    powi exponents cover 0, 1, negative and large values (multiply-by-squaring order matters for the bits)."""
    n = 80
    p = configs.voice_parameters(n)
    v = np.arange(n)
    w = configs.Workload("wrappers", [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_WR_VSUB), Stage(L.STAGE_WR_DIV), Stage(L.STAGE_WR_POWI),
                                      Stage(L.STAGE_WR_VDIV), Stage(L.STAGE_MUL_CONST)], n, 96, sample_type, 2)
    w.ctor = {0: p["freq"].reshape(n, 1), 1: (1.5 + 0.01 * v).reshape(n, 1), 2: (0.3 + 0.02 * v).reshape(n, 1),
              3: ((v % 13) - 4).astype(np.float64).reshape(n, 1), 4: (2.0 + 0.1 * v).reshape(n, 1), 5: np.full((n, 1), 1e-3 / n)}
    run_pair(knh, oracle, w, 3, None, L.MIX_LEFT_FOLD)


def test_powf_wrapper_and_pow_node_within_tolerance(knh, oracle):
    """WrPowf (wrappers_core/math.rs:508-584) and MathUGen Pow with a Constant (math.rs:75-85): device libm pow,
    a few ulp from the host's powf -- tolerance, like SinNumeric."""
    n = 64
    p = configs.voice_parameters(n)
    v = np.arange(n)
    w = configs.Workload("powf", [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_WR_MUL), Stage(L.STAGE_WR_ADD), Stage(L.STAGE_WR_POWF),
                                  Stage(L.STAGE_POW_CONST), Stage(L.STAGE_MUL_CONST)], n, 128, L.F32, 2)
    w.ctor = {0: p["freq"].reshape(n, 1), 1: np.full((n, 1), 0.4), 2: np.full((n, 1), 1.0), 3: (0.5 + 0.05 * v).reshape(n, 1),
              4: (1.0 + 0.02 * v).reshape(n, 1), 5: np.full((n, 1), 1.0 / n)}

    def ev(block, bank):
        if block == 1:
            bank.param_apply_many(v.astype(np.uint32), 4, 0, L.VALUE_FLOAT, 2.0 - 0.01 * v)
    run_pair(knh, oracle, w, 3, ev, voice_tol=2e-6)


@pytest.mark.parametrize("sample_type", [L.F32, L.F64])
def test_sample_delay_ring_in_hbm(knh, oracle, sample_type):
    """SampleDelay (delay.rs:14-50): per-voice rings of different lengths, delays of 0, 1, just below / at / above the
    tile length, half the ring, the whole ring; wrap-around of both pointers; delay changes at block starts and at
    sample-accurate offsets; a delay longer than the ring is ignored on both sides."""
    n, bs = 96, 64
    p = configs.voice_parameters(n)
    v = np.arange(n, dtype=np.uint32)
    w = configs.Workload("delay", [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_WR_MUL), Stage(L.STAGE_SAMPLE_DELAY, delayed_changes_per_block=2),
                                   Stage(L.STAGE_MUL_CONST)], n, bs, sample_type, 2)
    max_delay = 0.002 + 0.0001 * (v % 40)  # 96 .. 283 samples at 48 kHz, through Seconds::from_secs_f64
    ring = np.array([int((int(np.floor(s)) + int((s - np.floor(s)) * 282240000.0) / 282240000.0) * 48000.0) for s in max_delay])
    w.ctor = {0: p["freq"].reshape(n, 1), 1: np.full((n, 1), 0.5), 2: max_delay.reshape(n, 1), 3: np.full((n, 1), 1.0 / n)}
    kinds = [0, 1, 5, 15, 16, 17, 31, 32, 33, -2, -1, -3]  # samples; negative: relative to the ring length (-1 = whole ring)

    def delay_samples(shift):
        k = np.array([kinds[(i + shift) % len(kinds)] for i in range(n)])
        d = np.where(k >= 0, k, ring + 1 + k)
        d = np.where(k == -3, ring // 2, d)
        return np.minimum(d, ring)

    def ev(block, bank):
        if block == 0:
            bank.param_apply_many(v, 2, 0, L.VALUE_FLOAT, (delay_samples(0) + 0.25) / 48000.0)
        if block == 5:
            bank.param_apply_many(v, 2, 0, L.VALUE_FLOAT, (delay_samples(3) + 0.25) / 48000.0)
        if block == 7:  # sample-accurate change inside the block; every third voice asks for more than its ring holds
            too_long = (ring + 2.5) / 48000.0
            want = np.where(v % 3 == 0, too_long, (delay_samples(7) + 0.25) / 48000.0)
            bank.param_apply_many(v, 2, 0, L.VALUE_FLOAT, want, delays=(v % bs).astype(np.uint16))
    run_pair(knh, oracle, w, 12, ev, L.MIX_LEFT_FOLD)


def test_sample_delay_multi_block_launch_equals_single_blocks(knh):
    n, bs, blocks = 200, 128, 6
    p = configs.voice_parameters(n)
    w = configs.Workload("delay_mb", [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_SVF), Stage(L.STAGE_SAMPLE_DELAY), Stage(L.STAGE_MUL_CONST)],
                         n, bs, L.F32, 2)
    svf = np.stack([np.full(n, float(L.SVF_LOW)), p["cutoff"], p["q"], np.zeros(n)], axis=1)
    w.ctor = {0: p["freq"].reshape(n, 1), 1: svf, 2: np.full((n, 1), 0.0101), 3: np.full((n, 1), 1.0 / n)}
    d = (np.arange(n) * 7 % 480) / 48000.0 + 1e-6
    a = make_gpu(knh, w)
    b = make_gpu(knh, w)
    for bank in (a, b):
        bank.param_apply_many(np.arange(n, dtype=np.uint32), 2, 0, L.VALUE_FLOAT, d)
    one = np.stack([a.process_block()[0] for _ in range(blocks)])
    many, _ = b.process_blocks(blocks)
    assert_bit_equal(np.asarray(many).reshape(one.shape), one, "multi-block launch with a delay stage")
    a.close()
    b.close()


def test_c3_with_delay_line_against_oracle(knh, oracle):
    """The D3 bench chain (C3 with a SampleDelay behind the filter) on the pipelined kernel, per voice bit for bit."""
    w = configs.config("D3", n_voices=130, block_size=128)
    w.ctor[3] = np.full((130, 1), 0.02)  # 960-sample rings keep the oracle run short and make the pointers wrap
    v = np.arange(130, dtype=np.uint32)

    def ev(block, bank):
        if block == 0:
            bank.param_apply_many(v, 4, 3, L.VALUE_TRIGGER)
            bank.param_apply_many(v, 3, 0, L.VALUE_FLOAT, w.delay_times * 0.09)
        if block == 6:
            bank.param_apply_many(v, 4, 2, L.VALUE_TRIGGER)
    run_pair(knh, oracle, w, 12, ev, L.MIX_LEFT_FOLD)


@pytest.mark.parametrize("sample_type", [L.F32, L.F64])
def test_phasor_and_safety_limiter(knh, oracle, sample_type):
    """Phasor (osc.rs:172-214: f64 ramp, `while phase >= 1` wrap, negative and > sample-rate frequencies) into
    SafetyLimiter (dynamics.rs:9-31: clamp to +-1, NaN -> 0; a division by zero upstream supplies the NaN and infinities)."""
    n = 90
    v = np.arange(n, dtype=np.uint32)
    w = configs.Workload("phasor", [Stage(L.STAGE_PHASOR, delayed_changes_per_block=1), Stage(L.STAGE_WR_MUL), Stage(L.STAGE_WR_SUB),
                                    Stage(L.STAGE_DIV_CONST), Stage(L.STAGE_SAFETY_LIMITER), Stage(L.STAGE_MUL_CONST)], n, 96, sample_type, 2)
    freq = 20.0 + 37.0 * v
    w.ctor = {0: freq.reshape(n, 1), 1: np.full((n, 1), 3.0), 2: np.full((n, 1), 1.5),
              3: np.where(v % 9 == 0, 0.0, 0.5).reshape(n, 1),  # x / 0: +-inf and, where x == 0, NaN
              5: np.full((n, 1), 1.0 / n)}

    def ev(block, bank):
        if block == 2:  # more than one wrap per sample, and a ramp that runs backwards
            bank.param_apply_many(v, 0, 0, L.VALUE_FLOAT, np.where(v % 2 == 0, 48000.0 * 2.75, -333.0), delays=(v % 96).astype(np.uint16))
        if block == 4:
            bank.param_apply_many(v, 0, 0, L.VALUE_FLOAT, freq * 0.5)
    run_pair(knh, oracle, w, 6, ev, L.MIX_TREE)


@pytest.mark.parametrize("sample_type", [L.F32, L.F64])
def test_polyblep_all_waveforms(knh, oracle, sample_type):
    """PolyBlep (polyblep.rs:123-508): fourteen waveforms, pulse widths incl. the clamped extremes, frequency, pulse-width
    and waveform changes.  Waveforms built from + - * / and comparisons are bit-exact; the four that call sin (and any
    voice at or above sample_rate / 4, which the reference renders as a sine) are compared with a tolerance."""
    n, bs = 14 * 8, 128
    v = np.arange(n, dtype=np.uint32)
    wf = (v % 14).astype(np.float64)
    freq = 55.0 * 2.0 ** ((v // 14) * 0.9)  # 55 Hz .. 4.3 kHz
    w = configs.Workload("polyblep", [Stage(L.STAGE_POLYBLEP, delayed_changes_per_block=1), Stage(L.STAGE_MUL_CONST)], n, bs, sample_type, 2)
    w.ctor = {0: np.stack([wf, freq], axis=1), 1: np.full((n, 1), 1.0 / n)}
    g, o = make_gpu(knh, w), make_oracle(oracle, w)
    cur_wf, cur_freq = wf.copy(), freq.copy()
    for block in range(8):
        if block == 1:  # pulse widths across and beyond the clamps of tri2 / trap2, and 0 for trip's guard
            pw = np.choose(v % 6, [0.5, 0.1, 0.9, 0.99995, 0.00001, 0.0])
            for b in (g, o):
                b.param_apply_many(v, 0, 1, L.VALUE_FLOAT, pw)
        if block == 3:  # every third voice above sample_rate / 4 (rendered as a sine), at a sample-accurate offset
            sel = v[::3]
            cur_freq[::3] = 12000.0 + 100.0 * (sel % 7)
            for b in (g, o):
                b.param_apply_many(sel, 0, 0, L.VALUE_FLOAT, cur_freq[::3], delays=(sel % bs).astype(np.uint16))
        if block == 5:  # rotate the waveforms; 14 and 99 are out of range -> Sawtooth
            new_wf = (v + 5) % 16
            new_wf = np.where(new_wf == 15, 99, new_wf)
            cur_wf = np.where(new_wf < 14, new_wf, 0).astype(np.float64)
            for b in (g, o):
                b.param_apply_many(v, 0, 2, L.VALUE_INTEGER, ivalues=new_wf.astype(np.int64))
        _, gv, _ = g.process_block_voices()
        _, ov, _, _ = o.process_block()
        exact = ~np.isin(cur_wf, [1, 2, 9, 10]) & (cur_freq < 12000.0)
        if block in (3, 5):  # the block in which a change lands: the voice is exact or not depending on the frame
            exact &= False
        assert_bit_equal(gv[exact], ov[exact], f"block {block}: waveforms without sin")
        assert np.max(np.abs(gv.astype(np.float64) - ov)) <= 4e-6 / n, f"block {block}"
        assert np.abs(ov).max() > 0.1 / n
    g.close()
    o.close()


@pytest.mark.parametrize("kind", [L.STAGE_ALLPASS_DELAY, L.STAGE_ALLPASS_FB_DELAY])
@pytest.mark.parametrize("sample_type", [L.F32, L.F64])
def test_allpass_delay(knh, oracle, sample_type, kind):
    """AllpassDelay (delay.rs:93-206) and AllpassFeedbackDelay (the Schroeder allpass around it, :210-306; feedback set at
    block 1 and changed later): fractional delays through the allpass interpolator, the untouched start (read and
    write pointers together: a whole ring of delay), delays shorter than a tile, pointer wrap-around, changes at block
    starts and mid-block, a delay_time of the ring length or more ignored."""
    n, bs = 96, 64
    p = configs.voice_parameters(n)
    v = np.arange(n, dtype=np.uint32)
    w = configs.Workload("allpass", [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_WR_MUL), Stage(kind, delayed_changes_per_block=2),
                                     Stage(L.STAGE_MUL_CONST)], n, bs, sample_type, 2)
    max_delay = 0.003 + 0.0001 * (v % 30)
    ring = np.array([int(s * 282240000.0) * 48000 // 282240000 for s in max_delay])  # Seconds::to_samples for s < 1
    w.ctor = {0: p["freq"].reshape(n, 1), 1: np.full((n, 1), 0.5), 2: max_delay.reshape(n, 1), 3: np.full((n, 1), 1.0 / n)}
    frames = [0.7, 1.2, 5.49, 15.5, 31.9, 32.0, 33.25, 64.75, -1.0, -2.0]  # negative: relative to the ring (-1: just inside, -2: half)

    def delay_seconds(shift):
        k = np.array([frames[(i + shift) % len(frames)] for i in range(n)])
        d = np.where(k >= 0, k, np.where(k == -1.0, ring - 1.25, ring * 0.5 + 0.3))
        return d / 48000.0

    def ev(block, bank):
        if kind == L.STAGE_ALLPASS_FB_DELAY and block in (1, 8):
            bank.param_apply_many(v, 2, 1, L.VALUE_FLOAT, (0.3 + 0.005 * v) * (1.0 if block == 1 else -0.9))
        if block == 2:
            bank.param_apply_many(v, 2, 0, L.VALUE_FLOAT, delay_seconds(0))
        if block == 6:
            bank.param_apply_many(v, 2, 0, L.VALUE_FLOAT, delay_seconds(4), delays=(v % bs).astype(np.uint16))
        if block == 9:  # too long for the ring on every other voice: ignored there
            want = np.where(v % 2 == 0, (ring + 3.0) / 48000.0, delay_seconds(7))
            bank.param_apply_many(v, 2, 0, L.VALUE_FLOAT, want)
    run_pair(knh, oracle, w, 14, ev, L.MIX_LEFT_FOLD)


@pytest.mark.parametrize("sample_type", [L.F32, L.F64])
def test_buffer_reader_sampler_bank(knh, oracle, sample_type):
    """BufferReader<F, U1> (buffer.rs:19-191) on one shared Buffer at another sample rate: per-voice rate, start and
    length, looping and one-shot voices (done at the frame after the last one, then silence), every parameter."""
    n, bs = 100, 64
    v = np.arange(n, dtype=np.uint32)
    rng = np.random.default_rng(3)
    t = np.arange(3000) / 44100.0
    samples = 0.5 * np.sin(2 * np.pi * 300.0 * t) + 0.3 * np.sin(2 * np.pi * 1234.5 * t) + 0.1 * rng.uniform(-1, 1, 3000)
    w = configs.Workload("sampler", [Stage(L.STAGE_BUFFER_READER), Stage(L.STAGE_MUL_CONST)], n, bs, sample_type, 2)
    rate = 0.25 + 0.03 * v
    looping = (v % 3 == 0).astype(np.float64)
    start = np.where(v % 4 == 0, 0.011 + 0.0001 * v, 0.0)
    w.ctor = {0: np.stack([rate, looping, start], axis=1), 1: np.full((n, 1), 1.0 / n)}
    w.buffer = (0, samples, 44100.0)

    def ev(block, bank):
        if block == 3:
            bank.param_apply_many(v, 0, 0, L.VALUE_FLOAT, rate * 2.5)                 # rate
            bank.param_apply_many(v[::2], 0, 3, L.VALUE_FLOAT, 0.004 + 0.0002 * v[::2])  # duration_s
        if block == 5:
            bank.param_apply_many(v, 0, 2, L.VALUE_FLOAT, 0.02 + 0.0001 * v)         # start_s
            bank.param_apply_many(v, 0, 5, L.VALUE_TRIGGER)                           # t_restart
        if block == 8:
            bank.param_apply_many(v[1::2], 0, 4, L.VALUE_FLOAT, 0.05 + 0.0001 * v[1::2])  # end_s
            bank.param_apply_many(v, 0, 1, L.VALUE_BOOL, ivalues=(v % 2).astype(np.int64))  # looping
        if block == 10:
            bank.param_apply_many(v, 0, 5, L.VALUE_TRIGGER)
    run_pair(knh, oracle, w, 14, ev, L.MIX_LEFT_FOLD)


@pytest.mark.parametrize("sample_type", [L.F32, L.F64])
@pytest.mark.parametrize("kind", [L.STAGE_WHITE_NOISE, L.STAGE_PINK_NOISE, L.STAGE_BROWN_NOISE])
def test_noise_sources(knh, oracle, kind, sample_type):
    """WhiteNoise / PinkNoise / BrownNoise (noise.rs:26-156), one generator per voice seeded the way the reference's
    process-wide counter would (0, 1, 2, ...), through a filter and an envelope (a run-time fused pipeline), block after
    block: the integer generator and the f32 arithmetic around it are bit-identical to the oracle's restatement.
    (What that restatement is worth against the real `fastrand` crate: parity unpinned)"""
    n, bs = 200, 96
    v = np.arange(n, dtype=np.uint32)
    p = configs.voice_parameters(n)
    w = configs.Workload("noise", [Stage(kind), Stage(L.STAGE_WR_MUL), Stage(L.STAGE_SVF), Stage(L.STAGE_MUL_ENV_ASR)], n, bs, sample_type, 2)
    seeds = np.where(v % 50 == 49, 2.0 ** 40 + v, v).astype(np.float64)  # small counters, and a few seeds above 2^32
    w.ctor = {0: seeds.reshape(n, 1), 1: np.full((n, 1), 1.0 / n),
              2: np.stack([np.zeros(n), p["cutoff"], p["q"], np.zeros(n)], axis=1),
              3: np.stack([p["attack"], p["release"]], axis=1)}

    def ev(block, bank):
        if block == 0:
            bank.param_apply_many(v, 3, 3, L.VALUE_TRIGGER)
        if block == 4:
            bank.param_apply_many(v[::2], 3, 2, L.VALUE_TRIGGER, delays=(v[::2] % bs).astype(np.uint16))
    run_pair(knh, oracle, w, 7, ev, L.MIX_TREE)


@pytest.mark.parametrize("sample_type", [L.F32, L.F64])
def test_random_lin(knh, oracle, sample_type):
    """RandomLin (noise.rs:158-230): the two draws of new() and init() happen on the host, the rest on the device;
    frequencies from well below one value per block to more than one per sample-pair, changed sample-accurately."""
    n, bs = 130, 64
    v = np.arange(n, dtype=np.uint32)
    w = configs.Workload("randomlin", [Stage(L.STAGE_RANDOM_LIN, delayed_changes_per_block=2), Stage(L.STAGE_MUL_CONST)], n, bs, sample_type, 2)
    freq = 3.0 * 1.07 ** v  # 3 Hz .. 19 kHz
    w.ctor = {0: np.stack([v.astype(np.float64) + 7.0, freq], axis=1), 1: np.full((n, 1), 1.0 / n)}

    def ev(block, bank):
        if block == 2:
            bank.param_apply_many(v, 0, 0, L.VALUE_FLOAT, freq[::-1].copy(), delays=(v % bs).astype(np.uint16))
        if block == 4:
            bank.param_apply_many(v[::3], 0, 0, L.VALUE_FLOAT, np.full(len(v[::3]), 48000.0))  # a new value every sample
    run_pair(knh, oracle, w, 7, ev, L.MIX_TREE)


@pytest.mark.parametrize("which", ["zero", "whole_ring"])
def test_sample_delay_of_zero_or_the_whole_ring_in_every_voice(knh, oracle, which):
    """delay.rs:29-43 with delay 0 or delay = ring length: the sample just written is the one read.  With EVERY voice of a
    wavefront in that situation the tile-wise (16-byte) path of the device stage must not be taken (it reads before it
    writes); in the ring test above some other voice of the wavefront always kept the wave on the per-sample path."""
    n, bs = 70, 64
    p = configs.voice_parameters(n)
    v = np.arange(n, dtype=np.uint32)
    w = configs.Workload("delay0", [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_SAMPLE_DELAY), Stage(L.STAGE_MUL_CONST)], n, bs, L.F32, 2)
    max_delay = np.full(n, 0.004)  # 192 samples
    w.ctor = {0: p["freq"].reshape(n, 1), 1: max_delay.reshape(n, 1), 2: np.full((n, 1), 1.0 / n)}

    def ev(block, bank):
        if block == 0:
            bank.param_apply_many(v, 1, 0, L.VALUE_FLOAT, np.full(n, 0.0 if which == "zero" else 192.25 / 48000.0))
        if block == 3:
            bank.param_apply_many(v, 1, 0, L.VALUE_FLOAT, np.full(n, 40.25 / 48000.0))
        if block == 5:
            bank.param_apply_many(v, 1, 0, L.VALUE_FLOAT, np.full(n, 0.0 if which == "zero" else 192.25 / 48000.0))
    run_pair(knh, oracle, w, 8, ev, L.MIX_LEFT_FOLD)
