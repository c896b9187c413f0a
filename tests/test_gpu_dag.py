"""Voices that are small graphs (DAGs) rather than chains on one running signal: a second source in the stage list, stages
that name the signal they read (knh_stage_desc.input), MathUGen<_, U1, Op> of two signals (KNH_STAGE_MATH_*).  The oracle
builds the same voice as the reference's graph API would (one node per UGen, connect / math_nodes); bar: bit-identical per
voice for everything built from + - * and table lookups."""
import os

import numpy as np
import pytest

from helpers import assert_bit_equal, fire_all, make_gpu, make_oracle
from knaster_amd import _lib as L
from knaster_amd import configs
from knaster_amd.bank import Stage, TRIGGER

pytestmark = pytest.mark.gpu


def run_pair(knh, oracle, w, blocks, events=None, tol=None):
    g = make_gpu(knh, w, L.MIX_LEFT_FOLD)
    o = make_oracle(oracle, w)
    peak = 0.0
    for b in range(blocks):
        if events:
            events(b, g)
            events(b, o)
        g_out, g_voices, g_flags = g.process_block_voices()
        o_out, o_voices, o_flags, o_done = o.process_block()
        if tol is None:
            assert_bit_equal(g_voices, o_voices, f"{w.name} block {b} per-voice")
            assert_bit_equal(g_out, o_out, f"{w.name} block {b} left-fold mix")
        else:
            assert np.max(np.abs(g_voices.astype(np.float64) - o_voices)) <= tol
        np.testing.assert_array_equal(g.read_done_frames(), o_done)
        peak = max(peak, float(np.abs(o_voices).max()))
    g.close()
    o.close()
    assert peak > 1e-3
    return peak


@pytest.mark.parametrize("sample_type", [L.F32, L.F64])
@pytest.mark.parametrize("op", [L.STAGE_MATH_MUL, L.STAGE_MATH_ADD, L.STAGE_MATH_SUB])
def test_two_oscillators_combined(knh, oracle, op, sample_type):
    """`a * b`, `a + b`, `a - b` of two SinWt (graph_edit.rs:936-971): ring modulation and friends, 200 voices."""
    n = 200
    p = configs.voice_parameters(n)
    w = configs.Workload("ab", [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_SIN_WT), Stage(op, input=1, input2=2), Stage(L.STAGE_MUL_CONST)],
                         n, 96, sample_type, 2)
    w.ctor = {0: p["freq"].reshape(n, 1), 1: (p["freq"] * p["fm_ratio"]).reshape(n, 1), 3: np.full((n, 1), 0.5 / n)}
    run_pair(knh, oracle, w, 3)


@pytest.mark.parametrize("depth,n_voices,block_size", [(4, 130, 32), (12, 64, 128), (24, 3, 16)])
def test_fm_cascade_voice(knh, oracle, depth, n_voices, block_size):
    """The reference's "FM cascade" bench shape (graph_dsp_performance.rs:37-72) as one graph-shaped voice per lane."""
    w = configs.fm_cascade(depth, n_voices, block_size)
    st = w.stages
    assert knh.chain_ugen_count(st) == sum(2 if s.kind == L.STAGE_MUL_CONST else 1 for s in st)
    run_pair(knh, oracle, w, 3)


def test_one_signal_feeding_two_filters_and_an_envelope(knh, oracle):
    """Fan-out: one oscillator through a low-pass and a high-pass side by side, their difference through an EnvAsr, with
    parameter changes at the start of blocks and (WrPreciseTiming on the first filter) in the middle of them."""
    n = 150
    p = configs.voice_parameters(n)
    st = [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_WR_MUL),                 # 1, 2: the oscillator (and its wrapper)
          Stage(L.STAGE_SVF, delayed_changes_per_block=3),               # 3: low-pass of it
          Stage(L.STAGE_ONEPOLE_HPF, input=2),                           # 4: high-pass of the same signal
          Stage(L.STAGE_MATH_SUB, input=3, input2=4),                    # 5
          Stage(L.STAGE_MUL_ENV_ASR),                                    # 6
          Stage(L.STAGE_MATH_ADD, input=6, input2=2),                    # 7: plus the dry oscillator
          Stage(L.STAGE_MUL_CONST)]                                      # 8
    w = configs.Workload("fan", st, n, 64, L.F32, 2)
    w.ctor = {0: p["freq"].reshape(n, 1), 1: np.full((n, 1), 0.5), 2: np.stack([np.zeros(n), p["cutoff"], p["q"], np.zeros(n)], axis=1),
              5: np.tile([0.002, 0.004], (n, 1)), 7: np.full((n, 1), 1.0 / n)}
    v = np.arange(n, dtype=np.uint32)

    def ev(block, bank):
        if block == 0:
            fire_all(bank, n, 5, 3)
            bank.param_apply_many(v, 3, 0, L.VALUE_FLOAT, 300.0 + v)          # the high-pass cutoffs
        if block == 1:
            for voice in range(0, n, 7):
                bank.set_delay_within_block_for_param(voice, 2, 0, 5 + voice % 50)
                bank.param_apply(voice, 2, 0, 500.0 + 10.0 * voice)
        if block == 2:
            fire_all(bank, n, 5, 2)
            bank.param_apply_many(v[::2], 0, 0, L.VALUE_FLOAT, 100.0 + v[::2])
    run_pair(knh, oracle, w, 5, ev)


def random_dag(rng, n_stages):
    """A random feed-forward voice: sources, unary stages on any earlier signal, MathUGens of any two earlier signals."""
    st, ctor = [], {}
    for i in range(n_stages):
        have = len(st)
        r = rng.random()
        pick = lambda: int(rng.integers(1, have + 1))
        if have == 0 or r < 0.25:
            st.append(Stage(L.STAGE_SIN_WT))
            ctor[have] = [float(rng.uniform(50, 3000))]
        elif r < 0.55 and have >= 2:
            kind = [L.STAGE_MATH_MUL, L.STAGE_MATH_ADD, L.STAGE_MATH_SUB][int(rng.integers(0, 3))]
            st.append(Stage(kind, input=pick(), input2=pick()))
        else:
            kind = [L.STAGE_MUL_CONST, L.STAGE_ADD_CONST, L.STAGE_SUB_CONST, L.STAGE_SVF, L.STAGE_ONEPOLE_LPF, L.STAGE_SAFETY_LIMITER,
                    L.STAGE_MUL_ENV_AR][int(rng.integers(0, 7))]
            inp = 0 if rng.random() < 0.5 else pick()
            st.append(Stage(kind, input=inp))
            if kind in (L.STAGE_MUL_CONST, L.STAGE_ADD_CONST, L.STAGE_SUB_CONST):
                ctor[have] = [float(rng.uniform(-1.5, 1.5))]
            elif kind == L.STAGE_SVF:
                ctor[have] = [float(rng.integers(0, 6)), float(rng.uniform(200, 6000)), float(rng.uniform(0.5, 3.0)), 0.0]
            elif kind == L.STAGE_ONEPOLE_LPF:
                ctor[have] = [float(rng.uniform(200, 6000))]
            elif kind == L.STAGE_MUL_ENV_AR:
                ctor[have] = [float(rng.uniform(0.0005, 0.003)), float(rng.uniform(0.001, 0.01))]
    st.append(Stage(L.STAGE_SAFETY_LIMITER))  # keeps a runaway product finite
    return st, ctor


def with_audio_rate_parameters(rng, st):
    """Some parameters of a random voice linked to earlier signals of the voice (knh_stage_desc.ar_param / .input2): the ones
    whose setters are exact -- Constant.value, SinWt freq and phase_offset (negative and huge values saturate, NaN gives 0)."""
    out = []
    for i, x in enumerate(st):
        if i >= 1 and rng.random() < 0.45:
            drv = int(rng.integers(1, i + 1))
            if x.kind in (L.STAGE_MUL_CONST, L.STAGE_ADD_CONST, L.STAGE_SUB_CONST):
                x = Stage(x.kind, x.flags, x.delayed_changes_per_block, x.input, drv, ar_param=1)
            elif x.kind == L.STAGE_SIN_WT:
                x = Stage(x.kind, x.flags, x.delayed_changes_per_block, x.input, drv, ar_param=int(rng.integers(1, 3)))
        out.append(x)
    return out


@pytest.mark.parametrize("seed", range(int(os.environ.get("KNH_TEST_SEEDS", "12"))))
def test_random_dag_voices_with_audio_rate_parameters(knh, oracle, seed):
    rng = np.random.default_rng(3000 + seed)
    st, ctor = random_dag(rng, int(rng.integers(5, 13)))
    st = with_audio_rate_parameters(rng, st)
    assume_some = any(x.ar_param for x in st)
    n = int(rng.integers(1, 140))
    w = configs.Workload(f"dagar{seed}", st, n, int(rng.choice([16, 48, 64, 100])), L.F32 if seed % 3 else L.F64, 1)
    w.ctor = {s: np.tile(np.asarray(a, dtype=np.float64), (n, 1)) * (1.0 + 0.01 * np.arange(n)).reshape(n, 1) if st[s].kind == L.STAGE_SIN_WT
              else np.tile(np.asarray(a, dtype=np.float64), (n, 1)) for s, a in ctor.items()}
    envs = [i for i, s in enumerate(st) if s.kind == L.STAGE_MUL_ENV_AR]
    g = make_gpu(knh, w, L.MIX_LEFT_FOLD)
    o = make_oracle(oracle, w)
    for b in range(4):
        for bank in (g, o):
            if b in (0, 2):
                for e in envs:
                    fire_all(bank, n, e, 2)
            if b == 1:  # ordinary changes: ignored where a signal drives the parameter, applied elsewhere
                for i, x in enumerate(st):
                    if x.kind in (L.STAGE_MUL_CONST, L.STAGE_ADD_CONST, L.STAGE_SUB_CONST):
                        bank.param_apply_many(np.arange(n, dtype=np.uint32), i, 0, L.VALUE_FLOAT, np.full(n, 0.3 + 0.01 * i))
        _, gv, _ = g.process_block_voices()
        _, ov, _, od = o.process_block()
        assert_bit_equal(gv, ov, f"seed {seed} block {b} ({'with' if assume_some else 'without'} audio-rate parameters)")
        np.testing.assert_array_equal(g.read_done_frames(), od)
    g.close()
    o.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("KNH_TEST_SEEDS", "12"))))
def test_random_dag_voices_match_the_oracle(knh, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    st, ctor = random_dag(rng, int(rng.integers(4, 12)))
    n = int(rng.integers(1, 140))
    w = configs.Workload(f"dag{seed}", st, n, int(rng.choice([16, 48, 64, 100])), L.F32 if seed % 3 else L.F64, 1)
    w.ctor = {s: np.tile(np.asarray(a, dtype=np.float64), (n, 1)) * (1.0 + 0.01 * np.arange(n)).reshape(n, 1) if st[s].kind == L.STAGE_SIN_WT
              else np.tile(np.asarray(a, dtype=np.float64), (n, 1)) for s, a in ctor.items()}
    envs = [i for i, s in enumerate(st) if s.kind == L.STAGE_MUL_ENV_AR]

    def ev(block, bank):
        if block in (0, 2):
            for e in envs:
                fire_all(bank, n, e, 2)
    g = make_gpu(knh, w, L.MIX_LEFT_FOLD)
    o = make_oracle(oracle, w)
    for b in range(4):
        ev(b, g)
        ev(b, o)
        _, gv, _ = g.process_block_voices()
        _, ov, _, od = o.process_block()
        assert_bit_equal(gv, ov, f"seed {seed} block {b}")
        np.testing.assert_array_equal(g.read_done_frames(), od)
    g.close()
    o.close()


@pytest.mark.parametrize("seed,n_stages", [(0, 20), (1, 33)])
def test_larger_random_dag_voices(knh, oracle, seed, n_stages):
    """Voices of more than sixteen stages keep the whole-chain kernel's eight-sample visits (hiprtc's time grows with the visit
    length times the stage count): the same bits."""
    rng = np.random.default_rng(5000 + seed)
    st, ctor = random_dag(rng, n_stages)
    n = 70
    w = configs.Workload(f"bigdag{seed}", st, n, 48, L.F32, 1)
    w.ctor = {s: np.tile(np.asarray(a, dtype=np.float64), (n, 1)) * (1.0 + 0.01 * np.arange(n)).reshape(n, 1) if st[s].kind == L.STAGE_SIN_WT
              else np.tile(np.asarray(a, dtype=np.float64), (n, 1)) for s, a in ctor.items()}
    envs = [i for i, s in enumerate(st) if s.kind == L.STAGE_MUL_ENV_AR]
    g = make_gpu(knh, w, L.MIX_LEFT_FOLD)
    o = make_oracle(oracle, w)
    for b in range(3):
        for bank in (g, o):
            if b in (0, 2):
                for e in envs:
                    fire_all(bank, n, e, 2)
        _, gv, _ = g.process_block_voices()
        _, ov, _, od = o.process_block()
        assert_bit_equal(gv, ov, f"{n_stages}-stage voice, block {b}")
        np.testing.assert_array_equal(g.read_done_frames(), od)
    g.close()
    o.close()


def test_division_and_power_of_two_signals(knh, oracle):
    n = 64
    st = [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_SIN_WT), Stage(L.STAGE_ADD_CONST),          # 3: b + 2 (never near zero)
          Stage(L.STAGE_MATH_DIV, input=1, input2=3),                                        # 4: a / (b + 2): exact
          Stage(L.STAGE_MATH_POW, input=3, input2=1)]                                        # 5: (b + 2) ^ a: device pow
    w = configs.Workload("divpow", st, n, 64, L.F32, 1)
    w.ctor = {0: np.linspace(100, 900, n).reshape(n, 1), 1: np.linspace(50, 450, n).reshape(n, 1), 2: np.full((n, 1), 2.0)}
    run_pair(knh, oracle, w, 2, tol=2e-5)
    w.stages = st[:4]
    run_pair(knh, oracle, w, 2)


def test_graph_voices_have_a_stage_limit(knh):
    """Every stage of a FUSED graph-shaped voice unrolls into its kernel: the build time grows faster than the stage count, so
    the library refuses more than 512 stages instead of hanging -- unless the voice is made of SinWt oscillators and arithmetic
    only (the interpreter below takes up to 4 096 of those)."""
    st = [Stage(L.STAGE_SIN_WT)]
    for i in range(300):
        st += [Stage(L.STAGE_ONEPOLE_LPF), Stage(L.STAGE_MUL_CONST)]
    st += [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_MATH_ADD, input=len(st), input2=len(st) + 1)]  # (a second source: a graph)
    with pytest.raises(L.KnasterHipError) as e:
        knh.VoiceBank(st, 1, L.F32, 1)
    assert e.value.status == L.ERR_UNSUPPORTED_CHAIN
    w = configs.fm_cascade(700, 1, 128)
    assert len(w.stages) > 4096
    with pytest.raises(L.KnasterHipError) as e:
        knh.VoiceBank(w.stages, 1, L.F32, 1)
    assert e.value.status == L.ERR_UNSUPPORTED_CHAIN


@pytest.mark.parametrize("frame_jit", ["1", "0"])
@pytest.mark.parametrize("n_voices,block_size,sample_type", [(1, 128, L.F32), (3, 32, L.F32), (2, 16, L.F64)])
def test_the_reference_256_oscillator_fm_cascade(knh, oracle, monkeypatch, n_voices, block_size, sample_type, frame_jit):
    """knaster_benchmarks/benches/graph_dsp_performance.rs:37-72 as written: 256 SinWt and 1 275 math nodes in ONE voice (1 531
    stages).  Run a lane per frame -- as one straight-line kernel hiprtc builds from the stage list (voice_frame.hpp; the
    default) or by the interpreter (kernels_interp.hip; KNH_FRAME_JIT=0) -- bit-identical to the oracle's node-by-node graph."""
    monkeypatch.setenv("KNH_FRAME_JIT", frame_jit)
    w = configs.fm_cascade(256, n_voices, block_size, sample_type, add=19.0)  # (s + 19) * 0.05: the signal stays of order one
    assert len(w.stages) == 1531
    run_pair(knh, oracle, w, 3)
    # the reference's own constants: the signal is infinite, then NaN, from the 30th oscillator on -- the same infinities and
    # NaNs, bit for bit, on both sides
    w = configs.fm_cascade(256, n_voices, block_size, sample_type)
    g = make_gpu(knh, w, L.MIX_LEFT_FOLD)
    o = make_oracle(oracle, w)
    for b in range(2):
        _, gv, _ = g.process_block_voices()
        _, ov, _, _ = o.process_block()
        assert_bit_equal(gv, ov, f"reference constants, block {b}")
    g.close()
    o.close()


def arithmetic_dag(rng, n_stages):
    """A random feed-forward voice of what the interpreter runs: SinWt sources, x (op) value, a (op) b."""
    st, ctor = [], {}
    for i in range(n_stages):
        have = len(st)
        r = rng.random()
        pick = lambda: int(rng.integers(1, have + 1))
        if have == 0 or r < 0.3:
            st.append(Stage(L.STAGE_SIN_WT))
            ctor[have] = [float(rng.uniform(50, 3000))]
        elif r < 0.6 and have >= 2:
            kind = [L.STAGE_MATH_MUL, L.STAGE_MATH_ADD, L.STAGE_MATH_SUB][int(rng.integers(0, 3))]
            st.append(Stage(kind, input=pick(), input2=pick()))
        else:
            kind = [L.STAGE_MUL_CONST, L.STAGE_ADD_CONST, L.STAGE_SUB_CONST, L.STAGE_WR_MUL, L.STAGE_WR_ADD][int(rng.integers(0, 5))]
            wrapper = kind in (L.STAGE_WR_MUL, L.STAGE_WR_ADD)
            st.append(Stage(kind, input=0 if wrapper or rng.random() < 0.5 else pick()))
            ctor[have] = [float(rng.uniform(-1.2, 1.2))]
    return st, ctor


@pytest.mark.parametrize("seed", range(int(os.environ.get("KNH_TEST_SEEDS", "8"))))
def test_interpreter_equals_the_fused_kernel_and_the_oracle(knh, oracle, monkeypatch, seed):
    """KNH_INTERP=1 sends every graph-shaped voice the interpreter can run to it: random graphs, several blocks per launch,
    parameter changes at block starts (a new frequency, a new constant), both sample types, both mix orders."""
    rng = np.random.default_rng(4000 + seed)
    st, ctor = arithmetic_dag(rng, int(rng.integers(5, 40)))
    n = int(rng.integers(1, 70))
    bs = int(rng.choice([16, 48, 64, 100, 256]))
    w = configs.Workload(f"interp{seed}", st, n, bs, L.F32 if seed % 3 else L.F64, 1)
    w.ctor = {s: np.tile(np.asarray(a, dtype=np.float64), (n, 1)) * (1.0 + 0.01 * np.arange(n)).reshape(n, 1) for s, a in ctor.items()}
    if not any(x.kind in (L.STAGE_MATH_MUL, L.STAGE_MATH_ADD, L.STAGE_MATH_SUB) for x in st) and sum(x.kind == L.STAGE_SIN_WT for x in st) < 2:
        st.append(Stage(L.STAGE_SIN_WT))  # (make it a graph, not a chain)
        w.ctor[len(st) - 1] = np.full((n, 1), 333.0)
        st.append(Stage(L.STAGE_MATH_ADD, input=len(st), input2=len(st) - 1))
    sines = [i for i, x in enumerate(st) if x.kind == L.STAGE_SIN_WT]
    consts = [i for i, x in enumerate(st) if x.kind in (L.STAGE_MUL_CONST, L.STAGE_ADD_CONST, L.STAGE_SUB_CONST, L.STAGE_WR_MUL)]  # (WrAdd has no parameter)
    v = np.arange(n, dtype=np.uint32)

    def ev(block, bank):
        if block == 1:
            bank.param_apply_many(v[::2], sines[0], 0, L.VALUE_FLOAT, 200.0 + 3.0 * v[::2])
        if block == 2 and consts:
            bank.param_apply_many(v, consts[-1], 0, L.VALUE_FLOAT, np.linspace(-0.5, 0.5, n))
    outs = {}
    for form in ("fused", "interp", "frame"):
        monkeypatch.setenv("KNH_INTERP", "0" if form == "fused" else "1")
        monkeypatch.setenv("KNH_FRAME_JIT", "0" if form == "interp" else "1")
        g = make_gpu(knh, w, L.MIX_LEFT_FOLD)
        res = []
        for b in range(4):
            ev(b, g)
            res.append(g.process_block_voices()[:2])
        outs[form] = res
        g.close()
    o = make_oracle(oracle, w)
    for b in range(4):
        ev(b, o)
        o_out, o_voices, _, _ = o.process_block()
        for form in ("fused", "interp", "frame"):
            assert_bit_equal(outs[form][b][1], o_voices, f"seed {seed} {form} block {b} per-voice")
            assert_bit_equal(outs[form][b][0], o_out, f"seed {seed} {form} block {b} mix")
    o.close()
    # several blocks per launch, tree mix: each frame-parallel form against itself block by block
    monkeypatch.setenv("KNH_INTERP", "1")
    for fj in ("0", "1"):
        monkeypatch.setenv("KNH_FRAME_JIT", fj)
        a, b2 = make_gpu(knh, w), make_gpu(knh, w)
        one = []
        for blk in range(5):  # the same changes, given block by block and scheduled ahead for blocks 1 and 2 of one launch
            ev(blk, a)
            one.append(a.process_block()[0])
        one = np.stack(one)
        b2.param_apply_many(v[::2], sines[0], 0, L.VALUE_FLOAT, 200.0 + 3.0 * v[::2], block_offset=1)
        if consts:
            b2.param_apply_many(v, consts[-1], 0, L.VALUE_FLOAT, np.linspace(-0.5, 0.5, n), block_offset=2)
        many = b2.process_blocks(5)[0]
        assert_bit_equal(many, one, f"seed {seed}: 5 blocks in one launch (KNH_FRAME_JIT={fj})")
        a.close()
        b2.close()


def test_frame_parallel_voices_in_sharded_and_rank_banks(knh):
    """A bank of frame-parallel voices cut into host shards, spread over "devices" (all device 0 here) and as a one-rank rank
    bank: the same voices, mixed within the tolerance of the re-association (each range has a pairwise tree of its own)."""
    w = configs.fm_cascade(12, 300, 64, add=19.0)
    ref = make_gpu(knh, w)
    want = np.stack([ref.process_block()[0] for _ in range(3)])
    ref.close()
    assert np.abs(want).max() > 1e-3
    for kw in ({"host_threads": 3}, {"devices": [0, 0]}, {"rank": 0, "world": 1}):
        b = knh.VoiceBank(w.stages, w.n_voices, w.sample_type, w.out_channels, L.MIX_TREE, -1, False, **kw)
        for s, a in w.ctor.items():
            b.set_ctor_args(s, a)
        b.init(configs.SAMPLE_RATE, w.block_size)
        got = np.stack([b.process_block()[0] for _ in range(3)])
        b.close()
        assert np.max(np.abs(got - want)) <= 1e-5 * max(1.0, float(np.abs(want).max())), kw
