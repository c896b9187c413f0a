"""The reference's own known-answer vectors, fed straight through the C ABI to the HIP kernels.

Every expected value below is typed from the reference's test or bench source (file:line cited per test), not taken from
the oracle: these are the only numbers the reference itself holds for the path, and here they reach the device code
without the oracle in between.

The reference's test UGens are expressed with the bank's stages:
  TestNumUGen::new(c)        (knaster_core_dsp/src/test_utils.rs:8-41)   = SinWt(0 Hz) [a zero source: phase 0, step 0,
                             table[0] = sin(0) = +0.0] followed by ADD_CONST(c)
  TestInPlusParamGen::new()  (test_utils.rs:44-88: out = number + in, param 0 "number", in = 0)
                             = SinWt(0 Hz) followed by ADD_CONST(0), whose parameter 0 is the number
"""
import numpy as np
import pytest

from knaster_amd import _lib as L
from knaster_amd.bank import TRIGGER, Stage

pytestmark = pytest.mark.gpu

F32_EPS = float(np.finfo(np.float32).eps)


def bank_of(knh, stages, ctor, n_voices=1, block_size=4, sample_type=L.F32, out_channels=1, mix=L.MIX_LEFT_FOLD):
    b = knh.VoiceBank(stages, n_voices, sample_type, out_channels, mix)
    for s, a in ctor.items():
        b.set_ctor_args(s, np.tile(np.asarray(a, dtype=np.float64).reshape(1, -1), (n_voices, 1)))
    b.init(48000, block_size)
    return b


def first_sample(knh, wrapper_kind, number, value, sample_type):
    """TestNumUGen::new(number).<wrapper>(value).process(..)[0]"""
    b = bank_of(knh, [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_ADD_CONST), Stage(wrapper_kind)], {0: [0.0], 1: [number], 2: [value]},
                sample_type=sample_type)
    out, _ = b.process_block()
    b.close()
    assert np.all(out[0] == out[0, 0])  # a constant source
    return float(out[0, 0])


@pytest.mark.parametrize("sample_type", [L.F32, L.F64])
def test_wrapper_arithmetic(knh, sample_type):
    """knaster_core_dsp/src/wrappers_core.rs:124-164 (`wrapper_arithmetic`; the reference runs it with F = f64)."""
    assert first_sample(knh, L.STAGE_WR_ADD, 2.5, 2.5, sample_type) == 5.0
    assert first_sample(knh, L.STAGE_WR_MUL, 2.5, 3.0, sample_type) == 7.5
    assert first_sample(knh, L.STAGE_WR_DIV, 2.5, 5.0, sample_type) == 0.5
    assert first_sample(knh, L.STAGE_WR_VDIV, 2.5, 5.0, sample_type) == 2.0
    assert first_sample(knh, L.STAGE_WR_SUB, 6.0, 7.0, sample_type) == -1.0
    assert first_sample(knh, L.STAGE_WR_VSUB, 6.0, 7.0, sample_type) == 1.0
    # :141-147: |6.powf(2) - 36| < f32::EPSILON * 10 with f64 samples.  An f32 bank can only promise its own last place:
    # 36 has an ulp of 3.8e-6 in f32, so there the bound is two ulps of the result.
    powf = first_sample(knh, L.STAGE_WR_POWF, 6.0, 2.0, sample_type)
    assert abs(powf - 36.0) < (F32_EPS * 10.0 if sample_type == L.F64 else 2 * 3.8146973e-06)
    # :149-160: approx_eq!(f64, 6.powi(2), 36., epsilon = f32::EPSILON * 2, ulps = 2); the multiply-by-squaring loop is exact here
    assert first_sample(knh, L.STAGE_WR_POWI, 6.0, 2.0, sample_type) == 36.0
    # :161-162 `.wr(|s| s * 2.0 + 1.0)` is a host closure (WrClosure): opaque code cannot be fused, out of scope


EXPECTED_SAMPLE_ACCURATE = [0., 0., 0., 0., 0., 5., 6., 6., 8., 9., 10., 10., 10., 10., 10., 10.]


def schedule_five_changes(b, stage):
    # wrappers_core.rs:177-186 / :221-230
    for delay, value in ((5, 5.0), (6, 6.0), (8, 8.0), (9, 9.0), (10, 10.0)):
        b.set_delay_within_block_for_param(0, stage, 0, delay)
        b.param(0, stage, 0, value)


@pytest.mark.parametrize("n_voices", [1, 70])
def test_sample_accurate_parameters(knh, n_voices):
    """wrappers_core.rs:167-200: WrPreciseTiming::<10, _>::new(TestInPlusParamGen::new()), block 16, five delayed changes."""
    b = bank_of(knh, [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_ADD_CONST, delayed_changes_per_block=10)], {0: [0.0], 1: [0.0]},
                n_voices=n_voices, block_size=16)
    schedule_five_changes(b, 1)
    out, voices, _ = b.process_block_voices()
    assert voices[0].tolist() == EXPECTED_SAMPLE_ACCURATE
    for v in range(1, n_voices):  # the other voices received nothing
        assert not voices[v].any()
    assert out[0].tolist() == EXPECTED_SAMPLE_ACCURATE  # the left fold over one changing voice and silent ones
    # the queue is empty again: the next block holds the last value (precise_timing.rs:112 resets next_delay_i)
    out, _ = b.process_block()
    assert out[0].tolist() == [10.0] * 16
    b.close()


def test_sample_accurate_parameters_with_wrappers(knh):
    """wrappers_core.rs:202-250: the same under .wr_add(0).wr_sub(0).wr_div(1).wr_mul(1).wr_powf(1).wr_powi(1) -- every math
    wrapper must pass set_delay_within_block_for_param through to the WrPreciseTiming inside.  (The seventh, `.wr(|v| v)`,
    is a host closure: not expressible in a fused chain, and the identity.)  approx_eq!(f32, .., epsilon = 0.0002, ulps = 5)."""
    stages = [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_ADD_CONST, delayed_changes_per_block=10), Stage(L.STAGE_WR_ADD), Stage(L.STAGE_WR_SUB),
              Stage(L.STAGE_WR_DIV), Stage(L.STAGE_WR_MUL), Stage(L.STAGE_WR_POWF), Stage(L.STAGE_WR_POWI)]
    b = bank_of(knh, stages, {0: [0.0], 1: [0.0], 2: [0.0], 3: [0.0], 4: [1.0], 5: [1.0], 6: [1.0], 7: [1.0]}, block_size=16)
    schedule_five_changes(b, 1)
    out, _ = b.process_block()
    assert np.all(np.abs(out[0].astype(np.float64) - np.array(EXPECTED_SAMPLE_ACCURATE)) <= 0.0002)
    b.close()


@pytest.mark.parametrize("as_node", [False, True])
@pytest.mark.parametrize("n_voices", [1, 100])
def test_wrappers_vs_nodes_bench_asserts(knh, as_node, n_voices):
    """knaster_benchmarks/benches/wrappers_vs_nodes.rs:18-111: TestNumUGen(2.0).wr_mul(0.5) (:19, :67-70) or
    TestNumUGen(2.0) * TestNumUGen(0.5) (:38-41, :96-100) to graph output 0, once or 100 times additively;
    `output_block().channel_as_slice_mut(0)[block_size - 1]` is 1.0 (:25-28, :51-54) / 100.0 (:79-82, :108-111), block 32.
    The 100 additive connections are the reference's left fold of Add nodes: KNH_MIX_LEFT_FOLD."""
    block_size = 32
    last = Stage(L.STAGE_MUL_CONST) if as_node else Stage(L.STAGE_WR_MUL)
    b = bank_of(knh, [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_ADD_CONST), last], {0: [0.0], 1: [2.0], 2: [0.5]}, n_voices=n_voices,
                block_size=block_size)
    for _ in range(3):  # b.iter: run_without_inputs again and again
        out, _ = b.process_block()
        assert out[0, block_size - 1] == (100.0 if n_voices == 100 else 1.0)
    b.close()
    # the tree mix (the default) gives the same here: every partial sum of ones is exact
    b = bank_of(knh, [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_ADD_CONST), last], {0: [0.0], 1: [2.0], 2: [0.5]}, n_voices=n_voices,
                block_size=block_size, mix=L.MIX_TREE)
    out, _ = b.process_block()
    assert out[0, block_size - 1] == (100.0 if n_voices == 100 else 1.0)
    b.close()


def test_free_node_when_done(knh):
    """knaster_graph/src/graph.rs:2483-2513: EnvAsr::new(0.0, 0.0) with Done::FreeSelf, attack_time 0, release_time 0,
    t_restart and t_release set before the first block; after running, the node has freed itself -- i.e. the envelope
    called flags.mark_done.  Through the ABI: the bank reports the done flag (and its frame), then that every voice has
    stopped.  (Attacking -> t_release: release_scale = t = 0, t = 1; the first sample outputs 1^3 * 0 = 0 and t - 1 <= 0
    stops the envelope at frame 0, envelopes.rs:66-78,113-128.)"""
    n = 3
    b = bank_of(knh, [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_MUL_ENV_ASR)], {0: [440.0], 1: [0.0, 0.0]}, n_voices=n, block_size=16,
                out_channels=2)
    for v in range(n):
        b.param(v, 1, "attack_time", 0.0)
        b.param(v, 1, "release_time", 0.0)
        b.param(v, 1, "t_restart", TRIGGER)
        b.param(v, 1, "t_release", TRIGGER)
    out, flags = b.process_block()
    assert flags & L.FLAG_ANY_DONE
    assert b.read_done_frames().tolist() == [0] * n
    assert not out.any()
    for _ in range(9):  # graph.rs:2503-2505: ten runs in all
        out, flags = b.process_block()
        assert not (flags & L.FLAG_ANY_DONE)
        assert flags & L.FLAG_ALL_DONE
        assert not out.any()
    b.close()


def test_implement_a_gen_sine(knh):
    """knaster_core/examples/implement_a_gen.rs:14-35: a naive sine oscillator at 200 Hz / 48 kHz (the arithmetic of
    SinNumeric, osc.rs:263-270): the first frame is 0.0; after it, sample 63 of a 64-frame block is
    sin((200 / 48000) * TAU * 64) within f32::EPSILON."""
    b = bank_of(knh, [Stage(L.STAGE_SIN_NUMERIC)], {0: [200.0]}, block_size=64)
    out, _ = b.process_block(frames_to_process=1)  # osc.process(..): one frame
    assert out[0, 0] == 0.0
    out, _ = b.process_block()
    want = np.sin(np.float32(np.float32(np.float32(200.0) / np.float32(48000.0)) * np.float32(2 * np.pi)) * np.float32(64.0), dtype=np.float32)
    assert abs(float(out[0, 63]) - float(want)) < F32_EPS
    b.close()
