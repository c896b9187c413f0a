"""GPU tests at BASELINE.json's full sizes, through size-independent properties, plus the equivalences
between the library's own execution forms (single-wave vs wave-pipelined kernel, one launch per
block vs many blocks per launch, whole vs split blocks).  All comparisons are bit-exact."""
import numpy as np
import pytest

from helpers import assert_bit_equal, fire_all, make_gpu, make_oracle
from knaster_amd import _lib as L
from knaster_amd import configs

pytestmark = pytest.mark.gpu


def left_fold(rows: np.ndarray) -> np.ndarray:
    acc = rows[0].copy()
    for r in rows[1:]:
        acc = acc + r
    return acc


def tree_mix(voices: np.ndarray) -> np.ndarray:
    """The documented KNH_MIX_TREE order: the pairwise sum over the voice index (helpers.pairwise_sum)."""
    from helpers import pairwise_sum
    return pairwise_sum(voices)


def c3_script(w, block, bank, offset=0):
    v = np.arange(w.n_voices, dtype=np.uint32)
    if block == 0:
        bank.param_apply_many(v, w.restart[0], w.restart[1], L.VALUE_TRIGGER, block_offset=offset)
        if w.delay_times is not None:
            bank.param_apply_many(v, 3, 0, L.VALUE_FLOAT, w.delay_times, block_offset=offset)
    if block == 2 and w.delay_times is not None:  # delays shorter than a tile on some voices: the per-sample path
        bank.param_apply_many(v[::5], 3, 0, L.VALUE_FLOAT, (v[::5] % 40) / 48000.0 + 1e-6, block_offset=offset)
    if block == 3:
        bank.param_apply_many(v[::3], w.release[0], w.release[1], L.VALUE_TRIGGER, block_offset=offset)
    if block == 4:  # sample-accurate cutoff changes on half the voices (stage 2 is not wrapped -> immediate)
        p = configs.voice_parameters(w.n_voices)
        bank.param_apply_many(v[1::2], 2, 0, L.VALUE_FLOAT, p["cutoff"][1::2] * 0.5, None, None, block_offset=offset)


@pytest.mark.parametrize("pipeline", ["0", "1", "2"])
def test_full_size_c3_mix_orders_and_oracle_subset(knh, oracle, monkeypatch, pipeline):
    """16384 voices x 512 frames: (a) the tree mix equals the documented fold of the per-voice signals,
    (b) the left-fold mix equals the serial fold, (c) 96 sampled voices equal the oracle run on just
    those voices (voices are independent), all bit for bit."""
    monkeypatch.setenv("KNH_PIPELINE", pipeline)
    w = configs.config("C3")
    g_tree = make_gpu(knh, w, L.MIX_TREE)
    g_fold = make_gpu(knh, w, L.MIX_LEFT_FOLD)
    pick = np.unique(np.concatenate([np.arange(0, 32), np.arange(8191, 8223), np.arange(w.n_voices - 32, w.n_voices)]))
    sub = configs.Workload("sub", w.stages, len(pick), w.block_size, w.sample_type, w.out_channels,
                           {s: a[pick] for s, a in w.ctor.items()})
    o = make_oracle(oracle, sub, want_mix=False)
    for block in range(3):
        for bank, n in ((g_tree, w.n_voices), (g_fold, w.n_voices), (o, len(pick))):
            if block == 0:
                fire_all(bank, n, *w.restart)
            if block == 2:
                fire_all(bank, n, w.release[0], w.release[1])
        out_t, voices, _ = g_tree.process_block_voices()
        out_f, voices_f, _ = g_fold.process_block_voices()
        _, o_voices, _, _ = o.process_block()
        assert_bit_equal(voices, voices_f, "tree vs fold banks per-voice")
        assert_bit_equal(voices[pick], o_voices, f"block {block}: sampled voices vs oracle")
        assert_bit_equal(out_t[0], tree_mix(voices), f"block {block}: tree mix")
        assert_bit_equal(out_f[0], left_fold(voices), f"block {block}: left-fold mix")
        assert_bit_equal(out_t[0], out_t[1], "L == R")
        ref = voices.astype(np.float64).sum(axis=0)
        assert np.max(np.abs(out_t[0] - ref)) <= 1e-5 and np.max(np.abs(out_f[0] - ref)) <= 1e-5
        assert np.max(np.abs(ref)) > 1e-4  # not silence


def test_full_size_c2_sin_numeric(knh, oracle):
    """BASELINE.json config C2 at its full size: 1 024 voices of SinNumeric * gain, block 256, the kernel form the bench runs
    (phase | sin on eight wavefronts | mixer).  Every voice against the oracle within 1e-5 of its gain -- north_star's tolerance (the node's sin
    is the hardware's v_sin_f32, measured 8.7e-7 of full scale from the reference over every phase of (-2, 2),
    profiles/r03_micro_hw_sin.txt; the phase accumulator is exact, so the error does not grow over the blocks), the mix within 1e-5, and
    equal to the documented tree fold of the device's own per-voice signals bit for bit."""
    w = configs.config("C2")
    assert w.n_voices == 1024 and w.block_size == 256
    g = make_gpu(knh, w, L.MIX_TREE)
    o = make_oracle(oracle, w, want_mix=False)
    gain = 1.0 / w.n_voices
    for block in range(6):
        if block == 3:  # new frequencies on a third of the voices, on both
            v = np.arange(0, w.n_voices, 3, dtype=np.uint32)
            for bank in (g, o):
                bank.param_apply_many(v, 0, 0, L.VALUE_FLOAT, 110.0 + 0.37 * v)
        out, voices, _ = g.process_block_voices()
        _, o_voices, _, _ = o.process_block()
        assert np.max(np.abs(voices.astype(np.float64) - o_voices.astype(np.float64))) <= 1e-5 * gain, f"block {block}"
        assert_bit_equal(out[0], tree_mix(voices), f"block {block}: tree mix")
        assert np.max(np.abs(out[0].astype(np.float64) - o_voices.astype(np.float64).sum(axis=0))) <= 1e-5
        assert np.abs(o_voices).max() > 0.5 * gain
    g.close()
    o.close()


def test_full_size_c5_fm_with_sample_accurate_changes(knh, oracle):
    """BASELINE.json config C5 at its full size: 4 096 voices of audio-rate FM, block 128, every voice a delayed change every
    second block (resolved on the device: kernels_events.hip).  160 sampled voices -- whole wavefront groups at both ends and in
    the middle -- equal the oracle run on just those voices with just their changes, bit for bit, over 8 blocks given one by
    one; the same 8 blocks scheduled ahead in one launch give the same mix."""
    w = configs.config("C5")
    assert w.n_voices == 4096 and w.block_size == 128
    pick = np.unique(np.concatenate([np.arange(0, 64), np.arange(2040, 2072), np.arange(w.n_voices - 64, w.n_voices)]))
    sub = configs.Workload("sub", w.stages, len(pick), w.block_size, w.sample_type, w.out_channels, {s: a[pick] for s, a in w.ctor.items()})
    local = {int(v): i for i, v in enumerate(pick)}
    g = make_gpu(knh, w, L.MIX_TREE)
    g2 = make_gpu(knh, w, L.MIX_TREE)
    o = make_oracle(oracle, sub, want_mix=False)
    mixes = []
    for block in range(8):
        e = configs.c5_events(w, block)
        if e is not None:
            g.param_apply_many(e[0], e[1], e[2], e[3], e[4], None, e[5])
            g2.param_apply_many(e[0], e[1], e[2], e[3], e[4], None, e[5], block_offset=block)
            m = np.isin(e[0], pick)
            o.param_apply_many(np.array([local[int(v)] for v in e[0][m]], dtype=np.uint32), e[1][m], e[2][m], e[3][m], e[4][m], None, e[5][m])
        out, voices, _ = g.process_block_voices()
        _, o_voices, _, _ = o.process_block()
        assert_bit_equal(voices[pick], o_voices, f"block {block}: sampled voices vs oracle")
        assert_bit_equal(out[0], tree_mix(voices), f"block {block}: tree mix")
        mixes.append(out)
        assert np.abs(o_voices).max() > 1e-5
    ahead, _ = g2.process_blocks(8)
    assert_bit_equal(ahead, np.stack(mixes), "eight blocks scheduled ahead in one launch")
    for bank in (g, g2, o):
        bank.close()


@pytest.mark.parametrize("name,n_voices,block_size", [("C3", 1000, 512), ("C4", 300, 100), ("C5", 260, 128), ("C2", 200, 48),
                                                      ("D3", 500, 512), ("B3", 300, 256)])
def test_pipelined_kernel_equals_single_wave_kernel(knh, monkeypatch, name, n_voices, block_size):
    """KNH_PIPELINE 0 = one wavefront per 64 voices, 1 = linear wave pipeline, 2 = five-role pipeline where built;
    "1big" = the linear pipeline with 64-sample tiles and the fold in its last stage group (KNH_PIPE_BIG=1, where built);
    "1inplace" = 64-sample tiles, the last stage group works in place and a mixer wavefront folds (the default where built)."""
    w = configs.config(name, n_voices=n_voices, block_size=block_size)
    outs = {}
    for pipeline in ("0", "1", "2", "1big", "1inplace"):
        monkeypatch.setenv("KNH_PIPELINE", pipeline[0])
        monkeypatch.setenv("KNH_PIPE_BIG", "1" if pipeline.endswith("big") else ("2" if pipeline.endswith("inplace") else "0"))
        g = make_gpu(knh, w)
        res = []
        for block in range(6):
            if name in ("C3", "C4", "D3", "B3"):
                c3_script(w, block, g)
            if name == "C5":
                e = configs.c5_events(w, block)
                if e is not None:
                    g.param_apply_many(e[0], e[1], e[2], e[3], e[4], None, e[5])
            out, voices, flags = g.process_block_voices()
            res.append((out, voices, flags, g.read_done_frames()))
        outs[pipeline] = res
        g.close()
    for other in ("1", "2", "1big", "1inplace"):
        for (o0, v0, f0, d0), (o1, v1, f1, d1) in zip(outs["0"], outs[other]):
            assert_bit_equal(v0, v1, f"per-voice, pipeline {other}")
            assert_bit_equal(o0, o1, f"mix, pipeline {other}")
            assert f0 == f1 and np.array_equal(d0, d1)


@pytest.mark.parametrize("name,n_voices,block_size", [("C3", 1000, 512), ("C3", 129, 96), ("C4", 300, 100), ("C4", 449, 64)])
def test_two_voice_groups_per_workgroup_equal_the_single_wave_kernel(knh, monkeypatch, name, n_voices, block_size):
    """The pipeline with two 64-voice groups per workgroup (banks of more groups than the chip has CUs; KNH_PAIR=1 forces it
    for a small bank): per-voice signals, mix, flags and done frames are those of the one-wavefront-per-group kernel -- with
    an even and an odd number of groups (the last workgroup then holds one live group and one that only keeps the barriers
    company), a ragged last group, blocks that are not whole tiles, single blocks and several blocks per launch."""
    w = configs.config(name, n_voices=n_voices, block_size=block_size)
    outs = {}
    for form in ("single", "pair"):
        monkeypatch.setenv("KNH_PIPELINE", "0" if form == "single" else "1")
        monkeypatch.setenv("KNH_PAIR", "1" if form == "pair" else "0")
        g = make_gpu(knh, w)
        res = []
        for block in range(6):
            c3_script(w, block, g)
            out, voices, flags = g.process_block_voices()
            res.append((out, voices, flags, g.read_done_frames()))
        for block in range(6, 12):
            c3_script(w, block - 6, g, block - 6)
        many, _ = g.process_blocks(6)
        outs[form] = (res, many)
        g.close()
    for (o0, v0, f0, d0), (o1, v1, f1, d1) in zip(outs["single"][0], outs["pair"][0]):
        assert_bit_equal(v0, v1, "per-voice")
        assert_bit_equal(o0, o1, "mix")
        assert f0 == f1 and np.array_equal(d0, d1)
    assert_bit_equal(outs["single"][1], outs["pair"][1], "six blocks in one launch")
    assert np.abs(outs["pair"][1]).max() > 1e-6


@pytest.mark.parametrize("pipeline", ["0", "1", "2"])
@pytest.mark.parametrize("name,n_voices,block_size", [("C3", 700, 512), ("C3", 130, 100), ("C5", 200, 128)])
def test_many_blocks_per_launch_equals_block_by_block(knh, monkeypatch, pipeline, name, n_voices, block_size):
    monkeypatch.setenv("KNH_PIPELINE", pipeline)
    w = configs.config(name, n_voices=n_voices, block_size=block_size)
    n_blocks = 7

    def script(block, bank, offset):
        if name == "C3":
            c3_script(w, block, bank, offset)
        else:
            e = configs.c5_events(w, block)
            if e is not None:
                bank.param_apply_many(e[0], e[1], e[2], e[3], e[4], None, e[5], block_offset=offset)
    a = make_gpu(knh, w)
    single = []
    for block in range(n_blocks):
        script(block, a, 0)
        single.append(a.process_block()[0])
    b = make_gpu(knh, w)
    for block in range(n_blocks):
        script(block, b, block)
    multi, _ = b.process_blocks(n_blocks)
    for block in range(n_blocks):
        assert_bit_equal(multi[block], single[block], f"block {block}")
    # and the state left behind is the same: one more block from each
    assert_bit_equal(a.process_block_voices()[1], b.process_block_voices()[1], "state after the launch")
    # two launches of 3 + 4 blocks == one launch of 7
    c = make_gpu(knh, w)
    for block in range(n_blocks):
        script(block, c, block)   # offsets beyond the first launch carry over to the second
    first, _ = c.process_blocks(3)
    second, _ = c.process_blocks(4)
    assert_bit_equal(np.concatenate([first, second]), multi, "3 + 4 blocks")


@pytest.mark.parametrize("pipeline", ["0", "1", "2"])
def test_split_block_equals_whole_block(knh, monkeypatch, pipeline):
    """ctx.block partial processing (BlockMetadata::make_partial, ugen.rs:87-93): frames [0,k) then [k,B)."""
    monkeypatch.setenv("KNH_PIPELINE", pipeline)
    w = configs.config("C3", n_voices=200, block_size=256)
    whole, parts = make_gpu(knh, w), make_gpu(knh, w)
    for bank in (whole, parts):
        fire_all(bank, w.n_voices, *w.restart)
    for k in (1, 37, 64, 255):
        ref, _ = whole.process_block()
        out = np.zeros_like(ref)
        parts.process_block(frames_to_process=k, block_start_offset=0, out=out)
        parts.process_block(frames_to_process=w.block_size - k, block_start_offset=k, out=out)
        assert_bit_equal(out, ref, f"split at {k}")


def test_runs_are_deterministic(knh):
    w = configs.config("C3", n_voices=4096, block_size=512)
    outs = []
    for _ in range(2):
        g = make_gpu(knh, w)
        fire_all(g, w.n_voices, *w.restart)
        outs.append(g.process_blocks(4)[0])
        g.close()
    assert_bit_equal(outs[0], outs[1], "two runs")


def test_mix_is_linear_in_the_voice_set(knh):
    """Voices are independent: a bank of voices A+B mixes to (mix A) + (mix B) up to f32 reassociation, and exactly when
    A is a power of two of voices and B no more than that (they are the two subtrees under the root of the pairwise sum)."""
    w = configs.config("C3", n_voices=2048, block_size=128)
    full = make_gpu(knh, w)
    halves = []
    for lo in (0, 1024):
        h = configs.Workload("h", w.stages, 1024, w.block_size, w.sample_type, 2, {s: a[lo:lo + 1024] for s, a in w.ctor.items()})
        halves.append(make_gpu(knh, h))
    for bank, n in [(full, 2048), (halves[0], 1024), (halves[1], 1024)]:
        fire_all(bank, n, *w.restart)
    for _ in range(3):
        f = full.process_block()[0]
        a = halves[0].process_block()[0]
        b = halves[1].process_block()[0]
        assert_bit_equal(f, a + b, "mix(A u B) == mix(A) + mix(B)")


def test_runtime_fused_chain_matches_oracle(knh, oracle):
    """A chain with no pre-built kernel is fused at init time by hiprtc from the embedded device header."""
    from knaster_amd.bank import Stage

    n = 100
    p = configs.voice_parameters(n)
    w = configs.Workload("jit", [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_ONEPOLE_HPF), Stage(L.STAGE_ONEPOLE_LPF), Stage(L.STAGE_SVF),
                                 Stage(L.STAGE_MUL_ENV_AR), Stage(L.STAGE_DIV_CONST), Stage(L.STAGE_WR_ADD)], n, 96, L.F32, 2)
    w.ctor = {0: p["freq"].reshape(n, 1), 2: p["cutoff"].reshape(n, 1),
              3: np.stack([np.full(n, float(L.SVF_BAND)), p["cutoff"] * 0.5, p["q"], np.zeros(n)], axis=1),
              4: np.stack([p["attack"] * 0.05, p["release"] * 0.01], axis=1), 5: np.full((n, 1), 3.0), 6: np.full((n, 1), 0.001)}
    g = make_gpu(knh, w, L.MIX_LEFT_FOLD)
    o = make_oracle(oracle, w)
    v = np.arange(n, dtype=np.uint32)
    for block in range(5):
        for bank in (g, o):
            if block in (0, 3):
                bank.param_apply_many(v, 4, 2, L.VALUE_TRIGGER)
            if block == 2:
                bank.param_apply_many(v, 1, 0, L.VALUE_FLOAT, p["cutoff"] * 0.1)
        g_out, g_voices, _ = g.process_block_voices()
        o_out, o_voices, _, _ = o.process_block()
        assert_bit_equal(g_voices, o_voices, f"block {block} voices")
        assert_bit_equal(g_out, o_out, f"block {block} mix")
    assert np.max(np.abs(g_voices)) > 1e-4


def test_runtime_fused_kernel_equals_prebuilt_kernel(knh, monkeypatch):
    w = configs.config("C3", n_voices=500, block_size=256)
    outs = []
    for jit in ("0", "1"):
        monkeypatch.setenv("KNH_JIT", jit)
        monkeypatch.setenv("KNH_PIPELINE", "0")
        g = make_gpu(knh, w)
        fire_all(g, w.n_voices, *w.restart)
        outs.append(g.process_blocks(3)[0])
        g.close()
    assert_bit_equal(outs[0], outs[1], "hiprtc-built vs hipcc-built kernel")


@pytest.mark.parametrize("sample_type", [L.F32, L.F64])
def test_runtime_built_pipeline_equals_runtime_built_single_wave_kernel(knh, monkeypatch, sample_type):
    """A chain without a pre-built pipeline gets one at init time: hiprtc instantiates voice_pipe_kernel over stage
    groups cut by estimated cost (KNH_JIT_PIPE=0 keeps the single-wave form).  Same arithmetic, same bits; also
    against the hipcc-built pipeline of a chain that has one."""
    from knaster_amd.bank import Stage

    n, bs = 700, 160
    p = configs.voice_parameters(n)
    w = configs.Workload("jitpipe", [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_WR_MUL), Stage(L.STAGE_ONEPOLE_LPF), Stage(L.STAGE_SVF),
                                     Stage(L.STAGE_SAMPLE_DELAY), Stage(L.STAGE_MUL_ENV_AR, delayed_changes_per_block=2),
                                     Stage(L.STAGE_MUL_CONST)], n, bs, sample_type, 2)
    w.ctor = {0: p["freq"].reshape(n, 1), 1: np.full((n, 1), 0.7), 2: p["cutoff"].reshape(n, 1),
              3: np.stack([np.full(n, float(L.SVF_PEAK)), p["cutoff"] * 0.5, p["q"], np.zeros(n)], axis=1),
              4: np.full((n, 1), 0.011), 5: np.stack([p["attack"] * 0.1, p["release"] * 0.05], axis=1), 6: np.full((n, 1), 1.0 / n)}
    v = np.arange(n, dtype=np.uint32)
    outs = []
    for jit_pipe in ("0", "1"):
        monkeypatch.setenv("KNH_JIT_PIPE", jit_pipe)
        g = make_gpu(knh, w)
        g.param_apply_many(v, 4, 0, L.VALUE_FLOAT, (v % 500) / 48000.0 + 1e-6)
        res = []
        for block in range(5):
            if block in (0, 3):
                g.param_apply_many(v, 5, 2, L.VALUE_TRIGGER, delays=(v % bs).astype(np.uint16))
            res.append(g.process_block_voices())
        res.append((g.process_blocks(3)[0], None, 0))
        outs.append(res)
        g.close()
    for (o0, v0, f0), (o1, v1, f1) in zip(*outs):
        assert_bit_equal(o0, o1, "mix")
        if v0 is not None:
            assert_bit_equal(v0, v1, "per voice")
            assert f0 == f1
    assert np.max(np.abs(outs[0][2][1])) > 1e-6
    # a chain that has a hipcc-built pipeline: forcing run-time fusion must give the same bits
    w3 = configs.config("C3", n_voices=300, block_size=128, sample_type=sample_type)
    c3 = []
    for jit in ("0", "1"):
        monkeypatch.setenv("KNH_JIT", jit)
        monkeypatch.setenv("KNH_JIT_PIPE", "1")
        g = make_gpu(knh, w3)
        fire_all(g, w3.n_voices, *w3.restart)
        c3.append(g.process_blocks(4)[0])
        g.close()
    assert_bit_equal(c3[0], c3[1], "hiprtc-built vs hipcc-built pipeline")


@pytest.mark.parametrize("name,n_voices,block_size", [("C3", 1500, 512), ("C4", 700, 100), ("C5", 520, 128), ("C2", 300, 48),
                                                      ("D3", 900, 256), ("B3", 700, 128)])
def test_many_wave_kernels_equal_single_wave_kernel(knh, monkeypatch, name, n_voices, block_size):
    """KNH_WIDE=4/8/16: four, eight or sixteen 64-voice groups per workgroup (the builds used for very large banks)."""
    w = configs.config(name, n_voices=n_voices, block_size=block_size)
    outs = {}
    for wide in ("0", "4", "8", "16"):
        monkeypatch.setenv("KNH_PIPELINE", "0")
        monkeypatch.setenv("KNH_WIDE", wide)
        g = make_gpu(knh, w)
        res = []
        for block in range(5):
            if name in ("C3", "C4", "D3", "B3"):
                c3_script(w, block, g)
            if name == "C5":
                e = configs.c5_events(w, block)
                if e is not None:
                    g.param_apply_many(e[0], e[1], e[2], e[3], e[4], None, e[5])
            out, voices, flags = g.process_block_voices()
            res.append((out, voices, flags, g.read_done_frames()))
        res.append((g.process_blocks(3)[0], None, 0, None))
        outs[wide] = res
        g.close()
    for other in ("4", "8", "16"):
        for (o0, v0, f0, d0), (o1, v1, f1, d1) in zip(outs["0"], outs[other]):
            assert_bit_equal(o0, o1, f"mix, wide {other}")
            if v0 is not None:
                assert_bit_equal(v0, v1, f"per-voice, wide {other}")
                assert f0 == f1 and np.array_equal(d0, d1)


def test_long_run_stays_bit_exact(knh, oracle):
    """A few seconds of audio: 240 blocks of the C3 chain in 16-block launches with notes restarting and releasing at
    voice-dependent blocks and cutoff changes in between, every launch compared with the oracle's block-by-block run."""
    n, bs, per_launch, launches = 192, 256, 16, 15
    w = configs.config("C3", n_voices=n, block_size=bs)
    g = make_gpu(knh, w)
    o = make_oracle(oracle, w, want_mix=False)
    v = np.arange(n, dtype=np.uint32)
    p = configs.voice_parameters(n)
    rng = np.random.default_rng(99)
    for launch in range(launches):
        per_block = []
        for i in range(per_launch):
            blk = launch * per_launch + i
            ev = []
            on = v[(v * 7 + blk) % 23 == 0]
            off = v[(v * 5 + blk) % 31 == 0]
            if len(on):
                ev.append((on, w.restart[0], w.restart[1], L.VALUE_TRIGGER, None))
            if len(off):
                ev.append((off, w.release[0], w.release[1], L.VALUE_TRIGGER, None))
            if blk % 9 == 4:
                sel = v[rng.random(n) < 0.3]
                if len(sel):
                    ev.append((sel, 2, 0, L.VALUE_FLOAT, p["cutoff"][sel] * rng.uniform(0.5, 1.5, len(sel))))
            per_block.append(ev)
            for (sel, s, pi, kind, f) in ev:
                g.param_apply_many(sel, s, pi, kind, f, block_offset=i)
        got, _ = g.process_blocks(per_launch)
        for i in range(per_launch):
            for (sel, s, pi, kind, f) in per_block[i]:
                o.param_apply_many(sel, s, pi, kind, f)
            _, voices, _, _ = o.process_block()
            want = voices.astype(np.float64).sum(axis=0)
            tol = 1e-5 * max(1.0, float(np.abs(voices).max(axis=1).sum()))
            assert np.max(np.abs(got[i, 0].astype(np.float64) - want)) <= tol, f"launch {launch} block {i}"
    # and the state itself: one more single block, per voice, bit for bit
    _, gv, _ = g.process_block_voices()
    _, ov, _, _ = o.process_block()
    assert_bit_equal(gv, ov, "per-voice signals after 240 blocks")
    g.close()
    o.close()


@pytest.mark.parametrize("name,n_voices,block_size,threads", [("C5", 1100, 128, 4), ("C5", 4096, 128, 8), ("C3", 700, 96, 3),
                                                              ("C4", 300, 64, 2), ("C5", 100, 128, 5)])
def test_host_sharded_bank_equals_single_bank(knh, name, n_voices, block_size, threads):
    """knh_bank_create_sharded: K host threads, K voice ranges, K kernels -- every voice's samples, done frames and
    flags are those of the one-range bank, the mix is the sum of the ranges' tree mixes (tolerance of the tree mix),
    through the single-call and the batched entry points, one block or many per launch, whole and split blocks."""
    w = configs.config(name, n_voices=n_voices, block_size=block_size)
    tol = 2e-5 if w.sample_type == L.F32 else 1e-12

    def script(block, bank, offset):
        if name == "C5":
            e = configs.c5_events(w, block)
            if e is not None:
                bank.param_apply_many(e[0], e[1], e[2], e[3], e[4], None, e[5], block_offset=offset)
        else:
            c3_script(w, block, bank, offset)
    a = make_gpu(knh, w)
    b = make_gpu(knh, w, host_threads=threads)
    for block in range(5):
        script(block, a, 0)
        script(block, b, 0)
        if block == 3:  # a block in two parts
            half = block_size // 2
            oa0, va0, _ = a.process_block_voices(half, 0)
            oa1, va1, fa = a.process_block_voices(block_size - half, half)
            ob0, vb0, _ = b.process_block_voices(half, 0)
            ob1, vb1, fb = b.process_block_voices(block_size - half, half)
            assert_bit_equal(vb0[:, :half], va0[:, :half], "first part")
            assert_bit_equal(vb1[:, half:], va1[:, half:], "second part")
            np.testing.assert_allclose(ob0[:, :half], oa0[:, :half], rtol=0, atol=tol * max(1.0, float(np.abs(oa0).max())))
            np.testing.assert_allclose(ob1[:, half:], oa1[:, half:], rtol=0, atol=tol * max(1.0, float(np.abs(oa1).max())))
        else:
            oa, va, fa = a.process_block_voices()
            ob, vb, fb = b.process_block_voices()
            assert_bit_equal(vb, va, f"voices, block {block}")
            np.testing.assert_allclose(ob, oa, rtol=0, atol=tol * max(1.0, float(np.abs(oa).max())))
        assert fa == fb
        assert np.array_equal(a.read_done_frames(), b.read_done_frames())
    # several blocks per launch, changes addressed to later blocks, device-side output
    n_blocks = 6
    for k in range(n_blocks):
        script(5 + k, a, k)
        script(5 + k, b, k)
    ma, fa = a.process_blocks(n_blocks)
    mb, fb = b.process_blocks(n_blocks)
    np.testing.assert_allclose(mb, ma, rtol=0, atol=tol * max(1.0, float(np.abs(ma).max())))
    assert fa == fb
    assert_bit_equal(b.process_block_voices()[1], a.process_block_voices()[1], "state after the launch")
    # errors keep their codes
    with pytest.raises(L.KnasterHipError) as e:
        b.param_apply(n_voices, 0, 0, 1.0)
    assert e.value.status == L.ERR_OUT_OF_RANGE
    with pytest.raises(L.KnasterHipError) as e:
        b.param_apply(n_voices - 1, 99, 0, 1.0)
    assert e.value.status == L.ERR_OUT_OF_RANGE


@pytest.mark.parametrize("host_threads", [0, 3])
def test_batched_calls_for_later_blocks_in_any_order(knh, host_threads):
    """knh_bank_param_apply_many_at batches addressed to blocks of this launch and of later launches, submitted in scrambled
    block order, mixing parameters whose patches are made at once (triggers, SinWt freq, wr_mul) with parameters replayed
    when their block is assembled (every SvfFilter setter, the envelope times): same samples as the same calls made block
    by block, each block's calls in their submission order (GraphGen applies a block's events in arrival order,
    graph_gen.rs:110-166)."""
    n, bs, n_blocks = 300, 64, 10
    w = configs.config("C3", n_voices=n, block_size=bs)
    v = np.arange(n, dtype=np.uint32)
    F, T = L.VALUE_FLOAT, L.VALUE_TRIGGER
    calls = [  # (block, voices, stage, param, kind, values)
        (5, v, 3, 3, T, None),                          # t_restart again
        (2, v[::2], 2, 0, F, 400.0 + 3.0 * v[::2]),     # SVF cutoff (shadowed)
        (2, v, 0, 0, F, 110.0 + v),                     # SinWt freq
        (0, v, 3, 3, T, None),                          # t_restart
        (7, v[::3], 3, 2, T, None),                     # t_release
        (5, v, 3, 0, F, 0.001 + 1e-5 * v),              # attack_time (skip-if-unchanged shadow)
        (2, v[1::2], 2, 1, F, 0.7 + 0.01 * v[1::2]),    # SVF q, other voices
        (2, v[::2], 2, 1, F, 1.5 + 0.0 * v[::2]),       # SVF q after the cutoff of the same voices, same block
        (9, v, 1, 0, F, 0.5 / n + 0.0 * v),             # wr_mul
        (3, v[::5], 2, 3, L.VALUE_INTEGER, 2 + 0 * v[::5]),  # SVF filter type
        (8, v, 3, 2, T, None),
        (6, v, 0, 1, F, 1000.0 + v),                    # phase_offset
    ]

    def send(bank, c, offset):
        _, vs, stage, param, kind, vals = c
        iv = vals.astype(np.int64) if kind == L.VALUE_INTEGER else None
        fv = vals if kind == F else None
        bank.param_apply_many(vs, stage, param, kind, fv, iv, None, block_offset=offset)
    a = make_gpu(knh, w)
    single = []
    for block in range(n_blocks):
        for c in calls:
            if c[0] == block:
                send(a, c, 0)
        single.append(a.process_block()[0])
    for split in ([10], [4, 6], [3, 3, 4], [1, 1, 1, 1, 1, 1, 1, 1, 1, 1]):
        b = make_gpu(knh, w, host_threads=host_threads)
        for c in calls:
            send(b, c, c[0])
        got = np.concatenate([b.process_blocks(k)[0] for k in split])
        if host_threads:
            assert np.max(np.abs(got.astype(np.float64) - np.stack(single))) <= 1e-5
        else:
            for block in range(n_blocks):
                assert_bit_equal(got[block], single[block], f"split {split} block {block}")
        b.close()
    a.close()


def test_fan_pipeline_with_sample_accurate_changes_equals_single_wave_kernel(knh, monkeypatch):
    """SinNumeric * gain as the Fan pipeline runs it (phase accumulator in one wavefront, sin * gain on eight, each taking a
    window of every 64-sample tile) against the single-wave kernel, bit for bit, with sample-accurate changes landing in
    every part of a tile: new frequencies and phase offsets (the serial wavefront's parameters), new gains (parameters every
    one of the eight wavefronts holds), phase resets (state: the general path), on whole and ragged blocks."""
    from knaster_amd.bank import Stage, TRIGGER
    for block_size in (256, 200):
        n = 300
        w = configs.Workload("fan", [Stage(L.STAGE_SIN_NUMERIC, delayed_changes_per_block=4), Stage(L.STAGE_MUL_CONST, delayed_changes_per_block=4)],
                             n, block_size, L.F32, 2)
        p = configs.voice_parameters(n)
        w.ctor = {0: p["freq"].reshape(n, 1), 1: np.full((n, 1), 1.0 / n)}
        rng = np.random.default_rng(11)
        script = []
        for block in range(6):
            ops = []
            for _ in range(int(rng.integers(20, 200))):
                voice = int(rng.integers(0, n))
                stage = int(rng.integers(0, 2))
                if stage == 0:
                    param = int(rng.integers(0, 3))
                    value = [float(rng.uniform(50, 5000)), float(rng.uniform(0, 1)), None][param]
                else:
                    param, value = 0, float(rng.uniform(0.0, 2.0 / n))
                ops.append((voice, stage, param, value, int(rng.integers(0, block_size + 8))))
            script.append(ops)
        outs = {}
        for pipeline in ("0", "1"):
            monkeypatch.setenv("KNH_PIPELINE", pipeline)
            g = make_gpu(knh, w)
            res = []
            for block in range(6):
                for voice, stage, param, value, delay in script[block]:
                    g.set_delay_within_block_for_param(voice, stage, param, delay)
                    g.param_apply(voice, stage, param, TRIGGER if value is None else value)
                res.append(g.process_block_voices()[:2])
            outs[pipeline] = res
            g.close()
        for (o0, v0), (o1, v1) in zip(outs["0"], outs["1"]):
            assert_bit_equal(v0, v1, f"block_size {block_size}: per-voice")
            assert_bit_equal(o0, o1, f"block_size {block_size}: mix")
        assert np.abs(outs["1"][-1][0]).max() > 0


@pytest.mark.parametrize("shards", [1, 8])
def test_full_size_c4_f64_oracle_subset_and_sharded_sum(knh, oracle, shards):
    """BASELINE.json config C4 at its full size: 65 536 voices, f64, block 512.  (a) 96 sampled voices equal the oracle run on
    just those voices, bit for bit; (b) the mix equals the documented tree fold of the per-voice signals (bit for bit) and the
    f64 sum within 1e-12; (c) split over eight voice ranges as an 8-GPU run splits it (8 192 voices each, here all on this
    GPU through knh_bank_create_multi_device), the sum of the ranges' mixes equals the one-bank mix within 1e-12."""
    w = configs.config("C4")
    assert w.n_voices == 65536 and w.sample_type == L.F64
    if shards == 1:
        g = make_gpu(knh, w, L.MIX_TREE)
        pick = np.unique(np.concatenate([np.arange(0, 32), np.arange(32767, 32799), np.arange(w.n_voices - 32, w.n_voices)]))
        sub = configs.Workload("sub", w.stages, len(pick), w.block_size, w.sample_type, w.out_channels,
                               {s: a[pick] for s, a in w.ctor.items()})
        o = make_oracle(oracle, sub, want_mix=False)
        for block in range(3):
            for bank, n in ((g, w.n_voices), (o, len(pick))):
                if block == 0:
                    fire_all(bank, n, *w.restart)
                if block == 2:
                    fire_all(bank, n, w.release[0], w.release[1])
            out, voices, _ = g.process_block_voices()
            _, o_voices, _, _ = o.process_block()
            assert_bit_equal(voices[pick], o_voices, f"block {block}: sampled voices vs oracle")
            assert_bit_equal(out[0], tree_mix(voices), f"block {block}: tree mix")
            assert_bit_equal(out[0], out[1], "L == R")
            ref = voices.sum(axis=0)
            assert np.max(np.abs(out[0] - ref)) <= 1e-12 and np.max(np.abs(ref)) > 1e-4
        g.close()
        o.close()
    else:
        one = make_gpu(knh, w, L.MIX_TREE)
        many = knh.VoiceBank(w.stages, w.n_voices, w.sample_type, w.out_channels, L.MIX_TREE, devices=[0] * shards)
        for s, a in w.ctor.items():
            many.set_ctor_args(s, a)
        many.init(configs.SAMPLE_RATE, w.block_size)
        assert many.ranks() == shards
        for bank in (one, many):
            fire_all(bank, w.n_voices, *w.restart)
            bank.param_apply_many(np.arange(w.n_voices, dtype=np.uint32), w.release[0], w.release[1], L.VALUE_TRIGGER, block_offset=2)
        a, _ = one.process_blocks(4)
        b, _ = many.process_blocks(4)
        assert np.max(np.abs(a - b)) <= 1e-12 and np.max(np.abs(a)) > 1e-4
        one.close()
        many.close()


@pytest.mark.parametrize("kind", ["plain", "host_sharded", "multi_device", "rank"])
def test_pipelined_host_output_equals_blocking_calls(knh, kind):
    """knh_bank_process_blocks_begin / _end (two launches in flight, blocks copied to pinned memory behind the kernels) give
    the samples knh_bank_process_blocks gives, for every kind of bank; misuse is refused."""
    import ctypes
    w = configs.config("C3", n_voices=700, block_size=64)
    kw = {"plain": {}, "host_sharded": {"host_threads": 3}, "multi_device": {"devices": [0, 0]}, "rank": {"rank": 0, "world": 1}}[kind]

    def make():
        b = knh.VoiceBank(w.stages, w.n_voices, w.sample_type, w.out_channels, L.MIX_TREE, -1, False, **kw)
        for s, a in w.ctor.items():
            b.set_ctor_args(s, a)
        b.init(configs.SAMPLE_RATE, w.block_size)
        return b

    def events(bank, launch):  # the C3 script's blocks 0..5, one per launch, addressed to block 1 of the launch
        c3_script(w, launch, bank, 1 if sizes[launch] > 1 else 0)
    a, b = make(), make()
    sizes = [4, 1, 7, 3, 5, 2]
    want = []
    for launch, k in enumerate(sizes):
        events(a, launch)
        want.append(a.process_blocks(k)[0])
    scratch = np.zeros(8, dtype=np.float32)
    assert b._lib.knh_bank_process_blocks_end(b._h, scratch.ctypes.data_as(ctypes.c_void_p)) != L.OK  # nothing outstanding
    got = []
    for launch, k in enumerate(sizes):
        events(b, launch)
        b.process_blocks_begin(k)
        if launch == 1:
            assert b._lib.knh_bank_process_blocks_begin(b._h, 1, 0) != L.OK  # two outstanding is the limit
        if launch >= 1:
            got.append(b.process_blocks_end())
    got.append(b.process_blocks_end())
    for launch in range(len(sizes)):
        assert_bit_equal(got[launch], want[launch], f"{kind}: launch {launch}")
    assert np.abs(want[1]).max() > 0
    a.close()
    b.close()


@pytest.mark.parametrize("n_voices", [1, 3, 4, 5, 63, 64, 65, 100, 257, 1000, 4100, 16385])
def test_tree_mix_is_the_pairwise_sum_of_the_voices(knh, monkeypatch, n_voices):
    """KNH_MIX_TREE is a fixed order of additions (include/knaster_hip.h): the binary tree over the voice index, bit for bit
    the same in every kernel form."""
    from helpers import pairwise_sum
    w = configs.config("C3", n_voices=n_voices, block_size=96)
    for form in ({"KNH_PIPELINE": "0"}, {"KNH_PIPELINE": "1", "KNH_PIPE_BIG": "0"}, {"KNH_PIPELINE": "1", "KNH_PIPE_BIG": "1"},
                 {"KNH_PIPELINE": "1", "KNH_PIPE_BIG": "2"}, {"KNH_PIPELINE": "0", "KNH_WIDE": "4"}, {"KNH_JIT": "1"}):
        for k in ("KNH_PIPELINE", "KNH_PIPE_BIG", "KNH_WIDE", "KNH_JIT"):
            monkeypatch.delenv(k, raising=False)
        for k, val in form.items():
            monkeypatch.setenv(k, val)
        if n_voices > 5000 and "KNH_JIT" in form:
            continue
        w.block_size = 96 if form.get("KNH_PIPE_BIG", "0") == "0" else 128  # (the 64-sample-tile forms want whole tiles)
        g = make_gpu(knh, w)
        fire_all(g, n_voices, *w.restart)
        for block in range(2):
            out, voices, _ = g.process_block_voices()
            assert_bit_equal(out[0], pairwise_sum(voices), f"{n_voices} voices, {form}, block {block}")
        assert np.abs(out).max() > 0
        g.close()


@pytest.mark.parametrize("name,n_voices", [("C4", 333), ("P3", 150), ("P3", 1029), ("C2", 77), ("C5", 300)])
def test_tree_mix_pairwise_other_chains(knh, name, n_voices):
    """The same for f64 samples (C4), Pan2 chains (one tree per channel over the rounded products), the Fan pipeline (C2) and C5."""
    from helpers import pairwise_sum
    w = configs.config(name, n_voices=n_voices, block_size=64)
    g = make_gpu(knh, w)
    if w.restart:
        fire_all(g, n_voices, *w.restart)
    for block in range(2):
        out, voices, _ = g.process_block_voices()
        if voices.ndim == 3:
            for ch in range(2):
                assert_bit_equal(out[ch], pairwise_sum(voices[ch]), f"{name} channel {ch}")
        else:
            assert_bit_equal(out[0], pairwise_sum(voices), name)
    assert np.abs(out).max() > 0
    g.close()
