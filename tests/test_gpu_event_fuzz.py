"""Random parameter traffic against the oracle: batches of changes on random voices (repeats included, so that the
WrPreciseTiming queues fill, block and overflow), random parameters and triggers, random in-block delays, given block by
block to one bank (per-voice signals bit-identical to the oracle) and scheduled ahead inside multi-block launches to another
(its mix bit-identical to the first bank's).  KNH_TEST_SEEDS=200 for a soak run."""
import os

import numpy as np
import pytest

from helpers import assert_bit_equal, make_gpu, make_oracle
from knaster_amd import _lib as L
from knaster_amd import configs

pytestmark = pytest.mark.gpu

FLOATS = {"freq": (50.0, 3000.0), "phase_offset": (0.0, 1.0), "wr_mul": (-1.0, 1.0), "cutoff_freq": (200.0, 6000.0), "q": (0.5, 4.0),
          "attack_time": (0.001, 0.02), "release_time": (0.005, 0.1), "value": (-1.0, 1.0), "pan": (-1.0, 1.0), "gain": (-6.0, 6.0),
          "delay_time": (0.0, 0.0045), "feedback": (-0.8, 0.8), "time_scale": (0.5, 2.0), "pulse_width": (0.1, 0.9)}
TRIGGERS = {"t_restart", "t_release", "reset_phase", "t_stop", "t_calculate_coefficients"}
INTS = {"filter": (0, 9), "jump_to_segment": (0, 4)}   # PInteger parameters (SvfFilter's type, Envelope's segment)
BOOLS = {"looping"}


@pytest.mark.parametrize("seed", range(int(os.environ.get("KNH_TEST_SEEDS", "32"))))
def test_random_parameter_traffic(knh, oracle, monkeypatch, seed):
    rng = np.random.default_rng(7000 + seed)
    # every kernel form gets its share of the seeds (the switches are read when a bank is created)
    form = [{}, {"KNH_PIPE_BIG": "0"}, {"KNH_PIPE_BIG": "1"}, {"KNH_PIPELINE": "0"}, {"KNH_PIPELINE": "0", "KNH_WIDE": "4"}, {"KNH_JIT": "1"},
            {"KNH_JIT": "1", "KNH_JIT_PIPE": "0"}][(seed // 8) % 7]
    for k_, v_ in form.items():
        monkeypatch.setenv(k_, v_)
    name = ["C5", "C3", "RANDOM", "D3", "P3", "C4", "M1", "DAG", "SMOOTH", "ARITH", "RANDOM", "INPUT"][seed % 12]  # (B3's PolyBlep waveforms with a sin in them are tolerance-only)
    n = int(rng.integers(65, 260)) if seed % 3 else int(rng.integers(1, 140))
    bs = int(rng.choice([64, 128, 96])) if seed % 2 else int(rng.choice([16, 17, 33, 40, 100, 200, 256]))  # (ragged tiles too)
    first = []
    if name == "RANDOM":  # a random chain of the run-time fused kind (delay lines, segment envelopes, wrappers ..), its own sizes
        from test_gpu_random_chains import random_chain
        w, _rng, _changes, triggers = random_chain(1000 + seed)
        n, bs = w.n_voices, w.block_size
        first = [(s, restart) for (s, restart, _rel) in triggers]
    elif name == "DAG":  # a voice that is a small graph (filters, an envelope, two-signal arithmetic), some stages precise-timed
        from test_gpu_dag import random_dag
        st, ctor = random_dag(np.random.default_rng(500 + seed), int(rng.integers(4, 14)))
        from knaster_amd.bank import Stage
        st = [Stage(x.kind, x.flags, int(rng.integers(1, 3)), x.input, x.input2)
              if x.kind in (L.STAGE_SVF, L.STAGE_ONEPOLE_LPF, L.STAGE_MUL_ENV_AR, L.STAGE_SIN_WT) and rng.random() < 0.5 else x for x in st]
        w = configs.Workload(f"dag{seed}", st, n, bs, L.F32 if seed % 16 < 8 else L.F64, 1)
        w.ctor = {s_: np.tile(np.asarray(a_, dtype=np.float64), (n, 1)) * ((1.0 + 0.01 * np.arange(n)).reshape(n, 1) if st[s_].kind == L.STAGE_SIN_WT else 1.0)
                  for s_, a_ in ctor.items()}
        first = [(i_, 2) for i_, x in enumerate(st) if x.kind == L.STAGE_MUL_ENV_AR]
    elif name == "SMOOTH":  # WrSmoothParams (smooth_params.rs) on an oscillator, a filter (also precise-timed) and a constant
        from knaster_amd.bank import Stage
        SM = L.STAGE_FLAG_SMOOTH_PARAMS
        p_ = configs.voice_parameters(n)
        w = configs.Workload(f"smooth{seed}", [Stage(L.STAGE_SIN_WT, flags=SM), Stage(L.STAGE_SVF, flags=SM, delayed_changes_per_block=int(rng.integers(0, 4))),
                                               Stage(L.STAGE_MUL_CONST, flags=SM)], n, bs, L.F32 if seed % 24 < 12 else L.F64, 2)
        w.ctor = {0: p_["freq"].reshape(n, 1), 1: np.stack([np.zeros(n), p_["cutoff"], p_["q"], np.zeros(n)], axis=1), 2: np.full((n, 1), 1.0 / n)}
    elif name == "ARITH":  # oscillators and arithmetic only: the frame-parallel forms (voice_frame.hpp; the interpreter for every other seed of these)
        from test_gpu_dag import arithmetic_dag
        from knaster_amd.bank import Stage
        st, ctor = arithmetic_dag(np.random.default_rng(900 + seed), int(rng.integers(5, 40)))
        if sum(x.kind == L.STAGE_SIN_WT for x in st) < 2:
            st.append(Stage(L.STAGE_SIN_WT))
            ctor[len(st) - 1] = [333.0]
            st.append(Stage(L.STAGE_MATH_ADD, input=len(st), input2=len(st) - 1))
        w = configs.Workload(f"arith{seed}", st, n, bs, L.F32 if seed % 24 < 12 else L.F64, 1)
        w.ctor = {s_: np.tile(np.asarray(a_, dtype=np.float64), (n, 1)) * (1.0 + 0.01 * np.arange(n)).reshape(n, 1) for s_, a_ in ctor.items()}
        monkeypatch.setenv("KNH_FRAME_JIT", "1" if (seed // 12) % 2 else "0")
    elif name == "INPUT":  # a filter bank on the bank node's input 0, input 1 times a per-voice oscillator beside it
        from knaster_amd.bank import Stage
        p_ = configs.voice_parameters(n)
        st = [Stage(L.STAGE_INPUT), Stage(L.STAGE_SVF, delayed_changes_per_block=int(rng.integers(0, 3))),
              Stage(L.STAGE_MUL_ENV_ASR, delayed_changes_per_block=int(rng.integers(0, 3))),
              Stage(L.STAGE_INPUT), Stage(L.STAGE_SIN_WT), Stage(L.STAGE_MATH_MUL, input=4, input2=5), Stage(L.STAGE_MATH_ADD, input=3, input2=6),
              Stage(L.STAGE_MUL_CONST)]
        w = configs.Workload(f"input{seed}", st, n, bs, L.F32 if seed % 24 < 12 else L.F64, 1, in_channels=2)
        w.ctor = {0: np.zeros((n, 1)), 1: np.stack([np.full(n, 2.0), p_["cutoff"], p_["q"], np.zeros(n)], axis=1), 2: np.tile([0.002, 0.01], (n, 1)),
                  3: np.ones((n, 1)), 4: p_["freq"].reshape(n, 1), 7: np.full((n, 1), 1.0 / n)}
        first = [(2, 3)]
    else:
        w = configs.config(name, n_voices=n, block_size=bs, precise=int(rng.integers(0, 4)))
    sharded = seed % 5 == 4 and os.environ.get("KNH_FUZZ_NO_SHARD") != "1"  # the scheduled-ahead bank with its host work on three threads (three voice ranges)
    a = make_gpu(knh, w, L.MIX_LEFT_FOLD)   # block by block, per-voice signals
    # multi-block launches, changes scheduled ahead; for a fifth of the seeds the bank is cut into voice ranges: on three host
    # threads, on three "GPUs" (all of them device 0 here), or as the one rank of a one-rank job
    how = [{"host_threads": 3}, {"devices": [0, 0, 0]}, {"rank": 0, "world": 1}][(seed // 5) % 3] if sharded else {}
    b = make_gpu(knh, w, **how)
    c = make_gpu(knh, w)                    # block by block, the same (tree) mix as b
    o = make_oracle(oracle, w)
    targets = []  # (stage, param, name)
    for s in range(len(w.stages)):
        for p, pname in enumerate(a.stage_param_descriptions(s)):
            if pname in FLOATS or pname in TRIGGERS or pname in INTS or pname in BOOLS:
                targets.append((s, p, pname))
    if not targets:
        pytest.skip("a chain without a parameter this test knows how to set")
    # A delay armed on a wrapper's own parameter goes to the node it wraps (wrappers_core/math.rs:109-112), and if that node is a
    # WrPreciseTiming its next_delay array has no such index (precise_timing.rs:146-148: out of bounds): not a defined case.
    WRAPPERS = (L.STAGE_WR_MUL, L.STAGE_WR_ADD, L.STAGE_WR_SUB, L.STAGE_WR_VSUB, L.STAGE_WR_DIV, L.STAGE_WR_VDIV, L.STAGE_WR_POWF, L.STAGE_WR_POWI)
    no_delay = set()
    base = 0
    for s_i, st_ in enumerate(w.stages):
        if st_.kind in WRAPPERS:
            if w.stages[base].delayed_changes_per_block > 0:
                no_delay.add(s_i)
        else:
            base = s_i
    n_blocks = 12
    ins = None
    if getattr(w, "in_channels", 0):
        t_ = np.arange(n_blocks * bs).reshape(n_blocks, 1, bs) / 48000.0
        ins = np.concatenate([np.sin(2 * np.pi * 110.0 * (ch + 1) * t_) for ch in range(w.in_channels)], axis=1)
        ins = (0.5 * ins + 0.05 * rng.standard_normal(ins.shape)).astype(np.float64 if w.sample_type == L.F64 else np.float32)
    plan = []
    for blk in range(n_blocks):
        batches = []
        if blk == 0 and w.restart:
            v = np.arange(n, dtype=np.uint32)
            batches.append((v, w.restart[0], w.restart[1], L.VALUE_TRIGGER, None, None))
        if blk == 0:
            for (s0, p0) in first:
                batches.append((np.arange(n, dtype=np.uint32), s0, p0, L.VALUE_TRIGGER, None, None))
        for _ in range(int(rng.integers(0, 6))):
            s, p, pname = targets[int(rng.integers(0, len(targets)))]
            m = int(rng.integers(1, 2 * n)) if rng.random() < 0.8 else int(rng.integers(1, 13))
            v = rng.integers(0, n, m).astype(np.uint32)          # repeats: several changes of one node in one block
            if rng.random() < 0.5:
                v = np.sort(v)
            # (a fifth of the delayed batches reach past the end of the block: such a change is never applied, nor is
            # anything queued behind it on the same node -- precise_timing.rs:65-114)
            hi_delay = bs + bs // 2 if rng.random() < 0.2 else bs
            delays = rng.integers(0, hi_delay, m).astype(np.uint16) if rng.random() < 0.8 and s not in no_delay else None
            if name == "SMOOTH" and pname in FLOATS and rng.random() < 0.35:
                # a Smoothing value for the parameter: off, or linear over a few milliseconds at block or audio rate
                # (ParameterValue::Smoothing: seconds in the float, 0 none / 1 block rate / 2 audio rate in the integer)
                mode = rng.integers(0, 3, m).astype(np.int64)
                batches.append((v, s, p, L.VALUE_SMOOTHING, np.where(mode == 0, 0.0, rng.uniform(0.0005, 0.01, m)), delays, mode))
            elif pname in INTS:
                batches.append((v, s, p, L.VALUE_INTEGER, None, delays, rng.integers(*INTS[pname], m).astype(np.int64)))
            elif pname in BOOLS:
                batches.append((v, s, p, L.VALUE_BOOL, None, delays, rng.integers(0, 2, m).astype(np.int64)))
            elif pname in TRIGGERS:
                batches.append((v, s, p, L.VALUE_TRIGGER, None, delays))
            else:
                lo, hi = FLOATS[pname]
                batches.append((v, s, p, L.VALUE_FLOAT, rng.uniform(lo, hi, m), delays))
        plan.append(batches)
    # block by block
    mixes = []
    for blk in range(n_blocks):
        for (v, s, p, kind, f, d, *iv) in plan[blk]:
            for bank in (a, c, o):
                if bank is a and len(v) <= 12 and kind in (L.VALUE_FLOAT, L.VALUE_TRIGGER):
                    # the UGen entry points themselves, call by call (set_delay_within_block_for_param, then param_apply)
                    from knaster_amd.bank import TRIGGER
                    for q in range(len(v)):
                        if d is not None and d[q] > 0:
                            bank.set_delay_within_block_for_param(int(v[q]), s, p, int(d[q]))
                        bank.param_apply(int(v[q]), s, p, TRIGGER if kind == L.VALUE_TRIGGER else float(f[q]))
                    continue
                bank.param_apply_many(v, s, p, kind, f, iv[0] if iv else None, d)
        if ins is not None:
            for bank in (a, c, o):
                bank.set_input(ins[blk])
        _, av, af = a.process_block_voices()
        _, ov, _of, od = o.process_block()
        if np.isnan(ov).any():  # (a filter the random settings drove to NaN: NaN on both sides, whatever its sign and payload)
            assert np.array_equal(np.isnan(av), np.isnan(ov)), f"seed {seed} {name} block {blk}: NaNs in different places"
            av, ov = np.nan_to_num(av, nan=0.0), np.nan_to_num(ov, nan=0.0)
        assert_bit_equal(av, ov, f"seed {seed} {name} block {blk} per-voice")
        # (several envelopes of a graph-shaped voice finishing in one block: the last one in the reference's task order names
        # the done frame -- one UGenFlags for all tasks, graph_gen.rs:196-200; graph.rs calculate_node_order)
        np.testing.assert_array_equal(a.read_done_frames(), od)
        mixes.append(c.process_block()[0])
    # the same traffic scheduled ahead, in launches of 1-4 blocks
    blk = 0
    while blk < n_blocks:
        k = min(int(rng.integers(1, 5)), n_blocks - blk)
        for i in range(k):
            for (v, s, p, kind, f, d, *iv) in plan[blk + i]:
                b.param_apply_many(v, s, p, kind, f, iv[0] if iv else None, d, block_offset=i)
        if ins is not None:
            b.set_input(ins[blk:blk + k])
        out = b.process_blocks(k)[0]
        for i in range(k):
            if sharded:  # (each voice range has a tree of its own: equal up to the re-association)
                want = mixes[blk + i].astype(np.float64)  # (a filter driven to NaN by the random settings is NaN on both sides)
                assert np.array_equal(np.isfinite(out[i]), np.isfinite(want)), (seed, blk + i)
                fin = np.isfinite(want)
                if fin.any():
                    err = float(np.max(np.abs(out[i].astype(np.float64)[fin] - want[fin])))
                    assert err <= 1e-5 * max(1.0, float(np.abs(want[fin]).max())), (seed, blk + i, err)
            else:
                assert_bit_equal(out[i], mixes[blk + i], f"seed {seed} {name}: block {blk + i} of a {k}-block launch")
        blk += k
    for bank in (a, b, c, o):
        bank.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("KNH_TEST_SEEDS", "16"))))
def test_changes_between_partial_blocks_equal_delayed_changes(knh, monkeypatch, seed):
    """Two ways a host gets a change to land on frame k of a block, which the reference defines to be the same thing
    (precise_timing.rs:65-114 is literally the first done by the wrapper): processing frames [0, k), applying the change,
    processing [k, B) (BlockMetadata::make_partial, ugen.rs:87-93) -- or arming a delay of k on a WrPreciseTiming-wrapped
    node and processing the whole block.  Every stage of the chain is wrapped, one change per voice and block."""
    from knaster_amd.bank import Stage
    rng = np.random.default_rng(9000 + seed)
    form = [{}, {"KNH_PIPE_BIG": "0"}, {"KNH_PIPELINE": "0"}, {"KNH_PIPE_BIG": "1"}][seed % 4]
    for k_, v_ in form.items():
        monkeypatch.setenv(k_, v_)
    n = int(rng.integers(65, 200))
    bs = int(rng.choice([64, 128, 256]))
    p_ = configs.voice_parameters(n)
    st = [Stage(L.STAGE_SIN_WT, delayed_changes_per_block=1), Stage(L.STAGE_WR_MUL), Stage(L.STAGE_SVF, delayed_changes_per_block=1),
          Stage(L.STAGE_MUL_ENV_ASR, delayed_changes_per_block=1)]
    w = configs.Workload(f"partial{seed}", st, n, bs, L.F32 if seed % 8 < 4 else L.F64, 2)
    w.ctor = {0: p_["freq"].reshape(n, 1), 1: np.full((n, 1), 1.0 / n), 2: np.stack([np.zeros(n), p_["cutoff"], p_["q"], np.zeros(n)], axis=1),
              3: np.stack([p_["attack"], p_["release"]], axis=1)}
    whole, parts = make_gpu(knh, w), make_gpu(knh, w)
    v_all = np.arange(n, dtype=np.uint32)
    for bank in (whole, parts):
        bank.param_apply_many(v_all, 3, 3, L.VALUE_TRIGGER)
    targets = [(0, 0, "freq"), (0, 1, "phase_offset"), (2, 0, "cutoff_freq"), (2, 1, "q"), (3, 0, "attack_time"), (3, 1, "release_time"), (3, 2, "t_release"), (3, 3, "t_restart")]
    for blk in range(10):
        k = int(rng.integers(1, bs))
        s, p, pname = targets[int(rng.integers(0, len(targets)))]
        v = np.sort(rng.choice(n, int(rng.integers(1, n + 1)), replace=False)).astype(np.uint32)
        kind = L.VALUE_TRIGGER if pname.startswith("t_") else L.VALUE_FLOAT
        f = None if kind == L.VALUE_TRIGGER else rng.uniform(*FLOATS[pname], len(v))
        whole.param_apply_many(v, s, p, kind, f, None, np.full(len(v), k, dtype=np.uint16))
        ref, _ = whole.process_block()
        out = np.zeros_like(ref)
        parts.process_block(frames_to_process=k, block_start_offset=0, out=out)
        parts.param_apply_many(v, s, p, kind, f)
        parts.process_block(frames_to_process=bs - k, block_start_offset=k, out=out)
        assert_bit_equal(out, ref, f"seed {seed} block {blk}: {pname} on {len(v)} voices at frame {k} of {bs}")
    whole.close()
    parts.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("KNH_TEST_SEEDS", "12"))))
def test_bad_calls_are_refused_and_leave_no_trace(knh, seed):
    """Batches with out-of-range voices, stages, parameter indices and wrong value kinds among good calls: the call reports an
    error, the good calls of the batch take effect, the bad ones leave nothing behind -- the bank renders what a bank that
    only ever saw the good calls renders (and no kernel ever sees an index the host did not check)."""
    rng = np.random.default_rng(11000 + seed)
    name = ["C3", "C5", "D3", "M1"][seed % 4]
    n = int(rng.integers(10, 200))
    bs = int(rng.choice([64, 100, 128]))
    w = configs.config(name, n_voices=n, block_size=bs, precise=int(rng.integers(0, 3)))
    clean, dirty = make_gpu(knh, w), make_gpu(knh, w)
    targets = []
    for s in range(len(w.stages)):
        for p, pname in enumerate(clean.stage_param_descriptions(s)):
            if pname in FLOATS or pname in TRIGGERS:
                targets.append((s, p, pname))
    v_all = np.arange(n, dtype=np.uint32)
    if w.restart:
        for bank in (clean, dirty):
            bank.param_apply_many(v_all, w.restart[0], w.restart[1], L.VALUE_TRIGGER)
    for blk in range(8):
        for _ in range(int(rng.integers(1, 4))):
            s, p, pname = targets[int(rng.integers(0, len(targets)))]
            m = int(rng.integers(1, 3 * n))
            v = rng.integers(0, n, m).astype(np.uint32)
            st_ = np.full(m, s, dtype=np.uint32)
            pa = np.full(m, p, dtype=np.uint32)
            trig = pname in TRIGGERS
            kinds = np.full(m, L.VALUE_TRIGGER if trig else L.VALUE_FLOAT, dtype=np.uint32)
            f = np.zeros(m) if trig else rng.uniform(*FLOATS[pname], m)
            d = rng.integers(0, bs, m).astype(np.uint16)
            bad = rng.random(m) < 0.15
            how = rng.integers(0, 4, m)
            vb, sb, pb, kb = v.copy(), st_.copy(), pa.copy(), kinds.copy()
            vb[bad & (how == 0)] = n + rng.integers(0, 1000)
            sb[bad & (how == 1)] = len(w.stages) + rng.integers(0, 50)
            pb[bad & (how == 2)] = 7 + rng.integers(0, 50)
            kb[bad & (how == 3)] = L.VALUE_FLOAT if trig else L.VALUE_TRIGGER
            off = int(rng.integers(0, 2))
            good = ~bad
            if good.any():
                clean.param_apply_many(v[good], st_[good], pa[good], kinds[good], f[good], None, d[good], block_offset=off)
            if bad.any():
                with pytest.raises(L.KnasterHipError):
                    dirty.param_apply_many(vb, sb, pb, kb, f, None, d, block_offset=off)
            else:
                dirty.param_apply_many(vb, sb, pb, kb, f, None, d, block_offset=off)
        k = 2
        assert_bit_equal(dirty.process_blocks(k)[0], clean.process_blocks(k)[0], f"seed {seed} {name} blocks {2 * blk}..")
    clean.close()
    dirty.close()


@pytest.mark.parametrize("name,n,bs", [("C5", 300, 128), ("C5", 70, 100), ("M1", 200, 64)])
def test_device_resolved_change_queues_equal_the_hosts(knh, monkeypatch, name, n, bs):
    """The WrPreciseTiming queues of nodes whose setters need no host library call are resolved by kernels
    (kernels_events.hip; KNH_DEV_EVENTS=0: by the host, as before): the same calls -- batches with delays, repeats that fill
    and block the queues, single arm / value calls, calls scheduled ahead inside and beyond a multi-block launch -- give the
    same samples, bit for bit."""
    from knaster_amd.bank import Stage, TRIGGER
    rng = np.random.default_rng(4242)
    if name == "C5":
        w = configs.config("C5", n_voices=n, block_size=bs)
        targets = [(0, 0, (50.0, 3000.0)), (0, 1, (0.0, 60000.0)), (3, 1, (0.0, 60000.0)), (3, 0, (100.0, 900.0)), (0, 2, None), (3, 2, None)]
    else:
        st = [Stage(L.STAGE_SIN_WT, delayed_changes_per_block=3), Stage(L.STAGE_WR_MUL), Stage(L.STAGE_MUL_ENV_AR, delayed_changes_per_block=2),
              Stage(L.STAGE_MUL_CONST, delayed_changes_per_block=1)]
        w = configs.Workload("devq", st, n, bs, L.F32, 2)
        p = configs.voice_parameters(n)
        w.ctor = {0: p["freq"].reshape(n, 1), 1: np.full((n, 1), 0.5), 2: np.tile([0.001, 0.003], (n, 1)), 3: np.full((n, 1), 1.0 / n)}
        targets = [(0, 0, (50.0, 3000.0)), (0, 1, (0.0, 60000.0)), (0, 2, None), (2, 0, (0.0005, 0.004)), (2, 1, (0.001, 0.01)), (2, 2, None),
                   (3, 0, (0.0, 0.02)), (1, 0, (0.1, 0.9))]
    script = []
    for launch in range(5):
        calls = []
        for _ in range(int(rng.integers(3, 9))):
            stage, param, rng_ = targets[int(rng.integers(0, len(targets)))]
            k = int(rng.integers(1, n))
            voices = rng.integers(0, n, size=k).astype(np.uint32)  # repeats: queues fill, block and overflow
            delays = rng.choice([0, 0, 1, 7, bs // 2, bs - 1, bs, bs + 5], size=k).astype(np.uint16)
            vals = None if rng_ is None else rng.uniform(rng_[0], rng_[1], size=k)
            calls.append((int(rng.integers(0, 6)), voices, stage, param, vals, delays))  # block offsets 0..5: launches of 4 blocks, so some wait
        singles = [(int(rng.integers(0, n)), *targets[int(rng.integers(0, len(targets)))], int(rng.choice([0, 3, bs - 2]))) for _ in range(6)]
        script.append((calls, singles))
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("KNH_DEV_EVENTS", mode)
        g = make_gpu(knh, w)
        res = []
        for calls, singles in script:
            for off, voices, stage, param, vals, delays in calls:
                g.param_apply_many(voices, stage, param, L.VALUE_TRIGGER if vals is None else L.VALUE_FLOAT, vals, None, delays, block_offset=off)
            for voice, stage, param, rng_, delay in singles:
                g.set_delay_within_block_for_param(voice, stage, param, delay)
                g.param_apply(voice, stage, param, TRIGGER if rng_ is None else float(0.5 * (rng_[0] + rng_[1])))
            res.append(g.process_blocks(4)[0])
            res.append(g.process_block_voices()[1])  # and a single block with per-voice output
        outs[mode] = res
        g.close()
    for a, b in zip(outs["1"], outs["0"]):
        assert_bit_equal(a, b, "device-resolved against host-resolved")
    assert max(float(np.abs(x).max()) for x in outs["1"]) > 1e-5
