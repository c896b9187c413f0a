"""The per-block call on a RESIDENT kernel (voice_chain.hpp `Resident`, knh_bank_resident_stats): knh_bank_process_block is the
call the reference makes once per block (Task::run, knaster_graph/src/task.rs:25-31); where the bank's kernel form allows it the
kernel stays on its CUs between calls.  Everything here is held to the launch-per-call path (KNH_RESIDENT=0), bit for bit --
which the parity suites hold to the oracle."""
import time

import numpy as np
import pytest

from helpers import assert_bit_equal, fire_all, make_gpu
from knaster_amd import _lib as L
from knaster_amd import configs
from knaster_amd.bank import TRIGGER

pytestmark = pytest.mark.gpu


def _script(bank, w, block, rng_seed=5):
    """the block's parameter traffic: note on, changes on a few voices, note off, and on again"""
    n = w.n_voices
    v = np.arange(n, dtype=np.uint32)
    if block == 0:
        fire_all(bank, n, *w.restart)
    if block == 3 and w.stages[2].kind == L.STAGE_SVF:
        some = v[::7]
        bank.param_apply_many(some, 2, 0, L.VALUE_FLOAT, 400.0 + (some % 900))  # SvfFilter cutoffs
    if block == 4:
        bank.param_apply(n - 1, 0, 0, 333.0)
    if block == 6 and w.release:
        bank.param_apply_many(v, w.release[0], w.release[1], L.VALUE_TRIGGER)
    if block == 9:
        fire_all(bank, n, *w.restart)


def _render(knh, monkeypatch, name, n, bs, resident, blocks=12, splits=(), pause_after=None, other_calls=False):
    monkeypatch.setenv("KNH_RESIDENT", "1" if resident else "0")
    w = configs.config(name, n_voices=n, block_size=bs)
    g = make_gpu(knh, w, L.MIX_TREE)
    outs, flags = [], []
    for b in range(blocks):
        if w.restart:
            _script(g, w, b)
        if b in splits:  # the block in two partial calls (a splitting wrapper around the node)
            cut = bs // 3 + 1
            out = np.zeros((w.out_channels, bs), dtype=g.dtype)
            o1, f1 = g.process_block(cut, 0)
            o2, f2 = g.process_block(bs - cut, cut)
            out[:, :cut] = o1[:, :cut]
            out[:, cut:] = o2[:, cut:]
            f = f2
        else:
            out, f = g.process_block()
        outs.append(out.copy())
        flags.append(f)
        if pause_after is not None and b == pause_after:
            time.sleep(0.05)
        if other_calls and b == 5:
            outs.append(g.read_done_frames().astype(np.float64).reshape(1, -1)[:, :1])  # (an entry point that needs the device state)
            many, _ = g.process_blocks(3)
            outs.extend(list(many))
    done = g.read_done_frames()
    stats = g.resident_stats()
    g.close()
    return outs, flags, done, stats


# (C1 and the graph voice run on the one-wavefront kernel, the others on pipelines: pre-built in-place and mixer forms, a Fan pipeline, f64, Pan2)
CASES = [("C3", 70, 64), ("C3", 2100, 128), ("C3", 16384, 512), ("C4", 200, 96), ("M1", 600, 64), ("C2", 1024, 256), ("P3", 300, 64), ("C1", 1, 64),
         ("C1", 130, 200)]


@pytest.mark.parametrize("name,n,bs", CASES)
def test_resident_calls_equal_a_launch_per_call(knh, monkeypatch, name, n, bs):
    """Note cycles, parameter changes and two split blocks, block by block: same samples, flags and done frames with the
    kernel resident as with a launch per call; and the resident kernel really served the calls from ONE launch."""
    blocks = 12 if n < 10000 else 6
    a = _render(knh, monkeypatch, name, n, bs, True, blocks=blocks, splits=(2, 7))
    b = _render(knh, monkeypatch, name, n, bs, False, blocks=blocks, splits=(2, 7))
    for k, (x, y) in enumerate(zip(a[0], b[0])):
        assert_bit_equal(x, y, f"{name} block {k}")
    assert a[1] == b[1]
    np.testing.assert_array_equal(a[2], b[2])
    assert np.abs(np.stack(b[0])).max() > 1e-5
    assert b[3] == (0, 0)
    n_split = sum(1 for k in (2, 7) if k < blocks)
    assert a[3] == (blocks + n_split, 1), a[3]  # every call (the split blocks in two calls each) on one launch


def test_an_idle_resident_kernel_leaves_by_itself_and_the_next_call_starts_another(knh, monkeypatch):
    monkeypatch.setenv("KNH_RESIDENT_IDLE_US", "300")
    a = _render(knh, monkeypatch, "C3", 700, 64, True, pause_after=4)
    b = _render(knh, monkeypatch, "C3", 700, 64, False)
    for k, (x, y) in enumerate(zip(a[0], b[0])):
        assert_bit_equal(x, y, f"block {k}")
    np.testing.assert_array_equal(a[2], b[2])
    assert a[3][0] == 12 and a[3][1] >= 2, a[3]  # the pause outlasted the kernel's patience: a second launch


def test_other_entry_points_between_resident_calls(knh, monkeypatch):
    """read_done_frames and a multi-block launch in the middle: the resident kernel hands the state back, the next per-block
    call starts a new one."""
    a = _render(knh, monkeypatch, "C3", 700, 64, True, other_calls=True)
    b = _render(knh, monkeypatch, "C3", 700, 64, False, other_calls=True)
    assert len(a[0]) == len(b[0])
    for k, (x, y) in enumerate(zip(a[0], b[0])):
        assert_bit_equal(np.asarray(x), np.asarray(y), f"item {k}")
    assert a[3][1] == 2, a[3]


def test_two_banks_take_turns_on_the_device(knh, monkeypatch):
    """Two banks of one process called alternately: each launch asks the other bank's resident kernel to leave (it holds every
    CU's LDS); after that the banks stay with a launch per call for a while.  Same samples as ever."""
    w = configs.config("C3", n_voices=500, block_size=64)
    g1, g2, ref = make_gpu(knh, w, L.MIX_TREE), make_gpu(knh, w, L.MIX_TREE), make_gpu(knh, w, L.MIX_TREE)
    for b in range(6):
        for bank in (g1, g2, ref):
            _script(bank, w, b)
        monkeypatch.setenv("KNH_RESIDENT", "0")  # (a bank decides at its first process call)
        want, _ = ref.process_block()
        monkeypatch.setenv("KNH_RESIDENT", "1")
        o1, _ = g1.process_block()
        o2, _ = g2.process_block()
        assert_bit_equal(o1, want, f"bank 1 block {b}")
        assert_bit_equal(o2, want, f"bank 2 block {b}")
    assert g1.resident_stats()[1] >= 1
    for bank in (g1, g2, ref):
        bank.close()


def test_range_events_for_parts_of_the_bank(knh, monkeypatch):
    """A batch of envelope triggers for neighbouring voices reaches a resident pipeline kernel as ONE range event (ResCall,
    voice_chain.hpp) instead of an event per voice: ranges that cover parts of the bank, overlap (release, then restart, for
    the voices in both), arrive beside single changes (before and after them: the whole call is spelled out per voice then),
    more of them than a command can name, and in a block that is rendered in two partial calls."""
    def script(bank, w, b):
        n = w.n_voices
        v = np.arange(n, dtype=np.uint32)
        on, off = w.restart, w.release
        if b == 0:
            bank.param_apply_many(v[: n // 2], on[0], on[1], L.VALUE_TRIGGER)
            bank.param_apply_many(v[n // 2:], on[0], on[1], L.VALUE_TRIGGER)
        if b == 2:
            bank.param_apply_many(v[100:300], off[0], off[1], L.VALUE_TRIGGER)
            bank.param_apply_many(v[200:250], on[0], on[1], L.VALUE_TRIGGER)
        if b == 4:
            bank.param_apply_many(v, on[0], on[1], L.VALUE_TRIGGER)
            bank.param_apply(5, 0, 0, 500.0)
        if b == 5:
            bank.param_apply(7, 0, 0, 600.0)
            bank.param_apply_many(v, off[0], off[1], L.VALUE_TRIGGER)
        if b == 7:
            for i in range(20):
                bank.param_apply_many(v[i * 30: i * 30 + 40], on[0] if i % 3 else off[0], on[1] if i % 3 else off[1], L.VALUE_TRIGGER)
        if b == 8:  # (a split block: the triggers belong to its first part)
            bank.param_apply_many(v[3:690], on[0], on[1], L.VALUE_TRIGGER)
        if b == 9:  # knh_bank_param_apply_range: triggers (a range event as it stands) and a float for a run of voices (spelled out)
            bank.param_apply_range(40, 660, on[0], on[1], L.VALUE_TRIGGER)
            bank.param_apply_range(100, 200, 0, 0, L.VALUE_FLOAT, 321.0)
        if b == 10:
            bank.param_apply_many(v[::-1].copy(), off[0], off[1], L.VALUE_TRIGGER)  # falling order: not a range

    def render(resident):
        monkeypatch.setenv("KNH_RESIDENT", "1" if resident else "0")
        w = configs.config("C3", n_voices=700, block_size=64)
        g = make_gpu(knh, w, L.MIX_TREE)
        outs = []
        for b in range(12):
            script(g, w, b)
            if b == 8:
                o1, _ = g.process_block(20, 0)
                o2, _ = g.process_block(44, 20)
                out = o1.copy()
                out[:, 20:] = o2[:, 20:]
            else:
                out, _ = g.process_block()
            outs.append(out.copy())
        done, stats = g.read_done_frames(), g.resident_stats()
        g.close()
        return outs, done, stats
    a, b = render(True), render(False)
    for k, (x, y) in enumerate(zip(a[0], b[0])):
        assert_bit_equal(x, y, f"block {k}")
    np.testing.assert_array_equal(a[1], b[1])
    assert np.abs(np.stack(b[0])).max() > 1e-5
    assert a[2] == (13, 1), a[2]


def test_param_apply_range_is_the_batch_it_stands_for(knh, monkeypatch):
    """knh_bank_param_apply_range(v0, v1, ..) against knh_bank_param_apply_many for the same voices: triggers, floats, a filter
    parameter (per-voice coefficient patches), an empty range, bad ranges and kinds refused with the batch's codes."""
    monkeypatch.setenv("KNH_RESIDENT", "0")
    w = configs.config("C3", n_voices=300, block_size=64)
    a, b = make_gpu(knh, w, L.MIX_TREE), make_gpu(knh, w, L.MIX_TREE)
    v = np.arange(300, dtype=np.uint32)
    for blk in range(8):
        if blk == 0:
            a.param_apply_range(0, 300, w.restart[0], w.restart[1], L.VALUE_TRIGGER)
            b.param_apply_many(v, w.restart[0], w.restart[1], L.VALUE_TRIGGER)
        if blk == 2:
            a.param_apply_range(10, 200, 2, 0, L.VALUE_FLOAT, 1234.5)   # SvfFilter cutoff
            b.param_apply_many(v[10:200], 2, 0, L.VALUE_FLOAT, 1234.5)
            a.param_apply_range(5, 5, 0, 0, L.VALUE_FLOAT, 99.0)        # nobody
        if blk == 4:
            a.param_apply_range(7, 23, w.release[0], w.release[1], L.VALUE_TRIGGER)  # a short run, and a single voice
            b.param_apply_many(v[7:23], w.release[0], w.release[1], L.VALUE_TRIGGER)
            a.param_apply_range(250, 251, 0, 0, L.VALUE_FLOAT, 777.0)
            b.param_apply(250, 0, 0, 777.0)
        if blk == 5:
            for bad in ((290, 301, 0, 0, L.VALUE_FLOAT), (0, 10, 9, 0, L.VALUE_FLOAT), (0, 10, 0, 0, L.VALUE_TRIGGER), (20, 10, 0, 0, L.VALUE_FLOAT)):
                with pytest.raises(L.KnasterHipError):
                    a.param_apply_range(*bad, 1.0)
            b.param_apply_many(v[290:300], 0, 0, L.VALUE_FLOAT, 1.0)  # (the in-range part of the first bad call is applied, as a batch's would be)
        x, fa = a.process_block()
        y, fb = b.process_block()
        assert_bit_equal(x, y, f"block {blk}")
        assert fa == fb
    np.testing.assert_array_equal(a.read_done_frames(), b.read_done_frames())
    a.close()
    b.close()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("KNH_TEST_SEEDS", "10"))))
def test_random_traffic_on_a_resident_kernel(knh, monkeypatch, seed):
    """Seeded parameter traffic -- batches on random voices, runs of neighbouring voices, whole-bank triggers through every entry
    point, single calls, the same change for every voice call by call, blocks rendered in two or three partial calls -- on a
    resident kernel against a launch per call: same samples, flags and done frames.  (A call's events reach a resident kernel
    as per-voice lists, as range events, or as lists that turn out to be ranges: all three come up here.)"""
    rng0 = np.random.default_rng(9000 + seed)
    name, n, bs = [("C3", 700, 64), ("C3", 1300, 128), ("C4", 260, 96), ("P3", 300, 64)][seed % 4]
    blocks = 14

    def render(resident):
        rng = np.random.default_rng(4000 + seed)
        monkeypatch.setenv("KNH_RESIDENT", "1" if resident else "0")
        w = configs.config(name, n_voices=n, block_size=bs)
        g = make_gpu(knh, w, L.MIX_TREE)
        v = np.arange(n, dtype=np.uint32)
        outs, flags = [], []
        for b in range(blocks):
            for _ in range(int(rng.integers(0, 4))):
                kind = int(rng.integers(0, 7))
                on = w.restart if rng.random() < 0.6 else (w.release or w.restart)
                if kind == 0:
                    g.param_apply_range(0, n, on[0], on[1], L.VALUE_TRIGGER)
                elif kind == 1:
                    a = int(rng.integers(0, n - 20))
                    g.param_apply_range(a, int(rng.integers(a + 1, n + 1)), on[0], on[1], L.VALUE_TRIGGER)
                elif kind == 2:
                    a = int(rng.integers(0, n - 40))
                    g.param_apply_many(v[a:a + int(rng.integers(16, 200))], on[0], on[1], L.VALUE_TRIGGER)
                elif kind == 3:
                    sel = rng.choice(n, size=int(rng.integers(1, 60)), replace=False).astype(np.uint32)
                    g.param_apply_many(sel, 0, 0, L.VALUE_FLOAT, 100.0 + 900.0 * rng.random(sel.size))
                elif kind == 4:
                    for vv in rng.choice(n, size=int(rng.integers(1, 8)), replace=False):
                        g.param_apply(int(vv), on[0], on[1], TRIGGER)
                elif kind == 5:  # the same change for every voice, one call each: a list that is a range
                    f = 200.0 + 50.0 * int(rng.integers(0, 20))
                    for vv in range(n):
                        g.param_apply(vv, 0, 0, f)
                else:
                    g.param_apply_range(0, n, 0, 1, L.VALUE_FLOAT, float(rng.random()))  # phase offsets, spelled out per voice
            cuts = sorted(set(int(c) for c in rng.integers(1, bs, size=int(rng.integers(0, 3))))) if rng.random() < 0.3 else []
            out = np.zeros((w.out_channels, bs), dtype=g.dtype)
            f = 0
            for lo, hi in zip([0] + cuts, cuts + [bs]):
                o, f = g.process_block(hi - lo, lo)
                out[:, lo:hi] = o[:, lo:hi]
            outs.append(out)
            flags.append(f)
        done = g.read_done_frames()
        g.close()
        return outs, flags, done
    assert rng0 is not None
    a, b = render(True), render(False)
    for k, (x, y) in enumerate(zip(a[0], b[0])):
        assert_bit_equal(x, y, f"seed {seed} block {k}")
    assert a[1] == b[1]
    np.testing.assert_array_equal(a[2], b[2])
