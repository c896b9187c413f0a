"""The N > 1 path on CPU: world_size 2, gloo.  Each rank owns the contiguous voice range the LIBRARY gives it
(knh_shard_voice_range: what knh_bank_create_rank uses); the only collective is the sum-reduce of the per-rank mixed blocks.
The library's banks need a GPU (tests/test_gpu_multi.py runs knh_bank_create_rank / _custom there, two ranks in two
processes included), so here the per-rank compute is stood in for by the CPU oracle and the routing by a few lines of test
code (`_RankStandIn`): what is under test is the voice ranges, the route-by-range rule and the reduce."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shard_range(n_voices, rank, world):
    """[lo, hi) of `rank`: knh_shard_voice_range (needs the library, not a GPU)"""
    from knaster_amd import shard_voice_range

    first, count = shard_voice_range(n_voices, rank, world)
    return first, first + count


def test_shard_ranges_partition_the_voices(knh):
    """knh_shard_voice_range: whole 64-voice groups, as even as they go, contiguous, covering every voice once."""
    for n in (1, 7, 10, 63, 64, 65, 130, 1000, 16384, 65536, 100000):
        for world in (1, 2, 3, 4, 8):
            nxt = 0
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert lo == nxt and (lo % 64 == 0 or hi == lo) and lo <= hi <= n
                nxt = hi
            assert nxt == n
    assert shard_range(65536, 3, 8) == (24576, 32768)  # C4: 8192 voices per GPU
    assert shard_range(16384 * 8, 7, 8) == (114688, 131072)
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


class _RankStandIn:
    """What the library's rank bank does with a parameter stream and a launch (rank_bank.hpp), in test code: calls for this
    rank's voices go to the local bank with local indices, the rest is some other rank's business; after the launch the
    mixed blocks are sum-reduced to rank 0."""

    def __init__(self, local, n_voices_total, rank, world, reduce_fn):
        self.local, self.rank, self.world, self.reduce_fn = local, rank, world, reduce_fn
        self.lo, self.hi = shard_range(n_voices_total, rank, world)

    def param_apply_many(self, voices, stages, params, kinds, fvalues=None, ivalues=None, delays=None, block_offset=0):
        v = np.asarray(voices, dtype=np.int64)
        mine = (v >= self.lo) & (v < self.hi)
        if not mine.any():
            return 0
        pick = lambda a: None if a is None else (np.broadcast_to(np.asarray(a), v.shape)[mine])
        self.local.param_apply_many((v[mine] - self.lo).astype(np.uint32), pick(stages), pick(params), pick(kinds), pick(fvalues),
                                    pick(ivalues), pick(delays), block_offset=block_offset)
        return int(mine.sum())

    def process_blocks(self, n_blocks):
        out, flags = self.local.process_blocks(n_blocks)
        return (self.reduce_fn(out) if self.world > 1 else out), flags


def _gloo_reduce(arr):
    import torch
    import torch.distributed as dist

    t = torch.as_tensor(np.ascontiguousarray(arr))
    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
    return t.numpy()


class _OracleLocal:
    """An object with VoiceBank's multi-block surface, computed by the oracle (tests only)."""

    def __init__(self, oracle_py, w, lo, hi):
        self.w = w
        self.bank = oracle_py.OracleBank(w.stages, hi - lo, w.sample_type, w.out_channels, True, False)
        for s, a in w.ctor.items():
            self.bank.set_ctor_args(s, a[lo:hi])
        self.bank.init(48000, w.block_size)
        self.future = {}

    def param_apply_many(self, voices, stages, params, kinds, fvalues=None, ivalues=None, delays=None, block_offset=0):
        self.future.setdefault(block_offset, []).append((voices, stages, params, kinds, fvalues, ivalues, delays))

    def process_blocks(self, n_blocks):
        outs = []
        for b in range(n_blocks):
            for call in self.future.pop(b, []):
                self.bank.param_apply_many(*call)
            outs.append(self.bank.process_block()[0])
        self.future = {k - n_blocks: v for k, v in self.future.items()}
        return np.stack(outs), 0


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from knaster_amd import _lib as L
    from knaster_amd import configs
    from oracle import oracle_py

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = configs.config("C3", n_voices=70, block_size=64)
        lo, hi = shard_range(w.n_voices, rank, world)
        sb = _RankStandIn(_OracleLocal(oracle_py, w, lo, hi), w.n_voices, rank, world, _gloo_reduce)
        all_voices = np.arange(w.n_voices)
        n = sb.param_apply_many(all_voices, w.restart[0], w.restart[1], L.VALUE_TRIGGER)            # block 0
        assert n == hi - lo
        sb.param_apply_many(all_voices[::2], w.release[0], w.release[1], L.VALUE_TRIGGER, block_offset=2)
        sb.param_apply_many(np.array([5, 40]), 0, 0, L.VALUE_FLOAT, np.array([880.0, 1760.0]), block_offset=1)
        mix, _ = sb.process_blocks(4)
        dist.barrier()
        if rank == 0:
            q.put(mix)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_mix_equals_single_bank(oracle):
    import torch.multiprocessing as mp

    from knaster_amd import _lib as L
    from knaster_amd import configs

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    mix = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-bank reference: the per-voice signals summed in f64
    w = configs.config("C3", n_voices=70, block_size=64)
    ref = _OracleLocal(oracle, w, 0, 70)
    ref.bank.close()
    ref.bank = oracle.OracleBank(w.stages, 70, w.sample_type, w.out_channels, False, True)
    for s_, a in w.ctor.items():
        ref.bank.set_ctor_args(s_, a)
    ref.bank.init(48000, 64)
    allv = np.arange(70)
    ref.bank.param_apply_many(allv, w.restart[0], w.restart[1], L.VALUE_TRIGGER)
    want = []
    for b in range(4):
        if b == 1:
            ref.bank.param_apply_many(np.array([5, 40]), 0, 0, L.VALUE_FLOAT, np.array([880.0, 1760.0]))
        if b == 2:
            ref.bank.param_apply_many(allv[::2], w.release[0], w.release[1], L.VALUE_TRIGGER)
        _, voices, _, _ = ref.bank.process_block()
        want.append(voices.astype(np.float64).sum(axis=0))
    want = np.stack(want)
    assert mix.shape == (4, 2, 64)
    assert np.max(np.abs(want)) > 1e-3
    for c in range(2):
        assert np.max(np.abs(mix[:, c, :].astype(np.float64) - want)) <= 1e-5
