"""The sharding arithmetic of a voice bank spread over the GPUs of one node, restated in Python for hosts and tests.

The product's multi-GPU path lives in the library: knh_bank_create_rank (one process per GPU, RCCL ncclReduce of
each launch's stereo blocks, knaster_amd/csrc/rank_bank.hpp + comm.hip) and knh_bank_create_multi_device (one
process, peer-to-peer copies + a sum in range order, host_shards.hpp); bench.py runs the former.  This module holds
`shard_range` / `owner_of` -- the same voice ranges as knh_shard_voice_range, checked against it by
tests/test_sharded_gloo.py -- and `ShardedBank`, the same control flow (route by voice, process, sum-reduce) over any
object with a bank's surface, which the CPU tests drive with world_size 2 over gloo and a CPU stand-in as the local bank.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Optional, Tuple

import numpy as np


def shard_range(n_voices: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous voice range [lo, hi) owned by `rank`: the 64-voice groups (one wavefront each) are dealt out as evenly
    as they go -- rank r of R owns groups [r*G/R, (r+1)*G/R).  For N a multiple of 64*R this is [r*N/R, (r+1)*N/R).
    Mirrors knh_shard_voice_range (rank_bank.hpp)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    groups = (n_voices + 63) // 64
    return min(groups * rank // world * 64, n_voices), min(groups * (rank + 1) // world * 64, n_voices)


def owner_of(voice: int, n_voices: int, world: int) -> Tuple[int, int]:
    """(rank, local voice index) of a global voice: parameter/event routing on the host."""
    if not (0 <= voice < n_voices):
        raise IndexError("voice out of range")
    r = min(world - 1, (voice * world) // max(n_voices, 1))
    while shard_range(n_voices, r, world)[1] <= voice:
        r += 1
    while shard_range(n_voices, r, world)[0] > voice:
        r -= 1
    return r, voice - shard_range(n_voices, r, world)[0]


@dataclass
class ShardedBank:
    """`local` is this rank's bank (knaster_amd.VoiceBank on a GPU; any object with the same
    param_apply_many / process_blocks surface in tests).  Global voice indices in, rank-0 mix out."""
    local: object
    n_voices_total: int
    rank: int
    world: int
    reduce_fn: Optional[Callable] = None  # (array_like [n_blocks, ch, B]) -> reduced on rank 0 (identity if world == 1)

    @property
    def lo(self) -> int:
        return shard_range(self.n_voices_total, self.rank, self.world)[0]

    @property
    def hi(self) -> int:
        return shard_range(self.n_voices_total, self.rank, self.world)[1]

    def param_apply_many(self, voices, stages, params, kinds, fvalues=None, ivalues=None, delays=None, block_offset=0):
        """Routes the calls addressed to this rank's voices; the others are some other rank's business."""
        v = np.asarray(voices, dtype=np.int64)
        mine = (v >= self.lo) & (v < self.hi)
        if not mine.any():
            return 0
        pick = lambda a: None if a is None else (np.broadcast_to(np.asarray(a), v.shape)[mine])
        self.local.param_apply_many((v[mine] - self.lo).astype(np.uint32), pick(stages), pick(params), pick(kinds), pick(fvalues),
                                    pick(ivalues), pick(delays), block_offset=block_offset)
        return int(mine.sum())

    def process_blocks(self, n_blocks: int):
        """-> (mix [n_blocks, ch, B] valid on rank 0, local flags)"""
        out, flags = self.local.process_blocks(n_blocks)
        if self.world > 1:
            if self.reduce_fn is None:
                raise RuntimeError("world > 1 needs a reduce_fn")
            out = self.reduce_fn(out)
        return out, flags


def torch_reduce_fn(dst: int = 0, device: Optional[str] = None):
    """Sum-reduce to `dst` with torch.distributed (RCCL on GPUs, gloo on CPU)."""
    import torch
    import torch.distributed as dist

    def fn(arr):
        t = torch.as_tensor(np.ascontiguousarray(arr), device=device)
        dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM)
        return t.cpu().numpy()
    return fn
