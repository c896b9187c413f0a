"""How much of a bench launch period is host work (event scheduling + enqueue) and how much is the GPU."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import knaster_amd
from knaster_amd import _lib as L, configs

w = configs.config("C3")
b = knaster_amd.VoiceBank(w.stages, w.n_voices, w.sample_type, 2, L.MIX_TREE)
for s, a in w.ctor.items():
    b.set_ctor_args(s, a)
b.init(48000, 512)
v = np.arange(w.n_voices, dtype=np.uint32)
K = 64
t_sched = t_call = 0.0
b.timing_reset(True)
t0 = time.perf_counter()
n = 40
for launch in range(n):
    a0 = time.perf_counter()
    b.param_apply_many(v, w.restart[0], w.restart[1], L.VALUE_TRIGGER, block_offset=0)
    b.param_apply_many(v, w.release[0], w.release[1], L.VALUE_TRIGGER, block_offset=32)
    a1 = time.perf_counter()
    b.process_blocks_device(K)
    a2 = time.perf_counter()
    t_sched += a1 - a0
    t_call += a2 - a1
b.synchronize()
wall = time.perf_counter() - t0
kms, k = b.timing_read()
print(f"per 64-block launch: wall {1e3 * wall / n:.3f} ms, kernel {kms / k:.3f} ms, host scheduling {1e3 * t_sched / n:.3f} ms, "
      f"process_blocks_device call {1e3 * t_call / n:.3f} ms")
