"""Where the wall time of C5 (4096 voices, one sample-accurate change per voice every second block) goes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import knaster_amd
from knaster_amd import _lib as L, configs

w = configs.config("C5")
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 0  # host threads (knh_bank_create_sharded)
b = knaster_amd.VoiceBank(w.stages, w.n_voices, w.sample_type, w.out_channels, L.MIX_TREE, host_threads=threads)
for s, a in w.ctor.items():
    b.set_ctor_args(s, a)
b.init(configs.SAMPLE_RATE, w.block_size)
blocks = 32
t_gen = t_apply = t_proc = 0.0
step = 0
for launch in range(6):
    for i in range(blocks):
        t0 = time.perf_counter()
        e = configs.c5_events(w, step + i)
        if e is not None:
            e = b.prepare_many(e[0], e[1], e[2], e[3], e[4], None, e[5])
        t1 = time.perf_counter()
        if e is not None:
            b.param_apply_prepared(e, block_offset=i)
        t2 = time.perf_counter()
        if launch:
            t_gen += t1 - t0
            t_apply += t2 - t1
    step += blocks
    t0 = time.perf_counter()
    b.process_blocks_device(blocks)
    t1 = time.perf_counter()
    b.synchronize()
    t2 = time.perf_counter()
    if launch:
        t_proc += t1 - t0
        if launch == 5: print(f"launch {launch}: process_blocks_device call {1e3 * (t1 - t0):.2f} ms, sync {1e3 * (t2 - t1):.2f} ms")
n = 5
print(f"host_threads {max(1, threads)}, per 32-block launch: event generation and array conversion (python) {1e3 * t_gen / n:.2f} ms, param_apply_many {1e3 * t_apply / n:.2f} ms, "
      f"process_blocks_device (host part) {1e3 * t_proc / n:.2f} ms; events per launch {16 * w.n_voices}")
