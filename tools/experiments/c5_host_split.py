import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import knaster_amd
from knaster_amd import _lib as L, configs
w = configs.config("C5")
b = knaster_amd.VoiceBank(w.stages, w.n_voices, w.sample_type, w.out_channels, L.MIX_TREE, -1, False, 0)
for s, a in w.ctor.items(): b.set_ctor_args(s, a)
b.init(configs.SAMPLE_RATE, w.block_size)
blocks, launches = 32, 16
c5 = {}
for blk in range(blocks * (launches + 1)):
    e = configs.c5_events(w, blk)
    c5[blk] = None if e is None else b.prepare_many(e[0], e[1], e[2], e[3], e[4], None, e[5])
step = 0
t_apply = t_proc = 0.0
n_changes = 0
for l in range(launches + 1):
    t0 = time.perf_counter()
    for i in range(blocks):
        e = c5[step + i]
        if e is not None:
            b.param_apply_prepared(e, block_offset=i)
            if l: n_changes += w.n_voices
    t1 = time.perf_counter()
    b.process_blocks_device(blocks)
    t2 = time.perf_counter()
    step += blocks
    if l:
        t_apply += t1 - t0; t_proc += t2 - t1
b.synchronize()
print(f"changes {n_changes}: apply {t_apply/n_changes*1e9:.1f} ns/change, process (event assembly + launch) {t_proc/n_changes*1e9:.1f} ns/change; per launch apply {t_apply/launches*1e3:.3f} ms, process {t_proc/launches*1e3:.3f} ms")
