import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import knaster_amd
from knaster_amd import _lib as L, configs
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
blocks = 32
w = configs.config(name)
b = knaster_amd.VoiceBank(w.stages, w.n_voices, w.sample_type, w.out_channels, L.MIX_TREE, -1, False, 0)
for s, a in w.ctor.items():
    b.set_ctor_args(s, a)
b.init(configs.SAMPLE_RATE, w.block_size)
c5 = {}
for blk in range(blocks * 40):
    e = configs.c5_events(w, blk) if name == "C5" else None
    c5[blk] = None if e is None else b.prepare_many(e[0], e[1], e[2], e[3], e[4], None, e[5])
step = 0
t_ev = []; t_pr = []
for it in range(30):
    t0 = time.perf_counter()
    for i in range(blocks):
        e = c5[step + i]
        if e is not None:
            b.param_apply_prepared(e, block_offset=i)
    step += blocks
    t1 = time.perf_counter()
    b.process_blocks_device(blocks)
    t2 = time.perf_counter()
    t_ev.append((t1 - t0) * 1e6); t_pr.append((t2 - t1) * 1e6)
b.synchronize()
print(name, "host us per launch: events", np.median(t_ev[5:]), "process call", np.median(t_pr[5:]), "sum", np.median(np.array(t_ev[5:]) + np.array(t_pr[5:])))
b.close()
