import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import knaster_amd
from knaster_amd import _lib as L, configs
from helpers import make_gpu, fire_all
res = {}
for mode in ("1", "0"):
    os.environ["KNH_RESIDENT"] = mode
    w = configs.config("C4", n_voices=200, block_size=96)
    g = make_gpu(knaster_amd, w, L.MIX_TREE)
    fire_all(g, 200, *w.restart)
    outs = []
    for b in range(3):
        o, f = g.process_block()
        outs.append(o.copy())
    print(mode, "stats", g.resident_stats(), "block0[:4]", outs[0][0, :4], "block1[:4]", outs[1][0, :4])
    res[mode] = outs
    g.close()
for b in range(3):
    d = np.abs(res["1"][b] - res["0"][b])
    print("block", b, "max diff", d.max(), "first differing frame", int(np.argmax(d[0] > 0)) if d.max() > 0 else -1, "n differing", int((d[0] > 0).sum()))
