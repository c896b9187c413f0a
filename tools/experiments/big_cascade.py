"""A 4 075-stage FM cascade (680 oscillators per voice): build time of the frame-parallel kernel and parity of its two forms."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import knaster_amd
from knaster_amd import _lib as L, configs
from oracle import oracle_py
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 680
w = configs.fm_cascade(depth, 2, 128, add=19.0)
outs = {}
for form in ("jit", "interp"):
    os.environ["KNH_FRAME_JIT"] = "1" if form == "jit" else "0"
    t0 = time.perf_counter()
    b = knaster_amd.VoiceBank(w.stages, 2, w.sample_type, 1, L.MIX_LEFT_FOLD)
    for s, a in w.ctor.items(): b.set_ctor_args(s, a)
    b.init(48000, 128)
    t1 = time.perf_counter()
    outs[form] = [b.process_block()[0] for _ in range(2)]
    print(f"{form}: {len(w.stages)} stages, create+init {t1-t0:.2f} s")
    b.close()
o = oracle_py.OracleBank(w.stages, 2, w.sample_type, 1, True, False)
for s, a in w.ctor.items(): o.set_ctor_args(s, a)
o.init(48000, 128)
ref = [np.asarray(o.process_block()[0], dtype=np.float32) for _ in range(2)]  # (the left-fold mix of the two voices)
for blk in range(2):
    g, i, r = outs["jit"][blk], outs["interp"][blk], ref[blk]
    print(f"block {blk}: jit == interp {np.array_equal(g.view(np.uint32), i.view(np.uint32))}, jit == oracle {np.array_equal(g.view(np.uint32), r.view(np.uint32))}, "
          f"max |jit - oracle| {np.abs(g - r).max():.3g}, differing {np.count_nonzero(g.view(np.uint32) != r.view(np.uint32))} of {g.size}, peak {np.abs(r).max():.3g}")
