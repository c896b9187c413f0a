import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import knaster_amd as knh
from knaster_amd import _lib as L, configs
from helpers import make_gpu, make_oracle
from oracle import oracle_py
import test_gpu_event_fuzz as T
from test_gpu_random_chains import random_chain
names = {getattr(L, k): k[6:] for k in dir(L) if k.startswith("STAGE_") and isinstance(getattr(L, k), int)}
for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(65, 260)); bs = int(rng.choice([64, 128, 96]))
    w, _r, _c, triggers = random_chain(1000 + seed)
    n, bs = w.n_voices, w.block_size
    first = [(s, restart) for (s, restart, _rel) in triggers]
    print("seed", seed, "n", n, "bs", bs, "f64" if w.sample_type else "f32", [(names[x.kind], x.delayed_changes_per_block) for x in w.stages])
    a = make_gpu(knh, w, L.MIX_LEFT_FOLD); o = make_oracle(oracle_py, w)
    targets = []
    for s in range(len(w.stages)):
        for p, pname in enumerate(a.stage_param_descriptions(s)):
            if pname in T.FLOATS or pname in T.TRIGGERS: targets.append((s, p, pname))
    done = False
    for blk in range(12):
        batches = []
        if blk == 0:
            for (s0, p0) in first: batches.append((np.arange(n, dtype=np.uint32), s0, p0, L.VALUE_TRIGGER, None, None, "first"))
        for _ in range(int(rng.integers(0, 6))):
            s, p, pname = targets[int(rng.integers(0, len(targets)))]
            m = int(rng.integers(1, 2 * n)); v = rng.integers(0, n, m).astype(np.uint32)
            if rng.random() < 0.5: v = np.sort(v)
            delays = rng.integers(0, bs, m).astype(np.uint16) if rng.random() < 0.8 else None
            if pname in T.TRIGGERS: batches.append((v, s, p, L.VALUE_TRIGGER, None, delays, pname))
            else:
                lo, hi = T.FLOATS[pname]; batches.append((v, s, p, L.VALUE_FLOAT, rng.uniform(lo, hi, m), delays, pname))
        for (v, s, p, kind, f, d, nm) in batches:
            for bank in (a, o): bank.param_apply_many(v, s, p, kind, f, None, d)
        av = a.process_block_voices()[1]; ov = np.asarray(o.process_block()[1])
        bad = np.where(np.any(av.view(np.uint8).reshape(n, -1) != ov.astype(av.dtype).view(np.uint8).reshape(n, -1), axis=1))[0]
        if len(bad) and not done:
            done = True
            print("  first bad block", blk, "voices", bad[:10], "of", len(bad))
            for (v, s, p, kind, f, d, nm) in batches:
                hit = np.intersect1d(v, bad)
                print("     batch stage", s, nm, "m", len(v), "delays" if d is not None else "no delays", "hits bad voices:", len(hit), "mult of first bad:", int(np.sum(v == bad[0])),
                      "its delays:", (d[v == bad[0]] if d is not None else None))
            fb = bad[0]
            diff = np.where(av[fb] != ov[fb].astype(av.dtype))[0]
            print("     voice", fb, "differs at frames", diff[:8], "...", len(diff), "gpu", av[fb][diff[:3]], "oracle", ov[fb][diff[:3]])
    a.close(); o.close()
