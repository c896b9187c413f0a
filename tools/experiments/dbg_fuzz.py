import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import knaster_amd as knh
from knaster_amd import _lib as L, configs
from helpers import make_gpu
import test_gpu_event_fuzz as T
seed = int(sys.argv[1])
rng = np.random.default_rng(7000 + seed)
name = ["C5", "C3", "C5", "D3"][seed % 4]
n = int(rng.integers(65, 260)); bs = int(rng.choice([64, 128]))
w = configs.config(name, n_voices=n, block_size=bs, precise=int(rng.integers(1, 4)))
print(name, "n", n, "bs", bs, "precise", [s.delayed_changes_per_block for s in w.stages])
a = make_gpu(knh, w, L.MIX_LEFT_FOLD)
targets = []
for s in range(len(w.stages)):
    for p, pname in enumerate(a.stage_param_descriptions(s)):
        if pname in T.FLOATS or pname in T.TRIGGERS: targets.append((s, p, pname))
n_blocks = 12
plan = []
for blk in range(n_blocks):
    batches = []
    if blk == 0 and w.restart:
        v = np.arange(n, dtype=np.uint32); batches.append((v, w.restart[0], w.restart[1], L.VALUE_TRIGGER, None, None, "restart"))
    for _ in range(int(rng.integers(0, 4))):
        s, p, pname = targets[int(rng.integers(0, len(targets)))]
        m = int(rng.integers(1, 2 * n)); v = rng.integers(0, n, m).astype(np.uint32)
        if rng.random() < 0.5: v = np.sort(v)
        delays = rng.integers(0, bs, m).astype(np.uint16) if rng.random() < 0.8 else None
        if pname in T.TRIGGERS: batches.append((v, s, p, L.VALUE_TRIGGER, None, delays, pname))
        else:
            lo, hi = T.FLOATS[pname]; batches.append((v, s, p, L.VALUE_FLOAT, rng.uniform(lo, hi, m), delays, pname))
    plan.append(batches)
for blk, bt in enumerate(plan):
    print("block", blk, [(s, nm, len(v), "sorted" if np.all(np.diff(v.astype(np.int64)) >= 0) else "unsorted", "delays" if d is not None else "no delays") for (v, s, p, k, f, d, nm) in bt])
# which launch splits reproduce: try every launch layout of the failing region by brute force: block-by-block per-voice vs multi
def run(layout):
    g = make_gpu(knh, w, L.MIX_LEFT_FOLD if layout is None else L.MIX_TREE)
    outs = []
    blk = 0
    lay = [1] * n_blocks if layout is None else layout
    for k in lay:
        for i in range(k):
            for (v, s, p, kind, f, d, nm) in plan[blk + i]:
                g.param_apply_many(v, s, p, kind, f, None, d, block_offset=i)
        if k == 1 and layout is None:
            outs.append(g.process_block_voices()[1])
        else:
            o = g.process_blocks(k)[0]
            outs += [o[i] for i in range(k)]
        blk += k
    g.close()
    return outs
ref = run([1] * n_blocks)
for lay in ([4, 4, 4], [2] * 6, [3] * 4, [1, 1, 2] * 3, [1, 3] * 3, [2, 1, 1] * 3):
    got = run(lay)
    bad = [i for i in range(n_blocks) if not np.array_equal(got[i].view(np.uint32), ref[i].view(np.uint32))]
    print("layout", lay, "differing blocks", bad)
print("-- which batch of block 2")
full = plan[2]
import itertools
for keep in ([0], [1], [2], [0, 1], [0, 2], [1, 2]):
    plan[2] = [full[i] for i in keep]
    ref = run([1] * n_blocks); got = run([4, 4, 4])
    bad = [i for i in range(n_blocks) if not np.array_equal(got[i].view(np.uint32), ref[i].view(np.uint32))]
    print("block 2 batches", keep, "-> differing", bad[:3])
plan[2] = full
