import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import knaster_amd
from knaster_amd import _lib as L, configs
from knaster_amd.bank import Stage
from oracle import oracle_py
import test_gpu_dag as T
from helpers import make_gpu, make_oracle
names = {getattr(L, k): k[6:] for k in dir(L) if k.startswith("STAGE_") and isinstance(getattr(L, k), int)}
for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(4000 + seed)
    st, ctor = T.arithmetic_dag(rng, int(rng.integers(5, 40)))
    n = int(rng.integers(1, 70)); bs = int(rng.choice([16, 48, 64, 100, 256]))
    w = configs.Workload(f"interp{seed}", st, n, bs, L.F32 if seed % 3 else L.F64, 1)
    w.ctor = {s: np.tile(np.asarray(a, dtype=np.float64), (n, 1)) * (1.0 + 0.01 * np.arange(n)).reshape(n, 1) for s, a in ctor.items()}
    if not any(x.kind in (L.STAGE_MATH_MUL, L.STAGE_MATH_ADD, L.STAGE_MATH_SUB) for x in st) and sum(x.kind == L.STAGE_SIN_WT for x in st) < 2:
        st.append(Stage(L.STAGE_SIN_WT)); w.ctor[len(st) - 1] = np.full((n, 1), 333.0)
        st.append(Stage(L.STAGE_MATH_ADD, input=len(st), input2=len(st) - 1))
    print("seed", seed, "n", n, "bs", bs, [(i + 1, names[x.kind], x.input, x.input2) for i, x in enumerate(st)])
    o = oracle_py.OracleBank(w.stages, n, w.sample_type, 1, True, True)
    for s, a in w.ctor.items(): o.set_ctor_args(s, a)
    o.init(48000, bs)
    ov = np.asarray(o.process_block()[1])
    for form in ("fused", "interp", "frame"):
        os.environ["KNH_INTERP"] = "0" if form == "fused" else "1"
        os.environ["KNH_FRAME_JIT"] = "0" if form == "interp" else "1"
        g = make_gpu(knaster_amd, w, L.MIX_LEFT_FOLD)
        gv = g.process_block_voices()[1]
        print("   ", form, "== oracle:", bool(np.array_equal(gv.view(np.uint8), ov.astype(gv.dtype).view(np.uint8))), gv[0, :3], ov[0, :3])
        g.close()
