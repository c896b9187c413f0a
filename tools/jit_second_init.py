#!/usr/bin/env python3
"""knh_bank_init of a graph-shaped voice of N stages twice, each in a process of its own: the first compiles (in the helper
process) and writes the code-object cache, the second loads from it.  VERDICT r03 item 3: "second init of the 100-stage voice
< 100 ms".  usage: python tools/jit_second_init.py [stages=100]   (needs a GPU: init loads the module)"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
CHILD = f"""
import sys, time, ctypes as C
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
import numpy as np
import knaster_amd
from knaster_amd import _lib as L
import test_gpu_dag as D
rng = np.random.default_rng(77)
st, ctor = D.random_dag(rng, {N})
b = knaster_amd.VoiceBank(st, 128, L.F32, 1, L.MIX_TREE)
for s, a in ctor.items():
    b.set_ctor_args(s, np.tile(np.asarray(a, dtype=np.float64), (128, 1)))
# (the HIP runtime's own start-up -- context, first allocation, first module -- is not this library's: a one-voice bank of a
# pre-built chain comes up first)
w = knaster_amd.VoiceBank([knaster_amd.bank.Stage(L.STAGE_SIN_WT), knaster_amd.bank.Stage(L.STAGE_MUL_CONST)], 1, L.F32, 1, L.MIX_TREE)
w.set_ctor_args(0, np.full((1, 1), 440.0)); w.set_ctor_args(1, np.full((1, 1), 0.2))
t0 = time.perf_counter(); w.init(48000, 64); warm = time.perf_counter() - t0
w.process_block()
t0 = time.perf_counter(); b.init(48000, 64); dt = time.perf_counter() - t0
print('WARMUP_MS', warm * 1e3)
b2 = knaster_amd.VoiceBank(st, 128, L.F32, 1, L.MIX_TREE)
for s, a in ctor.items():
    b2.set_ctor_args(s, np.tile(np.asarray(a, dtype=np.float64), (128, 1)))
t0 = time.perf_counter(); b2.init(48000, 64); print('SAME_PROCESS_AGAIN_MS', (time.perf_counter() - t0) * 1e3)
v = [C.c_uint64(0) for _ in range(4)]
L.load().knh_jit_stats(*[C.byref(x) for x in v])
out, _ = b.process_block()
print('RESULT', len(st), dt, *[x.value for x in v], bool(np.isfinite(out).all()))
"""
with tempfile.TemporaryDirectory() as d:
    env = dict(os.environ, KNH_JIT_CACHE_DIR=d, AMD_COMGR_CACHE="0")
    for run in ("first (compiles)", "second (from the cache)"):
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        if not line:
            print(p.stdout[-2000:])
            sys.exit(1)
        f = line[0].split()
        for l in p.stdout.splitlines():
            if l.startswith(("WARMUP_MS", "SAME_PROCESS_AGAIN_MS")):
                print("   ", l)
        print(f"{run}: {f[1]} stages, knh_bank_init {float(f[2]) * 1e3:.1f} ms; jit stats compiles={f[3]} cache_hits={f[4]} helper_runs={f[5]} failures={f[6]}; output finite: {f[7]}")
