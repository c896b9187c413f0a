#!/usr/bin/env python3
"""Compiles the device kernels of random voices -- the chains and graphs of the seeded GPU tests, for more seeds -- with the
library's run-time fusion (hiprtc), each in a process of its own (tests/cpp/bin/jit_compile_check), WITHOUT a GPU: a compiler crash or a
compile error shows up here, not inside a host process at knh_bank_init.
usage: python tools/jit_compile_fuzz.py [seeds=300] [workers=8] [first_seed=0]"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import knaster_amd
from knaster_amd import _lib as L, configs
import test_gpu_dag as D
import test_gpu_random_chains as R

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
check = os.path.join(ROOT, "tests", "cpp", "bin", "jit_compile_check")  # built by tests/cpp/Makefile (__graft_entry__.build())


def signature(stages, sample_type):
    b = knaster_amd.VoiceBank(stages, 64, sample_type, 1, L.MIX_LEFT_FOLD)
    s = b.debug_signature()
    b.close()
    return s


jobs = set()
for seed in range(first, first + seeds):
    rng = np.random.default_rng(3000 + seed)
    st, _ = D.random_dag(rng, int(rng.integers(5, 13)))
    st = D.with_audio_rate_parameters(rng, st)
    jobs.add((signature(st, L.F32), "f32"))
    jobs.add((signature(st, L.F64), "f64"))
    rng = np.random.default_rng(1000 + seed)
    st, _ = D.random_dag(rng, int(rng.integers(4, 12)))
    jobs.add((signature(st, L.F32 if seed % 3 else L.F64), "f32" if seed % 3 else "f64"))
    w, _, _, _ = R.random_chain(seed)
    ty = "f64" if w.sample_type == L.F64 else "f32"
    jobs.add((signature(w.stages, w.sample_type), ty))
    jobs.add((signature(w.stages, w.sample_type), ty + " pipe"))  # ... and as the pipeline knh_bank_init builds for it
jobs = sorted(jobs)
print(len(jobs), "distinct kernels", flush=True)


# No core files from a crashing compiler: the limit is set once, here, and inherited (preexec_fn is not safe with threads).
# No device for the children: with none visible, loading the compiled kernel fails right behind the compile, which is all
# this tool wants -- wherever it runs, it is a compile-only check and never a crowd of processes on one GPU.
# The compiles run in the library's own process here (KNH_JIT_INPROCESS=1, no caches): this tool IS the helper process.
import resource
resource.setrlimit(resource.RLIMIT_CORE, (0, 0))
CHILD_ENV = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", KNH_JIT_INPROCESS="1", KNH_JIT_CACHE="0", AMD_COMGR_CACHE="0")


def run(job):
    sig, ty = job
    p = subprocess.run([check, sig] + [a for a in ty.split() if a != "f32"], cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1800,
                       env=CHILD_ENV)
    return job, p.returncode, p.stdout.decode(errors="replace")[-400:]


bad = 0
with ThreadPoolExecutor(workers) as ex:
    for n, (job, rc, out) in enumerate(ex.map(run, jobs)):
        if rc != 0:
            bad += 1
            print("rc", rc, job, out.strip().replace("\n", " | ")[:300], flush=True)
        if n % 50 == 49:
            print(n + 1, "compiled,", bad, "bad", flush=True)
print("done:", len(jobs), "kernels,", bad, "failed or crashed")
sys.exit(1 if bad else 0)
