#!/usr/bin/env python3
"""Times every BASELINE.json workload on one GPU (not the headline bench; see bench.py).
Prints one JSON line per config: UGen-samples/s with outputs left in HBM, 32 blocks per launch."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import knaster_amd
from knaster_amd import _lib as L, configs


def run(name, n_voices=None, launches=16, blocks=32, allow_fma=False, host_threads=0):
    w = configs.config(name, n_voices=n_voices)
    b = knaster_amd.VoiceBank(w.stages, w.n_voices, w.sample_type, w.out_channels, L.MIX_TREE, -1, allow_fma, host_threads)
    for s, a in w.ctor.items():
        b.set_ctor_args(s, a)
    b.init(configs.SAMPLE_RATE, w.block_size)
    v = np.arange(w.n_voices, dtype=np.uint32)
    if w.restart:
        b.param_apply_many(v, w.restart[0], w.restart[1], L.VALUE_TRIGGER)
    if w.delay_times is not None:
        b.param_apply_many(v, 3, 0, L.VALUE_FLOAT, w.delay_times)
    step = [0]
    # the application's own work of deciding what changes when is not the engine's: C5's event arrays are made up front
    c5 = {}
    if name == "C5":
        for blk in range(blocks * (launches + 3)):
            e = configs.c5_events(w, blk)
            c5[blk] = None if e is None else b.prepare_many(e[0], e[1], e[2], e[3], e[4], None, e[5])

    def events(k):
        for i in range(k):
            if name == "C5":
                e = c5[step[0] + i]
                if e is not None:
                    b.param_apply_prepared(e, block_offset=i)
            elif w.release and (step[0] + i) % 64 == 32:
                b.param_apply_many(v, w.release[0], w.release[1], L.VALUE_TRIGGER, block_offset=i)
            elif w.restart and (step[0] + i) % 64 == 0 and step[0] + i > 0:
                b.param_apply_many(v, w.restart[0], w.restart[1], L.VALUE_TRIGGER, block_offset=i)
        step[0] += k
    for _ in range(3):  # untimed: first-use allocations (both of the alternating record / list buffers), the clock
        events(blocks)
        b.process_blocks_device(blocks)
    b.synchronize()
    b.timing_reset(True)
    t0 = time.perf_counter()
    for _ in range(launches):
        events(blocks)
        b.process_blocks_device(blocks)
    b.synchronize()
    dt = time.perf_counter() - t0
    kms, n = b.timing_read()
    # UGens per voice as SURVEY.md 8(d) counts them for the BASELINE.json configs (C2: SinNumeric + gain; C5: modulator,
    # scale/offset math, carrier); any other workload: the reference nodes its chain stands for
    ugens = {"C1": 3, "C2": 2, "C3": 4, "C4": 4, "C5": 3}.get(name) or knaster_amd.chain_ugen_count(w.stages)
    work = float(w.n_voices) * w.block_size * ugens * blocks * launches
    rd, wr = b.algorithmic_bytes_per_voice_block()
    if w.delay_times is not None:  # the ring: one sample read and one written per frame
        rd += 4 * w.block_size
        wr += 4 * w.block_size
    print(json.dumps({"config": name, "voices": w.n_voices, "block_size": w.block_size, "sample_type": "f64" if w.sample_type else "f32",
                      "ugens_per_voice": ugens, "allow_fma": allow_fma, "host_threads": max(1, host_threads), "ugen_samples_per_s": work / dt,
                      "kernel_only_ugen_samples_per_s": work / (kms * 1e-3), "us_per_block_kernel": kms * 1e3 / (n * blocks),
                      "hbm_algorithmic_GBps": (rd + wr) * w.n_voices * blocks * n / (kms * 1e-3) / 1e9}), flush=True)
    b.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ab":  # the configs that run on kernels other than the headline one
        run("C3", n_voices=65536)
        run("C3", n_voices=262144, launches=4)
        run("D3")
        run("D3", n_voices=65536)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "only":  # tools/bench_configs.py only C5 C3 C2:1024 ...
        for spec in sys.argv[2:]:
            name, _, nv = spec.partition(":")
            run(name, n_voices=int(nv) if nv else None)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "c5":  # the host-bound config against the number of host threads
        for k in (0, 2, 4, 8, 12):
            run("C5", host_threads=k, launches=16)
        sys.exit(0)
    run("C1")
    run("C2")
    run("C3")
    run("C3", allow_fma=True)
    run("C3", n_voices=65536)
    run("C3", n_voices=262144, launches=4)
    run("C4", n_voices=8192)
    run("C4")
    run("C5")
    run("C5", host_threads=4)
    run("B3")
    run("D3")
    run("D3", n_voices=65536)
    run("D3", n_voices=262144, launches=4)
