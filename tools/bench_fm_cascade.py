#!/usr/bin/env python3
"""The reference's "256 FM cascade" bench (knaster_benchmarks/benches/graph_dsp_performance.rs:37-72) as graph-shaped voices:
    python tools/bench_fm_cascade.py [depth=256] [voices=1,64,4096] [block=128]
Prints, per voice count: fusion (hiprtc) time at init, kernel time per block, UGen-samples/s (UGens = the reference nodes the
voice stands for), and whether voice 0 equals the CPU oracle bit for bit (checker only)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import knaster_amd
from knaster_amd import _lib as L, configs

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 256
voices = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 64, 4096]
block = int(sys.argv[3]) if len(sys.argv) > 3 else 128
for nv in voices:
    w = configs.fm_cascade(depth, nv, block)
    b = knaster_amd.VoiceBank(w.stages, nv, w.sample_type, 1, L.MIX_TREE)
    for s, a in w.ctor.items():
        b.set_ctor_args(s, a)
    t0 = time.perf_counter()
    b.init(48000, block)
    t_init = time.perf_counter() - t0
    first, _ = b.process_blocks(2)
    b.timing_reset(True)
    n_launch, blocks = 4, 32
    t0 = time.perf_counter()
    for _ in range(n_launch):
        b.process_blocks_device(blocks)
    b.synchronize()
    wall = time.perf_counter() - t0
    kms, n = b.timing_read()
    ugens = knaster_amd.chain_ugen_count(w.stages)
    line = {"workload": f"FM cascade, {depth} oscillators per voice", "stages": len(w.stages), "ugens_per_voice": ugens, "voices": nv,
            "block_size": block, "init_s_incl_hiprtc": round(t_init, 2), "us_per_block_kernel": kms * 1e3 / (n * blocks),
            "ugen_samples_per_s_kernel": float(nv) * block * ugens * blocks * n / (kms * 1e-3),
            "ugen_samples_per_s_wall": float(nv) * block * ugens * blocks * n_launch / wall}
    if nv <= 64:
        from oracle import oracle_py
        o = oracle_py.OracleBank(w.stages, 1, w.sample_type, 1, True, False)
        for s, a in w.ctor.items():
            o.set_ctor_args(s, a[:1])
        o.init(48000, block)
        ref = np.stack([o.process_block()[0] for _ in range(2)])
        t0 = time.perf_counter()
        for _ in range(20):
            o.process_block()
        line["oracle_cpu_us_per_block_one_voice"] = (time.perf_counter() - t0) / 20 * 1e6  # the unfused node graph on one host core
        g1 = knaster_amd.VoiceBank(w.stages, 1, w.sample_type, 1, L.MIX_LEFT_FOLD)
        for s, a in w.ctor.items():
            g1.set_ctor_args(s, a[:1])
        g1.init(48000, block)
        got = np.stack([g1.process_block()[0] for _ in range(2)])
        line["voice0_bit_identical_to_oracle"] = bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32)))
    print(json.dumps(line), flush=True)
    b.close()
