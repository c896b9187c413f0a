#!/usr/bin/env python3
"""D3 = C3 with a SampleDelay (0.25 s ring per voice in HBM) behind the filter: the HBM-bound regime of the path.
One JSON line per bank size; run under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` for the traffic."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import knaster_amd
from knaster_amd import _lib as L, configs

BLOCKS = 32
for nv in [int(a) for a in sys.argv[1:]] or [16384, 65536, 262144]:
    w = configs.config("D3", n_voices=nv)
    b = knaster_amd.VoiceBank(w.stages, w.n_voices, w.sample_type, w.out_channels, L.MIX_TREE)
    for s, a in w.ctor.items():
        b.set_ctor_args(s, a)
    b.init(configs.SAMPLE_RATE, w.block_size)
    v = np.arange(nv, dtype=np.uint32)
    b.param_apply_many(v, w.restart[0], w.restart[1], L.VALUE_TRIGGER)
    b.param_apply_many(v, 3, 0, L.VALUE_FLOAT, w.delay_times)
    b.process_blocks_device(BLOCKS)
    b.synchronize()
    b.timing_reset(True)
    t0 = time.perf_counter()
    launches = 6
    for _ in range(launches):
        b.process_blocks_device(BLOCKS)
    b.synchronize()
    dt = time.perf_counter() - t0
    kms, n = b.timing_read()
    ugens = knaster_amd.chain_ugen_count(w.stages)
    work = float(nv) * w.block_size * ugens * BLOCKS * launches
    rd, wr = b.algorithmic_bytes_per_voice_block()
    ring = 8.0 * w.block_size  # one f32 read + one written per frame
    alg = (rd + wr + ring) * nv * BLOCKS
    print(json.dumps({"config": "D3", "voices": nv, "ugens_per_voice": ugens, "ugen_samples_per_s": work / dt,
                      "kernel_only_ugen_samples_per_s": work / (kms * 1e-3), "kernel_ms_per_launch": kms / n,
                      "algorithmic_bytes_per_launch": alg, "algorithmic_GBps": alg * n / (kms * 1e-3) / 1e9,
                      "ring_MiB": nv * 12000 * 4 / 2**20}), flush=True)
    b.close()
