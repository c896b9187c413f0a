#!/usr/bin/env python3
"""Kernel time of the C3 voice with a PolyBlep oscillator, one waveform for the whole bank, per waveform."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import knaster_amd
from knaster_amd import _lib as L, configs

NAMES = ["Sawtooth", "Sine", "Cosine", "Triangle", "Square", "Rectangle", "Ramp", "ModifiedTriangle", "ModifiedSquare",
         "HalfWaveRectifiedSine", "FullWaveRectifiedSine", "TriangularPulse", "TrapezoidFixed", "TrapezoidVariable"]
for wf in range(14):
    w = configs.config("B3")
    w.ctor[0][:, 0] = wf
    b = knaster_amd.VoiceBank(w.stages, w.n_voices, w.sample_type, 2, L.MIX_TREE)
    for s, a in w.ctor.items():
        b.set_ctor_args(s, a)
    b.init(48000, 512)
    v = np.arange(w.n_voices, dtype=np.uint32)
    b.param_apply_many(v, 3, 3, L.VALUE_TRIGGER)
    b.process_blocks_device(8)
    b.synchronize()
    b.timing_reset(True)
    b.process_blocks_device(16)
    b.synchronize()
    kms, k = b.timing_read()
    print(json.dumps({"waveform": NAMES[wf], "us_per_block_kernel": kms * 1e3 / 16}), flush=True)
    b.close()
