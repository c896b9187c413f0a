"""Busy cycles per 32-sample tile of every wavefront of the default pipeline kernel (C3).
Needs the diagnostic build: KNH_BUILD_STAMPS=1 python -m knaster_amd.build --force"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

import knaster_amd
from knaster_amd import _lib as L, configs

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
w = configs.config(name, n_voices=int(sys.argv[2]) if len(sys.argv) > 2 else None)
b = knaster_amd.VoiceBank(w.stages, w.n_voices, w.sample_type, 2, L.MIX_TREE)
for s, a in w.ctor.items():
    b.set_ctor_args(s, a)
b.init(48000, w.block_size)
v = np.arange(w.n_voices, dtype=np.uint32)
if w.restart:
    b.param_apply_many(v, w.restart[0], w.restart[1], L.VALUE_TRIGGER)
if w.delay_times is not None:
    b.param_apply_many(v, 3, 0, L.VALUE_FLOAT, w.delay_times)
for phase in ["attack (blocks 0-8)", "sustain (8-16)"]:
    b.process_blocks(8)
    print(phase, "busy cycles per tile [osc+gain, svf, env, mixer]:", b.debug_words()[4:8], "in/out per group:", b.debug_words()[8:14],
          "last group's arithmetic / fold:", b.debug_words()[14:16])
if w.release:
    b.param_apply_many(v, w.release[0], w.release[1], L.VALUE_TRIGGER)
b.process_blocks(8)
print("release", b.debug_words()[4:8], "in/out per group:", b.debug_words()[8:14])
b.process_blocks(32)
print("stopped", b.debug_words()[4:8], "in/out per group:", b.debug_words()[8:14])
b.timing_reset(True)
b.process_blocks(64)
print("the folding group (64-sample form): stage arithmetic and fold, cycles per tile:", b.debug_words()[14:16])
print("kernel ms for 64 blocks", b.timing_read())
