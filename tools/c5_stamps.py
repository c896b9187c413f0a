"""Busy cycles per 32-sample tile of the wavefronts of the C5 pipeline kernel (modulator | carrier | mixer), in launches
without and with sample-accurate changes.  Needs the diagnostic build: KNH_BUILD_STAMPS=1 python -m knaster_amd.build --force"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

import knaster_amd
from knaster_amd import _lib as L, configs

w = configs.config("C5")
b = knaster_amd.VoiceBank(w.stages, w.n_voices, w.sample_type, 2, L.MIX_TREE)
for s, a in w.ctor.items():
    b.set_ctor_args(s, a)
b.init(48000, w.block_size)
blocks = 32
b.process_blocks(blocks)
b.timing_reset(True)
b.process_blocks(blocks)
print("no changes:      busy cycles per tile [modulator, carrier, mixer]:", b.debug_words()[4:7], "in/out per group:", b.debug_words()[8:12],
      "kernel us per block", b.timing_read()[0] * 1e3 / blocks)
for density, label in ((2, "every 2nd block (C5)"), (1, "every block")):
    for rep in range(2):
        for i in range(blocks):
            if i % density == 0:
                e = configs.c5_events(w, 2 * i)
                b.param_apply_many(e[0], e[1], e[2], e[3], e[4], None, e[5], block_offset=i)
        b.timing_reset(True)
        b.process_blocks(blocks)
    print(f"changes {label}: busy cycles per tile [modulator, carrier, mixer]:", b.debug_words()[4:7], "in/out per group:", b.debug_words()[8:12],
          "kernel us per block", b.timing_read()[0] * 1e3 / blocks)
