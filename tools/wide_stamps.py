"""Cycles per 64 samples of one wavefront of the WHOLE-CHAIN kernel (voice_kernel<F, FMA, WAVES, ...>: banks of more
64-voice groups than the pipeline covers), per stage, for its tile stores and its fold.
Needs the diagnostic build:  python -m knaster_amd.build --variant=stamps ; KNH_LIB=knaster_amd/csrc/libknaster_hip_stamps.so
usage: python tools/wide_stamps.py C4 65536 [KNH_WIDE value]"""
import json
import os
import sys

sys.path.insert(0, os.getcwd())
if len(sys.argv) > 3:
    os.environ["KNH_WIDE"] = sys.argv[3]
import numpy as np

import knaster_amd
from knaster_amd import _lib as L, configs

name = sys.argv[1] if len(sys.argv) > 1 else "C4"
w = configs.config(name, n_voices=int(sys.argv[2]) if len(sys.argv) > 2 else None)
b = knaster_amd.VoiceBank(w.stages, w.n_voices, w.sample_type, 2, L.MIX_TREE)
for s, a in w.ctor.items():
    b.set_ctor_args(s, a)
b.init(48000, w.block_size)
v = np.arange(w.n_voices, dtype=np.uint32)
if w.restart:
    b.param_apply_many(v, w.restart[0], w.restart[1], L.VALUE_TRIGGER)
out = {"config": name, "voices": w.n_voices, "sample_type": "f64" if w.sample_type else "f32", "block_size": w.block_size,
       "KNH_WIDE": os.environ.get("KNH_WIDE", "default"), "unit": "shader cycles per 64 samples of one 64-voice wavefront (wavefront 0)",
       "stages": [int(s.kind) for s in w.stages], "phases": {}}


def words():
    d = [int(x) for x in b.debug_words()]
    n = len(w.stages)
    return {"per_stage": d[4:4 + n], "tile_stores": d[12], "fold": d[13], "everything": d[14], "stamped_share": d[15] / 64.0}


b.process_blocks_device(8)
out["phases"]["attack (blocks 0-8)"] = words()
b.process_blocks_device(8)
out["phases"]["sustain (8-16)"] = words()
if w.release:
    b.param_apply_many(v, w.release[0], w.release[1], L.VALUE_TRIGGER)
b.process_blocks_device(8)
out["phases"]["release"] = words()
b.process_blocks_device(32)
out["phases"]["stopped"] = words()
b.timing_reset(True)
if w.restart:
    b.param_apply_many(v, w.restart[0], w.restart[1], L.VALUE_TRIGGER)
if w.release:
    b.param_apply_many(v, w.release[0], w.release[1], L.VALUE_TRIGGER, block_offset=32)
b.process_blocks_device(64)
b.synchronize()
ms, n = b.timing_read()
out["note_cycle_64_blocks"] = words()
out["kernel_ms_64_blocks_stamped_build"] = ms / max(n, 1)
out["us_per_block_stamped_build"] = ms * 1e3 / max(n, 1) / 64
print(json.dumps(out, indent=1))
