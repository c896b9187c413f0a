"""Busy cycles per tile of the C3 pipeline's wavefronts by envelope situation (diagnostic build:
KNH_BUILD_STAMPS=1 python -m knaster_amd.build --force).  Every voice gets the same envelope times, so a window of
blocks holds one kind of tile: long release = Releasing tiles with no lane near its end; staggered short releases =
tiles in which lanes run out."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

import knaster_amd
from knaster_amd import _lib as L, configs


def run(label, attack, release, blocks_after_release):
    w = configs.config("C3")
    nv = w.n_voices
    w.ctor[3] = np.stack([np.broadcast_to(attack, (nv,)), np.broadcast_to(release, (nv,))], axis=1).astype(np.float64)
    b = knaster_amd.VoiceBank(w.stages, nv, w.sample_type, 2, L.MIX_TREE)
    for s, a in w.ctor.items():
        b.set_ctor_args(s, a)
    b.init(48000, 512)
    v = np.arange(nv, dtype=np.uint32)
    b.param_apply_many(v, w.restart[0], w.restart[1], L.VALUE_TRIGGER)
    b.process_blocks(8)
    att = b.debug_words()[4:7].copy()
    b.process_blocks(8)
    sus = b.debug_words()[4:7].copy()
    b.param_apply_many(v, w.release[0], w.release[1], L.VALUE_TRIGGER)
    b.process_blocks(blocks_after_release)
    rel = b.debug_words()[4:7].copy()
    print(f"{label}: [osc, svf, env(+fold)] attack window {att}, sustain {sus}, after release {rel}")
    b.close()


run("attack 1 s, release 4 s (no lane near a threshold)", 1.0, 4.0, 8)
lane = np.arange(16384) % 64
run("release ends staggered over 8 blocks (one lane per tile)", 0.002, (lane + 1) * 64 / 48000.0, 8)
run("every release ends inside block 4", 0.002, 4.5 * 512 / 48000.0, 8)
