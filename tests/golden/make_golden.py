#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small input/output vectors of the five BASELINE.json workloads.

PROVENANCE: the outputs come from the CPU oracle (oracle/), NOT from the Rust reference -- the image
has no Rust toolchain, so the reference cannot be run.  They pin the oracle against regressions and
give the GPU tests a target that does not need the oracle at run time.  Re-validate against the real
crate wherever `cargo` exists.

    python tests/golden/make_golden.py
    python tests/golden/make_golden.py --inputs   # only writes <case>.ctor.bin: the constructor arguments, for the Rust-side
                                                  # generator bindings/rust/knaster_hip/examples/dump_golden.rs (run where cargo exists)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from knaster_amd import _lib as L  # noqa: E402
from knaster_amd import configs  # noqa: E402

CASES = {
    # name: (config, voices, block_size, blocks)
    "c1_readme": ("C1", 1, 64, 3),
    "c2_sin_numeric": ("C2", 12, 48, 3),
    "c3_chain_f32": ("C3", 24, 96, 5),
    "c4_chain_f64": ("C4", 10, 64, 5),
    "c5_fm_events": ("C5", 16, 128, 4),
    "m1_many_sines_pan2": ("M1", 20, 64, 5),  # voices: [block][2][voice][frame] (left, right)
}


def script(case: str, w, block: int, bank):
    """The parameter events of `block`, identical for every backend."""
    v = np.arange(w.n_voices, dtype=np.uint32)
    if w.name in ("C3", "C4"):
        if block == 0:
            bank.param_apply_many(v, 3, 3, L.VALUE_TRIGGER)
        if block == 2:
            bank.param_apply_many(v[::2], 3, 2, L.VALUE_TRIGGER)
        if block == 3:
            bank.param_apply_many(v, 2, 0, L.VALUE_FLOAT, 300.0 + 50.0 * v)
    if w.name == "M1":  # many_sines.rs:64-92: new frequencies and envelope restarts while running; here also new pans
        if block in (0, 3):
            bank.param_apply_many(v, 2, 2, L.VALUE_TRIGGER)
        if block == 1:
            bank.param_apply_many(v[::3], 0, 0, L.VALUE_FLOAT, 110.0 * (1 + v[::3] % 7))
        if block == 2:
            bank.param_apply_many(v[1::2], 3, 0, L.VALUE_FLOAT, np.linspace(-1.0, 1.0, len(v[1::2])))
    if w.name == "C5":
        e = configs.c5_events(w, block)
        if e is not None:
            bank.param_apply_many(e[0], e[1], e[2], e[3], e[4], None, e[5])


def workload(case: str):
    name, nv, bs, blocks = CASES[case]
    return configs.config(name, n_voices=nv, block_size=bs), blocks


def write_inputs():
    """<case>.ctor.bin: u32 magic, n_voices, block_size, blocks, n_stages; per stage u32 n_args + f64[n_voices][n_args]."""
    import struct

    for case in CASES:
        w, blocks = workload(case)
        with open(os.path.join(HERE, case + ".ctor.bin"), "wb") as f:
            f.write(struct.pack("<5I", 0x4B4E4831, w.n_voices, w.block_size, blocks, len(w.stages)))
            for s in range(len(w.stages)):
                a = np.ascontiguousarray(w.ctor.get(s, np.zeros((w.n_voices, 0))), dtype="<f8").reshape(w.n_voices, -1)
                f.write(struct.pack("<I", a.shape[1]))
                f.write(a.tobytes())
        print(os.path.join(HERE, case + ".ctor.bin"))


def main():
    if "--inputs" in sys.argv:
        write_inputs()
        return
    from helpers import make_oracle
    from oracle import oracle_py

    for case in CASES:
        w, blocks = workload(case)
        o = make_oracle(oracle_py, w)
        voices, mix = [], []
        for b in range(blocks):
            script(case, w, b, o)
            out, vs, _, _ = o.process_block()
            voices.append(vs)
            mix.append(out)
        path = os.path.join(HERE, case + ".npz")
        np.savez_compressed(path, voices=np.stack(voices), mix=np.stack(mix))
        print(path, np.stack(voices).shape, np.stack(mix).shape)


if __name__ == "__main__":
    main()
