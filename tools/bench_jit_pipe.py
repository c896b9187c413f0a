#!/usr/bin/env python3
"""A chain with no pre-built kernel at 16 384 voices: run-time-built pipeline against the run-time-built single-wave kernel."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np

    import knaster_amd
    from knaster_amd import _lib as L, configs
    from knaster_amd.bank import Stage

    n, bs, blocks = 16384, 512, 32
    p = configs.voice_parameters(n)
    stages = [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_WR_MUL), Stage(L.STAGE_ONEPOLE_LPF), Stage(L.STAGE_SVF), Stage(L.STAGE_MUL_ENV_AR),
              Stage(L.STAGE_MUL_CONST)]
    b = knaster_amd.VoiceBank(stages, n, L.F32, 2, L.MIX_TREE)
    b.set_ctor_args(0, p["freq"].reshape(n, 1))
    b.set_ctor_args(1, np.full((n, 1), 0.7))
    b.set_ctor_args(2, p["cutoff"].reshape(n, 1))
    b.set_ctor_args(3, np.stack([np.full(n, float(L.SVF_BAND)), p["cutoff"] * 0.5, p["q"], np.zeros(n)], axis=1))
    b.set_ctor_args(4, np.stack([p["attack"], p["release"]], axis=1))
    b.set_ctor_args(5, np.full((n, 1), 1.0 / n))
    b.init(48000, bs)
    v = np.arange(n, dtype=np.uint32)
    b.param_apply_many(v, 4, 2, L.VALUE_TRIGGER)
    b.process_blocks_device(blocks)
    b.synchronize()
    b.timing_reset(True)
    for _ in range(6):
        b.process_blocks_device(blocks)
    b.synchronize()
    kms, k = b.timing_read()
    ugens = knaster_amd.chain_ugen_count(stages)
    print(json.dumps({"KNH_JIT_PIPE": os.environ.get("KNH_JIT_PIPE", "1"), "chain": "SinWt.wr_mul -> OnePoleLpf -> SvfFilter(Band) -> * EnvAr -> * c",
                      "voices": n, "us_per_block_kernel": kms * 1e3 / (k * blocks),
                      "kernel_only_ugen_samples_per_s": n * bs * ugens * blocks * k / (kms * 1e-3)}))
else:
    for mode in ("0", "1"):
        env = dict(os.environ, KNH_JIT_PIPE=mode)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
