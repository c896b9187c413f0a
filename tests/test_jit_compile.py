"""Run-time fusion without a GPU: the compile (hiprtc) is host work, so kernels of chains with no pre-built form can be
compiled here, each in a process of its own (tests/cpp/bin/jit_compile_check) -- a compiler crash would otherwise take the HOST
process down at knh_bank_init, as one did: the voice of tests/test_gpu_dag.py's seed 175 (two filters in a graph) crashed the
compiler's subregister renaming pass while the filter's two step encodings carried two-float values across their asm blocks.
tools/jit_compile_fuzz.py does the same for hundreds of random voices."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "tests", "cpp", "bin", "jit_compile_check")

SIGNATURES = [
    ("W@_,_,0S@0,_,0*@0,0,0S@0,_,1-@1,0,0X@0,_,0#2", "f32"),   # the voice that crashed the compiler: SinWt -> Svf -> x*x -> Svf -> (a - b) -> limiter
    ("W@_,_,0S@0,_,0*@0,0,0S@0,_,1-@1,0,0X@0,_,0#2", "f64"),
    ("WSLSA", "f32"),                                            # a plain chain with two filters and a one-pole between them
    ("WSLSA", "f32 pipe"),                                       # ... and as the pipeline knh_bank_init would cut it into
]


@pytest.mark.skipif(not os.path.exists(CHECK), reason="tests/cpp/bin/jit_compile_check not built (make -C tests/cpp)")
@pytest.mark.parametrize("signature,sample_type", SIGNATURES)
def test_kernels_of_tricky_voices_compile(signature, sample_type):
    p = subprocess.run([CHECK, signature] + [a for a in sample_type.split() if a != "f32"], cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=900)
    assert p.returncode == 0, f"{signature} ({sample_type}): rc {p.returncode}: {p.stdout.decode(errors='replace')[-600:]}"


def test_debug_signature_names_the_chain():
    """knh_bank_debug_signature: the string run-time fusion starts from (and tools/jit_compile_fuzz.py compiles) -- available
    without a device, before init."""
    import numpy as np  # noqa: F401
    import knaster_amd
    from knaster_amd import _lib as L, configs
    from knaster_amd.bank import Stage
    w = configs.config("C3", n_voices=64)
    b = knaster_amd.VoiceBank(w.stages, w.n_voices, w.sample_type, w.out_channels, L.MIX_TREE)
    assert b.debug_signature() == "WmSA"
    b.close()
    st = [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_SVF), Stage(L.STAGE_MATH_MUL, input=2, input2=2), Stage(L.STAGE_SVF),
          Stage(L.STAGE_MATH_SUB, input=4, input2=3), Stage(L.STAGE_SAFETY_LIMITER)]
    b = knaster_amd.VoiceBank(st, 3, L.F32, 1, L.MIX_LEFT_FOLD)
    assert b.debug_signature() == SIGNATURES[0][0]  # the voice of test_gpu_dag.py's seed 175
    b.close()
