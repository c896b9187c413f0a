"""The hand-written SVF steps let an instruction read a packed-f32 result with no wait state between them (voice_stages.hpp,
KNH_SVF_WAIT: the compiler's hazard table would pad an `s_nop` there, the hardware was measured not to need it --
tools/micro/svf_low_variants.hip, ONE wavefront per SIMD).  KNH_SVF_NOP=1 puts the `s_nop` back into kernels fused at run time;
here the two are compared bit for bit where the measurement did not reach: two and four wavefronts per SIMD (eight and sixteen
whole-chain wavefronts per workgroup), and the pipeline.  """
import numpy as np
import pytest

from helpers import assert_bit_equal, make_gpu
from knaster_amd import _lib as L
from knaster_amd import configs

pytestmark = pytest.mark.gpu


def render(knh, monkeypatch, nop, form):
    for k in ("KNH_JIT_WAVES", "KNH_PIPELINE", "KNH_SVF_NOP"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("KNH_JIT", "1")  # run-time fusion also for chains that have pre-built kernels
    if form != "pipeline":
        monkeypatch.setenv("KNH_PIPELINE", "0")
        monkeypatch.setenv("KNH_JIT_WAVES", form)
    if nop:
        monkeypatch.setenv("KNH_SVF_NOP", "1")
    n, bs, blocks = 2112, 256, 5  # 33 voice groups: three workgroups of sixteen, the last one ragged
    w = configs.config("C3", n_voices=n, block_size=bs)
    ty = np.full(n, float(L.SVF_LOW))
    ty[64:1088] = np.arange(1024) % 9  # wavefronts of low-pass voices (the seven-instruction step) beside wavefronts of all nine types
    w.ctor[2][:, 0] = ty
    g = make_gpu(knh, w, L.MIX_TREE)
    v = np.arange(n, dtype=np.uint32)
    g.param_apply_many(v, w.restart[0], w.restart[1], L.VALUE_TRIGGER)
    outs = []
    for b in range(blocks):
        if b == 3:
            g.param_apply_many(v, w.release[0], w.release[1], L.VALUE_TRIGGER)
        out, voices, _ = g.process_block_voices()
        outs.append((out.copy(), voices.copy()))
    g.close()
    return outs


@pytest.mark.parametrize("form", ["1", "4", "8", "16", "pipeline"])
def test_svf_steps_with_and_without_the_wait_state(knh, monkeypatch, form):
    a = render(knh, monkeypatch, False, form)
    b = render(knh, monkeypatch, True, form)
    for k, ((ao, av), (bo, bv)) in enumerate(zip(a, b)):
        assert_bit_equal(av, bv, f"form {form} block {k} per voice", strict_zero=True)
        assert_bit_equal(ao, bo, f"form {form} block {k} mix", strict_zero=True)
    assert np.abs(a[-2][1]).max() > 1e-6
