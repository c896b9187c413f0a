"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle;
see its header for provenance).  CPU: the oracle still reproduces them.  GPU: the HIP bank matches
them without the oracle in the loop."""
import os
import sys

import numpy as np
import pytest

from helpers import assert_bit_equal, make_gpu, make_oracle
from knaster_amd import _lib as L

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)
import make_golden  # noqa: E402


@pytest.mark.parametrize("case", sorted(make_golden.CASES))
def test_oracle_reproduces_golden(oracle, case):
    w, blocks = make_golden.workload(case)
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    o = make_oracle(oracle, w)
    for b in range(blocks):
        make_golden.script(case, w, b, o)
        out, voices, _, _ = o.process_block()
        assert_bit_equal(voices, g["voices"][b], f"{case} block {b} voices")
        assert_bit_equal(out, g["mix"][b], f"{case} block {b} mix")


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(make_golden.CASES))
def test_gpu_matches_golden(knh, case):
    w, blocks = make_golden.workload(case)
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    bank = make_gpu(knh, w, L.MIX_LEFT_FOLD)
    for b in range(blocks):
        make_golden.script(case, w, b, bank)
        out, voices, _ = bank.process_block_voices()
        if case == "c2_sin_numeric":  # device sin vs glibc sinf
            assert np.max(np.abs(voices.astype(np.float64) - g["voices"][b])) <= 1e-5
            assert np.max(np.abs(out.astype(np.float64) - g["mix"][b])) <= 1e-5
        else:
            assert_bit_equal(voices, g["voices"][b], f"{case} block {b} voices")
            assert_bit_equal(out, g["mix"][b], f"{case} block {b} mix")
    bank.close()
