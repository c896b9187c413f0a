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


@pytest.mark.parametrize("case", sorted(make_golden.CASES))
def test_reference_dump_matches_golden(case):
    """The pin that needs a Rust toolchain: bindings/rust/knaster_hip/examples/dump_golden.rs renders the same cases with the
    REAL crate and writes tests/golden/reference/<case>.ref.bin.  When those files are present the committed vectors (from
    this repository's oracle) are compared with them -- bit for bit, except c2 (SinNumeric's sin comes from the platform's
    libm on either side: 1e-6).  Absent (no cargo in the build image): skipped, and the waveform-level parity stays
    "unpinned" as DESIGN.md section 2 says."""
    path = os.path.join(GOLDEN, "reference", case + ".ref.bin")
    if not os.path.exists(path):
        pytest.skip("no reference dump (run the Rust example where cargo exists)")
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    dtype = g["voices"].dtype
    raw = np.fromfile(path, dtype=dtype.newbyteorder("<"))
    nv, nm = g["voices"].size, g["mix"].size
    assert raw.size == nv + nm, f"{raw.size} samples in the dump, {nv + nm} expected"
    voices, mix = raw[:nv].reshape(g["voices"].shape), raw[nv:].reshape(g["mix"].shape)
    if case == "c2_sin_numeric":
        assert np.max(np.abs(voices.astype(np.float64) - g["voices"])) <= 1e-6
        assert np.max(np.abs(mix.astype(np.float64) - g["mix"])) <= 1e-6
    else:
        assert_bit_equal(voices.astype(dtype), g["voices"], f"{case}: per-voice signals, reference against oracle")
        assert_bit_equal(mix.astype(dtype), g["mix"], f"{case}: left-fold mix, reference against oracle")
