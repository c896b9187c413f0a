"""Run-time fusion out of the host's blast radius (knaster_amd/csrc/jit_cache.hpp): the code-object cache on disk and the
compile in a helper process.  The compile is host work, so all of it but the last test runs without a GPU, through
tests/cpp/bin/jit_compile_check (a process that calls the library's own jit entry points)."""
import glob
import hashlib
import os
import subprocess
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "tests", "cpp", "bin", "jit_compile_check")
HELPER = os.path.join(ROOT, "knaster_amd", "csrc", "knh_jit_helper")
needs_check = pytest.mark.skipif(not os.path.exists(CHECK), reason="tests/cpp/bin/jit_compile_check not built (make -C tests/cpp)")


def run(args, cache_dir, **env):
    e = dict(os.environ, KNH_JIT_CACHE_DIR=str(cache_dir), AMD_COMGR_CACHE="0")  # (comgr's own cache off: the compiles below are real ones)
    e.pop("KNH_JIT_CACHE", None)
    e.update(env)
    t0 = time.perf_counter()
    p = subprocess.run([CHECK] + args, cwd="/tmp", env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, text=True)
    return p.returncode, p.stdout, time.perf_counter() - t0


@needs_check
def test_sha256_is_sha256():
    for text in ["", "abc", "a" * 55, "a" * 56, "a" * 63, "a" * 64, "a" * 65, "knaster" * 1000]:
        rc, out, _ = run(["--sha256", text], "/tmp")
        assert rc == 0 and out.strip() == hashlib.sha256(text.encode()).hexdigest(), text[:20]


@needs_check
def test_second_process_loads_from_disk_and_a_damaged_entry_is_a_miss(tmp_path):
    assert os.path.exists(HELPER), "knh_jit_helper is built beside the library (knaster_amd/build.py)"
    rc, out, cold = run(["WLHSA"], tmp_path)
    assert rc == 0 and "compiled" in out and "helper 1" in out and "disk 0" in out and "in-process 0" in out, out
    entries = glob.glob(str(tmp_path / "*.knhco"))
    assert len(entries) == 1 and not glob.glob(str(tmp_path / "job-*")), os.listdir(tmp_path)  # job and log files are gone
    rc, out, warm = run(["WLHSA"], tmp_path)
    assert rc == 0 and "disk 1" in out and "helper 0" in out, out
    assert warm < 0.1 + 0.25 * cold, f"cold {cold:.2f} s, from disk {warm:.2f} s"
    # another kernel identity (the same chain in f64) is another entry
    rc, out, _ = run(["WLHSA", "f64"], tmp_path)
    assert rc == 0 and "helper 1" in out and len(glob.glob(str(tmp_path / "*.knhco"))) == 2, out
    # a torn / tampered entry is not loaded: it is compiled again and replaced
    data = bytearray(open(entries[0], "rb").read())
    data[len(data) // 2] ^= 0x40
    open(entries[0], "wb").write(bytes(data))
    rc, out, _ = run(["WLHSA"], tmp_path)
    assert rc == 0 and "helper 1" in out and "disk 0" in out, out
    rc, out, _ = run(["WLHSA"], tmp_path)
    assert rc == 0 and "disk 1" in out, out
    open(entries[0], "wb").write(b"KNHCO001")  # a truncated one
    rc, out, _ = run(["WLHSA"], tmp_path)
    assert rc == 0 and "helper 1" in out, out


@needs_check
@pytest.mark.parametrize("how,needle", [("crash", "signal 6"), ("segv", "signal 11"), ("hang", "was killed")])
def test_a_compiler_that_dies_is_a_message_not_the_hosts_death(tmp_path, how, needle):
    """The helper is made to abort, to fault and to hang (KNH_JIT_HELPER_TEST): the process that asked for the kernel gets an
    error string -- which knh_bank_init turns into KNH_ERR_INTERNAL + knh_last_error -- and goes on living."""
    rc, out, dt = run(["WLHSA"], tmp_path, KNH_JIT_HELPER_TEST=how, KNH_JIT_TIMEOUT_S="2")
    assert rc == 1 and "FAILED: JIT_CRASH:" in out and needle in out, out  # rc 1 = jit_compile_check's own "failed", not a signal
    assert "jit stats:" in out  # ... printed by the caller after the failure: it is alive
    assert not glob.glob(str(tmp_path / "*.knhco")) and not glob.glob(str(tmp_path / "job-*"))
    assert dt < 30
    rc, out, _ = run(["WLHSA"], tmp_path)  # and without the fault the same request succeeds
    assert rc == 0 and "helper 1" in out, out


@needs_check
def test_without_a_helper_or_a_cache_the_compile_still_happens(tmp_path):
    rc, out, _ = run(["WLHSA"], tmp_path, KNH_JIT_INPROCESS="1")
    assert rc == 0 and "in-process 1" in out and len(glob.glob(str(tmp_path / "*.knhco"))) == 1, out  # (and feeds the cache)
    rc, out, _ = run(["WLHSA"], tmp_path, KNH_JIT_CACHE="0")
    assert rc == 0 and "helper 1" in out and "disk 0" in out, out
    rc, out, _ = run(["WLHSA"], tmp_path, KNH_JIT_HELPER="/nonexistent/knh_jit_helper", KNH_JIT_CACHE="0")
    assert rc == 1 and "could not start the JIT helper" in out, out
    rc, out, _ = run(["W?SA"], tmp_path)  # a compile ERROR (not a crash) keeps its own message
    assert rc == 1 and "JIT_CRASH" not in out, out


@pytest.mark.gpu
def test_bank_init_reports_a_compiler_crash_and_loads_the_second_time_from_disk(knh, tmp_path, monkeypatch):
    import numpy as np
    from knaster_amd import _lib as L
    from knaster_amd.bank import Stage
    monkeypatch.setenv("KNH_JIT_CACHE_DIR", str(tmp_path))
    st = [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_ONEPOLE_LPF), Stage(L.STAGE_ONEPOLE_HPF), Stage(L.STAGE_SVF), Stage(L.STAGE_ADD_CONST), Stage(L.STAGE_MUL_ENV_AR)]

    def bank():
        b = knh.VoiceBank(st, 130, L.F32, 2, L.MIX_TREE)
        b.set_ctor_args(0, np.full((130, 1), 220.0))
        b.set_ctor_args(1, np.full((130, 1), 900.0))
        b.set_ctor_args(3, np.tile([0.0, 700.0, 1.0, 0.0], (130, 1)))
        b.set_ctor_args(4, np.full((130, 1), 0.125))
        b.set_ctor_args(5, np.tile([0.01, 0.1], (130, 1)))
        return b
    monkeypatch.setenv("KNH_JIT_HELPER_TEST", "crash")
    b = bank()
    with pytest.raises(L.KnasterHipError) as e:
        b.init(48000, 64)
    assert e.value.status == L.ERR_INTERNAL and "JIT helper" in str(e.value)
    b.close()
    monkeypatch.delenv("KNH_JIT_HELPER_TEST")
    b = bank()
    t0 = time.perf_counter()
    b.init(48000, 64)
    cold = time.perf_counter() - t0
    out, _ = b.process_block()
    assert np.isfinite(out).all()
    b.close()
    assert glob.glob(str(tmp_path / "*.knhco"))
    # a second process: its bank of the same shape comes up from the disk cache
    code = ("import time, numpy as np, knaster_amd\nfrom knaster_amd import _lib as L\nfrom knaster_amd.bank import Stage\n"
            "st = [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_ONEPOLE_LPF), Stage(L.STAGE_ONEPOLE_HPF), Stage(L.STAGE_SVF), Stage(L.STAGE_ADD_CONST), Stage(L.STAGE_MUL_ENV_AR)]\n"
            "b = knaster_amd.VoiceBank(st, 130, L.F32, 2, L.MIX_TREE)\nb.set_ctor_args(3, np.tile([0.0, 700.0, 1.0, 0.0], (130, 1)))\n"
            "t0 = time.perf_counter(); b.init(48000, 64); print('INIT_S', time.perf_counter() - t0)\n"
            "import ctypes as C\nv = [C.c_uint64(0) for _ in range(4)]\nL.load().knh_jit_stats(*[C.byref(x) for x in v])\nprint('STATS', *[x.value for x in v])\n")
    p = subprocess.run(["python", "-c", code], cwd=ROOT, env=dict(os.environ, AMD_COMGR_CACHE="0"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout
    stats = [int(x) for x in p.stdout.split("STATS")[1].split()[:4]]
    assert stats[1] >= 1 and stats[2] == 0 and stats[3] == 0, p.stdout  # from disk: no compile of either kind
    warm = float(p.stdout.split("INIT_S")[1].split()[0])
    assert warm < 0.1 + 0.25 * cold or warm < 0.5, (cold, warm)
