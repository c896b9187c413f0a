"""The C++ host mirror of the reference's graph API (knaster_amd/host/knaster_host.hpp) over the C ABI:
tests/cpp/host_mirror_test.cpp is built by `make -C tests/cpp` (also by __graft_entry__.build())."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "bin", "host_mirror_test")


def _build(knh):
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True, capture_output=True)
    assert os.path.exists(BIN)


def test_graph_api_plans_voice_banks(knh):
    _build(knh)
    res = subprocess.run([BIN, "--plan"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    for name in ("plan_readme_example", "plan_groups_voices_by_chain_shape", "plan_rejects_what_is_not_a_voice_chain", "time_and_seconds",
                 "plan_noise_sources_take_seeds_in_construction_order", "plan_many_sines_with_pan2", "plan_voices_that_are_graphs"):
        assert f"ok   {name}" in res.stdout


@pytest.mark.gpu
def test_graph_api_end_to_end_on_gpu(knh):
    _build(knh)
    res = subprocess.run([BIN, "--gpu"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    for name in ("gpu_readme_example", "gpu_voice_graph_matches_reference_shaped_graph", "gpu_run_blocks_equals_block_by_block",
                 "gpu_heterogeneous_voices_mix_on_device", "gpu_many_sines_with_pan2", "gpu_voices_that_are_graphs", "gpu_reference_fm_cascade_256"):
        assert f"ok   {name}" in res.stdout


def test_shard_worker_pool_plain_and_under_thread_sanitizer():
    """knaster_amd/csrc/shard_workers.hpp (the threads of knh_bank_create_sharded) has no HIP in it: run it on the CPU,
    once as built for the library and once under ThreadSanitizer."""
    cpp = os.path.join(ROOT, "tests", "cpp")
    subprocess.run(["make", "-C", cpp, "bin/shard_workers_test", "bin/shard_workers_test_tsan"], check=True, capture_output=True)
    for name in ("shard_workers_test", "shard_workers_test_tsan"):
        res = subprocess.run([os.path.join(cpp, "bin", name)], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0 and "ok   shard_workers" in res.stdout, name + ":\n" + res.stdout + res.stderr
