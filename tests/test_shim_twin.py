"""The C ABI driven exactly as the Rust shim drives it: tests/cpp/shim_twin.hpp mirrors bindings/rust/knaster_hip/src/lib.rs
method by method, tests/cpp/shim_twin_test.cpp makes the reference's call sequence with it (events, then one
process_block per block: knaster_graph/src/graph_gen.rs:110-200, task.rs:25-31, processor.rs:142-179)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
BIN = os.path.join(CPP, "bin", "shim_twin_test")


def _build(knh):
    subprocess.run(["make", "-C", CPP, "bin/shim_twin_test"], check=True, capture_output=True)
    assert os.path.exists(BIN)


def test_no_exception_crosses_the_c_abi(knh):
    """std::bad_alloc inside knh_bank_create (address space capped with setrlimit) comes back as KNH_ERR_OUT_OF_MEMORY."""
    _build(knh)
    res = subprocess.run([BIN, "--cpu"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "ok   cpu_no_exception_crosses_the_abi" in res.stdout


def _body(text: str, start: int) -> str:
    """the brace-delimited block that opens at or after `start`"""
    i = text.index("{", start)
    depth, j = 0, i
    while True:
        depth += text[j] == "{"
        depth -= text[j] == "}"
        j += 1
        if depth == 0:
            return text[i:j]


def _calls(body: str):
    out = []
    for name in re.findall(r"\b(knh_[a-z_]+)\s*\(", body):
        if name != "knh_last_error" and (not out or out[-1] != name):
            out.append(name)
    return out


def test_shim_twin_makes_the_shims_calls():
    """Every UGen method of the Rust shim and its C++ twin name the same C entry points in the same order."""
    rust = open(os.path.join(ROOT, "bindings", "rust", "knaster_hip", "src", "lib.rs")).read()
    twin = open(os.path.join(CPP, "shim_twin.hpp")).read()
    pairs = {
        "init": (r"fn init\(&mut self, sample_rate: u32, block_size: usize\)", r"void init\(uint32_t sample_rate, size_t block_size\)"),
        "process_block": (r"fn process_block<InBlock, OutBlock>\(", r"int32_t process_block\(AudioCtx& ctx"),
        "param_apply": (r"fn param_apply\(&mut self, ctx: &mut AudioCtx, index: usize, value: ParameterValue\)", r"int32_t param_apply\(AudioCtx&"),
        "set_ar_param_buffer": (r"unsafe fn set_ar_param_buffer\(", r"void set_ar_param_buffer\(AudioCtx&"),
        "set_delay_within_block_for_param": (r"fn set_delay_within_block_for_param\(", r"int32_t set_delay_within_block_for_param\(AudioCtx&"),
        "new": (r"pub fn with_options\(", r"GpuVoiceBank\(const std::vector<knh_stage_desc>& stages"),
        "drop": (r"fn drop\(&mut self\)", r"~GpuVoiceBank\(\)"),
        "index": (r"pub fn index\(&self", r"size_t index\(uint32_t voice"),
        "param_apply_range": (r"pub fn param_apply_range\(&mut self, voice_begin: u32", r"int32_t param_apply_range\(uint32_t voice_begin"),
        "param_apply_many": (r"pub fn param_apply_many\(&mut self, indices: &\[usize\], value: ParameterValue\)", r"int32_t param_apply_many\(const std::vector<size_t>& indices"),
    }
    for method, (r_pat, c_pat) in pairs.items():
        r_m, c_m = re.search(r_pat, rust), re.search(c_pat, twin)
        assert r_m and c_m, method
        r_calls, c_calls = _calls(_body(rust, r_m.end())), _calls(_body(twin, c_m.end()))
        assert r_calls == c_calls, f"{method}: lib.rs calls {r_calls}, shim_twin.hpp calls {c_calls}"
    # what the shim declares about itself
    assert "type Inputs = I;" in rust and "in_channels: (I::USIZE + ar_slots) as u32" in rust
    assert "unsafe fn set_ar_param_buffer" in rust and "knh_bank_set_input" in rust


@pytest.mark.gpu
def test_reference_call_sequences_on_gpu(knh):
    _build(knh)
    res = subprocess.run([BIN, "--gpu"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    for name in ("gpu_readme_example_block_by_block", "gpu_c3_block_by_block_equals_one_launch", "gpu_partial_blocks",
                 "gpu_inputs_and_audio_rate_buffer"):
        assert f"ok   {name}" in res.stdout, res.stdout
