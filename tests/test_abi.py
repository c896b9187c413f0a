"""The C-ABI boundary on a machine without a GPU: the library loads, exports every symbol the
header declares (and the ctypes prototypes cover all of them), validates chains, and refuses to
compute -- there is no CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from knaster_amd import _lib as L
from knaster_amd import configs
from knaster_amd.bank import Stage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "knaster_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(knh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(knh):
    lib = C.CDLL(L.LIB_PATH)
    names = header_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in knaster_hip.h but not exported"
    assert set(names) == set(L.PROTOTYPES), "ctypes prototypes and header disagree"
    assert knh.lib.load().knh_abi_version() == L.KNH_ABI_VERSION


def test_no_oracle_in_product():
    """The product never links, includes or imports the oracle."""
    for root, _dirs, files in os.walk(os.path.join(ROOT, "knaster_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(root, f), errors="ignore").read()
                assert "oracle" not in text.lower().replace("oracle/_ref", ""), f"{f} mentions the oracle"


def test_chain_validation_and_ugen_count(knh):
    c3 = configs.config("C3", n_voices=4, block_size=16)
    assert knh.chain_ugen_count(c3.stages) == 4      # SinWt(+WrMul), SvfFilter, EnvAsr, MathUGen Mul
    assert knh.chain_ugen_count(configs.config("C1").stages) == 3  # SinWt, Constant, MathUGen Mul
    with pytest.raises(L.KnasterHipError) as e:      # processor without a source
        knh.VoiceBank([Stage(L.STAGE_SVF)], 4)
    assert e.value.status == L.ERR_INVALID_ARGUMENT
    # a second source mid-list starts a second signal of the voice (a graph, not a chain): accepted since ABI 2
    knh.VoiceBank([Stage(L.STAGE_SIN_WT), Stage(L.STAGE_SIN_WT), Stage(L.STAGE_MATH_MUL, input=1, input2=2)], 4).close()
    for bad in ([Stage(L.STAGE_SIN_WT), Stage(L.STAGE_SIN_WT), Stage(L.STAGE_MATH_MUL)],                    # operands not named
                [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_MUL_CONST, input=3)],                                   # reads a later stage
                [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_MUL_CONST, input2=1)],                                  # input2 on a unary stage
                [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_SIN_WT, input=1)],                                      # a source reads nothing
                [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_SIN_WT), Stage(L.STAGE_WR_MUL, input=1)]):              # a wrapper wraps its neighbour
        with pytest.raises(L.KnasterHipError) as e:
            knh.VoiceBank(bad, 4)
        assert e.value.status == L.ERR_INVALID_ARGUMENT
    assert knh.chain_ugen_count([Stage(L.STAGE_SIN_WT), Stage(L.STAGE_SIN_WT), Stage(L.STAGE_MATH_MUL, input=1, input2=2)]) == 3
    # a valid chain without a pre-built kernel is accepted: it is fused at init time (hiprtc)
    knh.VoiceBank([Stage(L.STAGE_SIN_NUMERIC), Stage(L.STAGE_ONEPOLE_HPF), Stage(L.STAGE_ONEPOLE_LPF),
                   Stage(L.STAGE_DIV_CONST)], 4).close()
    with pytest.raises(L.KnasterHipError):
        knh.VoiceBank(c3.stages, 0)
    with pytest.raises(L.KnasterHipError):
        knh.VoiceBank(c3.stages, 4, out_channels=3)


def test_bank_metadata_without_device(knh):
    w = configs.config("C3", n_voices=4, block_size=16)
    b = knh.VoiceBank(w.stages, 4)
    assert b.inputs() == 0 and b.outputs() == 2
    assert b.stage_param_descriptions(0) == ["freq", "phase_offset", "reset_phase"]          # osc.rs:126-140
    assert b.stage_param_descriptions(1) == ["wr_mul"]                                       # math.rs:79
    assert b.stage_param_descriptions(2) == ["cutoff_freq", "q", "gain", "filter", "t_calculate_coefficients"]
    assert b.stage_param_descriptions(3) == ["attack_time", "release_time", "t_release", "t_restart"]
    assert b.algorithmic_bytes_per_voice_block() == (68, 24)  # 92 B per voice per block
    with pytest.raises(L.KnasterHipError) as e:
        b.set_ctor_args(0, np.zeros((5, 1)))  # more voices than the bank has
    assert e.value.status == L.ERR_OUT_OF_RANGE
    with pytest.raises(L.KnasterHipError) as e:
        b.set_ctor_args(2, np.zeros((4, 1)))  # SvfFilter::new takes 4 arguments
    assert e.value.status == L.ERR_INVALID_ARGUMENT
    with pytest.raises(L.KnasterHipError) as e:
        b.param_apply(0, 0, 0, 440.0)
    assert e.value.status == L.ERR_NOT_INITIALISED
    with pytest.raises(L.KnasterHipError) as e:
        b.process_block()
    assert e.value.status == L.ERR_NOT_INITIALISED
    b.close()
    w64 = configs.config("C4", n_voices=4, block_size=16)
    b = knh.VoiceBank(w64.stages, 4, L.F64)
    assert b.algorithmic_bytes_per_voice_block() == (136, 48)
    b.close()


def test_compute_fails_loudly_without_gpu(knh):
    if knh.lib.load().knh_device_count() > 0:
        pytest.skip("a gfx950 device is present")
    w = configs.config("C1")
    b = knh.VoiceBank(w.stages, 1)
    for s, a in w.ctor.items():
        b.set_ctor_args(s, a)
    with pytest.raises(L.KnasterHipError) as e:
        b.init(48000, 64)
    assert e.value.status == L.ERR_NO_DEVICE
    assert "no CPU path" in str(e.value)
    b.close()


def test_workload_inputs_are_deterministic():
    a = configs.voice_parameters(16)
    b = configs.voice_parameters(16)
    for k in a:
        assert np.array_equal(a[k], b[k])
    assert a["freq"].min() >= 55.0 and a["freq"].max() <= 3520.0
    assert a["cutoff"].max() < 24000.0
    w = configs.config("C3")
    assert (w.n_voices, w.block_size, w.sample_type) == (16384, 512, L.F32)
    assert abs(float(w.ctor[1].sum()) - 1.0) < 1e-9  # sum of gains = 1


def _header_decls():
    """name -> number of arguments, from the header's function declarations."""
    text = open(os.path.join(ROOT, "include", "knaster_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for name, args in re.findall(r"\b(knh_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        args = args.strip()
        out[name] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def _header_constants():
    text = open(os.path.join(ROOT, "include", "knaster_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    vals = {}
    for name, v in re.findall(r"\b(KNH_[A-Z0-9_]+)\s*=\s*(1u\s*<<\s*[0-9]+|[0-9]+)", text):
        vals[name] = eval(v.replace("u", ""))
    vals["KNH_ABI_VERSION"] = int(re.search(r"#define KNH_ABI_VERSION (\d+)", text).group(1))
    return vals


def test_rust_ffi_matches_header():
    """bindings/rust/knaster_hip/src/ffi.rs cannot be compiled here (no rustc): keep it honest by comparing every
    declared function (name, argument count) and every constant with include/knaster_hip.h."""
    src = open(os.path.join(ROOT, "bindings", "rust", "knaster_hip", "src", "ffi.rs")).read()
    rust_fns = {}
    for name, args in re.findall(r"pub fn (knh_[a-z0-9_]+)\(([^)]*)\)", src):
        args = args.strip()
        rust_fns[name] = 0 if not args else args.count(":")
    assert rust_fns == _header_decls()
    rust_consts = {n: eval(v) for n, v in re.findall(r"pub const (KNH_[A-Z0-9_]+): [iu](?:16|32) = ([0-9]+|1 << [0-9]+);", src)}
    header = _header_constants()
    assert rust_consts, "no constants parsed"
    for name, value in rust_consts.items():
        assert header.get(name) == value, f"{name}: ffi.rs {value} vs header {header.get(name)}"
    for name in header:
        assert name in rust_consts, f"{name} missing from ffi.rs"
    # struct layouts: field order of the two repr(C) structs
    text = open(os.path.join(ROOT, "include", "knaster_hip.h")).read()
    for struct in ("knh_stage_desc", "knh_bank_desc"):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), text, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        c_fields = re.findall(r"\b([a-z_]+)\s*;", body)
        r_body = re.search(r"pub struct %s \{(.*?)\n\}" % struct, src, flags=re.S).group(1)
        r_fields = re.findall(r"pub ([a-z_]+):", r_body)
        assert c_fields == r_fields, struct


def test_graph_voice_stage_limits_without_device(knh):
    """Checked at knh_bank_create, before any device is touched: a fused graph-shaped voice holds at most 512 stages; one made of
    SinWt oscillators and arithmetic alone (it runs a lane per frame) up to 4 096 -- the reference's
    256-oscillator FM cascade is 1 531."""
    w = configs.fm_cascade(256, 1, 128)
    assert len(w.stages) == 1531
    knh.VoiceBank(w.stages, 1, L.F32, 1).close()
    assert knh.chain_ugen_count(w.stages) == 2041  # 256 SinWt + 510 Constants + 1 275 MathUGens (graph_dsp_performance.rs:37-72)
    too_long = configs.fm_cascade(700, 1, 128)
    with pytest.raises(L.KnasterHipError) as e:
        knh.VoiceBank(too_long.stages, 1, L.F32, 1)
    assert e.value.status == L.ERR_UNSUPPORTED_CHAIN
    st = [Stage(L.STAGE_SIN_WT)] + [Stage(L.STAGE_ONEPOLE_LPF), Stage(L.STAGE_MUL_CONST)] * 300
    st += [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_MATH_ADD, input=len(st), input2=len(st) + 1)]  # a filter in it: fused, and too long for that
    with pytest.raises(L.KnasterHipError) as e:
        knh.VoiceBank(st, 1, L.F32, 1)
    assert e.value.status == L.ERR_UNSUPPORTED_CHAIN


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference sources are only in the build container")
def test_rust_sources_name_items_the_reference_has():
    """bindings/rust/knaster_hip (the shim and examples/dump_golden.rs, which renders the golden cases with the real crate)
    cannot be compiled here: keep their `use` lines honest by looking every imported item up in the reference's sources, read
    as text -- the module file must exist and declare the name (`pub struct|enum|trait|fn|type|mod|const X`, or re-export it)."""
    crates = {"knaster_core": "knaster_core/src", "knaster_core_dsp": "knaster_core_dsp/src", "knaster_graph": "knaster_graph/src"}
    checked = 0
    for rel in ("examples/dump_golden.rs", "src/lib.rs"):
        src = open(os.path.join(ROOT, "bindings", "rust", "knaster_hip", rel)).read()
        for path, names in re.findall(r"^\s*use ((?:knaster_[a-z_]+)(?:::[a-z_0-9]+)*)::(\{[^}]*\}|[A-Za-z_0-9]+);", src, flags=re.M):
            parts = path.split("::")
            crate, mods = parts[0], parts[1:]
            if crate not in crates:
                continue
            items = [n.strip().split(" as ")[0] for n in names.strip("{}").split(",") if n.strip()]
            base = os.path.join(REFERENCE, crates[crate])
            # the module's file: src/a/b.rs, src/a/b/mod.rs, or (crate root / re-exported `ugens::*`) anywhere below src/
            candidates = []
            if mods and mods[0] != "typenum":
                for stem in (os.path.join(base, *mods), os.path.join(base, "ugens", *mods)):
                    candidates += [stem + ".rs", os.path.join(stem, "mod.rs")]
                candidates = [c for c in candidates if os.path.isfile(c)]
                assert candidates, f"{rel}: no module file for {path} under {base}"
            else:
                # (knaster_core's root re-exports knaster_primitives wholesale: `pub use knaster_primitives::*`, lib.rs:42)
                for b in [base] + ([os.path.join(REFERENCE, "knaster_primitives/src")] if crate == "knaster_core" else []):
                    for d, _, files in os.walk(b):
                        candidates += [os.path.join(d, f) for f in files if f.endswith(".rs")]
            text = "\n".join(open(c).read() for c in candidates)
            for item in items:
                if mods and mods[0] == "typenum":  # a re-exported third-party crate: only that the re-export exists
                    assert re.search(r"pub use typenum|pub extern crate typenum|pub use [a-z_:]*typenum", text), f"{rel}: {crate} does not re-export typenum"
                    continue
                pat = r"pub (?:struct|enum|trait|fn|type|mod|const|use [A-Za-z_0-9:{}, ]*)\s*\b%s\b" % re.escape(item)
                assert re.search(pat, text), f"{rel}: `{item}` (use {path}::..) is not declared in {[os.path.relpath(c, REFERENCE) for c in candidates][:4]}"
                checked += 1
    assert checked >= 10, checked
