#!/usr/bin/env python3
"""Same-box A/B of the C3 kernel between two builds of the library (boxes differ by ~2 %, so kernel changes of that size only
show in alternating runs on one box):   python tools/ab_c3.py tools/ab/libknaster_hip_r01.so knaster_amd/csrc/libknaster_hip.so
Loads each .so with ctypes directly (the old build has ABI 1: 8-byte stage descriptors), runs the bench's note cycle in 64-block
launches, and prints the voice kernel's mean duration per launch (HIP events inside the library), alternating A B A B."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from knaster_amd import configs


def open_lib(path):
    lib = C.CDLL(os.path.abspath(path))
    lib.knh_abi_version.restype = C.c_uint32
    abi = lib.knh_abi_version()
    fields = [("kind", C.c_uint16), ("flags", C.c_uint16), ("dcpb", C.c_uint16), ("reserved", C.c_uint16)]
    if abi >= 2:
        fields += [("input", C.c_uint16), ("input2", C.c_uint16)]
    Stage = type("Stage", (C.Structure,), {"_fields_": fields})

    class Desc(C.Structure):
        _fields_ = [("abi", C.c_uint32), ("n_voices", C.c_uint32), ("sample_type", C.c_uint32), ("n_stages", C.c_uint32),
                    ("stages", C.POINTER(Stage)), ("out_channels", C.c_uint32), ("mix_mode", C.c_uint32), ("device", C.c_int32), ("allow_fma", C.c_uint32),
                    ("in_channels", C.c_uint32)]  # (ABI 2; an ABI 1 library does not read that far)
    lib.knh_bank_create.argtypes = [C.POINTER(Desc), C.POINTER(C.c_void_p)]
    lib.knh_bank_set_ctor_args.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
    lib.knh_bank_init.argtypes = [C.c_void_p, C.c_uint32, C.c_size_t]
    lib.knh_bank_param_apply_many_at.argtypes = [C.c_void_p, C.c_uint32, C.c_size_t] + [C.c_void_p] * 7
    lib.knh_bank_process_blocks_device.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.knh_bank_synchronize.argtypes = [C.c_void_p]
    lib.knh_bank_timing_reset.argtypes = [C.c_void_p, C.c_int32]
    lib.knh_bank_timing_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.knh_bank_destroy.argtypes = [C.c_void_p]
    lib.knh_last_error.restype = C.c_char_p
    lib.knh_last_error.argtypes = [C.c_void_p]
    return lib, Stage, Desc, abi


def make(libt, w):
    lib, Stage, Desc, abi = libt
    arr = (Stage * len(w.stages))()
    for i, s in enumerate(w.stages):
        arr[i].kind, arr[i].flags, arr[i].dcpb = s.kind, s.flags, s.delayed_changes_per_block
    d = Desc(abi, w.n_voices, w.sample_type, len(w.stages), arr, w.out_channels, 0, -1, 0, 0)
    h = C.c_void_p()
    assert lib.knh_bank_create(C.byref(d), C.byref(h)) == 0, lib.knh_last_error(None)
    for s, a in w.ctor.items():
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert lib.knh_bank_set_ctor_args(h, s, 0, a.shape[0], a.ctypes.data_as(C.c_void_p), a.shape[1]) == 0
    assert lib.knh_bank_init(h, 48000, w.block_size) == 0, lib.knh_last_error(h)
    return h


def run(libt, h, w, launches):
    lib = libt[0]
    n = w.n_voices
    v = np.arange(n, dtype=np.uint32)
    k = np.full(n, 1, dtype=np.uint32)  # trigger
    def fire(stage, param, block):
        s = np.full(n, stage, dtype=np.uint32); p = np.full(n, param, dtype=np.uint32)
        assert lib.knh_bank_param_apply_many_at(h, block, n, v.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p),
                                                k.ctypes.data_as(C.c_void_p), None, None, None) == 0
    for _ in range(launches):
        fire(w.restart[0], w.restart[1], 0)
        fire(w.release[0], w.release[1], 32)
        assert lib.knh_bank_process_blocks_device(h, 64, 0, None, None) == 0, lib.knh_last_error(h)
    lib.knh_bank_synchronize(h)


def main():
    # "lib.so@KNH_PIPE_BIG=1": that environment switch is set while the bank of that side is created (the library reads its
    # switches in knh_bank_create / knh_bank_init), so two forms of one build can be compared too
    specs = [a.split("@") for a in sys.argv[1:3]]
    paths = [sp[0] for sp in specs]
    name = sys.argv[3] if len(sys.argv) > 3 else "C3"
    w = configs.config(name)
    libs = [open_lib(p) for p in paths]
    banks = []
    for l, sp in zip(libs, specs):
        sets = dict(kv.split("=", 1) for kv in sp[1:])
        os.environ.update(sets)
        banks.append(make(l, w))
        for k in sets:
            del os.environ[k]
    for l, h in zip(libs, banks):
        run(l, h, w, 200)  # clocks up, both warmed
    for rnd in range(4):
        for tag, l, h in zip("AB", libs, banks):
            l[0].knh_bank_timing_reset(h, 1)
            run(l, h, w, 100)
            ms, cnt = C.c_double(0), C.c_uint64(0)
            l[0].knh_bank_timing_read(h, C.byref(ms), C.byref(cnt))
            print(f"round {rnd} {tag} ({os.path.basename(sys.argv[1 + 'AB'.index(tag)])}): {ms.value / max(cnt.value, 1):.4f} ms per 64-block launch", flush=True)
    for l, h in zip(libs, banks):
        l[0].knh_bank_destroy(h)


if __name__ == "__main__":
    main()
