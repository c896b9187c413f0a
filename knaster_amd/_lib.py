"""ctypes binding of include/knaster_hip.h.  Plumbing only: every entry point is the C ABI's.

Loading fails loudly when the HIP library has not been built -- there is no
Python or CPU fallback for the hot path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KNH_LIB: another build of the same library (the stamped diagnostic build, an older one for A/B runs); never a fallback
LIB_PATH = os.environ.get("KNH_LIB") or os.path.join(_HERE, "csrc", "libknaster_hip.so")

KNH_ABI_VERSION = 4

# knh_status
OK, ERR_INVALID_ARGUMENT, ERR_OUT_OF_RANGE, ERR_UNSUPPORTED_CHAIN, ERR_DEVICE = 0, 1, 2, 3, 4
ERR_NOT_INITIALISED, ERR_NO_DEVICE, ERR_WRONG_VALUE_KIND, ERR_OUT_OF_MEMORY, ERR_INTERNAL = 5, 6, 7, 8, 9
# knh_sample_type
F32, F64 = 0, 1
# knh_value_kind
VALUE_FLOAT, VALUE_TRIGGER, VALUE_INTEGER, VALUE_BOOL, VALUE_SMOOTHING = 0, 1, 2, 3, 4
# knh_stage_kind
(STAGE_SIN_WT, STAGE_SIN_NUMERIC, STAGE_SVF, STAGE_ONEPOLE_LPF, STAGE_ONEPOLE_HPF, STAGE_MUL_ENV_ASR,
 STAGE_MUL_ENV_AR, STAGE_MUL_CONST, STAGE_ADD_CONST, STAGE_SUB_CONST, STAGE_DIV_CONST, STAGE_WR_MUL,
 STAGE_WR_ADD, STAGE_WR_SUB, STAGE_MUL_ENVELOPE, STAGE_WR_VSUB, STAGE_WR_DIV, STAGE_WR_VDIV, STAGE_WR_POWF,
 STAGE_WR_POWI, STAGE_POW_CONST, STAGE_SAMPLE_DELAY, STAGE_PHASOR, STAGE_SAFETY_LIMITER, STAGE_POLYBLEP, STAGE_ALLPASS_DELAY, STAGE_ALLPASS_FB_DELAY, STAGE_BUFFER_READER,
 STAGE_WHITE_NOISE, STAGE_PINK_NOISE, STAGE_BROWN_NOISE, STAGE_RANDOM_LIN, STAGE_PAN2,
 STAGE_MATH_ADD, STAGE_MATH_SUB, STAGE_MATH_MUL, STAGE_MATH_DIV, STAGE_MATH_POW, STAGE_INPUT) = range(39)
STAGE_FLAG_AR_FREQ = 1
STAGE_FLAG_SMOOTH_PARAMS = 2
# knh_svf_type
SVF_LOW, SVF_HIGH, SVF_BAND, SVF_NOTCH, SVF_PEAK, SVF_ALL, SVF_BELL, SVF_LOW_SHELF, SVF_HIGH_SHELF = range(9)
# knh_mix_mode
MIX_TREE, MIX_LEFT_FOLD = 0, 1
FLAG_ANY_DONE, FLAG_ALL_DONE = 1, 2

# constructor-argument count per stage kind (table in knaster_hip.h)
STAGE_CTOR_ARGS = {  # STAGE_MUL_ENVELOPE takes 4 + 2 * n_max (variable)
    STAGE_SIN_WT: 1, STAGE_SIN_NUMERIC: 1, STAGE_SVF: 4, STAGE_ONEPOLE_LPF: 1, STAGE_ONEPOLE_HPF: 0,
    STAGE_MUL_ENV_ASR: 2, STAGE_MUL_ENV_AR: 2, STAGE_MUL_CONST: 1, STAGE_ADD_CONST: 1, STAGE_SUB_CONST: 1,
    STAGE_DIV_CONST: 1, STAGE_WR_MUL: 1, STAGE_WR_ADD: 1, STAGE_WR_SUB: 1,
    STAGE_WR_VSUB: 1, STAGE_WR_DIV: 1, STAGE_WR_VDIV: 1, STAGE_WR_POWF: 1, STAGE_WR_POWI: 1, STAGE_POW_CONST: 1, STAGE_SAMPLE_DELAY: 1, STAGE_PHASOR: 1, STAGE_SAFETY_LIMITER: 0, STAGE_POLYBLEP: 2, STAGE_ALLPASS_DELAY: 1, STAGE_ALLPASS_FB_DELAY: 1, STAGE_BUFFER_READER: 3,
    STAGE_WHITE_NOISE: 1, STAGE_PINK_NOISE: 1, STAGE_BROWN_NOISE: 1, STAGE_RANDOM_LIN: 2, STAGE_PAN2: 1,
    STAGE_MATH_ADD: 0, STAGE_MATH_SUB: 0, STAGE_MATH_MUL: 0, STAGE_MATH_DIV: 0, STAGE_MATH_POW: 0, STAGE_INPUT: 1,
}


class StageDesc(C.Structure):
    _fields_ = [("kind", C.c_uint16), ("flags", C.c_uint16), ("delayed_changes_per_block", C.c_uint16),
                ("ar_param", C.c_uint16), ("input", C.c_uint16), ("input2", C.c_uint16)]


class BankDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("n_voices", C.c_uint32), ("sample_type", C.c_uint32),
                ("n_stages", C.c_uint32), ("stages", C.POINTER(StageDesc)), ("out_channels", C.c_uint32),
                ("mix_mode", C.c_uint32), ("device", C.c_int32), ("allow_fma", C.c_uint32), ("in_channels", C.c_uint32)]


# name -> (restype, argtypes): exactly the declarations of include/knaster_hip.h
PROTOTYPES = {
    "knh_abi_version": (C.c_uint32, []),
    "knh_device_count": (C.c_int32, []),
    "knh_status_string": (C.c_char_p, [C.c_int32]),
    "knh_last_error": (C.c_char_p, [C.c_void_p]),
    "knh_chain_ugen_count": (C.c_int32, [C.POINTER(StageDesc), C.c_uint32]),
    "knh_bank_create": (C.c_int32, [C.POINTER(BankDesc), C.POINTER(C.c_void_p)]),
    "knh_bank_create_sharded": (C.c_int32, [C.POINTER(BankDesc), C.c_uint32, C.POINTER(C.c_void_p)]),
    "knh_bank_set_ctor_args": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]),
    "knh_bank_set_buffer": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_double]),
    "knh_bank_init": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_size_t]),
    "knh_bank_destroy": (None, [C.c_void_p]),
    "knh_bank_inputs": (C.c_uint16, [C.c_void_p]),
    "knh_bank_outputs": (C.c_uint16, [C.c_void_p]),
    "knh_bank_stage_parameters": (C.c_uint16, [C.c_void_p, C.c_uint32]),
    "knh_bank_stage_param_description": (C.c_char_p, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "knh_bank_param_apply": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_int64]),
    "knh_bank_set_delay_within_block_for_param": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint16]),
    "knh_bank_param_apply_many": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p]),
    "knh_bank_param_apply_range": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_int64]),
    "knh_bank_process_block": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint32)]),
    "knh_jit_stats": (None, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "knh_bank_resident_stats": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "knh_bank_resident_trace": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "knh_bank_process_block_channels": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]),
    "knh_bank_process_block_device": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_void_p, C.c_void_p]),
    "knh_bank_process_block_voices": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_void_p, C.c_void_p,
                                                 C.POINTER(C.c_uint32)]),
    "knh_bank_process_blocks": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint32)]),
    "knh_bank_process_blocks_device": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]),
    "knh_bank_process_blocks_device_add": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]),
    "knh_device_malloc": (C.c_void_p, [C.c_size_t, C.c_int32]),
    "knh_device_free": (None, [C.c_void_p]),
    "knh_device_read": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "knh_bank_param_apply_many_at": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "knh_bank_read_done_frames": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "knh_bank_debug_words": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "knh_bank_debug_signature": (C.c_char_p, [C.c_void_p]),
    "knh_bank_synchronize": (C.c_int32, [C.c_void_p]),
    "knh_bank_timing_reset": (C.c_int32, [C.c_void_p, C.c_int32]),
    "knh_bank_timing_read": (C.c_int32, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "knh_bank_collective_timing_read": (C.c_int32, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "knh_bank_algorithmic_bytes_per_voice_block": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
}

COMM_ID_BYTES = 128
# knh_reduce_fn: int32 (*)(void* user, void* device_buf, size_t count, uint32 sample_type, uint32 root, void* hip_stream)
REDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p)
PROTOTYPES.update({
    "knh_bank_process_blocks_begin": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64]),
    "knh_bank_process_blocks_end": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "knh_bank_set_input": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "knh_bank_set_input_device": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "knh_bank_create_multi_device": (C.c_int32, [C.POINTER(BankDesc), C.POINTER(C.c_int32), C.c_uint32, C.POINTER(C.c_void_p)]),
    "knh_comm_unique_id": (C.c_int32, [C.c_void_p]),
    "knh_bank_create_rank": (C.c_int32, [C.POINTER(BankDesc), C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "knh_bank_create_rank_custom": (C.c_int32, [C.POINTER(BankDesc), C.c_uint32, C.c_uint32, REDUCE_FN, C.c_void_p, C.POINTER(C.c_void_p)]),
    "knh_shard_voice_range": (C.c_int32, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "knh_bank_ranks": (C.c_uint32, [C.c_void_p]),
    "knh_comm_create": (C.c_int32, [C.c_uint32, C.c_uint32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "knh_comm_destroy": (None, [C.c_void_p]),
    "knh_comm_last_error": (C.c_char_p, [C.c_void_p]),
    "knh_comm_world": (C.c_uint32, [C.c_void_p]),
    "knh_comm_rccl_version": (C.c_int32, []),
    "knh_comm_reduce_sum": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p]),
    "knh_comm_wait_buffer": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "knh_comm_wait": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "knh_comm_synchronize": (C.c_int32, [C.c_void_p]),
    "knh_comm_timing_reset": (C.c_int32, [C.c_void_p, C.c_int32]),
    "knh_comm_timing_read": (C.c_int32, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
})

_lib = None


class KnasterHipError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"knaster_hip status {status}: {message}")
        self.status = status


def load() -> C.CDLL:
    """dlopen csrc/libknaster_hip.so and bind every symbol of the header.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m knaster_amd.build` "
            "(hipcc --offload-arch=gfx950).  knaster_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.knh_abi_version() != KNH_ABI_VERSION:
        raise ImportError("libknaster_hip.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib
