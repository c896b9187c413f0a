//! Raw declarations of include/knaster_hip.h (ABI version 3).  One `pub fn` per exported symbol, same order as
//! the header; tests/test_abi.py fails when the two drift apart.
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_void};

#[repr(C)]
pub struct knh_bank {
    _private: [u8; 0],
}
/// An RCCL communicator wrapped by the library (one process per GPU).
#[repr(C)]
pub struct knh_comm {
    _private: [u8; 0],
}
pub const KNH_COMM_ID_BYTES: usize = 128;
/// The host's own sum-reduce for `knh_bank_create_rank_custom`.
pub type knh_reduce_fn = Option<unsafe extern "C" fn(user: *mut c_void, device_buf: *mut c_void, count: usize, sample_type: u32, root: u32, hip_stream: *mut c_void) -> i32>;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct knh_stage_desc {
    pub kind: u16,
    pub flags: u16,
    pub delayed_changes_per_block: u16,
    /// 1 + the float parameter that the signal `input2` drives at audio rate (`.ar_params()` + `link`), 0: none
    pub ar_param: u16,
    /// 0: the stage reads the output of the stage before it; k > 0: the output of stage k - 1
    pub input: u16,
    /// second operand of the KNH_STAGE_MATH_* stages (same numbering), 0 elsewhere
    pub input2: u16,
}

#[repr(C)]
pub struct knh_bank_desc {
    pub abi_version: u32,
    pub n_voices: u32,
    pub sample_type: u32,
    pub n_stages: u32,
    pub stages: *const knh_stage_desc,
    pub out_channels: u32,
    pub mix_mode: u32,
    pub device: i32,
    pub allow_fma: u32,
    /// UGen::Inputs of the bank node: channels the voices read through KNH_STAGE_INPUT stages
    pub in_channels: u32,
}

pub const KNH_ABI_VERSION: u32 = 4;

// knh_status
pub const KNH_OK: i32 = 0;
pub const KNH_ERR_INVALID_ARGUMENT: i32 = 1;
pub const KNH_ERR_OUT_OF_RANGE: i32 = 2;
pub const KNH_ERR_UNSUPPORTED_CHAIN: i32 = 3;
pub const KNH_ERR_DEVICE: i32 = 4;
pub const KNH_ERR_NOT_INITIALISED: i32 = 5;
pub const KNH_ERR_NO_DEVICE: i32 = 6;
pub const KNH_ERR_WRONG_VALUE_KIND: i32 = 7;
pub const KNH_ERR_OUT_OF_MEMORY: i32 = 8;
pub const KNH_ERR_INTERNAL: i32 = 9;

// knh_sample_type
pub const KNH_F32: u32 = 0;
pub const KNH_F64: u32 = 1;

// knh_value_kind
pub const KNH_VALUE_FLOAT: u32 = 0;
pub const KNH_VALUE_TRIGGER: u32 = 1;
pub const KNH_VALUE_INTEGER: u32 = 2;
pub const KNH_VALUE_BOOL: u32 = 3;
pub const KNH_VALUE_SMOOTHING: u32 = 4;

// knh_stage_kind
pub const KNH_STAGE_SIN_WT: u16 = 0;
pub const KNH_STAGE_SIN_NUMERIC: u16 = 1;
pub const KNH_STAGE_SVF: u16 = 2;
pub const KNH_STAGE_ONEPOLE_LPF: u16 = 3;
pub const KNH_STAGE_ONEPOLE_HPF: u16 = 4;
pub const KNH_STAGE_MUL_ENV_ASR: u16 = 5;
pub const KNH_STAGE_MUL_ENV_AR: u16 = 6;
pub const KNH_STAGE_MUL_CONST: u16 = 7;
pub const KNH_STAGE_ADD_CONST: u16 = 8;
pub const KNH_STAGE_SUB_CONST: u16 = 9;
pub const KNH_STAGE_DIV_CONST: u16 = 10;
pub const KNH_STAGE_WR_MUL: u16 = 11;
pub const KNH_STAGE_WR_ADD: u16 = 12;
pub const KNH_STAGE_WR_SUB: u16 = 13;
pub const KNH_STAGE_MUL_ENVELOPE: u16 = 14;
pub const KNH_STAGE_WR_VSUB: u16 = 15;
pub const KNH_STAGE_WR_DIV: u16 = 16;
pub const KNH_STAGE_WR_VDIV: u16 = 17;
pub const KNH_STAGE_WR_POWF: u16 = 18;
pub const KNH_STAGE_WR_POWI: u16 = 19;
pub const KNH_STAGE_POW_CONST: u16 = 20;
pub const KNH_STAGE_SAMPLE_DELAY: u16 = 21;
pub const KNH_STAGE_PHASOR: u16 = 22;
pub const KNH_STAGE_SAFETY_LIMITER: u16 = 23;
pub const KNH_STAGE_POLYBLEP: u16 = 24;
pub const KNH_STAGE_ALLPASS_DELAY: u16 = 25;
pub const KNH_STAGE_ALLPASS_FB_DELAY: u16 = 26;
pub const KNH_STAGE_BUFFER_READER: u16 = 27;
pub const KNH_STAGE_WHITE_NOISE: u16 = 28;
pub const KNH_STAGE_PINK_NOISE: u16 = 29;
pub const KNH_STAGE_BROWN_NOISE: u16 = 30;
pub const KNH_STAGE_RANDOM_LIN: u16 = 31;
pub const KNH_STAGE_PAN2: u16 = 32;
pub const KNH_STAGE_MATH_ADD: u16 = 33;
pub const KNH_STAGE_MATH_SUB: u16 = 34;
pub const KNH_STAGE_MATH_MUL: u16 = 35;
pub const KNH_STAGE_MATH_DIV: u16 = 36;
pub const KNH_STAGE_MATH_POW: u16 = 37;
pub const KNH_STAGE_INPUT: u16 = 38;
pub const KNH_STAGE_KIND_COUNT: u16 = 39;

// knh_svf_type = SvfFilterType, knaster_core_dsp/src/ugens/svf.rs:19-39
pub const KNH_SVF_LOW: u32 = 0;
pub const KNH_SVF_HIGH: u32 = 1;
pub const KNH_SVF_BAND: u32 = 2;
pub const KNH_SVF_NOTCH: u32 = 3;
pub const KNH_SVF_PEAK: u32 = 4;
pub const KNH_SVF_ALL: u32 = 5;
pub const KNH_SVF_BELL: u32 = 6;
pub const KNH_SVF_LOW_SHELF: u32 = 7;
pub const KNH_SVF_HIGH_SHELF: u32 = 8;

pub const KNH_STAGE_FLAG_AR_FREQ: u16 = 1 << 0;
pub const KNH_STAGE_FLAG_SMOOTH_PARAMS: u16 = 1 << 1;

// knh_mix_mode
pub const KNH_MIX_TREE: u32 = 0;
pub const KNH_MIX_LEFT_FOLD: u32 = 1;

pub const KNH_FLAG_ANY_DONE: u32 = 1 << 0;
pub const KNH_FLAG_ALL_DONE: u32 = 1 << 1;

#[link(name = "knaster_hip")]
unsafe extern "C" {
    pub fn knh_abi_version() -> u32;
    pub fn knh_device_count() -> i32;
    pub fn knh_status_string(status: i32) -> *const c_char;
    pub fn knh_last_error(bank: *const knh_bank) -> *const c_char;
    pub fn knh_chain_ugen_count(stages: *const knh_stage_desc, n_stages: u32) -> i32;
    pub fn knh_bank_create(desc: *const knh_bank_desc, out_bank: *mut *mut knh_bank) -> i32;
    pub fn knh_bank_create_sharded(desc: *const knh_bank_desc, host_threads: u32, out_bank: *mut *mut knh_bank) -> i32;
    pub fn knh_bank_set_ctor_args(bank: *mut knh_bank, stage: u32, first_voice: u32, count: u32, args: *const f64, n_args: u32) -> i32;
    pub fn knh_bank_set_buffer(bank: *mut knh_bank, stage: u32, samples: *const c_void, n_frames: usize, buffer_sample_rate: f64) -> i32;
    pub fn knh_bank_init(bank: *mut knh_bank, sample_rate: u32, block_size: usize) -> i32;
    pub fn knh_bank_destroy(bank: *mut knh_bank);
    pub fn knh_bank_inputs(bank: *const knh_bank) -> u16;
    pub fn knh_bank_outputs(bank: *const knh_bank) -> u16;
    pub fn knh_bank_stage_parameters(bank: *const knh_bank, stage: u32) -> u16;
    pub fn knh_bank_stage_param_description(bank: *const knh_bank, stage: u32, param: u32) -> *const c_char;
    pub fn knh_bank_param_apply(bank: *mut knh_bank, voice: u32, stage: u32, param: u32, kind: u32, fvalue: f64, ivalue: i64) -> i32;
    pub fn knh_bank_set_delay_within_block_for_param(bank: *mut knh_bank, voice: u32, stage: u32, param: u32, delay: u16) -> i32;
    pub fn knh_bank_param_apply_range(bank: *mut knh_bank, voice_begin: u32, voice_end: u32, stage: u32, param: u32, kind: u32, fvalue: f64, ivalue: i64) -> i32;
    pub fn knh_bank_param_apply_many(bank: *mut knh_bank, count: usize, voices: *const u32, stages: *const u32, params: *const u32, kinds: *const u32, fvalues: *const f64, ivalues: *const i64, delays: *const u16) -> i32;
    pub fn knh_bank_process_block(bank: *mut knh_bank, frames_to_process: usize, block_start_offset: usize, frame_clock: u64, out: *mut c_void, out_flags: *mut u32) -> i32;
    pub fn knh_jit_stats(memory_hits: *mut u64, disk_hits: *mut u64, helper_compiles: *mut u64, in_process_compiles: *mut u64);
    pub fn knh_bank_resident_stats(bank: *mut knh_bank, calls: *mut u64, launches: *mut u64) -> i32;
    pub fn knh_bank_resident_trace(bank: *mut knh_bank, ticks5: *mut u64) -> i32;
    pub fn knh_bank_process_block_channels(bank: *mut knh_bank, frames_to_process: usize, block_start_offset: usize, frame_clock: u64, out_channels: *const *mut c_void, out_flags: *mut u32) -> i32;
    pub fn knh_bank_process_block_device(bank: *mut knh_bank, frames_to_process: usize, block_start_offset: usize, frame_clock: u64, out_device: *mut c_void, hip_stream: *mut c_void) -> i32;
    pub fn knh_bank_process_block_voices(bank: *mut knh_bank, frames_to_process: usize, block_start_offset: usize, frame_clock: u64, out: *mut c_void, voices_out: *mut c_void, out_flags: *mut u32) -> i32;
    pub fn knh_bank_process_blocks(bank: *mut knh_bank, n_blocks: u32, frame_clock: u64, out: *mut c_void, out_flags: *mut u32) -> i32;
    pub fn knh_bank_process_blocks_device(bank: *mut knh_bank, n_blocks: u32, frame_clock: u64, out_device: *mut c_void, hip_stream: *mut c_void) -> i32;
    pub fn knh_bank_process_blocks_device_add(bank: *mut knh_bank, n_blocks: u32, frame_clock: u64, out_device: *mut c_void, hip_stream: *mut c_void) -> i32;
    pub fn knh_device_malloc(bytes: usize, device: i32) -> *mut c_void;
    pub fn knh_device_free(device_ptr: *mut c_void);
    pub fn knh_device_read(dst_host: *mut c_void, src_device: *const c_void, bytes: usize, hip_stream: *mut c_void) -> i32;
    pub fn knh_bank_param_apply_many_at(bank: *mut knh_bank, block_offset: u32, count: usize, voices: *const u32, stages: *const u32, params: *const u32, kinds: *const u32, fvalues: *const f64, ivalues: *const i64, delays: *const u16) -> i32;
    pub fn knh_bank_read_done_frames(bank: *mut knh_bank, done_frames: *mut u32) -> i32;
    pub fn knh_bank_debug_words(bank: *mut knh_bank, out16: *mut u32) -> i32;
    pub fn knh_bank_debug_signature(bank: *const knh_bank) -> *const core::ffi::c_char;
    pub fn knh_bank_synchronize(bank: *mut knh_bank) -> i32;
    pub fn knh_bank_timing_reset(bank: *mut knh_bank, enable: i32) -> i32;
    pub fn knh_bank_timing_read(bank: *mut knh_bank, kernel_ms: *mut f64, launches: *mut u64) -> i32;
    pub fn knh_bank_collective_timing_read(bank: *mut knh_bank, reduce_ms: *mut f64, reduces: *mut u64) -> i32;
    pub fn knh_bank_algorithmic_bytes_per_voice_block(bank: *const knh_bank, read_bytes: *mut u32, write_bytes: *mut u32) -> i32;
    // several GPUs of one node: one process owning them all, or one process per GPU with an RCCL reduce
    pub fn knh_bank_process_blocks_begin(bank: *mut knh_bank, n_blocks: u32, frame_clock: u64) -> i32;
    pub fn knh_bank_process_blocks_end(bank: *mut knh_bank, out: *mut c_void) -> i32;
    pub fn knh_bank_set_input(bank: *mut knh_bank, n_blocks: u32, input: *const c_void) -> i32;
    pub fn knh_bank_set_input_device(bank: *mut knh_bank, n_blocks: u32, in_device: *const c_void) -> i32;
    pub fn knh_bank_create_multi_device(desc: *const knh_bank_desc, devices: *const i32, n_devices: u32, out_bank: *mut *mut knh_bank) -> i32;
    pub fn knh_comm_unique_id(id: *mut u8) -> i32;
    pub fn knh_bank_create_rank(desc: *const knh_bank_desc, rank: u32, world: u32, comm_id: *const u8, out_bank: *mut *mut knh_bank) -> i32;
    pub fn knh_bank_create_rank_custom(desc: *const knh_bank_desc, rank: u32, world: u32, reduce: knh_reduce_fn, user: *mut c_void, out_bank: *mut *mut knh_bank) -> i32;
    pub fn knh_shard_voice_range(n_voices: u32, rank: u32, world: u32, first: *mut u32, count: *mut u32) -> i32;
    pub fn knh_bank_ranks(bank: *const knh_bank) -> u32;
    pub fn knh_comm_create(rank: u32, world: u32, id: *const u8, device: i32, out_comm: *mut *mut knh_comm) -> i32;
    pub fn knh_comm_destroy(comm: *mut knh_comm);
    pub fn knh_comm_last_error(comm: *const knh_comm) -> *const c_char;
    pub fn knh_comm_world(comm: *const knh_comm) -> u32;
    pub fn knh_comm_rccl_version() -> i32;
    pub fn knh_comm_reduce_sum(comm: *mut knh_comm, buf: *mut c_void, count: usize, sample_type: u32, root: u32, after_stream: *mut c_void) -> i32;
    pub fn knh_comm_wait_buffer(comm: *mut knh_comm, buf: *const c_void, stream: *mut c_void) -> i32;
    pub fn knh_comm_wait(comm: *mut knh_comm, stream: *mut c_void) -> i32;
    pub fn knh_comm_synchronize(comm: *mut knh_comm) -> i32;
    pub fn knh_comm_timing_reset(comm: *mut knh_comm, enable: i32) -> i32;
    pub fn knh_comm_timing_read(comm: *mut knh_comm, reduce_ms: *mut f64, reduces: *mut u64) -> i32;
}
