//! `GpuVoiceBank`: one Knaster node (`Inputs = I`, any typenum, `U0` by default; `Outputs = U2`) that evaluates N
//! independent voice chains on an MI355X through `libknaster_hip.so` (C ABI: `include/knaster_hip.h`).
//!
//! It implements `knaster_core::UGen` (knaster_core/src/ugen.rs:232-369) by forwarding every call of the
//! audio thread to one FFI call.  Pushed like any UGen:
//!
//! ```ignore
//! let bank = graph.edit(|g| { let b = g.push(GpuVoiceBank::<f32>::new(&chain, n, &ctor)?); b.to_graph_out(); b });
//! ```
//!
//! It stands for the ~6·N nodes the same patch costs on the CPU (per voice: SinWt, SvfFilter, EnvAsr,
//! MathUGen Mul, plus the MathUGen Add output chain, knaster_graph/src/graph.rs:827-872).
//!
//! ## Parameters
//! A bank holds N voices × S stages × up to `MAX_PARAMS` parameters; `UGen::Parameters` is a compile-time
//! typenum and the graph stores it as `u16` (`node.data.parameters`, graph.rs:1368), so the bank declares
//! `Parameters = U0` and is addressed through a *flat index*
//! `index = (voice * S + stage) * MAX_PARAMS + param`.  The audio-thread path does not bound-check indices
//! (`apply_parameter_change`, graph_gen.rs:269-305 → `DynUGen::param_apply`, dynugen.rs:99-101), and neither does
//! the typed `Handle<T>` returned by `Graph::push` (graph.rs:370) for `Param::Index` (`HandleTrait::set`,
//! handle.rs:218-222).  `graph.set(node, index, ..)` (graph.rs:1368) and `GraphEdit`'s `Parameter` (a `u16`
//! index checked against `Parameters::USIZE`, graph_edit.rs:769-781) do check, and therefore cannot reach a bank:
//!
//! ```ignore
//! let i = bank.index(voice, stage, "cutoff_freq")?;     // before the bank is moved into the graph
//! let handle = graph.push(bank);
//! handle.set((i, 800.0))?;
//! ```
//!
//! Lifting that restriction needs one of two small changes on the Knaster side: widen `Parameter::param_index`
//! to `usize`, or let a UGen report its parameter count at run time (`DynUGen::parameters()` instead of the
//! typenum).  Neither touches the audio path.
//!
//! ## Inputs and audio-rate parameters
//! `GpuVoiceBank<F, I>` has `I::USIZE` audio inputs: `process_block` packs the `input` block it is handed
//! (`input.channel_as_slice(ch)`, ugen.rs:263-284) and gives it to the library with `knh_bank_set_input` in front of
//! `knh_bank_process_block`; the voices read channel `ch` through a `KNH_STAGE_INPUT` stage.  An audio-rate parameter edge
//! (`bank.link(k, modulator)`, graph_edit.rs:735-754 -> `set_ar_param_buffer(ctx, k, ptr)`, task.rs:113-120) is one more
//! such channel: a bank made `with_ar_slots(.., n)` has `n` of them behind its `I` inputs, `link`'s index `k` is the slot,
//! and the chain routes slot `k` with `KNH_STAGE_INPUT` on channel `I::USIZE + k` (e.g. `INPUT -> MUL_CONST(depth) ->
//! ADD_CONST(f0) -> SIN_WT` flagged `KNH_STAGE_FLAG_AR_FREQ`: the reference's `SinWt::new(f).ar_params()` with its `freq`
//! linked).  The buffer drives that parameter of EVERY voice, as the one node it is linked to.  The pointer is read during
//! `process_block` only (`ugen.rs:321-325`: valid until replaced or the node is dropped).
//!
//! ## Realtime
//! `process_block` waits for the GPU.  Since round 4 it launches nothing: the library keeps a resident kernel on the device
//! (state in registers, sine table in LDS), a call is one command word, and the mixed block arrives in pinned host memory
//! frame by frame, each with a tag the call waits for -- 20.5 us per call for a 16 384-voice bank of 512-frame blocks, 6.6 us
//! for one voice
//! (`knh_bank_resident_stats`, `KNH_RESIDENT=0` for a launch per call).  Use it under the non-realtime driver
//! (`AudioProcessor::run_without_inputs` in a loop, processor.rs:142-179).  It never allocates.
//!
//! NOT compiled in this repository's build image (no Rust toolchain); see bindings/rust/README.md.
#![allow(clippy::missing_safety_doc)]

use core::ffi::c_void;
use core::marker::PhantomData;

use knaster_core::{
    AudioCtx, Block, BlockRead, Float, Frame, ParameterHint, ParameterSmoothing, ParameterValue, Rate, Size, UGen, UGenFlags,
    numeric_array::NumericArray,
    rt_log,
    typenum::{U0, U2, Unsigned},
};

pub mod ffi;
use ffi::*;

/// Upper bound of parameters per stage (BufferReader has 6: buffer.rs:62-103).
pub const MAX_PARAMS: usize = 6;

/// One stage of a voice chain (`knh_stage_desc`); the table of kinds is in `include/knaster_hip.h`.
pub type Stage = knh_stage_desc;

pub fn stage(kind: u16) -> Stage {
    Stage { kind, flags: 0, delayed_changes_per_block: 0, ar_param: 0, input: 0, input2: 0 }
}
pub trait StageExt {
    /// `.precise_timing::<N>()` on the stage's node (wrappers_core.rs:106-111)
    fn precise_timing(self, max_changes_per_block: u16) -> Self;
    /// `.ar_params()` + `.link("freq", running_signal)` (SinWt only)
    fn ar_freq(self) -> Self;
    /// `.smooth_params()` (wrappers_core.rs:63-65)
    fn smooth_params(self) -> Self;
    /// `.ar_params()` with the float parameter `param` linked to the output of stage `driver` of the same voice
    /// (`node.link(param, signal)`, graph_edit.rs:735-754; audio_rate.rs:11-85): `knh_stage_desc.ar_param` / `.input2`
    fn ar_param(self, param: u16, driver_stage: u16) -> Self;
}
impl StageExt for Stage {
    fn precise_timing(mut self, n: u16) -> Self {
        self.delayed_changes_per_block = n;
        self
    }
    fn ar_freq(mut self) -> Self {
        self.flags |= KNH_STAGE_FLAG_AR_FREQ;
        self
    }
    fn smooth_params(mut self) -> Self {
        self.flags |= KNH_STAGE_FLAG_SMOOTH_PARAMS;
        self
    }
    fn ar_param(mut self, param: u16, driver_stage: u16) -> Self {
        self.ar_param = param + 1;
        self.input2 = driver_stage + 1;
        self
    }
}

#[derive(Debug)]
pub struct BankError(pub String);
impl core::fmt::Display for BankError {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        write!(f, "knaster_hip: {}", self.0)
    }
}
impl std::error::Error for BankError {}

fn last_error(h: *const knh_bank) -> BankError {
    BankError(unsafe { std::ffi::CStr::from_ptr(knh_last_error(h)) }.to_string_lossy().into_owned())
}

pub struct GpuVoiceBank<F: Float, I: Size = U0> {
    h: *mut knh_bank,
    n_stages: usize,
    n_voices: u32,
    /// audio-rate parameter slots behind the `I` inputs (input channels `I::USIZE ..` of the library's bank)
    n_ar: usize,
    /// what `set_ar_param_buffer` handed over, per slot (null: nothing linked, the slot reads zeros)
    ar_bufs: Vec<*const F>,
    /// `[I::USIZE + n_ar][block_size]`, channel-major: the block `knh_bank_set_input` is given; sized in `init`
    in_pack: Vec<F>,
    block_size: usize,
    /// `UGen::init` returns nothing (ugen.rs:242-246): a failed `knh_bank_init` (no gfx950 device, out of device memory, a
    /// chain that cannot be fused) is kept here; the node then renders silence and says so once per block on the RT logger.
    init_error: Option<BankError>,
    /// `param_apply_many`'s arrays (voices, stages, params, kinds, floats, integers), kept between calls: no allocation per batch
    batch: (Vec<u32>, Vec<u32>, Vec<u32>, Vec<u32>, Vec<f64>, Vec<i64>),
    _f: PhantomData<(F, I)>,
}
// knaster requires `Node: Send` (knaster_graph/src/node.rs:194).  The handle is single-caller: the graph moves
// it between threads (control thread at push, audio thread afterwards, control thread again for the drop,
// task.rs:105-130) but never shares it.
unsafe impl<F: Float, I: Size> Send for GpuVoiceBank<F, I> {}

impl<F: Float, I: Size> GpuVoiceBank<F, I> {
    /// `ctor[stage]` = row-major `[n_voices][n_args]` constructor arguments (`SinWt::new(freq)` → `[freq]`,
    /// `SvfFilter::new(ty, cutoff, q, gain)` → `[ty as f64, cutoff, q, gain]`, ...; table in knaster_hip.h).
    /// The randomness sources (`KNH_STAGE_WHITE_NOISE`, `_PINK_NOISE`, `_BROWN_NOISE`, `_RANDOM_LIN`) take as first argument
    /// the seed their reference constructor would have drawn: `knaster_core_dsp::noise::next_randomness_seed() as f64`,
    /// once per voice, in the order the voices would have been constructed.
    pub fn new(stages: &[Stage], n_voices: u32, ctor: &[Vec<f64>]) -> Result<Self, BankError> {
        Self::with_options(stages, n_voices, ctor, 0, 0)
    }
    /// A bank that accepts `ar_slots` audio-rate parameter buffers (`handle.link(k, source)` for `k < ar_slots`): slot `k`
    /// is input channel `I::USIZE + k` of the chain's `KNH_STAGE_INPUT` stages (module docs).
    pub fn with_ar_slots(stages: &[Stage], n_voices: u32, ctor: &[Vec<f64>], ar_slots: usize) -> Result<Self, BankError> {
        Self::with_options(stages, n_voices, ctor, 0, ar_slots)
    }
    /// The same bank with the host side of its sample-accurate parameter changes (the `WrPreciseTiming` queues of
    /// every voice and the per-block event lists built from them) spread over `host_threads` threads; worth it when
    /// every voice receives changes every block (`knh_bank_create_sharded` in knaster_hip.h).
    pub fn with_host_threads(stages: &[Stage], n_voices: u32, ctor: &[Vec<f64>], host_threads: u32) -> Result<Self, BankError> {
        Self::with_options(stages, n_voices, ctor, host_threads, 0)
    }
    pub fn with_options(stages: &[Stage], n_voices: u32, ctor: &[Vec<f64>], host_threads: u32, ar_slots: usize) -> Result<Self, BankError> {
        if I::USIZE + ar_slots > 16 {
            return Err(BankError("a bank node has at most 16 input channels (inputs + audio-rate parameter slots)".into()));
        }
        let desc = knh_bank_desc {
            abi_version: KNH_ABI_VERSION,
            n_voices,
            sample_type: if core::mem::size_of::<F>() == 8 { KNH_F64 } else { KNH_F32 },
            n_stages: stages.len() as u32,
            stages: stages.as_ptr(),
            out_channels: 2,
            mix_mode: KNH_MIX_TREE,
            device: -1,
            allow_fma: 0,
            in_channels: (I::USIZE + ar_slots) as u32,
        };
        let mut h = core::ptr::null_mut();
        if unsafe { knh_bank_create_sharded(&desc, host_threads, &mut h) } != KNH_OK {
            return Err(last_error(core::ptr::null()));
        }
        for (s, args) in ctor.iter().enumerate() {
            let n_args = (args.len() / n_voices as usize) as u32;
            if n_args > 0 && unsafe { knh_bank_set_ctor_args(h, s as u32, 0, n_voices, args.as_ptr(), n_args) } != KNH_OK {
                let e = last_error(h);
                unsafe { knh_bank_destroy(h) };
                return Err(e);
            }
        }
        Ok(Self {
            h,
            n_stages: stages.len(),
            n_voices,
            n_ar: ar_slots,
            ar_bufs: vec![core::ptr::null(); ar_slots],
            in_pack: Vec::new(),
            block_size: 0,
            init_error: None,
            batch: Default::default(),
            _f: PhantomData,
        })
    }

    /// `Buffer::from_vec(samples, sample_rate)` for the chain's `BufferReader` stage: one single-channel buffer shared
    /// by every voice of the bank.  Before the bank is pushed (init runs at push time, graph.rs:462-475).
    pub fn set_buffer(&mut self, stage: usize, samples: &[F], buffer_sample_rate: f64) -> Result<(), BankError> {
        let rc = unsafe { knh_bank_set_buffer(self.h, stage as u32, samples.as_ptr() as *const c_void, samples.len(), buffer_sample_rate) };
        if rc != KNH_OK { Err(last_error(self.h)) } else { Ok(()) }
    }

    /// Flat parameter index of (`voice`, `stage`, parameter name), e.g. `"cutoff_freq"`, `"t_restart"`, `"wr_mul"`.
    pub fn index(&self, voice: u32, stage: usize, name: &str) -> Result<usize, BankError> {
        if voice >= self.n_voices || stage >= self.n_stages {
            return Err(BankError("voice or stage out of range".into()));
        }
        let n = unsafe { knh_bank_stage_parameters(self.h, stage as u32) } as usize;
        for p in 0..n {
            let d = unsafe { knh_bank_stage_param_description(self.h, stage as u32, p as u32) };
            if !d.is_null() && unsafe { std::ffi::CStr::from_ptr(d) }.to_bytes() == name.as_bytes() {
                return Ok(flat_index(voice as usize, self.n_stages, stage, p));
            }
        }
        Err(BankError(format!("DescriptionNotFound({name})"))) // ParameterError::DescriptionNotFound, ugen.rs:365
    }
    /// The number of reference UGen nodes one voice of this chain stands for (the unit of UGen-samples/s).
    pub fn ugens_per_voice(stages: &[Stage]) -> i32 {
        unsafe { knh_chain_ugen_count(stages.as_ptr(), stages.len() as u32) }
    }
    /// One parameter of the voices `[voice_begin, voice_end)` in ONE call and without an array: a bank's note-on
    /// (`bank.param_apply_range(0, n, env_stage, t_restart, ParameterValue::Trigger)`).  `param` = the parameter's position in
    /// the stage's `#[param]` order, as `index()` resolves it from its name.
    pub fn param_apply_range(&mut self, voice_begin: u32, voice_end: u32, stage: usize, param: usize, value: ParameterValue) -> Result<(), BankError> {
        let (kind, f, i) = encode_value(value);
        let rc = unsafe { knh_bank_param_apply_range(self.h, voice_begin, voice_end, stage as u32, param as u32, kind, f, i) };
        if rc != KNH_OK { Err(last_error(self.h)) } else { Ok(()) }
    }
    /// One parameter of many voices in ONE call: what a host does with a block's worth of `SchedulingEvent`s for this node
    /// instead of a `param_apply` per event (16 384 single calls cost 170 us of host time, one batched call 11).  `index[k]` as
    /// `index()` gives it; all of them must name the same (stage, parameter).  Envelope triggers for neighbouring voices in
    /// rising order travel to a resident kernel as one 32-byte range event.
    pub fn param_apply_many(&mut self, indices: &[usize], value: ParameterValue) -> Result<(), BankError> {
        let (kind, f, i) = encode_value(value);
        let n = indices.len();
        let (n_stages, b) = (self.n_stages, &mut self.batch);
        b.0.clear();
        b.1.clear();
        b.2.clear();
        for &ix in indices {
            let (param, rest) = (ix % MAX_PARAMS, ix / MAX_PARAMS);
            b.0.push((rest / n_stages) as u32);
            b.1.push((rest % n_stages) as u32);
            b.2.push(param as u32);
        }
        b.3.clear();
        b.3.resize(n, kind);
        b.4.clear();
        b.4.resize(n, f);
        b.5.clear();
        b.5.resize(n, i);
        let rc = unsafe { knh_bank_param_apply_many(self.h, n, b.0.as_ptr(), b.1.as_ptr(), b.2.as_ptr(), b.3.as_ptr(), b.4.as_ptr(), b.5.as_ptr(), core::ptr::null()) };
        if rc != KNH_OK { Err(last_error(self.h)) } else { Ok(()) }
    }
    /// Offline rendering: `n_blocks` consecutive blocks in one launch (events of those blocks must already be
    /// scheduled through `knh_bank_param_apply_many_at`).  `out` = `[n_blocks][2][block_size]`.
    pub fn process_blocks(&mut self, n_blocks: u32, frame_clock: u64, out: &mut [F]) -> Result<u32, BankError> {
        let mut flags = 0u32;
        let rc = unsafe { knh_bank_process_blocks(self.h, n_blocks, frame_clock, out.as_mut_ptr() as *mut c_void, &mut flags) };
        if rc != KNH_OK { Err(last_error(self.h)) } else { Ok(flags) }
    }
    pub fn raw(&self) -> *mut knh_bank {
        self.h
    }
    /// What `UGen::init` could not report: `Some` if the device side of the bank was not set up (the node renders silence).
    pub fn init_error(&self) -> Option<&BankError> {
        self.init_error.as_ref()
    }
    #[inline]
    fn split(&self, index: usize) -> (u32, u32, u32) {
        let (param, rest) = (index % MAX_PARAMS, index / MAX_PARAMS);
        ((rest / self.n_stages) as u32, (rest % self.n_stages) as u32, param as u32)
    }
}

/// A `ParameterValue` as the C ABI takes it: (KNH_VALUE_*, the float, the integer).
#[inline]
fn encode_value(value: ParameterValue) -> (u32, f64, i64) {
    match value {
        ParameterValue::Float(v) => (KNH_VALUE_FLOAT, v as f64, 0),
        ParameterValue::Trigger => (KNH_VALUE_TRIGGER, 0.0, 0),
        ParameterValue::Integer(v) => (KNH_VALUE_INTEGER, 0.0, v.0 as i64),
        ParameterValue::Bool(b) => (KNH_VALUE_BOOL, 0.0, b as i64),
        // needs KNH_STAGE_FLAG_SMOOTH_PARAMS on the stage (WrSmoothParams, smooth_params.rs:12-311)
        ParameterValue::Smoothing(ParameterSmoothing::None, _) => (KNH_VALUE_SMOOTHING, 0.0, 0),
        ParameterValue::Smoothing(ParameterSmoothing::Linear(s), Rate::BlockRate) => (KNH_VALUE_SMOOTHING, s as f64, 1),
        ParameterValue::Smoothing(ParameterSmoothing::Linear(s), Rate::AudioRate) => (KNH_VALUE_SMOOTHING, s as f64, 2),
    }
}

#[inline]
pub const fn flat_index(voice: usize, n_stages: usize, stage: usize, param: usize) -> usize {
    (voice * n_stages + stage) * MAX_PARAMS + param
}

impl<F: Float, I: Size> Drop for GpuVoiceBank<F, I> {
    fn drop(&mut self) {
        unsafe { knh_bank_destroy(self.h) }
    }
}

impl<F: Float, I: Size> UGen for GpuVoiceBank<F, I> {
    type Sample = F;
    type Inputs = I;
    type Outputs = U2;
    type Parameters = U0; // addressed through the flat index (module docs)

    // graph.rs:462-475 calls init on the control thread at push time; allocating is allowed (ugen.rs:242-246)
    fn init(&mut self, sample_rate: u32, block_size: usize) {
        if unsafe { knh_bank_init(self.h, sample_rate, block_size) } != KNH_OK {
            let e = last_error(self.h);
            log::error!("{}", e);
            self.init_error = Some(e);
        }
        // the input block handed to the library once per process_block: allocated here, on the control thread
        self.block_size = block_size;
        self.in_pack = vec![F::ZERO; (I::USIZE + self.n_ar) * block_size];
    }

    fn process(&mut self, ctx: &mut AudioCtx, _flags: &mut UGenFlags, _input: Frame<F, I>) -> Frame<F, U2> {
        // A bank is only ever run block-wise (as GraphGen itself, graph_gen.rs:241-249).
        rt_log!(ctx.logger(); "knaster_hip: GpuVoiceBank::process called; only process_block is supported");
        Frame::default()
    }

    fn process_block<InBlock, OutBlock>(&mut self, ctx: &mut AudioCtx, flags: &mut UGenFlags, input: &InBlock, output: &mut OutBlock)
    where
        InBlock: BlockRead<Sample = F> + ?Sized,
        OutBlock: Block<Sample = F> + ?Sized,
    {
        if self.init_error.is_some() {
            for ch in 0..2 {
                for s in output.channel_as_slice_mut(ch).iter_mut() {
                    *s = F::ZERO;
                }
            }
            rt_log!(ctx.logger(); "knaster_hip: the bank failed to initialise (GpuVoiceBank::init_error): silence");
            return;
        }
        // Inputs and audio-rate parameter buffers: one channel-major block [I + n_ar][block_size], handed over in front of
        // the process call (knh_bank_set_input copies it into pinned memory; the upload rides in the launch's stream).
        // A partial block (ctx.block_start_offset() > 0, e.g. under a splitting wrapper) hands over the frames it covers
        // at their place in the block: `input` is already the partial view (knaster_primitives/src/block.rs:269-302).
        let n_in = I::USIZE + self.n_ar;
        if n_in > 0 {
            let (bs, off, ftp) = (self.block_size, ctx.block_start_offset(), ctx.frames_to_process());
            for ch in 0..I::USIZE {
                let src = input.channel_as_slice(ch);
                let n = src.len().min(ftp).min(bs - off); // `src` starts at the partial block's first frame (block.rs:285-287)
                self.in_pack[ch * bs + off..ch * bs + off + n].copy_from_slice(&src[..n]);
            }
            for k in 0..self.n_ar {
                let dst = &mut self.in_pack[(I::USIZE + k) * bs..(I::USIZE + k + 1) * bs];
                let p = self.ar_bufs[k];
                if p.is_null() {
                    dst.fill(F::ZERO);
                } else {
                    // ugen.rs:321-325: at least block_size contiguous samples until replaced or dropped
                    dst.copy_from_slice(unsafe { core::slice::from_raw_parts(p, bs) });
                }
            }
            if unsafe { knh_bank_set_input(self.h, 1, self.in_pack.as_ptr() as *const c_void) } != KNH_OK {
                rt_log!(ctx.logger(); "knaster_hip: knh_bank_set_input failed");
            }
        }
        // One pointer per output channel, each at the first frame THIS call writes: `output` may be the whole
        // RawContiguousBlock (knaster_graph/src/block.rs:19-78) or, under a splitting wrapper, a PartialBlockMut whose slices
        // already start at the partial block's offset (knaster_primitives/src/block.rs:307-339, precise_timing.rs:98-110).
        // ctx.block_start_offset() says where those frames lie in the bank's block; it must not be applied to the pointers
        // a second time, and the Block trait promises nothing about channel 1 following channel 0 in memory.
        let out: [*mut c_void; 2] = [
            output.channel_as_slice_mut(0).as_mut_ptr() as *mut c_void,
            output.channel_as_slice_mut(1).as_mut_ptr() as *mut c_void,
        ];
        let mut f = 0u32;
        let rc = unsafe {
            knh_bank_process_block_channels(self.h, ctx.frames_to_process(), ctx.block_start_offset(), ctx.frame_clock(), out.as_ptr(), &mut f)
        };
        if rc != KNH_OK {
            rt_log!(ctx.logger(); "knaster_hip: process_block failed, status ", rc as f64);
        }
        if f & KNH_FLAG_ALL_DONE != 0 {
            flags.mark_done(0); // every voice's envelope has stopped: the bank as a whole is done (done.rs:33-45)
        }
    }

    fn param_hints() -> NumericArray<ParameterHint, U0> {
        NumericArray::default()
    }

    fn param_apply(&mut self, ctx: &mut AudioCtx, index: usize, value: ParameterValue) {
        let (voice, stage, param) = self.split(index);
        let (kind, f, i) = encode_value(value);
        if unsafe { knh_bank_param_apply(self.h, voice, stage, param, kind, f, i) } != KNH_OK {
            rt_log!(ctx.logger(); "knaster_hip: param_apply rejected, index ", index as f64);
        }
    }

    // An audio-rate parameter edge (`handle.link(k, source)`, graph_edit.rs:735-754; resolved to a pointer into the source
    // node's output block at commit, graph.rs:1532-1562; handed over when the new schedule is taken, task.rs:113-120).
    // `index` is the bank's audio-rate slot: the samples are packed behind the `I` inputs in process_block (module docs).
    unsafe fn set_ar_param_buffer(&mut self, ctx: &mut AudioCtx, index: usize, buffer: *const F) {
        if index < self.n_ar {
            self.ar_bufs[index] = buffer;
        } else {
            rt_log!(ctx.logger(); "knaster_hip: audio-rate parameter buffer for a slot the bank was not made with (with_ar_slots); ignored");
        }
    }

    // GraphGen::apply_parameter_change (graph_gen.rs:269-305) calls this first when the event's Time gives
    // 0 < delay < block_size, then param_apply: the same pair the C ABI mirrors, so `param.set_at(v, Time::..)`
    // is sample accurate for stages declared with `.precise_timing(n)`.
    fn set_delay_within_block_for_param(&mut self, _ctx: &mut AudioCtx, index: usize, delay: u16) {
        let (voice, stage, param) = self.split(index);
        unsafe { knh_bank_set_delay_within_block_for_param(self.h, voice, stage, param, delay) };
    }
}
