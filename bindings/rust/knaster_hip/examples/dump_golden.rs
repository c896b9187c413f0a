//! Reference-side vector generator: renders the six cases of `tests/golden/make_golden.py` with the REAL crate -- `graph.edit`,
//! `push`, the operators, `link`, `param().set/trig/set_after`, `AudioProcessor::run_without_inputs` -- and writes the same
//! `voices` / `mix` arrays as raw little-endian files that `tests/test_golden.py::test_reference_dump_matches_golden` compares
//! with the committed `tests/golden/*.npz` (which come from this repository's C++ oracle).
//!
//! This is the one step that turns "parity unpinned" into a pin, and it needs `cargo`, which the build image of this
//! repository does not have: the file is source only, written against the surveyed API (knaster_graph/src/graph_edit.rs,
//! processor.rs:142-179, wrappers_core.rs:26-111) and has never been compiled.
//!
//!     python tests/golden/make_golden.py --inputs        # writes tests/golden/<case>.ctor.bin (the constructor arguments)
//!     cargo run --release --features graph --example dump_golden -- <repo>/tests/golden
//!     python -m pytest tests/test_golden.py -k reference  # compares tests/golden/reference/<case>.ref.bin with <case>.npz
//!
//! Input  `<case>.ctor.bin`: u32 magic 0x4B4E4831, u32 n_voices, u32 block_size, u32 blocks, u32 n_stages, then per stage
//!         u32 n_args and n_voices * n_args f64 (row-major per voice).
//! Output `reference/<case>.ref.bin`: the per-voice signals `[blocks][(2)][n_voices][block_size]` then the mix
//!         `[blocks][2][block_size]`, as f32 (f64 for c4), little endian.
use std::fs;
use std::io::Write;
use std::path::{Path, PathBuf};

use knaster_core::typenum::{U0, U2};
use knaster_core::{Float, PTrigger, Seconds};
use knaster_core_dsp::envelopes::{EnvAr, EnvAsr};
use knaster_core_dsp::osc::{SinNumeric, SinWt};
use knaster_core_dsp::pan::Pan2;
use knaster_core_dsp::svf::{SvfFilter, SvfFilterType};
use knaster_core_dsp::wrappers_core::UGenWrapperCoreExt;
use knaster_graph::graph::Graph;
use knaster_graph::graph_edit::Parameter;
use knaster_graph::processor::{AudioProcessor, AudioProcessorOptions};

const SR: u32 = 48_000;

struct Case {
    n_voices: usize,
    block_size: usize,
    blocks: usize,
    ctor: Vec<(usize, Vec<f64>)>, // per stage: (n_args, [n_voices][n_args])
}
impl Case {
    fn arg(&self, stage: usize, voice: usize, k: usize) -> f64 {
        let (n, a) = &self.ctor[stage];
        a[voice * n + k]
    }
}

fn read_case(path: &Path) -> Case {
    let b = fs::read(path).expect("run `python tests/golden/make_golden.py --inputs` first");
    let u32_at = |o: usize| u32::from_le_bytes(b[o..o + 4].try_into().unwrap()) as usize;
    assert_eq!(u32_at(0), 0x4B4E_4831);
    let (n_voices, block_size, blocks, n_stages) = (u32_at(4), u32_at(8), u32_at(12), u32_at(16));
    let mut o = 20;
    let mut ctor = Vec::new();
    for _ in 0..n_stages {
        let n_args = u32_at(o);
        o += 4;
        let mut a = Vec::with_capacity(n_voices * n_args);
        for _ in 0..n_voices * n_args {
            a.push(f64::from_le_bytes(b[o..o + 8].try_into().unwrap()));
            o += 8;
        }
        ctor.push((n_args, a));
    }
    Case { n_voices, block_size, blocks, ctor }
}

/// What one voice exposes to the event script.
#[derive(Default)]
struct VoiceParams {
    restart: Option<Parameter>,
    release: Option<Parameter>,
    cutoff: Option<Parameter>,
    freq: Option<Parameter>,        // C2 / M1 oscillator, C5 modulator
    pan: Option<Parameter>,
    phase_offset: Option<Parameter>, // C5 carrier
}

fn new_graph<F: Float>(block_size: usize) -> (Graph<F>, AudioProcessor<F>) {
    let (graph, processor, _log) = AudioProcessor::<F>::new::<U0, U2>(AudioProcessorOptions {
        block_size,
        sample_rate: SR,
        ring_buffer_size: 1 << 14,
        ..Default::default()
    });
    (graph, processor)
}

/// Pushes voice `v` of `case` into `graph` the way a user of the reference writes it.
fn push_voice<F: Float>(name: &str, c: &Case, v: usize, graph: &mut Graph<F>) -> VoiceParams {
    let f = |x: f64| F::new(x);
    let mut p = VoiceParams::default();
    graph.edit(|g| match name {
        "c1_readme" => {
            // README.md:34-51
            let sine = g.push(SinWt::new(f(c.arg(0, v, 0))));
            (sine * c.arg(1, v, 0)).out([0, 0]).to_graph_out();
        }
        "c2_sin_numeric" => {
            let s = g.push(SinNumeric::new(f(c.arg(0, v, 0))));
            (s * c.arg(1, v, 0)).out([0, 0]).to_graph_out();
            p.freq = Some(s.param("freq"));
        }
        "c3_chain_f32" | "c4_chain_f64" => {
            let s = g.push(SinWt::new(f(c.arg(0, v, 0))).wr_mul(f(c.arg(1, v, 0))));
            let ty = SvfFilterType::Low; // make_golden.py: svf type 0
            let flt = g.push(SvfFilter::new(ty, f(c.arg(2, v, 1)), f(c.arg(2, v, 2)), f(c.arg(2, v, 3))));
            let env = g.push(EnvAsr::new(f(c.arg(3, v, 0)), f(c.arg(3, v, 1))));
            ((s >> flt) * env).out([0, 0]).to_graph_out();
            p.restart = Some(env.param("t_restart"));
            p.release = Some(env.param("t_release"));
            p.cutoff = Some(flt.param("cutoff_freq"));
        }
        "m1_many_sines_pan2" => {
            // knaster/examples/many_sines.rs:51-63
            let env = g.push(EnvAr::new(f(c.arg(2, v, 0)), f(c.arg(2, v, 1))));
            let sine = g.push(SinWt::new(f(c.arg(0, v, 0))).wr_mul(f(c.arg(1, v, 0))));
            let pan = g.push(Pan2::new(c.arg(3, v, 0) as f32));
            ((env * sine) >> pan).to_graph_out();
            p.restart = Some(env.param("t_restart"));
            p.freq = Some(sine.param("freq"));
            p.pan = Some(pan.param("pan"));
        }
        "c5_fm_events" => {
            // modulator * index + f0 -> carrier.ar_params() "freq"; both wrapped for sample-accurate changes
            let m = g.push(SinWt::new(f(c.arg(0, v, 0))).precise_timing::<4>());
            let carrier = g.push(SinWt::new(f(c.arg(3, v, 0))).ar_params().precise_timing::<4>());
            carrier.link("freq", m * c.arg(1, v, 0) + c.arg(2, v, 0));
            (carrier * c.arg(4, v, 0)).out([0, 0]).to_graph_out();
            p.freq = Some(m.param("freq"));
            p.phase_offset = Some(carrier.param("phase_offset"));
        }
        other => panic!("unknown case {other}"),
    });
    p
}

fn put<F: Float>(out: &mut Vec<u8>, x: F) {
    if core::mem::size_of::<F>() == 8 {
        out.extend_from_slice(&x.to_f64().to_le_bytes());
    } else {
        out.extend_from_slice(&x.to_f32().to_le_bytes());
    }
}

fn render<F: Float>(name: &str, dir: &Path) {
    let c = read_case(&dir.join(format!("{name}.ctor.bin")));
    let pan = name == "m1_many_sines_pan2";
    // one graph per voice (its own signal), and one graph holding them all (the reference's additive mix, graph.rs:827-872)
    let mut singles: Vec<(Graph<F>, AudioProcessor<F>, Vec<VoiceParams>)> = Vec::new();
    for v in 0..c.n_voices {
        let (mut g, p) = new_graph::<F>(c.block_size);
        let vp = push_voice(name, &c, v, &mut g);
        singles.push((g, p, vec![vp]));
    }
    let (mut all_graph, mut all_proc) = new_graph::<F>(c.block_size);
    let mut all_params: Vec<VoiceParams> = (0..c.n_voices).map(|v| push_voice(name, &c, v, &mut all_graph)).collect();
    let freq0: Vec<f64> = (0..c.n_voices).map(|v| c.arg(0, v, 0)).collect();
    let mut voices_bytes = Vec::new();
    let mut mix_bytes = Vec::new();
    for block in 0..c.blocks {
        // a single-voice graph sees its own voice's events, the big graph every voice's
        for (v, (_g, proc_, vp)) in singles.iter_mut().enumerate() {
            script_voice(name, &c, block, &mut vp[0], v, &freq0);
            proc_.run_without_inputs();
        }
        for (v, p) in all_params.iter_mut().enumerate() {
            script_voice(name, &c, block, p, v, &freq0);
        }
        all_proc.run_without_inputs();
        let channels = if pan { 2 } else { 1 };
        for ch in 0..channels {
            for (_g, proc_, _vp) in singles.iter() {
                for s in proc_.output_block().channel_as_slice(ch) {
                    put(&mut voices_bytes, *s);
                }
            }
        }
        for ch in 0..2 {
            for s in all_proc.output_block().channel_as_slice(ch) {
                put(&mut mix_bytes, *s);
            }
        }
    }
    let out_dir = dir.join("reference");
    fs::create_dir_all(&out_dir).unwrap();
    let mut f = fs::File::create(out_dir.join(format!("{name}.ref.bin"))).unwrap();
    f.write_all(&voices_bytes).unwrap();
    f.write_all(&mix_bytes).unwrap();
    println!("{name}: {} voices x {} blocks written", c.n_voices, c.blocks);
}

/// tests/golden/make_golden.py `script()`: the parameter events of voice `only` in front of `block` (voices are independent:
/// the batches of the Python script, voice by voice).
fn script_voice(name: &str, c: &Case, block: usize, p: &mut VoiceParams, v: usize, freq0: &[f64]) {
    let n = c.n_voices;
    match name {
        "c3_chain_f32" | "c4_chain_f64" => {
            if block == 0 {
                p.restart.as_mut().unwrap().trig().unwrap();
            }
            if block == 2 && v % 2 == 0 {
                p.release.as_mut().unwrap().trig().unwrap();
            }
            if block == 3 {
                p.cutoff.as_mut().unwrap().set(300.0 + 50.0 * v as f64).unwrap();
            }
        }
        "m1_many_sines_pan2" => {
            let odd: Vec<usize> = (1..n).step_by(2).collect();
            if block == 0 || block == 3 {
                p.restart.as_mut().unwrap().trig().unwrap();
            }
            if block == 1 && v % 3 == 0 {
                p.freq.as_mut().unwrap().set(110.0 * (1 + v % 7) as f64).unwrap();
            }
            if block == 2 && v % 2 == 1 {
                let k = odd.iter().position(|x| *x == v).unwrap();
                let t = if odd.len() > 1 { -1.0 + 2.0 * k as f64 / (odd.len() - 1) as f64 } else { -1.0 };
                p.pan.as_mut().unwrap().set(t).unwrap();
            }
        }
        "c5_fm_events" => {
            if block % 2 == 0 {
                let k = 1.0 + 0.01 * ((block / 2) % 7) as f64;
                let delay = (17 * v) % c.block_size;
                let (param, value) = if v % 2 == 0 {
                    (p.freq.as_mut().unwrap(), freq0[v] * k)
                } else {
                    (p.phase_offset.as_mut().unwrap(), 1000.0 * ((block / 2) % 16) as f64)
                };
                if delay == 0 {
                    param.set(value).unwrap();
                } else {
                    param.set_after(value, Seconds::from_samples(delay as u64, SR as u64)).unwrap();
                }
            }
        }
        _ => {}
    }
}

fn main() {
    let dir = PathBuf::from(std::env::args().nth(1).expect("usage: dump_golden <repo>/tests/golden"));
    render::<f32>("c1_readme", &dir);
    render::<f32>("c2_sin_numeric", &dir);
    render::<f32>("c3_chain_f32", &dir);
    render::<f64>("c4_chain_f64", &dir);
    render::<f32>("c5_fm_events", &dir);
    render::<f32>("m1_many_sines_pan2", &dir);
    // PTrigger is what `.trig()` sends (graph_edit.rs:1838-1848); named here so that the import documents it
    let _ = PTrigger;
}
