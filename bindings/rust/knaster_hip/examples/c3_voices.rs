//! BASELINE.json config C3 through the real graph: 16 384 voices of SinWt.wr_mul(1/N) -> SvfFilter(Low) -> * EnvAsr
//! as ONE node, rendered offline with AudioProcessor::run_without_inputs (processor.rs:142-179).
//! Illustrative: written against the surveyed API, never compiled (see ../README.md).
use knaster_core::{typenum::{U0, U2}, PTrigger};
use knaster_graph::{handle::HandleTrait, processor::{AudioProcessor, AudioProcessorOptions}};
use knaster_hip::{ffi::*, stage, GpuVoiceBank, StageExt};

fn main() -> Result<(), Box<dyn std::error::Error>> {
    let (n, block_size, sr) = (16_384u32, 512usize, 48_000u32);
    let (mut graph, mut audio_processor, _log) = AudioProcessor::<f32>::new::<U0, U2>(AudioProcessorOptions {
        block_size,
        sample_rate: sr,
        ring_buffer_size: 1 << 16, // one scheduling event per voice per note
        ..Default::default()
    });

    let chain = [stage(KNH_STAGE_SIN_WT), stage(KNH_STAGE_WR_MUL), stage(KNH_STAGE_SVF), stage(KNH_STAGE_MUL_ENV_ASR).precise_timing(2)];
    // per-voice constructor arguments, [n_voices][n_args] row-major per stage
    let mut ctor = vec![Vec::new(); chain.len()];
    for v in 0..n as usize {
        let u = v as f64 / n as f64;
        ctor[0].push(55.0 * (6.0 * u).exp2()); // SinWt::new(freq)
        ctor[1].push(1.0 / n as f64); // .wr_mul(gain)
        ctor[2].extend([0.0 /* SvfFilterType::Low */, 200.0 + 7800.0 * u, 0.5 + 3.5 * u, 0.0]);
        ctor[3].extend([0.002 + 0.02 * u, 0.05 + 0.25 * u]); // EnvAsr::new(attack, release)
    }
    let bank = GpuVoiceBank::<f32>::new(&chain, n, &ctor)?;
    let restart: Vec<usize> = (0..n).map(|v| bank.index(v, 3, "t_restart").unwrap()).collect();
    let release: Vec<usize> = (0..n).map(|v| bank.index(v, 3, "t_release").unwrap()).collect();

    let handle = graph.push(bank); // Handle<GpuVoiceBank<f32>>, graph.rs:370
    graph.edit(|g| g.handle(&handle).expect("node exists").to_graph_out());
    for i in &restart {
        handle.set((*i, PTrigger))?;
    }
    let mut peak = 0.0f32;
    for block in 0..64 {
        if block == 32 {
            for i in &release {
                handle.set((*i, PTrigger))?;
            }
        }
        audio_processor.run_without_inputs();
        let out = audio_processor.output_block();
        for s in out.channel_as_slice(0) {
            peak = peak.max(s.abs());
        }
    }
    println!("rendered 64 blocks of {n} voices, peak {peak}");
    Ok(())
}
