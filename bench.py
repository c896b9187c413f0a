#!/usr/bin/env python3
"""bench.py -- UGen-samples/s of the fused voice-bank path on N MI355X (BASELINE.json metric).

Workload: BASELINE.json configs[2] ("C3"): 16384 voices per GPU, chain
SinWt.wr_mul(1/N) -> SvfFilter(Low) -> * EnvAsr, block_size 512, f32, 48 kHz, synthetic
per-voice parameters (xorshift32).  One *step* = one pass of the hot path over one
batch: BLOCKS_PER_STEP (64) consecutive 512-frame blocks of every voice on every rank, rendered in ONE
launch (knh_bank_process_blocks_device: voice state stays in registers across the blocks of a launch;
results are bit-identical to one launch per block, tests/test_gpu_properties.py) -- the note cycle of
SURVEY.md 8(d) (t_restart at block 0, t_release at block 32).  `value` counts every block of every step.  Voices shard across ranks (one process per GPU); each rank
folds its own voices into stereo blocks and each launch's stereo blocks are sum-reduced to rank 0
over RCCL in one call (the reduce is latency-bound at 4 KiB per block).

Launch: `python bench.py` (N=1) or
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
VALU_PEAK_OPS = 78.6e12        # 157.3 TFLOP/s FP32 vector counts an FMA as 2: 78.6e12 non-fused ops/s
OPS_PER_UGEN_SAMPLE = 6.0      # ~23 flop + 3 iop + 1 gather per voice-sample / 4 UGens
# The filter wavefront of the C3 pipeline kernel issues ten VALU instructions per sample (five of them packed); a wavefront
# alone on its SIMD needs 44 shader-clock cycles for those ten (tools/micro/svf_chain.hip, profiles/r01_micro_svf_chain.txt),
# 18.3 ns at the 2.4 GHz the chip holds under this load (tools/micro/clock_share.hip).  That is the floor of the kernel's
# time per sample: the serial filter recurrence of one 64-voice group cannot be spread over more wavefronts.
SVF_STEP_CYCLES = 44.0
SHADER_CLOCK_GHZ = 2.4
PIPE_TILE = 64                 # samples per pipeline step of the shipped C3 kernel (voice_pipe.hpp)
BLOCKS_PER_STEP = 64           # blocks per step = per launch = per RCCL reduce (one note cycle)
REDUCE_EVERY = BLOCKS_PER_STEP
PREWARM_MS = 150.0             # untimed launches before the warm-up steps: the shader clock needs a few ms of load to come up


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32, help="timed steps; one step = one 64-block launch per rank")
    ap.add_argument("--warmup", type=int, default=4, help="untimed steps before the timed ones")
    ap.add_argument("--voices-per-gpu", type=int, default=16384)
    ap.add_argument("--block-size", type=int, default=512)
    ap.add_argument("--allow-fma", action="store_true", help="non-bit-exact FMA kernels (reported as such)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-blocks", type=int, default=0, help="0 = auto (about 10-20 s)")
    return ap.parse_args()


def cpu_baseline(w, cores: int, blocks: int):
    """The oracle (unfused reference-shaped graph, `cores` independent sequential schedulers) on a
    bounded sample of the same workload.  Checker code timed as a baseline -- never the product."""
    from oracle import oracle_py

    t0 = time.perf_counter()
    secs, _ = oracle_py.baseline_run(w.stages, w.n_voices, w.sample_type, w.out_channels, w.ctor, 48000, w.block_size,
                                     1, blocks, cores, w.restart, (w.release[0], w.release[1], blocks // 2))
    wall = time.perf_counter() - t0
    return secs, wall


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist

    import knaster_amd
    from knaster_amd import _lib as L
    from knaster_amd import configs

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the voice-bank path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    # KNH_BENCH_REHEARSE=1: several ranks share the visible GPUs and talk over gloo, to rehearse the N>1 control
    # flow on a one-GPU box.  Never set for a measurement: the reported line then says "rehearsal".
    rehearse = os.environ.get("KNH_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    nv, bs = args.voices_per_gpu, args.block_size
    # every rank owns a contiguous range of the global voice list: parameters are drawn for the whole
    # list (one xorshift stream, voices in index order) and sliced
    w_all = configs.config("C3", n_voices=nv * world, block_size=bs)
    lo, hi = rank * nv, (rank + 1) * nv
    bank = knaster_amd.VoiceBank(w_all.stages, nv, w_all.sample_type, w_all.out_channels, L.MIX_TREE, local_rank, args.allow_fma)
    for s, a in w_all.ctor.items():
        bank.set_ctor_args(s, a[lo:hi])
    bank.init(configs.SAMPLE_RATE, bs)
    ugens = knaster_amd.chain_ugen_count(w_all.stages)
    voices = np.arange(nv, dtype=np.uint32)

    stream = torch.cuda.current_stream()
    # the mixed stereo blocks of a launch: [REDUCE_EVERY][channels][block_size], resident in HBM; two of them so
    # that the RCCL reduce of one launch overlaps the next launch's kernels
    rings = [torch.zeros((REDUCE_EVERY, w_all.out_channels, bs), dtype=torch.float32, device=dev) for _ in range(2)]
    ring = rings[0]
    pending = [None, None]
    CYCLE = 64  # the note cycle of SURVEY.md 8(d): t_restart at block 0, t_release at block 32 of every 64 blocks

    def schedule(first_step: int, n: int):
        """Queue the parameter events of steps [first_step, first_step + n) for the next launch."""
        for i in range(n):
            phase = (first_step + i) % CYCLE
            if phase == 0:
                bank.param_apply_many(voices, w_all.restart[0], w_all.restart[1], L.VALUE_TRIGGER, block_offset=i)
            elif phase == CYCLE // 2:
                bank.param_apply_many(voices, w_all.release[0], w_all.release[1], L.VALUE_TRIGGER, block_offset=i)

    launch_no = [0]

    def run_steps(first_step: int, n: int):
        """n steps; one step = one launch of BLOCKS_PER_STEP blocks + one RCCL reduce of its stereo blocks."""
        for i in range(n):
            half = launch_no[0] & 1
            launch_no[0] += 1
            if pending[half] is not None:  # the reduce that last read this buffer must be done before it is rewritten
                pending[half].wait()
                pending[half] = None
            schedule((first_step + i) * BLOCKS_PER_STEP, BLOCKS_PER_STEP)
            bank.process_blocks_device(BLOCKS_PER_STEP, rings[half].data_ptr(), stream.cuda_stream)
            if world > 1 and not rehearse:
                pending[half] = dist.reduce(rings[half], dst=0, op=dist.ReduceOp.SUM, async_op=True)
            elif world > 1:  # gloo has no device-tensor reduce
                pending[half] = dist.all_reduce(rings[half], op=dist.ReduceOp.SUM, async_op=True)

    def drain():
        for h in range(2):
            if pending[h] is not None:
                pending[h].wait()
                pending[h] = None

    def fence():
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: bring the shader clock up (a kernel of under a few milliseconds right after an idle period runs ~25 % slow)
    t_pre = time.perf_counter()
    n_pre = 0
    while (time.perf_counter() - t_pre) * 1e3 < PREWARM_MS:
        run_steps(0, 2)
        drain()
        torch.cuda.synchronize()
        n_pre += 2
    run_steps(0, args.warmup)
    fence()
    bank.timing_reset(True)
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = bank.timing_read()
    bank.timing_reset(False)
    blocks_per_launch = float(BLOCKS_PER_STEP)
    total_blocks = args.steps * BLOCKS_PER_STEP

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        k = torch.tensor([kernel_ms / max(launches, 1)], dtype=torch.float64, device=dev)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
        kernel_avg_ms = float(k.item())
    else:
        kernel_avg_ms = kernel_ms / max(launches, 1)

    # The same path when the boundary hands the blocks to the HOST (knh_bank_process_blocks: D2H copy of
    # every launch's stereo blocks over PCIe + stream sync).  Reported beside `value`, never as `value`.
    host_rate = None
    if world == 1:
        n_host = 4
        schedule(0, REDUCE_EVERY)
        bank.process_blocks(REDUCE_EVERY)
        t1 = time.perf_counter()
        for i in range(n_host):
            schedule(0, REDUCE_EVERY)
            bank.process_blocks(REDUCE_EVERY)
        host_rate = float(nv) * bs * ugens * REDUCE_EVERY * n_host / (time.perf_counter() - t1)

    sane = bool(torch.isfinite(rings[0]).all().item() and torch.isfinite(rings[1]).all().item())
    # HBM bytes per launch from the committed PMC passes (newest round first), scaled per block: the counters are
    # per-launch totals of a 64-block launch of the same bank
    traffic, traffic_src = None, None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                prof = json.load(f)
            wl = prof.get("workload", {})
            if (wl.get("voices"), wl.get("block_size"), wl.get("sample_type", "f32")) == (nv, bs, "f32"):
                traffic = prof["voice_pipe_kernel"]["hbm_bytes_per_launch"] / float(wl["blocks_per_launch"]) * blocks_per_launch
                traffic_src = os.path.relpath(path, ROOT)
                break
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            continue
    if rank == 0:
        total_voices = nv * world
        ugen_samples = float(total_voices) * bs * ugens * total_blocks
        value = ugen_samples / elapsed
        rd, wr = bank.algorithmic_bytes_per_voice_block()
        # 92 B per voice per block (state read once + mutable state written once per block)
        # x the voice-blocks one launch processes
        alg_bytes_per_launch = float(rd + wr) * nv * blocks_per_launch
        achieved_gbs = alg_bytes_per_launch / (kernel_avg_ms * 1e-3) / 1e9 if kernel_avg_ms > 0 else 0.0
        kernel_rate = float(nv) * bs * ugens * blocks_per_launch / (kernel_avg_ms * 1e-3) if kernel_avg_ms > 0 else 0.0
        line = {
            "metric": "UGen-samples/sec (voices x block_size x UGens / s)",
            "value": value,
            "unit": "UGen-samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not rehearse else "synthetic (REHEARSAL: ranks share GPUs, gloo; not a measurement)",
            "config": {
                "workload": "C3: SinWt.wr_mul(1/N) -> SvfFilter(Low) -> * EnvAsr, stereo mix",
                "voices_per_gpu": nv, "voices_total": total_voices, "block_size": bs, "sample_rate": 48000,
                "ugens_per_voice": ugens, "mix": "two-level left fold (deterministic)",
                "step": f"one launch = {BLOCKS_PER_STEP} consecutive blocks of every voice (one note cycle); "
                        f"{total_blocks} blocks timed", "blocks_per_step": BLOCKS_PER_STEP, "blocks_timed": total_blocks,
                "residency": "value is measured with voice state, events and the mixed stereo blocks resident in HBM; the "
                             "PCIe-inclusive rate of the host-pointer boundary (knh_bank_process_blocks) is host_output",
                "prewarm": f"{n_pre} untimed launches (>= {PREWARM_MS:.0f} ms) before the warm-up steps, to bring the clock up",
                "arithmetic": "fma" if args.allow_fma else "exact (bit-identical per voice to the CPU oracle)",
                "parallelism": f"voices sharded over {world} rank(s); {REDUCE_EVERY} blocks per launch; RCCL sum-reduce of the "
                               f"stereo blocks once per launch",
                "events": "t_restart on every voice at block 0 and t_release at block 32 of every 64-block cycle",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": f"{traffic_src} (rocprofv3 FETCH_SIZE + WRITE_SIZE, separate --pmc passes, per launch)" if traffic else None,
                "kernel": "voice_pipe_kernel<float,false,64,true,Group<SinWt,MulVal>,Group<Svf>,Group<MulAsr>>",
                "kernel_avg_ms": kernel_avg_ms, "launches": launches, "blocks_per_launch": blocks_per_launch,
                "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                "note": "fused kernel moves 92 B per voice per block; it is bound by the instruction issue of its busiest "
                        "wavefront, not by HBM (see valu and issue)",
            },
            "valu": {
                "achieved_ops_per_s": kernel_rate * OPS_PER_UGEN_SAMPLE, "peak_ops_per_s": VALU_PEAK_OPS,
                "frac": kernel_rate * OPS_PER_UGEN_SAMPLE / VALU_PEAK_OPS,
                "ops_per_ugen_sample": OPS_PER_UGEN_SAMPLE, "kernel_only_ugen_samples_per_s": kernel_rate,
            },
            # What actually bounds this kernel at 16 384 voices (one 64-voice group per CU, one wavefront per SIMD): the
            # filter wavefront's instruction stream.
            "issue": {
                "bound": "instruction issue of the busiest wavefront (SVF), one wavefront per SIMD",
                "filter_step_cycles_per_sample_alone": SVF_STEP_CYCLES,
                "floor_ns_per_sample": SVF_STEP_CYCLES / SHADER_CLOCK_GHZ,
                "kernel_ns_per_sample": kernel_avg_ms * 1e6 / (blocks_per_launch * bs) if kernel_avg_ms > 0 else None,
                "frac": (SVF_STEP_CYCLES / SHADER_CLOCK_GHZ) / (kernel_avg_ms * 1e6 / (blocks_per_launch * bs))
                if kernel_avg_ms > 0 and not args.allow_fma else None,
                "tile_samples": PIPE_TILE,
                "note": "floor = the ten instructions of one filter step issued by a wavefront alone on its SIMD (44 cycles, "
                        "micro-benchmark) at 2.4 GHz; the rest of the kernel's time per sample is that wavefront's LDS hand-over, "
                        "block/event bookkeeping and barrier once per 64-sample tile, and tiles in which the envelope "
                        "wavefront (which also folds the voices) is the slower one",
            },
            "output_finite": sane,
            "host_output": None if host_rate is None else {
                "value": host_rate, "unit": "UGen-samples/s",
                "note": "PCIe-inclusive: each 64-block launch's stereo blocks copied to host memory and synchronised",
            },
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 1
            cores = max(1, min(cores, 16))  # the GPU box's CPU share for one GPU is 16 cores
            blocks = args.cpu_baseline_blocks
            if not blocks:  # calibrate to roughly 10-15 s of CPU work
                secs8, _ = cpu_baseline(w_all, cores, 8)
                blocks = int(max(16, min(4096, 8 * 12.0 / max(secs8, 1e-3))))
            secs, wall = cpu_baseline(w_all, cores, blocks)
            line["cpu_baseline"] = {
                "value": float(total_voices) * bs * ugens * blocks / secs, "unit": "UGen-samples/s", "cores": cores,
                "kind": "port",
                "sample": f"{blocks} blocks of the same {total_voices}-voice C3 graph, unfused reference-shaped node graph, "
                          f"{cores} independent sequential schedulers (oracle, g++ -O3 -ffp-contract=off)",
                "seconds": secs,
            }
            b1 = max(4, blocks // 16)
            secs1, _ = cpu_baseline(w_all, 1, b1)
            line["cpu_baseline_single_thread"] = {
                "value": float(total_voices) * bs * ugens * b1 / secs1, "unit": "UGen-samples/s", "cores": 1,
                "kind": "port", "sample": f"{b1} blocks, one sequential scheduler (the reference is single-threaded)",
            }
        print(json.dumps(line), flush=True)
    bank.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
