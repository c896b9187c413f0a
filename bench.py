#!/usr/bin/env python3
"""bench.py -- UGen-samples/s of the fused voice-bank path on N MI355X (BASELINE.json metric).

Workload (`value`): BASELINE.json configs[2] ("C3"): 16384 voices per GPU, chain
SinWt.wr_mul(1/N) -> SvfFilter(Low) -> * EnvAsr, block_size 512, f32, 48 kHz, synthetic per-voice parameters
(xorshift32); weak scaling (voices per GPU fixed).  One *step* = one pass of the hot path over one
batch: LAUNCHES_PER_STEP (4) launches of BLOCKS_PER_LAUNCH (64) consecutive 512-frame blocks of every voice on every rank
(knh_bank_process_blocks_device: voice state stays in registers across the blocks of a launch; results are
bit-identical to one launch per block, tests/test_gpu_properties.py) -- four note cycles of SURVEY.md 8(d)
(t_restart at block 0, t_release at block 32 of every 64).  `value` counts every block of every step.  (Four launches to
a step so that the driver's 20 timed steps last ~70 ms, not 17.)

Beside `value` the line carries: `host_output` -- the same path when the blocks are handed to the HOST, including
`per_block_value`, the rate of the call the reference actually makes (UGen::process_block once per block through
knh_bank_process_block, Task::run in knaster_graph/src/task.rs:25-31), for C3 and C1; and `configs` -- short legs of the
other BASELINE.json configurations (C1, C2, C5; C4 is `c4_strong`), each kernel-only and wall.

Several GPUs: one process per GPU; each rank creates its share of the bank with knh_bank_create_rank (contiguous
voice ranges, global voice indices) and the LIBRARY sums each launch's stereo blocks to rank 0 with RCCL's ncclReduce
on a stream of its own, overlapping the next launch (knaster_amd/csrc/comm.hip, rank_bank.hpp).  torch.distributed is
used for nothing on the data path: it hands the communicator id to the ranks, and provides the barrier and the
max-over-ranks of the timing.

The same line also carries BASELINE.json configs[3] ("C4": 65536 voices in all, f64, STRONG scaling over the ranks)
as `c4_strong` (skip with --no-c4; `--config C4` makes it the headline instead).

Launch: `python bench.py` (N=1) or
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import glob
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
# The filter wavefront of the C3 pipeline kernel issues seven VALU instructions per sample of a low-pass filter (four of
# them packed: Svf::tick_tile_low; the general step of the other eight filter types has ten); a wavefront alone
# on its SIMD needs 29 shader-clock cycles for them (tools/micro/svf_low_variants.hip, profiles/r03_micro_svf_low_variants.txt:
# ~4.15 per issue slot), 12.2 ns at the 2.4 GHz the chip holds under this load (tools/micro/clock_share.hip).  That is the
# floor of the kernel's time per sample: the serial filter recurrence of one 64-voice group cannot be spread over more
# wavefronts.
SVF_STEP_CYCLES = 29.25
SHADER_CLOCK_GHZ = 2.4
PIPE_TILE = 64                 # samples per pipeline step of the shipped C3 kernel (voice_pipe.hpp)
BLOCKS_PER_LAUNCH = 64         # blocks per launch = per RCCL reduce (one note cycle)
LAUNCHES_PER_STEP = 4          # a step = four such launches: the driver's 20 steps then time ~70 ms, not 17
BLOCKS_PER_STEP = BLOCKS_PER_LAUNCH * LAUNCHES_PER_STEP
PREWARM_MS = 150.0             # untimed launches before the warm-up steps: the shader clock needs a few ms of load to come up


def c4_issue(voices_per_gpu, block_size, kernel_us_per_block):
    """c4_strong.issue: the issue floor of a rank's share of C4 from profiles/r04_c4_floors.json (tools/c4_floors.py)."""
    out = {"regime": "pipeline: one 64-voice group per CU" if voices_per_gpu <= 16384 else "one whole-chain wavefront per SIMD",
           "kernel_us_per_block": kernel_us_per_block, "floor_cycles_per_sample": None, "floor_us_per_block": None, "frac_of_floor": None,
           "note": "strong scaling of this bank: per-GPU time per block is flat from 65 536 down to 16 385 voices per GPU (fewer "
                   "SIMDs busy, not faster ones), drops once a GPU's share fits the pipeline (16 384 voices or fewer: a 64-voice "
                   "group per CU) and is flat again below that"}
    try:
        with open(os.path.join(ROOT, "profiles", "r04_c4_floors.json")) as f:
            fl = json.load(f)
    except (OSError, ValueError):
        out["source"] = "profiles/r04_c4_floors.json missing: no floor quoted"
        return out
    reg = fl["pipeline"] if voices_per_gpu <= 16384 else fl["wide"]
    cyc = reg["floor_cycles_per_sample"]
    if voices_per_gpu > 65536:  # more than one wavefront per SIMD: their instructions share the SIMD's issue
        cyc *= voices_per_gpu / 65536.0
    out["floor_cycles_per_sample"] = cyc
    out["floor_us_per_block"] = cyc * block_size / (fl["shader_clock_ghz"] * 1e3)
    out["frac_of_floor"] = out["floor_us_per_block"] / kernel_us_per_block if kernel_us_per_block > 0 else None
    out["source"] = "profiles/r04_c4_floors.json <- " + reg["source"]
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32, help="timed steps; one step = four 64-block launches per rank")
    ap.add_argument("--warmup", type=int, default=4, help="untimed steps before the timed ones")
    ap.add_argument("--config", choices=["C3", "C4"], default="C3", help="headline workload: C3 weak scaling (default) or C4 strong scaling")
    ap.add_argument("--voices-per-gpu", type=int, default=16384, help="C3: voices per GPU (weak scaling)")
    ap.add_argument("--c4-voices", type=int, default=65536, help="C4: voices in all (strong scaling)")
    ap.add_argument("--block-size", type=int, default=512)
    ap.add_argument("--allow-fma", action="store_true", help="non-bit-exact FMA kernels (reported as such)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c4", action="store_true", help="skip the secondary C4 strong-scaling measurement")
    ap.add_argument("--no-configs", action="store_true", help="skip the short C1 / C2 / C5 legs and the per-block boundary legs")
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


class Env:
    """The process's place in the job: rank, device, and the little torch.distributed is used for."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            if self.rank == 0:
                print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}; launch with torch.distributed.run", file=sys.stderr)
            sys.exit(2)
        if not torch.cuda.is_available():
            print("bench.py: no GPU visible; the voice-bank path has no CPU fallback", file=sys.stderr)
            sys.exit(3)
        # KNH_BENCH_REHEARSE=1: several ranks share the visible GPUs (RCCL refuses two ranks on one GPU, so the sum goes
        # through a host-side gloo reduce handed to the library as its reduce function): rehearses the N > 1 control flow
        # on a one-GPU box (tests/test_gpu_multi.py).  Never set for a measurement: the line then says "rehearsal".
        self.rehearse = os.environ.get("KNH_BENCH_REHEARSE") == "1"
        if self.rehearse:
            self.local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.rehearse:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=self.dev)
        self.stream = torch.cuda.current_stream()

    def comm_id(self):
        """rank 0 makes the RCCL id (through the library); the host hands it round."""
        import knaster_amd

        if self.world == 1 or self.rehearse:
            return None
        box = [knaster_amd.comm_unique_id() if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def gloo_reduce_fn(self):
        """knh_reduce_fn for the rehearsal: device -> host, gloo reduce, host -> device on the root."""
        import ctypes as C

        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        torch, dist, rank = self.torch, self.dist, self.rank

        def fn(_user, buf, count, sample_type, root, stream):
            host = np.empty(count, dtype=np.float64 if sample_type == 1 else np.float32)
            if hip.hipStreamSynchronize(stream) != 0 or hip.hipMemcpy(host.ctypes.data, buf, host.nbytes, 2) != 0:
                return 4
            t = torch.from_numpy(host)
            dist.reduce(t, dst=root, op=dist.ReduceOp.SUM)
            if rank == root and hip.hipMemcpy(buf, host.ctypes.data, host.nbytes, 1) != 0:
                return 4
            return 0
        return fn

    def torch_reduce_fn(self, rings):
        """knh_reduce_fn fallback: the buffer the library hands over is one of the bench's own torch tensors."""
        dist = self.dist

        def fn(_user, buf, count, sample_type, root, stream):
            for r in rings:
                if r.data_ptr() == buf and r.numel() == count:
                    dist.reduce(r, dst=root, op=dist.ReduceOp.SUM)  # enqueued on the current stream = the bank's stream
                    return 0
            return 1
        return fn

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def gather_list(self, values):
        """-> [values of rank 0, values of rank 1, ...] on every rank"""
        if self.world == 1:
            return [list(values)]
        box = [None] * self.world
        self.dist.all_gather_object(box, list(values))
        return box

    def max_over_ranks(self, x: float) -> float:
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cpu" if self.rehearse else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def measure(env: Env, args, name: str, total_voices: int, bs: int, steps: int, warmup: int):
    """One workload on the job's ranks: returns a dict of raw measurements (every rank), or exits on failure."""
    import knaster_amd
    from knaster_amd import _lib as L
    from knaster_amd import configs

    torch = env.torch
    w_all = configs.config(name, n_voices=total_voices, block_size=bs)
    lo, cnt = knaster_amd.shard_voice_range(total_voices, env.rank, env.world)
    tdtype = torch.float64 if w_all.sample_type == L.F64 else torch.float32
    # the mixed stereo blocks of a launch: [BLOCKS_PER_LAUNCH][channels][block_size], resident in HBM; two of them, used
    # alternately, so that the library's reduce of one launch overlaps the next launch's kernels
    rings = [torch.zeros((BLOCKS_PER_LAUNCH, w_all.out_channels, bs), dtype=tdtype, device=env.dev) for _ in range(2)]

    def make_bank(collective: str):
        kwargs = dict(rank=env.rank, world=env.world)
        if env.world > 1 and env.rehearse:
            kwargs["reduce_fn"] = env.gloo_reduce_fn()
        elif env.world > 1 and collective == "torch":
            kwargs["reduce_fn"] = env.torch_reduce_fn(rings)
        else:
            kwargs["comm_id"] = env.comm_id()
        b = knaster_amd.VoiceBank(w_all.stages, total_voices, w_all.sample_type, w_all.out_channels, L.MIX_TREE, env.local_rank,
                                  args.allow_fma, **kwargs)
        for s, a in w_all.ctor.items():
            b.set_ctor_args(s, a)  # every rank hands over the whole list; the library keeps its range
        b.init(configs.SAMPLE_RATE, bs)
        return b

    # The library's own RCCL reduce is the path; should its communicator fail to come up in some environment, every rank
    # falls back (together) to torch.distributed's reduce handed to the library as its reduce function, and the line says so.
    collective, why = "library RCCL (ncclReduce on the communicator's own stream)", None
    bank = None
    try:
        bank = make_bank("native")
        ok = 1.0
    except L.KnasterHipError as e:
        ok, why = 0.0, str(e)
    if env.world > 1 and not env.rehearse and -env.max_over_ranks(-ok) < 1.0:  # min over ranks
        if bank is not None:
            bank.close()
        bank = make_bank("torch")
        collective = f"torch.distributed reduce as the library's reduce function (FALLBACK: the library's RCCL communicator failed: {why})"
    elif bank is None:
        raise RuntimeError(why)
    if env.rehearse and env.world > 1:
        collective = "host-side gloo reduce as the library's reduce function (rehearsal)"
    ugens = knaster_amd.chain_ugen_count(w_all.stages)
    mine = np.arange(lo, lo + cnt, dtype=np.uint32)  # global indices of this rank's voices
    restart = bank.prepare_many(mine, w_all.restart[0], w_all.restart[1], L.VALUE_TRIGGER)
    release = bank.prepare_many(mine, w_all.release[0], w_all.release[1], L.VALUE_TRIGGER)
    launch_no = [0]

    def schedule():
        """The parameter events of one launch (a 64-block note cycle): t_restart at block 0, t_release at block 32."""
        if cnt:
            bank.param_apply_prepared(restart, 0)
            bank.param_apply_prepared(release, BLOCKS_PER_LAUNCH // 2)

    def run_launches(n: int):
        for _ in range(n):
            half = launch_no[0] & 1
            launch_no[0] += 1
            schedule()
            bank.process_blocks_device(BLOCKS_PER_LAUNCH, rings[half].data_ptr(), env.stream.cuda_stream)

    def run_steps(n: int):
        run_launches(n * LAUNCHES_PER_STEP)

    def fence():
        bank.synchronize()  # this rank's kernels and reduces
        env.barrier()

    # untimed: bring the shader clock up (a kernel of under a few milliseconds right after an idle period runs ~25 % slow)
    # (every rank runs the same number of launches -- each launch holds a collective: the ranks agree after every pair)
    t_pre, n_pre = time.perf_counter(), 0
    while True:
        run_launches(2)
        bank.synchronize()
        n_pre += 2
        if env.max_over_ranks((time.perf_counter() - t_pre) * 1e3) >= PREWARM_MS:
            break
    run_steps(warmup)
    fence()
    bank.timing_reset(True)
    t0 = time.perf_counter()
    run_steps(steps)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = bank.timing_read()
    reduce_ms, reduces = bank.collective_timing_read()
    bank.timing_reset(False)
    elapsed = env.max_over_ranks(elapsed)
    kernel_avg_ms = env.max_over_ranks(kernel_ms / max(launches, 1))
    # every rank's own two numbers, so that a multi-GPU run explains itself: voice-kernel time and reduce time per launch
    per_rank = env.gather_list([kernel_ms / max(launches, 1), reduce_ms / max(reduces, 1) if reduces else 0.0, cnt])

    # The same path when the boundary hands the blocks to the HOST (knh_bank_process_blocks: D2H copy of
    # every launch's stereo blocks over PCIe + stream sync).  Reported beside `value`, never as `value`.
    host_rate = None
    if env.world == 1:
        n_host = 32
        schedule()
        bank.process_blocks(BLOCKS_PER_LAUNCH)
        schedule()
        bank.process_blocks_begin(BLOCKS_PER_LAUNCH)  # untimed: the pipelined path's buffers and stream are made on first use
        bank.process_blocks_end()
        t1 = time.perf_counter()
        for _ in range(n_host):  # launch by launch: each call returns with its blocks in host memory
            schedule()
            bank.process_blocks(BLOCKS_PER_LAUNCH)
        host_rate_blocking = float(total_voices) * bs * ugens * BLOCKS_PER_LAUNCH * n_host / (time.perf_counter() - t1)
        t1 = time.perf_counter()
        schedule()
        bank.process_blocks_begin(BLOCKS_PER_LAUNCH)
        for i in range(n_host):  # two launches in flight: launch i + 1 is enqueued before launch i's blocks are fetched
            if i + 1 < n_host:
                schedule()
                bank.process_blocks_begin(BLOCKS_PER_LAUNCH)
            host_blocks = bank.process_blocks_end()
        host_rate = float(total_voices) * bs * ugens * BLOCKS_PER_LAUNCH * n_host / (time.perf_counter() - t1)
        host_sane = bool(np.isfinite(host_blocks).all())
    sane = bool(torch.isfinite(rings[0]).all().item() and torch.isfinite(rings[1]).all().item())
    peak = float(max(rings[0].abs().max().item(), rings[1].abs().max().item()))
    rd, wr = bank.algorithmic_bytes_per_voice_block()
    out = dict(workload=w_all, ugens=ugens, elapsed=elapsed, kernel_avg_ms=kernel_avg_ms, launches=launches, host_rate=host_rate,
               host_rate_blocking=host_rate_blocking if env.world == 1 else None,
               sane=sane, peak=peak, bytes_per_voice_block=rd + wr, n_pre=n_pre, voices_rank0=cnt if env.rank == 0 else None,
               ranks_seen=bank.ranks(), total_voices=total_voices, steps=steps, collective=collective, per_rank=per_rank)
    bank.close()
    return out


def per_block_boundary(name: str, blocks: int = 2048):
    """The call the reference makes: UGen::process_block once per block (Task::run, knaster_graph/src/task.rs:25-31, under
    AudioProcessor::run_without_inputs, processor.rs:142-179) = one knh_bank_process_block[_channels] per block, the mixed block
    in HOST memory when it returns; the block's parameter events in front of it (graph_gen.rs:110-166), a block's triggers as
    one batched call.  Served by a resident kernel since round 4 (knh_bank_resident_stats).
    Two drivers: `twin` = tests/cpp/bin/shim_twin_test --bench, the C++ twin of the Rust shim making the shim's exact calls
    (what the round's targets are quoted on); `python` = this process through ctypes, everything preallocated.  Per-call wall
    time: median, mean and 99th percentile.  Returns a dict."""
    import ctypes as C
    import subprocess

    import knaster_amd
    from knaster_amd import _lib as L
    from knaster_amd import configs

    w = configs.config(name)
    bs = w.block_size
    ugens = knaster_amd.chain_ugen_count(w.stages)
    res = {"config": name, "voices": w.n_voices, "block_size": bs, "blocks": blocks, "unit": "UGen-samples/s"}
    twin = os.path.join(ROOT, "tests", "cpp", "bin", "shim_twin_test")
    if os.path.exists(twin):
        for label, extra_env in (("twin", {}), ("twin_launch_per_call", {"KNH_RESIDENT": "0"})):
            try:
                p = subprocess.run([twin, "--bench", name, str(blocks), "batched"], capture_output=True, text=True, timeout=120,
                                   env=dict(os.environ, **extra_env))
                d = json.loads(p.stdout.strip().splitlines()[-1])
                res[label] = {k: d[k] for k in ("us_per_block_p50", "us_per_block_mean", "us_per_block_p99", "us_per_block_min", "ugen_samples_per_s",
                                                "resident_calls", "device_us_after_the_voice_kernel_saw_the_command", "driver") if k in d}
            except Exception as e:  # noqa: BLE001 -- a bench must not die of its side leg
                res[label] = {"error": repr(e)[:200]}
    bank = knaster_amd.VoiceBank(w.stages, w.n_voices, w.sample_type, w.out_channels, L.MIX_TREE)
    for s, a in w.ctor.items():
        bank.set_ctor_args(s, a)
    bank.init(configs.SAMPLE_RATE, bs)
    v = np.arange(w.n_voices, dtype=np.uint32)
    restart = bank.prepare_many(v, w.restart[0], w.restart[1], L.VALUE_TRIGGER) if w.restart else None
    release = bank.prepare_many(v, w.release[0], w.release[1], L.VALUE_TRIGGER) if w.release else None
    out = np.zeros((w.out_channels, bs), dtype=np.float32)
    flags = C.c_uint32(0)
    fn, h, outp, fp = bank._lib.knh_bank_process_block, bank._h, out.ctypes.data_as(C.c_void_p), C.byref(flags)
    us = np.zeros(blocks)

    def run(n, clock0, timed):
        for blk in range(n):
            t0 = time.perf_counter()
            if restart is not None and blk % 64 == 0:
                bank.param_apply_prepared(restart, 0)
            if release is not None and blk % 64 == 32:
                bank.param_apply_prepared(release, 0)
            if fn(h, bs, 0, clock0 + blk * bs, outp, fp) != 0:
                raise RuntimeError("knh_bank_process_block failed")
            if timed:
                us[blk] = (time.perf_counter() - t0) * 1e6

    run(128, 0, False)  # warm-up (clock, first-use allocations)
    t0 = time.perf_counter()
    run(blocks, 128 * bs, True)
    dt = time.perf_counter() - t0
    calls, launches = bank.resident_stats()
    finite = bool(np.isfinite(out).all())
    bank.close()
    res["python"] = {"us_per_block_p50": float(np.median(us)), "us_per_block_mean": float(us.mean()), "us_per_block_p99": float(np.percentile(us, 99)),
                     "ugen_samples_per_s": float(w.n_voices) * bs * ugens * blocks / dt, "resident_calls": calls, "resident_launches": launches,
                     "output_finite": finite}
    best = res["twin"] if "twin" in res and "us_per_block_mean" in res.get("twin", {}) else res["python"]
    res["per_block_value"] = float(w.n_voices) * bs * ugens / (best["us_per_block_mean"] * 1e-6)
    res["us_per_call"] = best["us_per_block_mean"]
    res["us_per_call_p50"] = best["us_per_block_p50"]
    res["us_per_call_p99"] = best["us_per_block_p99"]
    res["driver_of_per_block_value"] = "twin" if best is res.get("twin") else "python"
    return res


def config_leg(name: str, launches: int = 16, blocks: int = 32, n_voices: int | None = None):
    """A short leg of another BASELINE.json configuration at its full size: `launches` launches of `blocks` blocks, outputs
    left in HBM; wall and kernel-only.  UGens per voice as SURVEY.md 8(d) counts them."""
    import knaster_amd
    from knaster_amd import _lib as L
    from knaster_amd import configs

    w = configs.config(name, n_voices=n_voices)
    b = knaster_amd.VoiceBank(w.stages, w.n_voices, w.sample_type, w.out_channels, L.MIX_TREE)
    for s, a in w.ctor.items():
        b.set_ctor_args(s, a)
    b.init(configs.SAMPLE_RATE, w.block_size)
    v = np.arange(w.n_voices, dtype=np.uint32)
    if w.restart:
        b.param_apply_many(v, w.restart[0], w.restart[1], L.VALUE_TRIGGER)
    if w.delay_times is not None:  # D3: every voice its own delay
        b.param_apply_many(v, 3, 0, L.VALUE_FLOAT, w.delay_times)
    # the application's own work of deciding what changes when is not the engine's: C5's event arrays are made up front
    c5 = {}
    if name == "C5":
        for blk in range(blocks * (launches + 3)):
            e = configs.c5_events(w, blk)
            c5[blk] = None if e is None else b.prepare_many(e[0], e[1], e[2], e[3], e[4], None, e[5])
    step = [0]

    def events(k):
        for i in range(k):
            e = c5.get(step[0] + i)
            if e is not None:
                b.param_apply_prepared(e, block_offset=i)
        step[0] += k
    for _ in range(3):  # untimed: first-use allocations (both of the alternating record / list buffers), the clock
        events(blocks)
        b.process_blocks_device(blocks)
    b.synchronize()
    b.timing_reset(True)
    t0 = time.perf_counter()
    for _ in range(launches):
        events(blocks)
        b.process_blocks_device(blocks)
    b.synchronize()
    dt = time.perf_counter() - t0
    kms, n = b.timing_read()
    ugens = {"C1": 3, "C2": 2, "C3": 4, "C4": 4, "C5": 3, "D3": 5}[name]
    work = float(w.n_voices) * w.block_size * ugens * blocks * launches
    rd, wr = b.algorithmic_bytes_per_voice_block()
    if w.delay_times is not None:  # the ring: one sample read and one written per frame
        rd += (8 if w.sample_type else 4) * w.block_size
        wr += (8 if w.sample_type else 4) * w.block_size
    b.close()
    return {"config": name, "workload": w.description, "voices": w.n_voices, "block_size": w.block_size, "dtype": "f64" if w.sample_type else "f32",
            "ugens_per_voice": ugens, "blocks_per_launch": blocks, "launches": launches, "value": work / dt, "unit": "UGen-samples/s",
            "kernel_only_value": work / (kms * 1e-3) if kms > 0 else None, "us_per_block_kernel": kms * 1e3 / (n * blocks) if n else None,
            "algorithmic_bytes_per_voice_block": rd + wr,
            "roofline_achieved_gbs": (rd + wr) * w.n_voices * blocks * n / (kms * 1e-3) / 1e9 if kms > 0 else None}


def traffic_from_profiles(nv: int, bs: int, sample_type: str, kernel_key: str = "voice_pipe_kernel"):
    """HBM bytes per 64-block launch from the committed PMC passes (newest round first), scaled per block: the counters
    are per-launch totals of a launch of the same bank (rocprofv3 FETCH_SIZE + WRITE_SIZE, separate --pmc passes)."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic*.json")), reverse=True):
        try:
            with open(path) as f:
                prof = json.load(f)
            wl = prof.get("workload", {})
            if (wl.get("voices"), wl.get("block_size"), wl.get("sample_type", "f32")) == (nv, bs, sample_type):
                return prof[kernel_key]["hbm_bytes_per_launch"] / float(wl["blocks_per_launch"]) * BLOCKS_PER_LAUNCH, os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            continue
    return None, None


def main():
    args = parse()
    env = Env(args)
    from knaster_amd import _lib as L

    bs, world = args.block_size, env.world
    headline = args.config
    if headline == "C3":
        m = measure(env, args, "C3", args.voices_per_gpu * world, bs, args.steps, args.warmup)
        secondary = None if args.no_c4 else measure(env, args, "C4", args.c4_voices, bs, max(4, args.steps // 2), max(1, args.warmup // 2))
    else:
        m = measure(env, args, "C4", args.c4_voices, bs, args.steps, args.warmup)
        secondary = None
    if env.rank == 0:
        w_all, ugens, total_voices = m["workload"], m["ugens"], m["total_voices"]
        f64 = w_all.sample_type == L.F64
        nv_rank = m["voices_rank0"]
        total_blocks = m["steps"] * BLOCKS_PER_STEP
        value = float(total_voices) * bs * ugens * total_blocks / m["elapsed"]
        kernel_avg_ms = m["kernel_avg_ms"]
        # 92 B per voice per block for C3 (state read once + mutable state written once per block; f64: 184 B)
        # x the voice-blocks one launch of one rank processes
        alg_bytes_per_launch = float(m["bytes_per_voice_block"]) * nv_rank * BLOCKS_PER_LAUNCH
        achieved_gbs = alg_bytes_per_launch / (kernel_avg_ms * 1e-3) / 1e9 if kernel_avg_ms > 0 else 0.0
        kernel_rate = float(nv_rank) * bs * ugens * BLOCKS_PER_LAUNCH / (kernel_avg_ms * 1e-3) if kernel_avg_ms > 0 else 0.0
        traffic, traffic_src = traffic_from_profiles(nv_rank, bs, "f64" if f64 else "f32")
        ns_per_sample = kernel_avg_ms * 1e6 / (BLOCKS_PER_LAUNCH * bs) if kernel_avg_ms > 0 else None
        line = {
            "metric": "UGen-samples/sec (voices x block_size x UGens / s)",
            "value": value,
            "unit": "UGen-samples/s",
            "n_gpus": world,
            "steps": m["steps"],
            "warmup": args.warmup,
            "ms_per_step": m["elapsed"] / m["steps"] * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if headline == "C3" else "strong",
            "vs_baseline": None,
            "dtype": "f64" if f64 else "f32",
            "data": "synthetic" if not env.rehearse else "synthetic (REHEARSAL: ranks share GPUs, host-side gloo reduce; not a measurement)",
            "config": {
                "workload": ("C3" if headline == "C3" else "C4") + ": SinWt.wr_mul(1/N) -> SvfFilter(Low) -> * EnvAsr, stereo mix"
                            + (", f64 samples" if f64 else ""),
                "voices_per_gpu": nv_rank, "voices_total": total_voices, "block_size": bs, "sample_rate": 48000,
                "ugens_per_voice": ugens, "mix": "pairwise tree over the voice index per GPU (KNH_MIX_TREE, deterministic), RCCL sum across GPUs",
                "step": f"{LAUNCHES_PER_STEP} launches per rank of {BLOCKS_PER_LAUNCH} consecutive blocks of every voice each (one note cycle "
                        f"per launch); {total_blocks} blocks timed", "blocks_per_step": BLOCKS_PER_STEP, "blocks_per_launch": BLOCKS_PER_LAUNCH,
                "blocks_timed": total_blocks,
                "residency": "value is measured with voice state, events and the mixed stereo blocks resident in HBM; the "
                             "PCIe-inclusive rate of the host-pointer boundary (knh_bank_process_blocks) is host_output",
                "prewarm": f"{m['n_pre']} untimed launches (>= {PREWARM_MS:.0f} ms) before the warm-up steps, to bring the clock up",
                "arithmetic": "fma" if args.allow_fma else "exact (bit-identical per voice to the CPU oracle)",
                "parallelism": f"voices sharded over {world} rank(s), one process per GPU (knh_bank_create_rank); "
                               f"{BLOCKS_PER_LAUNCH} blocks per launch; the library's ncclReduce of the stereo blocks to rank 0 once per "
                               f"launch, on its own stream",
                "ranks_seen_by_rccl": m["ranks_seen"], "collective": m["collective"],
                "per_rank": {"kernel_ms_per_launch": [r[0] for r in m["per_rank"]], "reduce_ms_per_launch": [r[1] for r in m["per_rank"]],
                             "voices": [r[2] for r in m["per_rank"]],
                             "note": "HIP-event device time per 64-block launch on each rank: its voice kernel, and the sum of the ranks' "
                                     "mixed blocks to rank 0 (ncclReduce on the communicator's stream; overlaps the next launch's kernel)"},
                "events": "t_restart on every voice at block 0 and t_release at block 32 of every 64-block cycle",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": f"{traffic_src} (rocprofv3 FETCH_SIZE + WRITE_SIZE, separate --pmc passes, per launch)" if traffic else None,
                # (the library's choice, bank.hip make_bank: the pipeline up to 768 voice groups in f32, 512 in f64)
                "kernel": ("voice_pipe_kernel<double,false,32,PIPE_INPLACE,...>" if f64 else
                           "voice_pipe_kernel<float,false,64,PIPE_INPLACE,Group<SinWt,MulVal>,Group<Svf>,Group<MulAsr>>")
                          if nv_rank <= (16384 if f64 else 32768) else "voice_kernel<..., WAVES = 4 or 8, SinWt, MulVal, Svf, MulAsr>",
                "kernel_avg_ms": kernel_avg_ms, "launches": m["launches"], "blocks_per_launch": float(BLOCKS_PER_LAUNCH),
                "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                "note": "fused kernel moves 92 B (f64: 184 B) per voice per block; it is bound by the instruction issue of its "
                        "busiest wavefront, not by HBM (see valu and issue)",
            },
            "valu": {
                "achieved_ops_per_s": kernel_rate * OPS_PER_UGEN_SAMPLE, "peak_ops_per_s": VALU_PEAK_OPS,
                "frac": kernel_rate * OPS_PER_UGEN_SAMPLE / VALU_PEAK_OPS,
                "ops_per_ugen_sample": OPS_PER_UGEN_SAMPLE, "kernel_only_ugen_samples_per_s": kernel_rate,
            },
            "output_finite": m["sane"], "output_peak": m["peak"],
            "host_output": None if m["host_rate"] is None else {
                "value": m["host_rate"], "unit": "UGen-samples/s", "blocking_value": m["host_rate_blocking"],
                "note": "PCIe-inclusive: each 64-block launch's stereo blocks end up in host memory. value: two launches in "
                        "flight (knh_bank_process_blocks_begin / _end); blocking_value: one blocking call per launch "
                        "(knh_bank_process_blocks)",
            },
        }
        if headline == "C3" and not f64:
            # What actually bounds this kernel at 16 384 voices (one 64-voice group per CU, one wavefront per SIMD): the
            # filter wavefront's instruction stream.
            line["issue"] = {
                "bound": "instruction issue of the busiest wavefront (SVF), one wavefront per SIMD",
                "filter_step_cycles_per_sample_alone": SVF_STEP_CYCLES,
                "floor_ns_per_sample": SVF_STEP_CYCLES / SHADER_CLOCK_GHZ,
                "kernel_ns_per_sample": ns_per_sample,
                "frac": (SVF_STEP_CYCLES / SHADER_CLOCK_GHZ) / ns_per_sample if ns_per_sample and not args.allow_fma else None,
                "tile_samples": PIPE_TILE,
                "note": "floor = the seven instructions of one low-pass filter step issued by a wavefront alone on "
                        "its SIMD (29 cycles, micro-benchmark) at 2.4 GHz; the rest of the kernel's time per sample is that wavefront's "
                        "LDS hand-over (16 loads, 16 stores of 1 KiB per 64-sample tile: ~13 cycles per sample, during which it issues "
                        "nothing else), block/event bookkeeping and the workgroup barrier once per tile.  The oscillator, envelope and "
                        "mixer wavefronts on the other three SIMDs are all faster and hide behind it "
                        "(profiles/r03_pipe_wave_busy_cycles.txt, _filter_alone.txt)",
            }
        if world == 1 and not args.no_configs:
            pb = [per_block_boundary("C3"), per_block_boundary("C1")]
            if line["host_output"] is not None:
                line["host_output"]["per_block_value"] = pb[0]["per_block_value"]
                line["host_output"]["per_block"] = pb
                line["host_output"]["note"] += ("; per_block_value: ONE knh_bank_process_block call per block (what the reference's Task::run "
                                                "does, task.rs:25-31), the block in host memory when the call returns -- C3, and C1 beside it; "
                                                "served by a resident kernel (no launch per call); per_block[i].twin = the C++ twin of the Rust shim, "
                                                ".twin_launch_per_call = the same with KNH_RESIDENT=0, .python = this process's ctypes loop")
            line["configs"] = [config_leg("C1"), config_leg("C2"), config_leg("C5")]
            # Banks beyond the headline size (not BASELINE.json configs: the same C3 voice, more of them; D3 = C3 with a
            # SampleDelay of 0.25 s per voice behind the filter, the path's HBM-bound regime): whole-chain wavefronts, four or
            # eight to a workgroup.  valu = the headline's measure (OPS_PER_UGEN_SAMPLE per UGen-sample against the non-fused
            # FP32 peak); hbm = algorithmic bytes (state + ring) against 8 TB/s.
            large = []
            for nm, nv, ln in (("C3", 65536, 8), ("C3", 262144, 4), ("D3", 65536, 6), ("D3", 262144, 4)):
                try:
                    leg = config_leg(nm, launches=ln, blocks=32, n_voices=nv)
                except Exception as e:  # (a box without the memory for 12 GB of rings)
                    large.append({"config": nm, "voices": nv, "error": str(e)[:200]})
                    continue
                ko = leg["kernel_only_value"] or 0.0
                leg["valu"] = {"achieved_ops_per_s": ko * OPS_PER_UGEN_SAMPLE, "peak_ops_per_s": VALU_PEAK_OPS, "frac": ko * OPS_PER_UGEN_SAMPLE / VALU_PEAK_OPS}
                leg["hbm"] = {"achieved_gbs": leg["roofline_achieved_gbs"], "peak_gbs": HBM_PEAK_GBS, "frac": (leg["roofline_achieved_gbs"] or 0.0) / HBM_PEAK_GBS}
                large.append(leg)
            line["large_banks"] = large
        if secondary is not None:
            s = secondary
            s_blocks = s["steps"] * BLOCKS_PER_STEP
            s_kernel_rate = float(s["voices_rank0"]) * bs * s["ugens"] * BLOCKS_PER_LAUNCH / (s["kernel_avg_ms"] * 1e-3) if s["kernel_avg_ms"] > 0 else 0.0
            line["c4_strong"] = {
                "workload": "C4: the same chain, f64 samples, 65 536 voices IN ALL, split over the ranks (BASELINE.json configs[3])",
                "value": float(s["total_voices"]) * bs * s["ugens"] * s_blocks / s["elapsed"], "unit": "UGen-samples/s",
                "scaling": "strong", "dtype": "f64", "n_gpus": world, "voices_total": s["total_voices"], "voices_per_gpu": s["voices_rank0"],
                "steps": s["steps"], "ms_per_step": s["elapsed"] / s["steps"] * 1e3, "kernel_avg_ms": s["kernel_avg_ms"],
                "kernel_only_ugen_samples_per_s_per_gpu": s_kernel_rate,
                "roofline_achieved_gbs": float(s["bytes_per_voice_block"]) * s["voices_rank0"] * BLOCKS_PER_LAUNCH / (s["kernel_avg_ms"] * 1e-3) / 1e9
                if s["kernel_avg_ms"] > 0 else 0.0,
                "ranks_seen_by_rccl": s["ranks_seen"], "output_finite": s["sane"],
                # What bounds a rank's share: an f64 wavefront alone on its SIMD issues an instruction every ~4.4 cycles.  Up to 256
                # voice groups per GPU (the pipeline, a group per CU) the filter wavefront -- its arithmetic plus its tile's way
                # through LDS -- sets a block's time WHATEVER the number of voices; beyond that (whole-chain wavefronts, one per
                # SIMD up to 65 536 voices) it is every stage's instructions.  Both floors come from measurements committed under
                # profiles/ (tools/c4_floors.py -> profiles/r04_c4_floors.json: per-wavefront stamps of the pipeline, SQ counters
                # of the whole-chain kernel), not from literals here.
                "issue": c4_issue(s["voices_rank0"], bs, s["kernel_avg_ms"] * 1e3 / BLOCKS_PER_LAUNCH),
                "per_rank": {"kernel_ms_per_launch": [r[0] for r in s["per_rank"]], "reduce_ms_per_launch": [r[1] for r in s["per_rank"]],
                             "voices": [r[2] for r in s["per_rank"]]},
            }
        if not args.no_cpu_baseline and world == 1 and headline == "C3":
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
    if world > 1:
        env.dist.destroy_process_group()


if __name__ == "__main__":
    main()
