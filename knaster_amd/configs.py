"""The five BASELINE.json workloads as voice-chain descriptors + deterministic synthetic inputs.

Inputs follow SURVEY.md section 8(d): per-voice parameters from the reference's xorshift32
(knaster_core_dsp/src/dsp/xorrng.rs:23-50: x^=x<<13; x^=x>>17; x^=x<<5; gen_f32 = x as f32 /
u32::MAX as f32), seed 0x9E3779B9, one stream, voices in index order, seven draws per voice:
freq, cutoff, q, attack, release, fm_ratio, fm_index.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np

from . import _lib as L
from .bank import Stage

SEED = 0x9E3779B9
SAMPLE_RATE = 48000


def xorshift32_stream(seed: int, n: int) -> np.ndarray:
    """n successive gen_u32() values."""
    x = seed & 0xFFFFFFFF or 17
    out = np.empty(n, dtype=np.uint32)
    for i in range(n):
        x ^= (x << 13) & 0xFFFFFFFF
        x ^= x >> 17
        x ^= (x << 5) & 0xFFFFFFFF
        out[i] = x
    return out


_PARAM_CACHE: Dict[tuple, Dict[str, np.ndarray]] = {}


def voice_parameters(n_voices: int, seed: int = SEED) -> Dict[str, np.ndarray]:
    key = (n_voices, seed)
    if key not in _PARAM_CACHE:
        _PARAM_CACHE[key] = _voice_parameters(n_voices, seed)
    return {k: v.copy() for k, v in _PARAM_CACHE[key].items()}


def _voice_parameters(n_voices: int, seed: int) -> Dict[str, np.ndarray]:
    raw = xorshift32_stream(seed, n_voices * 7).reshape(n_voices, 7)
    u = (raw.astype(np.float32) / np.float32(0xFFFFFFFF)).astype(np.float64)  # gen_f32
    return {
        "freq": 55.0 * np.exp2(6.0 * u[:, 0]),
        "cutoff": 200.0 + 7800.0 * u[:, 1],
        "q": 0.5 + 3.5 * u[:, 2],
        "attack": 0.002 + 0.02 * u[:, 3],
        "release": 0.05 + 0.25 * u[:, 4],
        "fm_ratio": 1.0 + 3.0 * u[:, 5],
        "fm_index": 500.0 * u[:, 6],
    }


@dataclass
class Workload:
    name: str
    stages: List[Stage]
    n_voices: int
    block_size: int
    sample_type: int
    out_channels: int = 2
    ctor: Dict[int, np.ndarray] = field(default_factory=dict)  # stage -> [n_voices, n_args]
    restart: tuple = ()  # (stage, param) trigger fired on every voice before block 0
    release: tuple = ()  # (stage, param, block) trigger fired on every voice before `block`
    description: str = ""
    delay_times: np.ndarray = None  # D3: per-voice SampleDelay delay_time set before block 0 (stage 3, param 0)
    buffer: tuple = None  # (stage, samples, sample_rate): the Buffer of a BufferReader stage
    in_channels: int = 0  # UGen::Inputs of the bank node (KNH_STAGE_INPUT stages read them)


def config(name: str, n_voices: int | None = None, block_size: int | None = None, sample_type: int | None = None,
           precise: int = 0) -> Workload:
    """name in {"C1".."C5"}; sizes default to the BASELINE.json values."""
    name = name.upper()
    defaults = {"C1": (1, 64, L.F32), "C2": (1024, 256, L.F32), "C3": (16384, 512, L.F32),
                "C4": (65536, 512, L.F64), "C5": (4096, 128, L.F32),
                "D3": (16384, 512, L.F32), "B3": (16384, 512, L.F32), "M1": (600, 64, L.F32), "P3": (16384, 512, L.F32)}
    nv, bs, st = defaults[name]
    nv = n_voices or nv
    bs = block_size or bs
    st = st if sample_type is None else sample_type
    p = voice_parameters(nv)
    gain = np.full(nv, 1.0 / nv)
    col = lambda a: np.asarray(a, dtype=np.float64).reshape(nv, -1)
    if name == "C1":  # README.md:34-51: SinWt(440) * 0.2 -> both outputs
        w = Workload(name, [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_MUL_CONST)], nv, bs, st,
                     description="SinWt(440)*0.2 -> stereo")
        w.ctor = {0: col(np.full(nv, 440.0)), 1: col(np.full(nv, 0.2))}
    elif name == "C2":  # SinNumeric + gain
        w = Workload(name, [Stage(L.STAGE_SIN_NUMERIC), Stage(L.STAGE_MUL_CONST)], nv, bs, st,
                     description="SinNumeric * gain")
        w.ctor = {0: col(p["freq"]), 1: col(gain)}
    elif name in ("C3", "C4"):  # SinWt.wr_mul(gain) -> SvfFilter(Low) -> * EnvAsr
        w = Workload(name, [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_WR_MUL), Stage(L.STAGE_SVF),
                            Stage(L.STAGE_MUL_ENV_ASR, delayed_changes_per_block=precise)], nv, bs, st,
                     description="SinWt.wr_mul(1/N) -> SvfFilter(Low) -> * EnvAsr")
        svf = np.stack([np.full(nv, float(L.SVF_LOW)), p["cutoff"], p["q"], np.zeros(nv)], axis=1)
        w.ctor = {0: col(p["freq"]), 1: col(gain), 2: svf, 3: np.stack([p["attack"], p["release"]], axis=1)}
        w.restart = (3, 3)
        w.release = (3, 2, 32)
    elif name == "M1":  # knaster/examples/many_sines.rs:51-63: (EnvAr(0.01, 0.1) * SinWt(f).wr_mul(amp)) >> Pan2(pan), 600 voices
        w = Workload(name, [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_WR_MUL), Stage(L.STAGE_MUL_ENV_AR), Stage(L.STAGE_PAN2)], nv, bs, st,
                     description="(EnvAr * SinWt.wr_mul(amp)) >> Pan2(pan) -> graph out (many_sines.rs)")
        u = (p["q"] - 0.5) / 3.5  # three of the uniform draws, mapped onto the example's ranges
        w.ctor = {0: col(3000.0 + 7000.0 * (p["cutoff"] - 200.0) / 7800.0), 1: col(0.01 + 0.005 * u),
                  2: np.tile([0.01, 0.1], (nv, 1)), 3: col(-1.0 + 2.0 * (p["fm_ratio"] - 1.0) / 3.0)}
        w.restart = (2, 2)
    elif name == "P3":  # not a BASELINE.json config: the C3 voice panned (Pan2 behind the envelope)
        w = Workload(name, [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_WR_MUL), Stage(L.STAGE_SVF),
                            Stage(L.STAGE_MUL_ENV_ASR, delayed_changes_per_block=precise), Stage(L.STAGE_PAN2)], nv, bs, st,
                     description="SinWt.wr_mul(1/N) -> SvfFilter(Low) -> * EnvAsr -> Pan2")
        svf = np.stack([np.full(nv, float(L.SVF_LOW)), p["cutoff"], p["q"], np.zeros(nv)], axis=1)
        w.ctor = {0: col(p["freq"]), 1: col(gain), 2: svf, 3: np.stack([p["attack"], p["release"]], axis=1),
                  4: col(-1.0 + 2.0 * (p["fm_ratio"] - 1.0) / 3.0)}
        w.restart = (3, 3)
        w.release = (3, 2, 32)
    elif name == "B3":  # not a BASELINE.json config: the C3 voice with a band-limited PolyBlep oscillator; one waveform per
        # 64-voice group (a wavefront executes every waveform its lanes hold, so a bank is laid out by waveform)
        w = Workload(name, [Stage(L.STAGE_POLYBLEP), Stage(L.STAGE_WR_MUL), Stage(L.STAGE_SVF),
                            Stage(L.STAGE_MUL_ENV_ASR, delayed_changes_per_block=precise)], nv, bs, st,
                     description="PolyBlep((voice / 64) % 14).wr_mul(1/N) -> SvfFilter(Low) -> * EnvAsr")
        svf = np.stack([np.full(nv, float(L.SVF_LOW)), p["cutoff"], p["q"], np.zeros(nv)], axis=1)
        w.ctor = {0: np.stack([(np.arange(nv) // 64) % 14, p["freq"]], axis=1).astype(np.float64), 1: col(gain), 2: svf,
                  3: np.stack([p["attack"], p["release"]], axis=1)}
        w.restart = (3, 3)
        w.release = (3, 2, 32)
    elif name == "D3":  # not a BASELINE.json config: C3 with a SampleDelay behind the filter (per-voice rings in HBM)
        w = Workload(name, [Stage(L.STAGE_SIN_WT), Stage(L.STAGE_WR_MUL), Stage(L.STAGE_SVF), Stage(L.STAGE_SAMPLE_DELAY),
                            Stage(L.STAGE_MUL_ENV_ASR, delayed_changes_per_block=precise)], nv, bs, st,
                     description="SinWt.wr_mul(1/N) -> SvfFilter(Low) -> SampleDelay(0.25 s) -> * EnvAsr")
        svf = np.stack([np.full(nv, float(L.SVF_LOW)), p["cutoff"], p["q"], np.zeros(nv)], axis=1)
        w.ctor = {0: col(p["freq"]), 1: col(gain), 2: svf, 3: col(np.full(nv, 0.25)), 4: np.stack([p["attack"], p["release"]], axis=1)}
        w.restart = (4, 3)
        w.release = (4, 2, 32)
        w.delay_times = 0.01 + 0.2 * (p["q"] - 0.5) / 3.5  # 10 .. 210 ms, different per voice
    elif name == "C5":  # modulator SinWt * index + carrier_freq -> carrier SinWt.ar_params() "freq"; * gain
        pr = precise or 4
        w = Workload(name, [Stage(L.STAGE_SIN_WT, delayed_changes_per_block=pr), Stage(L.STAGE_MUL_CONST),
                            Stage(L.STAGE_ADD_CONST),
                            Stage(L.STAGE_SIN_WT, flags=L.STAGE_FLAG_AR_FREQ, delayed_changes_per_block=pr),
                            Stage(L.STAGE_MUL_CONST)], nv, bs, st,
                     description="SinWt(mod) * index + f0 -> SinWt.ar_params().precise_timing() freq; * 1/N")
        w.ctor = {0: col(p["freq"] * p["fm_ratio"]), 1: col(p["fm_index"]), 2: col(p["freq"]), 3: col(p["freq"]),
                  4: col(gain)}
    else:
        raise KeyError(name)
    return w


def c5_events(w: Workload, block: int):
    """C5's sample-accurate changes for `block`: every second block each voice gets one delayed change
    at in-block frame (17*voice) mod block_size: even voices a new modulator freq, odd voices a new
    carrier phase_offset.  Returns arrays for param_apply_many or None."""
    if block % 2 != 0:
        return None
    nv, bs = w.n_voices, w.block_size
    voices = np.arange(nv, dtype=np.uint32)
    delays = ((17 * voices.astype(np.int64)) % bs).astype(np.uint16)
    stages = np.where(voices % 2 == 0, 0, 3).astype(np.uint32)
    params = np.where(voices % 2 == 0, 0, 1).astype(np.uint32)
    p = voice_parameters(nv)
    k = 1.0 + 0.01 * ((block // 2) % 7)
    fvalues = np.where(voices % 2 == 0, p["freq"] * p["fm_ratio"] * k, 1000.0 * ((block // 2) % 16))
    kinds = np.full(nv, L.VALUE_FLOAT, dtype=np.uint32)
    return voices, stages, params, kinds, fvalues.astype(np.float64), delays


def fm_cascade(depth: int, n_voices: int = 1, block_size: int = 128, sample_type: int = L.F32, detune: float = 0.001,
               add: float = 440.0, gain: float = 0.05) -> Workload:
    """knaster_benchmarks/benches/graph_dsp_performance.rs:37-72 ("256 FM cascade": depth = 256, one voice), as ONE voice that
    is a graph of `depth` oscillators:
         i = 0:  (c * s0).to_graph_out();  l = s0
         i > 0:  add = l * 440.0;  mul = s_i * l;  node = mul + add;  node.to_graph_out();  l = node * c
    The additive graph outputs are the reference's chain of Add nodes (graph.rs:850-864): acc = acc + node.  Voice v's
    oscillators are detuned by (1 + detune * v) so that voices differ.  With the reference's constants (add = 440, gain = 0.05) the
    signal grows 22-fold per oscillator and is infinite, then NaN, from the 30th on (the bench measures time, not sound);
    add = 19 keeps it of order one."""
    st, ctor = [], {}

    def push(stage, args=None):
        st.append(stage)
        if args is not None:
            ctor[len(st) - 1] = args
        return len(st)  # 1 + index: the value `input` takes to name this stage
    s0 = push(Stage(L.STAGE_SIN_WT), 220.0)
    acc = push(Stage(L.STAGE_MUL_CONST, input=s0), gain)
    last = s0
    for i in range(1, depth):
        add_ = push(Stage(L.STAGE_MUL_CONST, input=last), add)
        s = push(Stage(L.STAGE_SIN_WT), 220.0 + i)
        mul = push(Stage(L.STAGE_MATH_MUL, input=s, input2=last))
        node = push(Stage(L.STAGE_MATH_ADD, input=mul, input2=add_))
        if i + 1 < depth:
            last = push(Stage(L.STAGE_MUL_CONST, input=node), gain)
        acc = push(Stage(L.STAGE_MATH_ADD, input=acc, input2=node))
    w = Workload("FMC", st, n_voices, block_size, sample_type, 1, description=f"FM cascade of {depth} oscillators per voice")
    scale = 1.0 + detune * np.arange(n_voices)
    w.ctor = {s: (a * scale if st[s].kind == L.STAGE_SIN_WT else np.full(n_voices, a)).reshape(n_voices, 1) for s, a in ctor.items()}
    return w
