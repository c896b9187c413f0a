#!/usr/bin/env python3
"""The floors bench.py quotes for C4 (`c4_strong.issue.floor_*`), derived from the measurements committed under profiles/.

Two regimes:
  * up to 16 384 f64 voices per GPU -- the four-wavefront pipeline, one 64-voice group per CU.  What paces a block is the
    filter wavefront: its arithmetic plus its tile's way in and out of LDS, cycles per 32-sample tile from the per-wavefront
    stamps of profiles/r03_c4_pipe_wave_busy_cycles.txt (the later run: low-pass step of eleven f64 instructions).
  * beyond -- whole-chain wavefronts, one per SIMD up to 65 536 voices.  A wavefront alone on its SIMD issues an instruction
    every ISSUE_CYCLES cycles (tools/micro/valu_issue.hip; f64 instructions hold the SIMD four cycles, less than that), so
    the floor is (VALU + LDS wave-instructions per voice-sample) x ISSUE_CYCLES, the counts from the SQ counters of
    profiles/r04_c4_wide_sq_counters.json.
Writes profiles/r04_c4_floors.json; no GPU needed.
"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ISSUE_CYCLES = 4.4       # tools/micro/valu_issue.hip: one wavefront on a SIMD, dependent or independent instructions alike
SHADER_CLOCK_GHZ = 2.4   # tools/micro/clock_share.hip: what the chip holds under this load


def pipeline_floor():
    path = os.path.join(ROOT, "profiles", "r03_c4_pipe_wave_busy_cycles.txt")
    text = open(path).read()
    later = text[text.index("--- later in round 3"):]
    m = re.search(r"sustain.*?in/out per group: \[([\d\s]+)\] last group's arithmetic / fold: \[\s*(\d+)\s+(\d+)\]", later)
    io = [int(x) for x in m.group(1).split()]
    filt_in, filt_out = io[2], io[3]          # the filter is the second of three groups: [in0 out0 in1 out1 in2 out2]
    m2 = re.search(r"folding group \(64-sample form\): stage arithmetic and fold, cycles per tile: \[\s*(\d+)\s+(\d+)\]", later)
    arith = int(m2.group(1))                  # filter arithmetic per 32-sample tile (the stamped build's filter group)
    tile = 32
    cyc = (arith + filt_in + filt_out) / tile
    return {"source": "profiles/r03_c4_pipe_wave_busy_cycles.txt (later run, sustain)", "tile_samples": tile,
            "filter_arithmetic_cycles_per_tile": arith, "filter_tile_in_cycles": filt_in, "filter_tile_out_cycles": filt_out,
            "floor_cycles_per_sample": cyc}


def wide_floor():
    path = os.path.join(ROOT, "profiles", "r04_c4_wide_sq_counters.json")
    c = json.load(open(path))
    insts = c["valu_wave_insts_per_voice_sample"] + c["lds_wave_insts_per_voice_sample"]
    return {"source": "profiles/r04_c4_wide_sq_counters.json", "valu_wave_insts_per_voice_sample": c["valu_wave_insts_per_voice_sample"],
            "lds_wave_insts_per_voice_sample": c["lds_wave_insts_per_voice_sample"], "issue_cycles_per_instruction": ISSUE_CYCLES,
            "floor_cycles_per_sample": insts * ISSUE_CYCLES,
            "measured_wave_cycles_per_sample": c["wave_cycles_per_sample_of_a_wavefront"], "voices_at_one_wavefront_per_simd": 65536}


def main():
    out = {"shader_clock_ghz": SHADER_CLOCK_GHZ, "block_size": 512, "pipeline": pipeline_floor(), "wide": wide_floor()}
    for k in ("pipeline", "wide"):
        out[k]["floor_us_per_block_of_512"] = out[k]["floor_cycles_per_sample"] * 512 / (SHADER_CLOCK_GHZ * 1e3)
    path = os.path.join(ROOT, "profiles", "r04_c4_floors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
