#!/usr/bin/env python3
"""Per-launch SQ counters of one kernel from a rocprofv3 --pmc pass (counter_collection.csv), per voice-sample.
usage: python tools/sq_summary_any.py <csv> <kernel name substring> <voices> <block_size> <blocks per launch> [label]"""
import csv
import json
import sys

COUNTERS = ["SQ_INSTS_VALU", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_LDS"]
path, key, voices, bs, blocks = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
per = {}
name = None
with open(path) as f:
    for row in csv.DictReader(f):
        if key in row["Kernel_Name"] and row["Counter_Name"] in COUNTERS:
            name = row["Kernel_Name"]
            per.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
out = {"kernel": name, "label": sys.argv[6] if len(sys.argv) > 6 else "", "voices": voices, "block_size": bs, "blocks_per_launch": blocks}
for c, vals in sorted(per.items()):
    vals = vals[len(vals) // 2:]  # the later launches: steady state
    out[c] = sum(vals) / len(vals)
    out["launches_sampled"] = len(vals)
vs = float(voices) * bs * blocks
if "SQ_INSTS_VALU" in out:
    out["valu_wave_insts_per_voice_sample"] = out["SQ_INSTS_VALU"] * 64.0 / vs
    out["lds_wave_insts_per_voice_sample"] = out["SQ_INSTS_LDS"] * 64.0 / vs
    out["wave_cycles_per_sample_of_a_wavefront"] = out["SQ_WAVE_CYCLES"] * 4.0 / out["SQ_WAVES"] / (bs * blocks)
    out["active_valu_over_wave_cycles"] = out["SQ_ACTIVE_INST_VALU"] / out["SQ_WAVE_CYCLES"]
    out["wait_any_over_wave_cycles"] = out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"]
    out["wait_inst_any_over_wave_cycles"] = out["SQ_WAIT_INST_ANY"] / out["SQ_WAVE_CYCLES"]
    # the issue floor of a wavefront alone on its SIMD: one instruction per 4.4 cycles (profiles/r01_micro_valu_issue.txt), whatever it is
    out["issue_floor_cycles_per_sample"] = 4.4 * (out["valu_wave_insts_per_voice_sample"] + out["lds_wave_insts_per_voice_sample"])
out["notes"] = ["SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over wavefronts (MI355X_MICROARCH.md, cycle constants)",
                "rocprofv3 --pmc " + " ".join(COUNTERS) + " --kernel-trace -- python3 tools/bench_configs.py only <config>:<voices>  (32 blocks per launch; later half of the launches)"]
print(json.dumps(out, indent=1))
