#!/usr/bin/env python3
"""Per-launch SQ counters of the bench's kernels from the rocprofv3 PMC pass of tools/make_profiles.sh.
usage: python tools/sq_summary.py gpurun_out/profiles_new/pmc_SQ.csv > profiles/r03_sq_counters.json"""
import csv
import json
import sys

COUNTERS = ["SQ_INSTS_VALU", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_LDS"]
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/profiles_new/pmc_SQ.csv"
per = {}
with open(path) as f:
    for row in csv.DictReader(f):
        for key in ("voice_pipe_kernel", "fold_tree_kernel"):
            if key in row["Kernel_Name"] and row["Counter_Name"] in COUNTERS:
                per.setdefault(key, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
out = {}
for key, counters in per.items():
    e = out.setdefault(key, {})
    for name, vals in sorted(counters.items()):
        vals = vals[len(vals) // 2:]  # the later launches: steady state
        e[name] = sum(vals) / len(vals)
        e["launches_sampled"] = len(vals)
v = out.get("voice_pipe_kernel", {})
if "SQ_INSTS_VALU" in v:
    voice_samples = 16384 * 512 * 64
    v["valu_insts_per_wave"] = v["SQ_INSTS_VALU"] / v["SQ_WAVES"]
    v["valu_wave_insts_per_voice_sample"] = v["SQ_INSTS_VALU"] * 64.0 / voice_samples
    v["lds_wave_insts_per_voice_sample"] = v["SQ_INSTS_LDS"] * 64.0 / voice_samples
    v["active_valu_over_wave_cycles"] = v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]
    v["wait_inst_any_over_wave_cycles"] = v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]
    v["wait_any_over_wave_cycles"] = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]
out["command"] = ("tools/make_profiles.sh: rocprofv3 --pmc " + " ".join(COUNTERS) + " --kernel-trace --output-format csv -- python3 bench.py "
                  "--gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-c4 --no-configs (per launch of 64 blocks x 16 384 voices; later half of the launches); tools/sq_summary.py")
out["notes"] = [
    "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over wavefronts (MI355X_MICROARCH.md, cycle constants)",
    "1 024 wavefronts per launch: 256 workgroups x (oscillator, filter, envelope, mixer); the ratios are averages over the four roles -- "
    "the filter wavefront alone is busy ~95 % of a tile step (profiles/r03_pipe_wave_busy_cycles.txt)",
    "valu_wave_insts_per_voice_sample: wave-level VALU instructions x 64 lanes / (voices x frames x blocks): the instructions one voice-sample costs, all roles",
]
print(json.dumps(out, indent=1))
