#!/usr/bin/env python3
"""Per-launch HBM traffic of the bench's kernels from the two rocprofv3 PMC passes of tools/make_profiles.sh.
usage: python tools/pmc_summary.py gpurun_out/profiles_new > profiles/r02_hbm_traffic.json"""
import csv
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/profiles_new"
out = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    per = {}
    with open(os.path.join(d, f"pmc_{counter}.csv")) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            for key in ("voice_pipe_kernel", "fold_tree_kernel"):
                if key in name and row["Counter_Name"] == counter:
                    per.setdefault(key, []).append(float(row["Counter_Value"]))
    for key, vals in per.items():
        vals = vals[len(vals) // 2:]  # the later launches: steady state (the first ones include first-touch effects)
        e = out.setdefault(key, {})
        e[f"{counter}_KB_per_launch"] = round(sum(vals) / len(vals), 1)
        e["launches_sampled"] = len(vals)
v = out.get("voice_pipe_kernel", {})
if "FETCH_SIZE_KB_per_launch" in v and "WRITE_SIZE_KB_per_launch" in v:
    v["hbm_bytes_per_launch"] = (v["FETCH_SIZE_KB_per_launch"] + v["WRITE_SIZE_KB_per_launch"]) * 1024.0
out["command"] = ("tools/make_profiles.sh: rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- "
                  "python3 bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-c4 --no-configs (separate passes; one step = four 64-block launches); tools/pmc_summary.py")
out["workload"] = {"voices": 16384, "block_size": 512, "blocks_per_launch": 64, "chain": "WmSA", "sample_type": "f32"}
out["notes"] = [
    "FETCH_SIZE/WRITE_SIZE are in units of 1024 B as rocprofv3 reports them",
    "voice kernel reads are 4 B/lane dword loads (state SoA), not the 16 B/lane streaming pattern the gfx950 FETCH_SIZE x2 "
    "correction is calibrated for; no correction applied",
    "WRITE_SIZE of the voice kernel = per-wavefront partial rows: 64 blocks x 256 waves x 512 frames x 4 B = 33.5 MB, plus 0.4 MB of state",
    "event lists are read by the kernel from pinned host memory (PCIe), they do not appear in these memory-side counters",
]
print(json.dumps(out, indent=1))
