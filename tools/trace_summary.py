#!/usr/bin/env python3
"""The headline kernel's launches in a rocprofv3 --kernel-trace CSV of `bench.py`, by kind.
`bench.py` launches the same kernel for its 64-block steps (the timed region, the warm-up, the host-output legs), one block at
a time for the launch-per-call leg of the per-block boundary, and ONCE per bank as a resident launch that stays on the chip
for all of that bank's per-block calls (round 4); `--stats` averages over all of them.  This splits them: 0.3 .. 5 ms are
64-block launches, shorter ones single blocks, longer ones resident launches.
usage: python tools/trace_summary.py <..._kernel_trace.csv> > profiles/r04_bench_kernel_trace_summary.json"""
import csv
import json
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if "voice_pipe_kernel<float, false, 64, 2, 1" in r["Kernel_Name"] and "MulEnvT" in r["Kernel_Name"] and "Svf" in r["Kernel_Name"]:
            rows.append((r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6))
big = [d for _, d in rows if 0.3 <= d < 5.0]
small = [d for _, d in rows if d < 0.3]
resident = [d for _, d in rows if d >= 5.0]
tail = big[len(big) // 2:]
out = {
    "kernel": rows[0][0] if rows else None,
    "launches_of_64_blocks": {"count": len(big), "average_ms": sum(big) / len(big) if big else None,
                              "average_ms_later_half": sum(tail) / len(tail) if tail else None, "min_ms": min(big) if big else None, "max_ms": max(big) if big else None},
    "launches_of_one_block": {"count": len(small), "average_ms": sum(small) / len(small) if small else None},
    "resident_launches": {"count": len(resident), "ms_each": [round(d, 3) for d in resident]},
    "note": "the --stats average of this kernel (bench_kernel_stats.csv) is over all kinds; roofline.kernel_avg_ms of the bench line is the "
            "HIP-event average of the 64-block launches of the timed region",
}
print(json.dumps(out, indent=1))
