#!/bin/bash
# D3 (delay chain) kernel times and HBM counters -> gpurun_out/profiles_new/delay_* and, summarised, r04_delay_hbm.json there
#   bash tools/profile_delay.sh      (through gpurun; copy the summary into profiles/)
set -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/profiles_new"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SIZES="16384 65536 262144"
timeout -k 10 300 python3 "$ROOT/tools/bench_delay.py" $SIZES > "$OUT/delay_bench.jsonl" 2> "$OUT/delay_bench.err" || { tail -5 "$OUT/delay_bench.err"; exit 1; }
cat "$OUT/delay_bench.jsonl"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$OUT/dpmc_$C"
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/dpmc_$C" -o d3 -- python3 "$ROOT/tools/bench_delay.py" $SIZES > "$OUT/delay_pmc_$C.jsonl" 2> "$OUT/delay_pmc_$C.err" || { tail -5 "$OUT/delay_pmc_$C.err"; exit 1; }
  find "$OUT/dpmc_$C" -name "*counter_collection.csv" -exec cp {} "$OUT/delay_pmc_$C.csv" \;
  rm -rf "$OUT/dpmc_$C"
done
rm -rf "$OUT/dstats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/dstats" -o d3 -- python3 "$ROOT/tools/bench_delay.py" $SIZES > "$OUT/delay_under_rocprof.jsonl" 2> "$OUT/dstats.err" || { tail -5 "$OUT/dstats.err"; exit 1; }
find "$OUT/dstats" -name "*kernel_stats.csv" -exec cp {} "$OUT/delay_kernel_stats.csv" \;
rm -rf "$OUT/dstats"
python3 - "$OUT" <<'PY'
import csv, sys, collections, json
out = sys.argv[1]
bench = {json.loads(l)["voices"]: json.loads(l) for l in open(f"{out}/delay_bench.jsonl")}
acc = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{out}/delay_pmc_{c}.csv")):
        if "SampleDelay" in r["Kernel_Name"] and r["Counter_Name"] == c:
            per[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    for k, v in per.items():
        v = v[len(v) // 2:]  # the later launches (the rings are all touched by then)
        acc[k][c] = sum(v) / len(v)
        acc[k]["launches_sampled"] = len(v)
res = {"workload": "D3 = C3 with SampleDelay(0.25 s) behind the filter; per-voice delay_time 10..210 ms; f32; 32 blocks of 512 frames per launch",
       "command": "tools/profile_delay.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace (separate passes) -- python3 tools/bench_delay.py 16384 65536 262144; times from the plain run before them",
       "notes": ["FETCH_SIZE / WRITE_SIZE in units of 1024 B as rocprofv3 reports them",
                 "the ring traffic is now whole 128-byte lines, 16 bytes per lane (RingLines): the pattern MI355X_MICROARCH.md's gfx950 correction is calibrated for, so reads = 2 x FETCH_SIZE; WRITE_SIZE as reported",
                 "algorithmic ring bytes per launch = voices x 512 frames x 32 blocks x 4 B, read and written"],
       "kernels": []}
for (name, grid), d in sorted(acc.items(), key=lambda kv: kv[0][1]):
    waves = 4 if ", 4, knh_dev" in name else (8 if ", 8, knh_dev" in name else 0)
    per_wg = {0: 64, 4: 256, 8: 512}[waves]
    threads_per_wg = {0: 256, 4: 256, 8: 512}[waves]
    voices = None
    for nv in bench:
        groups = (nv + 63) // 64
        wgs = groups if waves == 0 else (groups + waves - 1) // waves
        if wgs * threads_per_wg == grid:
            voices = nv
    e = {"kernel": name[:150], "grid_size": grid, "voices": voices, "FETCH_SIZE_KB_per_launch": round(d.get("FETCH_SIZE", 0), 1),
         "WRITE_SIZE_KB_per_launch": round(d.get("WRITE_SIZE", 0), 1), "launches_sampled": d["launches_sampled"]}
    if voices:
        ring = voices * 512 * 32 * 4
        ms = bench[voices]["kernel_ms_per_launch"]
        rd, wr = 2 * e["FETCH_SIZE_KB_per_launch"] * 1024, e["WRITE_SIZE_KB_per_launch"] * 1024
        e.update({"kernel_ms_per_launch": ms, "algorithmic_ring_bytes_read_per_launch": ring, "algorithmic_ring_bytes_written_per_launch": ring,
                  "hbm_read_bytes_per_launch": rd, "hbm_written_bytes_per_launch": wr,
                  "read_over_algorithmic": rd / ring, "written_over_algorithmic": wr / ring,
                  "algorithmic_TBps": 2 * ring / (ms * 1e-3) / 1e12, "hbm_TBps_from_counters": (rd + wr) / (ms * 1e-3) / 1e12,
                  "frac_of_8TBps": 2 * ring / (ms * 1e-3) / 8e12})
    res["kernels"].append(e)
json.dump(res, open(f"{out}/r04_delay_hbm.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
