#!/bin/bash
# D3 (delay chain) kernel times and HBM counters -> gpurun_out/profiles_new/delay_*
set -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/profiles_new"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 "$ROOT/tools/bench_delay.py" > "$OUT/delay_bench.jsonl" 2> "$OUT/delay_bench.err" || { tail -5 "$OUT/delay_bench.err"; exit 1; }
cat "$OUT/delay_bench.jsonl"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$OUT/dpmc_$C"
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/dpmc_$C" -o d3 -- python3 "$ROOT/tools/bench_delay.py" 16384 262144 > "$OUT/delay_pmc_$C.jsonl" 2> "$OUT/delay_pmc_$C.err" || { tail -5 "$OUT/delay_pmc_$C.err"; exit 1; }
  find "$OUT/dpmc_$C" -name "*counter_collection.csv" -exec cp {} "$OUT/delay_pmc_$C.csv" \;
  rm -rf "$OUT/dpmc_$C"
done
python3 - "$OUT" <<'PY'
import csv, sys, collections
out = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{out}/delay_pmc_{c}.csv")):
        if "voice_" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:60], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(c, k, len(v), "mean KB per launch", sum(v) / len(v))
PY
