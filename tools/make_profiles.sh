#!/bin/bash
# Regenerates the measurement artefacts under gpurun_out/profiles_new/ on the GPU box:
#   bash tools/make_profiles.sh        (run through gpurun; copy what should be judged into profiles/)
# rocprofv3: the program goes directly after `--`; PMC counters in their own passes (no trace domains).
set -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/profiles_new"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5"   # the driver's command (one step = one 64-block launch)

echo "== bench (plain)"; (cd "$ROOT" && timeout -k 10 300 python3 bench.py > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err") || exit 1
echo "== kernel trace + stats"
rm -rf "$OUT/stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err" || { tail -5 "$OUT/stats.err"; exit 1; }
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \;
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C"
  rm -rf "$OUT/pmc_$C"
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o bench -- python3 "$ROOT/bench.py" --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-c4 > "$OUT/pmc_$C.json" 2> "$OUT/pmc_$C.err" || { tail -5 "$OUT/pmc_$C.err"; exit 1; }
  find "$OUT/pmc_$C" -name "*counter_collection.csv" -exec cp {} "$OUT/pmc_$C.csv" \;
done
echo "== all configs"; (cd "$ROOT" && timeout -k 10 400 python3 tools/bench_configs.py > "$OUT/all_configs.jsonl" 2> "$OUT/all_configs.err") || tail -3 "$OUT/all_configs.err"
echo "== micro"; (cd "$ROOT" && timeout -k 10 60 ./tools/micro/valu_issue > "$OUT/micro_valu_issue.txt" 2>&1; timeout -k 10 60 ./tools/micro/exec_mask > "$OUT/micro_exec_mask.txt" 2>&1)
rm -rf "$OUT"/stats "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE
ls -la "$OUT"
