#!/bin/bash
# Regenerates the measurement artefacts under gpurun_out/profiles_new/ on the GPU box:
#   bash tools/make_profiles.sh        (run through gpurun; copy what should be judged into profiles/)
# rocprofv3: the program goes directly after `--`; PMC counters in their own passes (no trace domains).
set -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/profiles_new"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5"   # the driver's command (one step = four 64-block launches)

echo "== bench (plain)"; (cd "$ROOT" && timeout -k 10 300 python3 bench.py > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err") || exit 1
echo "== kernel trace + stats"
rm -rf "$OUT/stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err" || { tail -5 "$OUT/stats.err"; exit 1; }
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \;
find "$OUT/stats" -name "*kernel_trace.csv" -exec python3 "$ROOT/tools/trace_summary.py" {} \; > "$OUT/bench_kernel_trace_summary.json"
echo "== kernel stats of the headline launches alone (no per-block / C1 / C2 / C5 / C4 legs: every launch of the kernel is a 64-block one)"
rm -rf "$OUT/stats_h"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_h" -o bench -- $BENCH --no-configs --no-c4 --no-cpu-baseline > "$OUT/bench_headline_only_under_rocprof.json" 2> "$OUT/stats_h.err" || { tail -5 "$OUT/stats_h.err"; exit 1; }
find "$OUT/stats_h" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_headline_only_kernel_stats.csv" \;
rm -rf "$OUT/stats_h"
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C"
  rm -rf "$OUT/pmc_$C"
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o bench -- python3 "$ROOT/bench.py" --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-c4 --no-configs > "$OUT/pmc_$C.json" 2> "$OUT/pmc_$C.err" || { tail -5 "$OUT/pmc_$C.err"; exit 1; }
  find "$OUT/pmc_$C" -name "*counter_collection.csv" -exec cp {} "$OUT/pmc_$C.csv" \;
done
echo "== pmc SQ"
rm -rf "$OUT/pmc_SQ"
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/pmc_SQ" -o bench -- python3 "$ROOT/bench.py" --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-c4 --no-configs > "$OUT/pmc_SQ.json" 2> "$OUT/pmc_SQ.err" || { tail -5 "$OUT/pmc_SQ.err"; exit 1; }
find "$OUT/pmc_SQ" -name "*counter_collection.csv" -exec cp {} "$OUT/pmc_SQ.csv" \;
echo "== all configs"; (cd "$ROOT" && timeout -k 10 400 python3 tools/bench_configs.py > "$OUT/all_configs.jsonl" 2> "$OUT/all_configs.err") || tail -3 "$OUT/all_configs.err"
echo "== per-block boundary (C++ twin of the Rust shim): the resident kernel, and a launch per call (KNH_RESIDENT=0)"
(cd "$ROOT" && for c in C3 C1; do timeout -k 10 120 tests/cpp/bin/shim_twin_test --bench $c 4096 batched > "$OUT/per_block_twin_$c.json" 2>&1; KNH_RESIDENT=0 timeout -k 10 120 tests/cpp/bin/shim_twin_test --bench $c 4096 batched > "$OUT/per_block_twin_${c}_launch_per_call.json" 2>&1; done; timeout -k 10 120 tests/cpp/bin/shim_twin_test --bench C3 4096 > "$OUT/per_block_twin_C3_single_call_events.json" 2>&1)
echo "== C5 kernel stats (the resolver kernels beside the voice kernel)"
rm -rf "$OUT/c5stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5stats" -o c5 -- python3 "$ROOT/tools/bench_configs.py" only C5 > "$OUT/c5_under_rocprof.json" 2> "$OUT/c5stats.err" || tail -3 "$OUT/c5stats.err"
find "$OUT/c5stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/c5_kernel_stats.csv" \;
rm -rf "$OUT/c5stats"
echo "== large banks (the two-groups-per-workgroup form against the forms of round 2)"
(cd "$ROOT" && for p in 0 2; do if [ $p = 2 ]; then unset KNH_PAIR; echo '{"KNH_PAIR": "default"}'; else export KNH_PAIR=$p; echo '{"KNH_PAIR": "0"}'; fi; timeout -k 10 200 python3 tools/bench_configs.py only C3:24576 C3:32768 C3:65536 C4:32768; done > "$OUT/pair_form.jsonl" 2>&1; unset KNH_PAIR)
echo "== micro"; (cd "$ROOT" && timeout -k 10 90 ./tools/micro/valu_issue > "$OUT/micro_valu_issue.txt" 2>&1; timeout -k 10 60 ./tools/micro/ring_lines > "$OUT/micro_ring_lines.txt" 2>&1)
rm -rf "$OUT"/stats "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE "$OUT"/pmc_SQ
ls -la "$OUT"
