#!/bin/bash
# Copies what tools/make_profiles.sh left under gpurun_out/profiles_new/ into profiles/ under this round's names.
#   bash tools/adopt_profiles.sh r03
set -e
R="${1:?round prefix, e.g. r03}"
cd "$(dirname "$0")/.."
N=gpurun_out/profiles_new
cp $N/bench_plain.json profiles/${R}_bench.json
cp $N/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
cp $N/bench_kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
cp $N/bench_kernel_trace_summary.json profiles/${R}_bench_kernel_trace_summary.json
cp $N/bench_headline_only_kernel_stats.csv profiles/${R}_bench_headline_only_kernel_stats.csv
cp $N/bench_headline_only_under_rocprof.json profiles/${R}_bench_headline_only_under_rocprof.json
cp $N/all_configs.jsonl profiles/${R}_all_configs.jsonl
cp $N/c5_kernel_stats.csv profiles/${R}_c5_kernel_stats.csv
cp $N/pair_form.jsonl profiles/${R}_pair_form.jsonl
for c in C3 C1; do
  l=$(echo $c | tr A-Z a-z)
  cp $N/per_block_twin_$c.json profiles/${R}_per_block_twin_$l.json
  cp $N/per_block_twin_${c}_launch_per_call.json profiles/${R}_per_block_twin_${l}_launch_per_call.json
done
cp $N/per_block_twin_C3_single_call_events.json profiles/${R}_per_block_twin_c3_single_call_events.json
python3 tools/pmc_summary.py $N > profiles/${R}_hbm_traffic.json
python3 tools/sq_summary.py $N/pmc_SQ.csv > profiles/${R}_sq_counters.json
git status --short profiles | head -30
