#!/bin/bash
# round 4, run 16: range events: resident tests, the twin, the per-block numbers
set -u
O=gpurun_out/r16; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_resident.py tests/test_shim_twin.py -x -q > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -3 $O/tests.txt
for i in 1 2; do
timeout -k 10 120 tests/cpp/bin/shim_twin_test --bench C3 4096 batched > $O/twin_c3_$i.json 2>&1 || exit 1
done
timeout -k 10 120 tests/cpp/bin/shim_twin_test --bench C3 4096 > $O/twin_c3_single.json 2>&1 || exit 1
KNH_RESIDENT=0 timeout -k 10 120 tests/cpp/bin/shim_twin_test --bench C3 4096 batched > $O/twin_c3_launch.json 2>&1 || exit 1
cat $O/twin_c3_*.json
