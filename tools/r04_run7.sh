set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
echo "== twin gpu"; timeout -k 10 120 tests/cpp/bin/shim_twin_test --gpu > $O/shim_twin_gpu.txt 2>&1; rc=$?; echo rc=$rc; tail -8 $O/shim_twin_gpu.txt
echo "== resident tests (all)"; timeout -k 10 300 python -m pytest -q -m gpu tests/test_gpu_resident.py > $O/resident_tests.txt 2>&1; rc=$?; echo rc=$rc; grep -E "passed|failed|FAILED|Error|warn" $O/resident_tests.txt | head -20
echo "== per-block twin"; for c in C3 C1; do for m in "" batched; do timeout -k 10 120 tests/cpp/bin/shim_twin_test --bench $c 2048 $m > $O/per_block_twin_res_${c}_${m:-single}.json 2>&1; cat $O/per_block_twin_res_${c}_${m:-single}.json; done; done
KNH_RESIDENT=0 timeout -k 10 120 tests/cpp/bin/shim_twin_test --bench C3 2048 batched > $O/per_block_twin_launch_C3_batched.json 2>&1; cat $O/per_block_twin_launch_C3_batched.json
true
