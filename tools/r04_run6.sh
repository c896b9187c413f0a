set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
echo "== census"; timeout -k 10 60 tools/micro/coresidency > $O/coresidency.txt 2>&1; echo rc=$?; cat $O/coresidency.txt
echo "== twin gpu"; timeout -k 10 120 tests/cpp/bin/shim_twin_test --gpu > $O/shim_twin_gpu.txt 2>&1; rc=$?; echo rc=$rc; tail -12 $O/shim_twin_gpu.txt
echo "== resident tests (all)"; timeout -k 10 300 python -m pytest -q -m gpu tests/test_gpu_resident.py > $O/resident_tests.txt 2>&1; rc=$?; echo rc=$rc; grep -E "passed|failed|FAILED|Error" $O/resident_tests.txt | head -20
true
