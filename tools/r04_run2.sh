set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
echo "== changed tests"; timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_inputs.py tests/test_gpu_multi.py::test_eight_ranks_in_one_process_split_c4 tests/test_gpu_properties.py::test_full_size_c2_sin_numeric tests/test_shim_twin.py > $O/changed_tests.txt 2>&1; echo rc=$?; tail -5 $O/changed_tests.txt
echo "== wide stamps"
for spec in "C4 65536" "C3 65536" "C3 262144" "C4 8192 4"; do set -- $spec; KNH_LIB=$R/knaster_amd/csrc/libknaster_hip_stamps.so timeout -k 10 120 python tools/wide_stamps.py $1 $2 ${3:-} > $O/wide_stamps_$1_$2.json 2>&1; echo "$spec rc=$?"; done
KNH_LIB=$R/knaster_amd/csrc/libknaster_hip_stamps.so timeout -k 10 120 python tools/pipe_stamps.py C4 8192 > $O/pipe_stamps_C4_8192.txt 2>&1; echo rc=$?
KNH_LIB=$R/knaster_amd/csrc/libknaster_hip_stamps.so timeout -k 10 120 python tools/pipe_stamps.py C3 > $O/pipe_stamps_C3.txt 2>&1; echo rc=$?
true
