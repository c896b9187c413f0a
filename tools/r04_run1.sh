set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
echo "== doorbell"; timeout -k 10 120 tools/micro/doorbell > $O/doorbell.txt 2>&1; echo rc=$?; tail -12 $O/doorbell.txt
echo "== shim twin gpu"; timeout -k 10 300 tests/cpp/bin/shim_twin_test --gpu > $O/shim_twin_gpu.txt 2>&1; echo rc=$?; tail -8 $O/shim_twin_gpu.txt
echo "== changed tests"; timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_inputs.py tests/test_gpu_multi.py::test_eight_ranks_in_one_process_split_c4 tests/test_gpu_properties.py::test_full_size_c2_sin_numeric tests/test_shim_twin.py > $O/changed_tests.txt 2>&1; echo rc=$?; tail -15 $O/changed_tests.txt
echo "== wide stamps"
for spec in "C4 65536" "C3 65536" "C3 262144" "C4 8192 4"; do set -- $spec; KNH_LIB=$R/knaster_amd/csrc/libknaster_hip_stamps.so KNH_WIDE=${3:-} timeout -k 10 120 python tools/wide_stamps.py $1 $2 ${3:-} > $O/wide_stamps_$1_$2.json 2>&1; echo "$spec rc=$?"; done
KNH_LIB=$R/knaster_amd/csrc/libknaster_hip_stamps.so timeout -k 10 120 python tools/pipe_stamps.py C4 8192 > $O/pipe_stamps_C4_8192.txt 2>&1; echo rc=$?
echo "== per-block twin baseline"; for c in C3 C1; do timeout -k 10 120 tests/cpp/bin/shim_twin_test --bench $c 2048 > $O/per_block_twin_base_$c.json 2>&1; cat $O/per_block_twin_base_$c.json; done
echo "== SQ counters C4 wide"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_c4
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_c4 -o c4 -- python3 $R/tools/bench_configs.py only C4:65536 C3:65536 > $O/pmc_c4.json 2> $O/pmc_c4.err; echo rc=$?
find $O/pmc_c4 -name "*counter_collection.csv" -exec cp {} $O/pmc_c4_sq.csv \; ; rm -rf $O/pmc_c4
cd $R
echo "== doorbell BAR probe"; DOORBELL_TRY_BAR=1 timeout -k 10 120 tools/micro/doorbell > $O/doorbell_bar.txt 2>&1; echo rc=$?; tail -12 $O/doorbell_bar.txt
true
