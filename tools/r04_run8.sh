set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
for nv in 16384 8192 2048; do TWIN_VOICES=$nv timeout -k 10 120 tests/cpp/bin/shim_twin_test --bench C3 1024 batched > $O/twin_res_nv$nv.json 2>&1; cat $O/twin_res_nv$nv.json; done
true
