set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
for spec in "C4 65536" "C3 65536" "C3 262144"; do set -- $spec; KNH_LIB=$R/knaster_amd/csrc/libknaster_hip_stamps.so timeout -k 10 120 python tools/wide_stamps.py $1 $2 > $O/wide_stamps_bf_$1_$2.json 2>&1; echo "$spec rc=$?"; done
true
