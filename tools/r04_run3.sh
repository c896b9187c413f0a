set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
echo "== full gpu suite"; timeout -k 10 900 python -m pytest -x -q -m gpu tests > $O/gpu_suite.txt 2>&1; echo rc=$?; tail -12 $O/gpu_suite.txt
echo "== wide configs"; timeout -k 10 300 python tools/bench_configs.py only C3:65536 C3:131072 C3:262144 C4:65536 C4:32768 C1 > $O/wide_configs_butterfly.jsonl 2>&1; cat $O/wide_configs_butterfly.jsonl
true
