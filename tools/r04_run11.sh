#!/bin/bash
# round 4, run 11: 32- against 64-sample visits at eight wavefronts per workgroup; packed-f32 micro-benchmarks; D3 as it stands
set -u
mkdir -p gpurun_out/r11
O=gpurun_out/r11
timeout -k 10 300 python tools/bench_configs.py only C3:131072 C3:262144 C3:65536 > $O/kt32.jsonl 2> $O/kt32.err || exit 1
KNH_LIB=$GRAFT_REPO_ROOT/knaster_amd/csrc/libknaster_hip_kt64.so timeout -k 10 300 python tools/bench_configs.py only C3:131072 C3:262144 C3:65536 > $O/kt64.jsonl 2> $O/kt64.err || exit 1
timeout -k 10 300 python tools/bench_configs.py only C3:131072 C3:262144 >> $O/kt32.jsonl 2>> $O/kt32.err || exit 1
KNH_LIB=$GRAFT_REPO_ROOT/knaster_amd/csrc/libknaster_hip_kt64.so timeout -k 10 300 python tools/bench_configs.py only C3:131072 C3:262144 >> $O/kt64.jsonl 2>> $O/kt64.err || exit 1
timeout -k 10 120 tools/micro/valu_issue > $O/valu_issue.txt 2>&1 || exit 1
timeout -k 10 120 tools/micro/pk_two_voices > $O/pk_two_voices.txt 2>&1 || exit 1
timeout -k 10 400 python tools/bench_delay.py 16384 65536 262144 > $O/delay.jsonl 2> $O/delay.err || exit 1
echo done
