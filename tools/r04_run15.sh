#!/bin/bash
# round 4, run 15: ring lines in the pipeline kernel too; batched headers; sink row
set -u
O=gpurun_out/r15; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ring_lines.py tests/test_gpu_parity.py -x -q -k "delay or allpass or ring" > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -3 $O/tests.txt
SIZES="16384 20480 32768 65536 131072 262144"
KNH_WIDE=0 timeout -k 10 300 python tools/bench_delay.py $SIZES > $O/delay_pipeline.jsonl 2> $O/delay_pipeline.err || exit 1
KNH_WIDE=4 timeout -k 10 300 python tools/bench_delay.py 20480 32768 65536 131072 > $O/delay_wide4.jsonl 2> $O/delay_wide4.err || exit 1
KNH_WIDE=8 timeout -k 10 300 python tools/bench_delay.py 65536 131072 262144 > $O/delay_wide8.jsonl 2> $O/delay_wide8.err || exit 1
for f in pipeline wide4 wide8; do echo == $f; python - <<PY
import json
for l in open("$O/delay_$f.jsonl"):
    d=json.loads(l); print(d["voices"], "%.3g UGen-samples/s  %.0f GB/s  %.1f us/block" % (d["kernel_only_ugen_samples_per_s"], d["algorithmic_GBps"], d["kernel_ms_per_launch"]*1e3/32))
PY
done
