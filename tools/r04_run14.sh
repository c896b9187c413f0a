#!/bin/bash
# round 4, run 14: which form for a delay chain at which size
set -u
O=gpurun_out/r14; mkdir -p $O
SIZES="20480 24576 32768 49152 65536 98304 131072"
timeout -k 10 300 python tools/bench_delay.py $SIZES > $O/delay_pipeline.jsonl 2> $O/delay_pipeline.err || exit 1
KNH_WIDE=4 timeout -k 10 300 python tools/bench_delay.py $SIZES > $O/delay_wide4.jsonl 2> $O/delay_wide4.err || exit 1
KNH_WIDE=8 timeout -k 10 300 python tools/bench_delay.py $SIZES > $O/delay_wide8.jsonl 2> $O/delay_wide8.err || exit 1
for f in pipeline wide4 wide8; do echo == $f; python - <<PY
import json
for l in open("$O/delay_$f.jsonl"):
    d=json.loads(l); print(d["voices"], "%.3g UGen-samples/s  %.0f GB/s  %.1f us/block" % (d["kernel_only_ugen_samples_per_s"], d["algorithmic_GBps"], d["kernel_ms_per_launch"]*1e3/32))
PY
done
