set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
for lib in libknaster_hip.so libknaster_hip_kt64.so libknaster_hip.so libknaster_hip_kt64.so; do echo "== $lib"; KNH_LIB=$R/knaster_amd/csrc/$lib timeout -k 10 200 python tools/bench_configs.py only C4:65536 C3:65536 C4:32768 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print(d['config'],d['voices'],round(d['us_per_block_kernel'],2))
    except Exception: print(l.strip()[:200])"; done
true
