# pipeline (KNH_WIDE=0) against the 4- and 8-group kernels between one and four voice groups per CU
for cfg in C3 C4; do for n in 16384 24576 32768 49152 65536; do for wide in 0 4 8; do
  echo "== $cfg voices=$n KNH_WIDE=$wide"; KNH_WIDE=$wide python tools/bench_configs.py only $cfg:$n | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['us_per_block_kernel'],2),'us/block', '%.3g'%d['kernel_only_ugen_samples_per_s'])"
done; done; done
