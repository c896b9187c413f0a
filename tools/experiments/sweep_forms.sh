# Kernel forms against each other beyond one voice group per CU (us per 512-frame block, kernel only):
#   pair  = two groups per workgroup, each its own pipeline (KNH_PAIR=1)      pipe = one group per workgroup, in rounds (KNH_WIDE=0)
#   w4/w8/w16 = whole-chain wavefronts, 4/8/16 groups per workgroup (KNH_WIDE)
# usage (on the GPU box): bash tools/experiments/sweep_forms.sh > gpurun_out/sweep_forms.txt
cd "$(dirname "$0")/../.."
for cfg in C3 C4; do for n in 20480 24576 32768 40960 49152 65536 98304 131072 262144; do
  line="$cfg voices=$n groups=$((n / 64)):"
  for form in pair pipe w4 w8 w16; do
    case $form in
      pair) envs="KNH_PAIR=1";;
      pipe) envs="KNH_PAIR=0 KNH_WIDE=0";;
      w4) envs="KNH_PAIR=0 KNH_WIDE=4";;
      w8) envs="KNH_PAIR=0 KNH_WIDE=8";;
      w16) envs="KNH_PAIR=0 KNH_WIDE=16";;
    esac
    if [ $cfg = C4 ] && [ $form = w16 ]; then continue; fi
    if [ $n -ge 131072 ] && { [ $form = pipe ] || [ $form = pair ]; }; then continue; fi
    r=$(env $envs timeout -k 5 120 python tools/bench_configs.py only $cfg:$n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['us_per_block_kernel'],2))" 2>/dev/null || echo "-")
    line="$line $form=$r"
  done
  echo "$line"
done; done
