set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
echo "== bench quick"; timeout -k 10 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_quick.json 2> $O/bench_quick.err; echo rc=$?; tail -c 600 $O/bench_quick.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_c4
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_c4 -o c4 -- python3 $R/tools/bench_configs.py only C4:65536 C3:65536 C3:262144 > $O/pmc_c4.json 2> $O/pmc_c4.err; echo rc=$?
find $O/pmc_c4 -name "*counter_collection.csv" -exec cp {} $O/pmc_c4_sq.csv \; ; rm -rf $O/pmc_c4
cd $R
python tools/sq_summary_any.py $O/pmc_c4_sq.csv "voice_kernel<double, false, 4" 65536 512 32 "C4 65536 f64 voices, four whole-chain wavefronts per workgroup" > $O/c4_wide_sq.json
python tools/sq_summary_any.py $O/pmc_c4_sq.csv "voice_kernel<float, false, 4" 65536 512 32 "C3 65536 f32 voices, four whole-chain wavefronts per workgroup" > $O/c3_65536_sq.json
python tools/sq_summary_any.py $O/pmc_c4_sq.csv "voice_kernel<float, false, 8" 262144 512 32 "C3 262144 f32 voices, eight whole-chain wavefronts per workgroup" > $O/c3_262144_sq.json
cat $O/c4_wide_sq.json | head -30
true
