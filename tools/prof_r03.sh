# rocprofv3 recipe behind profiles/r03_h / r03_i (run from the repo root on the GPU box: gpurun -- 'bash tools/prof_r03.sh'); output under gpurun_out/r03prof
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03prof
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "bench done"
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats_msm -o s --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline > $O/bench_msm.json 2> $O/bench_msm.err || exit 1
echo "stats msm done"
rocprofv3 --kernel-trace --stats -d $O/stats_ntt -o s --output-format csv -- python3 $R/tools/ntt_bench.py fr 24 10 > $O/ntt24.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -d $O/stats_fourstep -o s --output-format csv -- python3 $R/tools/four_step_local_bench.py 26 8 > $O/fourstep.log 2>&1 || exit 1
cd $R
for d in stats_msm stats_ntt stats_fourstep; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); python tools/summarize_prof.py stats $f $O/$d.md; done
echo "summaries done"
