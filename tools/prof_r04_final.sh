# rocprofv3 recipe behind profiles/r04_f_* (final state of round 4; run from the repo root on the GPU box: gpurun -- 'bash tools/prof_r04_final.sh'); output under gpurun_out/r04final
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04final
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "bench done"
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats_msm -o s --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline > $O/bench_msm.json 2> $O/bench_msm.err || exit 1
echo "stats msm done"
rocprofv3 --kernel-trace --stats -d $O/stats_ntt -o s --output-format csv -- python3 $R/tools/ntt_bench.py fr 24 10 > $O/ntt24.log 2>&1 || exit 1
echo "stats ntt done"
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python3 $R/tools/ab_msm.py 20 3 > $O/trace.log 2>&1 || exit 1
cd $R
for d in stats_msm stats_ntt; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); python tools/summarize_prof.py stats $f $O/$d.md; done
python tools/summarize_prof.py timeline $(find $O/trace -name "*kernel_trace.csv" | head -1) msm_digits msm_pyramid_tail $O/timeline.md
echo "summaries done"
