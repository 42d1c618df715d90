# rocprofv3 recipe behind profiles/r05_i_* / r05_h_* / r05_g_* (final state of round 5): run from the repo root on the GPU box
#   gpurun -- 'bash tools/prof_r05_final.sh';  output under gpurun_out/r05final;  then in the container: python tools/copy_r05_final_profiles.py
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05final
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "bench done"
python bench.py --in-process-leg --in-process-slots 8 --config4-log-n 26 > $O/in_process_8slots.json 2> $O/in_process.err || exit 1
echo "in-process leg done"
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats_msm -o s --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline > $O/bench_msm.json 2> $O/bench_msm.err || exit 1
echo "stats msm done"
rocprofv3 --kernel-trace --stats -d $O/stats_ntt -o s --output-format csv -- python3 $R/tools/ntt_bench.py fr 24 10 > $O/ntt24.log 2>&1 || exit 1
echo "stats ntt done"
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python3 $R/tools/ab_msm.py 20 3 > $O/trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/plonk -o s --output-format csv -- python3 $R/tools/plonk_bench.py 16 auto > $O/plonk.log 2>&1 || exit 1
echo "plonk trace done"
cd $R
for d in stats_msm stats_ntt; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); python tools/summarize_prof.py stats $f $O/$d.md; done
python tools/summarize_prof.py timeline $(find $O/trace -name "*kernel_trace.csv" | head -1) msm_digits msm_pyramid_tail $O/timeline.md
python3 tools/plonk_timeline.py $(find $O/plonk -name "*kernel_trace.csv" | head -1) $O/plonk_timeline.md > /dev/null
python3 tools/plonk_bench.py 16 auto > $O/plonk_plain.txt 2>&1
echo "summaries done"
bash tools/prof_r05_traffic.sh | tail -3
