# rocprofv3 recipe behind profiles/r02_* (run from the repo root on the GPU box:  gpurun -- 'bash tools/prof_r02.sh'); output under gpurun_out/r02prof
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02prof
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "bench done"
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats_msm -o s --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline > $O/bench_msm.json 2> $O/bench_msm.err || exit 1
echo "stats msm done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 $R/tools/ab_msm.py 20 2 > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 $R/tools/ab_msm.py 20 2 > $O/pmc_write.log 2>&1 || exit 1
echo "pmc 2^20 done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch24 -o p --output-format csv -- python3 $R/tools/ab_msm.py 24 1 > $O/pmc_fetch24.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write24 -o p --output-format csv -- python3 $R/tools/ab_msm.py 24 1 > $O/pmc_write24.log 2>&1 || exit 1
echo "pmc 2^24 done"
rocprofv3 --kernel-trace --stats -d $O/stats_ntt -o s --output-format csv -- python3 $R/tools/ntt_bench.py fr 24 10 > $O/ntt24.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -d $O/stats_fourstep -o s --output-format csv -- python3 $R/tools/four_step_local_bench.py 26 8 > $O/fourstep.log 2>&1 || exit 1
cd $R
for d in stats_msm stats_ntt stats_fourstep; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); python tools/summarize_prof.py stats $f $O/$d.md; done
python tools/summarize_prof.py pmc $(find $O/pmc_fetch $O/pmc_write -name "*counter_collection.csv") $O/pmc20.md
python tools/summarize_prof.py pmc $(find $O/pmc_fetch24 $O/pmc_write24 -name "*counter_collection.csv") $O/pmc24.md
echo "summaries done"
