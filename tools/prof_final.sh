# rocprofv3 recipe behind profiles/ (run from the repo root on the GPU box:  gpurun -- 'bash tools/prof_final.sh'); output under gpurun_out/
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/final
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "bench done" 
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats_default -o s --output-format csv -- python3 $R/bench.py > $O/bench_prof.json 2> $O/bench_prof.err || exit 1
echo "stats default done"
rocprofv3 --kernel-trace --stats -d $O/stats_msm -o s --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline > $O/bench_msm.json 2> $O/bench_msm.err || exit 1
echo "stats msm done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 $R/tools/ab_msm.py 20 2 > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 $R/tools/ab_msm.py 20 2 > $O/pmc_write.log 2>&1 || exit 1
echo "pmc done"
cd $R
python tools/grid_bench.py > $O/grid.md 2> $O/grid.err || exit 1
echo "grid done"
