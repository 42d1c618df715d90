# rocprofv3 recipe behind profiles/ (run from the repo root on the GPU box:  gpurun -- 'bash tools/prof_plonk_stats.sh'); output under gpurun_out/
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/plonk_stats
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/s -o s --output-format csv -- python3 $R/tools/plonk_bench.py 16 auto > $O/log.txt 2>&1
echo rc=$?
