# PLONK 2^16: fresh launch timeline of one zkp_plonk_prove + plain timing
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05l_plonk
mkdir -p $O
python3 tools/plonk_bench.py 16 auto > $O/plain.txt 2>&1; tail -5 $O/plain.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/s -o s --output-format csv -- python3 $R/tools/plonk_bench.py 16 auto > $O/log.txt 2>&1
echo rc=$?
cd $R
python3 tools/plonk_timeline.py $(ls $O/s/*kernel_trace.csv | head -1) $O/timeline.md | head -40
