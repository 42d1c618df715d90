export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/trace_fri
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/trace_fri/t -o t --output-format csv -- python3 $R/tools/fri_bench.py > $R/gpurun_out/trace_fri/log.txt 2>&1
cat $R/gpurun_out/trace_fri/log.txt | tail -5
