# launch + copy timeline of one host-scalar MSM at 2^LOG (default 20) -- output gpurun_out/r05_h2d_timeline.md
LOG=${1:-20}
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r05_h2d
rm -rf $O; mkdir -p $O
python3 tools/h2d_timeline.py $LOG 8 > $O/plain.txt 2>&1 || { cat $O/plain.txt; exit 1; }
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $O/trace -o t --output-format csv -- python3 $R/tools/h2d_timeline.py $LOG 3 > $O/traced.txt 2>&1 || { tail -5 $O/traced.txt; exit 1; }
cd $R
{ cat $O/plain.txt | tail -1; tail -1 $O/traced.txt; python3 tools/h2d_timeline.py --summarise $O/trace; } > gpurun_out/r05_h2d_timeline.md
rm -rf $O/trace
cat gpurun_out/r05_h2d_timeline.md
