# HBM traffic of the Fr NTT passes at 2^24 (BASELINE configs[2]: "rocprof HBM GB/s reported"): FETCH_SIZE and WRITE_SIZE in separate passes
# gpurun -- 'bash tools/prof_r03_ntt_traffic.sh'; output under gpurun_out/r03ntt
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03ntt
mkdir -p $O
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c -d $O/$c -o p --output-format csv -- python3 $R/tools/ntt_bench.py fr 24 10 > $O/$c.log 2>&1
    rc=$?
    echo "$c rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit 1; fi
done
cd $R
python tools/summarize_prof.py pmc $(find $O/FETCH_SIZE $O/WRITE_SIZE -name "*counter_collection.csv") $O/pmc_ntt24.md
cat $O/pmc_ntt24.md
