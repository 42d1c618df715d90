# Calibration of FETCH_SIZE for random 128-byte gathers: gpurun -- 'bash tools/prof_r03_gather.sh'; output under gpurun_out/r03gather
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03gather
mkdir -p $O
run() {  # run <label> <command...>: a pass that fails quickly (unknown counter) is skipped, one that was killed ends the script
    local label=$1; shift
    timeout -k 10 420 "$@" > $O/$label.log 2>&1
    local rc=$?
    echo "$label rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit 1; fi
}
cd /tmp
run plain $R/bench_micro/gather128 3
rocprofv3 -L > $O/counters.txt 2>&1
grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*UTCL[A-Z0-9_]*\|[A-Z0-9_]*UTCL2[A-Z0-9_]*" $O/counters.txt | sort -u > $O/counter_names.txt
run fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p_fetch -o g --output-format csv -- $R/bench_micro/gather128 3
run rdreq rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/p_rdreq -o g --output-format csv -- $R/bench_micro/gather128 3
run hitmiss rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/p_hit -o g --output-format csv -- $R/bench_micro/gather128 3
run bubble rocprofv3 --kernel-trace --pmc TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum -d $O/p_dram -o g --output-format csv -- $R/bench_micro/gather128 3
run utcl1 rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum -d $O/p_utcl1 -o g --output-format csv -- $R/bench_micro/gather128 3
cd $R
for d in p_fetch p_rdreq p_hit p_dram p_utcl1; do
    f=$(find $O/$d -name "*counter_collection.csv" 2>/dev/null | head -1)
    [ -n "$f" ] && python tools/summarize_prof.py pmc $f $O/$d.md
done
echo done
