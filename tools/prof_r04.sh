# rocprofv3 recipe behind profiles/r04_* (run from the repo root on the GPU box: gpurun -- 'bash tools/prof_r04.sh'); output under gpurun_out/r04prof
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04prof
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "bench done"
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats_msm -o s --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline > $O/bench_msm.json 2> $O/bench_msm.err || exit 1
echo "stats msm done"
rocprofv3 --kernel-trace --stats -d $O/stats_ntt -o s --output-format csv -- python3 $R/tools/ntt_bench.py fr 24 10 > $O/ntt24.log 2>&1 || exit 1
echo "stats ntt done"
# HBM traffic of msm_accumulate (now with the clock stamps and one more argument), FETCH_SIZE and WRITE_SIZE in separate passes
for ln in 20 24; do
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $O/${c}_$ln -o p --output-format csv -- python3 $R/tools/ab_msm.py $ln 1 > $O/${c}_$ln.log 2>&1
        rc=$?
        echo "$c $ln rc=$rc"
        if [ $rc -ne 0 ]; then echo "failed: stopping"; exit 1; fi
    done
done
# the bucket reduction's level kernels: issue, wait, instruction-fetch and memory counters (one set per pass)
rocprofv3 -L > $O/counters_available.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $O/pyr_p$i -o p --output-format csv -- python3 $R/tools/ab_msm.py 20 2 > $O/pyr_p$i.log 2>&1
    echo "pyramid counters pass $i ($set) rc=$?"
done
cd $R
for d in stats_msm stats_ntt; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); python tools/summarize_prof.py stats $f $O/$d.md; done
for ln in 20 24; do
    python tools/summarize_prof.py pmc $(find $O/FETCH_SIZE_$ln $O/WRITE_SIZE_$ln -name "*counter_collection.csv") $O/pmc_$ln.md
done
python tools/summarize_prof.py pmc $(find $O/pyr_p* -name "*counter_collection.csv") $O/pyr_counters.md
echo "summaries done"
