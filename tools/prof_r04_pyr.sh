# rocprofv3 recipe behind profiles/r04_e (run from the repo root on the GPU box: gpurun -- 'bash tools/prof_r04_pyr.sh'); output under gpurun_out/r04pyr
# The bucket reduction of one 2^20 MSM (2^19 buckets): launch-by-launch timeline, then the issue / wait / memory counters of its level kernels.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04pyr
mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python3 $R/tools/ab_msm.py 20 3 > $O/trace.log 2>&1 || { echo "trace failed"; exit 1; }
echo "trace done"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_IFETCH SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python3 $R/tools/ab_msm.py 20 2 > $O/p$i.log 2>&1
    rc=$?
    echo "pyramid counters pass $i ($set) rc=$rc"
    if [ $rc -ne 0 ]; then echo "failed: stopping"; exit 1; fi
done
cd $R
python tools/summarize_prof.py timeline $(find $O/trace -name "*kernel_trace.csv" | head -1) msm_digits msm_pyramid_tail $O/timeline.md
python tools/summarize_prof.py pmc_by_grid msm_pyramid $(find $O/p* -name "*counter_collection.csv") $O/pyr_counters.md
python tools/summarize_prof.py pmc_by_grid msm_accumulate $(find $O/p* -name "*counter_collection.csv") $O/acc_counters.md
echo "summaries done"
