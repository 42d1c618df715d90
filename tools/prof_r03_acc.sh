# msm_accumulate at 2^20 (20-bit windows) and 2^24 (22-bit): memory-side request counts and wave stall split (profiles/r03_*)
# gpurun -- 'bash tools/prof_r03_acc.sh'; output under gpurun_out/r03acc
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03acc
mkdir -p $O
cd /tmp
pass() {  # pass <label> <log_n> <counters...>
    local label=$1 ln=$2; shift; shift
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" -d $O/$label -o p --output-format csv -- python3 $R/tools/ab_msm.py $ln 2 > $O/$label.log 2>&1
    local rc=$?
    echo "$label rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit 1; fi
}
for ln in 20 24; do
    pass ea_$ln $ln TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
    pass tcc_$ln $ln TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
    pass tcp_$ln $ln TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum
    pass utcl_$ln $ln TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
    pass sq_$ln $ln SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
done
cd $R
for ln in 20 24; do
    python tools/summarize_prof.py pmc $(find $O/ea_$ln $O/tcc_$ln $O/tcp_$ln $O/utcl_$ln $O/sq_$ln -name "*counter_collection.csv") $O/acc_$ln.md
done
echo done
