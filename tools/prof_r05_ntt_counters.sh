# rocprofv3 recipe behind profiles/r05_ntt_*: the SQ counters of the Fr transform passes at 2^24 (dynamic instruction counts by unit,
# busy / wait cycles) -- run from the repo root on the GPU box:  gpurun -- 'bash tools/prof_r05_ntt_counters.sh'; output under gpurun_out/
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r05_ntt_pmc
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT/p1 -o p1 --output-format csv -- python3 $R/tools/ntt_bench.py fr 24 4 > $OUT/p1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/p2 -o p2 --output-format csv -- python3 $R/tools/ntt_bench.py fr 24 4 > $OUT/p2.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/p3 -o p3 --output-format csv -- python3 $R/tools/ntt_bench.py fr 24 4 > $OUT/p3.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_ANY -d $OUT/p4 -o p4 --output-format csv -- python3 $R/tools/ntt_bench.py fr 24 4 > $OUT/p4.log 2>&1
cd $R
python3 tools/pmc_summary.py $OUT | grep -E "ntt_pass" > $OUT/summary.txt
cat $OUT/summary.txt
