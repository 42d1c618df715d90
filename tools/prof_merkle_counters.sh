# rocprofv3 recipe behind profiles/ (run from the repo root on the GPU box:  gpurun -- 'bash tools/prof_merkle_counters.sh'); output under gpurun_out/
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pmc_merkle
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_LDS -d $O/p1 -o p --output-format csv -- python3 $R/tools/fri_bench.py > $O/p1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/p2 -o p --output-format csv -- python3 $R/tools/fri_bench.py > $O/p2.log 2>&1
echo rc=$?
