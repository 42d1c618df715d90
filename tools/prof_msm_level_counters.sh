# rocprofv3 recipe behind profiles/ (run from the repo root on the GPU box:  gpurun -- 'bash tools/prof_msm_level_counters.sh'); output under gpurun_out/
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pmc_pyr
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/p1 -o p --output-format csv -- python3 $R/tools/ab_msm.py 20 2 > $O/p1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/p2 -o p --output-format csv -- python3 $R/tools/ab_msm.py 20 2 > $O/p2.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_IFETCH SQ_INSTS_SALU -d $O/p3 -o p --output-format csv -- python3 $R/tools/ab_msm.py 20 2 > $O/p3.log 2>&1 &&
rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $O/p4 -o p --output-format csv -- python3 $R/tools/ab_msm.py 20 2 > $O/p4.log 2>&1
echo rc=$?
