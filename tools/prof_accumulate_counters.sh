# rocprofv3 recipe behind profiles/ (run from the repo root on the GPU box:  gpurun -- 'bash tools/prof_accumulate_counters.sh'); output under gpurun_out/
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pmc_acc
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pmc_acc/p1 -o p1 --output-format csv -- python3 $R/tools/ab_msm.py 20 2 > $R/gpurun_out/pmc_acc/p1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD -d $R/gpurun_out/pmc_acc/p2 -o p2 --output-format csv -- python3 $R/tools/ab_msm.py 20 2 > $R/gpurun_out/pmc_acc/p2.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU -d $R/gpurun_out/pmc_acc/p3 -o p3 --output-format csv -- python3 $R/tools/ab_msm.py 20 2 > $R/gpurun_out/pmc_acc/p3.log 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_acc | grep -E "accumulate|pyramid_kernel|binsort|partscatter" > gpurun_out/pmc_acc/summary.txt
cat gpurun_out/pmc_acc/summary.txt
