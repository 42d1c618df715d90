# rocprofv3 recipe behind profiles/ (run from the repo root on the GPU box:  gpurun -- 'bash tools/prof_msm_timeline.sh'); output under gpurun_out/
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/trace_msm
cd /tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/trace_msm/t -o t --output-format csv -- python3 $R/tools/ab_msm.py 20 3 > $R/gpurun_out/trace_msm/log.txt 2>&1
echo rc=$?
