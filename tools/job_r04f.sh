mkdir -p gpurun_out/r04f
python -m pytest tests/test_gpu_parity.py tests/test_gpu_plonk.py -m gpu -x -q > gpurun_out/r04f/tests.log 2>&1 || { tail -20 gpurun_out/r04f/tests.log; exit 1; }
tail -2 gpurun_out/r04f/tests.log
NOSOLO=$PWD/zkp-implementation_amd/libzkp_hip_nosolo.so
for ln in 20 16 12 24; do
  for i in 1 2 3; do
    python tools/ab_msm.py $ln 30 2>/dev/null | tail -1
    ZKP_HIP_LIB=$NOSOLO python tools/ab_msm.py $ln 30 2>/dev/null | tail -1
  done
done > gpurun_out/r04f/ab_solo.txt 2>&1
cat gpurun_out/r04f/ab_solo.txt | cut -c1-260
bash tools/prof_r04_pyr.sh > gpurun_out/r04pyr.log 2>&1; tail -12 gpurun_out/r04pyr.log
