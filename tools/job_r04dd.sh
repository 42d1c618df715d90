# A/B: msm_pyramid_kernel (lane-per-add levels of the bucket reduction) at four waves per SIMD (128 VGPRs, 116 B of scratch) against three (146 VGPRs) -- output under gpurun_out/r04dd
mkdir -p gpurun_out/r04dd
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k msm > gpurun_out/r04dd/tests.log 2>&1 || { tail -30 gpurun_out/r04dd/tests.log; exit 1; }
tail -1 gpurun_out/r04dd/tests.log
OLD=$PWD/zkp-implementation_amd/libzkp_hip_base.so
run() { python tools/ab_msm.py $1 $3 2>/dev/null | tail -1 | sed "s/^/[$2] /"; }
for ln in 20 24 22; do
  reps=30; [ $ln -ge 22 ] && reps=10
  for i in 1 2 3; do
    run $ln "four waves" $reps
    ZKP_HIP_LIB=$OLD run $ln "three waves" $reps
  done
done > gpurun_out/r04dd/ab.txt 2>&1
grep -o "^\[[a-z ]*\]\|n=2^[0-9]* [0-9.]* ms\|'msm_bucket_reduce': [0-9.]*" gpurun_out/r04dd/ab.txt | paste - - -
