# A/B: sort tiles -- binsort 16384 (ZKP_BS_TILE), partscatter 12288 (ZKP_PS_TILE), both -- output under gpurun_out/r04x
mkdir -p gpurun_out/r04x
D=$PWD/zkp-implementation_amd
for v in ps12 ps12bs16; do
  ZKP_HIP_LIB=$D/libzkp_hip_$v.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" > gpurun_out/r04x/tests_$v.log 2>&1 || { tail -30 gpurun_out/r04x/tests_$v.log; exit 1; }
  tail -1 gpurun_out/r04x/tests_$v.log
done
run() { python tools/ab_msm.py $1 $3 2>/dev/null | tail -1 | sed "s/^/[$2] /"; }
for ln in 20 22 24 26 16; do
  reps=30; [ $ln -ge 22 ] && reps=10; [ $ln -ge 26 ] && reps=3
  for i in 1 2; do
    run $ln "default" $reps
    ZKP_HIP_LIB=$D/libzkp_hip_bs16.so run $ln "bs16" $reps
    ZKP_HIP_LIB=$D/libzkp_hip_ps12.so run $ln "ps12" $reps
    ZKP_HIP_LIB=$D/libzkp_hip_ps12bs16.so run $ln "ps12bs16" $reps
  done
done > gpurun_out/r04x/ab.txt 2>&1
grep -o "^\[[a-z0-9 ]*\]\|n=2^[0-9]* [0-9.]* ms\|'msm_sort': [0-9.]*\|'msm_accumulate': [0-9.]*" gpurun_out/r04x/ab.txt | paste - - - -
