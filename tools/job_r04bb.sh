# A/B: one accumulation chain per column in the Fq28 product (ZKP_FQ28_ONE_CHAIN) against the two chains of the product library, kernel pinned to three waves per SIMD -- output under gpurun_out/r04bb
mkdir -p gpurun_out/r04bb
V=$PWD/zkp-implementation_amd/libzkp_hip_onechain.so
ZKP_HIP_LIB=$V python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" > gpurun_out/r04bb/tests.log 2>&1 || { tail -30 gpurun_out/r04bb/tests.log; exit 1; }
tail -1 gpurun_out/r04bb/tests.log
run() { python tools/ab_msm.py $1 $3 2>/dev/null | tail -1 | sed "s/^/[$2] /"; }
for ln in 20 24 16; do
  reps=30; [ $ln -ge 22 ] && reps=10
  for i in 1 2 3; do
    ZKP_HIP_LIB=$V run $ln "one chain" $reps
    run $ln "two chains" $reps
  done
done > gpurun_out/r04bb/ab.txt 2>&1
grep -o "^\[[a-z ]*\]\|n=2^[0-9]* [0-9.]* ms\|'msm_accumulate': [0-9.]*\|'msm_bucket_reduce': [0-9.]*\|'accumulate_kcycles': [0-9.]*" gpurun_out/r04bb/ab.txt | paste - - - - -
