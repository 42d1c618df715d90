# A/B: binsort tile of 16384 entries (ZKP_BS_TILE) against the default 8192 -- output under gpurun_out/r04w
mkdir -p gpurun_out/r04w
V=$PWD/zkp-implementation_amd/libzkp_hip_bs16.so
ZKP_HIP_LIB=$V python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" > gpurun_out/r04w/tests.log 2>&1 || { tail -30 gpurun_out/r04w/tests.log; exit 1; }
tail -2 gpurun_out/r04w/tests.log
run() { python tools/ab_msm.py $1 $3 2>/dev/null | tail -1 | sed "s/^/[$2] /"; }
for ln in 20 22 24 16; do
  reps=30; [ $ln -ge 22 ] && reps=10
  for i in 1 2 3; do
    ZKP_HIP_LIB=$V run $ln "tile 16384" $reps
    run $ln "tile 8192" $reps
  done
done > gpurun_out/r04w/ab.txt 2>&1
grep -o "^\[[a-z0-9 ]*\]\|n=2^[0-9]* [0-9.]* ms\|'msm_sort': [0-9.]*\|'msm_accumulate': [0-9.]*" gpurun_out/r04w/ab.txt | paste - - - -
