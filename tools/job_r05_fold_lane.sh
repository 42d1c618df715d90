# A/B: the fold of split bucket parts with one lane per add (msm_fold_parts_lane_kernel) from N adds per launch on (ZKP_FOLD_LANE_MIN),
# against four lanes per add everywhere.  Parity first (every fold step on the new kernel), then PLONK 2^16 and small MSM batches.
# Output gpurun_out/r05_fold_lane.txt
out=gpurun_out/r05_fold_lane.txt
: > $out
ZKP_FOLD_LANE_MIN=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plonk.py -m gpu -x -q -k "msm or plonk or kzg or commit" > gpurun_out/r05_fold_lane_tests.log 2>&1 || { tail -30 gpurun_out/r05_fold_lane_tests.log; exit 1; }
tail -2 gpurun_out/r05_fold_lane_tests.log >> $out
for i in 1 2 3; do
  for v in 4000000000 150000 65536 1; do
    echo "[lane from $v]" >> $out
    ZKP_FOLD_LANE_MIN=$v python tools/plonk_bench.py 16 auto 2>/dev/null | tail -1 | grep -o "'generate_proof_ms_with_transcript': [0-9.]*\|'generate_proof_ms': [0-9.]*\|'msm_accumulate': {'ms': [0-9.]*\|'msm_bucket_reduce': {'ms': [0-9.]*" | paste - - - - >> $out
  done
done
for v in 4000000000 65536 1; do
  echo "[small MSMs, lane from $v]" >> $out
  ZKP_FOLD_LANE_MIN=$v python tools/small_msm_bench.py 2>/dev/null | tail -12 >> $out
done
cat $out
