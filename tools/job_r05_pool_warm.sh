# A/B: the host pool woken before the MSM's last kernel ends (ZKP_POOL_NO_WARM=1 = as before) -- PLONK 2^16 one-call proofs and batches of 2^16 MSMs
out=gpurun_out/r05_pool_warm.txt
python -m pytest tests/test_gpu_plonk.py tests/test_gpu_parity.py -m gpu -x -q -k "plonk or batch or commit" > gpurun_out/r05_pool_warm_tests.log 2>&1 || { tail -20 gpurun_out/r05_pool_warm_tests.log; exit 1; }
tail -1 gpurun_out/r05_pool_warm_tests.log > $out
for i in 1 2 3 4; do
  for v in 1 ""; do
    [ -n "$v" ] && export ZKP_POOL_NO_WARM=1 || unset ZKP_POOL_NO_WARM
    echo "[$([ -n "$v" ] && echo "cold pool" || echo "warm pool")]" >> $out
    python tools/plonk_bench.py 16 auto 2>/dev/null | tail -1 | grep -o "'generate_proof_ms_with_transcript': [0-9.]*\|'generate_proof_ms_with_transcript_median': [0-9.]*\|'prove_ms': [0-9.]*\|'msm_tail_host': {'ms': [0-9.]*" | paste - - - - >> $out
  done
done
unset ZKP_POOL_NO_WARM
for v in 1 ""; do
  [ -n "$v" ] && export ZKP_POOL_NO_WARM=1 || unset ZKP_POOL_NO_WARM
  echo "[small MSM batches, $([ -n "$v" ] && echo "cold pool" || echo "warm pool")]" >> $out
  python tools/small_msm_bench.py 2>/dev/null | tail -3 >> $out
done
cat $out
