# PLONK 2^16: does splitting every bucket's run over four lanes (instead of two for the batches of three) balance the accumulate?
for v in default 2 1; do
  echo "== ZKP_MSM_SPLIT_LOG=$v"
  if [ $v = default ]; then python3 tools/plonk_bench.py 16 auto 2>&1 | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print(d['prove_ms'], d['generate_proof_ms_with_transcript'], d['round_ms'], d['phase_ms_one_proof']['msm_accumulate'], d['phase_ms_one_proof']['msm_bucket_reduce'])"
  else ZKP_MSM_SPLIT_LOG=$v python3 tools/plonk_bench.py 16 auto 2>&1 | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print(d['prove_ms'], d['generate_proof_ms_with_transcript'], d['round_ms'], d['phase_ms_one_proof']['msm_accumulate'], d['phase_ms_one_proof']['msm_bucket_reduce'])"; fi
done
