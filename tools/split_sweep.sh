# sweep of window width x run split for small MSMs and the PLONK prover (run on the GPU box from the repo root); output gpurun_out/split_sweep.txt
O=gpurun_out/split_sweep.txt
: > $O
for ln in 16; do
  for c in 14 15 16 17 18; do
    for s in 0 1 2; do
      echo "== n=2^$ln c=$c split_log=$s" >> $O
      ZKP_MSM_SPLIT_LOG=$s timeout -k 10 120 python tools/small_msm_bench.py $ln $c 2>/dev/null >> $O || exit 1
    done
  done
done
for c in 15 16 17 18; do
  for s in 0 1 2; do
    echo "== plonk 2^16 c=$c split_log=$s" >> $O
    ZKP_MSM_SPLIT_LOG=$s timeout -k 10 120 python tools/plonk_bench.py 16 $c 2>/dev/null | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1])
print(round(d['prove_ms'],3), round(d['generate_proof_ms_with_transcript'],3), d['round_ms'], {k:v['ms'] for k,v in d['phase_ms_one_proof'].items()}, d['verified_with_pairings'])
" >> $O || exit 1
  done
done
