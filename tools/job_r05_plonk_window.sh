# PLONK 2^16: window width of the expanded SRS (automatic = 16 bits) swept again on the round-5 kernels -- output gpurun_out/r05_plonk_window.txt
out=gpurun_out/r05_plonk_window.txt
: > $out
for i in 1 2 3; do
  for w in auto 15 14 17; do
    echo "[window $w]" >> $out
    python tools/plonk_bench.py 16 $w 2>/dev/null | tail -1 | grep -o "'generate_proof_ms_with_transcript': [0-9.]*\|'msm_accumulate': {'ms': [0-9.]*\|'msm_bucket_reduce': {'ms': [0-9.]*\|'msm_sort': {'ms': [0-9.]*" | paste - - - - >> $out
  done
done
cat $out
