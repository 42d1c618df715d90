# A/B: a short first scalar range for RESIDENT scalars (ZKP_MSM_FIRST_PCT), so that only the first range's digits + sort are exposed and the
# rest sorts under the first accumulate -- output gpurun_out/r05_first_range.txt
run() { python tools/ab_msm.py $1 $3 2>/dev/null | tail -1 | sed "s/^/[$2] /"; }
for ln in 24 22 20; do
  reps=10; [ $ln -le 20 ] && reps=30
  for i in 1 2; do
    run $ln "one range" $reps
    for pct in 6 12 25 50; do ZKP_MSM_FIRST_PCT=$pct run $ln "first $pct%" $reps; done
  done
done > gpurun_out/r05_first_range.txt 2>&1
grep -o "^\[[a-z0-9 %]*\]\|n=2^[0-9]* [0-9.]* ms\|'msm_sort': [0-9.]*\|'msm_accumulate': [0-9.]*" gpurun_out/r05_first_range.txt | paste - - - -
