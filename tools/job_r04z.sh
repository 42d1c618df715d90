# window width at 2^21 .. 2^23 with this round's kernels: 20-bit (13 slices, 2^19 buckets) against 22-bit (12 slices, 2^21 buckets) -- output under gpurun_out/r04z
mkdir -p gpurun_out/r04z
run() { python tools/ab_msm.py $1 10 2>/dev/null | tail -1 | sed "s/^/[$2] /"; }
for ln in 21 22 23; do
  for i in 1 2; do
    ZKP_AB_C=20 run $ln "c=20"
    ZKP_AB_C=22 run $ln "c=22"
  done
done > gpurun_out/r04z/ab.txt 2>&1
grep -o "^\[[a-z0-9= ]*\]\|n=2^[0-9]* [0-9.]* ms\|'msm_sort': [0-9.]*\|'msm_accumulate': [0-9.]*\|'msm_bucket_reduce': [0-9.]*" gpurun_out/r04z/ab.txt | paste - - - - -
