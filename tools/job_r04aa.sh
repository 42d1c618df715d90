# chunks of the counting sort (ZKP_MSM_NCHUNK; default 512 per bucket set) with the 147 KB first-pass tiles: one workgroup per CU -- output under gpurun_out/r04aa
mkdir -p gpurun_out/r04aa
run() { python tools/ab_msm.py $1 $3 2>/dev/null | tail -1 | sed "s/^/[$2] /"; }
for ln in 20 24; do
  reps=30; [ $ln -ge 22 ] && reps=10
  for i in 1 2; do
    for nc in 512 256 128 1024; do ZKP_MSM_NCHUNK=$nc run $ln "nchunk $nc" $reps; done
  done
done > gpurun_out/r04aa/ab.txt 2>&1
grep -o "^\[[a-z0-9= ]*\]\|n=2^[0-9]* [0-9.]* ms\|'msm_sort': [0-9.]*\|'msm_accumulate': [0-9.]*" gpurun_out/r04aa/ab.txt | paste - - - -
