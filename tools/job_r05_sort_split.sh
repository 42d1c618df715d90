# (apply tools/patches/r05_sort_split_entries.patch first; libzkp_variant_base.so = the unpatched build)
# A/B: sort entries as two arrays (4-byte index|sign + 2-byte low bits) against one 8-byte record (libzkp_variant_base.so = the commit before)
# parity first, then alternating runs on one box -- output gpurun_out/r05_sort_split.txt
out=gpurun_out/r05_sort_split.txt
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" > gpurun_out/r05_sort_split_tests.log 2>&1 || { tail -30 gpurun_out/r05_sort_split_tests.log; exit 1; }
tail -1 gpurun_out/r05_sort_split_tests.log > $out
V=$PWD/zkp-implementation_amd/libzkp_variant_base.so
run() { python tools/ab_msm.py $1 $3 2>/dev/null | tail -1 | sed "s/^/[$2] /"; }
for ln in 20 22 24 16 26; do
  reps=30; [ $ln -ge 22 ] && reps=10; [ $ln -ge 26 ] && reps=3
  for i in 1 2; do
    ZKP_HIP_LIB=$V run $ln "record 8 B" $reps
    run $ln "split 4+2 B" $reps
  done
done >> $out 2>&1
grep -o "^\[[a-zA-Z0-9 +]*\]\|n=2^[0-9]* [0-9.]* ms\|'msm_sort': [0-9.]*\|'msm_accumulate': [0-9.]*\|passed.*" $out | paste - - - -
