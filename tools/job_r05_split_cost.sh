# what a second scalar range costs in the accumulate at 2^20 (resident scalars, no copies): one range / 20+80 / 50+50, overlapped and serialised
out=gpurun_out/r05_split_cost.txt
: > $out
run() { python tools/ab_msm.py 20 30 2>/dev/null | tail -1 | grep -o "n=2^20 [0-9.]* ms\|'msm_sort': [0-9.]*\|'msm_accumulate': [0-9.]*\|'accumulate_mhz': [0-9.]*\|'accumulate_kcycles': [0-9.]*" | paste - - - - - | sed "s/^/[$1] /"; }
for i in 1 2; do
  run "one range" >> $out
  ZKP_MSM_FIRST_PCT=20 run "20 + 80" >> $out
  ZKP_MSM_FIRST_PCT=20 ZKP_MSM_NO_OVERLAP=1 run "20 + 80 serial" >> $out
  ZKP_MSM_FIRST_PCT=50 ZKP_MSM_NO_OVERLAP=1 run "50 + 50 serial" >> $out
  ZKP_MSM_FIRST_PCT=80 ZKP_MSM_NO_OVERLAP=1 run "80 + 20 serial" >> $out
done
cat $out
