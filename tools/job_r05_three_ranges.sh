# zkp_msm_g1 (host scalars): a second short range before the rest (ZKP_MSM_FEED_FIRST_PCT / ZKP_MSM_FEED_SECOND_PCT) against 25 % + 75 %
# Output gpurun_out/r05_three_ranges.txt
out=gpurun_out/r05_three_ranges.txt
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" > gpurun_out/r05_three_ranges_tests.log 2>&1 || { tail -30 gpurun_out/r05_three_ranges_tests.log; exit 1; }
tail -1 gpurun_out/r05_three_ranges_tests.log > $out
ZKP_MSM_FEED_FIRST_PCT=6 ZKP_MSM_FEED_SECOND_PCT=24 python3 tests/soak/fuzz_msm.py 51 80 2>&1 | tail -1 >> $out || { tail -5 $out; exit 1; }
for ln in 20 22 24; do
  reps=20; [ $ln -ge 22 ] && reps=8; [ $ln -ge 24 ] && reps=4
  for i in 1 2; do
    for cfg in "25 0" "5 25" "8 27" "10 30" "6 20" "12 33" "15 0"; do
      set -- $cfg
      ZKP_MSM_FEED_FIRST_PCT=$1 ZKP_MSM_FEED_SECOND_PCT=$2 python3 tools/h2d_timeline.py $ln $reps 2>/dev/null | tail -1 | sed "s/^/[first $1 % second $2 %] /" >> $out
    done
  done
done
cat $out
