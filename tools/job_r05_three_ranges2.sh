# finer sweep of the two short ranges of zkp_msm_g1 at 2^22 / 2^24 (after the 2^24 timeline: the last sort should start when the accumulate before it ends)
out=gpurun_out/r05_three_ranges2.txt
: > $out
for ln in 24 22; do
  reps=4; [ $ln -le 22 ] && reps=8
  for i in 1 2; do
    for cfg in "10 30" "10 20" "12 20" "12 25" "14 24" "14 18" "9 15"; do
      set -- $cfg
      ZKP_MSM_FEED_FIRST_PCT=$1 ZKP_MSM_FEED_SECOND_PCT=$2 python3 tools/h2d_timeline.py $ln $reps 2>/dev/null | tail -1 | sed "s/^/[first $1 % second $2 %] /" >> $out
    done
  done
done
cat $out
