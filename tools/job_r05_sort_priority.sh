# (needs the three-line patch that creates Ctx::sort_stream with hipStreamCreateWithPriority when ZKP_SORT_STREAM_PRIORITY is set; not kept)
# A/B: the digits + sort of the next scalar range on a HIGH-priority stream (ZKP_SORT_STREAM_PRIORITY) -- under an accumulate a sort gets only
# the slots draining workgroups leave and runs at a third of its speed (profiles/r05_o).  Output gpurun_out/r05_sort_priority.txt
out=gpurun_out/r05_sort_priority.txt
: > $out
run() { python tools/ab_msm.py $1 $3 2>/dev/null | tail -1 | cut -c1-330 | sed "s/^/[$2] /"; }
for i in 1 2; do
  run 26 "default priority, resident 2^26" 3 >> $out
  ZKP_SORT_STREAM_PRIORITY=1 run 26 "high priority, resident 2^26" 3 >> $out
done
for ln in 24 22; do
  reps=4; [ $ln -le 22 ] && reps=8
  for i in 1 2; do
    python3 tools/h2d_timeline.py $ln $reps 2>/dev/null | tail -1 | sed "s/^/[default priority] /" >> $out
    ZKP_SORT_STREAM_PRIORITY=1 python3 tools/h2d_timeline.py $ln $reps 2>/dev/null | tail -1 | sed "s/^/[high priority] /" >> $out
  done
done
cut -c1-250 $out
