# A/B: bucket sums handed from one scalar range to the next in an entry-major array (256 contiguous bytes per bucket) instead of through the
# plane-major bucket array (libzkp_variant_base.so = the commit before).  Parity first, then alternating runs on one box.
# Output gpurun_out/r05_range_handover.txt
out=gpurun_out/r05_range_handover.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi_slot.py -m gpu -x -q -k "msm or kzg or slot" > gpurun_out/r05_range_handover_tests.log 2>&1 || { tail -30 gpurun_out/r05_range_handover_tests.log; exit 1; }
tail -1 gpurun_out/r05_range_handover_tests.log > $out
python3 tests/soak/fuzz_msm.py 31 150 2>&1 | tail -1 >> $out || { tail -5 $out; exit 1; }
V=$PWD/zkp-implementation_amd/libzkp_variant_base.so
for ln in 20 22 24; do
  reps=20; [ $ln -ge 22 ] && reps=8; [ $ln -ge 24 ] && reps=4
  for i in 1 2 3; do
    ZKP_HIP_LIB=$V python3 tools/h2d_timeline.py $ln $reps 2>/dev/null | tail -1 | sed "s/^/[plane-major hand-over] /" >> $out
    python3 tools/h2d_timeline.py $ln $reps 2>/dev/null | tail -1 | sed "s/^/[entry-major hand-over] /" >> $out
  done
done
run() { python tools/ab_msm.py $1 $3 2>/dev/null | tail -1 | cut -c1-330 | sed "s/^/[$2] /"; }
for i in 1 2; do
  ZKP_HIP_LIB=$V run 26 "plane-major, resident 2^26" 3 >> $out
  run 26 "entry-major, resident 2^26" 3 >> $out
done
ZKP_HIP_LIB=$V run 20 "plane-major, resident 2^20 one range" 30 >> $out
run 20 "entry-major, resident 2^20 one range" 30 >> $out
cut -c1-260 $out
