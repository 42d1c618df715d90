# A/B: the uploads of the later scalar ranges of zkp_msm_g1 issued by an uploader thread before the first range's kernels are enqueued
# (libzkp_variant_base.so = the commit before: issued in line after them); then the share of the first range swept again.
# Output gpurun_out/r05_uploader.txt
out=gpurun_out/r05_uploader.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi_slot.py -m gpu -x -q -k "msm or kzg or slot" > gpurun_out/r05_uploader_tests.log 2>&1 || { tail -30 gpurun_out/r05_uploader_tests.log; exit 1; }
tail -1 gpurun_out/r05_uploader_tests.log > $out
python3 tests/soak/fuzz_msm.py 41 120 2>&1 | tail -1 >> $out || { tail -5 $out; exit 1; }
V=$PWD/zkp-implementation_amd/libzkp_variant_base.so
for ln in 20 22 24; do
  reps=20; [ $ln -ge 22 ] && reps=8; [ $ln -ge 24 ] && reps=4
  for i in 1 2 3; do
    ZKP_HIP_LIB=$V python3 tools/h2d_timeline.py $ln $reps 2>/dev/null | tail -1 | sed "s/^/[in line, first 20 %] /" >> $out
    for pct in 20 25 30; do
      ZKP_MSM_FEED_FIRST_PCT=$pct python3 tools/h2d_timeline.py $ln $reps 2>/dev/null | tail -1 | sed "s/^/[uploader, first $pct %] /" >> $out
    done
  done
done
cat $out
