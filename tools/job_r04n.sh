# host-scalar MSM (zkp_msm_g1, PCIe-inclusive): share of the scalars in the first of the two upload ranges -- output under gpurun_out/r04n
mkdir -p gpurun_out/r04n
python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r04n/tests.log 2>&1 || { tail -30 gpurun_out/r04n/tests.log; exit 1; }
tail -2 gpurun_out/r04n/tests.log
for i in 1 2; do
  for pct in 25 0 10 15 20 30 35 40; do
    ZKP_MSM_FEED_FIRST_PCT=$pct python tools/pcie_rate.py 20 2>/dev/null | grep -v amdgpu | sed "s/^/[first $pct %] /"
  done
  ZKP_MSM_FEED_RANGES=1 python tools/pcie_rate.py 20 2>/dev/null | grep "host scalars" | sed "s/^/[one range] /"
done > gpurun_out/r04n/pcie.txt 2>&1
grep "host scalars through\|phases" gpurun_out/r04n/pcie.txt | cut -c1-250
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['extra']['msm_h2d_inclusive'])"
