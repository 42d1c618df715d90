# randomised GPU soaks after the clk_end fix and the lane fold: the MSM fuzz now also draws scalar ranges, a short first range, serialised
# ranges and profiling (tests/soak/fuzz_msm.py) -- output gpurun_out/r05_soak2.log
{
python3 tests/soak/fuzz_msm.py 21 300 2>&1 | tail -1 &&
python3 tests/soak/fuzz_msm.py 22 300 2>&1 | tail -1 &&
python3 tests/soak/fuzz_ntt.py 21 100 2>&1 | tail -1 &&
python3 tests/soak/fuzz_plonk.py 21 30 2>&1 | tail -1
} > gpurun_out/r05_soak2.log 2>&1
rc=$?
cat gpurun_out/r05_soak2.log
exit $rc
