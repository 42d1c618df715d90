# longer randomised GPU soak on the round's last binary (new seeds): output gpurun_out/r05_soak3.log
{
python3 tests/soak/fuzz_msm.py 71 600 2>&1 | tail -1 &&
python3 tests/soak/fuzz_msm.py 72 600 2>&1 | tail -1 &&
python3 tests/soak/fuzz_ntt.py 71 400 2>&1 | tail -1 &&
python3 tests/soak/fuzz_plonk.py 71 80 2>&1 | tail -1 &&
python3 tests/soak/fuzz_fri.py 71 60 2>&1 | tail -1
} > gpurun_out/r05_soak3.log 2>&1
rc=$?
cat gpurun_out/r05_soak3.log
exit $rc
