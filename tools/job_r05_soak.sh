# randomised GPU soaks on the round's final binary (tests/soak/README.md)
python3 tests/soak/fuzz_msm.py 11 300 2>&1 | tail -1
python3 tests/soak/fuzz_msm.py 12 150 2>&1 | tail -1
python3 tests/soak/fuzz_ntt.py 11 200 2>&1 | tail -1
python3 tests/soak/fuzz_plonk.py 11 30 2>&1 | tail -1
python3 tests/soak/fuzz_fri.py 11 30 2>&1 | tail -1
