# (1) soak of the NTT / MSM / PLONK paths on this round's binary; (2) experiment: 8-column tiles (256-byte runs, 72 KB of LDS, two
# workgroups per CU) for pass 0 / pass 1 / the last pass of the Fr transform -- EXP_T3 bit p = strided pass p, bit 3 = last pass
python3 tests/soak/fuzz_ntt.py 5 160 2>&1 | tail -2
python3 tests/soak/fuzz_msm.py 5 120 2>&1 | tail -2
python3 tests/soak/fuzz_plonk.py 5 12 2>&1 | tail -2
export ZKP_HIP_LIB=$PWD/zkp-implementation_amd/libzkp_variant_t3.so
for m in 0 1 3 8 9 11; do echo "== EXP_T3=$m"; EXP_T3=$m python tools/ab_ntt.py 22 24 26 | tail -1; done
