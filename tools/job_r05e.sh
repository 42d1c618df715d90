# A/B of the Fr transform with the butterflies' products as interleaved multiply-add chains (fr29_mul2, product library) against two
# plain C products per pair (-DZKP_FR29_PLAIN_C: rounds 1-4), same box, alternating
echo "== mul2 (product library)"; python tools/ab_ntt.py 16 18 20 22 24 26
echo "== plain C"; ZKP_HIP_LIB=$PWD/zkp-implementation_amd/libzkp_variant_plainc.so python tools/ab_ntt.py 16 18 20 22 24 26
echo "== mul2 again"; python tools/ab_ntt.py 24
echo "== plain C again"; ZKP_HIP_LIB=$PWD/zkp-implementation_amd/libzkp_variant_plainc.so python tools/ab_ntt.py 24
