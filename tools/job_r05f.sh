# where the time of a Fr transform pass goes: the product library against timing-only builds (results wrong on purpose) without the
# field products, and without any tile work (load + store only)
echo "== product library"; python tools/ab_ntt.py 20 24 26 | tail -1
echo "== no field products (xor instead)"; ZKP_HIP_LIB=$PWD/zkp-implementation_amd/libzkp_variant_EXP_NOMUL.so python tools/ab_ntt.py 20 24 26 | tail -1
echo "== no tile work at all (load, convert, store)"; ZKP_HIP_LIB=$PWD/zkp-implementation_amd/libzkp_variant_EXP_NOMULDEXP_NOTILE.so python tools/ab_ntt.py 20 24 26 | tail -1
echo "== product library again"; python tools/ab_ntt.py 24 | tail -1
