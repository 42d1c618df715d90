# msm_accumulate with the six plain products of an insertion as strict multiply-add chains in asm (product library) against the
# compiler's schedule (-DZKP_ACC_PLAIN_PRODUCTS), alternating on one box; parity first
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "msm" 2>&1 | tail -2
for rep in 1 2; do
  echo "== chains (product library)"; python tools/ab_msm.py 20 3 | tail -2; python tools/ab_msm.py 24 2 | tail -1
  echo "== compiler's schedule"; ZKP_HIP_LIB=$PWD/zkp-implementation_amd/libzkp_variant_accplain.so python tools/ab_msm.py 20 3 | tail -2; ZKP_HIP_LIB=$PWD/zkp-implementation_amd/libzkp_variant_accplain.so python tools/ab_msm.py 24 2 | tail -1
done
