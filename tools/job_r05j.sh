# persistent, software-pipelined Fr passes (product library) against the round-4 structure with the round-5 product (variant) -- parity first
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ntt or poly_mul or four_step or fri" 2>&1 | tail -3
echo "== persistent + pipelined (product library)"; python tools/ab_ntt.py 12 16 18 20 22 24 26 | tail -1
echo "== round-4 structure, plain C product"; ZKP_HIP_LIB=$PWD/zkp-implementation_amd/libzkp_variant_plainc.so python tools/ab_ntt.py 12 16 18 20 22 24 26 | tail -1
echo "== persistent again"; python tools/ab_ntt.py 24 26 | tail -1
python tools/ntt_bench.py gl 24 20; python tools/ntt_bench.py gl 26 10; python tools/ntt_bench.py gl 20 50
ZKP_HIP_LIB=$PWD/zkp-implementation_amd/libzkp_variant_plainc.so python tools/ntt_bench.py gl 24 20; ZKP_HIP_LIB=$PWD/zkp-implementation_amd/libzkp_variant_plainc.so python tools/ntt_bench.py gl 26 10; ZKP_HIP_LIB=$PWD/zkp-implementation_amd/libzkp_variant_plainc.so python tools/ntt_bench.py gl 20 50
