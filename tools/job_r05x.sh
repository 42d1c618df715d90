# experiment: every strided-pass workgroup also touches the input lines of the tile `lookahead` workgroups ahead (async global -> dummy LDS loads)
export ZKP_HIP_LIB=$PWD/zkp-implementation_amd/libzkp_variant_la.so
for la in 0 512 1024 2048 4096; do echo "== lookahead $la"; EXP_LOOKAHEAD=$la python tools/ab_ntt.py 22 24 26 | tail -1; done
echo "== lookahead 0 again"; EXP_LOOKAHEAD=0 python tools/ab_ntt.py 24 26 | tail -1
