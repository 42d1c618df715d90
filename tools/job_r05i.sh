# the Fr passes with every global access folded into a 1 MiB window (L2-resident; results wrong on purpose): what the transform costs
# when memory is free
echo "== product library"; python tools/ab_ntt.py 20 24 26 | tail -1
echo "== all global accesses inside 1 MiB"; ZKP_HIP_LIB=$PWD/zkp-implementation_amd/libzkp_variant_l2.so python tools/ab_ntt.py 20 24 26 | tail -1
echo "== product library again"; python tools/ab_ntt.py 24 | tail -1
