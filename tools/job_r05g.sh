# do the workgroups that share a CU run their load / compute / store phases in lock step?  first-generation workgroups delayed by 0..3 x N x 3.5 us
echo "== product library"; python tools/ab_ntt.py 20 24 26 | tail -1
for v in 1 2; do echo "== stagger $v"; ZKP_HIP_LIB=$PWD/zkp-implementation_amd/libzkp_variant_stagger$v.so python tools/ab_ntt.py 20 24 26 | tail -1; done
echo "== product library again"; python tools/ab_ntt.py 24 | tail -1
