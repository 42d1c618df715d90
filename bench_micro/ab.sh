for i in 1 2; do
  timeout -k 10 200 python tools/ab_msm.py 20 && ZKP_HIP_LIB=$PWD/zkp-implementation_amd/libzkp_3c.so timeout -k 10 200 python tools/ab_msm.py 20 || exit 1
done
