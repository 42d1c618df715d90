for lb in 8 9 10 8 9 10; do
  ZKP_SORT_LO_BITS=$lb timeout -k 10 200 python tools/ab_msm.py 20 || exit 1
done
for lb in 8 9 10; do
  ZKP_SORT_LO_BITS=$lb timeout -k 10 200 python tools/ab_msm.py 24 5 || exit 1
done
