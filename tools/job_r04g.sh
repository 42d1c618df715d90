# A/B of the bucket reduction's last-levels launch (single-wave workgroups, DPP quad moves) -- output under gpurun_out/r04g
mkdir -p gpurun_out/r04g
python -m pytest tests/test_gpu_parity.py tests/test_gpu_plonk.py -m gpu -x -q > gpurun_out/r04g/tests.log 2>&1 || { tail -30 gpurun_out/r04g/tests.log; exit 1; }
tail -2 gpurun_out/r04g/tests.log
OLD=$PWD/zkp-implementation_amd/libzkp_hip_nosolo.so
run() { python tools/ab_msm.py $1 30 2>/dev/null | tail -1 | sed "s/^/[$2] /"; }
for ln in 20 16; do
  for i in 1 2; do
    run $ln "new 64x64"
    ZKP_HIP_LIB=$OLD run $ln "round-3 tail, ds_bpermute"
    ZKP_PYR_TAIL_THREADS=512 ZKP_PYR_TAIL_BLOCKS=8 run $ln "512x8 + DPP"
    ZKP_PYR_TAIL_THREADS=64 ZKP_PYR_TAIL_BLOCKS=32 run $ln "64x32"
    ZKP_PYR_TAIL_THREADS=64 ZKP_PYR_TAIL_BLOCKS=128 run $ln "64x128"
    ZKP_PYR_TAIL_THREADS=128 ZKP_PYR_TAIL_BLOCKS=32 run $ln "128x32"
    ZKP_PYR_TAIL_THREADS=256 ZKP_PYR_TAIL_BLOCKS=16 run $ln "256x16"
    ZKP_PYR_TAIL_HALF=128 ZKP_PYR_TAIL_BLOCKS=128 run $ln "64x128 from 128 pairs"
    ZKP_PYR_TAIL_HALF=256 ZKP_PYR_TAIL_BLOCKS=256 run $ln "64x256 from 256 pairs"
    ZKP_PYR_TAIL_HALF=32 run $ln "64x64 from 32 pairs"
    ZKP_PYR_TAIL_HALF=1 run $ln "no tail launch (every level its own launch)"
  done
done > gpurun_out/r04g/ab_tail.txt 2>&1
cut -c1-200 gpurun_out/r04g/ab_tail.txt
for i in 1 2; do
  python tools/plonk_bench.py 16 2>/dev/null | tail -1 | cut -c1-600 | sed "s/^/[new] /"
  ZKP_HIP_LIB=$OLD python tools/plonk_bench.py 16 2>/dev/null | tail -1 | cut -c1-600 | sed "s/^/[round-3 tail] /"
done > gpurun_out/r04g/ab_plonk.txt 2>&1
cat gpurun_out/r04g/ab_plonk.txt
