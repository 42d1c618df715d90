# sort tiles adopted (binsort 16384, partscatter 12288 / 8192 / 4096 by partition count): parity incl. 23-bit windows, then ZKP_SORT_LO_BITS at 2^20 -- output under gpurun_out/r04y
mkdir -p gpurun_out/r04y
python -m pytest tests/test_gpu_parity.py tests/test_gpu_plonk.py -m gpu -x -q > gpurun_out/r04y/tests.log 2>&1 || { tail -30 gpurun_out/r04y/tests.log; exit 1; }
tail -2 gpurun_out/r04y/tests.log
timeout -k 10 300 python tests/soak/fuzz_msm.py 431 150 > gpurun_out/r04y/soak.log 2>&1; tail -1 gpurun_out/r04y/soak.log
run() { python tools/ab_msm.py $1 30 2>/dev/null | tail -1 | sed "s/^/[$2] /"; }
for ln in 20 18; do
  for i in 1 2; do
    run $ln "lo_bits default"
    ZKP_SORT_LO_BITS=10 run $ln "lo_bits 10"
    ZKP_SORT_LO_BITS=8 run $ln "lo_bits 8"
  done
done > gpurun_out/r04y/ab.txt 2>&1
grep -o "^\[[a-z0-9_ ]*\]\|n=2^[0-9]* [0-9.]* ms\|'msm_sort': [0-9.]*\|'msm_accumulate': [0-9.]*" gpurun_out/r04y/ab.txt | paste - - - -
