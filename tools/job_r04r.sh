# A/B: second insertion with the affine + affine formula in both accumulate kernels (lane: g1_28_mmadd peel; quad: first product round skipped) -- output under gpurun_out/r04r
mkdir -p gpurun_out/r04r
python -m pytest tests/test_gpu_parity.py tests/test_gpu_plonk.py -m gpu -x -q > gpurun_out/r04r/tests.log 2>&1 || { tail -30 gpurun_out/r04r/tests.log; exit 1; }
tail -2 gpurun_out/r04r/tests.log
OLD=$PWD/zkp-implementation_amd/libzkp_hip_base.so
run() { python tools/ab_msm.py $1 $3 2>/dev/null | tail -1 | sed "s/^/[$2] /"; }
for ln in 12 14 16 20; do
  for i in 1 2 3; do
    run $ln "affine second insertion" 30
    ZKP_HIP_LIB=$OLD run $ln "base" 30
  done
done > gpurun_out/r04r/ab.txt 2>&1
grep -o "^\[[a-z ]*\]\|n=2^[0-9]* [0-9.]* ms\|'msm_accumulate': [0-9.]*\|'accumulate_kcycles': [0-9.]*" gpurun_out/r04r/ab.txt | paste - - - -
pl() { python tools/plonk_bench.py 16 $1 2>/dev/null | tail -1 | python -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('$2', 'prove_ms %.3f' % d['prove_ms'], 'second-proof', d.get('generate_proof_ms_with_transcript'), {k: v['ms'] for k, v in d['phase_ms_one_proof'].items()})"; }
for i in 1 2 3; do
  pl auto "[affine second insertion]"
  ZKP_HIP_LIB=$OLD pl auto "[base]"
done > gpurun_out/r04r/ab_plonk.txt 2>&1
cat gpurun_out/r04r/ab_plonk.txt
python tools/small_msm_bench.py 2>/dev/null | grep batch | sed "s/^/[new] /"
ZKP_HIP_LIB=$OLD python tools/small_msm_bench.py 2>/dev/null | grep batch | sed "s/^/[base] /"
