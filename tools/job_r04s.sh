# A/B: dedicated squaring (fq28_sqr_norm) in the XYZZ additions -- output under gpurun_out/r04s
mkdir -p gpurun_out/r04s
python -m pytest tests/test_gpu_parity.py tests/test_gpu_plonk.py -m gpu -x -q > gpurun_out/r04s/tests.log 2>&1 || { tail -30 gpurun_out/r04s/tests.log; exit 1; }
tail -2 gpurun_out/r04s/tests.log
OLD=$PWD/zkp-implementation_amd/libzkp_hip_nosqr.so
run() { python tools/ab_msm.py $1 $3 2>/dev/null | tail -1 | sed "s/^/[$2] /"; }
for ln in 20 24 16 12; do
  reps=30; [ $ln -ge 22 ] && reps=10
  for i in 1 2 3; do
    run $ln "squaring" $reps
    ZKP_HIP_LIB=$OLD run $ln "products only" $reps
  done
done > gpurun_out/r04s/ab.txt 2>&1
grep -o "^\[[a-z ]*\]\|n=2^[0-9]* [0-9.]* ms\|'msm_accumulate': [0-9.]*\|'msm_bucket_reduce': [0-9.]*\|'accumulate_kcycles': [0-9.]*" gpurun_out/r04s/ab.txt | paste - - - - -
pl() { python tools/plonk_bench.py 16 $1 2>/dev/null | tail -1 | python -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('$2', 'prove_ms %.3f' % d['prove_ms'], 'second-proof', d.get('generate_proof_ms_with_transcript'), {k: v['ms'] for k, v in d['phase_ms_one_proof'].items()})"; }
for i in 1 2 3; do
  pl auto "[squaring]"
  ZKP_HIP_LIB=$OLD pl auto "[products only]"
done > gpurun_out/r04s/ab_plonk.txt 2>&1
cat gpurun_out/r04s/ab_plonk.txt
