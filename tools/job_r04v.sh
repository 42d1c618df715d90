# A/B: exceptional cases of the cooperative (four lanes per add) XYZZ addition without the inlined scalar add (93-141 VGPRs instead of 198-223) -- output under gpurun_out/r04v
mkdir -p gpurun_out/r04v
python -m pytest tests/test_gpu_parity.py tests/test_gpu_plonk.py -m gpu -x -q > gpurun_out/r04v/tests.log 2>&1 || { tail -30 gpurun_out/r04v/tests.log; exit 1; }
tail -2 gpurun_out/r04v/tests.log
OLD=$PWD/zkp-implementation_amd/libzkp_hip_base.so
run() { python tools/ab_msm.py $1 $3 2>/dev/null | tail -1 | sed "s/^/[$2] /"; }
for ln in 12 16 20 24; do
  reps=30; [ $ln -ge 22 ] && reps=10
  for i in 1 2 3; do
    run $ln "lean quad add" $reps
    ZKP_HIP_LIB=$OLD run $ln "base" $reps
  done
done > gpurun_out/r04v/ab.txt 2>&1
grep -o "^\[[a-z ]*\]\|n=2^[0-9]* [0-9.]* ms\|'msm_accumulate': [0-9.]*\|'msm_bucket_reduce': [0-9.]*" gpurun_out/r04v/ab.txt | paste - - - -
pl() { python tools/plonk_bench.py 16 $1 2>/dev/null | tail -1 | python -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('$2', 'prove_ms %.3f' % d['prove_ms'], 'second-proof', d.get('generate_proof_ms_with_transcript'), {k: v['ms'] for k, v in d['phase_ms_one_proof'].items()})"; }
for i in 1 2 3; do
  pl auto "[lean quad add]"
  ZKP_HIP_LIB=$OLD pl auto "[base]"
done > gpurun_out/r04v/ab_plonk.txt 2>&1
cat gpurun_out/r04v/ab_plonk.txt
python tools/small_msm_bench.py 2>/dev/null | grep batch | sed "s/^/[new] /"
ZKP_HIP_LIB=$OLD python tools/small_msm_bench.py 2>/dev/null | grep batch | sed "s/^/[base] /"
