# A/B: second insertion of a run with the affine + affine formula (g1_28_mmadd) -- output under gpurun_out/r04q
mkdir -p gpurun_out/r04q
python -m pytest tests/test_gpu_parity.py tests/test_gpu_plonk.py -m gpu -x -q > gpurun_out/r04q/tests.log 2>&1 || { tail -30 gpurun_out/r04q/tests.log; exit 1; }
tail -2 gpurun_out/r04q/tests.log
OLD=$PWD/zkp-implementation_amd/libzkp_hip_base.so
run() { python tools/ab_msm.py $1 $3 2>/dev/null | tail -1 | sed "s/^/[$2] /"; }
for ln in 20 22 24 18; do
  reps=30; [ $ln -ge 22 ] && reps=10
  for i in 1 2 3; do
    run $ln "mmadd peel" $reps
    ZKP_HIP_LIB=$OLD run $ln "base" $reps
  done
done > gpurun_out/r04q/ab.txt 2>&1
cut -c1-270 gpurun_out/r04q/ab.txt
pl() { python tools/plonk_bench.py 16 $1 2>/dev/null | tail -1 | python -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('$2', 'prove_ms %.3f' % d['prove_ms'], 'second-proof', d.get('generate_proof_ms_with_transcript'), {k: v['ms'] for k, v in d['phase_ms_one_proof'].items()})"; }
for i in 1 2 3; do
  pl auto "[mmadd peel]"
  ZKP_HIP_LIB=$OLD pl auto "[base]"
done > gpurun_out/r04q/ab_plonk.txt 2>&1
cat gpurun_out/r04q/ab_plonk.txt
