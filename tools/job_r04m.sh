# A/B: software prefetch of index + point in msm_accumulate_quad (latency-bound small MSMs) -- output under gpurun_out/r04m
mkdir -p gpurun_out/r04m
python -m pytest tests/test_gpu_parity.py tests/test_gpu_plonk.py -m gpu -x -q > gpurun_out/r04m/tests.log 2>&1 || { tail -30 gpurun_out/r04m/tests.log; exit 1; }
tail -2 gpurun_out/r04m/tests.log
OLD=$PWD/zkp-implementation_amd/libzkp_hip_base.so
run() { python tools/ab_msm.py $1 30 2>/dev/null | tail -1 | sed "s/^/[$2] /"; }
for ln in 16 12 14 18; do
  for i in 1 2; do
    run $ln "prefetch"
    ZKP_HIP_LIB=$OLD run $ln "base"
  done
done > gpurun_out/r04m/ab_quad.txt 2>&1
cut -c1-260 gpurun_out/r04m/ab_quad.txt
pl() { python tools/plonk_bench.py 16 $1 2>/dev/null | tail -1 | python -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('$2', 'prove_ms %.3f' % d['prove_ms'], 'second-proof', d.get('generate_proof_ms_with_transcript'), {k: v['ms'] for k, v in d['phase_ms_one_proof'].items()})"; }
for i in 1 2 3; do
  pl auto "[prefetch, expanded SRS]"
  ZKP_HIP_LIB=$OLD pl auto "[base, expanded SRS]"
done > gpurun_out/r04m/ab_plonk.txt 2>&1
cat gpurun_out/r04m/ab_plonk.txt
python tools/small_msm_bench.py > gpurun_out/r04m/small_new.txt 2>&1; tail -12 gpurun_out/r04m/small_new.txt
ZKP_HIP_LIB=$OLD python tools/small_msm_bench.py > gpurun_out/r04m/small_base.txt 2>&1; tail -12 gpurun_out/r04m/small_base.txt
