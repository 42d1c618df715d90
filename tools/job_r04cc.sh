# A/B: MSM results written by the last kernel straight into pinned host memory (no device-to-host copy launch) -- output under gpurun_out/r04cc
mkdir -p gpurun_out/r04cc
python -m pytest tests/test_gpu_parity.py tests/test_gpu_plonk.py tests/test_gpu_multi_slot.py tests/test_abi_c_caller.py -m gpu -x -q > gpurun_out/r04cc/tests.log 2>&1 || { tail -30 gpurun_out/r04cc/tests.log; exit 1; }
tail -2 gpurun_out/r04cc/tests.log
OLD=$PWD/zkp-implementation_amd/libzkp_hip_base.so
run() { python tools/ab_msm.py $1 30 2>/dev/null | tail -1 | sed "s/^/[$2] /"; }
for ln in 20 16 12; do
  for i in 1 2 3; do
    run $ln "zero copy"
    ZKP_HIP_LIB=$OLD run $ln "memcpy"
  done
done > gpurun_out/r04cc/ab.txt 2>&1
grep -o "^\[[a-z ]*\]\|n=2^[0-9]* [0-9.]* ms" gpurun_out/r04cc/ab.txt | paste - -
pl() { python tools/plonk_bench.py 16 auto 2>/dev/null | tail -1 | python -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('$1', 'prove_ms %.3f' % d['prove_ms'], 'second-proof', d.get('generate_proof_ms_with_transcript'))"; }
for i in 1 2 3; do pl "[zero copy]"; ZKP_HIP_LIB=$OLD pl "[memcpy]"; done
