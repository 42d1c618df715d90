# A/B: plonk_quotient_kernel (268 VGPRs, one wave per SIMD) forced to two waves per SIMD -- build the variant with
#   __attribute__((amdgpu_waves_per_eu(2))) on the kernel as zkp-implementation_amd/libzkp_variant_q2.so (256 VGPRs, 14 spilled, 60 B scratch)
# Result (profiles/r05_q): no difference, not kept.  Output gpurun_out/r05_quot2.txt
out=gpurun_out/r05_quot2.txt
: > $out
V=$PWD/zkp-implementation_amd/libzkp_variant_q2.so
ZKP_HIP_LIB=$V python -m pytest tests/test_gpu_plonk.py -m gpu -x -q > gpurun_out/r05_quot2_tests.log 2>&1 || { tail -20 gpurun_out/r05_quot2_tests.log; exit 1; }
tail -1 gpurun_out/r05_quot2_tests.log >> $out
for i in 1 2 3; do
  for v in "" $V; do
    [ -n "$v" ] && export ZKP_HIP_LIB=$v || unset ZKP_HIP_LIB
    echo "[$([ -n "$v" ] && echo "quotient at 2 waves per SIMD" || echo "shipped")]" >> $out
    python tools/plonk_bench.py 16 auto 2>/dev/null | tail -1 | grep -o "'generate_proof_ms_with_transcript': [0-9.]*\|'generate_proof_ms_with_transcript_median': [0-9.]*\|'prove_ms': [0-9.]*" | paste - - - >> $out
  done
done
cat $out
