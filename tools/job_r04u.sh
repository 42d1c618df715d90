pl() { python tools/plonk_bench.py 16 auto 2>/dev/null | tail -1 | python -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('$1', 'prove_ms %.3f' % d['prove_ms'], 'second-proof', d.get('generate_proof_ms_with_transcript'), {k: v['ms'] for k, v in d['phase_ms_one_proof'].items()})"; }
for i in 1 2 3; do
  pl "[default]"
  ZKP_MSM_SPLIT_LOG=2 pl "[split 4 everywhere]"
done
python tools/small_msm_bench.py 2>/dev/null | grep batch | sed "s/^/[default] /"
ZKP_MSM_SPLIT_LOG=2 python tools/small_msm_bench.py 2>/dev/null | grep batch | sed "s/^/[split 4] /"
ZKP_MSM_SPLIT_LOG=1 python tools/small_msm_bench.py 2>/dev/null | grep batch | sed "s/^/[split 2] /"
