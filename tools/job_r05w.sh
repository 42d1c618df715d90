# host polls the MSM's result flags in pinned memory instead of hipStreamSynchronize: parity, then A/B by environment switch
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plonk.py tests/test_gpu_multi_slot.py -x -q -m gpu -k "msm or plonk or sharded or kzg" 2>&1 | tail -2
for i in 1 2; do
echo "== poll"; python tools/ab_msm.py 20 3 | tail -1 | cut -c1-120; python tools/small_msm_bench.py 16 2>&1 | tail -2
python3 tools/plonk_bench.py 16 auto 2>&1 | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('plonk', d['prove_ms'], d['generate_proof_ms_with_transcript'], d['round_ms'])"
echo "== stream wait"; ZKP_MSM_NO_POLL=1 python tools/ab_msm.py 20 3 | tail -1 | cut -c1-120; ZKP_MSM_NO_POLL=1 python tools/small_msm_bench.py 16 2>&1 | tail -2
ZKP_MSM_NO_POLL=1 python3 tools/plonk_bench.py 16 auto 2>&1 | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('plonk', d['prove_ms'], d['generate_proof_ms_with_transcript'], d['round_ms'])"
done
