timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05u_gputests.log 2>&1; tail -3 gpurun_out/r05u_gputests.log
python3 tools/plonk_bench.py 16 auto 2>&1 | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print(d['prove_ms'], d['generate_proof_ms_with_transcript'], d['round_ms'], d['phase_ms_one_proof']['msm_accumulate'], d['phase_ms_one_proof']['msm_bucket_reduce'], d['verified_with_pairings'])"
python3 tools/plonk_bench.py 16 auto 2>&1 | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print(d['prove_ms'], d['generate_proof_ms_with_transcript'], d['round_ms'])"
python3 tools/ab_msm.py 20 3 | tail -1 | cut -c1-330
