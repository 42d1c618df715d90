# A/B: the four small read-backs of a PLONK proof waited for by polling a flag a one-thread kernel writes (ZKP_PLONK_NO_POLL=1 = the runtime's stream wait)
out=gpurun_out/r05_plonk_poll.txt
python -m pytest tests/test_gpu_plonk.py tests/test_gpu_parity.py -m gpu -x -q -k "plonk or kzg or open" > gpurun_out/r05_plonk_poll_tests.log 2>&1 || { tail -20 gpurun_out/r05_plonk_poll_tests.log; exit 1; }
tail -1 gpurun_out/r05_plonk_poll_tests.log > $out
python3 tests/soak/fuzz_plonk.py 61 20 2>&1 | tail -1 >> $out || { tail -5 $out; exit 1; }
for i in 1 2 3 4; do
  for v in 1 ""; do
    [ -n "$v" ] && export ZKP_PLONK_NO_POLL=1 || unset ZKP_PLONK_NO_POLL
    echo "[$([ -n "$v" ] && echo "stream wait" || echo "polling")]" >> $out
    python tools/plonk_bench.py 16 auto 2>/dev/null | tail -1 | grep -o "'generate_proof_ms_with_transcript': [0-9.]*\|'generate_proof_ms_with_transcript_median': [0-9.]*\|'prove_ms': [0-9.]*" | paste - - - >> $out
  done
done
cat $out
