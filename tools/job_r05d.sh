python bench.py --steps 20 --warmup 10 --grid-max-log-n 0 > gpurun_out/r05d_bench_nogrid.json 2> gpurun_out/r05d_bench.err && python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05d_bench_nogrid.json') if l.startswith('{')][0])
print(d['ms_per_step'], json.dumps(d['extra']['ntt_fr'], indent=1))
PY
bash tools/prof_r05_ntt_counters.sh
