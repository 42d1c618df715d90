# round-5 checkpoint: whole GPU suite, then the default bench line
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05o_gputests.log 2>&1; tail -4 gpurun_out/r05o_gputests.log
python bench.py > gpurun_out/r05o_bench_default.json 2> gpurun_out/r05o_bench.err; tail -c 300 gpurun_out/r05o_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05o_bench_default.json') if l.startswith('{')][0])
e=d['extra']
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'exact', d['bit_exact_full'])
print('roofline', {k:d['roofline'][k] for k in ('frac','avg_kernel_ms','shader_clock_mhz','traffic','traffic_source')})
print('issue', d['roofline']['integer_issue']['frac_in_cycles'], 'phases', d['roofline']['phase_ms'])
print('h2d', e['msm_h2d_inclusive'])
print('unexp', e['msm_unexpanded_bases']['ms_per_step'], 'batch4', e['msm_batch_of_4']['ms_per_msm'])
print('grid', {k:(v.get('ms_per_msm'), v.get('bit_exact_full'), v.get('traffic_source','')[-60:]) for k,v in e['msm_grid'].items() if isinstance(v,dict)})
n=e['ntt_fr']; print('ntt', n['roundtrip_ms'], n['roundtrip_ms_median'], n.get('shader_clock_mhz'), n.get('integer_issue',{}).get('frac_in_cycles'), n['traffic_source'][-80:])
print('nttgrid', {k:v.get('forward_ms') for k,v in e['ntt_grid'].items() if isinstance(v,dict)})
print('fri', e['fri']['prove_ms'], 'plonk', e['plonk']['prove_ms'], e['plonk']['generate_proof_ms_with_transcript'], e['plonk']['round_ms'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['gpu_bit_exact_on_sample'])
PY
