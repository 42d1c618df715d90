# second A/B of the last-levels launch: defaults 256 x 16 (four waves per workgroup), many-window (per-window mode) cases -- output under gpurun_out/r04h
mkdir -p gpurun_out/r04h
python -m pytest tests/test_gpu_parity.py tests/test_gpu_plonk.py -m gpu -x -q > gpurun_out/r04h/tests.log 2>&1 || { tail -30 gpurun_out/r04h/tests.log; exit 1; }
tail -2 gpurun_out/r04h/tests.log
OLD=$PWD/zkp-implementation_amd/libzkp_hip_nosolo.so
run() { python tools/ab_msm.py $1 30 2>/dev/null | tail -1 | sed "s/^/[$2] /"; }
for ln in 20 16 12; do
  for i in 1 2; do
    run $ln "new 256x16"
    ZKP_HIP_LIB=$OLD run $ln "round-3 tail, ds_bpermute"
    ZKP_PYR_TAIL_THREADS=256 ZKP_PYR_TAIL_BLOCKS=8 run $ln "256x8"
    ZKP_PYR_TAIL_THREADS=256 ZKP_PYR_TAIL_BLOCKS=32 run $ln "256x32"
    ZKP_PYR_TAIL_HALF=128 ZKP_PYR_TAIL_BLOCKS=32 run $ln "256x32 from 128 pairs"
  done
done > gpurun_out/r04h/ab_tail.txt 2>&1
cut -c1-200 gpurun_out/r04h/ab_tail.txt
pl() { python tools/plonk_bench.py 16 $1 2>/dev/null | tail -1 | python -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('$2', 'prove_ms %.3f' % d['prove_ms'], 'second-proof', d.get('generate_proof_ms_with_transcript'), {k: v['ms'] for k, v in d['phase_ms_one_proof'].items()})"; }
for i in 1 2; do
  pl auto "[new, expanded SRS]"
  ZKP_HIP_LIB=$OLD pl auto "[round-3 tail, expanded SRS]"
  pl 0 "[new, plain bases: 48 bucket sets]"
  ZKP_HIP_LIB=$OLD pl 0 "[round-3 tail, plain bases]"
  ZKP_PYR_TAIL_BLOCKS=8 pl 0 "[256x8, plain bases]"
done > gpurun_out/r04h/ab_plonk.txt 2>&1
cat gpurun_out/r04h/ab_plonk.txt
for i in 1 2; do
  ZKP_BENCH_EXPAND=0 python bench.py --no-extra --no-cpu-baseline --expand-bases 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[new, 2^20 plain bases]', d['ms_per_step'], d['roofline']['phase_ms'])"
  ZKP_HIP_LIB=$OLD python bench.py --no-extra --no-cpu-baseline --expand-bases 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[round-3 tail, 2^20 plain bases]', d['ms_per_step'], d['roofline']['phase_ms'])"
done > gpurun_out/r04h/ab_plain.txt 2>&1
cat gpurun_out/r04h/ab_plain.txt
