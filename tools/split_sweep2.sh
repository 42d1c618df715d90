# window width per SRS size with the automatic run split (GPU box, repo root); output gpurun_out/split_sweep2.txt
O=gpurun_out/split_sweep2.txt
: > $O
run() { echo "== n=2^$1 c=$2 ${3:-auto}" >> $O; if [ -n "$3" ]; then ZKP_MSM_SPLIT_LOG=$3 timeout -k 10 120 python tools/small_msm_bench.py $1 $2 2>/dev/null >> $O || exit 1; else timeout -k 10 120 python tools/small_msm_bench.py $1 $2 2>/dev/null >> $O || exit 1; fi; }
run 16 16 2; run 16 16 3; run 16 16; run 16 15 3
for c in 10 12 14 16; do run 10 $c; done
for c in 12 13 14 16; do run 12 $c; done
for c in 13 14 15 16; do run 14 $c; done
for c in 14 15 16 18; do run 15 $c; done
for c in 16 18 19; do run 17 $c; done
for c in 16 18 19 20; do run 18 $c; done
for c in 18 19 20; do run 19 $c; done
