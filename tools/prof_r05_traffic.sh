# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes as the guide prescribes) of msm_accumulate at 2^20 / 2^22 / 2^24 / 2^26 and of the
# Fr transform passes at 2^24 on THIS binary.  gpurun -- 'bash tools/prof_r05_traffic.sh'; output under
# gpurun_out/r05traffic; `python tools/update_traffic.py gpurun_out/r05traffic profiles/r05_g_pmc_hbm.md` turns it into profiles/traffic.json
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05traffic
mkdir -p $O
cd /tmp
for ln in 20 22 24 26; do
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c -d $O/${c}_$ln -o p --output-format csv -- python3 $R/tools/ab_msm.py $ln 1 > $O/${c}_$ln.log 2>&1
        rc=$?
        echo "$c $ln rc=$rc"
        if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit 1; fi
    done
done
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c -d $O/${c}_ntt24 -o p --output-format csv -- python3 $R/tools/ntt_bench.py fr 24 10 > $O/${c}_ntt24.log 2>&1
    rc=$?
    echo "$c ntt24 rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit 1; fi
done
cd $R
for ln in 20 22 24 26 ntt24; do
    python tools/summarize_prof.py pmc $(find $O/FETCH_SIZE_$ln $O/WRITE_SIZE_$ln -name "*counter_collection.csv") $O/pmc_$ln.md
done
python tools/kernel_regs.py > $O/regs.txt
echo done
