# HBM traffic of msm_accumulate per launch at the north-star sizes (FETCH_SIZE and WRITE_SIZE in separate passes, as the guide prescribes)
# gpurun -- 'bash tools/prof_r03_traffic.sh'; output under gpurun_out/r03traffic; tools/update_traffic.py turns it into profiles/traffic.json
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03traffic
mkdir -p $O
cd /tmp
for ln in 20 22 24 26; do
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c -d $O/${c}_$ln -o p --output-format csv -- python3 $R/tools/ab_msm.py $ln 1 > $O/${c}_$ln.log 2>&1
        rc=$?
        echo "$c $ln rc=$rc"
        if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit 1; fi
    done
done
cd $R
for ln in 20 22 24 26; do
    python tools/summarize_prof.py pmc $(find $O/FETCH_SIZE_$ln $O/WRITE_SIZE_$ln -name "*counter_collection.csv") $O/pmc_$ln.md
done
echo done
