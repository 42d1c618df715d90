# After the clk_end fix (ff.hpp): the configurations that faulted (profiles/r05_l_profile_mode_fault.md) with the shipped library, profiling on
# for every MSM of the second half.  Stops at the first failure -- output gpurun_out/r05_profile_fault_check.txt
out=gpurun_out/r05_profile_fault_check.txt
: > $out
python tools/check_smem_long_branch.py >> $out 2>&1 || { cat $out; exit 1; }
for cfg in "ZKP_MSM_FIRST_PCT=6" "ZKP_MSM_FIRST_PCT=12" "ZKP_MSM_FIRST_PCT=6 ZKP_MSM_NO_OVERLAP=1" "ZKP_MSM_FIRST_PCT=6"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 300 python tools/ab_msm.py 24 10 >> $out 2>&1 || { echo "FAILED rc=$?" >> $out; cut -c1-200 $out | tail -20; exit 1; }
done
echo "== profiled MSMs at 2^20 (300) and the NTT clock stamps (200 transforms at 2^20)" >> $out
timeout -k 10 300 python - >> $out 2>&1 <<'PY' || { echo "FAILED rc=$?" >> $out; cut -c1-200 $out | tail -20; exit 1; }
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "zkp-implementation_amd"))
import torch, bench, zkp_hip as zkp
zkp.init()
dev = torch.device("cuda", 0)
n = 1 << 20
ks = bench.rand_fr_tensor(torch, n, 1, dev); sc = bench.rand_fr_tensor(torch, n, 2, dev)
pts = torch.zeros(n * 12, dtype=torch.int64, device=dev)
zkp.g1_fixed_base_mul_dev(ks, n, pts); torch.cuda.synchronize()
bases = zkp.G1Bases.from_device(pts, n); bases.precompute(0)
ref = zkp.msm_g1_dev(bases, sc, n)
zkp.profile_reset(); zkp.profile_enable(True)
for i in range(300):
    out = zkp.msm_g1_dev(bases, sc, n)
    assert (out[0] == ref[0]).all()
    if i % 50 == 49: zkp.profile_reset()
x = bench.rand_fr_tensor(torch, n, 3, dev)
for i in range(200):
    zkp.ntt_fr_dev(x, 20, inverse=bool(i & 1))
torch.cuda.synchronize()
print("clock reads:", zkp.profile_clock_read("msm_accumulate"), zkp.profile_clock_read("ntt_fr_pass"))
zkp.profile_enable(False)
print("OK 300 profiled MSMs + 200 profiled transforms")
PY
cut -c1-200 $out | tail -20
