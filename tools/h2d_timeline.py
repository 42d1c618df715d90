"""One host-scalar MSM (zkp_msm_g1: scalars in pageable host memory) at 2^LOG, for a launch + copy timeline under rocprofv3:
   rocprofv3 --kernel-trace --memory-copy-trace -d DIR -o t --output-format csv -- python3 tools/h2d_timeline.py [LOG] [REPS]
   python3 tools/h2d_timeline.py --summarise DIR   (last call: kernels and copies on one time axis)"""
import os, sys, glob, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    d = sys.argv[2]
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][:60]))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "copy")))
    rows.sort()
    # the last MSM = from the last msm_digits_kernel that follows a gap back to the end
    tails = [i for i, r in enumerate(rows) if "msm_pyramid_tail" in r[2] or "msm_collect" in r[2]]
    prev_end = tails[-2] if len(tails) >= 2 else -1          # the last MSM starts after the bucket reduction of the one before
    first = next(i for i, r in enumerate(rows) if i > prev_end and "msm_digits" in r[2])
    # include the copy that precedes it
    while first > 0 and rows[first - 1][2].startswith("C") and first - 1 > prev_end: first -= 1
    t0 = rows[first][0]
    print("| what | start us | duration us | gap before us |\n|---|---|---|---|")
    prev_end = t0
    for s, e, name in rows[first:]:
        print(f"| {name} | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {max(0, s - prev_end) / 1e3:.1f} |")
        prev_end = max(prev_end, e)
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "zkp-implementation_amd"))
import time, numpy as np, torch
import bench, zkp_hip as zkp
ln = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
zkp.init()
dev = torch.device("cuda", 0)
n = 1 << ln
ks = bench.rand_fr_tensor(torch, n, 1000 + ln, dev)
sc = bench.rand_fr_tensor(torch, n, 2000 + ln, dev)
pts = torch.zeros(n * 12, dtype=torch.int64, device=dev)
zkp.g1_fixed_base_mul_dev(ks, n, pts); torch.cuda.synchronize()
bases = zkp.G1Bases.from_device(pts, n); bases.precompute(0)
h = sc.cpu().numpy().view(np.uint64).reshape(n, 4).copy()
ref = zkp.msm_g1_dev(bases, sc, n)
for _ in range(3): out = zkp.msm_g1(bases, h)
assert (out[0] == ref[0]).all()
t0 = time.perf_counter()
for _ in range(reps): out = zkp.msm_g1(bases, h)
print(f"host-scalar MSM 2^{ln}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call")
