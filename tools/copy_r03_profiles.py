#!/usr/bin/env python3
"""Copy the summaries tools/prof_r03.sh left under gpurun_out/r03prof/ into profiles/ (tracked), with a header line each."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "r03prof")
P = os.path.join(ROOT, "profiles")
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
rd = lambda f: open(os.path.join(O, f)).read()
clean = lambda t: "\n".join(l for l in t.splitlines() if "amdgpu.ids" not in l and not l.startswith(("W2026", "E2026")))
msm = json.load(open(os.path.join(O, "bench_msm.json")))
open(os.path.join(P, "r03_h_kernel_stats_bench_msm_only.md"), "w").write(
    f"# r03_h — `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-extra --no-cpu-baseline` (round 3, source at {commit})\n\n"
    f"The same run printed `roofline.avg_kernel_ms` = {msm['roofline']['avg_kernel_ms']:.4f} ms for msm_accumulate_kernel (HIP events on the "
    "launch stream, 20 timed steps); rocprofv3's average over its 23 calls (3 warm-up + 20 timed) is below.  Kernels named `Cijk_*` / "
    "`at::native::*` are torch's (the float64 products of the known-answer check in zkp_hip/trapdoor.py, tensor fills), outside the timed "
    "region.\n\n" + rd("stats_msm.md"))
open(os.path.join(P, "r03_h_bench_default.json"), "w").write(rd("bench.json"))
open(os.path.join(P, "r03_i_kernel_stats_ntt.md"), "w").write(
    "# r03_i — kernel stats of one Fr NTT 2^24 (`tools/ntt_bench.py fr 24 10`) and of the per-rank kernels of the four-step 2^26 transform "
    f"with the 2^8 x 2^18 split (`tools/four_step_local_bench.py 26 8`), source at {commit}\n\n" + rd("stats_ntt.md") + "\n" + rd("stats_fourstep.md") +
    "\n```\n" + clean(rd("ntt24.log")) + "\n" + clean(rd("fourstep.log")) + "\n```\n")
print("profiles updated from", O, "at", commit)
