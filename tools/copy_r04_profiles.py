#!/usr/bin/env python3
"""Copy the summaries tools/prof_r04.sh left under gpurun_out/r04prof/ into profiles/ (tracked), with a header line each."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "r04prof")
P = os.path.join(ROOT, "profiles")
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
rd = lambda f: open(os.path.join(O, f)).read()
clean = lambda t: "\n".join(l for l in t.splitlines() if "amdgpu.ids" not in l and not l.startswith(("W2026", "E2026")))
msm = json.load(open(os.path.join(O, "bench_msm.json")))
r = msm["roofline"]
open(os.path.join(P, "r04_b_kernel_stats_bench_msm_only.md"), "w").write(
    f"# r04_b — `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-extra --no-cpu-baseline` (round 4, source at {commit})\n\n"
    f"The same run printed `roofline.avg_kernel_ms` = {r['avg_kernel_ms']:.4f} ms for msm_accumulate_kernel (HIP events on the launch stream, 20 timed "
    f"steps) at `roofline.shader_clock_mhz` = {r['shader_clock_mhz']:.0f} MHz (in-kernel stamps: {r['accumulate_simd_cycles_per_insertion']:.1f} SIMD-cycles "
    "per insertion); rocprofv3's average over its 23 calls (3 warm-up + 20 timed) is below.  `mad_rate_probe_kernel` is the issue-rate probe the bench "
    "line's `integer_issue` peak comes from (zkp_probe_mad_rate, 1 warm-up + 20 launches right after the timed region).  Kernels named `Cijk_*` / "
    "`at::native::*` are torch's (the float64 products of the known-answer check in zkp_hip/trapdoor.py, tensor fills), outside the timed region.\n\n"
    + rd("stats_msm.md"))
open(os.path.join(P, "r04_b_bench_default.json"), "w").write(rd("bench.json"))
open(os.path.join(P, "r04_c_kernel_stats_ntt.md"), "w").write(
    f"# r04_c — kernel stats of one Fr NTT 2^24 (`tools/ntt_bench.py fr 24 10`), source at {commit} (the NTT kernels are those of round 3)\n\n"
    + rd("stats_ntt.md") + "\n```\n" + clean(rd("ntt24.log")) + "\n```\n")
keep = lambda t: "\n".join(l for l in t.splitlines() if l.startswith(("Per-dispatch", "under-counts", "| kernel", "|---")) or "msm_" in l)
open(os.path.join(P, "r04_d_pmc_hbm_msm.md"), "w").write(
    f"# r04_d — HBM traffic of the MSM kernels from the PMC counters, 2^20 (20-bit windows) and 2^24 (22-bit), source at {commit}\n\n"
    "`rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate passes over `tools/ab_msm.py LOG 1` (tools/prof_r04.sh); bytes per launch = "
    "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 (the gfx950 FETCH_SIZE correction, confirmed for 128-byte gathers in profiles/r03_c).  `msm_accumulate_kernel` now "
    "carries the clock stamps and one more argument; its traffic is that of round 3 (2.79 / 35.8 GB).\n\n## 2^20\n\n" + keep(rd("pmc_20.md")) +
    "\n\n## 2^24\n\n" + keep(rd("pmc_24.md")) + "\n")
print("profiles updated from", O, "at", commit)
