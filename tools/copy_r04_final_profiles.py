#!/usr/bin/env python3
"""Copy the summaries tools/prof_r04_final.sh left under gpurun_out/r04final/ into profiles/ (tracked), with a header line each."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "r04final")
P = os.path.join(ROOT, "profiles")
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
rd = lambda f: open(os.path.join(O, f)).read()
clean = lambda t: "\n".join(l for l in t.splitlines() if "amdgpu.ids" not in l and not l.startswith(("W2026", "E2026")))
msm = json.loads(rd("bench_msm.json").strip().splitlines()[-1])
r = msm["roofline"]
open(os.path.join(P, "r04_f_kernel_stats_bench_msm_only.md"), "w").write(
    f"# r04_f — `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-extra --no-cpu-baseline` (round 4, final source, at {commit})\n\n"
    f"The same run printed `roofline.avg_kernel_ms` = {r['avg_kernel_ms']:.4f} ms for msm_accumulate_kernel (HIP events on the launch stream, {msm['steps']} timed "
    f"steps) at `roofline.shader_clock_mhz` = {r['shader_clock_mhz']:.0f} MHz (in-kernel stamps: {r['accumulate_simd_cycles_per_insertion']:.1f} SIMD-cycles "
    f"per insertion), `integer_issue.frac_in_cycles` = {r['integer_issue']['frac_in_cycles']:.3f}; rocprofv3's average over all its calls ({msm['warmup']} warm-up + "
    f"{msm['steps']} timed + the short pass that records the other phases) is below.  "
    "`mad_rate_probe_kernel` is the issue-rate probe the bench line's `integer_issue` peak comes from.  Kernels named `Cijk_*` / `at::native::*` are torch's "
    "(the known-answer check in zkp_hip/trapdoor.py, tensor fills), outside the timed region.\n\n" + rd("stats_msm.md")
    + "\n## One MSM of that workload, launch by launch (`rocprofv3 --kernel-trace -- python3 tools/ab_msm.py 20 3`, last MSM)\n\n" + rd("timeline.md"))
open(os.path.join(P, "r04_f_bench_default.json"), "w").write(rd("bench.json").strip().splitlines()[-1] + "\n")
open(os.path.join(P, "r04_f_kernel_stats_ntt.md"), "w").write(
    f"# r04_f — kernel stats of one Fr NTT 2^24 (`tools/ntt_bench.py fr 24 10`), source at {commit} (the NTT kernels are those of round 3)\n\n"
    + rd("stats_ntt.md") + "\n```\n" + clean(rd("ntt24.log")) + "\n```\n")
print("profiles updated from", O, "at", commit)
