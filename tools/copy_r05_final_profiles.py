#!/usr/bin/env python3
"""Copy what tools/prof_r05_final.sh left under gpurun_out/r05final/ (and r05traffic/) into profiles/ (tracked), with a header each."""
import ast
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "r05final")
P = os.path.join(ROOT, "profiles")
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
rd = lambda f: open(os.path.join(O, f)).read()
clean = lambda t: "\n".join(l for l in t.splitlines() if "amdgpu.ids" not in l and not l.startswith(("W2026", "E2026")))
last_json = lambda f: [l for l in rd(f).splitlines() if l.startswith("{")][-1]
msm = json.loads(last_json("bench_msm.json"))
r = msm["roofline"]
open(os.path.join(P, "r05_i_kernel_stats_bench_msm_only.md"), "w").write(
    f"# r05_i — `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-extra --no-cpu-baseline` (round 5, final source, at {commit})\n\n"
    f"The same run printed `roofline.avg_kernel_ms` = {r['avg_kernel_ms']:.4f} ms for msm_accumulate_kernel (HIP events on the launch stream, {msm['steps']} timed "
    f"steps) at `roofline.shader_clock_mhz` = {r['shader_clock_mhz']:.0f} MHz (in-kernel stamps: {r['accumulate_simd_cycles_per_insertion']:.1f} SIMD-cycles "
    f"per insertion), `integer_issue.frac_in_cycles` = {r['integer_issue']['frac_in_cycles']:.3f}; rocprofv3's average over all its calls ({msm['warmup']} warm-up + "
    f"{msm['steps']} timed + the short pass that records the other phases) is below.  "
    "`mad_rate_probe_kernel` is the issue-rate probe the bench line's `integer_issue` peak comes from.  Kernels named `Cijk_*` / `at::native::*` are torch's "
    "(the known-answer check in zkp_hip/trapdoor.py, tensor fills), outside the timed region.\n\n" + rd("stats_msm.md")
    + "\n## One MSM of that workload, launch by launch (`rocprofv3 --kernel-trace -- python3 tools/ab_msm.py 20 3`, last MSM)\n\n" + rd("timeline.md"))
open(os.path.join(P, "r05_i_bench_default.json"), "w").write(last_json("bench.json") + "\n")
open(os.path.join(P, "r05_c_in_process_8slots_2_26.json"), "w").write(last_json("in_process_8slots.json") + "\n")
open(os.path.join(P, "r05_i_kernel_stats_ntt.md"), "w").write(
    f"# r05_i — kernel stats of one Fr NTT 2^24 (`tools/ntt_bench.py fr 24 10`), source at {commit}\n\n"
    + rd("stats_ntt.md") + "\n```\n" + clean(rd("ntt24.log")) + "\n```\n")
plain = ast.literal_eval([l for l in rd("plonk_plain.txt").splitlines() if l.startswith("{")][-1])
open(os.path.join(P, "r05_h_plonk_timeline.md"), "w").write(
    f"# r05_h — launch timeline of one `zkp_plonk_prove` at 2^16 gates (round 5, final source, at {commit})\n\n"
    "`rocprofv3 --kernel-trace -- python3 tools/plonk_bench.py 16 auto`, last proof (the one-call prover with its own transcript), `tools/plonk_timeline.py`.  Under the\n"
    "profiler every launch is at least 4.4 µs and the proof takes longer than in a plain run; the same binary without the profiler on the same box:\n"
    f"`generate_proof_ms_with_transcript` = **{plain['generate_proof_ms_with_transcript']:.3f} ms**, five round entries {plain['prove_ms']:.3f} ms, rounds {plain['round_ms']},\n"
    f"phases of one proof {plain['phase_ms_one_proof']}.\n\n"
    "Side-stream work (the coset transforms of a, b, c under round 1's commitments and of z under round 2's) overlaps the MSM kernels: 'GPU busy' counts the union.\n\n"
    + rd("plonk_timeline.md"))
subprocess.check_call(["python3", os.path.join(ROOT, "tools", "copy_r05_profiles.py")], cwd=ROOT)
print("profiles updated from", O, "at", commit)
