#!/usr/bin/env python3
"""profiles/r04_e_bucket_reduce_counters.md from what tools/prof_r04_pyr.sh and tools/job_r04g.sh / job_r04h.sh left under gpurun_out/."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
rd = lambda *p: open(os.path.join(G, *p)).read()


def table(md):
    rows = [[c.strip() for c in l.strip().strip("|").split("|")] for l in md.splitlines() if l.startswith("|") and not l.startswith("|---")]
    head, body = rows[0], rows[1:]
    return [dict(zip(head, r)) for r in body]


SIMDS = 1024
out = ["# r04_e — where the bucket reduction's time is: launch timeline, counters per level, geometry of the last-levels launch (VERDICT r3 #5)\n",
       "2^20-term MSM over an SRS expanded at 20 bits: 2^19 buckets, 19 levels.  Launch l performs (l + 1) x 2^(18-l) XYZZ additions "
       "(one pairwise level of the pyramid plus one halving of each of the l + 1 odd-sum arrays, `csrc/msm.hpp: pyr_item`): 2^20 additions of "
       "14.5 field products in all, 768 B moved per addition (two 256-byte operands in, one out).\n",
       "## 1. One MSM, launch by launch (`rocprofv3 --kernel-trace`, `tools/prof_r04_pyr.sh`, third of three MSMs; levels stored de-interleaved, source at 841f31a + the "
       "experimental one-workgroup finish, which changes nothing measurable)\n", rd("r04pyr", "timeline.md")]
cnt = table(rd("r04pyr", "pyr_counters.md")) + table(rd("r04pyr", "acc_counters.md"))
out.append("\n## 2. Counters per launch size (`--pmc`, one set per pass, `tools/ab_msm.py 20 2`: 9 dispatches per row, 18 where levels 0 and 1 share a size)\n")
out.append("Units: SQ_WAVE_CYCLES, SQ_ACTIVE_INST_*, SQ_WAIT_* count in quad-cycles (4 shader cycles) summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs; "
           "FETCH_SIZE x 2 x 1024 and WRITE_SIZE x 1024 are bytes (gfx950 correction for full-line reads).  `VALU busy` = 4 x SQ_ACTIVE_INST_VALU / "
           "(kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8.\n")
out.append("| kernel | additions | us (counter run) | VALU instr per wave | VALU busy | wave waits on memory (SQ_WAIT_ANY / SQ_WAVE_CYCLES) | waits for issue (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES) | MB read | MB written | TB/s |")
out.append("|---|---|---|---|---|---|---|---|---|---|")
for r in cnt:
    f = lambda k: float(r[k])
    items = int(r["work-items"])
    adds = items if "quad" not in r["kernel"] and "tail" not in r["kernel"] and "accumulate" not in r["kernel"] else (items // 4 if "accumulate" not in r["kernel"] else 0)
    cyc = f("GRBM_GUI_ACTIVE") / 8
    busy = 4 * f("SQ_ACTIVE_INST_VALU") / (cyc * SIMDS)
    mb_r, mb_w = 2 * f("FETCH_SIZE") * 1024 / 1e6, f("WRITE_SIZE") * 1024 / 1e6
    us = f("avg us (counter runs)")
    label = "levels 12-18 (1 771)" if "tail" in r["kernel"] else (adds if adds else "13.6 M insertions")
    out.append(f"| {r['kernel']} | {label} | {us:.1f} | {f('SQ_INSTS_VALU') / f('SQ_WAVES'):.0f} | {busy:.0%} | "
               f"{f('SQ_WAIT_ANY') / f('SQ_WAVE_CYCLES'):.0%} | {f('SQ_WAIT_INST_ANY') / f('SQ_WAVE_CYCLES'):.0%} | {mb_r:.1f} | {mb_w:.1f} | {(mb_r + mb_w) / us:.2f} |")
out.append("")
body = open(os.path.join(ROOT, "profiles", "r04_e_reading.md.in")).read()
out.append(body)
for name, d in (("first sweep (`tools/job_r04g.sh`; defaults then 64 x 64)", "r04g"), ("second sweep (`tools/job_r04h.sh`; defaults 256 x 16)", "r04h"),
                ("third sweep (`tools/job_r04i.sh`; relaxed polling, padded counters, defaults 256 x 16 = committed)", "r04i")):
    p = os.path.join(G, d, "ab_tail.txt")
    if not os.path.exists(p):
        continue
    out.append(f"\n### {name}\n\n| variant | n | MSM ms | bucket reduction ms (two runs) |\n|---|---|---|---|")
    agg = {}
    for l in open(p):
        m = re.match(r"\[(.*?)\] .*n=2\^(\d+) ([\d.]+) ms.*'msm_bucket_reduce': ([\d.]+)", l)
        if m:
            agg.setdefault((m.group(1), m.group(2)), []).append((m.group(3), m.group(4)))
    for (v, n), xs in agg.items():
        out.append(f"| {v} | 2^{n} | {' / '.join(x[0] for x in xs)} | {' / '.join(x[1] for x in xs)} |")
    for extra in ("ab_plonk.txt", "ab_plain.txt"):
        q = os.path.join(G, d, extra)
        if os.path.exists(q) and d in ("r04h", "r04i"):
            out.append("\n```\n" + "\n".join(l[:400] for l in open(q).read().splitlines()) + "\n```")
open(os.path.join(ROOT, "profiles", "r04_e_bucket_reduce_counters.md"), "w").write("\n".join(out) + "\n")
print("written")
