#!/usr/bin/env python3
"""profiles/traffic.json entries for msm_accumulate from the rocprofv3 PMC summaries of tools/prof_r03_traffic.sh
(gpurun_out/r03traffic/pmc_<log_n>.md): bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the 2 x being the gfx950 FETCH_SIZE
correction that bench_micro/gather128.hip confirmed for random 128-byte records (profiles/r03_c_gather128_calibration.md).
usage: update_traffic.py <dir with pmc_*.md> <profile file to cite>"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src_dir, cite = sys.argv[1], sys.argv[2]
path = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(path))
for ln in (20, 22, 24, 26):
    f = os.path.join(src_dir, f"pmc_{ln}.md")
    if not os.path.exists(f):
        continue
    vals = {}
    for line in open(f):
        m = re.match(r"\| msm_accumulate_kernel \| (\d+) \| ([\d.]+) \| (\w+) \| ([\d.]+) \|", line)
        if m:
            vals[m.group(3)] = float(m.group(4)) * 1024
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        c = 20 if ln < 22 else 22
        key = f"msm_accumulate_log{ln}_c{c}"
        t[key] = int(2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"])
        t["_source"][key] = cite + (" (per launch = one range of 2^24 scalars; a 2^26 MSM makes four)" if ln == 26 else "")
t.pop("msm_accumulate_log24_c22_per_range", None)
t["_source"].pop("msm_accumulate_log24_c22_per_range", None)
json.dump(t, open(path, "w"), indent=1)
print(json.dumps({k: v for k, v in t.items() if k.startswith("msm_")}, indent=1))
