#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the short summaries committed under profiles/.

usage: summarize_prof.py stats <kernel_stats.csv> <out.md>
       summarize_prof.py pmc <counter_collection.csv> [...] <out.md>
"""
import csv
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    name = name.replace("zkp::", "")
    return name[:70]


def stats(path, out):
    rows = list(csv.DictReader(open(path)))
    with open(out, "w") as f:
        f.write("| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n")
        for r in rows:
            f.write(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | "
                    f"{float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['Percentage']):.2f} |\n")


def pmc(paths, out):
    agg = defaultdict(lambda: defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if "at::native" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]:
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[k]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(out, "w") as f:
        f.write("Per-dispatch averages.  FETCH_SIZE / WRITE_SIZE are rocprofv3's raw values in KiB; on gfx950 FETCH_SIZE\n"
                "under-counts wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section), other widths uncalibrated.\n\n")
        f.write("| kernel | dispatches | avg us | counter | avg raw (KiB) | avg MB |\n|---|---|---|---|---|---|\n")
        for k, cs in sorted(agg.items()):
            for cn, vals in sorted(cs.items()):
                if cn == "_dur_us":
                    continue
                avg = sum(vals) / len(vals)
                dur = sum(cs["_dur_us"]) / len(cs["_dur_us"])
                f.write(f"| {k} | {len(vals)} | {dur:.1f} | {cn} | {avg:.1f} | {avg * 1024 / 1e6:.2f} |\n")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2:-1], sys.argv[-1])
