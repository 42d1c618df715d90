#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the short summaries committed under profiles/.

usage: summarize_prof.py stats <kernel_stats.csv> <out.md>
       summarize_prof.py pmc <counter_collection.csv> [...] <out.md>
       summarize_prof.py pmc_by_grid <kernel name part> <counter_collection.csv> [...] <out.md>
       summarize_prof.py timeline <kernel_trace.csv> <first kernel> <last kernel> <out.md>
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




def _grid(r):
    """total work-items of a dispatch row (kernel_trace.csv has Grid_Size_X/Y/Z, counter_collection.csv has Grid_Size)"""
    if "Grid_Size" in r and r["Grid_Size"]:
        return int(r["Grid_Size"])
    g = 1
    for ax in "XYZ":
        g *= int(r.get("Grid_Size_" + ax) or r.get("Grid_Size_" + ax.lower()) or 1)
    return g


def pmc_by_grid(paths, out, pattern):
    """Like pmc, but one row per (kernel, grid size): the launches of a level kernel differ by level."""
    agg = defaultdict(lambda: defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            if pattern not in r["Kernel_Name"]:
                continue
            k = (short(r["Kernel_Name"]), _grid(r))
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[k]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    names = sorted({c for cs in agg.values() for c in cs if c != "_dur_us"})
    with open(out, "w") as f:
        f.write("| kernel | work-items | dispatches | avg us (counter runs) | " + " | ".join(names) + " |\n|---|---|---|---|" + "---|" * len(names) + "\n")
        for (k, g), cs in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
            dur = sum(cs["_dur_us"]) / len(cs["_dur_us"])
            cells = [f"{sum(cs[c]) / len(cs[c]):.4g}" if cs.get(c) else "-" for c in names]
            f.write(f"| {k} | {g} | {max(len(cs[c]) for c in names)} | {dur:.1f} | " + " | ".join(cells) + " |\n")


def timeline(path, out, first, last):
    """Dispatch-by-dispatch timeline of the LAST run of kernels first .. last in a kernel_trace.csv (one MSM): start offset, duration, gap."""
    rows = [r for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if last in r["Kernel_Name"]]
    if not ends:
        raise SystemExit("no dispatch of " + last)
    hi = ends[-1]
    lo = max(i for i in range(hi + 1) if first in rows[i]["Kernel_Name"])
    t0 = int(rows[lo]["Start_Timestamp"])
    prev_end = None
    with open(out, "w") as f:
        f.write("| # | kernel | work-items | start us | duration us | idle before us |\n|---|---|---|---|---|---|\n")
        for n, r in enumerate(rows[lo:hi + 1]):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:.1f}"
            f.write(f"| {n} | {short(r['Kernel_Name'])} | {_grid(r)} | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {gap} |\n")
            prev_end = e
        f.write(f"\nfirst start to last end: {(prev_end - t0) / 1e3:.1f} us\n")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "pmc_by_grid":   # pmc_by_grid PATTERN csv... out.md
        pmc_by_grid(sys.argv[3:-1], sys.argv[-1], sys.argv[2])
    elif sys.argv[1] == "timeline":      # timeline kernel_trace.csv FIRST LAST out.md
        timeline(sys.argv[2], sys.argv[5], sys.argv[3], sys.argv[4])
    else:
        pmc(sys.argv[2:-1], sys.argv[-1])
