#!/usr/bin/env python3
"""Timeline of the last PLONK proof in a rocprofv3 kernel trace (tools/prof_plonk_stats.sh): python3 tools/plonk_timeline.py TRACE.csv [OUT.md]
Finds the last run of kernels that starts with the first kernel of round 1 (fr_add_blinding after the memsets) and lists every
kernel with its start, duration and the idle gap before it; sums busy and idle time."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    n = n.replace("zkp::", "")
    n = re.sub(r"Fp<FrParams>\s*", "Fr", n)
    return n[:60]


# split into bursts separated by > 2 ms of idle (proofs are separated by host work: transcripts, verification, set-up)
bursts, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur[-8:]) > 2_000_000:
        bursts.append(cur)
        cur = []
    cur.append(e)
bursts.append(cur)
cands = [b for b in bursts if any("plonk_quotient" in e[2] for e in b)]
b = cands[-1]
t0 = b[0][0]
out = ["| kernel | start us | duration us | idle before us |", "|---|---|---|---|"]
busy_end, busy, idle, gaps = b[0][0], 0, 0, []
for s, e, n in b:
    gap = max(0, s - busy_end)
    idle += gap
    busy += max(0, e - max(s, busy_end))
    if gap > 3000:
        gaps.append((gap / 1e3, short(n)))
    out.append(f"| {short(n)} | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {gap / 1e3:.1f} |")
    busy_end = max(busy_end, e)
total = (busy_end - t0) / 1e3
head = [f"last proof: {len(b)} kernels, {total:.1f} us from the first kernel's start to the last one's end; GPU busy {busy / 1e3:.1f} us, idle {idle / 1e3:.1f} us",
        "idle gaps above 3 us (us, kernel that follows): " + ", ".join(f"{g:.0f} {n}" for g, n in gaps), ""]
text = "\n".join(head + out) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
print("\n".join(head))
agg = {}
for s, e, n in b:
    k = short(n)
    a = agg.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{k:62s} x{c:3d} {t / 1e3:8.1f} us")
