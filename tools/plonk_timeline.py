#!/usr/bin/env python3
"""Timeline of the last PLONK proof in a rocprofv3 kernel trace (tools/prof_plonk_stats.sh):
  python3 tools/plonk_timeline.py TRACE.csv [OUT.md]
The last proof of tools/plonk_bench.py is the one zkp_plonk_prove runs with its own transcript, followed by the verifier's circuit
commitments.  It spans from the blinding kernel of round 1 (the one before the last plonk_quotient_kernel) to the read-back after the
second MSM tail kernel that follows that quotient kernel (round 5).  Prints every kernel with start, duration and the idle time before
it, the idle gaps above 15 us, and the busy time per kernel name."""
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
    return n[:48] or "(memset / copy)"


q = max(i for i, e in enumerate(ev) if "plonk_quotient" in e[2])
s = max(i for i, e in enumerate(ev[:q]) if "plonk_blind" in e[2] and e[0] < ev[q][0] - 1_000_000)  # round 1's, not round 2's
tails = [i for i, e in enumerate(ev) if "pyramid_tail" in e[2] and i > q]
k = tails[1]  # round 3's MSM, round 5's MSM (its last launch writes the result points straight into pinned host memory)
t0 = ev[s][0]
busy_end, busy, idle, gaps, agg = t0, 0, 0, [], {}
out = ["| kernel | start us | duration us | idle before us |", "|---|---|---|---|"]
for a, b, n in ev[s:k + 1]:
    gap = max(0, a - busy_end)
    idle += gap
    busy += max(0, b - max(a, busy_end))
    if gap > 15000:
        gaps.append(f"{gap / 1e3:.0f} us before {short(n)}")
    out.append(f"| {short(n)} | {(a - t0) / 1e3:.1f} | {(b - a) / 1e3:.1f} | {gap / 1e3:.1f} |")
    x = agg.setdefault(short(n), [0, 0])
    x[0] += 1
    x[1] += b - a
    busy_end = max(busy_end, b)
head = [f"Last proof (zkp_plonk_prove, transcript on the host): {k + 1 - s} kernels, {(busy_end - t0) / 1e3:.0f} us from the first kernel of round 1 to "
        f"the read-back of round 5's commitments; GPU busy {busy / 1e3:.0f} us, idle {idle / 1e3:.0f} us (every kernel is >= 4.4 us under the profiler).",
        "", "Idle gaps above 15 us (host work between dependent steps): " + "; ".join(gaps), "",
        "| kernel | launches | busy us |", "|---|---|---|"]
head += [f"| {n} | {c} | {t / 1e3:.1f} |" for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])]
text = "\n".join(head + [""] + out) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
print("\n".join(head))
