#!/usr/bin/env python3
"""profiles/traffic.json from the rocprofv3 PMC summaries of tools/prof_r05_traffic.sh (gpurun_out/r05traffic/pmc_<log_n>.md and
pmc_ntt24.md): bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the 2 x being the gfx950 FETCH_SIZE correction that
bench_micro/gather128.hip confirmed for random 128-byte records (profiles/r03_c_gather128_calibration.md) and the guide gives for
coalesced streams.  Every entry records the commit it was measured at and a hash of the kernel's source files: bench.py compares that
hash with the files it runs from and marks an entry older than the kernel's last change (`traffic_stale`).
usage: update_traffic.py <dir with pmc_*.md> <profile file to cite>"""
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zkp-implementation_amd", "csrc")
KERNEL_SOURCES = {"msm_accumulate": ["msm.hpp", "g1_28.hpp", "fq28.hpp", "fq28_mul_asm.inc", "fq28_mul2x_asm.inc", "fq28_sqr_asm.inc", "fq28_mul2_asm.inc", "ff.hpp"], "ntt_fr": ["ntt.hpp", "fr29.hpp", "fr29_mul2_asm.inc", "ff.hpp"]}


def kernel_hash(family):
    h = hashlib.sha256()
    for f in KERNEL_SOURCES[family]:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    src_dir, cite = sys.argv[1], sys.argv[2]
    path = os.path.join(ROOT, "profiles", "traffic.json")
    t = json.load(open(path))
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "zkp-implementation_amd/csrc"], capture_output=True, text=True).stdout.strip()
    t.setdefault("_commit", {})
    t.setdefault("_kernel_src_hash", {})

    def put(key, value, family, note=""):
        t[key] = int(value)
        t["_source"][key] = cite + note
        t["_commit"][key] = commit + ("+uncommitted kernel changes" if dirty else "")
        t["_kernel_src_hash"][key] = kernel_hash(family)

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
            put(f"msm_accumulate_log{ln}_c{c}", 2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"], "msm_accumulate",
                " (per launch = one range of 2^24 scalars; a 2^26 MSM makes four)" if ln == 26 else "")
    f = os.path.join(src_dir, "pmc_ntt24.md")
    if os.path.exists(f):
        per = {}
        for line in open(f):
            m = re.match(r"\| (ntt_pass_\w+)<[^|]*\| (\d+) \| ([\d.]+) \| (\w+) \| ([\d.]+) \|", line)
            if m:
                per.setdefault(m.group(1), {})[m.group(4)] = float(m.group(5)) * 1024
        if all(k in per and "FETCH_SIZE" in per[k] and "WRITE_SIZE" in per[k] for k in ("ntt_pass_strided", "ntt_pass_last")):
            strided = 2 * per["ntt_pass_strided"]["FETCH_SIZE"] + per["ntt_pass_strided"]["WRITE_SIZE"]
            last = 2 * per["ntt_pass_last"]["FETCH_SIZE"] + per["ntt_pass_last"]["WRITE_SIZE"]
            put("ntt_fr_pass_log24", strided, "ntt_fr", ": average strided pass (pass 0 also reads the 512 MiB twiddle matrix)")
            put("ntt_fr_transform_log24", 2 * strided + last, "ntt_fr",
                ": two strided passes + the last pass of one 2^24 transform, 2 x FETCH_SIZE + WRITE_SIZE")
    json.dump(t, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in t.items() if not k.startswith("_")}, indent=1))


if __name__ == "__main__":
    main()
