#!/usr/bin/env python3
"""Static instruction mix, by class, of the gfx950 kernels inside a built library (reads the code object out of the HIP fat binary and
disassembles it with llvm-objdump):
    python tools/isa_histogram.py [lib.so] [--md] [name filter ...]
Classes: the multiply-add a field product is made of (v_mad_u64_u32), 64-bit adds and shifts (half rate like the multiply-add,
bench_micro/issue_rate.hip), 32-bit masks / shifts / adds (full rate), moves and selects, LDS, global memory, scalar and control.
The NTT passes have (almost) no data-dependent control flow -- every lane executes every round -- so the static mix of the kernel
body is the dynamic mix up to the trip counts of the load / store loops; the dynamic totals per element come from the SQ counters
(tools/prof_r05_ntt_counters.sh)."""
import collections
import re
import struct
import subprocess
import sys

CLASSES = [
    ("mad64 (v_mad_u64_u32)", lambda o: o.startswith("v_mad_u64_u32") or o.startswith("v_mad_i64_i32")),
    ("add64 (v_lshl_add_u64, add/addc pairs)", lambda o: o.startswith(("v_lshl_add_u64", "v_add_co", "v_addc_co", "v_sub_co", "v_subb_co", "v_subrev_co"))),
    ("shift64 (v_lshrrev_b64 ...)", lambda o: o.startswith(("v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64"))),
    ("mul32 (v_mul_lo/hi, 24-bit mads)", lambda o: o.startswith(("v_mul_lo", "v_mul_hi", "v_mad_u32_u24", "v_mad_i32_i24", "v_mul_u32_u24", "v_mad_u32_u16"))),
    ("mask (v_and, v_bfe, v_or, v_xor, v_not, v_bfi, v_bitop3)", lambda o: o.startswith(("v_and_", "v_bfe_", "v_or_", "v_xor_", "v_not_", "v_bfi_", "v_bitop3", "v_and_or", "v_or3"))),
    ("shift32 (v_lshr, v_lshl, v_alignbit, v_lshl_or, v_lshl_add_u32, v_ashr)", lambda o: o.startswith(("v_lshrrev_b32", "v_lshlrev_b32", "v_alignbit", "v_lshl_or", "v_lshl_add_u32", "v_ashrrev_i32", "v_add_lshl"))),
    ("add32 (v_add, v_sub, v_add3)", lambda o: o.startswith(("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add3_u32", "v_add_nc", "v_sub_nc", "v_add_i32", "v_sub_i32"))),
    ("move / select / compare", lambda o: o.startswith(("v_mov_", "v_cndmask", "v_cmp", "v_readfirstlane", "v_readlane", "v_writelane", "v_accvgpr", "v_swap", "v_min", "v_max", "v_perm", "v_bfrev", "v_ffb", "v_mbcnt"))),
    ("other VALU", lambda o: o.startswith("v_")),
    ("LDS (ds_*)", lambda o: o.startswith("ds_")),
    ("global / scratch memory", lambda o: o.startswith(("global_", "buffer_", "flat_", "scratch_"))),
    ("wait / nop / barrier", lambda o: o.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep"))),
    ("scalar and control", lambda o: o.startswith("s_")),
]


def code_object(path):
    data = open(path, "rb").read()
    i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    n = struct.unpack_from("<Q", data, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", data, off)
        off += 24
        triple = data[off:off + tl].decode()
        off += tl
        if "gfx950" in triple:
            open("/tmp/zkp_co_hist.o", "wb").write(data[i + o:i + o + sz])
            return "/tmp/zkp_co_hist.o"
    raise SystemExit("no gfx950 code object in " + path)


def main():
    args = [a for a in sys.argv[1:] if a != "--md"]
    md = "--md" in sys.argv
    path = args[0] if args and args[0].endswith(".so") else "zkp-implementation_amd/libzkp_hip.so"
    filters = [a for a in args if not a.endswith(".so")]
    dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--mcpu=gfx950", code_object(path)], capture_output=True, text=True).stdout
    for blk in re.split(r"\n(?=[0-9a-f]{16} <)", dis):
        m = re.match(r"[0-9a-f]{16} <(\S+)>:", blk)
        if not m or (filters and not any(f in m.group(1) for f in filters)):
            continue
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        ops = collections.Counter()
        for line in blk.splitlines()[1:]:
            t = line.strip().split()
            if t and re.match(r"[a-z]", t[0]):
                ops[t[0]] += 1
        total = sum(ops.values())
        by = collections.OrderedDict((c, 0) for c, _ in CLASSES)
        detail = collections.defaultdict(collections.Counter)
        for o, c in ops.items():
            for cname, pred in CLASSES:
                if pred(o):
                    by[cname] += c
                    detail[cname][o] += c
                    break
        valu = sum(c for o, c in ops.items() if o.startswith("v_"))
        if md:
            print(f"\n### `{name}` — {total} instructions, {valu} VALU\n")
            print("| class | count | % of all | % of VALU | most frequent |")
            print("|---|---|---|---|---|")
            for cname, c in by.items():
                if c:
                    top = ", ".join(f"{o} {k}" for o, k in detail[cname].most_common(3))
                    isv = cname.startswith(("mad64", "add64", "shift64", "mul32", "mask", "shift32", "add32", "move", "other VALU"))
                    print(f"| {cname} | {c} | {100 * c / total:.1f} | {(100 * c / valu):.1f} | {top} |" if isv else
                          f"| {cname} | {c} | {100 * c / total:.1f} | | {top} |")
        else:
            print(f"{name}: {total} instructions, {valu} VALU")
            for cname, c in by.items():
                if c:
                    print(f"    {cname:70s} {c:6d}  {100 * c / total:5.1f} %")


if __name__ == "__main__":
    main()
