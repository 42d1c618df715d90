#!/usr/bin/env python3
"""Static check of the built library for a scalar-memory / long-branch register race (found in round 5, profiles/r05_l_profile_mode_fault.md):

    s_memrealtime s[0:1]            <- result returns asynchronously (lgkmcnt)
    ...                             <- no s_waitcnt lgkmcnt(0): the value is dead on this path
    s_getpc_b64   s[0:1]            <- LLVM's branch relaxation scavenged the same pair for a long jump
    s_add_u32     s0, s0, <offset>
    s_addc_u32    s1, s1, -1
    s_setpc_b64   s[0:1]            <- a late return of the load has replaced half of the target: the wave jumps into the void

SIInsertWaitcnts runs before branch relaxation, so nothing waits.  This script disassembles every gfx950 kernel of the library and, for
every long-branch sequence, walks the straight-line code before it (up to the last s_waitcnt that drains lgkmcnt, at most 64 instructions)
for a scalar-memory instruction whose destination overlaps the pair the jump uses.  Exit code 1 and one line per hit.
    python tools/check_smem_long_branch.py [lib.so]"""
import re
import struct
import subprocess
import sys

SMEM = ("s_load_", "s_buffer_load_", "s_memtime", "s_memrealtime", "s_scratch_load_", "s_atc_probe")


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
            open("/tmp/zkp_co_lb.o", "wb").write(data[i + o:i + o + sz])
            return "/tmp/zkp_co_lb.o"
    raise SystemExit("no gfx950 code object in " + path)


def sregs(operand):
    """s[4:7] -> {4,5,6,7}; s12 -> {12}; anything else -> {}"""
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", operand)
    return {int(m.group(1))} if m else set()


def drains_lgkm(line):
    """s_waitcnt with lgkmcnt(0) (or the bare form, which waits for everything)"""
    if not line.startswith("s_waitcnt"):
        return False
    return "lgkmcnt(0)" in line or line.strip() == "s_waitcnt" or re.fullmatch(r"s_waitcnt\s+0(x0)?", line.strip()) is not None


def scan(dis):
    hits, kernels, long_branches = [], 0, 0
    for blk in re.split(r"\n(?=[0-9a-f]{16} <)", dis):
        m = re.match(r"[0-9a-f]{16} <(\S+)>:", blk)
        if not m:
            continue
        kernels += 1
        ins = []
        for line in blk.splitlines()[1:]:
            t = line.split("//")[0].strip()
            if t and re.match(r"[a-z]", t):
                ins.append(t)
        for k, t in enumerate(ins):
            if not t.startswith("s_getpc_b64"):
                continue
            long_branches += 1
            pair = sregs(t.split()[1].rstrip(","))
            for j in range(k - 1, max(-1, k - 65), -1):
                p = ins[j]
                if drains_lgkm(p) or p.startswith(("s_branch", "s_endpgm", "s_setpc_b64")):  # drained, or no fall-through from above
                    break
                if p.startswith(SMEM):
                    ops = p.split(None, 1)[1].split(",") if " " in p else []
                    dst = sregs(ops[0].strip()) if ops else set()
                    if dst & pair:
                        hits.append(f"{m.group(1)}: `{p}` still in flight at `{t}` ({k - j} instructions later)")
    return hits, kernels, long_branches


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "zkp-implementation_amd/libzkp_hip.so"
    dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--mcpu=gfx950", code_object(path)], capture_output=True, text=True).stdout
    hits, kernels, lb = scan(dis)
    for h in hits:
        print("HAZARD", h)
    print(f"{path}: {kernels} functions, {lb} long-branch sequences, {len(hits)} with a scalar-memory result in flight into their register pair")
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main())
