#!/usr/bin/env python3
"""Register / scratch usage of the gfx950 kernels inside a built library (reads the code object out of the HIP fat binary):
    python tools/kernel_regs.py [lib.so] [name filter ...]"""
import re
import struct
import subprocess
import sys

path = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else "zkp-implementation_amd/libzkp_hip.so"
filters = [a for a in sys.argv[1:] if not a.endswith(".so")]
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
        open("/tmp/zkp_co.o", "wb").write(data[i + o:i + o + sz])
txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", "/tmp/zkp_co.o"], capture_output=True, text=True).stdout
for blk in txt.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name or (filters and not any(f in name.group(1) for f in filters)):
        continue
    dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
    g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [0, "?"])[1]
    print(f"{dem.split('(')[0][:70]:70s} vgpr {g('vgpr_count'):>4} sgpr {g('sgpr_count'):>4} spill {g('vgpr_spill_count'):>3} "
          f"scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6}")
