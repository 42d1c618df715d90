"""Print VGPR/occupancy/scratch of every kernel: parses `hipcc -Rpass-analysis=kernel-resource-usage` remarks.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c csrc/api.hip -Rpass-analysis=kernel-resource-usage \
           -o /tmp/x.o 2> /tmp/res.txt && python tools/kernel_resources.py /tmp/res.txt"""
import re
import sys

t = open(sys.argv[1]).read()
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split()[0]

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"

    print("%-58s vgpr %4s agpr %4s spill %3s scratch %5s occ %2s lds %6s" % (
        name[:58], g("VGPRs"), g("AGPRs"), g("VGPR Spill"), g(r"ScratchSize \[bytes/lane\]"),
        g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
