"""Registers, scratch and occupancy of every kernel as the compiler reports them.
Usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c rt355.hip 2> res.txt; python tools/kernel_resources.py res.txt"""
import re
import sys

txt = open(sys.argv[1]).read()
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip()

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    row = [g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")]
    print(f"{name[:78]:78s} vgpr {row[0]:>4} agpr {row[1]:>3} sgpr {row[2]:>4} scratch {row[3]:>5} occ {row[4]:>2} lds {row[5]}")
