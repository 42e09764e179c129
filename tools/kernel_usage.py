"""Registers / scratch / occupancy of the kernels of one translation unit, from hipcc -Rpass-analysis=kernel-resource-usage:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -c co-zkvms_amd/csrc/poly.hip -o /tmp/poly.o -Rpass-analysis=kernel-resource-usage 2> usage.txt
  python tools/kernel_usage.py usage.txt [name-regex]"""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else "."
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    if re.search(pat, name):
        print("%-90s VGPR %4s AGPR %3s scratch %5s occ %s" % (name[:90], g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
