#!/usr/bin/env python
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr log)."""
import re
import sys

txt = open(sys.argv[1]).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
KEYS = [("vgpr", r"VGPRs"), ("agpr", r"AGPRs"), ("sgpr", r"SGPRs"),
        ("scratch", r"ScratchSize \[bytes/lane\]"), ("occ", r"Occupancy \[waves/SIMD\]"),
        ("lds", r"LDS Size \[bytes/block\]")]
for b in blocks:
    name = b.split("\n")[0].strip()
    vals = []
    for label, pat in KEYS:
        m = re.search(pat + r": (\d+)", b)
        vals.append(f"{label}={m.group(1) if m else '?':>6}")
    print(f"{name[:70]:72s} " + " ".join(vals))
