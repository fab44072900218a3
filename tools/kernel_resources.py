#!/usr/bin/env python3
"""Prints VGPR/SGPR/scratch/LDS/occupancy for every kernel of a .hip file compiled for gfx950."""
import os
import re
import subprocess
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "obs-color-monitor_amd/csrc/scope_kernels.hip"
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-I", os.path.dirname(os.path.abspath(src)), "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
keys = {"VGPRs": "vgpr", "TotalSGPRs": "sgpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occ",
        "LDS Size [bytes/block]": "lds"}
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for k, short in keys.items():
        m = re.search(re.escape(k) + r": (\d+)", line)
        if m and cur is not None:
            cur[short] = m.group(1)
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    print(f"{name:52s} " + " ".join(f"{s}={r.get(s)}" for s in keys.values()))
