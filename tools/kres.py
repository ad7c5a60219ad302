#!/usr/bin/env python3
"""Per-kernel VGPR / scratch / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` (CPU-only check).
usage: python tools/kres.py texttoaudiogrounding_amd/csrc/conv.hip [name filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math", "-ffp-contract=off",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = cur.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(\w[\w \[\]/]*): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print(f"{k:60s} VGPR {v.get('VGPRs',0):4d} AGPR {v.get('AGPRs',0):3d} SGPR {v.get('TotalSGPRs',0):4d} scratch {v.get('ScratchSize [bytes/lane]',0):4d} "
              f"occ {v.get('Occupancy [waves/SIMD]',0)} LDS {v.get('LDS Size [bytes/block]',0)}")
