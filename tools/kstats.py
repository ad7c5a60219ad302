#!/usr/bin/env python3
"""Print a per-step summary of a rocprofv3 --stats kernel_stats.csv:  tools/kstats.py file.csv [n_steps] [top]"""
import csv
import sys

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 34
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6 / steps:.3f} ms/step over {steps} step(s)")
for r in rows[:top]:
    print(f"{r['Name'][:86]:86s} {int(r['Calls']) / steps:7.1f}/step {float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step "
          f"avg {float(r['AverageNs']) / 1e3:9.1f} us")
