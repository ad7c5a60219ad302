#!/usr/bin/env python3
"""Print a per-step summary of a rocprofv3 --stats kernel_stats.csv:  tools/kstats.py file.csv [n_steps] [top]"""
import csv
import sys

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 34
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6 / steps:.3f} ms/step over {steps} step(s)  (averages include the warm-up steps: a kernel's FIRST "
      f"launch carries its code-object load -- 'steady' = the average without the slowest launch, 'min' = the fastest)")
for r in rows[:top]:
    calls, total, mx = int(r["Calls"]), float(r["TotalDurationNs"]), float(r.get("MaxNs", 0) or 0)
    steady = (total - mx) / (calls - 1) if calls > 1 and mx > 0 else total / max(calls, 1)
    print(f"{r['Name'][:86]:86s} {calls / steps:7.1f}/step {total / 1e6 / steps:8.3f} ms/step "
          f"avg {float(r['AverageNs']) / 1e3:9.1f} us  steady {steady / 1e3:9.1f}  min {float(r.get('MinNs', 0) or 0) / 1e3:9.1f}")
