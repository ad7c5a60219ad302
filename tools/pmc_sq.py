#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (SQ counters of tools/pmc_sq.txt).
    tools/pmc_sq.py <counter_collection.csv> [name filter]"""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "conv3x3"
acc = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    if flt not in k:
        continue
    m = re.search(r"(\w+<[^>]*>|\w+)\(", k.replace("(anonymous namespace)::", ""))
    k = m.group(1) if m else k
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(acc):
    c = acc[k]
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    wc = c.get("SQ_WAVE_CYCLES", 1.0)
    line = f"{k:48s}"
    if gui:
        line += f" mfma_util {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (gui * 128):.3f}"
    line += (f" wait_any {c.get('SQ_WAIT_ANY', 0) / wc:.2f} wait_inst {c.get('SQ_WAIT_INST_ANY', 0) / wc:.2f}"
             f" active {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} lds_issue_wait {c.get('SQ_WAIT_INST_LDS', 0) / wc:.3f}")
    if c.get("SQ_LDS_IDX_ACTIVE"):
        line += f" bank_conflict/lds_active {c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.3f}"
        if gui:
            line += f" lds_active/cu_cycle {c['SQ_LDS_IDX_ACTIVE'] / (gui * 32):.3f}"
    print(line)
