#!/usr/bin/env python3
"""Issue-side SQ counters per kernel (is a streaming kernel VALU-bound or memory-bound?).
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY ...
    tools/pmc_valu.py <counter_collection.csv> [substring ...]"""
import csv, re, sys
from collections import defaultdict
flt = sys.argv[2:] or ["pool", "bnrelu", "c1_", "reduce2", "mean_w", "logmel", "gemm"]
acc = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if not any(t in k for t in flt):
        continue
    m = re.search(r"(\w+<[^>]*>|\w+)\(", k.replace("(anonymous namespace)::", ""))
    k = m.group(1) if m else k
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(acc):
    c = acc[k]
    wc, gui = c.get("SQ_WAVE_CYCLES", 1.0), c.get("GRBM_GUI_ACTIVE", 1.0)
    print("%-58s valu_active %.2f any_active %.2f wait_any %.2f wait_inst %.2f valu_insts/simd_cycle %.2f" % (
        k[:58], c.get("SQ_ACTIVE_INST_VALU", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc, c.get("SQ_WAIT_ANY", 0) / wc,
        c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_INSTS_VALU", 0) / (gui * 128)))
