#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, as
MI355X_MICROARCH.md prescribes).  Units: the counters are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the
bytes of wide (16 B/lane) coalesced reads -> x2 correction on the read side.
    tools/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv out.json [n_steps_in_run]
"""
import collections
import csv
import json
import re
import sys


def kname(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9:]+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:40]


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = agg[kname(r["Kernel_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1; a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg


f = per_kernel(sys.argv[1], "FETCH_SIZE")
w = per_kernel(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
res = {}
tf = tw = 0.0
for k in sorted(f, key=lambda k: -f[k][0]):
    fe, n, dur = f[k]
    we, n2, _ = w.get(k, [0.0, 1, 0])
    fb, wb = fe * 1024 * 2 / n, we * 1024 / max(n2, 1)
    tf += fe * 1024 * 2; tw += we * 1024
    res[k] = {"launches_in_run": n, "fetch_GB_per_launch": round(fb / 1e9, 4), "write_GB_per_launch": round(wb / 1e9, 4),
              "avg_us_under_pmc": round(dur / n / 1e3, 1)}
    if fb + wb > 5e7:
        print(f"{k:48s} n={n:3d} fetch {fb / 1e9:7.3f} GB write {wb / 1e9:7.3f} GB /launch, {(fb + wb) / max(dur / n, 1):7.2f} GB/ms")
print(f"per step: fetch {tf / steps / 1e9:.2f} GB, write {tw / steps / 1e9:.2f} GB")
import datetime
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from texttoaudiogrounding_amd.lib import build_id as csrc_sha256          # the profile is only valid for THESE kernel sources
res["_meta"] = {"csrc_sha256": csrc_sha256(), "collected_utc": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"), "steps_in_run": steps,
                "fetch_GB_per_step": round(tf / steps / 1e9, 3), "write_GB_per_step": round(tw / steps / 1e9, 3),
                "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md, HBM section); "
                       "counters are KiB; FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B); per launch = sum / launches",
                "note": sys.argv[5] if len(sys.argv) > 5 else ""}
json.dump(res, open(sys.argv[3], "w"), indent=1)
