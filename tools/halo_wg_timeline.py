#!/usr/bin/env python3
"""Lifetimes of EVERY workgroup of one fp32 halo-conv launch (library built with -DTAG_HALO_PROF): start / end on the 100 MHz
realtime clock and the CU each one ran on -> mean life, the occupancy of a CU over the launch, the gaps between successive
workgroups of a CU slot.     bash tools/run_halo_prof.sh timeline        (GPU box, from the repo root)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from texttoaudiogrounding_amd import ops, lib
dev = torch.device("cuda:0")
L = ctypes.CDLL(lib.LIB_PATH)
B = 64
for (H, W, Cin, Cout, pro) in [(1001, 64, 64, 64, 1), (1001, 64, 64, 64, 0), (500, 32, 128, 128, 1), (250, 8, 512, 512, 1)]:
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    s, t = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
    wf, wd = ops.pack_conv_weight(w, W=W)
    for _ in range(4):
        ops.conv3x3_stats(x, wf, Cout, pro, s if pro else None, t if pro else None, want_stats=True)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (4 * 65536))()
    L.tag_debug_get_halo_wg(buf)
    a = np.frombuffer(buf, dtype=np.uint64).copy()
    th = 128 // W
    n = min(65536, B * ((H + th - 1) // th) * max(1, Cout // (128 if Cout >= 128 else 64)))
    st = a[:n].astype(np.int64)
    en = a[65536:65536 + n].astype(np.int64)
    ident = a[2 * 65536:2 * 65536 + n].astype(np.int64)
    # CU identity: XCC (bits 16..19) | SE (13..15) | SH (12) | CU (8..11); slot = that + SIMD/wave slot of wave 0 (bits 0..5)
    cu = ident & 0xfff00 | 0
    first = a[3 * 65536:3 * 65536 + n].astype(np.int64)
    life = (en - first) / 100.0                                        # us, first instruction -> after the epilogue
    pre = (st - first) / 100.0
    order = np.argsort(first)
    print(f"   first instruction -> pipeline start (index arithmetic, geometry, BN table): mean {pre.mean():.2f} us; the 768 workgroups of the "
          f"first residency round (idle chip) {pre[order[:768]].mean():.2f} us, the rest {pre[order[768:]].mean():.2f} us")
    t0, t1 = first.min(), en.max()
    span = (t1 - t0) / 100.0
    print(f"{H}x{W} {Cin}->{Cout}: {n} workgroups, launch span {span:.0f} us, life mean {life.mean():.1f} us  p10 {np.percentile(life, 10):.1f}  "
          f"p50 {np.percentile(life, 50):.1f}  p90 {np.percentile(life, 90):.1f}")
    cus = np.unique(cu)
    busy = np.array([life[cu == c].sum() for c in cus])
    print(f"   {len(cus)} CUs seen; workgroup-time per CU / span = mean {busy.mean() / span:.2f} (3.0 = three resident the whole launch), "
          f"min {busy.min() / span:.2f}, max {busy.max() / span:.2f}")
    # a CU's timeline: sort by start, resident count over time
    c = cus[len(cus) // 2]
    m = cu == c
    ev = sorted([(v, 1) for v in first[m]] + [(v, -1) for v in en[m]])
    cur, last, acc = 0, ev[0][0], {}
    for tt, d in ev:
        acc[cur] = acc.get(cur, 0) + (tt - last)
        cur += d
        last = tt
    tot = sum(acc.values())
    print("   one CU: share of its active span with k resident workgroups: " + "  ".join(f"{k}: {v / tot:.2f}" for k, v in sorted(acc.items())))
