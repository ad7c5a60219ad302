#!/usr/bin/env python3
"""Phase clocks of ONE workgroup's wave 0 of the row-streaming conv kernel (library built by tools/run_rows_prof.sh):
    TAG_HIP_LIB=texttoaudiogrounding_amd/libtag_rowsprof.so python tools/conv_rows_prof.py      (GPU box, repo root)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texttoaudiogrounding_amd import ops, lib
ops.CONV_MATH, ops.ACT_DTYPE = "bf16", "bf16"
dev = torch.device("cuda:0")
L = ctypes.CDLL(lib.LIB_PATH)
B = 64
names = ["barriers", "MFMA phase", "wait DMA", "transform", "pack+window+stats", "output stores", "DMA issue", "-"]
for (H, W, Cin, Cout, pro) in [(1001, 64, 64, 64, 1), (1001, 64, 64, 64, 0), (500, 32, 64, 128, 0)]:
    x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    s, t = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
    wf, wd = ops.pack_conv_weight(w, W=W)
    for _ in range(3):
        ops.conv3x3_stats(x, wf, Cout, pro, s if pro else None, t if pro else None, want_stats=True)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 12)()
    L.tag_debug_get_rows_prof(buf)
    v = list(buf)[:8]
    tot, T = sum(v), max(1, int(buf[8]))
    print(f"{H}x{W} {Cin}->{Cout} pro={pro}: workgroup {buf[9]} clk = {buf[10] / 100.0:.1f} us ({buf[9] / max(1, buf[10]) / 10.0:.2f} GHz); "
          f"{T} rows of wave 0 (its group: every other row), {tot / T:.0f} clk per row pair  " +
          "  ".join(f"{n} {x_ / T:.0f} ({100 * x_ / tot:.0f}%)" for n, x_ in zip(names, v)))
