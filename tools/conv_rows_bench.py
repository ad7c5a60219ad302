#!/usr/bin/env python3
"""A/B of the bf16-storage conv launches that conv_rows.hip (row-streaming kernel, weights in registers, LDS-DMA ring) takes
against the tile kernel of conv_x3.hip, on the Cnn8Rnn layer shapes at B = 64, plus a bit-level comparison of the two kernels'
outputs / statistics on the same inputs.   python tools/conv_rows_bench.py [--quick]      (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texttoaudiogrounding_amd import ops
from texttoaudiogrounding_amd.lib import query

ops.CONV_MATH, ops.ACT_DTYPE = "bf16", "bf16"
dev = torch.device("cuda:0")
B = 8 if "--quick" in sys.argv else 64
# (H, W, Cin, Cout, kind): kind f0 = forward prologue 0 + statistics, f1 = forward prologue 1 + statistics,
# d = dgrad + BatchNorm-backward sums, p = plain dgrad (no epilogue)
LAUNCHES = [(1001, 64, 64, 64, "f1"), (1001, 64, 64, 64, "d"), (500, 32, 64, 128, "f0")]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {0: 0.0, 1: 0.0}
for (H, W, Cin, Cout, kind) in LAUNCHES:
    torch.manual_seed(H + Cin + Cout)
    x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * (1.0 / (3.0 * Cin ** 0.5))
    s, t = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
    wf, _ = ops.pack_conv_weight(w, W=W)
    if kind == "d":
        yref = torch.randn(B, H, W, Cout, device=dev).bfloat16()
        gamma, beta = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
        st = ops.BNStat()
        st.train = True
        st.mean, st.invstd = torch.randn(Cout, device=dev) * 0.1, torch.rand(Cout, device=dev) + 0.5
        st.scale, st.shift = gamma * st.invstd, beta - st.mean * gamma * st.invstd
        fn = lambda: ops.conv3x3_dgrad_bnrelu_backward(x, wf, yref, st, gamma, defer_apply=True)
    elif kind == "p":
        fn = lambda: (ops.conv3x3(x, wf, Cout),)
    else:
        pro = int(kind[1])
        fn = lambda: ops.conv3x3_stats(x, wf, Cout, pro, s if pro else None, t if pro else None, want_stats=True)
    res, us = {}, {}
    for on in (0, 1):
        query("tag_conv_rows_enable", on)
        out = fn()
        torch.cuda.synchronize()
        if kind in ("f0", "f1"):
            y, part = out
            bst = ops.bn_stats(y.view(-1, Cout), torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev), None, None, True,
                               partials=part)
            res[on] = (y.float(), bst.mean.clone(), bst.invstd.clone())
        elif kind == "d":
            res[on] = (out[0].float(), out[1].clone(), out[2].clone())
        else:
            res[on] = (out[0].float(),)
        us[on] = timeit(fn)
        tot[on] += us[on]
    query("tag_conv_rows_enable", 1)
    diffs = []
    for a, b_ in zip(res[0], res[1]):
        den = a.abs().max().item() + 1e-30
        diffs.append((a - b_).abs().max().item() / den)
    nbad = int((res[0][0] != res[1][0]).sum().item())
    fl = 2.0 * B * H * W * 9 * Cin * Cout
    print(f"{H:5d}x{W:2d} {Cin:3d}->{Cout:3d} {kind:2s}: tile {us[0]:7.1f} us ({fl / us[0] / 1e6:6.0f} TF/s)   rows {us[1]:7.1f} us "
          f"({fl / us[1] / 1e6:6.0f} TF/s)   x{us[0] / us[1]:.2f}   max rel diff {' '.join(f'{d:.1e}' for d in diffs)}  "
          f"(outputs differing: {nbad} of {res[0][0].numel()})", flush=True)
print(f"total: tile {tot[0]:.0f} us, rows {tot[1]:.0f} us")
