#!/usr/bin/env python3
"""A/B of the bf16-storage weight-gradient kernels on the Cnn8Rnn layer shapes at B = 64: conv_wgrad_dma.hip (operands by
LDS-DMA) against the register-staged kernel of conv_x3.hip; interleaved rounds, the two results compared.
    python tools/conv_wgrad_bench.py [--quick]      (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texttoaudiogrounding_amd import ops
from texttoaudiogrounding_amd.lib import query

ops.CONV_MATH, ops.ACT_DTYPE = "bf16", "bf16"
dev = torch.device("cuda:0")
B = 8 if "--quick" in sys.argv else 64
SHAPES = [(1001, 64, 64, 64, 1), (500, 32, 64, 128, 0), (500, 32, 128, 128, 1), (250, 16, 128, 256, 0), (250, 16, 256, 256, 1),
          (250, 8, 256, 512, 0), (250, 8, 512, 512, 1)]


def run(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {0: 0.0, 1: 0.0}
for (H, W, Cin, Cout, pro) in SHAPES:
    torch.manual_seed(H + Cin + Cout)
    x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
    dy = torch.randn(B, H, W, Cout, device=dev).bfloat16()
    s, t = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
    fn = lambda: ops.conv3x3_wgrad(x, dy, pro, s if pro else None, t if pro else None)
    res, best = {}, {0: 1e9, 1: 1e9}
    for on in (0, 1):
        query("tag_wgrad_dma_enable", 2 * on)
        res[on] = fn().clone()
    for rnd in range(4):                                   # interleaved rounds, best of
        for on in (0, 1):
            query("tag_wgrad_dma_enable", 2 * on)
            run(fn, 3)
            best[on] = min(best[on], run(fn, 8))
    query("tag_wgrad_dma_enable", 1)
    for on in (0, 1):
        tot[on] += best[on]
    d = (res[0] - res[1]).abs().max().item() / (res[0].abs().max().item() + 1e-30)
    fl = 2.0 * B * H * W * 9 * Cin * Cout
    print(f"{H:5d}x{W:2d} {Cin:3d}->{Cout:3d} pro={pro}: staged {best[0]:7.1f} us ({fl / best[0] / 1e6:6.0f} TF/s)   dma {best[1]:7.1f} us "
          f"({fl / best[1] / 1e6:6.0f} TF/s)   x{best[0] / best[1]:.2f}   max rel diff {d:.1e}", flush=True)
print(f"total: staged {tot[0]:.0f} us, dma {tot[1]:.0f} us")
