#!/usr/bin/env python3
"""Micro-benchmark of the MFMA conv kernels at the Cnn8Rnn layer shapes (HIP events, in-process).
    python tools/conv_bench.py [--batch 64] [--reps 5] [--only fwd|wgrad]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from texttoaudiogrounding_amd import ops  # noqa: E402

SHAPES = [  # (H, W, Cin, Cout) of every 3x3 conv with Cin >= 32, forward orientation
    (1001, 64, 64, 64), (500, 32, 64, 128), (500, 32, 128, 128), (250, 16, 128, 256), (250, 16, 256, 256),
    (250, 8, 256, 512), (250, 8, 512, 512)]


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--math", default="fp32", help="fp32 | x3 (bf16x3-split MFMA forward/dgrad)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.batch
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    for (H, W, Cin, Cout) in SHAPES:
        x = torch.randn(B, H, W, Cin, device=dev)
        dy = torch.randn(B, H, W, Cout, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
        s, t = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
        ops.CONV_MATH = a.math
        wf, wd = ops.pack_conv_weight(w, W=W)
        flop = 2.0 * B * H * W * 9 * Cin * Cout
        res = {}
        if a.only in ("", "fwd"):
            res["fwd"] = timeit(lambda: ops.conv3x3(x, wf, Cout, 1, s, t), a.reps)
            res["dgrad"] = timeit(lambda: ops.conv3x3(dy, wd, Cin), a.reps)
        if a.only in ("", "wgrad"):
            res["wgrad"] = timeit(lambda: ops.conv3x3_wgrad(x, dy, 1, s, t), a.reps)
        line = f"{H:5d}x{W:<3d} {Cin:4d}->{Cout:<4d} {flop / 1e9:8.1f} GFLOP |"
        for k, ms in res.items():
            line += f" {k} {ms:7.3f} ms {flop / ms / 1e9:6.1f} TF |"
            tot[k][0] += ms
            tot[k][1] += flop
        print(line, flush=True)
        del x, dy
    for k, (ms, fl) in tot.items():
        if ms:
            print(f"TOTAL {k}: {ms:.2f} ms, {fl / ms / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
