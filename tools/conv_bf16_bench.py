#!/usr/bin/env python3
"""Times the one-product bf16-storage conv kernels (forward orientation, with BatchNorm statistics) on the Cnn8Rnn layer shapes.
    python tools/conv_bf16_bench.py [--lib path/to/libtag_hip.so]   (GPU box)"""
import os, sys
if "--lib" in sys.argv:
    os.environ["TAG_HIP_LIB"] = sys.argv[sys.argv.index("--lib") + 1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texttoaudiogrounding_amd import ops

ops.CONV_MATH, ops.ACT_DTYPE = "bf16", "bf16"
dev = torch.device("cuda:0")
SHAPES = [(1001, 64, 64, 64), (500, 32, 64, 128), (500, 32, 128, 128), (250, 16, 128, 256), (250, 16, 256, 256), (250, 8, 256, 512),
          (250, 8, 512, 512)]
B = 64
tot = 0.0
for (H, W, Cin, Cout) in SHAPES:
    x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    s, t = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
    wf, wd = ops.pack_conv_weight(w, W=W)
    for pro in (0, 1):
        fn = lambda: ops.conv3x3_stats(x, wf, Cout, pro, s if pro else None, t if pro else None, want_stats=True)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        tot += us
        print(f"{H}x{W} {Cin}->{Cout} pro={pro}: {us:7.1f} us  {2.0 * B * H * W * 9 * Cin * Cout / us / 1e6:7.1f} TFLOP/s")
print(f"total {tot:.0f} us")
