#!/usr/bin/env python3
"""Time PyTorch-ROCm's own (MIOpen) fp32 3x3 convolutions on the Cnn8Rnn layer shapes at B=64, 10 s clips:
what the reference's nn.Conv2d would run on this GPU.  Comparison point only (not part of the product or the tests)."""
import sys
import time
import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
shapes = [(1001, 64, 64, 64), (500, 32, 64, 128), (500, 32, 128, 128), (250, 16, 128, 256), (250, 16, 256, 256),
          (250, 8, 256, 512), (250, 8, 512, 512)]
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
for fmt in (torch.contiguous_format, torch.channels_last):
    for k in tot:
        tot[k] = 0.0
    for (H, W, Ci, Co) in shapes:
        x = torch.randn(B, Ci, H, W, device=dev).contiguous(memory_format=fmt)
        w = torch.randn(Co, Ci, 3, 3, device=dev).contiguous(memory_format=fmt)
        dy = torch.randn(B, Co, H, W, device=dev).contiguous(memory_format=fmt)
        flop = 2.0 * B * H * W * 9 * Ci * Co
        res = {}
        for name, fn in (("fwd", lambda: F.conv2d(x, w, None, 1, 1)),
                         ("dgrad", lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])),
                         ("wgrad", lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))):
            t0 = time.time()
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            tune = time.time() - t0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            res[name] = ms
            tot[name] += ms
            print(f"{'NHWC' if fmt == torch.channels_last else 'NCHW'} {H}x{W} {Ci}->{Co} {name:5s} {ms:8.3f} ms  {flop / ms / 1e9:7.1f} TFLOP/s  (first calls {tune:.1f} s)", flush=True)
    print(f"== {'NHWC' if fmt == torch.channels_last else 'NCHW'} totals: " + ", ".join(f"{k} {v:.2f} ms" for k, v in tot.items()), flush=True)
