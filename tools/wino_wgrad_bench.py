#!/usr/bin/env python3
"""Fused Winograd weight gradient (csrc/conv_wino_fused.hip, prologue 1) on the step's layer shapes at batch B: ms per launch and
executed TFLOP/s; `TAG_HIP_LIB=<variant .so> TAG_ALLOW_STALE_LIB=1` times an ablation build.  python tools/wino_wgrad_bench.py [B] [shape ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from texttoaudiogrounding_amd import ops  # noqa: E402
from texttoaudiogrounding_amd.ops import call, ptr, query  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
SHAPES = [(250, 8, 512, 512), (250, 8, 256, 512), (250, 16, 256, 256), (250, 16, 128, 256), (500, 32, 128, 128), (500, 32, 64, 128),
          (1001, 64, 64, 64)]
sel = [int(a) for a in sys.argv[2:]] or range(len(SHAPES))
out = []
for i in sel:
    H, W, Cin, Cout = SHAPES[i]
    x = torch.randn(B, H, W, Cin, device=dev)
    dy = torch.randn(B, H, W, Cout, device=dev)
    scale, shift = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.3
    dw = torch.empty(Cout, Cin, 3, 3, device=dev)
    ws = torch.empty(query("tag_conv3x3_wino_wgrad_ws_bytes", B, H, W, Cin, Cout) // 4 + 16, device=dev)
    fn = lambda: call("tag_conv3x3_wino_wgrad", ptr(x), 1, ptr(scale), ptr(shift), ptr(dy), ptr(dw), B, H, W, Cin, Cout, ptr(ws), None)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    fl = 2.0 * 16 * B * ((H + 1) // 2) * ((W + 1) // 2) * Cin * Cout
    out.append(f"{H}x{W} {Cin}->{Cout}: {t:.3f} ms {fl / t * 1e-9:.1f} TF/s")
    del x, dy
print(os.environ.get("TAG_HIP_LIB", "product"), "|", " | ".join(out))
ops.check_async_errors()
