#!/usr/bin/env python3
"""Fused Winograd forward (csrc/conv_wino_fused.hip; prologue 1 + BatchNorm statistics) on the layer shapes of the step at batch B:
ms per launch and executed TFLOP/s (2 * 16 * T * Cin * Cout).  `TAG_HIP_LIB=<variant .so> TAG_ALLOW_STALE_LIB=1` times an ablation
build (tools/wino_fused_abl.sh).    python tools/wino_fused_bench.py [B] [shape index ...]"""
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


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = []
for i in sel:
    H, W, Cin, Cout = SHAPES[i]
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * (2.0 / (9 * Cin)) ** 0.5
    scale, shift = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.3
    uf = torch.empty(16, Cin, Cout, device=dev)
    call("tag_pack_conv_weight_wino", ptr(w), ptr(uf), None, Cin, Cout)
    P = query("tag_conv3x3_wino_stats_rows", B, H, W, Cout)
    y = torch.empty(B, H, W, Cout, device=dev)
    st = torch.zeros(P * (3 * Cout + 1), device=dev)
    ws = torch.empty(16, device=dev)
    t = timeit(lambda: call("tag_conv3x3_wino_forward", ptr(x), ptr(uf), 1, ptr(scale), ptr(shift), ptr(y), ptr(st), B, H, W, Cin, Cout,
                            ptr(ws), None))
    fl = 2.0 * 16 * B * ((H + 1) // 2) * ((W + 1) // 2) * Cin * Cout
    out.append(f"{H}x{W} {Cin}->{Cout}: {t:.3f} ms {fl / t * 1e-9:.1f} TF/s")
    del x, y
print(os.environ.get("TAG_HIP_LIB", "product"), "|", " | ".join(out))
ops.check_async_errors()
