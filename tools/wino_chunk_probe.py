#!/usr/bin/env python3
"""Does cutting a Winograd launch into batch chunks whose transform planes fit the 256 MB Infinity Cache pay?  The 512 -> 512 forward
(no statistics) at B = 64 as one launch and as 2 / 4 / 8 / 16 chunks sharing one workspace."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from texttoaudiogrounding_amd import ops  # noqa: E402
from texttoaudiogrounding_amd.ops import call, ptr, query  # noqa: E402

dev = torch.device("cuda:0")
B = 64
for (H, W, Cin, Cout) in [(250, 8, 512, 512), (250, 16, 256, 256)]:
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.01
    uf = torch.empty(16, Cin, Cout, device=dev)
    call("tag_pack_conv_weight_wino", ptr(w), ptr(uf), None, Cin, Cout)
    y = torch.empty(B, H, W, Cout, device=dev)
    for nch in (1, 2, 4, 8, 16):
        nb = B // nch
        ws = torch.empty(query("tag_conv3x3_wino_ws_bytes", nb, H, W, Cin, Cout) // 4, device=dev)

        def run():
            for c in range(nch):
                call("tag_conv3x3_wino_forward", ptr(x[c * nb:(c + 1) * nb]), ptr(uf), 0, None, None, ptr(y[c * nb:(c + 1) * nb]), None, nb,
                     H, W, Cin, Cout, ptr(ws), None)
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(f"{H}x{W} {Cin}->{Cout}: {nch:2d} chunk(s) of {nb} clips ({ws.numel() * 4 / 2**20:.0f} MB planes): {e0.elapsed_time(e1) / 10:.3f} ms")
