#!/usr/bin/env python3
"""Phase clocks of ONE workgroup of the one-product bf16 conv kernel (library built with -DTAG_X3_PROF):
    bash tools/run_x3_prof.sh   (GPU box, from the repo root)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texttoaudiogrounding_amd import ops, lib
ops.CONV_MATH, ops.ACT_DTYPE = "bf16", "bf16"
dev = torch.device("cuda:0")
L = ctypes.CDLL(lib.LIB_PATH)
B = 64
for (H, W, Cin, Cout) in [(1001, 64, 64, 64), (500, 32, 128, 128), (250, 16, 256, 256), (250, 8, 512, 512)]:
    x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    s, t = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
    wf, wd = ops.pack_conv_weight(w, W=W)
    for _ in range(3):
        ops.conv3x3_stats(x, wf, Cout, 1, s, t, want_stats=True)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 10)()
    L.tag_debug_get_x3_prof(buf)
    names = ["prologue", "MFMA loop", "barrier", "store_patch", "barrier", "pre-epilogue barrier", "pack + LDS write", "16-B stores", "statistics"]
    v = list(buf)[:9]
    tot = sum(v)
    print(f"{H}x{W} {Cin}->{Cout}: total {tot} clk  " + "  ".join(f"{n} {x_} ({100 * x_ / tot:.0f}%)" for n, x_ in zip(names, v)))

print("wgrad (one-product, bf16 storage):")
for (H, W, Cin, Cout) in [(1001, 64, 64, 64), (500, 32, 128, 128), (250, 16, 256, 256), (250, 8, 512, 512)]:
    x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
    dy = torch.randn(B, H, W, Cout, device=dev).bfloat16()
    s, t = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
    for _ in range(2):
        ops.conv3x3_wgrad(x, dy, 1, s, t)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 10)()
    L.tag_debug_get_x3_prof(buf)
    names = ["prologue", "issue next-next loads", "MFMA chunk", "barrier", "strip rebuild / window", "epilogue (partial stores)", "wait for the next chunk's loads", "prologue math + LDS stores"]
    v = list(buf)[:8]
    tot = sum(v)
    print(f"{H}x{W} {Cin}->{Cout} ({buf[8]} chunks of 64 px): total {tot} clk  " + "  ".join(f"{n} {x_} ({100 * x_ / tot:.0f}%)" for n, x_ in zip(names, v)))
