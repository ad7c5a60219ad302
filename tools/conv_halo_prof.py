#!/usr/bin/env python3
"""Phase clocks of ONE workgroup of the exact-fp32 halo conv kernel (library built with -DTAG_HALO_PROF):
    bash tools/run_halo_prof.sh   (GPU box, from the repo root)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texttoaudiogrounding_amd import ops, lib
dev = torch.device("cuda:0")
L = ctypes.CDLL(lib.LIB_PATH)
B = 64
for (H, W, Cin, Cout) in [(1001, 64, 64, 64), (500, 32, 128, 128), (250, 16, 256, 256), (250, 8, 512, 512)]:
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    s, t = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
    wf, wd = ops.pack_conv_weight(w, W=W)
    for _ in range(int(os.environ.get("TAG_PROF_REPS", "2"))):
        ops.conv3x3_stats(x, wf, Cout, 1, s, t, want_stats=True)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 12)()
    L.tag_debug_get_halo_prof(buf)
    names = ["prologue", "MFMA taps", "barrier", "operand stores", "barrier", "output stores", "statistics"]
    v = list(buf)[:7]
    tot = sum(v)
    print(f"   prologue of a sampled workgroup: loads issued at {buf[7]}, barrier + loads landed at {buf[8]}, operands stored at {buf[9]} clk")
    if buf[10]:
        print(f"   shader clock over that workgroup's life: {tot / (buf[10] / 100e6) / 1e9:.2f} GHz ({tot} clk in {buf[10] / 100:.1f} us)")
    print(f"{H}x{W} {Cin}->{Cout}: total {tot} clk  " + "  ".join(f"{n} {x_} ({100 * x_ / tot:.0f}%)" for n, x_ in zip(names, v)))
