#!/usr/bin/env python3
"""Ablation of the conv forward kernel (guide: 'ablate before optimising'): builds private copies of conv.hip with
-DTAG_ABLATE=n (on the GPU box) and times one layer shape.   python tools/ablate_conv.py"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "texttoaudiogrounding_amd", "csrc")
OUT = "/tmp/ablate"
os.makedirs(OUT, exist_ok=True)
VARIANTS = {0: "baseline (2 LDS buffers, 2 WG/CU)", 100: "1 LDS buffer, 3 WG/CU", 0.5: "baseline again", 100.5: "1 buffer again"}
shapes = [(64, 250, 16, 256, 256), (64, 1001, 64, 64, 64)]
dev = torch.device("cuda:0")
for n, label in VARIANTS.items():
    so = f"{OUT}/conv_{n}.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-fno-fast-math", "-ffp-contract=off", f"-DTAG_ABLATE={int(n) % 100}", f"-DTAG_NBUF={1 if n >= 100 else 2}", os.path.join(CSRC, "conv.hip"),
                           os.path.join(CSRC, "tag_lib.hip"), "-o", so], stderr=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    P, I = ctypes.c_void_p, ctypes.c_int
    lib.tag_conv3x3_forward.argtypes = [P, P, I, P, P, P, P, I, I, I, I, I, P]
    lib.tag_pack_conv_weight.argtypes = [P, P, P, I, I, P]
    line = f"{label:34s}"
    for (B, H, W, Cin, Cout) in shapes:
        x = torch.randn(B, H, W, Cin, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
        wf = torch.empty(9, Cin, Cout, device=dev)
        y = torch.empty(B, H, W, Cout, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        lib.tag_pack_conv_weight(w.data_ptr(), wf.data_ptr(), None, Cin, Cout, st)
        run = lambda: lib.tag_conv3x3_forward(x.data_ptr(), wf.data_ptr(), 0, None, None, y.data_ptr(), None, B, H, W, Cin, Cout, st)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += f" | {Cin}->{Cout}@{H}x{W}: {ms:6.3f} ms {2.0 * B * H * W * 9 * Cin * Cout / ms / 1e9:6.1f} TF"
    print(line, flush=True)
