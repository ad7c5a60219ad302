#!/usr/bin/env python3
"""Times the small (non-conv) kernels of the B=64 training step in isolation with HIP events (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texttoaudiogrounding_amd import ops
from oracle import tag_oracle as O

dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


which = sys.argv[1:] or ["logmel"]
if "logmel" in which:
    for kind, S in (("cnn8rnn", 320000), ("crnn", 320000)):
        p = O.FRONTEND[kind]
        window, fb = O.frontend_tables(kind)
        wave = 0.1 * torch.randn(64, S, device=dev)
        w, f = window.to(dev), fb.to(dev)
        us = timeit(lambda: ops.logmel(wave, p["n_fft"], p["win_length"], p["hop_length"], w, f))
        print(f"logmel[{kind}] B=64 x 10 s: {us:.1f} us")
if "gru" in which:
    import math
    B, T, I, H = 64, 250, 512, 256
    g = torch.Generator().manual_seed(0)
    k = 1 / math.sqrt(H)
    rnn = []
    for _ in range(2):
        rnn += [((torch.rand(3 * H, I, generator=g) * 2 - 1) * k).to(dev), ((torch.rand(3 * H, H, generator=g) * 2 - 1) * k).to(dev),
                ((torch.rand(3 * H, generator=g) * 2 - 1) * k).to(dev), ((torch.rand(3 * H, generator=g) * 2 - 1) * k).to(dev)]
    x = torch.randn(B * T, I, device=dev)
    y, sv = ops.gru_bidir_forward(x, rnn, B, T, True)
    dy = torch.randn_like(y)
    print(f"GRU fwd (proj GEMM + persistent kernel) B=64 T=250: {timeit(lambda: ops.gru_bidir_forward(x, rnn, B, T, True)):.1f} us")
    print(f"GRU bwd (persistent kernel + GEMMs): {timeit(lambda: ops.gru_bidir_backward(dy, x, sv)):.1f} us")
if "c1" in which:
    B, H = 64, 1001
    x = torch.randn(B, H, 64, device=dev)
    dy = torch.randn(B, H, 64, 64, device=dev)
    w = torch.randn(64, 1, 3, 3, device=dev) * 0.1
    cs, ct = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
    us = timeit(lambda: ops.conv3x3_c1_backward(x, dy, w, cs, ct))
    print(f"conv_c1 backward (fp32 dy 1.05 GB): {us:.1f} us = {dy.numel() * 4 / us / 1e6:.2f} TB/s")
    dyb = dy.bfloat16()
    us = timeit(lambda: ops.conv3x3_c1_backward(x, dyb, w, cs, ct))
    print(f"conv_c1 backward (bf16 dy 0.52 GB): {us:.1f} us = {dyb.numel() * 2 / us / 1e6:.2f} TB/s")
