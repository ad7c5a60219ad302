#!/usr/bin/env python3
"""Per-layer A/B of the one-read pool backward (GPU box): the dgrad conv with / without the pool-backward sums in its epilogue,
and the pool backward with / without its own reduction pass, at the three Cnn8Rnn shapes where the fusion applies (B = 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texttoaudiogrounding_amd import ops

dev = torch.device("cuda:0")
B = 64


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# (Hf, Wf, C of the lower block, ph) ; the conv above: Cin_above -> C at (Hf/ph, Wf/2)
for Hf, Wf, C, ph, Cabove in ((1001, 64, 64, 2, 128), (500, 32, 128, 2, 256), (250, 16, 256, 1, 512)):
    H, W = Hf // ph, Wf // 2
    y = torch.randn(B, Hf, Wf, C, device=dev)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    st = ops.bn_stats(y.view(-1, C), gamma, beta, None, None, True)
    dyab = torch.randn(B, H, W, Cabove, device=dev)
    w = torch.randn(Cabove, C, 3, 3, device=dev) * 0.05
    _, wd = ops.pack_conv_weight(w, W=W)
    t_plain = timeit(lambda: ops.conv3x3(dyab, wd, C))
    t_fused = timeit(lambda: ops.conv3x3_dgrad_poolsums(dyab, wd, y, st, ph, 2, 0.2, 1234))
    dx, part = ops.conv3x3_dgrad_poolsums(dyab, wd, y, st, ph, 2, 0.2, 1234)
    t_two = timeit(lambda: ops.bnrelu_pool_backward(y, st, gamma, dx, ph, 2, 0.2, 1234))
    t_apply = timeit(lambda: ops.bnrelu_pool_backward(y, st, gamma, dx, ph, 2, 0.2, 1234, partials=part))
    print(f"block below {Hf}x{Wf}x{C} window {ph}x2 | dgrad {Cabove}->{C} at {H}x{W}: plain {t_plain:.3f} ms, with sums {t_fused:.3f} ms "
          f"(+{t_fused - t_plain:.3f}) | pool backward: two-pass {t_two:.3f} ms, apply + fold {t_apply:.3f} ms (-{t_two - t_apply:.3f}) "
          f"| net {t_fused - t_plain - (t_two - t_apply):+.3f} ms", flush=True)
