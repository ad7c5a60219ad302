#!/usr/bin/env python3
"""Block 1, first conv (Cin = 1) backward at the bench shape: the two-pass form (tag_bnrelu_backward_apply over da, then
tag_conv3x3_c1_backward) against the fused one (tag_conv3x3_c1_backward_bnrelu), fp32 and bf16 storage.

    python tools/c1_bwd_bench.py [B] [frames]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from texttoaudiogrounding_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1001
dev = torch.device("cuda:0")
torch.manual_seed(0)
W = C = 64
x = torch.randn(B, H, W, device=dev) * 10 - 30
cs, ct = torch.rand(W, device=dev) * 0.1 + 0.05, torch.randn(W, device=dev)
w = torch.randn(C, 1, 3, 3, device=dev) / 3
gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3


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
    return e0.elapsed_time(e1) / n


for dt in (torch.float32, torch.bfloat16):
    y = (torch.randn(B, H, W, C, device=dev) * 2 + 0.5).to(dt)
    da = torch.randn(B, H, W, C, device=dev).to(dt)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    st = ops.bn_stats(y.float().view(-1, C), gamma, beta, rm, rv, True, 1e-5, 0.1)
    _, dg, db = ops.bnrelu_backward(y, st, gamma, da, inplace=False)
    sfx = "_bf16" if dt == torch.bfloat16 else ""
    dyb = torch.empty_like(da)

    def apply_():
        ops.call("tag_bnrelu_backward_apply" + sfx, ops.ptr(y), ops.ptr(st.scale), ops.ptr(st.shift), ops.ptr(st.mean),
                 ops.ptr(st.invstd), ops.ptr(gamma), ops.ptr(da), ops.ptr(dyb), ops.ptr(dg), ops.ptr(db), B * H * W, C, 1)
    t_apply = timeit(apply_)
    t_plain = timeit(lambda: ops.conv3x3_c1_backward(x, dyb, w, cs, ct))
    t_fused = timeit(lambda: ops.conv3x3_c1_backward(x, da, w, cs, ct, bn_bwd=(y, st, gamma, dg, db)))
    print(f"{str(dt):16s} apply {t_apply:.3f} ms + c1_backward {t_plain:.3f} ms = {t_apply + t_plain:.3f} ms;  fused {t_fused:.3f} ms")
ops.check_async_errors()
