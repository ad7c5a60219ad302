#!/usr/bin/env python3
"""Times the attention core of match.CrossAttention (tag_mha_cross_forward / _backward) at the benched shape
(B = 64, T' = 250, E = 512, 8 heads, L = 4 tokens padded to the phrase length):  TAG_MHA_MFMA=0|1 python tools/mha_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texttoaudiogrounding_amd.lib import call, ptr, query
dev = torch.device("cuda:0")
for (B, T, L, E, H, p) in [(64, 250, 4, 512, 8, 0.1), (64, 250, 32, 512, 8, 0.1)]:
    g = torch.Generator(device=dev).manual_seed(1)
    q = torch.randn(B, T, E, device=dev, generator=g); k = torch.randn(B, L, E, device=dev, generator=g)
    v = torch.randn(B, L, E, device=dev, generator=g); dctx = torch.randn(B, T, E, device=dev, generator=g)
    klen = torch.randint(1, L + 1, (B,), device=dev, generator=g)
    attn = torch.empty(B, T, H, L, device=dev); ctx = torch.empty(B, T, E, device=dev)
    dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    ws = torch.empty(query("tag_mha_cross_backward_ws_bytes", B, T, L, E) // 4 + 4, device=dev)
    fwd = lambda: call("tag_mha_cross_forward", ptr(q), ptr(k), ptr(v), ptr(klen), ptr(attn), ptr(ctx), B, T, L, E, H, p, 7)
    bwd = lambda: call("tag_mha_cross_backward", ptr(q), ptr(k), ptr(v), ptr(attn), ptr(dctx), ptr(klen), ptr(dq), ptr(dk), ptr(dv),
                       B, T, L, E, H, p, 7, ptr(ws))
    for name, fn in (("forward", fwd), ("backward", bwd)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"MFMA={os.environ.get('TAG_MHA_MFMA', '1')} B={B} T={T} L={L} E={E} H={H}: {name} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us  "
              f"(checksum {float((ctx if name == 'forward' else dq).double().abs().sum()):.6e})")
