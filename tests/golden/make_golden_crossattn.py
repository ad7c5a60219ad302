#!/usr/bin/env python3
"""tests/golden/cross_attention.npz: outputs and gradients of the REFERENCE's match.CrossAttention
(/root/reference/models/match.py:63-88: nn.MultiheadAttention + residual + LayerNorm + Linear(E,1) + sigmoid) on small
seeded cases, in fp32 and with an fp64 twin, for both parameter layouts (kvdim = None -> packed in_proj_weight; kvdim != E
-> separate q/k/v_proj_weight); asserts that oracle.tag_oracle.match_cross_attention equals the reference.  dropout = 0
(the reference's torch dropout stream cannot be replayed; the HIP path's own masks are checked against the oracle on the
GPU).  Build container only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
from oracle import tag_oracle as O  # noqa: E402

ref_import.install()
from models.match import CrossAttention  # noqa: E402  (the reference)

out = {}
for case, (E, H, Dk) in {"packed": (64, 2, None), "kv": (64, 4, 96), "wide": (128, 2, None)}.items():   # head dims 32, 16, 64
    B, T, L = 3, 13, 5
    g = torch.Generator().manual_seed(31 + E)
    torch.manual_seed(7 + E)
    ref = CrossAttention(E, H, 0.0, kvdim=Dk)
    with torch.no_grad():
        for p in ref.parameters():                                     # a livelier state than the default init
            p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.ndim > 1 else 0.5))
        ref.norm.weight.add_(1.0)
    audio = torch.randn(B, T, E, generator=g)
    token = torch.randn(B, L, Dk or E, generator=g)
    text_len = torch.tensor([5, 1, 3])
    dsim = torch.randn(B, T, generator=g)
    st = {"match_fn." + k: v.detach().clone() for k, v in ref.state_dict().items()}
    out[f"{case}/cfg"] = np.array([E, H, Dk or E, B, T, L])
    out[f"{case}/audio"], out[f"{case}/token"] = audio.numpy(), token.numpy()
    out[f"{case}/text_len"], out[f"{case}/dsim"] = text_len.numpy(), dsim.numpy()
    for k, v in st.items():
        out[f"{case}/w/{k}"] = v.numpy()
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        m = CrossAttention(E, H, 0.0, kvdim=Dk).to(dt)
        m.load_state_dict({k: v.to(dt) for k, v in ref.state_dict().items()})
        a = audio.detach().clone().to(dt).requires_grad_(True)
        t = token.detach().clone().to(dt).requires_grad_(True)
        sim = m({"audio_emb": a, "text_emb": {"token_emb": t}, "text_len": text_len})
        sim.backward(dsim.to(dt))
        out[f"{case}/sim_{tag}"] = sim.detach().numpy()
        if dt == torch.float64:                        # gradients: the fp64 twin, stored as fp32 (fixture size)
            out[f"{case}/daudio"], out[f"{case}/dtoken"] = a.grad.float().numpy(), t.grad.float().numpy()
            for n, p in m.named_parameters():
                out[f"{case}/grad/match_fn.{n}"] = p.grad.float().numpy()
        else:
            out[f"{case}/grad_floor"] = np.array(0.0)
            f32_grads = {n: p.grad.double() for n, p in m.named_parameters()}
            f32_grads["daudio"], f32_grads["dtoken"] = a.grad.double(), t.grad.double()
        so = O.match_cross_attention({k: v.to(dt) for k, v in st.items()}, audio.to(dt), token.to(dt), text_len, H)
        err = (so - sim).abs().max().item()
        print(f"{case} {tag}: oracle vs reference {err:.2e}; sim range [{sim.min().item():.3f}, {sim.max().item():.3f}]")
        assert err < (2e-6 if dt == torch.float32 else 1e-12)
    # the reference's own fp32-vs-fp64 distance, max over all gradient tensors (max-normalised): the floor of the test
    worst = 0.0
    for n, p in m.named_parameters():
        worst = max(worst, ((f32_grads[n] - p.grad).abs().max() / (p.grad.abs().max() + 1e-300)).item())
    out[f"{case}/grad_floor"] = np.array(worst)
    print(f"{case}: reference fp32-vs-fp64 gradient floor {worst:.2e}")
np.savez_compressed(os.path.join(HERE, "cross_attention.npz"), **out)
print("wrote cross_attention.npz", os.path.getsize(os.path.join(HERE, "cross_attention.npz")))
