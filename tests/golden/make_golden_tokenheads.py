#!/usr/bin/env python3
"""tests/golden/token_heads.npz: the REFERENCE's match.ExpNegL2 / match.DotProduct with text_level="token"
(/root/reference/models/match.py:10-60; one text vector per frame, as a cross-encoder hands over) and its
MaxMarginRankingLoss with fix_norm=False (/root/reference/losses.py:226-264), outputs and gradients from the fp64 twin plus the
fp32 outputs.  Build container only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

ref_import.install()
from losses import MaxMarginRankingLoss  # noqa: E402  (the reference)
from models.match import DotProduct, ExpNegL2  # noqa: E402

out = {}
g = torch.Generator().manual_seed(77)
B, T, D = 3, 11, 96
audio, text, dsim = torch.randn(B, T, D, generator=g), torch.randn(B, T, D, generator=g), torch.randn(B, T, generator=g)
text[1, 3] = audio[1, 3]                              # a zero distance: the ExpNegL2 gradient is 0/0 there in the reference
out["audio"], out["text"], out["dsim"] = audio.numpy(), text.numpy(), dsim.numpy()
cases = {"expnegl2_norm": ExpNegL2(l2norm=True, text_level="token"), "expnegl2_raw": ExpNegL2(l2norm=False, text_level="token"),
         "dot_norm": DotProduct(l2norm=True, scale=True, text_level="token"),
         "dot_norm_noscale": DotProduct(l2norm=True, scale=False, text_level="token")}
for name, head in cases.items():
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        a, t = audio.to(dt).clone().requires_grad_(True), text.to(dt).clone().requires_grad_(True)
        sim = head({"audio_emb": a, "text_emb": {"token_emb": t}})
        sim.backward(dsim.to(dt))
        out[f"{name}/sim_{tag}"] = sim.detach().numpy()
        if dt == torch.float64:
            out[f"{name}/daudio"], out[f"{name}/dtext"] = a.grad.numpy(), t.grad.numpy()
    print(name, "sim range", out[f"{name}/sim_f64"].min(), out[f"{name}/sim_f64"].max(),
          "nan in grads:", np.isnan(out[f"{name}/daudio"]).sum())
for n, lam, margin in ((6, 0.7, 0.3), (9, 1.0, 1.0)):
    x = torch.randn(n, n, generator=g, dtype=torch.float64)
    for fix in (True, False):
        xx = x.clone().requires_grad_(True)
        loss = MaxMarginRankingLoss(margin=margin, fix_norm=fix, lamda1=lam)({"sim": xx})
        loss.backward()
        key = f"mm_n{n}_fix{int(fix)}"
        out[f"{key}/x"], out[f"{key}/cfg"] = x.numpy(), np.array([margin, lam])
        out[f"{key}/loss"], out[f"{key}/dx"] = loss.detach().numpy(), xx.grad.numpy()
        print(key, float(loss))
np.savez_compressed(os.path.join(HERE, "token_heads.npz"), **out)
print("wrote token_heads.npz", os.path.getsize(os.path.join(HERE, "token_heads.npz")))
