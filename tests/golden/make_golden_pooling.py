#!/usr/bin/env python3
"""tests/golden/sim_pooling.npz: outputs and input-gradients of EVERY reducer of the REFERENCE's models/sim_pooling.py
(twelve (B,B,T,N) reducers + MultiTextLinearSoft / MultiTextMax) and of the four *_with_lens pooling modes
MultiTextBiEncoder selects (models/utils.py:22-84, models/audio_text_model.py:205-215), imported from /root/reference, in
fp64; asserts that oracle.tag_oracle.sim_pooling / SEQ_POOL equal the reference.  Build container only."""
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
import models.sim_pooling as ref_pool  # noqa: E402  (the reference)
import models.utils as ref_utils  # noqa: E402

MODES = {"AudioMeanTextMean": ("mean", "mean"), "AudioMeanTextSum": ("mean", "sum"), "AudioMaxTextMean": ("max", "mean"),
         "AudioMaxTextMax": ("max", "max"), "AudioMaxTextSum": ("max", "sum"), "AudioMaxTextMeanSum": ("max", "mean_sum"),
         "AudioLinearSoftTextMean": ("linear_softmax", "mean"), "AudioLinearSoftTextSum": ("linear_softmax", "sum"),
         "AudioExpSoftTextMean": ("exp_softmax", "mean"), "AudioExpSoftTextSum": ("exp_softmax", "sum")}
B, T, N = 4, 13, 5
g = torch.Generator().manual_seed(23)
sim = torch.rand(B, B, T, N, generator=g, dtype=torch.float64) * 0.98 + 0.01
audio_len = torch.tensor([13, 7, 10, 1])          # max == T (max_with_lens builds its mask from max(lens))
text_len = torch.tensor([5, 1, 3, 4])             # max == N
dout = torch.randn(B, B, generator=g, dtype=torch.float64)
out = {"sim": sim.numpy(), "audio_len": audio_len.numpy(), "text_len": text_len.numpy(), "dout": dout.numpy()}
for name, (am, tm) in MODES.items():
    s = sim.clone().requires_grad_(True)
    y = getattr(ref_pool, name)()({"sim": s, "audio_len": audio_len, "text_len": text_len})
    y.backward(dout)
    out[f"{name}/out"], out[f"{name}/dsim"] = y.detach().numpy(), s.grad.numpy()
    mine = O.sim_pooling(sim, audio_len, text_len, am, tm)
    err = (mine - y.detach()).abs().max().item()
    print(f"{name:26s} oracle vs reference {err:.1e}")
    assert err < 1e-13
# MultiText reducers and the four pooling modes: frame_sim (B, T, n_txt) pooled over the frames
fs = torch.rand(B, T, N, generator=g, dtype=torch.float64) * 0.98 + 0.01
dclip = torch.randn(B, N, generator=g, dtype=torch.float64)
out["frame_sim"], out["dclip"] = fs.numpy(), dclip.numpy()
for mode, fn in {"linear_softmax": ref_utils.linear_softmax_with_lens, "max": ref_utils.max_with_lens,
                 "mean": ref_utils.mean_with_lens, "exp_softmax": ref_utils.exp_softmax_with_lens}.items():
    s = fs.clone().requires_grad_(True)
    y = fn(s, audio_len)
    y.backward(dclip)
    out[f"pool_{mode}/out"], out[f"pool_{mode}/dsim"] = y.detach().numpy(), s.grad.numpy()
    err = (O.SEQ_POOL[mode](fs, audio_len) - y.detach()).abs().max().item()
    print(f"{mode + '_with_lens':26s} oracle vs reference {err:.1e}")
    assert err < 1e-13
for name in ("MultiTextLinearSoft", "MultiTextMax"):
    s = fs.transpose(1, 2).clone().requires_grad_(True)                      # (B, n_txt, T)
    y = getattr(ref_pool, name)()({"sim": s, "audio_len": audio_len})
    y.backward(dclip)
    out[f"{name}/out"], out[f"{name}/dsim"] = y.detach().numpy(), s.grad.numpy()
# EmbeddingAgg(aggregation="attention"): the reference's AttentionPooling (models/text_encoder.py:46-58)
from models.text_encoder import AttentionPooling  # noqa: E402  (the reference; sentence_transformers stubbed)
D, L = 96, 5
torch.manual_seed(3)
ap = AttentionPooling(D).double()
with torch.no_grad():
    ap.fc.weight.mul_(8.0)
x = torch.randn(B, L, D, generator=g, dtype=torch.float64).requires_grad_(True)
lens = torch.tensor([5, 1, 3, 4])
dpool = torch.randn(B, D, generator=g, dtype=torch.float64)
y = ap(x, lens)
y.backward(dpool)
out.update({"attnpool/x": x.detach().numpy(), "attnpool/lens": lens.numpy(), "attnpool/w": ap.fc.weight.detach().numpy(),
            "attnpool/b": ap.fc.bias.detach().numpy(), "attnpool/dout": dpool.numpy(), "attnpool/out": y.detach().numpy(),
            "attnpool/dx": x.grad.numpy(), "attnpool/dw": ap.fc.weight.grad.numpy(), "attnpool/db": ap.fc.bias.grad.numpy()})
err = (O.attention_pooling(x.detach(), lens, ap.fc.weight.detach(), ap.fc.bias.detach()) - y.detach()).abs().max().item()
print(f"{'AttentionPooling':26s} oracle vs reference {err:.1e}")
assert err < 1e-13
np.savez_compressed(os.path.join(HERE, "sim_pooling.npz"), **out)
print("wrote sim_pooling.npz", os.path.getsize(os.path.join(HERE, "sim_pooling.npz")))
