#!/usr/bin/env python3
"""tests/golden/weak_heads.npz: the REFERENCE's weak-supervision head chain (models.match.DotProduct on the expanded audio,
models.utils.linear_softmax_with_lens, losses.ClipBceLoss -- exactly what MultiTextBiEncoder.forward composes,
models/audio_text_model.py:150-215) on a small seeded case, fp32 and fp64, with gradients; asserts the oracle's
restatement equals it.  Build container only."""
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
from models.match import DotProduct  # noqa: E402  (the reference)
from models.utils import linear_softmax_with_lens  # noqa: E402
from losses import ClipBceLoss  # noqa: E402

B, N, T, D = 3, 5, 13, 64
g = torch.Generator().manual_seed(21)
audio = torch.randn(B, T, D, generator=g) * 1.5
text = torch.randn(B * N, D, generator=g) * 1.5
length = torch.tensor([13, 8, 11])
label = (torch.rand(B, N, generator=g) < 0.4).float()
out = {}
for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
    a = audio.detach().clone().to(dt).requires_grad_(True)
    t = text.detach().clone().to(dt).requires_grad_(True)
    ae = a.unsqueeze(1).expand(-1, N, -1, -1).reshape(B * N, T, D)
    fs = DotProduct()({"audio_emb": ae, "text_emb": {"seq_emb": t}})
    frame_sim = fs.reshape(B, N, -1).transpose(1, 2)
    clip = linear_softmax_with_lens(frame_sim, length)
    loss = ClipBceLoss()({"clip_sim": clip, "label": label.to(dt)})
    loss.backward()
    out.update({f"frame_sim_{tag}": frame_sim.detach().numpy(), f"clip_sim_{tag}": clip.detach().numpy(),
                f"loss_{tag}": loss.item(), f"daudio_{tag}": a.grad.numpy(), f"dtext_{tag}": t.grad.numpy()})
    fo, co = O.multitext_head(audio.to(dt), text.to(dt), length, N)
    lo = O.clip_bce_loss(co, label.to(dt))
    err = max((fo - frame_sim).abs().max().item(), (co - clip).abs().max().item(), abs(lo.item() - loss.item()))
    print(f"{tag}: oracle vs reference {err:.2e}; clip_sim range [{clip.min().item():.3f}, {clip.max().item():.3f}]")
    assert err < (1e-6 if dt == torch.float32 else 1e-13)
np.savez_compressed(os.path.join(HERE, "weak_heads.npz"), audio=audio.numpy(), text=text.numpy(), length=length.numpy(),
                    label=label.numpy(), n_text=N, **out)
print("wrote weak_heads.npz")
