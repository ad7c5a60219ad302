#!/usr/bin/env python3
"""tests/golden/align_heads.npz: the REFERENCE's align-by-phrase head chain (models.align.DotProduct ->
models.sim_pooling.AudioMeanTextMean -> losses.MaxMarginRankingLoss, as AudioTextAlignByPhrase.forward composes them,
models/audio_text_model.py:944-976) on a small seeded case in fp32 and fp64 with gradients; asserts the oracle equals it."""
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
from models.align import DotProduct  # noqa: E402  (the reference)
from models.sim_pooling import AudioMeanTextMean  # noqa: E402
from losses import MaxMarginRankingLoss  # noqa: E402

B, T, N, D = 4, 9, 3, 32
g = torch.Generator().manual_seed(33)
audio = torch.randn(B, T, D, generator=g)
text = torch.randn(B, N, D, generator=g)
audio_len = torch.tensor([9, 5, 7, 9])
text_len = [3, 1, 2, 3]
out = {}
for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
    a = audio.detach().clone().to(dt).requires_grad_(True)
    t = text.detach().clone().to(dt).requires_grad_(True)
    m = DotProduct(scaled=True)(a, t)
    sim = AudioMeanTextMean()({"sim": m, "audio_len": audio_len, "text_len": text_len})
    loss = MaxMarginRankingLoss(margin=0.1)({"sim": sim})
    loss.backward()
    out.update({f"matrix_{tag}": m.detach().numpy(), f"sim_{tag}": sim.detach().numpy(), f"loss_{tag}": loss.item(),
                f"daudio_{tag}": a.grad.numpy(), f"dtext_{tag}": t.grad.numpy()})
    mo = O.align_dot_product(audio.to(dt), text.to(dt), scaled=True)
    so = O.audio_mean_text_mean(mo, audio_len, text_len)
    lo = O.max_margin_ranking_loss(so, 0.1, 1.0)
    err = max((mo - m).abs().max().item(), (so - sim).abs().max().item(), abs(lo.item() - loss.item()))
    print(f"{tag}: oracle vs reference {err:.2e}; loss {loss.item():.5f}")
    assert err < (1e-6 if dt == torch.float32 else 1e-13)
np.savez_compressed(os.path.join(HERE, "align_heads.npz"), audio=audio.numpy(), text=text.numpy(), audio_len=audio_len.numpy(),
                    text_len=np.array(text_len), **out)
print("wrote align_heads.npz")
