#!/usr/bin/env python3
"""tests/golden/cross_encoder.npz: outputs and gradients of the REFERENCE's CrossAttentionGating + DotProduct(text_level=
"token") (imported from /root/reference: models/cross_encoder.py, models/match.py) on a small seeded case, in fp32 and
with an fp64 twin; asserts that oracle.tag_oracle's restatement equals the reference.  Build container only."""
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
from models.cross_encoder import CrossAttentionGating  # noqa: E402  (the reference)
from models.match import DotProduct  # noqa: E402

D, B, T, L = 64, 3, 11, 4
st = O.init_cross_state(seed=17, dim=D, scale=3.0)
g = torch.Generator().manual_seed(5)
audio = torch.randn(B, T, D, generator=g)
token = torch.randn(B, L, D, generator=g)
audio_len = torch.tensor([11, 7, 9])
text_len = torch.tensor([4, 1, 3])
dsim = torch.randn(B, T, generator=g)

out = {}
for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
    ce = CrossAttentionGating(D).to(dt)
    ce.load_state_dict({k[len("cross_encoder."):]: v.to(dt) for k, v in st.items()})
    a = audio.detach().clone().to(dt).requires_grad_(True)
    t = token.detach().clone().to(dt).requires_grad_(True)
    enc = ce({"audio_emb": a, "text_emb": {"token_emb": t}, "audio_len": audio_len, "text_len": text_len})
    sim = DotProduct(text_level="token")({"audio_emb": enc["audio_emb"], "text_emb": enc["text_emb"]})
    sim.backward(dsim.to(dt))
    out[f"audio_out_{tag}"] = enc["audio_emb"].detach().numpy()
    out[f"text_out_{tag}"] = enc["text_emb"]["token_emb"].detach().numpy()
    out[f"sim_{tag}"] = sim.detach().numpy()
    out[f"daudio_{tag}"] = a.grad.numpy()
    out[f"dtoken_{tag}"] = t.grad.numpy()
    for n, p in ce.named_parameters():
        out[f"grad_{tag}/cross_encoder.{n}"] = p.grad.numpy()
    # the restatement must equal the reference
    st_d = {k: v.to(dt) for k, v in st.items()}
    ao, to = O.cross_attention_gating(st_d, audio.detach().to(dt), token.detach().to(dt), audio_len, text_len)
    so = O.match_dot_product_token(ao, to)
    err = max((ao - enc["audio_emb"]).abs().max().item(), (to - enc["text_emb"]["token_emb"]).abs().max().item(),
              (so - sim).abs().max().item())
    print(f"{tag}: oracle vs reference max abs err {err:.2e}; sim range [{sim.min().item():.3f}, {sim.max().item():.3f}]")
    assert err < (1e-5 if dt == torch.float32 else 1e-12)
np.savez_compressed(os.path.join(HERE, "cross_encoder.npz"), audio=audio.numpy(), token=token.numpy(),
                    audio_len=audio_len.numpy(), text_len=text_len.numpy(), dsim=dsim.numpy(),
                    **{"w/" + k: v.numpy() for k, v in st.items()}, **out)
print("wrote cross_encoder.npz", os.path.getsize(os.path.join(HERE, "cross_encoder.npz")))
