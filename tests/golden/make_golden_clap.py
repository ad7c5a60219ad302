#!/usr/bin/env python3
"""Generate tests/golden/clap_text_tiny.npz by running the REAL Hugging Face modules the reference's LaionClapEncoder wraps
(models/hf_modeling_grounding.py:183-199: ClapModel.text_model + ClapModel.text_projection) with random weights on a tiny
configuration, and assert that oracle/clap_text_oracle.py reproduces them.  Run in the build container only
(`transformers` is a third-party package of this image; nothing here reads /root/reference at test time)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import clap_text_oracle as C  # noqa: E402

from transformers.models.clap.configuration_clap import ClapTextConfig  # noqa: E402
from transformers.models.clap.modeling_clap import ClapProjectionLayer, ClapTextModel  # noqa: E402
import transformers  # noqa: E402

torch.manual_seed(11)
cfg = ClapTextConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=160, vocab_size=120,
                     max_position_embeddings=24, projection_dim=32)
cfg._attn_implementation = "eager"
text = ClapTextModel(cfg, add_pooling_layer=True).eval()
proj = ClapProjectionLayer(cfg).eval()
with torch.no_grad():                      # default init is tiny (std 0.02): widen so that every stage matters
    for p in list(text.parameters()) + list(proj.parameters()):
        p.mul_(2.5).add_(0.03 * torch.randn_like(p))
ids, mask = C.synthetic_tokens(5, 9, seed=3, vocab=120)
with torch.no_grad():
    out = text(input_ids=ids, attention_mask=mask)
    token_emb = proj(out.last_hidden_state)
    seq_emb = torch.nn.functional.normalize(proj(out.pooler_output), dim=-1)
st = {"model." + k: v for k, v in text.state_dict().items() if k != "embeddings.token_type_ids"}
st.update({"projection." + k: v for k, v in proj.state_dict().items()})
o = C.laion_clap_encoder_forward(st, ids, mask, cfg.num_attention_heads, cfg.layer_norm_eps)
h, pooled = C.text_model_forward(st, ids, mask, cfg.num_attention_heads, cfg.layer_norm_eps, prefix="model.")
err = max((h - out.last_hidden_state).abs().max().item(), (pooled - out.pooler_output).abs().max().item(),
          (o["token_emb"] - token_emb).abs().max().item(), (o["seq_emb"] - seq_emb).abs().max().item())
st64 = {k: v.double() for k, v in st.items()}
o64 = C.laion_clap_encoder_forward(st64, ids, mask, cfg.num_attention_heads, cfg.layer_norm_eps)
floor = (o64["token_emb"] - token_emb.double()).abs().max().item()
print(f"transformers {transformers.__version__}: oracle vs ClapTextModel max abs err {err:.2e}; "
      f"fp32-vs-fp64 floor on token_emb {floor:.2e} (|max| {token_emb.abs().max().item():.2f})")
assert err < 5e-6
valid = mask.bool()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "clap_text_tiny.npz"),
                    input_ids=ids.numpy(), attention_mask=mask.numpy(), n_heads=cfg.num_attention_heads,
                    eps=cfg.layer_norm_eps, last_hidden_state=out.last_hidden_state.numpy(),
                    pooler_output=out.pooler_output.numpy(), token_emb=token_emb.numpy(), seq_emb=seq_emb.numpy(),
                    transformers_version=transformers.__version__,
                    **{"w/" + k: v.numpy() for k, v in st.items()})
print("wrote clap_text_tiny.npz", os.path.getsize(os.path.join(ROOT, "tests", "golden", "clap_text_tiny.npz")), "bytes")
