"""Cross-encoder (mirror of models/cross_encoder.py:5-79 in the reference; BASELINE configs[3]): additive audio->text
attention + sigmoid cross-gating, plugged into ``BiEncoder(cross_encoder=CrossAttentionGating(D))`` with
``match.DotProduct(text_level="token")``.  Same constructor arguments, parameter names and forward contract; the
arithmetic (forward and backward) runs in libtag_hip.so through ``ops.CrossEncoderFunction``."""
import torch
import torch.nn as nn

from .. import ops


class Seq2SeqAttention(nn.Module):
    def __init__(self, d_q, d_kv, d_attn):
        super().__init__()
        self.h2attn = nn.Linear(d_q + d_kv, d_attn)
        self.v = nn.Parameter(torch.randn(d_attn))

    def forward(self, query, kv, query_len, kv_len):
        raise RuntimeError("evaluated inside CrossAttentionGating's fused HIP node (ops.CrossEncoderFunction)")


class CrossGating(nn.Module):
    def __init__(self, d_model) -> None:
        super().__init__()
        self.fc_u = nn.Linear(d_model, d_model)
        self.fc_s = nn.Linear(d_model, d_model)

    def forward(self, u, s):
        raise RuntimeError("evaluated inside CrossAttentionGating's fused HIP node (ops.CrossEncoderFunction)")


class CrossAttentionGating(nn.Module):
    def __init__(self, embed_dim):
        super().__init__()
        self.attn = Seq2SeqAttention(embed_dim, embed_dim, embed_dim)
        self.gating = CrossGating(embed_dim)

    def forward(self, input_dict):
        audio_emb = input_dict["audio_emb"]
        text_emb = input_dict["text_emb"]
        if isinstance(text_emb, dict):
            text_emb = text_emb["token_emb"]
        a, g = self.attn, self.gating
        audio_out, text_out = ops.CrossEncoderFunction.apply(
            audio_emb, text_emb, input_dict["audio_len"], input_dict["text_len"], a.h2attn.weight, a.h2attn.bias, a.v,
            g.fc_u.weight, g.fc_u.bias, g.fc_s.weight, g.fc_s.bias)
        return {"audio_emb": audio_out, "text_emb": {"token_emb": text_out}}
