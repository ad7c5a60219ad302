"""Cross-encoder (mirror of models/cross_encoder.py:5-79 in the reference; BASELINE configs[3]): additive audio->text
attention + sigmoid cross-gating, plugged into ``BiEncoder(cross_encoder=CrossAttentionGating(D))`` with
``match.DotProduct(text_level="token")``.  Same constructor arguments, parameter names and forward contracts -- every
module runs standalone (``Seq2SeqAttention.forward``, ``CrossGating.forward``) and ``CrossAttentionGating`` composes them
like the reference; the arithmetic (forward and backward) runs in libtag_hip.so through ``ops.Seq2SeqAttentionFunction``
/ ``ops.CrossGatingFunction``."""
import torch
import torch.nn as nn

from .. import ops


class Seq2SeqAttention(nn.Module):
    def __init__(self, d_q, d_kv, d_attn):
        super().__init__()
        self.h2attn = nn.Linear(d_q + d_kv, d_attn)
        self.v = nn.Parameter(torch.randn(d_attn))

    def forward(self, query, kv, query_len, kv_len):
        """query (B,Lq,d_q), kv (B,Lk,d_kv) -> (B,Lq,d_kv) (models/cross_encoder.py:11-42)."""
        return ops.Seq2SeqAttentionFunction.apply(query, kv, query_len, kv_len, self.h2attn.weight, self.h2attn.bias, self.v)


class CrossGating(nn.Module):
    def __init__(self, d_model) -> None:
        super().__init__()
        self.fc_u = nn.Linear(d_model, d_model)
        self.fc_s = nn.Linear(d_model, d_model)

    def forward(self, u, s):
        """-> (u * sigmoid(fc_s(s)), s * sigmoid(fc_u(u))) (models/cross_encoder.py:52-57)."""
        return ops.CrossGatingFunction.apply(u, s, self.fc_u.weight, self.fc_u.bias, self.fc_s.weight, self.fc_s.bias)


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
        text_emb = self.attn(audio_emb, text_emb, input_dict["audio_len"], input_dict["text_len"])
        audio_emb, text_emb = self.gating(audio_emb, text_emb)
        return {"audio_emb": audio_emb, "text_emb": {"token_emb": text_emb}}
