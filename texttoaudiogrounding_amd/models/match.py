"""Frame x phrase similarity heads (mirror of models/match.py:10-88 in the reference)."""
import torch
import torch.nn as nn

from .. import ops
from .. import torch_ops  # noqa: F401  (registers torch.ops.tag.*)


def _token_text(input_dict):
    """text_level='token': the reference broadcasts audio (B,T,D) against token_emb, which only type-checks when there is one
    text vector per frame (the cross-encoder's output, (B,T,D)) -- the same restriction holds here, loudly."""
    text = input_dict["text_emb"]["token_emb"]
    if text.shape != input_dict["audio_emb"].shape:
        raise RuntimeError(f"text_level='token' needs one text vector per audio frame (a cross-encoder output): audio "
                           f"{tuple(input_dict['audio_emb'].shape)} vs token_emb {tuple(text.shape)}")
    return text


class ExpNegL2(nn.Module):
    def __init__(self, l2norm=True, text_level="seq") -> None:
        super().__init__()
        self.l2norm = l2norm
        self.text_level = text_level

    def forward(self, input_dict):
        if self.text_level == "token":
            return ops.RowPairFunction.apply(input_dict["audio_emb"], _token_text(input_dict), 1, self.l2norm, False)
        return torch.ops.tag.frame_match(input_dict["audio_emb"], input_dict["text_emb"]["seq_emb"], 1, bool(self.l2norm), False)


class DotProduct(nn.Module):
    def __init__(self, l2norm=False, scale=True, text_level="seq") -> None:
        super().__init__()
        self.l2norm = l2norm
        self.scale = scale
        self.text_level = text_level

    def forward(self, input_dict):
        if self.text_level == "token":
            # after a cross-encoder the text is one vector per frame (B,T,D): (audio * text).sum(-1)  (models/match.py:53-59)
            text = _token_text(input_dict)
            if self.l2norm:
                return ops.RowPairFunction.apply(input_dict["audio_emb"], text, 0, True, self.scale)
            return ops.RowDotFunction.apply(input_dict["audio_emb"], text, self.scale)
        return torch.ops.tag.frame_match(input_dict["audio_emb"], input_dict["text_emb"]["seq_emb"], 0, bool(self.l2norm),
                                         bool(self.scale))


class CrossAttention(nn.Module):
    """Mirror of models/match.py:63-88: same constructor, same sub-module / parameter names (``attn`` is an
    nn.MultiheadAttention used as the parameter container, ``norm``, ``linear``) so reference checkpoints load; the forward
    runs in libtag_hip.so (projection GEMMs on the MFMA + csrc/mha.hip)."""

    def __init__(self, embed_dim, num_heads, dropout, kvdim=None) -> None:
        super().__init__()
        self.attn = nn.MultiheadAttention(embed_dim, num_heads, dropout, batch_first=True, kdim=kvdim, vdim=kvdim)
        self.dropout = nn.Dropout(dropout)
        self.norm = nn.LayerNorm(embed_dim)
        self.linear = nn.Linear(embed_dim, 1)
        self.embed_dim, self.num_heads, self.p = embed_dim, num_heads, dropout

    def forward(self, input_dict):
        audio = input_dict["audio_emb"]
        text = input_dict["text_emb"]["token_emb"]
        m, E = self.attn, self.embed_dim
        if m.in_proj_weight is not None:
            # kvdim == embed_dim: nn.MultiheadAttention holds ONE in_proj_weight (3E, E); its three row blocks enter the HIP
            # node as (differentiable) views and autograd stitches their gradients back into in_proj_weight.grad
            wq, wk, wv = m.in_proj_weight[:E], m.in_proj_weight[E:2 * E], m.in_proj_weight[2 * E:]
        else:
            wq, wk, wv = m.q_proj_weight, m.k_proj_weight, m.v_proj_weight
        return ops.CrossAttentionHeadFunction.apply(audio, text, input_dict["text_len"], self.num_heads, self.p,
                                                    self.training, wq, wk, wv, m.in_proj_bias, m.out_proj.weight,
                                                    m.out_proj.bias, self.norm.weight, self.norm.bias, self.linear.weight,
                                                    self.linear.bias)
