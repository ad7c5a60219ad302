"""Word-embedding text encoder (mirror of models/text_encoder.py:14-43,61-88 in the reference)."""
from typing import Dict

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .. import torch_ops  # noqa: F401  (registers torch.ops.tag.*)
from .utils import init_weights


def _ids_to_device(text, table):
    """Token ids as the kernels take them (int64, contiguous, on the table's device).  Ids still on the host (what the
    reference's collate function hands over) are checked right here, and an id outside the table raises at once like
    nn.Embedding (models/text_encoder.py:39); device-resident ids are checked by the kernel, which can only raise a sticky
    flag: ops.check_async_errors() (StrongRunner.loss_value, segments_for_thresholds)."""
    text = torch.as_tensor(text)
    if not text.is_cuda and text.numel():
        lo, hi = int(text.min()), int(text.max())
        if lo < 0 or hi >= table.shape[0]:
            raise IndexError(f"index out of range in self: token ids span [{lo}, {hi}], the table has {table.shape[0]} rows")
    return text.long().to(table.device).contiguous()


class EmbeddingLayer(nn.Module):
    def __init__(self, vocab_size: int, embed_dim: int, pretrained_embedding: str = None,
                 freeze_embedding: bool = False):
        super().__init__()
        self.embed_dim = embed_dim
        self.core = nn.Embedding(vocab_size, embed_dim)
        self.apply(init_weights)
        if pretrained_embedding is not None:
            self.load_pretrained_embedding(pretrained_embedding, freeze_embedding)

    def load_pretrained_embedding(self, weight: str, freeze: bool = True):
        w = np.load(weight)
        if w.shape != tuple(self.core.weight.shape):
            raise AssertionError(f"expect embedding with shape {tuple(self.core.weight.shape)} but {w.shape} is given")
        self.core = nn.Embedding.from_pretrained(torch.as_tensor(w, dtype=torch.float), freeze)

    def forward(self, input_dict: Dict):
        """models/text_encoder.py:39-43: ``core(tokens.long())`` for token ids of any shape -> (*tokens.shape, embed_dim); every
        position is looked up, padding included (row 0 of the table).  The gather is the token_emb half of the fused
        gather + mean kernel (``tag_embed_mean_forward``; EmbeddingAgg uses both halves in one launch), its backward the
        deterministic gather of ``tag_embed_tokens_backward``."""
        table = self.core.weight
        text = _ids_to_device(input_dict["text"], table)
        shape = tuple(text.shape)
        ids = text.reshape(-1, shape[-1] if text.dim() > 1 else 1)
        if ids.numel() == 0:
            return table.new_zeros(*shape, table.shape[1])
        lens = torch.full((ids.shape[0],), ids.shape[1], dtype=torch.long, device=table.device)
        if ops.DIRECT_GRADS:
            _, tok = ops.EmbedMeanFunction.apply(table, ids, lens, True)
        else:
            _, tok = torch.ops.tag.embed_mean(table, ids, lens)
        return tok.view(*shape, table.shape[1])


class AttentionPooling(nn.Module):
    """Parameter container with the reference's layout (models/text_encoder.py:46-58): ``fc = Linear(emb_dim, 1)``."""

    def __init__(self, emb_dim):
        super().__init__()
        self.fc = nn.Linear(emb_dim, 1)

    def forward(self, x, lens):
        lens = torch.as_tensor(lens).long().to(x.device).contiguous()
        return ops.AttnPoolFunction.apply(x, lens, self.fc.weight, self.fc.bias)


class EmbeddingAgg(nn.Module):
    def __init__(self, vocab_size, embed_dim, pretrained_embedding: str = None, freeze_embedding: bool = False,
                 aggregation: str = "mean"):
        super().__init__()
        self.embedding = EmbeddingLayer(vocab_size, embed_dim, pretrained_embedding, freeze_embedding)
        self.embed_dim = self.embedding.embed_dim
        self.agg = aggregation
        if aggregation == "attention":
            self.attn = AttentionPooling(embed_dim)
        elif aggregation != "mean":
            raise Exception(f"{aggregation} not supported")            # as the reference (models/text_encoder.py:87-88)

    def forward(self, input_dict):
        table = self.embedding.core.weight
        dev = table.device
        text = _ids_to_device(input_dict["text"], table)
        lens = torch.as_tensor(input_dict["text_len"]).long().to(dev).contiguous()
        if ops.DIRECT_GRADS:          # StrongRunner: scatter the table gradient straight into its flat-gradient rows
            seq, tok = ops.EmbedMeanFunction.apply(table, text, lens, True)
        else:
            seq, tok = torch.ops.tag.embed_mean(table, text, lens)
        if self.agg == "attention":
            seq = self.attn(tok, lens)
        return {"token_emb": tok, "seq_emb": seq}
