"""Cross-batch frame x text similarity (mirror of models/align.py:7-31 in the reference)."""
import torch
import torch.nn as nn

from .. import torch_ops  # noqa: F401  (registers torch.ops.tag.*)


class DotProduct(nn.Module):
    def __init__(self, l2norm=False, scaled=False) -> None:
        super().__init__()
        self.l2norm = l2norm
        self.scaled = scaled

    def forward(self, audio: torch.Tensor, text: torch.Tensor, **kwargs):
        a_bs, _, a_dim = audio.size()
        t_bs, _, t_dim = text.size()
        assert a_bs == t_bs
        assert a_dim == t_dim
        return torch.ops.tag.align_dot(audio, text, bool(self.l2norm), bool(self.scaled))
