"""Frame x phrase similarity heads (mirror of models/match.py:10-60 in the reference)."""
import torch.nn as nn

from .. import ops


def _seq_text(input_dict, text_level):
    if text_level != "seq":
        raise NotImplementedError("the HIP heads implement text_level='seq' (the strong/BiEncoder path)")
    return input_dict["text_emb"]["seq_emb"]


class ExpNegL2(nn.Module):
    def __init__(self, l2norm=True, text_level="seq") -> None:
        super().__init__()
        self.l2norm = l2norm
        self.text_level = text_level

    def forward(self, input_dict):
        return ops.MatchFunction.apply(input_dict["audio_emb"], _seq_text(input_dict, self.text_level), 1,
                                       self.l2norm, False)


class DotProduct(nn.Module):
    def __init__(self, l2norm=False, scale=True, text_level="seq") -> None:
        super().__init__()
        self.l2norm = l2norm
        self.scale = scale
        self.text_level = text_level

    def forward(self, input_dict):
        if self.text_level == "token":
            # after a cross-encoder the text is one vector per frame (B,T,D): (audio * text).sum(-1)  (models/match.py:53-59)
            text = input_dict["text_emb"]["token_emb"]
            if self.l2norm or text.shape != input_dict["audio_emb"].shape:
                raise NotImplementedError("token-level DotProduct is implemented for the cross-encoder output "
                                          "(text (B,T,D), l2norm=False)")
            return ops.RowDotFunction.apply(input_dict["audio_emb"], text, self.scale)
        return ops.MatchFunction.apply(input_dict["audio_emb"], _seq_text(input_dict, self.text_level), 0,
                                       self.l2norm, self.scale)
