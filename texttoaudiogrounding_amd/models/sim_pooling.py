"""Pooling of the cross-batch similarity matrix (mirror of models/sim_pooling.py:6-22 in the reference)."""
import torch
import torch.nn as nn

from .. import ops


class AudioMeanTextMean(nn.Module):
    def forward(self, input):
        sim = input["sim"]                                   # (B, B, T, N)
        dev = sim.device
        audio_len = torch.as_tensor(input["audio_len"]).long().to(dev).contiguous()
        text_len = torch.as_tensor(input["text_len"]).long().to(dev).contiguous()
        return ops.MeanMeanPoolFunction.apply(sim, audio_len, text_len)
