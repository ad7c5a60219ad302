"""Pooling of similarity matrices (mirror of models/sim_pooling.py:6-204 in the reference -- all twelve reducers plus the two
MultiText ones): an audio-axis reducer over the valid frames followed by a text-axis reducer over the valid tokens, one
HIP kernel pair (tag_sim_pool_forward / _backward) parameterised by the two modes."""
import torch
import torch.nn as nn

from .. import ops


class _PairPool(nn.Module):
    """sim (B, B, T, N): rows (clip a, caption b); audio_len by clip, text_len by caption -> (B, B)."""
    audio_mode = "mean"
    text_mode = "mean"

    def forward(self, input):
        sim = input["sim"]
        B, _, T, N = sim.shape
        dev = sim.device
        audio_len = torch.as_tensor(input["audio_len"]).long().to(dev).contiguous()
        text_len = torch.as_tensor(input["text_len"]).long().to(dev).contiguous()
        out = ops.SimPoolFunction.apply(sim.reshape(B * B, T, N), audio_len, text_len, B, B,
                                        ops.POOL_MODES[self.audio_mode], ops.TEXT_MODES[self.text_mode])
        return out.view(B, B)


class AudioMeanTextMean(_PairPool):
    audio_mode, text_mode = "mean", "mean"


class AudioMeanTextSum(_PairPool):
    audio_mode, text_mode = "mean", "sum"


class AudioMaxTextMean(_PairPool):
    audio_mode, text_mode = "max", "mean"


class AudioMaxTextMax(_PairPool):
    audio_mode, text_mode = "max", "max"


class AudioMaxTextSum(_PairPool):
    audio_mode, text_mode = "max", "sum"


class AudioMaxTextMeanSum(_PairPool):
    audio_mode, text_mode = "max", "mean_sum"


class AudioLinearSoftTextMean(_PairPool):
    audio_mode, text_mode = "linear_softmax", "mean"


class AudioLinearSoftTextSum(_PairPool):
    audio_mode, text_mode = "linear_softmax", "sum"


class AudioExpSoftTextMean(_PairPool):
    audio_mode, text_mode = "exp_softmax", "mean"


class AudioExpSoftTextSum(_PairPool):
    audio_mode, text_mode = "exp_softmax", "sum"


class _MultiTextPool(nn.Module):
    """sim (B, n_txt, T): pooled over the frames < audio_len[b] for every phrase -> (B, n_txt)
    (models/sim_pooling.py:191-204: ``*_with_lens(sim.transpose(1, 2), audio_len)``)."""
    audio_mode = "linear_softmax"

    def forward(self, input):
        sim = input["sim"]
        B, n_txt, T = sim.shape
        audio_len = torch.as_tensor(input["audio_len"]).long().to(sim.device).contiguous()
        out = ops.SimPoolFunction.apply(sim.reshape(B * n_txt, T, 1), audio_len, None, n_txt, 1,
                                        ops.POOL_MODES[self.audio_mode], -1)
        return out.view(B, n_txt)


class MultiTextLinearSoft(_MultiTextPool):
    audio_mode = "linear_softmax"


class MultiTextMax(_MultiTextPool):
    audio_mode = "max"
