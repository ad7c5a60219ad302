"""Audio encoders behind the reference interface (models/audio_encoder.py:89-232 in the reference).

``Cnn8Rnn(sample_rate, freeze_cnn, freeze_bn, pretrained, output_fn)`` keeps the reference's
constructor, attributes (``embed_dim``, ``downsample_ratio``, ``time_resolution``), forward
contract ``forward(input_dict) -> {"embedding": (B,T',512), "length": (B,)}`` and state-dict keys,
while the arithmetic (log-mel -> bn0 -> 4 conv blocks -> mean -> fc1 -> BiGRU, forward and backward)
runs in the gfx950 kernels of libtag_hip.so through one autograd node.
"""
import math
import sys
from typing import Dict

import torch
import torch.nn as nn

from .. import ops
from .. import torch_ops  # noqa: F401  (registers torch.ops.tag.*)
from .panns import ConvBlock, init_bn, init_layer


def _hz_to_mel_slaney(f):
    f_sp = 200.0 / 3
    if f >= 1000.0:
        return 1000.0 / f_sp + math.log(f / 1000.0) / (math.log(6.4) / 27.0)
    return f / f_sp


def slaney_mel_filterbank(n_freqs, f_min, f_max, n_mels, sample_rate):
    """(n_freqs, n_mels) slaney-scale, slaney-normalised triangles = the ``fb`` buffer torchaudio's
    MelScale(norm="slaney", mel_scale="slaney") holds (models/audio_encoder.py:113-123)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_mel_slaney(f_min), _hz_to_mel_slaney(f_max), n_mels + 2)
    f_sp = 200.0 / 3
    min_log_mel = 1000.0 / f_sp
    f_pts = f_sp * m_pts
    is_log = m_pts >= min_log_mel
    f_pts[is_log] = 1000.0 * torch.exp((math.log(6.4) / 27.0) * (m_pts[is_log] - min_log_mel))
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.min(down, up), min=0.0)
    return fb * (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])).unsqueeze(0)


class _MelFrontendBuffers(nn.Module):
    """Holds the two persistent buffers a real torchaudio MelSpectrogram contributes to the reference's
    state dict (``melspec_extractor.spectrogram.window`` / ``melspec_extractor.mel_scale.fb``) so that
    published checkpoints load with strict=True."""

    def __init__(self, window, fb):
        super().__init__()
        self.spectrogram = nn.Module()
        self.spectrogram.register_buffer("window", window)
        self.mel_scale = nn.Module()
        self.mel_scale.register_buffer("fb", fb)


def htk_mel_filterbank(n_freqs, f_min, f_max, n_mels, sample_rate):
    """(n_freqs, n_mels) HTK-scale triangles without normalisation = torchaudio's MelScale defaults, which is what
    the reference's CrnnEncoder frontend uses (models/audio_encoder.py:29-35)."""
    hz2mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(hz2mel(f_min), hz2mel(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    return torch.clamp(torch.min(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]), min=0.0)


def cdur_block(cin, cout, kernel_size=3, padding=1):
    """Parameter container with the reference's layout (models/audio_encoder.py:16-22): [BN, conv, LeakyReLU]."""
    return nn.Sequential(nn.BatchNorm2d(cin), nn.Conv2d(cin, cout, kernel_size=kernel_size, padding=padding, bias=False),
                         nn.LeakyReLU(inplace=True, negative_slope=0.1))


class CrnnEncoder(nn.Module):
    """The encoder the strong eg_config instantiates (models/audio_encoder.py:25-86): same constructor, attributes,
    forward contract and state-dict keys (cnn.{0,2,3,5,6}.{0,1}.*, gru.*); arithmetic in libtag_hip.so."""

    def __init__(self, sample_rate, embed_dim):
        super().__init__()
        from .utils import init_weights
        self.downsample_ratio = 4
        self.n_fft = 2048
        self.win_length = 40 * sample_rate // 1000
        self.hop_length = 20 * sample_rate // 1000
        self.time_resolution = 4 * self.hop_length / sample_rate
        if self.win_length > self.n_fft:
            raise ValueError("win_length must not exceed n_fft=2048")
        self.melspec_extractor = _MelFrontendBuffers(
            torch.hann_window(self.win_length),
            htk_mel_filterbank(self.n_fft // 2 + 1, 0.0, float(sample_rate // 2), 64, sample_rate))
        self.embed_dim = embed_dim
        self.cnn = nn.Sequential(cdur_block(1, 32), nn.LPPool2d(4, (2, 4)), cdur_block(32, 128), cdur_block(128, 128),
                                 nn.LPPool2d(4, (2, 4)), cdur_block(128, 128), cdur_block(128, 128),
                                 nn.LPPool2d(4, (1, 4)), nn.Dropout(0.3))
        self.gru = nn.GRU(128, embed_dim // 2, bidirectional=True, batch_first=True)
        self.dropout_p = 0.3
        self.apply(init_weights)

    window = property(lambda self: self.melspec_extractor.spectrogram.window)
    mel_fb = property(lambda self: self.melspec_extractor.mel_scale.fb)

    def _bn_modules(self):
        return [self.cnn[i][0] for i in (0, 2, 3, 5, 6)]

    def _flat_params(self):
        ps = []
        for i in (0, 2, 3, 5, 6):
            ps += [self.cnn[i][0].weight, self.cnn[i][0].bias, self.cnn[i][1].weight]
        for sfx in ("", "_reverse"):
            ps += [getattr(self.gru, f"weight_ih_l0{sfx}"), getattr(self.gru, f"weight_hh_l0{sfx}"),
                   getattr(self.gru, f"bias_ih_l0{sfx}"), getattr(self.gru, f"bias_hh_l0{sfx}")]
        return ps

    def forward(self, input_dict: Dict):
        if self.training:
            ops.bump_bn_counters(self, self._bn_modules())
        x = torch_ops.run_encoder(torch.ops.tag.crnn_encoder, self, input_dict["waveform"], self._flat_params())
        length = torch.div(torch.as_tensor(input_dict["waveform_len"]), self.hop_length, rounding_mode="floor") + 1
        length = torch.div(length, self.downsample_ratio, rounding_mode="floor")
        return {"embedding": x, "length": length}


class Cnn8Rnn(nn.Module):
    def __init__(self, sample_rate: int, freeze_cnn: bool = False, freeze_bn: bool = False,
                 pretrained: "str | None" = None, output_fn=sys.stdout.write):
        super().__init__()
        self.downsample_ratio = 4
        self.time_resolution = 0.04
        self.freeze_cnn = freeze_cnn
        self.freeze_bn = freeze_bn
        self.hop_length = int(0.010 * sample_rate)
        self.win_length = int(0.032 * sample_rate)
        self.n_fft = self.win_length
        if self.n_fft not in (1024, 2048):
            raise ValueError(f"the HIP log-mel frontend supports n_fft 1024/2048 (sample_rate 32000/64000), got {self.n_fft}")
        f_max = 14000 if sample_rate == 32000 else int(sample_rate / 2)
        self.melspec_extractor = _MelFrontendBuffers(
            torch.hann_window(self.win_length),
            slaney_mel_filterbank(self.n_fft // 2 + 1, 50.0, float(f_max), 64, sample_rate))
        self.bn0 = nn.BatchNorm2d(64)
        self.conv_block1 = ConvBlock(1, 64)
        self.conv_block2 = ConvBlock(64, 128)
        self.conv_block3 = ConvBlock(128, 256)
        self.conv_block4 = ConvBlock(256, 512)
        self.fc1 = nn.Linear(512, 512, bias=True)
        self.rnn = nn.GRU(512, 256, bidirectional=True, batch_first=True)
        self.embed_dim = 512
        self.dropout_p = (0.2, 0.5)     # F.dropout sites of the reference forward (:203-215)
        init_bn(self.bn0)
        init_layer(self.fc1)
        if pretrained is not None:
            self.load_pretrained(pretrained, output_fn)
        if self.freeze_cnn:
            for p in self.parameters():
                p.requires_grad = False
            for p in self.rnn.parameters():
                p.requires_grad = True

    # ---- checkpoint interchange: shape-matched key merge, tolerant of missing/extra keys ----
    def load_pretrained(self, ckpt_path, output_fn=sys.stdout.write):
        state = torch.load(ckpt_path, map_location="cpu")
        state = state.get("model", state)
        own = self.state_dict()
        matched = {k: v for k, v in state.items() if k in own and own[k].shape == v.shape}
        output_fn(f"Cnn8Rnn: loading {len(matched)}/{len(own)} tensors from {ckpt_path}\n")
        own.update(matched)
        self.load_state_dict(own)

    def train(self, mode: bool = True):
        super().train(mode)
        if self.freeze_bn:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self

    @property
    def window(self):
        return self.melspec_extractor.spectrogram.window

    @property
    def mel_fb(self):
        return self.melspec_extractor.mel_scale.fb

    def _flat_params(self):
        ps = [self.bn0.weight, self.bn0.bias]
        for i in range(1, 5):
            blk = getattr(self, f"conv_block{i}")
            ps += [blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, blk.conv2.weight, blk.bn2.weight, blk.bn2.bias]
        ps += [self.fc1.weight, self.fc1.bias]
        for sfx in ("", "_reverse"):
            ps += [getattr(self.rnn, f"weight_ih_l0{sfx}"), getattr(self.rnn, f"weight_hh_l0{sfx}"),
                   getattr(self.rnn, f"bias_ih_l0{sfx}"), getattr(self.rnn, f"bias_hh_l0{sfx}")]
        return ps

    def forward(self, input_dict: Dict):
        waveform = input_dict["waveform"]
        if self.training and input_dict["specaug"]:
            raise NotImplementedError("SpecAugment is off on the strongly-supervised path (run_strong.py:101-103)")
        if self.training and input_dict.get("mixup_lambda", None) is not None:
            raise NotImplementedError("mixup is not used on the strongly-supervised path")
        if self.training and not self.freeze_bn:
            ops.bump_bn_counters(self, [self.bn0, *(getattr(self, f"conv_block{i}").bn1 for i in range(1, 5)),
                                        *(getattr(self, f"conv_block{i}").bn2 for i in range(1, 5))])
        x = torch_ops.run_encoder(torch.ops.tag.cnn8rnn_encoder, self, waveform, self._flat_params())
        length = torch.div(torch.as_tensor(input_dict["waveform_len"]), self.hop_length, rounding_mode="floor") + 1
        length = torch.div(length, self.downsample_ratio, rounding_mode="floor")
        return {"embedding": x, "length": length}


Cnn8_Rnn = Cnn8Rnn   # stale alias used by some of the reference's eg_configs
