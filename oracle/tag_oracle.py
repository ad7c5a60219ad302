"""CPU restatement (oracle) of the text-to-audio-grounding hot path.

TEST INFRASTRUCTURE ONLY -- never imported by the product package.

Plain PyTorch eager on the CPU, written functionally over a flat ``state`` dict
whose keys and shapes are the reference's ``state_dict`` names, so the same
weights can be loaded into the imported reference (tests/golden/make_golden.py)
and into the HIP path.  Every function cites the reference lines it restates
(paths relative to /root/reference).

Pinning status
--------------
* Rows F3, A1-A5, T1-T2, M0-M3, R1, L1, O1, P1 of SURVEY.md section 8: PINNED by the
  golden vectors under tests/golden/, produced by importing the reference itself
  in the build container (tests/golden/make_golden.py).
* Rows F1/F2 (MelSpectrogram / AmplitudeToDB): the arithmetic lives in
  torchaudio, a third-party dependency that is NOT vendored in the reference,
  not listed in its requirements.txt and not installed here: PARITY UNPINNED.
  ``melscale_fbanks``/``mel_spectrogram``/``amplitude_to_db`` below follow
  torchaudio's published algorithm (torchaudio.functional.melscale_fbanks,
  transforms.Spectrogram -> torch.stft, transforms.AmplitudeToDB) anchored on
  the reference's call sites models/audio_encoder.py:29-37,113-124,68-69,183-184.
  The filterbank is cross-checked against transformers.audio_utils in
  tests/test_oracle_golden.py.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# Frontend (rows F1, F2) -- torchaudio semantics
# --------------------------------------------------------------------------


def _hz_to_mel(freq: float, mel_scale: str) -> float:
    if mel_scale == "htk":
        return 2595.0 * math.log10(1.0 + freq / 700.0)
    f_sp = 200.0 / 3
    mels = freq / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    if freq >= min_log_hz:
        mels = min_log_mel + math.log(freq / min_log_hz) / logstep
    return mels


def _mel_to_hz(mels: torch.Tensor, mel_scale: str) -> torch.Tensor:
    if mel_scale == "htk":
        return 700.0 * (10.0 ** (mels / 2595.0) - 1.0)
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    log_t = mels >= min_log_mel
    freqs[log_t] = min_log_hz * torch.exp(logstep * (mels[log_t] - min_log_mel))
    return freqs


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int,
                    sample_rate: int, norm: Optional[str], mel_scale: str) -> torch.Tensor:
    """(n_freqs, n_mels) triangular filterbank, torchaudio algorithm."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = _hz_to_mel(f_min, mel_scale)
    m_max = _hz_to_mel(f_max, mel_scale)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = _mel_to_hz(m_pts, mel_scale)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.max(torch.zeros(1), torch.min(down, up))
    if norm == "slaney":
        enorm = 2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])
        fb = fb * enorm.unsqueeze(0)
    return fb


#: frontend parameter sets (models/audio_encoder.py:107-123 and :29-35)
FRONTEND = {
    "cnn8rnn": dict(sample_rate=32000, n_fft=1024, win_length=1024, hop_length=320,
                    f_min=50.0, f_max=14000.0, n_mels=64, norm="slaney", mel_scale="slaney"),
    "crnn": dict(sample_rate=32000, n_fft=2048, win_length=1280, hop_length=640,
                 f_min=0.0, f_max=16000.0, n_mels=64, norm=None, mel_scale="htk"),
}


def frontend_tables(kind: str):
    """hann window (win_length,) periodic and fb (n_fft//2+1, n_mels) for a parameter set."""
    p = FRONTEND[kind]
    window = torch.hann_window(p["win_length"])
    fb = melscale_fbanks(p["n_fft"] // 2 + 1, p["f_min"], p["f_max"], p["n_mels"],
                         p["sample_rate"], p["norm"], p["mel_scale"])
    return window, fb


def mel_spectrogram(waveform: torch.Tensor, kind: str, window=None, fb=None) -> torch.Tensor:
    """power mel spectrogram (B, n_mels, F); torchaudio MelSpectrogram(power=2, center=True)."""
    p = FRONTEND[kind]
    if window is None or fb is None:
        window, fb = frontend_tables(kind)
    window = window.to(waveform.dtype)
    fb = fb.to(waveform.dtype)
    spec = torch.stft(waveform, p["n_fft"], p["hop_length"], p["win_length"], window=window,
                      center=True, pad_mode="reflect", normalized=False, onesided=True,
                      return_complex=True)
    power = spec.abs().pow(2.0)                       # (B, bins, F)
    return torch.matmul(power.transpose(-1, -2), fb).transpose(-1, -2)


def amplitude_to_db(x: torch.Tensor) -> torch.Tensor:
    """AmplitudeToDB(stype='power', top_db=None): 10 log10(clamp(x,1e-10)) - 10 log10(max(1e-10,1))."""
    x_db = 10.0 * torch.log10(torch.clamp(x, min=1e-10))
    x_db = x_db - 10.0 * math.log10(max(1e-10, 1.0))
    return x_db


def logmel(waveform: torch.Tensor, kind: str) -> torch.Tensor:
    """(B, n_mels, F) in dB -- rows F1+F2."""
    return amplitude_to_db(mel_spectrogram(waveform, kind))


# --------------------------------------------------------------------------
# State construction (reference init distributions; NOT the reference RNG order)
# --------------------------------------------------------------------------

CNN8_CHANNELS = [(1, 64), (64, 128), (128, 256), (256, 512)]


def _xavier_uniform(shape, gen):
    # nn.init.xavier_uniform_ (models/panns.py:5-11)
    if len(shape) == 4:
        rf = shape[2] * shape[3]
        fan_in, fan_out = shape[1] * rf, shape[0] * rf
    else:
        fan_out, fan_in = shape
    a = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=gen) * 2 - 1) * a


def _uniform(shape, bound, gen):
    return (torch.rand(shape, generator=gen) * 2 - 1) * bound


def init_state(seed: int = 0, vocab_size: int = 5221, text_dim: int = 512, shared_dim: int = 512,
               add_proj: bool = False, logit_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Random state for BiEncoder(Cnn8Rnn, EmbeddingAgg(mean), match) with reference key names.

    ``logit_gain`` scales the embedding table so that frame logits span several
    units (a random-init model gives frame_sim in [0.5001, 0.5042]).
    """
    g = torch.Generator().manual_seed(seed)
    st: Dict[str, torch.Tensor] = {}
    ae = "audio_encoder."

    def bn(prefix, c, randomize):
        st[prefix + "weight"] = torch.ones(c) if not randomize else 0.5 + torch.rand(c, generator=g)
        st[prefix + "bias"] = torch.zeros(c) if not randomize else 0.2 * torch.randn(c, generator=g)
        st[prefix + "running_mean"] = torch.zeros(c)
        st[prefix + "running_var"] = torch.ones(c)
        st[prefix + "num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    randomize = logit_gain != 1.0
    bn(ae + "bn0.", 64, randomize)
    for i, (cin, cout) in enumerate(CNN8_CHANNELS, start=1):
        st[f"{ae}conv_block{i}.conv1.weight"] = _xavier_uniform((cout, cin, 3, 3), g)
        st[f"{ae}conv_block{i}.conv2.weight"] = _xavier_uniform((cout, cout, 3, 3), g)
        bn(f"{ae}conv_block{i}.bn1.", cout, randomize)
        bn(f"{ae}conv_block{i}.bn2.", cout, randomize)
    st[ae + "fc1.weight"] = _xavier_uniform((512, 512), g)
    st[ae + "fc1.bias"] = torch.zeros(512) if not randomize else 0.1 * torch.randn(512, generator=g)
    k = 1.0 / math.sqrt(256)
    for sfx in ("", "_reverse"):
        st[f"{ae}rnn.weight_ih_l0{sfx}"] = _uniform((768, 512), k, g)
        st[f"{ae}rnn.weight_hh_l0{sfx}"] = _uniform((768, 256), k, g)
        st[f"{ae}rnn.bias_ih_l0{sfx}"] = _uniform((768,), k, g)
        st[f"{ae}rnn.bias_hh_l0{sfx}"] = _uniform((768,), k, g)
    # nn.Embedding + kaiming_uniform_ (models/utils.py:19-20): bound = sqrt(6 / fan_in), fan_in = D
    st["text_encoder.embedding.core.weight"] = _uniform((vocab_size, text_dim),
                                                        math.sqrt(6.0 / text_dim), g) * logit_gain
    if text_dim != 512 or add_proj:
        for name, din in (("audio_proj", 512), ("text_proj", text_dim)):
            b = 1.0 / math.sqrt(din)
            st[name + ".weight"] = _uniform((shared_dim, din), b, g)
            st[name + ".bias"] = _uniform((shared_dim,), b, g)
    return st


def state_to(st, dtype=None, requires_grad=False):
    out = {}
    for k, v in st.items():
        if v.is_floating_point():
            v = v.detach().clone().to(dtype or v.dtype)
            if requires_grad and "running_" not in k:
                v.requires_grad_(True)
        else:
            v = v.clone()
        out[k] = v
    return out


# --------------------------------------------------------------------------
# Cnn8Rnn (rows F3, A1-A5)
# --------------------------------------------------------------------------


def _bn(x, st, prefix, training, momentum=0.1, eps=1e-5):
    # nn.BatchNorm2d semantics; running stats are updated in place when training
    return F.batch_norm(x, st[prefix + "running_mean"], st[prefix + "running_var"],
                        st[prefix + "weight"], st[prefix + "bias"], training, momentum, eps)


def _dropout(x, p, training, masks, name):
    """Dropout with an injectable keep-mask so another implementation's RNG can be replayed."""
    if not training or p == 0.0:
        return x
    if masks is not None and name in masks:
        return x * masks[name].to(x.dtype) / (1.0 - p)
    return F.dropout(x, p=p, training=True)


def dropout_keep_mask(seed: int, n: int, p: float) -> np.ndarray:
    """The HIP path's counter-based keep mask restated on the CPU (texttoaudiogrounding_amd/csrc/tag_common.h:
    tag_mix64 / tag_keep): element i is kept iff u >= p with u = (splitmix64(seed * 0xD1342543DE82EF95 + i) >> 40) / 2^24,
    compared in fp32.  Not torch's Philox stream (the reference's F.dropout draws from that, models/audio_encoder.py:203-215);
    the oracle replays THESE masks so that a dropout-on step can be compared element by element."""
    M = np.uint64(0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.arange(n, dtype=np.uint64)) & M
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return u >= np.float32(p)


def dropout_keep_mask4(seed: int, n: int, p: float) -> np.ndarray:
    """The keep mask of the POOLED activations' dropout (csrc/tag_common.h tag_keep4_bits / tag_keep4): one splitmix64 per 4
    consecutive elements, element i kept iff ((splitmix64(seed * 0xD1342543DE82EF95 + (i >> 2)) >> 16 (i & 3)) & 0xFFFF) >=
    ceil(p * 2^16), p in fp32."""
    M = np.uint64(0xFFFFFFFFFFFFFFFF)
    ng = (n + 3) // 4
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.arange(ng, dtype=np.uint64)) & M
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M
        z = z ^ (z >> np.uint64(31))
    u = np.stack([(z >> np.uint64(16 * j)) & np.uint64(0xFFFF) for j in range(4)], axis=1).reshape(-1)[:n]
    thr = np.uint64(int(np.ceil(np.float32(p) * np.float32(65536.0))))
    return u >= thr


def cnn8rnn_dropout_masks(seeds, B: int, n_frames: int, p_drop=(0.2, 0.5), dtype=torch.float32):
    """The five keep masks of one Cnn8Rnn training step for the HIP path's seeds, in the oracle's NCHW layout
    (the HIP kernels index the channels-last pooled outputs (B,H,W,C) and the (B*T',512) mean-pooled rows flat)."""
    shapes, H, W = [], n_frames, 64
    for i, (ph, pw) in enumerate([(2, 2), (2, 2), (1, 2), (1, 2)]):
        H, W = H // ph, W // pw
        shapes.append((B, H, W, 64 << i))
    masks = {}
    for i, shp in enumerate(shapes):
        m = dropout_keep_mask4(seeds[i], int(np.prod(shp)), p_drop[0]).reshape(shp)
        masks[f"drop{i + 1}"] = torch.from_numpy(m).permute(0, 3, 1, 2).to(dtype)
    m = dropout_keep_mask(seeds[4], B * shapes[3][1] * 512, p_drop[1]).reshape(B, shapes[3][1], 512)
    masks["drop5"] = torch.from_numpy(m).to(dtype)
    return masks


def _imposed_max_pool(a, pool_size, index):
    """max_pool2d (kernel = stride, floor) with the arg-max IMPOSED: index (B,C,Ho,Wo) = position dh * pw + dw inside the window."""
    ph, pw = pool_size
    B, C, H, W = a.shape
    Ho, Wo = H // ph, W // pw
    win = a[:, :, :Ho * ph, :Wo * pw].reshape(B, C, Ho, ph, Wo, pw).permute(0, 1, 2, 4, 3, 5).reshape(B, C, Ho, Wo, ph * pw)
    return torch.gather(win, 4, index.long().unsqueeze(-1)).squeeze(-1)


def conv_block(x, st, prefix, pool_size, training, taps=None, decisions=None, bn_training=None):
    """ConvBlock.forward with pool_type='avg+max' (models/panns.py:46-62).

    decisions (test aid, tests/test_gpu_path.py): the hard decisions of ANOTHER implementation imposed on this one -- keys
    "relu/<prefix>bn1", "relu/<prefix>bn2" (0/1 masks, NCHW) and "argmax/<prefix>" (window position per pooled element).  Two
    fp32 implementations whose BatchNorm outputs differ by round-off can take different sides of a ReLU / arg-max whose operands
    are closer than that round-off; with the decisions imposed the remaining difference is round-off only."""
    dec = decisions or {}
    bn_training = training if bn_training is None else bn_training      # Cnn8Rnn(freeze_bn=True): BatchNorm in eval mode while training
    y1 = F.conv2d(x, st[prefix + "conv1.weight"], None, 1, 1)
    b1 = _bn(y1, st, prefix + "bn1.", bn_training)
    a1 = b1 * dec["relu/" + prefix + "bn1"].to(b1.dtype) if "relu/" + prefix + "bn1" in dec else F.relu(b1)
    y2 = F.conv2d(a1, st[prefix + "conv2.weight"], None, 1, 1)
    b2 = _bn(y2, st, prefix + "bn2.", bn_training)
    a2 = b2 * dec["relu/" + prefix + "bn2"].to(b2.dtype) if "relu/" + prefix + "bn2" in dec else F.relu(b2)
    mx = (_imposed_max_pool(a2, pool_size, dec["argmax/" + prefix]) if "argmax/" + prefix in dec
          else F.max_pool2d(a2, kernel_size=pool_size))
    out = F.avg_pool2d(a2, kernel_size=pool_size) + mx
    if taps is not None:
        taps[prefix + "conv1"] = y1
        taps[prefix + "conv2"] = y2
        taps[prefix + "pool"] = out
    return out


def output_length(waveform_len, hop_length: int, downsample_ratio: int = 4) -> torch.Tensor:
    """models/audio_encoder.py:219-227 (and :77-84): floor((floor(len / hop) + 1) / 4), int64."""
    length = torch.div(torch.as_tensor(waveform_len), hop_length, rounding_mode="floor") + 1
    return torch.div(length, downsample_ratio, rounding_mode="floor")


def cnn8rnn_forward(st, waveform, waveform_len, training=False, p_drop=(0.2, 0.5),
                    masks=None, taps=None, prefix="audio_encoder.", bn_training=None):
    """Cnn8Rnn.forward with specaug=False and no mixup (models/audio_encoder.py:178-232).  bn_training=False with training=True is
    Cnn8Rnn(freeze_bn=True).train() (models/audio_encoder.py:159-169: every BatchNorm module stays in eval mode)."""
    bn_training = training if bn_training is None else bn_training
    x = logmel(waveform, "cnn8rnn")                   # (B, 64, F)
    if taps is not None:
        taps["logmel"] = x
    x = x.transpose(1, 2).unsqueeze(1)                # (B, 1, F, 64)
    x = x.transpose(1, 3)
    x = _bn(x, st, prefix + "bn0.", bn_training)
    x = x.transpose(1, 3)
    if taps is not None:
        taps["bn0"] = x
    pools = [(2, 2), (2, 2), (1, 2), (1, 2)]
    for i, ps in enumerate(pools, start=1):
        x = conv_block(x, st, f"{prefix}conv_block{i}.", ps, training, taps, decisions=masks, bn_training=bn_training)
        x = _dropout(x, p_drop[0], training, masks, f"drop{i}")
    x = torch.mean(x, dim=3)                          # (B, 512, T')
    x = x.transpose(1, 2)
    x = _dropout(x, p_drop[1], training, masks, "drop5")
    x = F.linear(x, st[prefix + "fc1.weight"], st[prefix + "fc1.bias"])
    x = x * masks["relu/" + prefix + "fc1"].to(x.dtype) if masks is not None and "relu/" + prefix + "fc1" in masks else F.relu(x)
    if taps is not None:
        taps["fc1"] = x
    x = gru_bidir(x, st, prefix + "rnn.")
    length = output_length(waveform_len, FRONTEND["cnn8rnn"]["hop_length"])
    return {"embedding": x, "length": length}


# --------------------------------------------------------------------------
# BASELINE configs[2] as this build realises it ("bf16 mode"): NOT a reference path -- the reference is
# fp32 everywhere -- but the SAME forward arithmetic with the mode's rounding points made explicit, so that the HIP bf16 path
# can be held to "bf16 rounding at exactly these places and nothing else" instead of a loose budget against the fp64 oracle:
#   * raw conv outputs and pooled activations are STORED as bf16 (round-to-nearest-even of the fp32 accumulator / pool value);
#   * BatchNorm batch statistics come from the accumulators BEFORE that rounding; the affine is applied to the stored value;
#   * the operand a 3x3 conv multiplies is bf16: relu(bn(y)) is rounded once more when a BN+ReLU prologue feeds a conv, conv
#     weights are rounded to bf16 (the Cin = 1 conv runs on fp32 VALU: neither its input nor its weights are rounded);
#   * fc1 and the GRU input projection round both GEMM operands to bf16 and accumulate in fp32; the recurrence, the heads and
#     the loss are fp32.
# Evaluate in float64 so that the only differences from the device are its fp32 accumulation order and rounding ties.
# --------------------------------------------------------------------------
def _q_bf16(t):
    return t.to(torch.bfloat16).to(t.dtype)


def _bn_affine(y, st, prefix, training, eps=1e-5):
    """(scale, shift) of BatchNorm2d over an NCHW tensor: batch statistics (biased variance) or the running ones."""
    if training:                                   # (y may be None when the running statistics are asked for)
        mean, var = y.mean(dim=(0, 2, 3)), y.var(dim=(0, 2, 3), unbiased=False)
    else:
        mean, var = st[prefix + "running_mean"], st[prefix + "running_var"]
    scale = st[prefix + "weight"] / torch.sqrt(var + eps)
    return scale.view(1, -1, 1, 1), (st[prefix + "bias"] - mean * scale).view(1, -1, 1, 1)


def cnn8rnn_forward_bf16_mode(st, waveform, waveform_len, training=False, prefix="audio_encoder."):
    """The Cnn8Rnn forward with the bf16 mode's storage / operand roundings (dropout off); same contract as cnn8rnn_forward."""
    x = logmel(waveform, "cnn8rnn").transpose(1, 2).unsqueeze(1)      # (B, 1, F, 64)
    s0, t0 = _bn_affine(x.transpose(1, 3), st, prefix + "bn0.", training)
    x = (x.transpose(1, 3) * s0 + t0).transpose(1, 3)                  # bn0 over the mel axis, fp32 on the device
    pools = [(2, 2), (2, 2), (1, 2), (1, 2)]
    for i, ps in enumerate(pools, start=1):
        bp = f"{prefix}conv_block{i}."
        w1, w2 = st[bp + "conv1.weight"], st[bp + "conv2.weight"]
        y1f = F.conv2d(x, w1 if i == 1 else _q_bf16(w1), None, 1, 1)   # block 1: fp32 VALU conv of the fp32 log-mel
        s1, t1 = _bn_affine(y1f, st, bp + "bn1.", training)
        a1 = _q_bf16(F.relu(_q_bf16(y1f) * s1 + t1))                    # stored bf16 -> affine + ReLU -> bf16 operand
        y2f = F.conv2d(a1, _q_bf16(w2), None, 1, 1)
        s2, t2 = _bn_affine(y2f, st, bp + "bn2.", training)
        a2 = F.relu(_q_bf16(y2f) * s2 + t2)                             # the pool reads the stored value; no operand rounding
        x = _q_bf16(F.avg_pool2d(a2, kernel_size=ps) + F.max_pool2d(a2, kernel_size=ps))
    x = torch.mean(x, dim=3).transpose(1, 2)                           # (B, T', 512), fp32 on the device
    x = F.relu(F.linear(_q_bf16(x), _q_bf16(st[prefix + "fc1.weight"]), st[prefix + "fc1.bias"]))
    gst = dict(st)
    for sfx in ("", "_reverse"):
        gst[prefix + "rnn.weight_ih_l0" + sfx] = _q_bf16(st[prefix + "rnn.weight_ih_l0" + sfx])
    x = gru_bidir(_q_bf16(x), gst, prefix + "rnn.")
    return {"embedding": x, "length": output_length(waveform_len, FRONTEND["cnn8rnn"]["hop_length"])}


# ---- the bf16 mode's BACKWARD rounding points, stage by stage (round 4; the backward twin of cnn8rnn_forward_bf16_mode).  Every
# function takes the tensors a device stage reads AS STORED (bf16 values widened to float64, NCHW) plus the fp32 per-channel
# constants, computes in float64 and rounds exactly where the kernels round.  (models/panns.py:46-62 backward.)

def bf16_bnrelu_pool_backward(y, scale, shift, mean, invstd, gamma, dout, ps):
    """Backward of  out = avg_pool(a) + max_pool(a),  a = relu(y * scale + shift)  (ConvBlock pool_type 'avg+max') and of the
    train-mode BatchNorm in front of it, as tag_bnrelu_pool_backward_bf16 does it (csrc/bn_pool.hip PoolBwdCtx::compute,
    pool_bwd_reduce_kernel, pool_bwd_apply_kernel): the max gradient goes to the FIRST maximum in scan order (ATen's rule),
    sums over the batch in high precision, ONE rounding -- of dy to bf16 on store.  -> (dy (bf16-rounded), dgamma, dbeta)."""
    v = lambda t: t.view(1, -1, 1, 1)
    a = (y * v(scale) + v(shift)).detach().requires_grad_(True)
    r = F.relu(a)
    (F.avg_pool2d(r, kernel_size=ps) + F.max_pool2d(r, kernel_size=ps)).backward(dout)
    g = a.grad                                                      # dz: pool routing and the ReLU mask
    xh = (y - v(mean)) * v(invstd)
    n = y.shape[0] * y.shape[2] * y.shape[3]
    dbeta, dgamma = g.sum(dim=(0, 2, 3)), (g * xh).sum(dim=(0, 2, 3))
    dy = v(gamma * invstd) * (g - v(dbeta) / n - xh * v(dgamma) / n)
    return _q_bf16(dy), dgamma, dbeta


def bf16_conv_dgrad(dy, w, in_shape):
    """Input gradient of the 3x3 convolution on the bf16 MFMA: bf16 operands (dy as stored, the weights rounded), wide
    accumulation; UNROUNDED (the epilogue sums of tag_conv3x3_dgrad_bnsums_bf16 use the accumulators, the store rounds)."""
    return torch.nn.grad.conv2d_input(in_shape, _q_bf16(w), dy, 1, 1)


def bf16_dgrad_bnrelu_backward(da_f, yref, scale, shift, mean, invstd, gamma):
    """What follows the dgrad conv whose result flows into relu(bn(yref)) (tag_conv3x3_dgrad_bnsums_bf16 +
    tag_bn_grad_from_partials + tag_bnrelu_backward_apply_bf16; csrc/conv_x3.hip EPI == 1, csrc/conv_rows.hip EPI == 2,
    csrc/bn_pool.hip bnrelu_bwd_apply_kernel): dgamma / dbeta sum g = da * [bn(yref) > 0] from the UNROUNDED accumulators, the
    apply pass re-reads the STORED (bf16) da.  -> (dy (bf16-rounded), dgamma, dbeta, da (bf16-rounded, what the conv stored))."""
    v = lambda t: t.view(1, -1, 1, 1)
    mask = (yref * v(scale) + v(shift)) > 0
    xh = (yref - v(mean)) * v(invstd)
    gf = torch.where(mask, da_f, torch.zeros_like(da_f))
    n = yref.shape[0] * yref.shape[2] * yref.shape[3]
    dbeta, dgamma = gf.sum(dim=(0, 2, 3)), (gf * xh).sum(dim=(0, 2, 3))
    da_q = _q_bf16(da_f)
    gq = torch.where(mask, da_q, torch.zeros_like(da_q))
    dy = v(gamma * invstd) * (gq - v(dbeta) / n - xh * v(dgamma) / n)
    return _q_bf16(dy), dgamma, dbeta, da_q


def bf16_conv_wgrad(x, dy, wshape, scale=None, shift=None):
    """Weight gradient on the bf16 MFMA: operands as stored; with a producer prologue the activation relu(x * scale + shift)
    is formed in fp32 and ROUNDED to bf16 before it multiplies (the same operand the forward conv used); fp32 result."""
    if scale is not None:
        x = _q_bf16(F.relu(x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)))
    return torch.nn.grad.conv2d_weight(x, wshape, dy, 1, 1)


def gru_bidir(x, st, prefix):
    """nn.GRU(512, 256, bidirectional=True, batch_first=True), h0 = 0, all T' steps (row A4)."""
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    flat = [st[prefix + n] for n in names] + [st[prefix + n + "_reverse"] for n in names]
    hidden = flat[1].shape[1]
    h0 = x.new_zeros(2, x.shape[0], hidden)
    out, _ = torch.gru(x, h0, flat, True, 1, 0.0, False, True, True)
    return out


def gru_bidir_manual(x, st, prefix):
    """Explicit gate arithmetic (PyTorch order r, z, n); used to cross-check ``gru_bidir``."""
    B, T, _ = x.shape
    outs = []
    for sfx, order in (("", range(T)), ("_reverse", range(T - 1, -1, -1))):
        w_ih, w_hh = st[f"{prefix}weight_ih_l0{sfx}"], st[f"{prefix}weight_hh_l0{sfx}"]
        b_ih, b_hh = st[f"{prefix}bias_ih_l0{sfx}"], st[f"{prefix}bias_hh_l0{sfx}"]
        H = w_hh.shape[1]
        h = x.new_zeros(B, H)
        ys = [None] * T
        for t in order:
            gi = F.linear(x[:, t], w_ih, b_ih)
            gh = F.linear(h, w_hh, b_hh)
            r = torch.sigmoid(gi[:, :H] + gh[:, :H])
            z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
            n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
            h = (1 - z) * n + z * h
            ys[t] = h
        outs.append(torch.stack(ys, 1))
    return torch.cat(outs, -1)


# --------------------------------------------------------------------------
# CrnnEncoder (row A1') -- what the strong eg_config literally instantiates
# --------------------------------------------------------------------------


def init_crnn_state(seed=0, embed_dim=256, prefix="audio_encoder."):
    g = torch.Generator().manual_seed(seed)
    st = {}
    chans = {0: (1, 32), 2: (32, 128), 3: (128, 128), 5: (128, 128), 6: (128, 128)}
    for idx, (cin, cout) in chans.items():
        p = f"{prefix}cnn.{idx}."
        st[p + "0.weight"] = torch.ones(cin)
        st[p + "0.bias"] = torch.zeros(cin)
        st[p + "0.running_mean"] = torch.zeros(cin)
        st[p + "0.running_var"] = torch.ones(cin)
        st[p + "0.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
        std = math.sqrt(2.0 / (cin * 9))              # kaiming_normal_ (models/utils.py:6-7)
        st[p + "1.weight"] = torch.randn((cout, cin, 3, 3), generator=g) * std
    H = embed_dim // 2
    k = 1.0 / math.sqrt(H)
    for sfx in ("", "_reverse"):
        st[f"{prefix}gru.weight_ih_l0{sfx}"] = _uniform((3 * H, 128), k, g)
        st[f"{prefix}gru.weight_hh_l0{sfx}"] = _uniform((3 * H, H), k, g)
        st[f"{prefix}gru.bias_ih_l0{sfx}"] = _uniform((3 * H,), k, g)
        st[f"{prefix}gru.bias_hh_l0{sfx}"] = _uniform((3 * H,), k, g)
    return st


def _lppool4(x, k):
    # nn.LPPool2d(4, k): (sum_window x^4)^(1/4), stride = kernel, floor
    return F.lp_pool2d(x, 4.0, k)


def crnn_forward(st, waveform, waveform_len, training=False, p_drop=0.3, masks=None, taps=None,
                 prefix="audio_encoder."):
    """CrnnEncoder.forward (models/audio_encoder.py:16-22,39-49,66-86)."""
    x = logmel(waveform, "crnn").transpose(1, 2).unsqueeze(1)   # (B,1,F,64)
    if taps is not None:
        taps["logmel"] = x

    def block(x, idx):
        p = f"{prefix}cnn.{idx}."
        x = _bn(x, st, p + "0.", training)
        x = F.conv2d(x, st[p + "1.weight"], None, 1, 1)
        return F.leaky_relu(x, 0.1)

    x = _lppool4(block(x, 0), (2, 4))
    x = _lppool4(block(block(x, 2), 3), (2, 4))
    x = _lppool4(block(block(x, 5), 6), (1, 4))
    x = _dropout(x, p_drop, training, masks, "drop")
    x = x.transpose(1, 2).contiguous().flatten(-2)
    if taps is not None:
        taps["cnn"] = x
    x = gru_bidir(x, st, prefix + "gru.")
    length = output_length(waveform_len, FRONTEND["crnn"]["hop_length"])
    return {"embedding": x, "length": length}


# --------------------------------------------------------------------------
# Text encoder (rows T1, T2), heads (M0-M3), loss (R1, L1)
# --------------------------------------------------------------------------


def length_mask(lens, max_length=None) -> torch.Tensor:
    """models/utils.py:22-30: arange(L) < len."""
    lens = torch.as_tensor(lens)
    if max_length is None:
        max_length = int(lens.max().item())
    return torch.arange(max_length).unsqueeze(0) < lens.view(-1, 1)


def embedding_agg_mean(st, text, text_len, prefix="text_encoder."):
    """EmbeddingAgg(aggregation='mean') (models/text_encoder.py:79-88; models/utils.py:33-58)."""
    table = st[prefix + "embedding.core.weight"]
    embs = F.embedding(text.long(), table)            # (B, L, D)
    lens = torch.as_tensor(text_len)
    mask = length_mask(lens, embs.size(1)).unsqueeze(-1).to(embs.dtype)
    seq = (embs * mask).sum(1) / lens.view(-1, 1).to(embs.dtype)
    return {"token_emb": embs, "seq_emb": seq}


def match_exp_neg_l2(audio, text, l2norm=True):
    """match.ExpNegL2 (models/match.py:16-33), text_level='seq'."""
    if l2norm:
        audio = F.normalize(audio, dim=-1)
        text = F.normalize(text, dim=-1)
    diff = audio - text.unsqueeze(1)
    return torch.exp(-torch.norm(diff, dim=-1))


def match_dot_product(audio, text, l2norm=False, scale=True, return_logit=False):
    """match.DotProduct (models/match.py:43-60), text_level='seq'."""
    if l2norm:
        audio = F.normalize(audio, dim=-1)
        text = F.normalize(text, dim=-1)
    score = (audio * text.unsqueeze(1)).sum(-1)
    if scale:
        score = score / math.sqrt(audio.size(-1))
    if return_logit:
        return score
    return torch.sigmoid(score).clamp(1e-7, 1.0)


def align_dot_product(audio, text, l2norm=False, scaled=False):
    """align.DotProduct (models/align.py:14-31): (B,T,D),(B,N,D) -> (B,B,T,N)."""
    if l2norm:
        audio = F.normalize(audio, dim=-1)
        text = F.normalize(text, dim=-1)
    B, T, D = audio.shape
    N = text.shape[1]
    score = audio.reshape(-1, D) @ text.reshape(-1, D).t()
    if scaled:
        score = score / math.sqrt(D)
    score = torch.sigmoid(score).clamp(1e-7, 1.0)
    return score.reshape(B, T, B, N).transpose(1, 2)


def biencoder_forward(st, batch, match="dot", audio="cnn8rnn", training=False, p_drop=None,
                      masks=None, taps=None, bn_training=None):
    """BiEncoder.forward (models/audio_text_model.py:58-98), cross_encoder=None, upsample=False."""
    if audio == "cnn8rnn":
        kw = {} if p_drop is None else {"p_drop": p_drop}
        if bn_training is not None:
            kw["bn_training"] = bn_training
        ao = cnn8rnn_forward(st, batch["waveform"], batch["waveform_len"], training, masks=masks,
                             taps=taps, **kw)
    else:
        kw = {} if p_drop is None else {"p_drop": p_drop}
        ao = crnn_forward(st, batch["waveform"], batch["waveform_len"], training, masks=masks,
                          taps=taps, **kw)
    audio_emb = ao["embedding"]
    te = embedding_agg_mean(st, batch["text"], batch["text_len"])
    seq = te["seq_emb"]
    if "audio_proj.weight" in st:
        audio_emb = F.linear(audio_emb, st["audio_proj.weight"], st["audio_proj.bias"])
        seq = F.linear(seq, st["text_proj.weight"], st["text_proj.bias"])
    if taps is not None:
        taps["audio_emb"] = audio_emb
        taps["seq_emb"] = seq
    if match == "dot":
        if taps is not None:
            taps["logit"] = match_dot_product(audio_emb, seq, return_logit=True)
        sim = match_dot_product(audio_emb, seq)
    elif match == "expnegl2":
        sim = match_exp_neg_l2(audio_emb, seq)
    else:
        raise ValueError(match)
    return {"frame_sim": sim, "length": ao["length"]}


def runner_truncate(output, label):
    """Runner.forward label alignment (python_scripts/training/run_strong.py:107-118)."""
    frame_sim = output["frame_sim"]
    tt = min(frame_sim.size(1), label.size(1))
    return {"frame_sim": frame_sim[..., :tt], "label": label[..., :tt],
            "length": torch.clamp(output["length"], 1, tt)}


def frame_bce_loss(frame_sim, label, length):
    """FrameBceLoss.forward (losses.py:12-24)."""
    loss = F.binary_cross_entropy(frame_sim, label, reduction="none")
    mask = length_mask(length).to(frame_sim.dtype)
    if mask.shape[1] < loss.shape[1]:     # generate_length_mask uses max(length) columns
        mask = F.pad(mask, (0, loss.shape[1] - mask.shape[1]))
    return (loss * mask).sum() / mask.sum()


def train_step_loss(st, batch, match="dot", audio="cnn8rnn", training=True, p_drop=None,
                    masks=None, taps=None, bn_training=None):
    """zero_grad -> forward -> FrameBceLoss, the timed unit of BASELINE.md section 3."""
    out = biencoder_forward(st, batch, match, audio, training, p_drop, masks, taps, bn_training)
    out = runner_truncate(out, batch["label"])
    return frame_bce_loss(out["frame_sim"], out["label"], out["length"]), out


# --------------------------------------------------------------------------
# Synthetic batch (BASELINE.md section 3 / SURVEY section 8d)
# --------------------------------------------------------------------------


def synthetic_batch(batch_size: int, n_samples: int = 320000, seed: int = 1234, ragged: bool = False,
                    hop: int = 320, vocab_size: int = 5221):
    g = torch.Generator().manual_seed(seed)
    waveform = 0.1 * torch.randn(batch_size, n_samples, generator=g)
    if ragged:
        lens = torch.randint(n_samples // 2, n_samples + 1, (batch_size,), generator=g)
        lens[0] = n_samples
        for i in range(batch_size):
            waveform[i, lens[i]:] = 0.0
    else:
        lens = torch.full((batch_size,), n_samples, dtype=torch.long)
    text = torch.randint(2, vocab_size, (batch_size, 4), generator=g)
    text_len = 1 + torch.arange(batch_size) % 4
    for i in range(batch_size):
        text[i, text_len[i]:] = 0                     # pad id 0 (utils/build_vocab.py:42-43)
    t_out = (n_samples // hop + 1) // 4
    label = (torch.rand(batch_size, t_out, generator=g) < 0.5).float()
    return {"waveform": waveform, "waveform_len": lens.numpy(), "text": text,
            "text_len": text_len.numpy(), "label": label}


# --------------------------------------------------------------------------
# Post-processing (row P1) -- integer work, numpy
# --------------------------------------------------------------------------


def binarize(x: np.ndarray, threshold) -> np.ndarray:
    """eval_util.binarize (utils/eval_util.py:47-52 -> sklearn.preprocessing.binarize):
    strict ``x > threshold`` evaluated in float64 (the thresholds are np.float64 scalars,
    run_strong.py:203-205, which promote the float32 scores), result 0/1 in x's dtype."""
    x = np.asarray(x)
    return (x.astype(np.float64) > np.float64(threshold)).astype(x.dtype)


def median_filter_1d(b: np.ndarray, window_size: int) -> np.ndarray:
    """scipy.ndimage.median_filter along time with mode='reflect' (utils/eval_util.py:55-63).

    Window of ``size`` samples covering [i - size//2, i - size//2 + size - 1], boundary
    'reflect' = (d c b a | a b c d | d c b a); element of rank size//2 of the sorted window.
    """
    b = np.asarray(b)
    if window_size <= 1:
        return b.copy()
    T = b.shape[-1]
    left = window_size // 2
    idx = np.arange(-left, T + window_size - left - 1)
    period = 2 * T
    idx = np.mod(idx, period)
    idx = np.where(idx >= T, period - 1 - idx, idx)
    padded = b[..., idx]
    win = np.lib.stride_tricks.sliding_window_view(padded, window_size, axis=-1)
    return np.sort(win, axis=-1)[..., window_size // 2].astype(b.dtype)


def find_contiguous_regions(a: np.ndarray) -> np.ndarray:
    """utils/eval_util.py:18-44 -> (K, 2) int64 rows [onset, offset)."""
    a = np.asarray(a).astype(bool)
    out = []
    start = None
    for i, v in enumerate(a):
        if v and start is None:
            start = i
        elif not v and start is not None:
            out.append((start, i))
            start = None
    if start is not None:
        out.append((start, a.size))
    return np.asarray(out, dtype=np.int64).reshape(-1, 2)


def connect_clusters(x: np.ndarray, n: int) -> np.ndarray:
    """utils/eval_util.py:74-116: merge regions whose gap (next onset - cur offset) <= n."""
    x = np.asarray(x)
    reg = find_contiguous_regions(x)
    out = np.zeros_like(x, dtype=int)
    if len(reg) == 0:
        return out
    start, end = reg[0]
    merged = []
    for cur, nxt in zip(reg[:-1], reg[1:]):
        end = nxt[1]
        if nxt[0] - cur[1] > n:
            merged.append((start, cur[1]))
            start = nxt[0]
    merged.append((start, end))
    for s, e in merged:
        out[s:e] = 1
    return out


def segments(frame_sim_row: np.ndarray, threshold, window_size: int, n_connect: int) -> np.ndarray:
    """The per-(clip, threshold) chain of run_strong.py:234-252."""
    b = median_filter_1d(binarize(frame_sim_row, threshold), window_size)
    return find_contiguous_regions(connect_clusters(b, n_connect))


def eval_thresholds(n_thresholds: int = 50) -> np.ndarray:
    """run_strong.py:203-205."""
    return np.arange(1 / (n_thresholds * 2), 1, 1 / n_thresholds)


# ---------------------------------------------------------------------------------------------------------------
# BASELINE configs[3]: cross-encoder (models/cross_encoder.py:5-79) plugged into BiEncoder (models/audio_text_model.py:
# 74-77) with match.DotProduct(text_level="token") (models/match.py:43-60)
# ---------------------------------------------------------------------------------------------------------------
def seq2seq_attention(st, query, kv, query_len, kv_len, prefix="cross_encoder.attn."):
    """Seq2SeqAttention.forward (models/cross_encoder.py:12-42): additive attention of every audio frame over the phrase
    tokens.  score[b,q,k] = v . tanh(W [query[b,q] ; kv[b,k]] + bias), rows q >= query_len and columns k >= kv_len filled
    with -1e10, softmax over k, out = attn @ kv.  W = h2attn.weight (D_attn, D_q + D_kv) is applied as two products."""
    W, b, v = st[prefix + "h2attn.weight"], st[prefix + "h2attn.bias"], st[prefix + "v"]
    dq = query.shape[-1]
    aq = F.linear(query, W[:, :dq])                                   # (B,Tq,Da)
    ak = F.linear(kv, W[:, dq:], b)                                   # (B,Lk,Da)
    score = (torch.tanh(aq.unsqueeze(2) + ak.unsqueeze(1)) * v).sum(-1)          # (B,Tq,Lk)
    Tq, Lk = query.shape[1], kv.shape[1]
    qm = torch.arange(Tq)[None, :] < torch.as_tensor(query_len).view(-1, 1)
    km = torch.arange(Lk)[None, :] < torch.as_tensor(kv_len).view(-1, 1)
    score = score.masked_fill(~qm.unsqueeze(-1), -1e10).masked_fill(~km.unsqueeze(1), -1e10)
    attn = torch.softmax(score, dim=-1)
    return torch.bmm(attn, kv)


def cross_attention_gating(st, audio_emb, token_emb, audio_len, text_len, prefix="cross_encoder."):
    """CrossAttentionGating.forward (models/cross_encoder.py:67-79): text <- attention(audio -> tokens); then CrossGating
    (:51-57): s_out = s * sigmoid(fc_u(u)), u_out = u * sigmoid(fc_s(s)) with u = audio, s = attended text."""
    s = seq2seq_attention(st, audio_emb, token_emb, audio_len, text_len, prefix + "attn.")
    g_u = torch.sigmoid(F.linear(audio_emb, st[prefix + "gating.fc_u.weight"], st[prefix + "gating.fc_u.bias"]))
    g_s = torch.sigmoid(F.linear(s, st[prefix + "gating.fc_s.weight"], st[prefix + "gating.fc_s.bias"]))
    return audio_emb * g_s, s * g_u


def match_dot_product_token(audio, text, scale=True):
    """match.DotProduct with text_level='token' after a cross-encoder: text is (B,T,D), one vector per frame."""
    score = (audio * text).sum(-1)
    if scale:
        score = score / math.sqrt(audio.size(-1))
    return torch.sigmoid(score).clamp(1e-7, 1.0)


def match_cross_attention(st, audio, token, text_len, num_heads, prefix="match_fn.", attn_keep=None, res_keep=None,
                          p_drop=0.0):
    """match.CrossAttention (models/match.py:63-88): nn.MultiheadAttention(E, H, p, batch_first=True, kdim = vdim = kvdim)
    restated -- q/k/v projections (one in_proj_weight (3E,E) when kvdim == E, else q/k/v_proj_weight) + in_proj_bias,
    per-head softmax(q k^T / sqrt(E/H)) with -inf on tokens >= text_len, dropout on the weights, out_proj; then
    audio + dropout(out), LayerNorm (eps 1e-5), Linear(E,1), sigmoid.  attn_keep (B,T,H,L) / res_keep (B,T,E): keep masks
    replayed from the HIP generator for a dropout-on comparison (None = no dropout)."""
    B, T, E = audio.shape
    L = token.shape[1]
    H, dh = num_heads, E // num_heads
    if prefix + "attn.in_proj_weight" in st:
        w = st[prefix + "attn.in_proj_weight"]
        wq, wk, wv = w[:E], w[E:2 * E], w[2 * E:]
    else:
        wq, wk, wv = (st[prefix + f"attn.{n}_proj_weight"] for n in "qkv")
    b = st[prefix + "attn.in_proj_bias"]
    q = F.linear(audio, wq, b[:E]).view(B, T, H, dh).transpose(1, 2)            # (B,H,T,dh)
    k = F.linear(token, wk, b[E:2 * E]).view(B, L, H, dh).transpose(1, 2)
    v = F.linear(token, wv, b[2 * E:]).view(B, L, H, dh).transpose(1, 2)
    score = q @ k.transpose(-1, -2) / math.sqrt(dh)                              # (B,H,T,L)
    pad = ~length_mask(torch.as_tensor(text_len), L).to(torch.bool)             # (B,L) True = padding
    score = score.masked_fill(pad[:, None, None, :], float("-inf"))
    attn = torch.softmax(score, dim=-1)
    if attn_keep is not None:
        attn = attn * attn_keep.permute(0, 2, 1, 3).to(attn.dtype) / (1.0 - p_drop)
    ctx = (attn @ v).transpose(1, 2).reshape(B, T, E)
    out = F.linear(ctx, st[prefix + "attn.out_proj.weight"], st[prefix + "attn.out_proj.bias"])
    if res_keep is not None:
        out = out * res_keep.to(out.dtype) / (1.0 - p_drop)
    z = F.layer_norm(audio + out, (E,), st[prefix + "norm.weight"], st[prefix + "norm.bias"], 1e-5)
    return torch.sigmoid(F.linear(z, st[prefix + "linear.weight"], st[prefix + "linear.bias"])).squeeze(-1)


def init_cross_state(seed, dim=512, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    k = 1.0 / math.sqrt(2 * dim)
    u = lambda *s, a=1.0: (torch.rand(*s, generator=g) * 2 - 1) * a
    return {"cross_encoder.attn.h2attn.weight": u(dim, 2 * dim, a=k * scale), "cross_encoder.attn.h2attn.bias": u(dim, a=k),
            "cross_encoder.attn.v": torch.randn(dim, generator=g),
            "cross_encoder.gating.fc_u.weight": u(dim, dim, a=scale / math.sqrt(dim)),
            "cross_encoder.gating.fc_u.bias": u(dim, a=1 / math.sqrt(dim)),
            "cross_encoder.gating.fc_s.weight": u(dim, dim, a=scale / math.sqrt(dim)),
            "cross_encoder.gating.fc_s.bias": u(dim, a=1 / math.sqrt(dim))}


# ---------------------------------------------------------------------------------------------------------------
# Weak supervision: MultiTextBiEncoder head (models/audio_text_model.py:150-215),
# linear_softmax_with_lens (models/utils.py:22-40,75-76) and ClipBceLoss (losses.py:38-43)
# ---------------------------------------------------------------------------------------------------------------
def linear_softmax_with_lens(features, lens):
    """features (B,T,...) -> sum_{t<len} f^2 / sum_{t<len} f."""
    lens = torch.as_tensor(lens)
    mask = (torch.arange(features.shape[1])[None, :] < lens.view(-1, 1)).to(features.dtype)
    while mask.ndim < features.ndim:
        mask = mask.unsqueeze(-1)
    return (features * features * mask).sum(1) / (features * mask).sum(1)


def attention_pooling(x, lens, w, b):
    """AttentionPooling (models/text_encoder.py:46-58): fc = Linear(D,1); -1e10 fill beyond the length; softmax; weighted sum."""
    score = F.linear(x, w, b).squeeze(-1)
    mask = length_mask(torch.as_tensor(lens), x.size(1)).to(torch.bool)
    score = score.masked_fill(~mask, -1e10)
    return (x * torch.softmax(score, dim=1).unsqueeze(-1)).sum(1)


def upsample_linear(frame_sim, ratio):
    """BiEncoder(upsample=True) (models/audio_text_model.py:90-97)."""
    return F.interpolate(frame_sim.unsqueeze(1), frame_sim.size(1) * ratio, mode="linear", align_corners=False).squeeze(1)


def sum_with_lens(features, lens):
    """models/utils.py:33-46."""
    mask = length_mask(torch.as_tensor(lens), features.size(1)).to(features.dtype)
    while mask.ndim < features.ndim:
        mask = mask.unsqueeze(-1)
    return (features * mask).sum(1)


def mean_with_lens(features, lens):
    """models/utils.py:49-58."""
    s = sum_with_lens(features, lens)
    lens = torch.as_tensor(lens)
    while lens.ndim < s.ndim:
        lens = lens.unsqueeze(1)
    return s / lens.to(features.dtype)


def max_with_lens(features, lens):
    """models/utils.py:61-72 (positions >= len are -inf; torch.max picks the first maximum)."""
    mask = length_mask(torch.as_tensor(lens), features.size(1)).to(torch.bool)
    f = features.clone()
    f[~mask] = float("-inf")
    return f.max(1)[0]


def exp_softmax_with_lens(features, lens):
    """models/utils.py:79-84."""
    normed = features - features.max(1, keepdim=True)[0]
    e = torch.exp(normed)
    weight = e / sum_with_lens(e, lens).unsqueeze(1)
    return sum_with_lens(weight * features, lens)


SEQ_POOL = {"mean": mean_with_lens, "max": max_with_lens, "linear_softmax": linear_softmax_with_lens,
            "exp_softmax": exp_softmax_with_lens}


def sim_pooling(sim, audio_len, text_len, audio_mode, text_mode):
    """The reducers of models/sim_pooling.py:6-189 on a (B,B,T,N) matrix: <audio_mode>_with_lens over the frames, then
    mean / sum / max / mean+sum over the tokens."""
    B, _, T, N = sim.shape
    x = sim.reshape(B * B, T, N)
    al = torch.as_tensor(audio_len).unsqueeze(1).expand(B, B).reshape(-1)
    x = SEQ_POOL[audio_mode](x, al)                               # (B*B, N)
    tl = torch.as_tensor(text_len).repeat(B)
    if text_mode == "mean":
        x = mean_with_lens(x, tl)
    elif text_mode == "sum":
        x = sum_with_lens(x, tl)
    elif text_mode == "max":
        x = max_with_lens(x, tl)
    else:
        x = sum_with_lens(x, tl) + mean_with_lens(x, tl)
    return x.reshape(B, B)


def multitext_head(audio_emb, seq_emb, length, n_text, scale=True):
    """audio (B,T,D), seq_emb (B*N,D): expand the audio per phrase, DotProduct(seq), reshape to (B,T,N), pool."""
    B, T, D = audio_emb.shape
    a = audio_emb.unsqueeze(1).expand(-1, n_text, -1, -1).reshape(B * n_text, T, D)
    fs = match_dot_product(a, seq_emb, scale=scale)                       # (B*N, T)
    frame_sim = fs.reshape(B, n_text, T).transpose(1, 2)                 # (B, T, N)
    return frame_sim, linear_softmax_with_lens(frame_sim, length)


def clip_bce_loss(clip_sim, label):
    return F.binary_cross_entropy(clip_sim, label)


def audio_mean_text_mean(sim, audio_len, text_len):
    """sim_pooling.AudioMeanTextMean (models/sim_pooling.py:6-22): (B,B,T,N) -> (B,B); row a uses audio_len[a], column b
    text_len[b]."""
    B, _, T, N = sim.shape
    am = (torch.arange(T)[None, :] < torch.as_tensor(audio_len).view(-1, 1)).to(sim.dtype)        # (B,T)
    tm = (torch.arange(N)[None, :] < torch.as_tensor(text_len).view(-1, 1)).to(sim.dtype)         # (B,N)
    s = (sim * am[:, None, :, None]).sum(2) / torch.as_tensor(audio_len).to(sim.dtype).view(-1, 1, 1)     # (B,B,N)
    return (s * tm[None, :, :]).sum(2) / torch.as_tensor(text_len).to(sim.dtype).view(1, -1)


def max_margin_ranking_loss(x, margin=1.0, lamda1=1.0):
    """MaxMarginRankingLoss(fix_norm=True) (losses.py:226-264)."""
    n = x.shape[0]
    d = torch.diag(x).view(-1, 1)
    off = ~torch.eye(n, dtype=torch.bool)
    t1 = F.relu(margin - (d - x))[off]
    t2 = F.relu(margin - (d - lamda1 * x.t()))[off]
    return torch.cat([t1, t2]).mean()


# --------------------------------------------------------------------------
# Winograd F(2x2, 3x3) restatement of the ConvBlock convolutions (row A1) -- the ALGORITHM of csrc/conv_wino.hip stage by
# stage (input transform, 16 products, output transform; the weight gradient as the adjoint), in the dtype of its inputs.
# The convolution being restated is nn.Conv2d(kernel 3, stride 1, padding 1, bias=False) of models/panns.py:29-38 as applied in
# models/panns.py:49-50; tests/test_oracle_golden.py pins these functions to F.conv2d and to autograd's weight gradient.
# --------------------------------------------------------------------------
WINO_BT = ((1.0, 0.0, -1.0, 0.0), (0.0, 1.0, 1.0, 0.0), (0.0, -1.0, 1.0, 0.0), (0.0, 1.0, 0.0, -1.0))
WINO_G = ((1.0, 0.0, 0.0), (0.5, 0.5, 0.5), (0.5, -0.5, 0.5), (0.0, 0.0, 1.0))
WINO_AT = ((1.0, 1.0, 1.0, 0.0), (0.0, 1.0, -1.0, -1.0))


def _wino_mats(like):
    k = dict(dtype=like.dtype, device=like.device)
    return torch.tensor(WINO_BT, **k), torch.tensor(WINO_G, **k), torch.tensor(WINO_AT, **k)


def _wino_tiles(x):
    """x (B,C,H,W) -> the 4 x 4 input windows (B,C,th,tw,4,4) of the 2 x 2 output tiles (zero outside the image), th = ceil(H/2)."""
    B, C, H, W = x.shape
    th, tw = (H + 1) // 2, (W + 1) // 2
    xp = F.pad(x, (1, 2 * tw - W + 1, 1, 2 * th - H + 1))
    return xp.unfold(2, 4, 2).unfold(3, 4, 2)


def winograd_conv3x3(x, w):
    """y = conv2d(x, w, padding=1) as  A^T [ (G g G^T) (.) (B^T d B) ] A  per 2 x 2 output tile (csrc/conv_wino_fused.hip:
    wino_fused_kernel forms the same four stages inside one launch; csrc/conv_wino.hip: wino_pack_kernel, wino_input_kernel, the 16
    products, wino_output_kernel).  x (B,Cin,H,W), w (Cout,Cin,3,3)."""
    Bt, G, At = _wino_mats(x)
    B, _, H, W = x.shape
    U = torch.einsum("ri,ocij,sj->ocrs", G, w, G)                     # (Cout,Cin,4,4)
    V = torch.einsum("ri,bcthij,sj->bcthrs", Bt, _wino_tiles(x), Bt)  # (B,Cin,th,tw,4,4)
    M = torch.einsum("bcthrs,ocrs->bothrs", V, U)                     # 16 products over the input channels
    Y = torch.einsum("ar,bothrs,es->bothae", At, M, At)               # (B,Cout,th,tw,2,2)
    th, tw = Y.shape[2], Y.shape[3]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w.shape[0], 2 * th, 2 * tw)[:, :, :H, :W]


def winograd_conv3x3_wgrad(x, dy):
    """dL/dw of y = conv2d(x, w, padding=1) given dL/dy:  G^T [ sum over tiles (A dY A^T) (.) (B^T d B) ] G  -- the adjoint of
    winograd_conv3x3 in w (csrc/conv_wino_fused.hip: wino_fused_wgrad_kernel; csrc/conv_wino.hip: wino_dy_kernel, the K-sliced products,
    wino_wgrad_finish_kernel)."""
    Bt, G, At = _wino_mats(x)
    B, Co, H, W = dy.shape
    th, tw = (H + 1) // 2, (W + 1) // 2
    dyp = F.pad(dy, (0, 2 * tw - W, 0, 2 * th - H)).reshape(B, Co, th, 2, tw, 2).permute(0, 1, 2, 4, 3, 5)   # (B,Co,th,tw,2,2)
    D = torch.einsum("ar,bothae,es->bothrs", At, dyp, At)             # A dY A^T with A = At^T
    V = torch.einsum("ri,bcthij,sj->bcthrs", Bt, _wino_tiles(x), Bt)
    dU = torch.einsum("bothrs,bcthrs->ocrs", D, V)
    return torch.einsum("ri,ocrs,sj->ocij", G, dU, G)                 # G^T dU G
