#!/usr/bin/env python3
"""Stage-by-stage comparison of the bf16 mode's eval forward with the rounding-point emulation of the oracle
(oracle.tag_oracle.cnn8rnn_forward_bf16_mode): where does the device first deviate by more than rounding ties?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import tag_oracle as O
from texttoaudiogrounding_amd import ops
from tests.test_gpu_path import build_hip_model

ops.CONV_MATH, ops.ACT_DTYPE = "bf16", "bf16"
dev = torch.device("cuda:0")
st = O.init_state(seed=13, logit_gain=30.0)
batch = O.synthetic_batch(4, 64000, seed=21, ragged=True)
model = build_hip_model(st, "dot", dev).eval()
mod = model.audio_encoder
s64 = O.state_to(st, torch.float64)
q = O._q_bf16
P = "audio_encoder."


def cmp(name, dev_t, ref_nchw):
    d = dev_t.double().cpu()
    r = ref_nchw.permute(0, 2, 3, 1) if ref_nchw.dim() == 4 else ref_nchw
    e = (d - r).abs()
    print(f"{name:22s} max {e.max().item():.3e} mean {e.mean().item():.3e}  |ref| max {r.abs().max().item():.3f}  "
          f"elements off by > 1 bf16 ulp: {(e > r.abs() * 2 ** -7 + 1e-6).double().mean().item():.2e}")


with torch.no_grad():
    wave = batch["waveform"].to(dev)
    lm = ops.logmel(wave, mod.n_fft, mod.win_length, mod.hop_length, mod.window, mod.mel_fb)
    x_ref = O.logmel(batch["waveform"].double(), "cnn8rnn").transpose(1, 2).unsqueeze(1)       # (B,1,F,64)
    cmp("logmel", lm, x_ref.squeeze(1))
    s0, t0 = O._bn_affine(x_ref.transpose(1, 3), s64, P + "bn0.", False)
    x_ref = (x_ref.transpose(1, 3) * s0 + t0).transpose(1, 3)
    bn = lambda y, bnm: ops.bn_stats(y.view(-1, y.shape[-1]), bnm.weight.detach(), bnm.bias.detach(), bnm.running_mean, bnm.running_var,
                                     False, bnm.eps, bnm.momentum)
    st0 = ops.bn_stats(lm.view(-1, 64), mod.bn0.weight.detach(), mod.bn0.bias.detach(), mod.bn0.running_mean, mod.bn0.running_var, False,
                       mod.bn0.eps, mod.bn0.momentum)
    x = None
    pools = [(2, 2), (2, 2), (1, 2), (1, 2)]
    for i, ps in enumerate(pools, start=1):
        blk = getattr(mod, f"conv_block{i}")
        bp = f"{P}conv_block{i}."
        w1, w2 = s64[bp + "conv1.weight"], s64[bp + "conv2.weight"]
        if i == 1:
            y1, _ = ops.conv3x3_c1_stats(lm, blk.conv1.weight.detach(), st0.scale, st0.shift, want_stats=False, out_dtype=torch.bfloat16)
        else:
            wf, _ = ops.pack_conv_weight(blk.conv1.weight.detach(), want_dgrad=False, W=x.shape[2])
            y1, _ = ops.conv3x3_stats(x, wf, blk.conv1.weight.shape[0], want_stats=False)
        y1f = F.conv2d(x_ref, w1 if i == 1 else q(w1), None, 1, 1)
        cmp(f"block{i}.conv1 (stored)", y1, q(y1f))
        s1 = bn(y1, blk.bn1)
        r1s, r1t = O._bn_affine(y1f, s64, bp + "bn1.", False)
        a1 = q(F.relu(q(y1f) * r1s + r1t))
        wf2, _ = ops.pack_conv_weight(blk.conv2.weight.detach(), want_dgrad=False, W=y1.shape[2])
        y2, _ = ops.conv3x3_stats(y1, wf2, y1.shape[3], prologue=1, scale=s1.scale, shift=s1.shift, want_stats=False)
        y2f = F.conv2d(a1, q(w2), None, 1, 1)
        cmp(f"block{i}.conv2 (stored)", y2, q(y2f))
        # the same conv2 fed the DEVICE's own y1 (isolates this stage from upstream tie flips)
        y1d = y1.double().cpu().permute(0, 3, 1, 2)
        y2f_own = F.conv2d(q(F.relu(y1d * r1s + r1t)), q(w2), None, 1, 1)
        cmp(f"  '' fed device y1", y2, q(y2f_own))
        s2 = bn(y2, blk.bn2)
        r2s, r2t = O._bn_affine(y2f, s64, bp + "bn2.", False)
        a2 = F.relu(q(y2f) * r2s + r2t)
        xo = ops.bnact_pool(y2, s2, ps[0], ps[1], act=1, pool=0)
        x_ref = q(F.avg_pool2d(a2, kernel_size=ps) + F.max_pool2d(a2, kernel_size=ps))
        cmp(f"block{i}.pool (stored)", xo, x_ref)
        y2d = y2.double().cpu().permute(0, 3, 1, 2)
        a2o = F.relu(y2d * r2s + r2t)
        cmp(f"  '' fed device y2", xo, q(F.avg_pool2d(a2o, kernel_size=ps) + F.max_pool2d(a2o, kernel_size=ps)))
        x = xo
