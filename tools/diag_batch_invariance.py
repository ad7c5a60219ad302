#!/usr/bin/env python3
"""Which inference kernel depends on the batch composition?  Every stage of the eval-mode Cnn8Rnn forward is run on a full
pass (B clips) and on a sub-batch (clips picked from it), the sub-batch stage being FED THE SLICE of the full pass's input --
so a difference is charged to the stage that produced it -- and compared bit for bit.

    python tools/diag_batch_invariance.py [B] [seconds]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from texttoaudiogrounding_amd import ops  # noqa: E402
from texttoaudiogrounding_amd.models.audio_encoder import Cnn8Rnn  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
SEC = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
dev = torch.device("cuda:0")
torch.manual_seed(0)
mod = Cnn8Rnn(32000).to(dev).eval()
for m in mod.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.running_mean.normal_(0, 0.1)
        m.running_var.uniform_(0.5, 1.5)
wave = 0.1 * torch.randn(B, int(SEC * 32000), device=dev)
pick = [0, 1, B // 2, B - 1]


def report(name, full, sub):
    same = torch.equal(full[pick], sub)
    d = (full[pick].float() - sub.float()).abs().max().item()
    print(f"{name:28s} {'bit-identical' if same else 'DIFFERS'}  max|d| = {d:.3e}  shape {tuple(full.shape)}")


def bn(y, bnm):
    return ops.bn_stats(y.view(-1, y.shape[-1]), bnm.weight.detach(), bnm.bias.detach(), bnm.running_mean, bnm.running_var, False,
                        bnm.eps, bnm.momentum)


with torch.no_grad():
    lm = ops.logmel(wave, mod.n_fft, mod.win_length, mod.hop_length, mod.window, mod.mel_fb)
    report("logmel", lm, ops.logmel(wave[pick].contiguous(), mod.n_fft, mod.win_length, mod.hop_length, mod.window, mod.mel_fb))
    st0 = ops.bn_stats(lm.view(-1, 64), mod.bn0.weight.detach(), mod.bn0.bias.detach(), mod.bn0.running_mean, mod.bn0.running_var,
                       False, mod.bn0.eps, mod.bn0.momentum)
    x = None
    for i in range(4):
        blk = getattr(mod, f"conv_block{i + 1}")
        if i == 0:
            y1, _ = ops.conv3x3_c1_stats(lm, blk.conv1.weight.detach(), st0.scale, st0.shift, want_stats=False)
            y1s, _ = ops.conv3x3_c1_stats(lm[pick].contiguous(), blk.conv1.weight.detach(), st0.scale, st0.shift, want_stats=False)
        else:
            wf, _ = ops.pack_conv_weight(blk.conv1.weight.detach(), want_dgrad=False, W=x.shape[2])
            y1, _ = ops.conv3x3_stats(x, wf, blk.conv1.weight.shape[0], want_stats=False)
            y1s, _ = ops.conv3x3_stats(x[pick].contiguous(), wf, blk.conv1.weight.shape[0], want_stats=False)
        report(f"block{i + 1}.conv1", y1, y1s)
        s1 = bn(y1, blk.bn1)
        wf2, _ = ops.pack_conv_weight(blk.conv2.weight.detach(), want_dgrad=False, W=y1.shape[2])
        C = y1.shape[3]
        y2, _ = ops.conv3x3_stats(y1, wf2, C, prologue=1, scale=s1.scale, shift=s1.shift, want_stats=False)
        y2s, _ = ops.conv3x3_stats(y1[pick].contiguous(), wf2, C, prologue=1, scale=s1.scale, shift=s1.shift, want_stats=False)
        report(f"block{i + 1}.conv2", y2, y2s)
        s2 = bn(y2, blk.bn2)
        ph, pw = ops.CNN8_POOLS[i]
        xo = ops.bnact_pool(y2, s2, ph, pw, act=1, pool=0)
        report(f"block{i + 1}.pool", xo, ops.bnact_pool(y2[pick].contiguous(), s2, ph, pw, act=1, pool=0))
        del y1, y2, y1s, y2s
        x = xo
    Bx, Tp, Wp, C = x.shape

    def mean_w(t):
        out = torch.empty(t.shape[0] * Tp, C, device=dev)
        ops.call("tag_mean_w_forward", ops.ptr(t), t.shape[0] * Tp, Wp, C, 0.0, 0, ops.ptr(out))
        return out.view(t.shape[0], Tp, C)
    xm = mean_w(x)
    report("mean over mel", xm, mean_w(x[pick].contiguous()))
    fw, fb = mod.fc1.weight.detach(), mod.fc1.bias.detach()

    def fc1(t):
        M = t.shape[0] * Tp
        return ops.gemm(t.reshape(M, C), fw, M, fw.shape[0], C, transB=True, bias=fb, act=1).view(t.shape[0], Tp, -1)
    fc = fc1(xm)
    report("fc1 GEMM", fc, fc1(xm[pick].contiguous()))
    rnn = [p.detach() for p in (mod.rnn.weight_ih_l0, mod.rnn.weight_hh_l0, mod.rnn.bias_ih_l0, mod.rnn.bias_hh_l0,
                                mod.rnn.weight_ih_l0_reverse, mod.rnn.weight_hh_l0_reverse, mod.rnn.bias_ih_l0_reverse,
                                mod.rnn.bias_hh_l0_reverse)]
    Hh = rnn[1].shape[1]
    w_ih, b_ih = torch.cat([rnn[0], rnn[4]], 0), torch.cat([rnn[2], rnn[6]], 0)

    def gi_of(t):
        M = t.shape[0] * Tp
        return ops.gemm(t.reshape(M, -1), w_ih, M, 6 * Hh, t.shape[-1], transB=True, bias=b_ih).view(t.shape[0], Tp, -1)
    gi = gi_of(fc)
    report("GRU input GEMM", gi, gi_of(fc[pick].contiguous()))
    y, _ = ops.gru_bidir_forward(fc.reshape(Bx * Tp, -1), rnn, Bx, Tp, False)
    ys, _ = ops.gru_bidir_forward(fc[pick].reshape(len(pick) * Tp, -1), rnn, len(pick), Tp, False)
    report("BiGRU (GEMM + recurrence)", y, ys)
    ops.check_async_errors()
