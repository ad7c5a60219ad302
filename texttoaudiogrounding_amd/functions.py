"""torch.autograd nodes of the hot path: forward() / backward() of each node are sequences of dispatch.py launches; parameter
gradients are delivered through engine.py's direct-gradient sinks.  Cnn8RnnFunction (the whole audio encoder as one node) and its
weight-gradient side-stream helper, CrnnFunction, the text / match / loss heads, the cross-encoder nodes.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import settings as cfg
from .lib import call, ptr, query
from .engine import (F32, TagFunction, _chk, _deliver, _empty, _flush, _ready, _side_stream, _sinks, _ws, new_seed,
                     side_stream_enabled)
from .dispatch import (BF16, BNStat, C1_BWD_FUSED_SHAPE, _sfx, _wino_u, act_bf16, bn_act_backward, bn_param_grad,
                       bn_stats, bnact_pool, bnrelu_pool_backward, check_pass_size, colsum, conv3x3,
                       conv3x3_bnrelu_pool_eval, conv3x3_c1, conv3x3_c1_backward, conv3x3_c1_dgrad, conv3x3_c1_stats,
                       conv3x3_c1_wgrad, conv3x3_dgrad_bnrelu_backward, conv3x3_dgrad_poolsums, conv3x3_stats,
                       conv3x3_wgrad, embed_mean_backward_into, embed_mean_forward, eval_pool_fusable, gemm,
                       gru_bidir_backward, gru_bidir_forward, logmel, lppool_leaky_backward, pack_conv_weight,
                       pool_sums_fusable, relu_backward)

# ------------------------------------------------------------------------------------------------
# Cnn8Rnn: the whole audio encoder as one autograd node (rows F1-F3, A1-A4 forward + backward)
# ------------------------------------------------------------------------------------------------

CNN8_POOLS = [(2, 2), (2, 2), (1, 2), (1, 2)]


class _SideWgrad:
    """Runs conv3x3_wgrad calls on the side stream; join() makes the main stream wait for all of them.

    Lifetime of the tensors the side stream reads or writes: they are allocated on the MAIN stream, and the caching allocator
    would hand their memory to the main stream's next allocation the moment Python drops them.  They are therefore kept alive in
    ``self.keep`` until join() has made the main stream wait for the side stream -- from then on a release on the main stream is
    ordered after every side-stream access.  (Round 4 used ``Tensor.record_stream`` instead.  That defers the reuse of a block
    until the HOST sees the side stream's event complete; a host that enqueues K unsynchronised steps runs far ahead of the
    GPU, sees none complete and takes NEW memory for every step (measured: 53-75 GB of reserve for 3-6 GB of tensors after 30-60
    steps -- 100-150 GB on a fast box --, 22-26 ms of host time per step inside hipMalloc), and in every second of a row of
    `bench.py --conv-math x3 --steps 30` processes ONE such hipMalloc blocked for 0.7-2.6 s (the driver still reclaiming the
    previous process's reserve): kernels at their normal durations, the main thread asleep (docs/experiments_r05.md).)"""

    def __init__(self, device):
        self.on = side_stream_enabled()
        self.main = torch.cuda.current_stream(device)
        self.side = _side_stream(device) if self.on else None
        self.pending = []
        self.keep = []

    def wgrad(self, x, dy, prologue=0, scale=None, shift=None, out=None):
        if not self.on:
            return conv3x3_wgrad(x, dy, prologue, scale, shift, out=out)
        dw = out if out is not None else _empty(dy.shape[3], x.shape[3], 3, 3, like=x)
        self.pending.append((x, dy, prologue, scale, shift, dw))
        if not cfg.WGRAD_LAG:
            self.release()
        return dw

    def run(self, fn, tensors=()):
        """``fn()`` on the side stream, ordered after everything enqueued on the main stream so far (inline when the side
        stream is off).  ``tensors``: main-stream allocations fn reads (kept alive until join())."""
        if not self.on:
            fn()
            return
        self.release()
        self.side.wait_stream(self.main)
        with torch.cuda.stream(self.side):
            fn()
        self._hold(tensors)

    def release(self):
        """Launch the queued wgrads on the side stream, ordered after everything enqueued on the main stream so far."""
        if not self.pending:
            return
        self.side.wait_stream(self.main)                 # x, dy (and the BN constants) are ready
        with torch.cuda.stream(self.side):
            for x, dy, prologue, scale, shift, dw in self.pending:
                conv3x3_wgrad(x, dy, prologue, scale, shift, out=dw)
        for item in self.pending:
            self._hold(item)
        self.pending = []

    def _hold(self, tensors):
        for t in tensors:
            if isinstance(t, torch.Tensor):
                if cfg.SIDE_RECORD_STREAM:
                    t.record_stream(self.side)
                else:
                    self.keep.append(t)

    def join(self):
        if self.on:
            self.release()
            self.main.wait_stream(self.side)
            self.keep = []                               # released on the main stream, ordered after the wait


class Cnn8RnnFunction(TagFunction):
    """params order: bn0.w, bn0.b, 4 x (conv1.w, bn1.w, bn1.b, conv2.w, bn2.w, bn2.b), fc1.w, fc1.b,
    rnn (w_ih, w_hh, b_ih, b_hh) x (fwd, reverse)."""

    @staticmethod
    def forward(ctx, waveform, mod, *params):
        wave = _chk(waveform, "waveform")
        check_pass_size(wave.shape[0], wave.shape[1] // mod.hop_length + 1)
        training = mod.training
        bn_train = training and not mod.freeze_bn
        p = [_chk(t.detach(), "parameter") for t in params]
        bn0_w, bn0_b = p[0], p[1]
        blocks = [p[2 + 6 * i: 8 + 6 * i] for i in range(4)]
        fc_w, fc_b = p[26], p[27]
        rnn = p[28:36]
        drop = mod.dropout_p if training else (0.0, 0.0)
        seeds = [new_seed() for _ in range(5)] if training and (drop[0] > 0 or drop[1] > 0) else [0] * 5
        need_grad = any(ctx.needs_input_grad[2:])

        lm = logmel(wave, mod.n_fft, mod.win_length, mod.hop_length, mod.window, mod.mel_fb)   # (B,F,64)
        B, Fr, NM = lm.shape
        st0 = bn_stats(lm.view(B * Fr, NM), bn0_w, bn0_b, mod.bn0.running_mean, mod.bn0.running_var, bn_train,
                       mod.bn0.eps, mod.bn0.momentum)
        x = None
        acts = []
        for i, (c1w, g1, b1, c2w, g2, b2) in enumerate(blocks):
            blk = getattr(mod, f"conv_block{i + 1}")
            if i == 0:
                y1, part1 = conv3x3_c1_stats(lm, c1w, st0.scale, st0.shift, want_stats=bn_train,
                                             out_dtype=BF16 if act_bf16() else F32)
                wf1 = wd1 = None
            else:
                wf1, wd1 = pack_conv_weight(c1w, want_dgrad=need_grad, W=x.shape[2])
                y1, part1 = conv3x3_stats(x, wf1, c1w.shape[0], want_stats=bn_train,
                                          inference=not need_grad and not bn_train and drop[0] == 0.0)
            Bx, H, W, C = y1.shape
            s1 = bn_stats(y1.view(-1, C), g1, b1, blk.bn1.running_mean, blk.bn1.running_var, bn_train, blk.bn1.eps,
                          blk.bn1.momentum, partials=part1)
            wf2, wd2 = pack_conv_weight(c2w, want_dgrad=need_grad, W=y1.shape[2])
            ph, pw = CNN8_POOLS[i]
            if not need_grad and not bn_train and drop[0] == 0.0 and eval_pool_fusable(y1, wf2, ph, pw):
                # inference (models/hf_modeling_grounding.py; evaluation between epochs): bn2's affine is known before the conv
                # runs, so conv2 pools its own output tile -- y2, the block's largest tensor, is never written or read back
                s2 = bn_stats(g2.view(1, C), g2, b2, blk.bn2.running_mean, blk.bn2.running_var, False, blk.bn2.eps, blk.bn2.momentum)
                x = conv3x3_bnrelu_pool_eval(y1, wf2, C, s2, ph, pw, prologue=1, scale=s1.scale, shift=s1.shift)
                continue
            y2, part2 = conv3x3_stats(y1, wf2, C, prologue=1, scale=s1.scale, shift=s1.shift, want_stats=bn_train)
            s2 = bn_stats(y2.view(-1, C), g2, b2, blk.bn2.running_mean, blk.bn2.running_var, bn_train, blk.bn2.eps,
                          blk.bn2.momentum, partials=part2)
            xo = bnact_pool(y2, s2, ph, pw, act=1, pool=0, drop_p=drop[0], seed=seeds[i])
            if need_grad:                      # inference: intermediates die here (30 s clips x 64 are GBs per layer)
                acts.append((x, y1, s1, y2, s2, wd1, wd2))
            x = xo
        Bx, Tp, Wp, C = x.shape
        xm = _empty(Bx * Tp, C, like=x)
        call("tag_mean_w_forward" + _sfx(x), ptr(x), Bx * Tp, Wp, C, float(drop[1]), seeds[4], ptr(xm))
        M = Bx * Tp
        fc = gemm(xm, fc_w, M, fc_w.shape[0], C, transB=True, bias=fc_b, act=1)
        y, gsave = gru_bidir_forward(fc, rnn, Bx, Tp, need_grad)
        if need_grad:
            ctx.saved = dict(lm=lm, st0=st0, acts=acts, x_last=x, xm=xm, fc=fc, gsave=gsave, p=p, drop=drop,
                             seeds=seeds, sinks=_sinks(params), params=params if cfg.DIRECT_GRADS else None)
        mod._last_dropout = dict(p=drop, seeds=seeds)
        return y

    @staticmethod
    def backward(ctx, dy):
        sv = ctx.saved
        ctx.saved = None
        p = sv["p"]
        drop, seeds = sv["drop"], sv["seeds"]
        dy = _chk(dy, "grad_output")
        grads: List[Optional[torch.Tensor]] = [None] * len(p)
        sk, prm = sv["sinks"], sv["params"]
        fc = sv["fc"]
        sw = _SideWgrad(dy.device)
        dfc, ggru = gru_bidir_backward(dy, fc, sv["gsave"], outs=sk[28:36], side=sw if cfg.SIDE_PARAM_GRADS >= 1 else None)
        for k in range(8):
            _deliver(grads, sk, 28 + k, ggru[k])
        M = fc.shape[0]
        dfc = relu_backward(fc, dfc)
        xm = sv["xm"]
        fc_w = p[26]
        if cfg.SIDE_PARAM_GRADS >= 2 and sk[26] is not None and sk[27] is not None:
            # fc1's parameter gradients are off the dx chain too: beside the passes below, on the side stream
            sw.run(lambda: (gemm(dfc, xm, fc_w.shape[0], fc_w.shape[1], M, transA=True, lda=fc_w.shape[0], out=sk[26]),
                            colsum(dfc, M, fc_w.shape[0], out=sk[27])), (dfc, xm))
            _deliver(grads, sk, 26, sk[26])
            _deliver(grads, sk, 27, sk[27])
        else:
            _deliver(grads, sk, 26, gemm(dfc, xm, fc_w.shape[0], fc_w.shape[1], M, transA=True, lda=fc_w.shape[0], out=sk[26]))
            _deliver(grads, sk, 27, colsum(dfc, M, fc_w.shape[0], out=sk[27]))
        if prm is not None:
            # the persistent GRU backward is enqueued: from here on a bucket's all-reduce may run beside the kernels of
            # this stream (never beside the spinning GRU workgroups: the collective is ordered after them)
            _ready(prm[26:36])
            _flush()
        if not any(ctx.needs_input_grad[2:28]):
            # Cnn8Rnn(freeze_cnn=True) (models/audio_encoder.py:164-168: everything but the GRU frozen): no parameter below the
            # GRU takes a gradient and the waveform never does -- the conv stack's backward (97 % of the step) is not run
            sw.join()
            return (None, None, *grads)
        dxm = gemm(dfc, fc_w, M, fc_w.shape[1], fc_w.shape[0])
        x_last = sv["x_last"]
        Bx, Tp, Wp, C = x_last.shape
        dx = torch.empty_like(x_last)
        call("tag_mean_w_backward" + _sfx(dx), ptr(dxm), Bx * Tp, Wp, C, float(drop[1]), seeds[4], ptr(dx))
        # ---- conv blocks, last to first ----
        lm, st0 = sv["lm"], sv["st0"]
        poolpart = None                            # sums of block i's pool backward, taken by block i+1's dgrad conv
        for i in range(3, -1, -1):
            x_in, y1, s1, y2, s2, wd1, wd2 = sv["acts"][i]
            c1w, g1, b1, c2w, g2, b2 = p[2 + 6 * i: 8 + 6 * i]
            o = 2 + 6 * i
            ph, pw = CNN8_POOLS[i]
            C = y2.shape[3]
            dy2, dg2, db2 = bnrelu_pool_backward(y2, s2, g2, dx, ph, pw, drop[0], seeds[i], dg_out=sk[o + 4], db_out=sk[o + 5],
                                                 partials=poolpart)
            poolpart = None
            _deliver(grads, sk, o + 4, dg2)
            _deliver(grads, sk, o + 5, db2)
            del dx
            _deliver(grads, sk, o + 3, sw.wgrad(y1, dy2, prologue=1, scale=s1.scale, shift=s1.shift, out=sk[o + 3]))
            # block 1: its first conv has ONE consumer of dy1 (the Cin = 1 backward), which applies bn1's backward itself
            defer = i == 0 and cfg.FUSE_C1_BN_BWD and (y1.shape[2], y1.shape[3]) == C1_BWD_FUSED_SHAPE
            res = conv3x3_dgrad_bnrelu_backward(dy2, wd2, y1, s1, g1, dg_out=sk[o + 1], db_out=sk[o + 2],
                                                after_conv=sw.release, defer_apply=defer)
            dy1, dg1, db1 = res[:3]
            applied = res[3] if defer else True
            del dy2
            _deliver(grads, sk, o + 1, dg1)
            _deliver(grads, sk, o + 2, db1)
            if i > 0:
                _deliver(grads, sk, o, sw.wgrad(x_in, dy1, out=sk[o]))
                below = sv["acts"][i - 1]          # (x, y1, s1, y2, s2, ...) of the block whose pooled output x_in is
                if pool_sums_fusable(dy1, wd1, below[3], *CNN8_POOLS[i - 1]):
                    # (direct halo-tile kernel or, for the deep layers, the Winograd form: both carry the sums in their epilogue)
                    dx, poolpart = conv3x3_dgrad_poolsums(dy1, wd1, below[3], below[4], *CNN8_POOLS[i - 1], drop[0], seeds[i - 1])
                elif _wino_u(wd1, dy1, x_in.shape[3], count=False) is not None:
                    dx = conv3x3(dy1, wd1, x_in.shape[3], training_launch=True)
                else:
                    dx = conv3x3(dy1, wd1, x_in.shape[3])
                sw.release()
            else:
                dw0, dbn0 = conv3x3_c1_backward(lm, dy1, c1w, st0.scale, st0.shift, out=sk[2],   # dbn0: (B,F,64) grad wrt bn0 output
                                                bn_bwd=None if applied else (y1, s1, g1, dg1, db1))
                _deliver(grads, sk, 2, dw0)
                Bq, Fr, NM = lm.shape
                dg0, db0 = bn_param_grad(lm.view(Bq * Fr, NM), dbn0.view(Bq * Fr, NM), st0, dg_out=sk[0], db_out=sk[1])
                _deliver(grads, sk, 0, dg0)
                _deliver(grads, sk, 1, db0)
            del dy1
            sv["acts"][i] = None
            sw.release()                           # every gradient kernel of this block is enqueued before _ready
            if prm is not None:
                _ready(prm[o:o + 6] + ((prm[0], prm[1]) if i == 0 else ()))
                _flush()
        sw.join()
        return (None, None, *grads)


# ------------------------------------------------------------------------------------------------
# CrnnEncoder (row A1'): cdur_block = BN -> conv3x3 -> LeakyReLU(0.1), LPPool2d(4), Dropout(0.3), BiGRU(128)
# ------------------------------------------------------------------------------------------------
CRNN_POOLS = [(2, 4), (2, 4), (1, 4)]


class CrnnFunction(TagFunction):
    """params order: 5 x (bn.w, bn.b, conv.w) for cnn.{0,2,3,5,6}, then gru (w_ih, w_hh, b_ih, b_hh) x (fwd, reverse).

    Layer plan (channels-last): lm -> [bn0 scalar | conv 1->32] -> LP(2,4) -> [bn | conv 32->128] -> [leaky,bn | conv]
    -> LP(2,4) -> [bn | conv] -> [leaky,bn | conv] -> LP(1,4)+dropout -> GRU.  Every BatchNorm is folded into the
    A-operand load of the conv that follows it (prologue 3 after a pool, prologue 2 after a conv)."""

    @staticmethod
    def forward(ctx, waveform, mod, *params):
        wave = _chk(waveform, "waveform")
        training = mod.training
        p = [_chk(t.detach(), "parameter") for t in params]
        blk = [p[3 * i: 3 * i + 3] for i in range(5)]
        rnn = p[15:23]
        bns = mod._bn_modules()
        drop = mod.dropout_p if training else 0.0
        seed = new_seed() if training and drop > 0 else 0

        lm = logmel(wave, mod.n_fft, mod.win_length, mod.hop_length, mod.window, mod.mel_fb)   # (B,F,64)
        B, Fr, NM = lm.shape

        def stats(x2d, i, pre_op):
            return bn_stats(x2d, blk[i][0], blk[i][1], bns[i].running_mean, bns[i].running_var, training, bns[i].eps,
                            bns[i].momentum, pre_op)

        st = [None] * 5
        st[0] = stats(lm.view(-1, 1), 0, 0)                       # BatchNorm2d(1): one scalar affine
        cs, ct = st[0].scale.expand(NM).contiguous(), st[0].shift.expand(NM).contiguous()
        y0 = conv3x3_c1(lm, blk[0][2], cs, ct)                     # (B,F,64,32)
        p1 = bnact_pool(y0, None, 2, 4, act=2, pool=1)            # leaky + LPPool -> (B,F/2,16,32)
        st[1] = stats(p1.view(-1, p1.shape[3]), 1, 0)
        wf1, wd1 = pack_conv_weight(blk[1][2], W=p1.shape[2])
        y1 = conv3x3(p1, wf1, 128, prologue=3, scale=st[1].scale, shift=st[1].shift)
        st[2] = stats(y1.view(-1, 128), 2, 1)
        wf2, wd2 = pack_conv_weight(blk[2][2], W=y1.shape[2])
        y2 = conv3x3(y1, wf2, 128, prologue=2, scale=st[2].scale, shift=st[2].shift)
        p2 = bnact_pool(y2, None, 2, 4, act=2, pool=1)            # (B,F/4,4,128)
        st[3] = stats(p2.view(-1, 128), 3, 0)
        wf3, wd3 = pack_conv_weight(blk[3][2], W=p2.shape[2])
        y3 = conv3x3(p2, wf3, 128, prologue=3, scale=st[3].scale, shift=st[3].shift)
        st[4] = stats(y3.view(-1, 128), 4, 1)
        wf4, wd4 = pack_conv_weight(blk[4][2], W=y3.shape[2])
        y4 = conv3x3(y3, wf4, 128, prologue=2, scale=st[4].scale, shift=st[4].shift)
        p3 = bnact_pool(y4, None, 1, 4, act=2, pool=1, drop_p=drop, seed=seed)     # (B,T',1,128)
        Bx, Tp = p3.shape[0], p3.shape[1]
        x2d = p3.view(Bx * Tp, -1)
        need_grad = any(ctx.needs_input_grad[2:])
        y, gsave = gru_bidir_forward(x2d, rnn, Bx, Tp, need_grad)
        if need_grad:
            ctx.saved = dict(lm=lm, cs=cs, ct=ct, st=st, y=[y0, y1, y2, y3, y4], pool=[p1, p2, p3], wd=[wd1, wd2, wd3, wd4],
                             x2d=x2d, gsave=gsave, p=p, drop=drop, seed=seed, sinks=_sinks(params),
                             params=params if cfg.DIRECT_GRADS else None)
        mod._last_dropout = dict(p=drop, seeds=[seed])
        return y

    @staticmethod
    def backward(ctx, dy):
        sv = ctx.saved
        ctx.saved = None
        p, st, ys, pools, wd = sv["p"], sv["st"], sv["y"], sv["pool"], sv["wd"]
        blk = [p[3 * i: 3 * i + 3] for i in range(5)]
        grads: List[Optional[torch.Tensor]] = [None] * len(p)
        dy = _chk(dy, "grad_output")
        # every gradient is written straight into its flat-gradient view when the parameter has one (sk[k]; None = returned to
        # autograd): the 23 per-parameter copies of the former form were 0.11 ms of a 5.3 ms step (tools/step_timeline.py)
        sk = sv["sinks"]
        dx2d, grads[15:23] = gru_bidir_backward(dy, sv["x2d"], sv["gsave"], outs=sk[15:23])
        y0, y1, y2, y3, y4 = ys
        p1, p2, p3 = pools
        # block 6 (cnn.6): conv(bn(leaky(y3)))
        dy4 = lppool_leaky_backward(y4, dx2d.view(p3.shape), 1, 4, sv["drop"], sv["seed"])
        grads[14] = conv3x3_wgrad(y3, dy4, prologue=2, scale=st[4].scale, shift=st[4].shift, out=sk[14])
        du = conv3x3(dy4, wd[3], 128)
        dy3, grads[12], grads[13] = bn_act_backward(y3, 1, st[4], blk[4][0], du, dg_out=sk[12], db_out=sk[13])
        # block 5 (cnn.5): conv(bn(p2))
        grads[11] = conv3x3_wgrad(p2, dy3, prologue=3, scale=st[3].scale, shift=st[3].shift, out=sk[11])
        du = conv3x3(dy3, wd[2], 128)
        dp2, grads[9], grads[10] = bn_act_backward(p2, 0, st[3], blk[3][0], du, dg_out=sk[9], db_out=sk[10])
        dy2 = lppool_leaky_backward(y2, dp2, 2, 4)
        # block 3 (cnn.3)
        grads[8] = conv3x3_wgrad(y1, dy2, prologue=2, scale=st[2].scale, shift=st[2].shift, out=sk[8])
        du = conv3x3(dy2, wd[1], 128)
        dy1, grads[6], grads[7] = bn_act_backward(y1, 1, st[2], blk[2][0], du, dg_out=sk[6], db_out=sk[7])
        # block 2 (cnn.2)
        grads[5] = conv3x3_wgrad(p1, dy1, prologue=3, scale=st[1].scale, shift=st[1].shift, out=sk[5])
        du = conv3x3(dy1, wd[0], p1.shape[3])
        dp1, grads[3], grads[4] = bn_act_backward(p1, 0, st[1], blk[1][0], du, dg_out=sk[3], db_out=sk[4])
        dy0 = lppool_leaky_backward(y0, dp1, 2, 4)
        # block 0 (cnn.0): conv(bn_scalar(lm))
        lm = sv["lm"]
        grads[2] = conv3x3_c1_wgrad(lm, dy0, sv["cs"], sv["ct"], out=sk[2])
        du0 = conv3x3_c1_dgrad(dy0, blk[0][2])                                     # (B,F,64) grad wrt bn output
        B, Fr, NM = lm.shape
        st0c = BNStat()
        st0c.mean, st0c.invstd = st[0].mean.expand(NM).contiguous(), st[0].invstd.expand(NM).contiguous()
        dgc, dbc = bn_param_grad(lm.view(B * Fr, NM), du0.view(B * Fr, NM), st0c)
        grads[0], grads[1] = dgc.sum().view(1), dbc.sum().view(1)                   # 64 columns share one channel
        for k in range(len(grads)):
            if grads[k] is not None:
                _deliver(grads, sk, k, grads[k])
        _ready(sv["params"])
        return (None, None, *grads)


# ------------------------------------------------------------------------------------------------
# small heads
# ------------------------------------------------------------------------------------------------

class LinearFunction(TagFunction):
    """nn.Linear on the MFMA GEMM (audio_proj / text_proj, models/audio_text_model.py:45-46,78-87)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x2 = _chk(x, "x").view(-1, x.shape[-1])
        w_, b_ = _chk(w.detach(), "weight"), (_chk(b.detach(), "bias") if b is not None else None)
        M, K = x2.shape
        N = w_.shape[0]
        y = gemm(x2, w_, M, N, K, transB=True, bias=b_)
        ctx.save_for_backward(x2, w_)
        ctx.has_bias = b is not None
        ctx.xshape = x.shape
        ctx.sinks = _sinks([x, w, b])
        ctx.params = [w, b] if cfg.DIRECT_GRADS else None
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        M, K = x2.shape
        N = w.shape[0]
        dy2 = _chk(dy, "grad").view(M, N)
        dx = gemm(dy2, w, M, K, N).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        sk = ctx.sinks
        g = [dx, None, None]
        _deliver(g, sk, 1, gemm(dy2, x2, N, K, M, transA=True, lda=N, out=sk[1]))
        if ctx.has_bias:
            _deliver(g, sk, 2, colsum(dy2, M, N, out=sk[2]))
        _ready(ctx.params)
        return tuple(g)


# Each head's forward / backward arithmetic lives in ONE plain function below.  texttoaudiogrounding_amd/torch_ops.py
# registers them as PyTorch operators (torch.ops.tag.embed_mean / frame_match / align_dot / frame_bce + *_backward, with
# autograd formulas) and the reference-shaped modules (models/match.py, models/align.py, losses.py, models/text_encoder.py)
# call THOSE operators; the only autograd.Function kept here is EmbedMeanFunction, the direct-gradient variant that scatters
# straight into the flat-gradient rows of the table (an operator may not mutate hidden state).

class EmbedMeanFunction(TagFunction):
    """embed_mean with direct gradients (StrongRunner): the table gradient is scattered straight into the (zeroed)
    flat-gradient rows; same kernels as torch.ops.tag.embed_mean."""

    @staticmethod
    def forward(ctx, table, text, text_len, want_tokens):
        seq, tok = embed_mean_forward(table.detach(), text, text_len, want_tokens)
        ctx.save_for_backward(text, text_len)
        ctx.vd = tuple(table.shape)
        ctx.sinks = _sinks([table])
        ctx.table = table
        ctx.set_materialize_grads(False)
        return seq, tok

    @staticmethod
    def backward(ctx, dseq, dtok):
        text, text_len = ctx.saved_tensors
        sink = ctx.sinks[0]
        direct = sink is not None
        dtab = sink if direct else torch.zeros(*ctx.vd, device=text.device, dtype=F32)
        embed_mean_backward_into(dtab, dseq, dtok, text, text_len)
        if direct:
            _ready([ctx.table])
            return None, None, None, None
        return dtab, None, None, None


class Seq2SeqAttentionFunction(TagFunction):
    """Seq2SeqAttention.forward (models/cross_encoder.py:11-42): additive attention of every query row over the key/value
    rows, ``score[b,q,k] = v . tanh(W [query_q ; kv_k] + b)``, the two -1e10 mask fills, softmax over k, ``out = attn @ kv``.
    The reference materialises the (B, Lq*Lk, Dq+Dkv) concatenation; here ``W = [Wq | Wk]`` is applied as two MFMA GEMMs and
    cross.hip does the rest.  query (B,Lq,Dq), kv (B,Lk,Dkv) -> (B,Lq,Dkv).  params: h2attn.weight (Da, Dq+Dkv),
    h2attn.bias (Da), v (Da)."""

    @staticmethod
    def forward(ctx, query, kv, query_len, kv_len, w_h, b_h, v):
        a, t = _chk(query, "query"), _chk(kv, "kv")
        B, T, D = a.shape
        L, Dk = t.shape[1], t.shape[2]
        Da = w_h.shape[0]
        ctx.sinks = _sinks([w_h, b_h, v])
        ctx.params = [w_h, b_h, v] if cfg.DIRECT_GRADS else None
        w_h, b_h, v = (_chk(x.detach(), "parameter") for x in (w_h, b_h, v))
        if w_h.shape[1] != D + Dk or b_h.shape != (Da,) or v.shape != (Da,):
            raise RuntimeError("Seq2SeqAttention: inconsistent dimensions")
        dev = a.device
        ql = torch.as_tensor(query_len).long().to(dev).contiguous()
        kl = torch.as_tensor(kv_len).long().to(dev).contiguous()
        aq = gemm(a, w_h, B * T, Da, D, transB=True, ldb=D + Dk)
        ak = gemm(t, w_h[:, D:], B * L, Da, Dk, transB=True, ldb=D + Dk, bias=b_h)
        attn = _empty(B, T, L, like=a)
        cx = _empty(B, T, Dk, like=a)
        call("tag_addattn_forward", ptr(aq), ptr(ak), ptr(v), ptr(t), ptr(ql), ptr(kl), ptr(attn), ptr(cx), B, T, L, Da, Dk)
        ctx.save_for_backward(a, t, aq, ak, attn, ql, kl, w_h, v)
        return cx

    @staticmethod
    def backward(ctx, dcx):
        a, t, aq, ak, attn, ql, kl, w_h, v = ctx.saved_tensors
        B, T, D = a.shape
        L, Dk = t.shape[1], t.shape[2]
        Da, M = w_h.shape[0], B * T
        dcx = _chk(dcx, "grad")
        daq, dak = _empty(B, T, Da, like=a), _empty(B, L, Da, like=a)
        dkv, dv = _empty(B, L, Dk, like=a), _empty(Da, like=a)
        ws = _ws(query("tag_addattn_backward_ws_bytes", B, T, L, Da, Dk), a)
        call("tag_addattn_backward", ptr(aq), ptr(ak), ptr(v), ptr(t), ptr(attn), ptr(dcx), ptr(ql), ptr(kl), ptr(daq),
             ptr(dak), ptr(dkv), ptr(dv), B, T, L, Da, Dk, ptr(ws))
        dw_h = _empty(Da, D + Dk, like=a)
        gemm(daq, a, Da, D, M, transA=True, lda=Da, out=dw_h, ldc=D + Dk)
        gemm(dak, t, Da, Dk, B * L, transA=True, lda=Da, out=dw_h[:, D:], ldc=D + Dk)
        db_h = colsum(dak, B * L, Da)
        da = gemm(daq, w_h, M, D, Da, ldb=D + Dk).view(B, T, D)
        gemm(dak, w_h[:, D:], B * L, Dk, Da, ldb=D + Dk, out=dkv, accumulate=True)
        g = [dw_h, db_h, dv]
        for k in range(3):
            _deliver(g, ctx.sinks, k, g[k])
        _ready(ctx.params)
        return (da, dkv, None, None, *g)


class CrossGatingFunction(TagFunction):
    """CrossGating.forward (models/cross_encoder.py:45-57): ``s_out = s * sigmoid(fc_u(u))``, ``u_out = u * sigmoid(fc_s(s))``
    -- two MFMA GEMMs with the sigmoid epilogue + tag_mul / tag_gate_backward.  u, s (..., D) -> (u_out, s_out)."""

    @staticmethod
    def forward(ctx, u, s, w_u, b_u, w_s, b_s):
        a, cx = _chk(u, "u"), _chk(s, "s")
        D = a.shape[-1]
        if cx.shape != a.shape or w_u.shape != (D, D) or w_s.shape != (D, D):
            raise RuntimeError("CrossGating: inconsistent dimensions")
        ctx.sinks = _sinks([w_u, b_u, w_s, b_s])
        ctx.params = [w_u, b_u, w_s, b_s] if cfg.DIRECT_GRADS else None
        w_u, b_u, w_s, b_s = (_chk(x.detach(), "parameter") for x in (w_u, b_u, w_s, b_s))
        M = a.numel() // D
        g_u = gemm(a, w_u, M, D, D, transB=True, bias=b_u, act=5)
        g_s = gemm(cx, w_s, M, D, D, transB=True, bias=b_s, act=5)
        u_out, s_out = torch.empty_like(a), torch.empty_like(cx)
        call("tag_mul", ptr(a), ptr(g_s), ptr(u_out), a.numel())
        call("tag_mul", ptr(cx), ptr(g_u), ptr(s_out), cx.numel())
        ctx.save_for_backward(a, cx, g_u, g_s, w_u, w_s)
        return u_out, s_out

    @staticmethod
    def backward(ctx, du_out, ds_out):
        a, cx, g_u, g_s, w_u, w_s = ctx.saved_tensors
        D = a.shape[-1]
        M = a.numel() // D
        du_out, ds_out = _chk(du_out, "grad"), _chk(ds_out, "grad")
        da, dz_s = torch.empty_like(a), torch.empty_like(a)
        call("tag_gate_backward", ptr(du_out), ptr(a), ptr(g_s), ptr(da), 0, ptr(dz_s), a.numel())       # u_out = u * g_s
        dcx, dz_u = torch.empty_like(cx), torch.empty_like(cx)
        call("tag_gate_backward", ptr(ds_out), ptr(cx), ptr(g_u), ptr(dcx), 0, ptr(dz_u), cx.numel())    # s_out = s * g_u
        dw_s = gemm(dz_s, cx, D, D, M, transA=True, lda=D)
        db_s = colsum(dz_s, M, D)
        gemm(dz_s, w_s, M, D, D, out=dcx, accumulate=True)
        dw_u = gemm(dz_u, a, D, D, M, transA=True, lda=D)
        db_u = colsum(dz_u, M, D)
        gemm(dz_u, w_u, M, D, D, out=da, accumulate=True)
        g = [dw_u, db_u, dw_s, db_s]
        for k in range(4):
            _deliver(g, ctx.sinks, k, g[k])
        _ready(ctx.params)
        return (da, dcx, *g)


class CrossAttentionHeadFunction(TagFunction):
    """match.CrossAttention (models/match.py:63-88): nn.MultiheadAttention(E, H, p, batch_first, kdim = vdim = kvdim) of every
    audio frame over the phrase tokens, ``audio + dropout(out)``, LayerNorm, Linear(E,1), sigmoid -> (B,T).
    params = (wq (E,E), wk (E,Dk), wv (E,Dk), in_proj_bias (3E), out_proj.weight, out_proj.bias, norm.weight, norm.bias,
    linear.weight (1,E), linear.bias (1)); wq/wk/wv may be row blocks of one in_proj_weight (kvdim = E)."""

    @staticmethod
    def forward(ctx, audio, token, text_len, num_heads, drop_p, training, *params):
        a, t = _chk(audio, "audio_emb"), _chk(token, "token_emb")
        B, T, E = a.shape
        L, Dk = t.shape[1], t.shape[2]
        sinks = _sinks(params)
        wq, wk, wv, b_in, wo, bo, g, be, wl, bl = (_chk(x.detach(), "parameter") for x in params)
        if wq.shape != (E, E) or wk.shape != (E, Dk) or wv.shape != (E, Dk) or wo.shape != (E, E) or E % num_heads:
            raise RuntimeError("CrossAttention: inconsistent dimensions")
        kl = torch.as_tensor(text_len).long().to(a.device).contiguous()
        M, ML = B * T, B * L
        p = float(drop_p) if training else 0.0
        seeds = [new_seed(), new_seed()] if p > 0 else [0, 0]
        q = gemm(a, wq, M, E, E, transB=True, bias=b_in[:E])
        k = gemm(t, wk, ML, E, Dk, transB=True, bias=b_in[E:2 * E])
        v = gemm(t, wv, ML, E, Dk, transB=True, bias=b_in[2 * E:])
        attn = _empty(B, T, num_heads, L, like=a)
        cx = _empty(B, T, E, like=a)
        call("tag_mha_cross_forward", ptr(q), ptr(k), ptr(v), ptr(kl), ptr(attn), ptr(cx), B, T, L, E, num_heads, p, seeds[0])
        r = gemm(cx, wo, M, E, E, transB=True, bias=bo)
        sim = _empty(B, T, like=a)
        mu, rstd = _empty(M, like=a), _empty(M, like=a)
        call("tag_resln_head_forward", ptr(a), ptr(r), ptr(g), ptr(be), ptr(wl), ptr(bl), ptr(sim), ptr(mu), ptr(rstd), M, E,
             1e-5, p, seeds[1])
        ctx.save_for_backward(a, t, q, k, v, attn, cx, r, sim, mu, rstd, kl, wq, wk, wv, wo, g, be, wl)
        ctx.cfg = (num_heads, p, seeds)
        ctx.sinks = sinks
        ctx.params = list(params) if cfg.DIRECT_GRADS else None
        return sim

    @staticmethod
    def backward(ctx, dsim):
        a, t, q, k, v, attn, cx, r, sim, mu, rstd, kl, wq, wk, wv, wo, g, be, wl = ctx.saved_tensors
        H, p, seeds = ctx.cfg
        B, T, E = a.shape
        L, Dk = t.shape[1], t.shape[2]
        M, ML = B * T, B * L
        dsim = _chk(dsim, "grad")
        da, dr = torch.empty_like(a), torch.empty_like(a)
        gw, gg, gb = _empty(M, E, like=a), _empty(M, E, like=a), _empty(M, E, like=a)
        ds = _empty(M, like=a)
        call("tag_resln_head_backward", ptr(a), ptr(r), ptr(g), ptr(be), ptr(wl), ptr(mu), ptr(rstd), ptr(sim), ptr(dsim),
             ptr(da), ptr(dr), ptr(gw), ptr(gg), ptr(gb), ptr(ds), M, E, p, seeds[1])
        d_wl = colsum(gw, M, E).view(1, E)
        d_g, d_be = colsum(gg, M, E), colsum(gb, M, E)
        d_bl = colsum(ds, M, 1)
        # out_proj
        d_wo = gemm(dr, cx, E, E, M, transA=True, lda=E)
        d_bo = colsum(dr, M, E)
        dcx = gemm(dr, wo, M, E, E)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = _ws(query("tag_mha_cross_backward_ws_bytes", B, T, L, E), a)
        call("tag_mha_cross_backward", ptr(q), ptr(k), ptr(v), ptr(attn), ptr(dcx), ptr(kl), ptr(dq), ptr(dk), ptr(dv), B, T, L,
             E, H, p, seeds[0], ptr(ws))
        d_wq = gemm(dq, a, E, E, M, transA=True, lda=E)
        d_wk = gemm(dk, t, E, Dk, ML, transA=True, lda=E)
        d_wv = gemm(dv, t, E, Dk, ML, transA=True, lda=E)
        d_bin = torch.cat([colsum(dq, M, E), colsum(dk, ML, E), colsum(dv, ML, E)])
        gemm(dq, wq, M, E, E, out=da, accumulate=True)                     # d audio: residual branch + query projection
        dt = gemm(dk, wk, ML, Dk, E)
        gemm(dv, wv, ML, Dk, E, out=dt, accumulate=True)
        grads = [d_wq, d_wk, d_wv, d_bin, d_wo, d_bo, d_g, d_be, d_wl, d_bl]
        for i in range(len(grads)):
            _deliver(grads, ctx.sinks, i, grads[i])
        _ready(ctx.params)
        return (da, dt.view(B, L, Dk), None, None, None, None, *grads)


class RowDotFunction(TagFunction):
    """match.DotProduct with text_level='token' after a cross-encoder: one text vector per frame (models/match.py:43-60)."""

    @staticmethod
    def forward(ctx, audio, text, scale):
        a, t = _chk(audio, "audio_emb"), _chk(text, "token_emb")
        B, T, D = a.shape
        sim = _empty(B, T, like=a)
        call("tag_rowdot_sigmoid_forward", ptr(a), ptr(t), ptr(sim), B * T, D, int(scale))
        ctx.save_for_backward(a, t)
        ctx.scale = int(scale)
        return sim

    @staticmethod
    def backward(ctx, dsim):
        a, t = ctx.saved_tensors
        B, T, D = a.shape
        da, dt = torch.empty_like(a), torch.empty_like(t)
        call("tag_rowdot_sigmoid_backward", ptr(a), ptr(t), ptr(_chk(dsim, "grad")), ptr(da), ptr(dt), B * T, D, ctx.scale)
        return da, dt, None


class RowPairFunction(TagFunction):
    """Either head with text_level='token' in general (models/match.py:16-33, 43-60): text (B,T,D) holds one vector per frame.
    kind 0 = DotProduct, 1 = ExpNegL2; optional F.normalize of both operands."""

    @staticmethod
    def forward(ctx, audio, text, kind, l2norm, scale):
        a, t = _chk(audio, "audio_emb"), _chk(text, "token_emb")
        B, T, D = a.shape
        sim = _empty(B, T, like=a)
        call("tag_rowpair_forward", ptr(a), ptr(t), ptr(sim), B * T, D, int(kind), int(bool(l2norm)), int(bool(scale)))
        ctx.save_for_backward(a, t)
        ctx.cfg = (int(kind), int(bool(l2norm)), int(bool(scale)))
        return sim

    @staticmethod
    def backward(ctx, dsim):
        a, t = ctx.saved_tensors
        B, T, D = a.shape
        da, dt = torch.empty_like(a), torch.empty_like(t)
        call("tag_rowpair_backward", ptr(a), ptr(t), ptr(_chk(dsim, "grad")), ptr(da), ptr(dt), B * T, D, *ctx.cfg)
        return da, dt, None, None, None


class MatchGroupFunction(TagFunction):
    """DotProduct head of MultiTextBiEncoder (models/audio_text_model.py:150-190): N phrases per clip scored against the
    same audio embedding.  audio (B,T,D), text (B*N,D) -> sim (B*N,T)."""

    @staticmethod
    def forward(ctx, audio, text, N, scale):
        a, t = _chk(audio, "audio_emb"), _chk(text, "text_emb")
        B, T, D = a.shape
        if t.shape != (B * N, D):
            raise RuntimeError(f"text_emb must be (B*N, D) = ({B * N}, {D}), got {tuple(t.shape)}")
        sim = _empty(B * N, T, like=a)
        call("tag_match_group_forward", ptr(a), ptr(t), ptr(sim), int(scale), B, N, T, D)
        ctx.save_for_backward(a, t)
        ctx.cfg = (N, int(scale))
        return sim

    @staticmethod
    def backward(ctx, dsim):
        a, t = ctx.saved_tensors
        N, scale = ctx.cfg
        B, T, D = a.shape
        da, dt = torch.empty_like(a), torch.empty_like(t)
        call("tag_match_group_backward", ptr(a), ptr(t), ptr(_chk(dsim, "grad")), ptr(da), ptr(dt), scale, B, N, T, D)
        return da, dt, None, None


class LinearSoftmaxPoolFunction(TagFunction):
    """linear_softmax_with_lens (models/utils.py:75-76): rows (R,T) of frame probabilities -> (R,), row r uses
    length[r // group]."""

    @staticmethod
    def forward(ctx, fs, length, group):
        f = _chk(fs, "frame_sim")
        R, T = f.shape
        clip = _empty(R, like=f)
        call("tag_linear_softmax_pool_forward", ptr(f), ptr(length), ptr(clip), R, T, group)
        ctx.save_for_backward(f, length)
        ctx.group = group
        return clip

    @staticmethod
    def backward(ctx, dclip):
        f, length = ctx.saved_tensors
        R, T = f.shape
        dfs = torch.empty_like(f)
        call("tag_linear_softmax_pool_backward", ptr(f), ptr(length), ptr(_chk(dclip, "grad")), ptr(dfs), R, T, ctx.group)
        return dfs, None, None


class MeanMeanPoolFunction(TagFunction):
    """sim_pooling.AudioMeanTextMean (models/sim_pooling.py:6-22): (B,B,T,N) -> (B,B)."""

    @staticmethod
    def forward(ctx, sim, audio_len, text_len):
        s = _chk(sim, "sim")
        B, _, T, N = s.shape
        out = _empty(B, B, like=s)
        call("tag_meanmean_pool_forward", ptr(s), ptr(audio_len), ptr(text_len), ptr(out), B, T, N)
        ctx.save_for_backward(audio_len, text_len)
        ctx.shape = (B, T, N)
        return out

    @staticmethod
    def backward(ctx, dout):
        audio_len, text_len = ctx.saved_tensors
        B, T, N = ctx.shape
        dsim = torch.empty(B, B, T, N, device=dout.device, dtype=F32)
        call("tag_meanmean_pool_backward", ptr(_chk(dout, "grad")), ptr(audio_len), ptr(text_len), ptr(dsim), B, T, N)
        return dsim, None, None


class AttnPoolFunction(TagFunction):
    """AttentionPooling (models/text_encoder.py:46-58): softmax(fc(x)) over the valid tokens, weighted sum -> (B,D)."""

    @staticmethod
    def forward(ctx, x, lens, w, b):
        xs = _chk(x, "token_emb")
        B, L, D = xs.shape
        w_, b_ = _chk(w.detach(), "fc.weight").view(-1), _chk(b.detach(), "fc.bias")
        weight, out = _empty(B, L, like=xs), _empty(B, D, like=xs)
        call("tag_attnpool_forward", ptr(xs), ptr(lens), ptr(w_), ptr(b_), ptr(weight), ptr(out), B, L, D)
        ctx.save_for_backward(xs, w_, weight)
        ctx.sinks = _sinks([w, b])
        ctx.params = [w, b] if cfg.DIRECT_GRADS else None
        return out

    @staticmethod
    def backward(ctx, dout):
        xs, w_, weight = ctx.saved_tensors
        B, L, D = xs.shape
        dx, gw, gb = torch.empty_like(xs), _empty(B, D, like=xs), _empty(B, like=xs)
        call("tag_attnpool_backward", ptr(xs), ptr(w_), ptr(weight), ptr(_chk(dout, "grad")), ptr(dx), ptr(gw), ptr(gb), B, L, D)
        g = [colsum(gw, B, D).view(1, D), colsum(gb, B, 1)]
        for i in range(2):
            _deliver(g, ctx.sinks, i, g[i])
        _ready(ctx.params)
        return dx, None, g[0], g[1]


class UpsampleLinearFunction(TagFunction):
    """F.interpolate(x.unsqueeze(1), T * ratio, mode="linear", align_corners=False).squeeze(1) on (R,T) frame scores."""

    @staticmethod
    def forward(ctx, x, ratio):
        xs = _chk(x, "frame_sim")
        R, T = xs.shape
        out = _empty(R, T * ratio, like=xs)
        call("tag_upsample_linear_forward", ptr(xs), ptr(out), R, T, int(ratio))
        ctx.cfg = (R, T, int(ratio))
        return out

    @staticmethod
    def backward(ctx, dout):
        R, T, ratio = ctx.cfg
        dx = torch.empty(R, T, device=dout.device, dtype=F32)
        call("tag_upsample_linear_backward", ptr(_chk(dout, "grad")), ptr(dx), R, T, ratio)
        return dx, None


class GroupExpandFunction(TagFunction):
    """(B, ...) -> (B*N, ...): every clip's rows repeated for its N phrases (MultiTextBiEncoder with a cross-encoder,
    models/audio_text_model.py:165-168); backward sums the N copies in a fixed order."""

    @staticmethod
    def forward(ctx, x, n):
        xs = _chk(x, "audio_emb")
        B = xs.shape[0]
        R = xs.numel() // B
        out = _empty(B * n, *xs.shape[1:], like=xs)
        call("tag_group_expand_forward", ptr(xs), ptr(out), B, int(n), R)
        ctx.cfg = (xs.shape, int(n), R)
        return out

    @staticmethod
    def backward(ctx, dout):
        shape, n, R = ctx.cfg
        dx = torch.empty(shape, device=dout.device, dtype=F32)
        call("tag_group_expand_backward", ptr(_chk(dout, "grad")), ptr(dx), shape[0], n, R)
        return dx, None


class SimPoolFunction(TagFunction):
    """General similarity pooling (tag_sim_pool_*): sim (R,T,N) -> (R) or, with tmode = -1, (R,N).
    amode 0 mean / 1 max / 2 linear_softmax / 3 exp_softmax over the frames < alen[r // a_div];
    tmode 0 mean / 1 sum / 2 max / 3 mean+sum over the tokens < tlen[r % t_mod]."""

    @staticmethod
    def forward(ctx, sim, alen, tlen, a_div, t_mod, amode, tmode):
        s = _chk(sim, "sim")
        R, T, N = s.shape
        out = _empty(R, N, like=s) if tmode < 0 else _empty(R, like=s)
        call("tag_sim_pool_forward", ptr(s), ptr(alen), ptr(tlen), ptr(out), R, T, N, a_div, t_mod, amode, tmode)
        ctx.save_for_backward(s, alen, tlen if tlen is not None else alen)
        ctx.cfg = (a_div, t_mod, amode, tmode, tlen is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        s, alen, tlen = ctx.saved_tensors
        a_div, t_mod, amode, tmode, has_t = ctx.cfg
        R, T, N = s.shape
        dsim = torch.empty_like(s)
        call("tag_sim_pool_backward", ptr(s), ptr(alen), ptr(tlen) if has_t else None, ptr(_chk(dout, "grad")), ptr(dsim), R, T,
             N, a_div, t_mod, amode, tmode)
        return dsim, None, None, None, None, None, None


POOL_MODES = {"mean": 0, "max": 1, "linear_softmax": 2, "exp_softmax": 3}
TEXT_MODES = {"mean": 0, "sum": 1, "max": 2, "mean_sum": 3}


class MaxMarginFunction(TagFunction):
    """MaxMarginRankingLoss (losses.py:226-264) on an (n,n) similarity matrix; fix_norm drops the diagonal pairs."""

    @staticmethod
    def forward(ctx, x, margin, lamda1, fix_norm=True):
        xs = _chk(x, "sim")
        n = xs.shape[0]
        loss = _empty(1, like=xs)
        call("tag_maxmargin_forward", ptr(xs), n, float(margin), float(lamda1), int(bool(fix_norm)), ptr(loss))
        ctx.save_for_backward(xs)
        ctx.cfg = (float(margin), float(lamda1), int(bool(fix_norm)))
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        (xs,) = ctx.saved_tensors
        dx = torch.empty_like(xs)
        call("tag_maxmargin_backward", ptr(xs), xs.shape[0], ctx.cfg[0], ctx.cfg[1], ctx.cfg[2],
             ptr(_chk(dloss.reshape(1), "grad")), ptr(dx))
        return dx, None, None, None
