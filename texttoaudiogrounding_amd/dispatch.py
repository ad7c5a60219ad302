"""Functional wrappers over the C ABI of libtag_hip.so -- one Python function per kernel family, no autograd -- and the
DISPATCH RULES between kernel forms (direct / Winograd convolution, fused / two-pass epilogues, fp32 / bf16-MFMA arithmetics;
the switches live in settings.py).  Also what the per-kernel parity tests call.  PyTorch is plumbing here: device memory and
the current HIP stream; there is no eager fallback -- a CPU tensor or a missing library raises.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import settings as cfg
from .lib import call, ptr, query
from .engine import (F32, _chk, _empty, _timed, _ws)

# ------------------------------------------------------------------------------------------------
# thin functional wrappers (no autograd) -- also what the per-kernel parity tests call
# ------------------------------------------------------------------------------------------------

def waveform_f16_to_f32_padded(clips, device, length=None):
    """Ragged float16 clips (a list of 1-D numpy / torch float16 arrays, as WaveformStore.fetch_f16 returns them) ->
    (waveform (B,S) float32 zero-padded on ``device``, waveform_len (B,) int64 on ``device``), S = ``length`` or the longest
    clip.  The float16 samples are copied to the device back to back (half the bytes of the padded float32 batch the
    reference's collate function builds on the host) and widened + padded there (tag_waveform_f16_to_f32_padded)."""
    import numpy as np
    arrs = [c.numpy() if isinstance(c, torch.Tensor) else np.asarray(c) for c in clips]
    if any(a.dtype != np.float16 or a.ndim != 1 for a in arrs):
        raise RuntimeError("waveform_f16_to_f32_padded: clips must be 1-D float16 arrays (the pack's storage type)")
    lens = [a.shape[0] for a in arrs]
    S = int(length) if length is not None else max(lens)
    off = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.long)
    packed = torch.from_numpy(np.concatenate(arrs) if len(arrs) > 1 else arrs[0].copy())
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("waveform_f16_to_f32_padded: the target must be the MI355X (cuda) device; no CPU fallback")
    packed_d, off_d = packed.to(dev, non_blocking=True), off.to(dev, non_blocking=True)
    out = torch.empty(len(arrs), S, device=dev, dtype=F32)
    lens_d = torch.empty(len(arrs), device=dev, dtype=torch.long)
    call("tag_waveform_f16_to_f32_padded", ptr(packed_d), ptr(off_d), len(arrs), S, ptr(out), ptr(lens_d))
    return out, lens_d


def logmel(wave, n_fft, win_length, hop, window, fb, want_power=False):
    wave = _chk(wave, "waveform")
    B, S = wave.shape
    Fr = S // hop + 1
    n_mels = fb.shape[1]
    out = _empty(B, Fr, n_mels, like=wave)
    power = _empty(B, Fr, n_mels, like=wave) if want_power else None
    call("tag_logmel_forward", ptr(wave), B, S, n_fft, win_length, hop, ptr(window), ptr(fb), n_mels, ptr(out),
         ptr(power))
    return (out, power) if want_power else out


class BNStat:
    """Per-channel statistics / fused affine of one BatchNorm application."""
    __slots__ = ("mean", "invstd", "scale", "shift", "train")


def bn_stats(x2d, gamma, beta, running_mean, running_var, training, eps=1e-5, momentum=0.1, pre_op=0,
             partials=None) -> BNStat:
    """x2d: (rows, C) view of a channels-last tensor.  partials = (P, buffer) from conv3x3(..., want_stats=True): the batch
    statistics come from the conv kernel's epilogue instead of another pass over x2d."""
    rows, C = x2d.shape
    st = BNStat()
    st.train = bool(training)
    st.scale = _empty(C, like=x2d)
    st.shift = _empty(C, like=x2d)
    if training:
        st.mean = _empty(C, like=x2d)
        st.invstd = _empty(C, like=x2d)
        if x2d.dtype != F32 and not (partials is not None and pre_op == 0):
            raise RuntimeError("bf16 activations: BatchNorm batch statistics come from the producing conv kernel's epilogue")
        if partials is not None and pre_op == 0:
            ws = _ws(query("tag_bn_stats_from_partials_ws_bytes", partials[0], C), x2d)
            call("tag_bn_stats_from_partials", ptr(partials[1]), partials[0], C, ptr(gamma), ptr(beta), eps, momentum,
                 ptr(running_mean), ptr(running_var), ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift), ptr(ws))
        else:
            ws = _ws(query("tag_bn_stats_ws_bytes", rows, C), x2d)
            call("tag_bn_stats", ptr(x2d), rows, C, pre_op, ptr(gamma), ptr(beta), eps, momentum, ptr(running_mean),
                 ptr(running_var), ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift), ptr(ws))
    else:
        call("tag_bn_eval_affine", ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), eps, C, ptr(st.scale),
             ptr(st.shift))
        st.mean = running_mean
        st.invstd = _empty(C, like=x2d)
        dummy = _empty(C, like=x2d)
        call("tag_bn_eval_affine", None, None, ptr(running_mean), ptr(running_var), eps, C, ptr(st.invstd), ptr(dummy))
    return st


_X3_PRODUCTS = {"x3": 6, "x9": 9, "bf16": 1}
BF16 = torch.bfloat16


def gemm_bf16() -> bool:
    return cfg.GEMM_MATH == "bf16" or (cfg.GEMM_MATH == "auto" and cfg.ACT_DTYPE == "bf16")


def act_bf16() -> bool:
    return cfg.CONV_MATH == "bf16" and cfg.ACT_DTYPE == "bf16"


def _sfx(t) -> str:
    """Entry-point suffix for an activation tensor: '' (fp32) or '_bf16'."""
    return "_bf16" if t.dtype == BF16 else ""


def _x3_ok(W, K, N):
    return cfg.CONV_MATH in _X3_PRODUCTS and W in (8, 16, 32, 64) and K % 32 == 0 and N % 64 == 0


class _X3Pack:
    """Weight pack of the bf16-MFMA kernels: the byte blob + the product count it was made for."""
    dtype = torch.uint8

    def __init__(self, blob, products):
        self.blob, self.products = blob, products


def pack_conv_weight(w, want_dgrad=True, W=None):
    """(Cout,Cin,3,3) -> (forward pack, dgrad pack).  A pack is fp32 (9,K,N) for the exact kernels or a uint8 blob of
    pre-split bf16 fragments for the x3 kernels (when CONV_MATH == "x3" and the layer shape allows it)."""
    Cout, Cin = w.shape[0], w.shape[1]
    fx3, dx3 = _x3_ok(W, Cin, Cout), want_dgrad and _x3_ok(W, Cout, Cin)
    wf = wd = None
    if fx3 or dx3:
        nbytes = query("tag_conv3x3_x3_pack_bytes", Cin, Cout)
        xf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        xd = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        npr = _X3_PRODUCTS[cfg.CONV_MATH]
        call("tag_pack_conv_weight_x3", ptr(w), ptr(xf), ptr(xd), Cin, Cout, npr)
        wf, wd = (_X3Pack(xf, npr) if fx3 else None), (_X3Pack(xd, npr) if dx3 else None)
    if wf is None or (want_dgrad and wd is None):
        pf = _empty(9, Cin, Cout, like=w)
        pd = _empty(9, Cout, Cin, like=w) if want_dgrad else None
        call("tag_pack_conv_weight", ptr(w), ptr(pf), ptr(pd), Cin, Cout)
        wf = pf if wf is None else wf
        wd = pd if wd is None else wd
        if W is not None and _wino_shape(W, Cin, Cout) and wf is pf and (not want_dgrad or wd is pd):
            # the direct packs stay what they are (inference, pool-sum epilogues, fallbacks); the Winograd-domain weights ride along
            uf = _empty(16, Cin, Cout, like=w)
            ud = _empty(16, Cout, Cin, like=w) if want_dgrad else None
            call("tag_pack_conv_weight_wino", ptr(w), ptr(uf), ptr(ud), Cin, Cout)
            pf.wino_u = uf
            if want_dgrad:
                pd.wino_u = ud
    return wf, (wd if want_dgrad else None)


def _wino_shape(W, Cin, Cout) -> bool:
    return (cfg.CONV_WINOGRAD and cfg.CONV_MATH == "fp32" and W in (8, 16, 32, 64) and min(Cin, Cout) >= cfg.WINO_MIN_C
            and max(Cin, Cout) >= cfg.WINO_MIN_CMAX)


def _wino_flop(B, H, W, Cin, Cout) -> float:
    """FLOP a Winograd launch EXECUTES on the matrix pipe (16 products of T x Cin x Cout; bench.py's roofline counts these, not the
    2.25 x larger direct-convolution figure)."""
    return 2.0 * 16 * B * ((H + 1) // 2) * ((W + 1) // 2) * Cin * Cout


def _wino_u(wpack, x, Cout, count=True, any_size=False):
    """The Winograd-domain weights riding on a direct pack (pack_conv_weight) when this launch may use them, else None.
    any_size: the inference forward -- no tile threshold (see CONV_WINOGRAD_EVAL)."""
    u = getattr(wpack, "wino_u", None)
    if u is None or not cfg.CONV_WINOGRAD or cfg.CONV_MATH != "fp32" or x.dtype != F32:
        return None
    B, H, W, Cin = x.shape
    if any_size and not cfg.CONV_WINOGRAD_EVAL:
        return None
    if ((not any_size and B * ((H + 1) // 2) * ((W + 1) // 2) * Cout < cfg.WINO_MIN_WORK)
            or not query("tag_conv3x3_wino_ok", 1 if any_size else B, H, W, Cin, Cout)):     # (inference launches are cut by batch)
        return None
    if count:
        cfg.WINO_LAUNCHES += 1
    return u


def conv3x3(x, wpack, Cout, prologue=0, scale=None, shift=None, training_launch=False):
    """y = conv(prologue(x)).  training_launch: a launch of the training step (a dgrad conv): may take the Winograd form."""
    return conv3x3_stats(x, wpack, Cout, prologue, scale, shift, want_stats=False, training_launch=training_launch)[0]


def _batch_chunks(B, bytes_per_clip):
    """Batch slices [b0, b1) whose tensors stay under WINO_MAX_BYTES (the fused Winograd kernels' descriptor range)."""
    nb = max(1, min(B, cfg.WINO_MAX_BYTES // max(1, bytes_per_clip)))
    return [(b0, min(B, b0 + nb)) for b0 in range(0, B, nb)]


def conv3x3_stats(x, wpack, Cout, prologue=0, scale=None, shift=None, want_stats=True, training_launch=False, inference=False):
    """(y, partials): y = conv(prologue(x)) and, when want_stats, the BatchNorm partial statistics of y that the kernel
    writes in its epilogue ((P, buffer), or None when this shape has no fused statistics) -> bn_stats(..., partials=...)."""
    B, H, W, Cin = x.shape
    y = _empty(B, H, W, Cout, like=x, dtype=x.dtype)
    x3 = wpack.dtype == torch.uint8
    part = None
    u = None
    if not x3 and inference and not want_stats:
        u = _wino_u(wpack, x, Cout, any_size=True)
        if u is not None:                      # inference forward: batch cuts keep every tensor inside the descriptor range
            for b0, b1 in _batch_chunks(B, H * W * max(Cin, Cout) * 4):
                ws = _ws(query("tag_conv3x3_wino_ws_bytes", b1 - b0, H, W, Cin, Cout), x)
                with _timed(("conv3x3_wino", b1 - b0, H, W, Cin, Cout), _wino_flop(b1 - b0, H, W, Cin, Cout)):
                    call("tag_conv3x3_wino_forward", ptr(x[b0:b1]), ptr(u), prologue, ptr(scale), ptr(shift), ptr(y[b0:b1]), None,
                         b1 - b0, H, W, Cin, Cout, ptr(ws), None)
            return y, None
    elif ((want_stats and cfg.FUSE_BN_STATS) or training_launch) and not x3:
        u = _wino_u(wpack, x, Cout)
    if u is not None:
        if want_stats and cfg.FUSE_BN_STATS:
            P = query("tag_conv3x3_wino_stats_rows", B, H, W, Cout)
            part = (P, _empty(P * (3 * Cout + 1), like=x))
        ws = _ws(query("tag_conv3x3_wino_ws_bytes", B, H, W, Cin, Cout), x)
        with _timed(("conv3x3_wino", B, H, W, Cin, Cout), _wino_flop(B, H, W, Cin, Cout)):
            call("tag_conv3x3_wino_forward", ptr(x), ptr(u), prologue, ptr(scale), ptr(shift), ptr(y), ptr(part[1]) if part else None,
                 B, H, W, Cin, Cout, ptr(ws), None)
        return y, part
    if want_stats and cfg.FUSE_BN_STATS:
        if x3 and x.dtype == BF16:
            P = query("tag_conv3x3_x3_bf16_stats_rows", B, H, W, Cin, Cout, prologue)
        else:
            P = query("tag_conv3x3_x3_stats_rows" if x3 else "tag_conv3x3_stats_rows", B, H, W, Cout)
        if P > 0:
            part = (P, _empty(P * (3 * Cout + 1), like=x))
    sp = ptr(part[1]) if part else None
    if x.dtype == BF16:
        if not (x3 and wpack.products == 1):
            raise RuntimeError("bf16 activations need the one-product bf16 conv kernels (CONV_MATH='bf16') and an image width "
                               "of 8/16/32/64")
        with _timed(("conv3x3_x3_kernel", B, H, W, Cin, Cout), 2.0 * B * H * W * 9 * Cin * Cout):
            call("tag_conv3x3_forward_x3_bf16", ptr(x), ptr(wpack.blob), prologue, ptr(scale), ptr(shift), ptr(y), sp, B, H,
                 W, Cin, Cout)
    elif x3:
        with _timed(("conv3x3_x3_kernel", B, H, W, Cin, Cout), 2.0 * B * H * W * 9 * Cin * Cout):
            call("tag_conv3x3_forward_x3", ptr(x), ptr(wpack.blob), prologue, ptr(scale), ptr(shift), ptr(y), sp, B, H, W,
                 Cin, Cout, wpack.products)
    else:
        kname = "conv3x3_halo_kernel" if W in (4, 8, 16, 32, 64) else "conv3x3_fwd_kernel"   # dispatch rule of the C side
        with _timed((kname, B, H, W, Cin, Cout), 2.0 * B * H * W * 9 * Cin * Cout):
            call("tag_conv3x3_forward", ptr(x), ptr(wpack), prologue, ptr(scale), ptr(shift), ptr(y), sp, B, H, W, Cin,
                 Cout)
    return y, part


def conv3x3_dgrad_bnrelu_backward(dy_in, wpack, yref, st: BNStat, gamma, dg_out=None, db_out=None, after_conv=None,
                                  defer_apply=False):
    """The dgrad convolution da = conv(dy_in, wpack) followed by the backward of relu(bn(yref)):
    returns (dy_ref, dgamma, dbeta) with dy_ref = dL/d yref (written in place over da).  Exact-fp32 halo-tile shapes
    fold the per-channel sums into the conv epilogue; other shapes / arithmetics run the conv and tag_bnrelu_backward.
    defer_apply: a 4-tuple (t, dgamma, dbeta, applied) comes back; on the fused paths applied is False and t is still da --
    the caller's next kernel applies the BatchNorm + ReLU backward itself (conv3x3_c1_backward(bn_bwd=...))."""
    B, H, W, Cin = dy_in.shape
    C = yref.shape[3]
    fused = (cfg.FUSE_BN_BWD_SUMS and wpack.dtype != torch.uint8 and st.train and W in (8, 16, 32, 64)
             and query("tag_conv3x3_stats_rows", B, H, W, C) > 0)
    if (cfg.FUSE_BN_BWD_SUMS and dy_in.dtype == BF16 and wpack.dtype == torch.uint8 and wpack.products == 1 and st.train
            and yref.dtype == BF16 and W in (8, 16, 32, 64)):
        # BASELINE configs[2] mode: the same fusion on the one-product bf16 kernels (sums from the fp32 accumulators)
        P = query("tag_conv3x3_x3_bf16_stats_rows", B, H, W, Cin, C, 0)
        da = _empty(B, H, W, C, like=dy_in, dtype=BF16)
        part = _empty(P * 2 * C, like=dy_in)
        with _timed(("conv3x3_x3_kernel", B, H, W, Cin, C), 2.0 * B * H * W * 9 * Cin * C):
            call("tag_conv3x3_dgrad_bnsums_bf16", ptr(dy_in), ptr(wpack.blob), ptr(da), ptr(yref), ptr(st.scale), ptr(st.shift),
                 ptr(st.mean), ptr(st.invstd), ptr(part), B, H, W, Cin, C)
        if after_conv is not None:
            after_conv()
        dg = dg_out if dg_out is not None else _empty(C, like=da)
        db = db_out if db_out is not None else _empty(C, like=da)
        ws = _ws(query("tag_bn_grad_from_partials_ws_bytes", P, C), da)
        call("tag_bn_grad_from_partials", ptr(part), P, C, ptr(dg), ptr(db), ptr(ws))
        if defer_apply:
            return da, dg, db, False
        call("tag_bnrelu_backward_apply_bf16", ptr(yref), ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd),
             ptr(gamma), ptr(da), ptr(da), ptr(dg), ptr(db), B * H * W, C, int(st.train))
        return da, dg, db
    if not fused:
        da = conv3x3(dy_in, wpack, C)
        if after_conv is not None:
            after_conv()
        res = bnrelu_backward(yref, st, gamma, da, dg_out=dg_out, db_out=db_out)
        return (*res, True) if defer_apply else res
    u = _wino_u(wpack, dy_in, C)
    da = _empty(B, H, W, C, like=dy_in)
    if u is not None:
        P = query("tag_conv3x3_wino_stats_rows", B, H, W, C)
        part = _empty(P * 2 * C, like=dy_in)
        ws = _ws(query("tag_conv3x3_wino_ws_bytes", B, H, W, Cin, C), dy_in)
        with _timed(("conv3x3_wino", B, H, W, Cin, C), _wino_flop(B, H, W, Cin, C)):
            call("tag_conv3x3_wino_dgrad_bnsums", ptr(dy_in), ptr(u), ptr(da), ptr(yref), ptr(st.scale), ptr(st.shift),
                 ptr(st.mean), ptr(st.invstd), ptr(part), B, H, W, Cin, C, ptr(ws))
    else:
        P = query("tag_conv3x3_stats_rows", B, H, W, C)
        part = _empty(P * 2 * C, like=dy_in)
        with _timed(("conv3x3_halo_kernel", B, H, W, Cin, C), 2.0 * B * H * W * 9 * Cin * C):
            call("tag_conv3x3_dgrad_bnsums", ptr(dy_in), ptr(wpack), ptr(da), ptr(yref), ptr(st.scale), ptr(st.shift),
                 ptr(st.mean), ptr(st.invstd), ptr(part), B, H, W, Cin, C)
    if after_conv is not None:
        after_conv()
    dg = dg_out if dg_out is not None else _empty(C, like=da)
    db = db_out if db_out is not None else _empty(C, like=da)
    ws = _ws(query("tag_bn_grad_from_partials_ws_bytes", P, C), da)
    call("tag_bn_grad_from_partials", ptr(part), P, C, ptr(dg), ptr(db), ptr(ws))
    if defer_apply:
        return da, dg, db, False
    rows = B * H * W
    call("tag_bnrelu_backward_apply", ptr(yref), ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd), ptr(gamma),
         ptr(da), ptr(da), ptr(dg), ptr(db), rows, C, int(st.train))
    return da, dg, db


def pool_sums_fusable(dy_in, wpack, yref, ph, pw):
    """Can the dgrad conv of (dy_in, wpack) carry the pool-backward sums of the block below (raw output yref, window ph x pw)?
    Exact-fp32 halo-tile shapes, windows 1x2 / 2x2."""
    B, H, W, Cin = dy_in.shape
    if not (cfg.FUSE_POOL_BWD_SUMS and W in (8, 16, 32, 64) and pw == 2 and ph in (1, 2) and H == yref.shape[1] // ph
            and W == yref.shape[2] // pw):
        return False
    if dy_in.dtype == BF16:       # BASELINE configs[2] mode: the one-product bf16 tile kernel with the staged output tile
        return (cfg.FUSE_POOL_BWD_SUMS_BF16 and yref.dtype == BF16 and wpack.dtype == torch.uint8 and getattr(wpack, "products", 0) == 1
                and query("tag_conv3x3_dgrad_poolsums_bf16_rows", B, H, W, Cin, yref.shape[3]) > 0)
    return (dy_in.dtype == F32 and yref.dtype == F32 and wpack.dtype != torch.uint8
            and query("tag_conv3x3_stats_rows", B, H, W, yref.shape[3]) > 0)


def conv3x3_dgrad_poolsums(dy_in, wpack, yref, st: BNStat, ph, pw, drop_p=0.0, seed=0, pool=0):
    """dx = conv(dy_in, wpack) -- the gradient of the pooled (and dropped-out) output of the block whose second conv produced
    yref -- and, from the conv's epilogue, the partial sums (P, buffer) of that block's BatchNorm+ReLU+pool backward
    -> bnrelu_pool_backward(..., partials=...)."""
    B, H, W, Cin = dy_in.shape
    _, Hf, Wf, C = yref.shape
    if dy_in.dtype == BF16:
        P = query("tag_conv3x3_dgrad_poolsums_bf16_rows", B, H, W, Cin, C)
        dx = _empty(B, H, W, C, like=dy_in, dtype=BF16)
        part = _empty(P * 2 * C, like=dy_in)
        with _timed(("conv3x3_x3_kernel", B, H, W, Cin, C), 2.0 * B * H * W * 9 * Cin * C):
            call("tag_conv3x3_dgrad_poolsums_bf16", ptr(dy_in), ptr(wpack.blob), ptr(dx), ptr(yref), ptr(st.scale), ptr(st.shift),
                 ptr(st.mean), ptr(st.invstd), ptr(part), B, H, W, Cin, C, Hf, Wf, ph, pw, int(pool), float(drop_p), seed)
        return dx, (P, part)
    u = _wino_u(wpack, dy_in, C)
    if u is not None:                  # Winograd dgrad: the sums come from its output transform (conv_wino.hip, EPI == 2)
        P = query("tag_conv3x3_wino_stats_rows", B, H, W, C)
        dx = _empty(B, H, W, C, like=dy_in)
        part = _empty(P * 2 * C, like=dy_in)
        ws = _ws(query("tag_conv3x3_wino_ws_bytes", B, H, W, Cin, C), dy_in)
        with _timed(("conv3x3_wino", B, H, W, Cin, C), _wino_flop(B, H, W, Cin, C)):
            call("tag_conv3x3_wino_dgrad_poolsums", ptr(dy_in), ptr(u), ptr(dx), ptr(yref), ptr(st.scale), ptr(st.shift), ptr(st.mean),
                 ptr(st.invstd), ptr(part), B, H, W, Cin, C, Hf, Wf, ph, pw, int(pool), float(drop_p), seed, ptr(ws))
        return dx, (P, part)
    P = query("tag_conv3x3_stats_rows", B, H, W, C)
    dx = _empty(B, H, W, C, like=dy_in)
    part = _empty(P * 2 * C, like=dy_in)
    with _timed(("conv3x3_halo_kernel", B, H, W, Cin, C), 2.0 * B * H * W * 9 * Cin * C):
        call("tag_conv3x3_dgrad_poolsums", ptr(dy_in), ptr(wpack), ptr(dx), ptr(yref), ptr(st.scale), ptr(st.shift), ptr(st.mean),
             ptr(st.invstd), ptr(part), B, H, W, Cin, C, Hf, Wf, ph, pw, int(pool), float(drop_p), seed)
    return dx, (P, part)


def eval_pool_fusable(x, wpack, ph, pw, pool=0):
    """Exact-fp32 halo-tile shapes, windows 1x2 / 2x2, the three pool types."""
    B, H, W, _ = x.shape
    return (cfg.FUSE_EVAL_POOL and x.dtype == F32 and wpack.dtype != torch.uint8 and W in (8, 16, 32, 64) and pw == 2 and ph in (1, 2)
            and H // ph > 0 and pool in (0, 2, 3) and query("tag_conv3x3_stats_rows", B, H, W, 64) > 0)


def conv3x3_bnrelu_pool_eval(x, wpack, Cout, st: BNStat, ph, pw, prologue=0, scale=None, shift=None, pool=0):
    """pool(relu(bn_eval(conv(prologue(x))))) in ONE kernel: nothing of the (B,H,W,Cout) conv output touches HBM."""
    B, H, W, Cin = x.shape
    out = _empty(B, H // ph, W // pw, Cout, like=x)
    u = _wino_u(wpack, x, Cout, any_size=True)
    if u is not None:
        for b0, b1 in _batch_chunks(B, H * W * max(Cin, Cout) * 4):
            ws = _ws(query("tag_conv3x3_wino_ws_bytes", b1 - b0, H, W, Cin, Cout), x)
            with _timed(("conv3x3_wino", b1 - b0, H, W, Cin, Cout), _wino_flop(b1 - b0, H, W, Cin, Cout)):
                call("tag_conv3x3_wino_forward_bnrelu_pool_eval", ptr(x[b0:b1]), ptr(u), prologue, ptr(scale), ptr(shift),
                     ptr(out[b0:b1]), ptr(st.scale), ptr(st.shift), b1 - b0, H, W, Cin, Cout, ph, pw, int(pool), ptr(ws))
        return out
    with _timed(("conv3x3_halo_kernel", B, H, W, Cin, Cout), 2.0 * B * H * W * 9 * Cin * Cout):
        call("tag_conv3x3_forward_bnrelu_pool_eval", ptr(x), ptr(wpack), prologue, ptr(scale), ptr(shift), ptr(out), ptr(st.scale),
             ptr(st.shift), B, H, W, Cin, Cout, ph, pw, int(pool))
    return out


def conv3x3_wgrad(x, dy, prologue=0, scale=None, shift=None, out=None):
    B, H, W, Cin = x.shape
    Cout = dy.shape[3]
    dw = out if out is not None else _empty(Cout, Cin, 3, 3, like=x)
    if x.dtype == BF16:
        if dy.dtype != BF16 or not (W in (8, 16, 32, 64) and Cin % 64 == 0 and Cout % 64 == 0):
            raise RuntimeError("bf16 wgrad: both operands must be bf16, width 8/16/32/64, channels multiples of 64")
        ws = _ws(query("tag_conv3x3_wgrad_x3_ws_bytes", B, H, W, Cin, Cout), x)
        with _timed(("conv3x3_wgrad_x3_kernel", B, H, W, Cin, Cout), 2.0 * B * H * W * 9 * Cin * Cout):
            call("tag_conv3x3_wgrad_x3_bf16", ptr(x), prologue, ptr(scale), ptr(shift), ptr(dy), ptr(dw), B, H, W, Cin, Cout,
                 ptr(ws))
        return dw
    if cfg.CONV_MATH in _X3_PRODUCTS and W in (8, 16, 32, 64) and Cin % 64 == 0 and Cout % 64 == 0:
        ws = _ws(query("tag_conv3x3_wgrad_x3_ws_bytes", B, H, W, Cin, Cout), x)
        with _timed(("conv3x3_wgrad_x3_kernel", B, H, W, Cin, Cout), 2.0 * B * H * W * 9 * Cin * Cout):
            call("tag_conv3x3_wgrad_x3", ptr(x), prologue, ptr(scale), ptr(shift), ptr(dy), ptr(dw), B, H, W, Cin, Cout,
                 _X3_PRODUCTS[cfg.CONV_MATH], ptr(ws))
        return dw
    if (x.dtype == F32 and dy.dtype == F32 and _wino_shape(W, Cin, Cout)
            and B * ((H + 1) // 2) * ((W + 1) // 2) * max(Cin, Cout) >= cfg.WINO_MIN_WORK and query("tag_conv3x3_wino_ok", B, H, W, Cin, Cout)):
        cfg.WINO_LAUNCHES += 1
        ws = _ws(query("tag_conv3x3_wino_wgrad_ws_bytes", B, H, W, Cin, Cout), x)
        with _timed(("conv3x3_wino_wgrad", B, H, W, Cin, Cout), _wino_flop(B, H, W, Cin, Cout)):
            call("tag_conv3x3_wino_wgrad", ptr(x), prologue, ptr(scale), ptr(shift), ptr(dy), ptr(dw), B, H, W, Cin, Cout, ptr(ws),
                 None)
        return dw
    ws = _ws(query("tag_conv3x3_wgrad_ws_bytes", B, H, W, Cin, Cout), x)
    # profile family: the all-taps decomposition (conv3x3_wgrad_alltaps_kernel at W = 8 / 16, its row-ring form
    # conv3x3_wgrad_rowring_kernel at W = 32 / 64) or the per-tap fallback
    kname = "conv3x3_wgrad_alltaps_kernel" if W in (8, 16, 32, 64) else "conv3x3_wgrad_kernel"
    with _timed((kname, B, H, W, Cin, Cout), 2.0 * B * H * W * 9 * Cin * Cout):
        call("tag_conv3x3_wgrad", ptr(x), prologue, ptr(scale), ptr(shift), ptr(dy), ptr(dw), B, H, W, Cin, Cout,
             ptr(ws))
    return dw


def conv3x3_c1(x, w, col_scale=None, col_shift=None):
    return conv3x3_c1_stats(x, w, col_scale, col_shift, want_stats=False)[0]


def conv3x3_c1_stats(x, w, col_scale=None, col_shift=None, want_stats=True, out_dtype=F32):
    """(y, partials) of the Cin = 1 convolution; partials = (P, buffer) when the kernel wrote the BatchNorm statistics of
    y itself (W == 64, Cout == 64), else None.  out_dtype bf16: y stored as bf16 (statistics from the fp32 values)."""
    B, H, W = x.shape
    Cout = w.shape[0]
    y = _empty(B, H, W, Cout, like=x, dtype=out_dtype)
    if out_dtype == BF16:
        P = query("tag_conv3x3_c1_stats_rows", B, H, W, Cout)
        if P <= 0:
            raise RuntimeError("bf16 activations: the Cin = 1 convolution is implemented for 64 mel bins x 64 channels")
        part = (P, _empty(P * (3 * Cout + 1), like=x)) if want_stats else None
        call("tag_conv3x3_c1_forward_stats_bf16", ptr(x), ptr(col_scale), ptr(col_shift), ptr(w), ptr(y),
             ptr(part[1]) if part else None, B, H, W, Cout)
        return y, part
    P = query("tag_conv3x3_c1_stats_rows", B, H, W, Cout) if (want_stats and cfg.FUSE_BN_STATS) else 0
    if P > 0:
        part = (P, _empty(P * (3 * Cout + 1), like=x))
        call("tag_conv3x3_c1_forward_stats", ptr(x), ptr(col_scale), ptr(col_shift), ptr(w), ptr(y), ptr(part[1]), B, H, W,
             Cout)
        return y, part
    call("tag_conv3x3_c1_forward", ptr(x), ptr(col_scale), ptr(col_shift), ptr(w), ptr(y), B, H, W, Cout)
    return y, None


def conv3x3_c1_wgrad(x, dy, col_scale=None, col_shift=None, out=None):
    B, H, W = x.shape
    Cout = dy.shape[3]
    dw = out if out is not None else _empty(Cout, 1, 3, 3, like=x)
    ws = _ws(query("tag_conv3x3_c1_wgrad_ws_bytes", B, H, W, Cout), x)
    call("tag_conv3x3_c1_wgrad", ptr(x), ptr(col_scale), ptr(col_shift), ptr(dy), ptr(dw), B, H, W, Cout, ptr(ws))
    return dw


C1_BWD_FUSED_SHAPE = (64, 64)            # (mel bins, channels) the one-pass Cin = 1 backward is written for


def conv3x3_c1_backward(x, dy, w, col_scale=None, col_shift=None, out=None, bn_bwd=None):
    """(dw, dx) of the Cin = 1 convolution; one fused pass over dy when the shape allows (W == 64, Cout == 64).
    bn_bwd = (yref, st, gamma, dgamma, dbeta): `dy` is da = dL/d relu(bn(yref)) and the BatchNorm + ReLU backward is applied
    while it is loaded (tag_conv3x3_c1_backward_bnrelu; fused shape only)."""
    B, H, W = x.shape
    Cout = dy.shape[3]
    if (W, Cout) == C1_BWD_FUSED_SHAPE:
        dw = out if out is not None else _empty(Cout, 1, 3, 3, like=x)
        dx = _empty(B, H, W, like=x)
        ws = _ws(query("tag_conv3x3_c1_backward_ws_bytes", B, H, W, Cout), x)
        if bn_bwd is not None:
            yref, st, gamma, dg, db = bn_bwd
            if yref.dtype != dy.dtype or yref.shape != dy.shape:
                raise RuntimeError("conv3x3_c1_backward: yref and da must agree in dtype and shape")
            call("tag_conv3x3_c1_backward_bnrelu" + _sfx(dy), ptr(x), ptr(col_scale), ptr(col_shift), ptr(dy), ptr(yref),
                 ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd), ptr(gamma), ptr(dg), ptr(db), int(st.train),
                 ptr(w), ptr(dw), ptr(dx), B, H, W, Cout, ptr(ws))
            return dw, dx
        call("tag_conv3x3_c1_backward" + _sfx(dy), ptr(x), ptr(col_scale), ptr(col_shift), ptr(dy), ptr(w), ptr(dw), ptr(dx),
             B, H, W, Cout, ptr(ws))
        return dw, dx
    if bn_bwd is not None:
        raise RuntimeError("conv3x3_c1_backward: bn_bwd needs the fused 64 x 64 shape")
    if dy.dtype == BF16:
        raise RuntimeError("bf16 activations: the Cin = 1 backward is implemented for 64 mel bins x 64 channels")
    return conv3x3_c1_wgrad(x, dy, col_scale, col_shift), conv3x3_c1_dgrad(dy, w)


def conv3x3_c1_dgrad(dy, w):
    B, H, W, Cout = dy.shape
    dx = _empty(B, H, W, like=dy)
    call("tag_conv3x3_c1_dgrad", ptr(dy), ptr(w), ptr(dx), B, H, W, Cout)
    return dx


def bnact_pool(y, st: Optional[BNStat], ph, pw, act=1, pool=0, drop_p=0.0, seed=0):
    B, H, W, C = y.shape
    out = _empty(B, H // ph, W // pw, C, like=y, dtype=y.dtype)
    call("tag_bnact_pool_forward" + _sfx(y), ptr(y), ptr(st.scale) if st else None, ptr(st.shift) if st else None, ptr(out), B,
         H, W, C, ph, pw, act, pool, float(drop_p), seed)
    return out


def bnrelu_pool_backward(y, st: BNStat, gamma, dout, ph, pw, drop_p=0.0, seed=0, dg_out=None, db_out=None, pool=0,
                         partials=None):
    """partials = (P, buffer) from conv3x3_dgrad_poolsums: the sums were taken by the conv that produced dout; only the apply
    pass runs here."""
    B, H, W, C = y.shape
    if dout.dtype != y.dtype:
        raise RuntimeError("bnrelu_pool_backward: y and dout must share their storage type")
    dy = _empty(B, H, W, C, like=y, dtype=y.dtype)
    dg = dg_out if dg_out is not None else _empty(C, like=y)
    db = db_out if db_out is not None else _empty(C, like=y)
    if partials is not None:
        P, part = partials
        ws = _ws(query("tag_bn_grad_from_partials_ws_bytes", P, C), y)
        call("tag_bn_grad_from_partials", ptr(part), P, C, ptr(dg), ptr(db), ptr(ws))
        call("tag_bnrelu_pool_backward_apply" + _sfx(y), ptr(y), ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd),
             ptr(gamma), ptr(dout), ptr(dy), ptr(dg), ptr(db), B, H, W, C, ph, pw, int(pool), float(drop_p), seed, int(st.train))
        return dy, dg, db
    ws = _ws(query("tag_bn_backward_ws_bytes", B * H * W, C), y)
    call("tag_bnrelu_pool_backward" + _sfx(y), ptr(y), ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd), ptr(gamma),
         ptr(dout), ptr(dy), ptr(dg), ptr(db), B, H, W, C, ph, pw, int(pool), float(drop_p), seed, int(st.train), ptr(ws))
    return dy, dg, db


def bnrelu_backward(y, st: BNStat, gamma, da, inplace=True, dg_out=None, db_out=None):
    C = y.shape[-1]
    rows = y.numel() // C
    dy = da if inplace else torch.empty_like(da)
    dg = dg_out if dg_out is not None else _empty(C, like=y)
    db = db_out if db_out is not None else _empty(C, like=y)
    ws = _ws(query("tag_bn_backward_ws_bytes", rows, C), y)
    if da.dtype != y.dtype:
        raise RuntimeError("bnrelu_backward: y and da must share their storage type")
    call("tag_bnrelu_backward" + _sfx(y), ptr(y), ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd), ptr(gamma),
         ptr(da), ptr(dy), ptr(dg), ptr(db), rows, C, int(st.train), ptr(ws))
    return dy, dg, db


def bn_param_grad(x2d, dy2d, st: BNStat, dg_out=None, db_out=None):
    rows, C = x2d.shape
    dg = dg_out if dg_out is not None else _empty(C, like=x2d)
    db = db_out if db_out is not None else _empty(C, like=x2d)
    ws = _ws(query("tag_bn_backward_ws_bytes", rows, C), x2d)
    call("tag_bn_param_grad", ptr(x2d), ptr(dy2d), rows, C, ptr(st.mean), ptr(st.invstd), ptr(dg), ptr(db), ptr(ws))
    return dg, db


def dropout_mask(seed, shape, p, device, pooled=False):
    """The keep mask (0/1 bytes) a kernel draws for `seed`; pooled=True: the generator of the pooled activations' dropout
    (one hash per 4 elements; bnact_pool / bnrelu_pool_backward), else the per-element one (mean_w, dropout, attention)."""
    n = int(math.prod(shape))
    m = torch.empty(n, device=device, dtype=torch.uint8)
    call("tag_dropout_mask_pooled" if pooled else "tag_dropout_mask", seed, n, float(p), ptr(m))
    return m.view(*shape)


def gemm(A, B, M, N, K, transA=False, transB=False, lda=None, ldb=None, out=None, ldc=None, bias=None, act=0,
         accumulate=False):
    """Row-major C(M,N) = act(op(A) op(B) + bias) [+ C].  A/B may be strided views (lda/ldb).

    Split-K (a workspace) is offered to the library only for transA products -- the weight-gradient shape, whose reduction
    runs over the batch rows.  A product with A stored (M,K) has the batch in M: without K slices every output row is one
    fixed-order sum over k whatever M is, so forward passes are BATCH-INVARIANT (a clip scores bit-identically alone, in a
    64-clip pass or in a ragged remainder; tools/diag_batch_invariance.py, test_grounding_model_30s_full_pass_b67)."""
    lda = lda if lda is not None else (M if transA else K)
    ldb = ldb if ldb is not None else (K if transB else N)
    if out is None:
        out = _empty(M, N, like=A)
    ldc = ldc if ldc is not None else N
    nws = query("tag_gemm_ws_bytes", M, N, K) if transA else 0
    ws = _ws(nws, A) if nws else None
    call("tag_gemm_bf16" if gemm_bf16() else "tag_gemm", ptr(A), lda, int(transA), ptr(B), ldb, int(transB), ptr(out), ldc, M,
         N, K, ptr(bias), act, int(accumulate), ptr(ws))
    return out


def colsum(x, M, N, ld=None, out=None):
    out = out if out is not None else _empty(N, like=x)
    ws = _ws(query("tag_colsum_ws_bytes", M, N), x)
    call("tag_colsum", ptr(x), ld if ld is not None else N, M, N, ptr(out), ptr(ws))
    return out


def relu_backward(y, dy):
    call("tag_relu_backward", ptr(y), ptr(dy), ptr(dy), y.numel())
    return dy


def segments(frame_sim, thresholds, window_size, n_connect):
    """P1 on the device.  Returns (regions (B,NT,maxK,2) int64, counts (B,NT) int32)."""
    frame_sim = _chk(frame_sim, "frame_sim")
    B, T = frame_sim.shape
    th = torch.as_tensor(thresholds, dtype=torch.float64).to(frame_sim.device)
    NT = th.numel()
    maxk = (T + 1) // 2
    regions = torch.zeros(B, NT, maxk, 2, device=frame_sim.device, dtype=torch.int64)
    counts = torch.zeros(B, NT, device=frame_sim.device, dtype=torch.int32)
    call("tag_segments", ptr(frame_sim), T, B, T, ptr(th), NT, int(window_size), int(n_connect), ptr(regions),
         ptr(counts), maxk)
    return regions, counts


def align_dot(audio, text, l2norm=False, scaled=False):
    """align.DotProduct forward (models/align.py:14-31): (B,T,D),(B,N,D) -> (B,B,T,N); F.normalize of both operands first
    when l2norm (row kernels), then the MFMA GEMM with the [/sqrt D ->] sigmoid -> clamp -> (B,B,T,N) scatter epilogue."""
    audio, text = _chk(audio, "audio"), _chk(text, "text")
    B, T, D = audio.shape
    N = text.shape[1]
    if l2norm:
        audio, text = _l2norm_rows(audio, B * T, D), _l2norm_rows(text, B * N, D)
    out = _empty(B, B, T, N, like=audio)
    call("tag_align_dot_forward", ptr(audio), ptr(text), ptr(out), 0, int(scaled), B, T, N, D, None)
    return out


# ------------------------------------------------------------------------------------------------
# bidirectional GRU (row A4): input projection GEMM + persistent recurrence, and its backward
# ------------------------------------------------------------------------------------------------

#: persistent scratch of the GRU kernels per (device, B, H, pass): recurrent-weight transpose, exchange granules and a
#: STICKY error word (last 256 bytes; zeroed once here, raised by a persistent kernel whose bounded spin ran out and never
#: cleared by the library) -> check_async_errors()
_gru_scratch = {}


_GRU_SCRATCH_MAX = 8        # (B, H, pass) combinations kept per process; ragged epochs vary T, which the scratch ignores


def _gru_ws(B, T, Hh, like, which):
    """Scratch of one persistent GRU launch.  tag_gru_ws_bytes does not depend on T, so the cache key does not either (a
    ragged epoch pads every batch to its own longest clip); the least recently used entry is dropped beyond
    _GRU_SCRATCH_MAX.  Kernels run in stream order, so equal-shape GRUs may share one scratch."""
    key = (like.device.type, like.device.index, B, Hh, which)
    ws = _gru_scratch.pop(key, None)
    if ws is None:
        nbytes = query("tag_gru_ws_bytes", B, T, Hh)
        ws = torch.zeros((nbytes + 7) // 8, device=like.device, dtype=torch.float64)
        ws._tag_err_index = (nbytes - 256) // 4          # int32 index of the sticky error word
        while len(_gru_scratch) >= _GRU_SCRATCH_MAX:
            old_key = next(iter(_gru_scratch))
            _check_gru_word(old_key, _gru_scratch.pop(old_key))      # an evicted scratch must not take a raised flag with it
    _gru_scratch[key] = ws                                # (re-)inserted last = most recently used
    return ws


def _check_gru_word(key, ws):
    word = ws.view(torch.int32)[ws._tag_err_index: ws._tag_err_index + 1]
    err = word.cpu()
    if query("tag_gru_timed_out", err.data_ptr()):
        word.zero_()
        was_fast = query("tag_gru_disable_xcd_fast")      # later launches publish write-through (correct under every placement)
        raise RuntimeError(f"persistent GRU kernel timed out waiting for a neighbouring workgroup (B,H,pass = {key[2:]}): "
                           "its workgroups were not co-resident or an L2-resident exchange granule was read stale; outputs of "
                           "that step are NaN and the optimiser skipped it"
                           + ("; the same-XCD L2 publishing is now OFF for this process" if was_fast else ""))


def check_async_errors():
    """Host-side check of the sticky device error words (synchronises): raises RuntimeError when a persistent GRU kernel
    timed out waiting for its neighbours (its outputs were poisoned with NaN and the Adam kernel skipped the step) or an
    embedding lookup saw a token id outside the table.  Called by StrongRunner whenever it hands a loss VALUE to the host."""
    for key, ws in list(_gru_scratch.items()):
        _check_gru_word(key, ws)
    for dev, flag in _embed_err.items():
        if int(flag.cpu().item()) != 0:
            flag.zero_()
            raise IndexError("embedding lookup: token id out of range (nn.Embedding would raise; models/text_encoder.py:39)")


_embed_err = {}


def _embed_flag(like):
    key = (like.device.type, like.device.index)
    if key not in _embed_err:
        _embed_err[key] = torch.zeros(1, device=like.device, dtype=torch.int32)
    return _embed_err[key]


def _joined(a, b, shape):
    """``torch.cat / stack([a, b])`` as a VIEW when b lies right behind a in the same storage (runner.FlatParams lays the
    two directions of an nn.GRU out that way), else a copy: no concatenation kernels per step on the flat-parameter path."""
    if (a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and a.device == b.device
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and b.storage_offset() == a.storage_offset() + a.numel()):
        return torch.empty(0, device=a.device, dtype=a.dtype).set_(a.untyped_storage(), a.storage_offset(), tuple(shape))
    return torch.cat([a.reshape(-1), b.reshape(-1)]).view(*shape)


def bump_bn_counters(owner, bns):
    """``num_batches_tracked += 1`` of the BatchNorm modules ``bns`` (all in train mode) as ONE kernel: the nine 0-dim
    int64 buffers are re-homed (once per device move) as views of one flat tensor kept on ``owner``; state_dict keys,
    load_state_dict and the rank-0 buffer broadcast see the same buffers as before."""
    flat = getattr(owner, "_tag_nbt_flat", None)
    ok = (flat is not None and flat.numel() == len(bns) and flat.device == bns[0].num_batches_tracked.device
          and all(m._buffers["num_batches_tracked"].data_ptr() == flat.data_ptr() + 8 * i for i, m in enumerate(bns)))
    if not ok:
        flat = torch.stack([m.num_batches_tracked.detach().to(torch.long) for m in bns])
        for i, m in enumerate(bns):
            m._buffers["num_batches_tracked"] = flat[i]
        owner._tag_nbt_flat = flat
    flat += 1


def max_clips_per_pass(frames: int) -> int:
    """Largest batch the kernels take in one launch.  Round 4: the conv kernels add a 64-bit per-image base to 32-bit offsets
    INSIDE the image, so what is left is the 31-bit PIXEL index of the BatchNorm / pool passes over the largest tensor, the
    first block's (B, frames, 64 mel) pixels -- 33 520 clips of 10 s (rounds 1-3: 32-bit byte offsets over the whole batch, 261
    clips of 10 s in fp32).  Memory is the practical limit: ~75 MB of saved activations per 10 s clip in fp32."""
    return max(1, (2 ** 31 - 1) // (int(frames) * 64))


def check_pass_size(B: int, frames: int):
    """Loud and early instead of TAG_EINVAL from the first kernel: a batch beyond the 31-bit pixel index must be split by the
    CALLER (train-mode BatchNorm statistics are per forward pass, so the split is not invisible; the inference wrapper
    models/hf_modeling_grounding.py splits into passes of 64 for memory -- eval-mode BatchNorm makes its passes independent)."""
    lim = max_clips_per_pass(frames)
    if B > lim:
        raise RuntimeError(f"batch of {B} clips x {frames} frames exceeds the kernels' 31-bit pixel index: at most {lim} clips of "
                           "this length per forward pass. Split the batch (gradient accumulation over sub-batches; note that "
                           "train-mode BatchNorm statistics are then per sub-batch, as they would be with a smaller batch in the "
                           "reference)")


def gru_bidir_forward(x2d, rnn, B, T, need_grad):
    """x2d (B*T, I); rnn = [w_ih, w_hh, b_ih, b_hh] x (forward, reverse).  Returns y (B,T,2H) and the saved state."""
    Hh = rnn[1].shape[1]
    M = B * T
    I = rnn[0].shape[1]
    w_ih = _joined(rnn[0], rnn[4], (6 * Hh, I))            # (2*3H, I)
    b_ih = _joined(rnn[2], rnn[6], (6 * Hh,))
    w_hh = _joined(rnn[1], rnn[5], (2, 3 * Hh, Hh))
    b_hh = _joined(rnn[3], rnn[7], (2, 3 * Hh))
    gi = gemm(x2d, w_ih, M, 6 * Hh, x2d.shape[1], transB=True, bias=b_ih)
    y = _empty(B, T, 2 * Hh, like=x2d)
    gates = _empty(B, T, 2, 4 * Hh, like=x2d) if need_grad else None
    wsr = _gru_ws(B, T, Hh, x2d, "fwd")
    call("tag_gru_forward", ptr(gi), ptr(w_hh), ptr(b_hh), ptr(y), ptr(gates), ptr(wsr), B, T, Hh)
    return y, (dict(gates=gates, y=y, w_ih=w_ih, w_hh=w_hh, Hh=Hh) if need_grad else None)


def _adjacent_view(a, b, shape):
    """The view over ``a`` and ``b`` as ONE tensor when b lies right behind a in the same storage, else None."""
    if (a is not None and b is not None and a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype
            and a.device == b.device and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and b.storage_offset() == a.storage_offset() + a.numel()):
        return torch.empty(0, device=a.device, dtype=a.dtype).set_(a.untyped_storage(), a.storage_offset(), tuple(shape))
    return None


def gru_bidir_backward(dy, x2d, sv, outs=None, side=None):
    """Returns (dx2d, [8 parameter gradients in nn.GRU order]).  outs: optional 8 destination tensors (flat-gradient
    views); a gradient whose destination is given is written there directly -- the bias gradients too when the two
    directions' sinks are adjacent (runner.FlatParams lays them out that way): one column sum fills both.
    side: a _SideWgrad; when EVERY gradient has its destination, the parameter-gradient work (2 column sums + 4 GEMMs, all
    off the dx chain) runs on its side stream beside the memory-bound passes that follow on the main stream."""
    Hh, y, gates = sv["Hh"], sv["y"], sv["gates"]
    B, T, _ = y.shape
    M = B * T
    dgi = _empty(B, T, 2, 3 * Hh, like=y)
    dgh = _empty(B, T, 2, 3 * Hh, like=y)
    hprev = _empty(B, T, 2, Hh, like=y)
    scratch = _gru_ws(B, T, Hh, y, "bwd")
    call("tag_gru_backward", ptr(dy), ptr(y), ptr(gates), ptr(sv["w_hh"]), ptr(dgi), ptr(dgh), ptr(hprev),
         ptr(scratch), B, T, Hh)
    I = x2d.shape[1]
    outs = list(outs) if outs is not None else [None] * 8
    g = [None] * 8
    bih_sink = _adjacent_view(outs[2], outs[6], (6 * Hh,))
    bhh_sink = _adjacent_view(outs[3], outs[7], (6 * Hh,))
    all_direct = bih_sink is not None and bhh_sink is not None and all(o is not None for o in outs)

    def param_grads():
        db_ih = colsum(dgi, M, 6 * Hh, out=bih_sink)
        db_hh = colsum(dgh, M, 6 * Hh, out=bhh_sink)
        for d in range(2):
            ai = dgi.view(M, 6 * Hh)[:, d * 3 * Hh:]
            a = dgh.view(M, 6 * Hh)[:, d * 3 * Hh:]
            hb = hprev.view(M, 2 * Hh)[:, d * Hh:]
            g[4 * d + 0] = gemm(ai, x2d, 3 * Hh, I, M, transA=True, lda=6 * Hh, out=outs[4 * d + 0])
            g[4 * d + 1] = gemm(a, hb, 3 * Hh, Hh, M, transA=True, lda=6 * Hh, ldb=2 * Hh, out=outs[4 * d + 1])
            g[4 * d + 2] = outs[4 * d + 2] if bih_sink is not None else db_ih[d * 3 * Hh:(d + 1) * 3 * Hh]
            g[4 * d + 3] = outs[4 * d + 3] if bhh_sink is not None else db_hh[d * 3 * Hh:(d + 1) * 3 * Hh]
    if side is not None and all_direct:
        side.run(param_grads, (dgi, dgh, hprev, x2d))
    else:
        param_grads()
    dx = gemm(dgi, sv["w_ih"], M, I, 6 * Hh)
    return dx, g


# ------------------------------------------------------------------------------------------------
# One conv3x3 -> BatchNorm -> ReLU (-> pool) stage as a standalone operator: what SURVEY.md section 8(b) lists as
# ``conv3x3_bn_relu[_pool]`` and what ConvBlock.forward (models/panns.py:46-62) is made of.  The fused Cnn8Rnn node (functions.py)
# never materialises relu(bn(y)); this stage does (its output IS that tensor, pooled), so that it composes like an nn.Module.
# ------------------------------------------------------------------------------------------------
POOL_TYPES = {"avg+max": 0, "avg": 2, "max": 3}
POOL_SIZES = {(1, 1), (1, 2), (2, 1), (2, 2)}      # (time, mel) windows instantiated for forward AND backward in bn_pool.hip


def conv_bn_relu_pool_forward(x, w, gamma, beta, running_mean, running_var, training, momentum, eps, ph, pw, pool):
    """x channels-last (B,H,W,Cin) fp32; w (Cout,Cin,3,3).  -> (out (B,H/ph,W/pw,Cout), y raw conv output, BNStat).
    running statistics are updated in place when training (nn.BatchNorm2d semantics)."""
    x = _chk(x, "x")
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    if (ph, pw) not in POOL_SIZES:
        raise RuntimeError(f"conv3x3_bn_relu_pool: pool_size {(ph, pw)} has no kernel instance (built: {sorted(POOL_SIZES)})")
    if Cin == 1:
        y, part = conv3x3_c1_stats(x.view(B, H, W), w, want_stats=training)
    elif Cin % 32 == 0:
        wf, _ = pack_conv_weight(w, want_dgrad=False, W=W)
        y, part = conv3x3_stats(x, wf, Cout, want_stats=training)
    else:
        raise RuntimeError(f"conv3x3_bn_relu_pool: in_channels must be 1 or a multiple of 32, got {Cin}")
    st = bn_stats(y.view(-1, Cout), gamma, beta, running_mean, running_var, training, eps, momentum, partials=part)
    out = bnact_pool(y, st, ph, pw, act=1, pool=pool)
    return out, y, st


def conv_bn_relu_pool_backward(dout, x, w, y, st: BNStat, gamma, ph, pw, pool, need_dx=True):
    """-> (dx or None, dw, dgamma, dbeta): BatchNorm/ReLU/pool backward (two passes over y), weight gradient and input
    gradient of the stage above."""
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    dy, dg, db = bnrelu_pool_backward(y, st, gamma, _chk(dout, "grad_output"), ph, pw, pool=pool)
    if Cin == 1:
        dw = conv3x3_c1_wgrad(x.view(B, H, W), dy)
        dx = conv3x3_c1_dgrad(dy, w).view(B, H, W, 1) if need_dx else None
    else:
        dw = conv3x3_wgrad(x, dy)
        dx = None
        if need_dx:
            _, wd = pack_conv_weight(w, want_dgrad=True, W=W)
            dx = conv3x3(dy, wd, Cin)
    return dx, dw, dg, db


# ------------------------------------------------------------------------------------------------
# launches of CrnnEncoder (row A1') and of the text / match / loss heads
# ------------------------------------------------------------------------------------------------

def bn_act_backward(x, pre_op, st: BNStat, gamma, du, dg_out=None, db_out=None):
    C = x.shape[-1]
    rows = x.numel() // C
    dx = torch.empty_like(x)
    dg = dg_out if dg_out is not None else _empty(C, like=x)
    db = db_out if db_out is not None else _empty(C, like=x)
    ws = _ws(query("tag_bn_backward_ws_bytes", rows, C), x)
    call("tag_bn_act_backward", ptr(x), pre_op, ptr(st.mean), ptr(st.invstd), ptr(gamma), ptr(du), ptr(dx), ptr(dg),
         ptr(db), rows, C, int(st.train), ptr(ws))
    return dx, dg, db


def lppool_leaky_backward(y, dout, ph, pw, drop_p=0.0, seed=0):
    B, H, W, C = y.shape
    dy = torch.empty_like(y)
    call("tag_lppool_leaky_backward", ptr(y), ptr(dout), ptr(dy), B, H, W, C, ph, pw, float(drop_p), seed)
    return dy


def embed_mean_forward(table, text, text_len, want_tokens=True):
    """nn.Embedding gather + mean over the valid tokens (rows T1/T2): -> (seq_emb (B,D), token_emb (B,L,D) or None)."""
    tab = _chk(table, "embedding table")
    if not text.is_cuda:
        raise RuntimeError("embed_mean: token ids must live on the device (no CPU fallback)")
    B, L = text.shape
    V, D = tab.shape
    seq = _empty(B, D, like=tab)
    tok = _empty(B, L, D, like=tab) if want_tokens else None
    call("tag_embed_check_ids", ptr(text), B * L, V, ptr(_embed_flag(tab)))     # nn.Embedding raises; see check_async_errors
    call("tag_embed_mean_forward", ptr(text), ptr(text_len), ptr(tab), ptr(tok), ptr(seq), B, L, D, V)
    return seq, tok


def embed_mean_backward_into(dtab, dseq, dtok, text, text_len):
    """Adds the seq_emb / token_emb gradients into ``dtab`` (V,D) -- a zeroed tensor or the zeroed flat-gradient rows of the
    table.  Deterministic (fixed-order per-row sums, no atomics)."""
    B, L = text.shape
    V, D = dtab.shape
    if dseq is not None:
        call("tag_embed_mean_backward", ptr(_chk(dseq, "grad")), ptr(text), ptr(text_len), ptr(dtab), B, L, D, V)
    if dtok is not None:
        call("tag_embed_tokens_backward", ptr(_chk(dtok, "grad")), ptr(text), ptr(dtab), B, L, D, V)
    return dtab


def match_forward(audio, text, kind, l2norm, scale):
    """match.DotProduct (kind 0) / match.ExpNegL2 (kind 1), text_level='seq' (models/match.py:16-33,43-60): (B,T,D),(B,D) -> (B,T)."""
    a, t = _chk(audio, "audio_emb"), _chk(text, "text_emb")
    B, T, D = a.shape
    sim = _empty(B, T, like=a)
    call("tag_match_forward", ptr(a), ptr(t), ptr(sim), int(kind), int(l2norm), int(scale), B, T, D)
    return sim


def match_backward(audio, text, sim, dsim, kind, l2norm, scale):
    a, t = _chk(audio, "audio_emb"), _chk(text, "text_emb")
    B, T, D = a.shape
    da, dt = torch.empty_like(a), torch.empty_like(t)
    call("tag_match_backward", ptr(a), ptr(t), ptr(sim), ptr(_chk(dsim, "grad")), ptr(da), ptr(dt), int(kind), int(l2norm),
         int(scale), B, T, D)
    return da, dt


def frame_bce_forward(sim, label, length, Tt):
    """FrameBceLoss (losses.py:12-24) on (frame_sim[:, :Tt], label[:, :Tt], clamp(length, 1, Tt)) -> 0-dim loss."""
    s, lab = _chk(sim, "frame_sim"), _chk(label, "label")
    loss = _empty(1, like=s)
    call("tag_frame_bce_forward", ptr(s), s.shape[1], ptr(lab), lab.shape[1], ptr(length), s.shape[0], int(Tt), ptr(loss))
    return loss.view(())


def frame_bce_backward(sim, label, length, Tt, dloss):
    s, lab = _chk(sim, "frame_sim"), _chk(label, "label")
    ds = torch.empty_like(s)
    call("tag_frame_bce_backward", ptr(s), s.shape[1], ptr(lab), lab.shape[1], ptr(length), s.shape[0], int(Tt),
         ptr(_chk(dloss.reshape(1), "grad")), ptr(ds))
    return ds


def _l2norm_rows(x, rows, D):
    y = torch.empty_like(x)
    call("tag_l2norm_rows_forward", ptr(x), ptr(y), rows, D)
    return y


def align_dot_backward(audio, text, out, dout, l2norm, scaled):
    """Gradient of align.DotProduct (models/align.py:14-31): d score from (out, dout), then two MFMA GEMMs against the
    (re-normalised when l2norm) operands and the backward of F.normalize."""
    a, t = _chk(audio, "audio"), _chk(text, "text")
    B, T, D = a.shape
    N = t.shape[1]
    an, tn = (_l2norm_rows(a, B * T, D), _l2norm_rows(t, B * N, D)) if l2norm else (a, t)
    ds = _empty(B * T, B * N, like=a)
    call("tag_align_dot_dscore", ptr(out), ptr(_chk(dout, "grad")), ptr(ds), int(scaled), B, T, N, D)
    da = gemm(ds, tn.view(B * N, D), B * T, D, B * N)                       # (B*T, D)
    dt = gemm(ds, an.view(B * T, D), B * N, D, B * T, transA=True, lda=B * N)   # (B*N, D)
    if l2norm:
        da2, dt2 = torch.empty_like(da), torch.empty_like(dt)
        call("tag_l2norm_rows_backward", ptr(a), ptr(da), ptr(da2), B * T, D)
        call("tag_l2norm_rows_backward", ptr(t), ptr(dt), ptr(dt2), B * N, D)
        da, dt = da2, dt2
    return da.view(B, T, D), dt.view(B, N, D)


# ------------------------------------------------------------------------------------------------
# optimiser step on flat buffers (O1)
# ------------------------------------------------------------------------------------------------

def grad_sumsq(flat_grad):
    out = torch.empty(1, device=flat_grad.device, dtype=torch.float64)
    ws = _ws(query("tag_sumsq_ws_bytes", flat_grad.numel()), flat_grad)
    call("tag_sumsq", ptr(flat_grad), flat_grad.numel(), ptr(out), ptr(ws))
    return out


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, gnorm_sq=None, max_norm=0.0, grad_scale=1.0):
    call("tag_adam_step", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, step, ptr(gnorm_sq),
         float(max_norm), float(grad_scale))
