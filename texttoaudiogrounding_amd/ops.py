"""torch.autograd bindings over the C ABI of libtag_hip.so.

PyTorch is plumbing here: device memory (torch.empty), the current HIP stream and autograd
bookkeeping.  All arithmetic of the hot path runs in the hand-written gfx950 kernels; there is
no eager fallback -- a CPU tensor or a missing library raises.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch

from . import lib
from .lib import call, ptr, query

F32 = torch.float32


def _chk(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the MI355X (cuda) device, got {t.device}; "
                           "the HIP path has no CPU fallback")
    if t.dtype != F32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


def _empty(*shape, like: torch.Tensor, dtype=F32):
    return torch.empty(shape, device=like.device, dtype=dtype)


def _ws(nbytes: int, like: torch.Tensor):
    return torch.empty((max(int(nbytes), 16) + 7) // 8, device=like.device, dtype=torch.float64)


#: when set to a dict by bench.py, MFMA kernel launches are bracketed by HIP events recorded on the
#: launch stream: key -> list of (start_event, end_event, algorithmic_flops)
PROFILE = None


class _timed:
    def __init__(self, key, flops):
        self.key, self.flops = key, flops

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream())

    def __exit__(self, *a):
        if PROFILE is not None:
            self.e1.record(torch.cuda.current_stream())
            PROFILE.setdefault(self.key, []).append((self.e0, self.e1, self.flops))


#: data-parallel rank folded into every dropout seed (set by runner.StrongRunner): ranks seeded alike by
#: torch.manual_seed still draw different masks for their different clips
SEED_RANK = 0


def new_seed() -> int:
    """Dropout seed drawn from torch's global CPU generator (so torch.manual_seed controls it), decorrelated per rank."""
    s = int(torch.randint(0, 2 ** 62, (1,)).item())
    return (s + SEED_RANK * 0x9E3779B97F4A7C15) % (2 ** 62)


# ------------------------------------------------------------------------------------------------
# Direct gradients.  runner.FlatParams gives every trainable parameter a view of ONE flat gradient buffer
# (``p._tag_grad_sink``).  While DIRECT_GRADS is on (StrongRunner.forward_backward, after its zero_grad) the autograd
# nodes below write parameter gradients straight into those views and return None for them: no AccumulateGrad
# ``grad += new`` kernels, and a node can announce "these gradients are final" (GRAD_READY) so that the data-parallel
# all-reduce of a bucket starts while the rest of backward is still running (runner.GradBuckets).
# ------------------------------------------------------------------------------------------------
DIRECT_GRADS = False
GRAD_READY = None        # callable(list of parameters) -> None: their gradient kernels are enqueued
GRAD_FLUSH = None        # callable() -> None: a safe point to launch the all-reduce of every complete bucket


#: per training step (begin_direct_step): how many autograd nodes claimed each parameter in the forward pass, and which
#: sinks have been written in the backward pass.  A parameter seen by ONE node gets its gradient written in place; a
#: parameter shared by several nodes (a Linear applied twice, an encoder called twice) is NOT delivered directly by any of
#: them -- every contribution goes back to autograd, whose AccumulateGrad sums them into p.grad (= the same flat view) --
#: and is not announced to the gradient buckets early (GradBuckets.finish() exchanges it after backward).
_CLAIMS = {}
_WRITTEN = set()
_AUTOGRAD_SEEN = set()      # sinks (data_ptr) into which plain autograd accumulated a gradient this step (second_writer_guard)


def begin_direct_step():
    """Called by StrongRunner.forward_backward after zero_grad, before the forward pass."""
    _CLAIMS.clear()
    _WRITTEN.clear()
    _AUTOGRAD_SEEN.clear()


class _LazySinks:
    """List-like view of the flat-gradient sinks of a node's parameters, resolved when INDEXED (i.e. in backward, when the
    claim counts of the whole forward pass are known)."""

    def __init__(self, params, direct):
        self.params, self.direct = list(params), direct
        if direct:
            for t in self.params:
                if isinstance(t, torch.Tensor) and t.requires_grad and getattr(t, "_tag_grad_sink", None) is not None:
                    _CLAIMS[id(t)] = _CLAIMS.get(id(t), 0) + 1

    def _one(self, t):
        if not self.direct or not (isinstance(t, torch.Tensor) and t.requires_grad):
            return None
        sink = getattr(t, "_tag_grad_sink", None)
        return sink if (sink is not None and _CLAIMS.get(id(t), 0) == 1) else None

    def __len__(self):
        return len(self.params)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._one(t) for t in self.params[i]]
        return self._one(self.params[i])


#: True while the outermost tag autograd node being applied records a graph (set by TagFunction.apply): inside
#: Function.forward grad mode is always off and ctx.needs_input_grad only mirrors requires_grad, so this is the one place
#: the caller's ``torch.no_grad()`` is visible
_RECORDING = True


class TagFunction(torch.autograd.Function):
    """Base of the autograd nodes of this module: remembers whether the CALLER records a graph."""

    @classmethod
    def apply(cls, *args, **kwargs):
        global _RECORDING
        prev, _RECORDING = _RECORDING, torch.is_grad_enabled()
        try:
            return super().apply(*args, **kwargs)
        finally:
            _RECORDING = prev


def _sinks(params):
    """Per input parameter: its flat-gradient view, or None (frozen parameter / direct gradients off / a parameter claimed
    by more than one node in this forward pass).  A forward pass that records no graph (``torch.no_grad()``, the first pass of
    a checkpointed segment) claims nothing -- its node never runs backward, and a claim would silently switch the
    parameter's real node to the AccumulateGrad route and disable the early bucket launch."""
    return _LazySinks(params, DIRECT_GRADS and _RECORDING)


def second_writer_guard(p):
    """Tensor hook of every flat-buffer parameter (runner.FlatParams): it sees the gradient autograd is about to accumulate
    into ``p.grad`` -- None when every node delivered in place.  A parameter claimed by exactly ONE HIP node is delivered in
    place, so a defined gradient arriving for it means a second, plain-torch consumer of the same parameter (a tied weight,
    a regulariser on p) is adding into the very view the node overwrites with copy_: the sum would depend on the order of
    the two writes.  Raise instead of training on a silently wrong gradient.

    The race is decided on what HAPPENED in this step, not on the claim alone: the error is raised when both writers really
    wrote -- here if the node's in-place delivery came first (the sink is in _WRITTEN), in _deliver if autograd's came first.  A
    parameter claimed by a HIP node whose output never takes part in this backward (a metric-only forward under grad mode) and
    also used by a plain torch op has ONE writer and trains normally."""
    def hook(g):
        if g is None or not DIRECT_GRADS or _CLAIMS.get(id(p), 0) != 1:
            return
        sink = getattr(p, "_tag_grad_sink", None)
        if sink is None:
            return
        if sink.data_ptr() in _WRITTEN:
            raise RuntimeError(_SECOND_WRITER_MSG)
        _AUTOGRAD_SEEN.add(sink.data_ptr())
    return hook


_SECOND_WRITER_MSG = ("direct gradients: a parameter delivered in place by a HIP autograd node also received a gradient through "
                      "plain autograd in the same step (tied weight / regulariser on the parameter); the two writers race on one "
                      "flat-gradient view -- run this model with ops.DIRECT_GRADS off")


def _deliver(grads, sinks, i, val):
    """Gradient ``val`` of input i: copied into its sink (the node then returns None) or returned to autograd."""
    sink = sinks[i]
    if sink is not None:
        key = sink.data_ptr()
        if key in _WRITTEN:
            raise RuntimeError("direct gradients: a flat-gradient sink was written twice in one step (a retained graph run "
                               "twice?); plain autograd would have accumulated -- run this pattern with ops.DIRECT_GRADS off")
        if key in _AUTOGRAD_SEEN:
            raise RuntimeError(_SECOND_WRITER_MSG)
        _WRITTEN.add(key)
        if val.data_ptr() != key:
            sink.copy_(val.view_as(sink))
        grads[i] = None
    else:
        grads[i] = val


def _ready(params):
    if GRAD_READY is not None and params:
        GRAD_READY([t for t in params if isinstance(t, torch.Tensor) and _CLAIMS.get(id(t), 0) == 1])


def _flush():
    if GRAD_FLUSH is not None:
        GRAD_FLUSH()


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


# Arithmetic of the 3x3 convolutions (forward, dgrad, wgrad): "fp32" = exact fp32 MFMA (default); opt-in, on the bf16
# MFMA with fp32 accumulation (conv_x3.hip): "x3" = fp32 operands split exactly into 3 bf16 terms, 6 partial products;
# "x9" = all 9 partial products; "bf16" = operands rounded to bf16, one product (BASELINE configs[2] arithmetic).
CONV_MATH = os.environ.get("TAG_CONV_MATH", "fp32")
_X3_PRODUCTS = {"x3": 6, "x9": 9, "bf16": 1}
# Storage of the big activations of the conv stack (raw conv outputs, pooled block outputs and their gradients):
# "fp32" (default) or "bf16" = BASELINE configs[2] proper -- bf16 tensors in HBM, fp32 accumulation / BatchNorm statistics /
# GRU / heads / loss / master weights.  Only meaningful with CONV_MATH == "bf16" (the one-product bf16 MFMA kernels);
# with any other conv arithmetic the setting is ignored.
ACT_DTYPE = os.environ.get("TAG_ACT_DTYPE", "fp32")
BF16 = torch.bfloat16


#: BASELINE configs[2] mode only: GEMM operands (nn.Linear fc1 / projections, GRU input projections and their backward GEMMs)
#: rounded to bf16 on the bf16 MFMA with fp32 accumulation, as autocast would; "auto" = on exactly when ACT_DTYPE is bf16
GEMM_MATH = os.environ.get("TAG_GEMM_MATH", "auto")


def gemm_bf16() -> bool:
    return GEMM_MATH == "bf16" or (GEMM_MATH == "auto" and ACT_DTYPE == "bf16")


def act_bf16() -> bool:
    return CONV_MATH == "bf16" and ACT_DTYPE == "bf16"


def _sfx(t) -> str:
    """Entry-point suffix for an activation tensor: '' (fp32) or '_bf16'."""
    return "_bf16" if t.dtype == BF16 else ""


def _x3_ok(W, K, N):
    return CONV_MATH in _X3_PRODUCTS and W in (8, 16, 32, 64) and K % 32 == 0 and N % 64 == 0


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
        npr = _X3_PRODUCTS[CONV_MATH]
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


FUSE_BN_STATS = os.environ.get("TAG_FUSE_BN_STATS", "1") != "0"

#: Winograd F(2x2,3x3) form of the 3x3 convolutions, all fp32 (csrc/conv_wino_fused.hip: ONE kernel per launch, the transforms inside
#: the product kernel; round 5's plane form, csrc/conv_wino.hip, remains for channel counts the fused kernels do not take): forward
#: (training: + BatchNorm statistics; inference: + BatchNorm / ReLU / pool), dgrad (+ BatchNorm-backward or pool-backward sums) and
#: weight gradient.  2.25 x fewer MFMA FLOP than the direct halo-tile kernels; since round 6 faster on EVERY layer with >= 64
#: channels on both sides (tools/wino_bench.py, B = 64: x1.5 ... x1.9 per launch).  "0" = direct kernels only.
CONV_WINOGRAD = os.environ.get("TAG_CONV_WINOGRAD", "1") != "0"
#: channel rule: the smaller count >= WINO_MIN_C and the larger >= WINO_MIN_CMAX
WINO_MIN_C = int(os.environ.get("TAG_WINO_MIN_C", "64"))
WINO_MIN_CMAX = int(os.environ.get("TAG_WINO_MIN_CMAX", "64"))
#: ... and only training launches of at least this much work, tiles x output channels (tiles = B * ceil(H/2) * ceil(W/2); 2^20 = one
#: 64-tile x 64-cout workgroup of the fused kernel per CU): smaller launches cannot fill the chip with those blocks and keep the
#: direct kernel.  (At B = 64 every layer is 30 ... 120 times above it; the 2-clip fixtures of the parity tests are below it and
#: are ALSO run with the rule forced to 1 and the step's decisions imposed on the oracle: tests/test_gpu_path.py.)
WINO_MIN_WORK = int(os.environ.get("TAG_WINO_MIN_WORK", str(1 << 20)))
#: the inference forward (BatchNorm in eval mode, nothing saved) of the same layers as Winograd too, at EVERY launch size: the choice
#: must not depend on the batch, or the same clip would score differently in a 4-clip and in a 64-clip pass (the forward is
#: batch-invariant, tests/test_gpu_infer.py)
CONV_WINOGRAD_EVAL = os.environ.get("TAG_CONV_WINOGRAD_EVAL", "1") != "0"
#: the fused kernels address a tensor through a buffer descriptor (32-bit byte offsets): launches without per-batch sums are cut into
#: batch slices below this many bytes per tensor (any cut gives the same rows: every tile is computed independently of the others)
WINO_MAX_BYTES = int(os.environ.get("TAG_WINO_MAX_BYTES", str((1 << 31) - (1 << 20))))
#: launches that took the Winograd path since import (tests assert that the benched-size step really runs through it)
WINO_LAUNCHES = 0


def _wino_shape(W, Cin, Cout) -> bool:
    return (CONV_WINOGRAD and CONV_MATH == "fp32" and W in (8, 16, 32, 64) and min(Cin, Cout) >= WINO_MIN_C
            and max(Cin, Cout) >= WINO_MIN_CMAX)


def _wino_flop(B, H, W, Cin, Cout) -> float:
    """FLOP a Winograd launch EXECUTES on the matrix pipe (16 products of T x Cin x Cout; bench.py's roofline counts these, not the
    2.25 x larger direct-convolution figure)."""
    return 2.0 * 16 * B * ((H + 1) // 2) * ((W + 1) // 2) * Cin * Cout


def _wino_u(wpack, x, Cout, count=True, any_size=False):
    """The Winograd-domain weights riding on a direct pack (pack_conv_weight) when this launch may use them, else None.
    any_size: the inference forward -- no tile threshold (see CONV_WINOGRAD_EVAL)."""
    u = getattr(wpack, "wino_u", None)
    if u is None or not CONV_WINOGRAD or CONV_MATH != "fp32" or x.dtype != F32:
        return None
    B, H, W, Cin = x.shape
    if any_size and not CONV_WINOGRAD_EVAL:
        return None
    if ((not any_size and B * ((H + 1) // 2) * ((W + 1) // 2) * Cout < WINO_MIN_WORK)
            or not query("tag_conv3x3_wino_ok", 1 if any_size else B, H, W, Cin, Cout)):     # (inference launches are cut by batch)
        return None
    if count:
        global WINO_LAUNCHES
        WINO_LAUNCHES += 1
    return u


def conv3x3(x, wpack, Cout, prologue=0, scale=None, shift=None, training_launch=False):
    """y = conv(prologue(x)).  training_launch: a launch of the training step (a dgrad conv): may take the Winograd form."""
    return conv3x3_stats(x, wpack, Cout, prologue, scale, shift, want_stats=False, training_launch=training_launch)[0]


def _batch_chunks(B, bytes_per_clip):
    """Batch slices [b0, b1) whose tensors stay under WINO_MAX_BYTES (the fused Winograd kernels' descriptor range)."""
    nb = max(1, min(B, WINO_MAX_BYTES // max(1, bytes_per_clip)))
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
    elif ((want_stats and FUSE_BN_STATS) or training_launch) and not x3:
        u = _wino_u(wpack, x, Cout)
    if u is not None:
        if want_stats and FUSE_BN_STATS:
            P = query("tag_conv3x3_wino_stats_rows", B, H, W, Cout)
            part = (P, _empty(P * (3 * Cout + 1), like=x))
        ws = _ws(query("tag_conv3x3_wino_ws_bytes", B, H, W, Cin, Cout), x)
        with _timed(("conv3x3_wino", B, H, W, Cin, Cout), _wino_flop(B, H, W, Cin, Cout)):
            call("tag_conv3x3_wino_forward", ptr(x), ptr(u), prologue, ptr(scale), ptr(shift), ptr(y), ptr(part[1]) if part else None,
                 B, H, W, Cin, Cout, ptr(ws), None)
        return y, part
    if want_stats and FUSE_BN_STATS:
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


#: BatchNorm-backward sums in the dgrad conv epilogue (tag_conv3x3_dgrad_bnsums) instead of a separate two-tensor pass
FUSE_BN_BWD_SUMS = os.environ.get("TAG_FUSE_BN_BWD", "1") != "0"
#: block 1: bn1's backward applied inside the Cin = 1 conv backward (tag_conv3x3_c1_backward_bnrelu) instead of a separate pass
FUSE_C1_BN_BWD = os.environ.get("TAG_FUSE_C1_BN_BWD", "1") != "0"


def conv3x3_dgrad_bnrelu_backward(dy_in, wpack, yref, st: BNStat, gamma, dg_out=None, db_out=None, after_conv=None,
                                  defer_apply=False):
    """The dgrad convolution da = conv(dy_in, wpack) followed by the backward of relu(bn(yref)):
    returns (dy_ref, dgamma, dbeta) with dy_ref = dL/d yref (written in place over da).  Exact-fp32 halo-tile shapes
    fold the per-channel sums into the conv epilogue; other shapes / arithmetics run the conv and tag_bnrelu_backward.
    defer_apply: a 4-tuple (t, dgamma, dbeta, applied) comes back; on the fused paths applied is False and t is still da --
    the caller's next kernel applies the BatchNorm + ReLU backward itself (conv3x3_c1_backward(bn_bwd=...))."""
    B, H, W, Cin = dy_in.shape
    C = yref.shape[3]
    fused = (FUSE_BN_BWD_SUMS and wpack.dtype != torch.uint8 and st.train and W in (8, 16, 32, 64)
             and query("tag_conv3x3_stats_rows", B, H, W, C) > 0)
    if (FUSE_BN_BWD_SUMS and dy_in.dtype == BF16 and wpack.dtype == torch.uint8 and wpack.products == 1 and st.train
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


#: the reduction half of the pool backward in the epilogue of the dgrad conv that PRODUCES the pooled gradient
#: (tag_conv3x3_dgrad_poolsums) instead of a pass of its own over the largest tensors (pool_bwd_reduce_kernel)
FUSE_POOL_BWD_SUMS = os.environ.get("TAG_FUSE_POOL_BWD", "1") != "0"
#: the same for the bf16-storage kernels (tag_conv3x3_dgrad_poolsums_bf16): built and tested, OFF by default -- the bf16 convs are
#: HBM / power-bound, the epilogue's window reads are not hidden there and the step time is level (10.62 vs 10.62-10.70 ms) while the
#: conv family's own time grows by what the removed pass cost (docs/experiments_r05.md)
FUSE_POOL_BWD_SUMS_BF16 = os.environ.get("TAG_FUSE_POOL_BWD_BF16", "0") != "0"


def pool_sums_fusable(dy_in, wpack, yref, ph, pw):
    """Can the dgrad conv of (dy_in, wpack) carry the pool-backward sums of the block below (raw output yref, window ph x pw)?
    Exact-fp32 halo-tile shapes, windows 1x2 / 2x2."""
    B, H, W, Cin = dy_in.shape
    if not (FUSE_POOL_BWD_SUMS and W in (8, 16, 32, 64) and pw == 2 and ph in (1, 2) and H == yref.shape[1] // ph
            and W == yref.shape[2] // pw):
        return False
    if dy_in.dtype == BF16:       # BASELINE configs[2] mode: the one-product bf16 tile kernel with the staged output tile
        return (FUSE_POOL_BWD_SUMS_BF16 and yref.dtype == BF16 and wpack.dtype == torch.uint8 and getattr(wpack, "products", 0) == 1
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


#: inference: conv2 of a ConvBlock writes the POOLED relu(bn(.)) straight from its output tile (tag_conv3x3_forward_bnrelu_pool_eval)
FUSE_EVAL_POOL = os.environ.get("TAG_FUSE_EVAL_POOL", "1") != "0"


def eval_pool_fusable(x, wpack, ph, pw, pool=0):
    """Exact-fp32 halo-tile shapes, windows 1x2 / 2x2, the three pool types."""
    B, H, W, _ = x.shape
    return (FUSE_EVAL_POOL and x.dtype == F32 and wpack.dtype != torch.uint8 and W in (8, 16, 32, 64) and pw == 2 and ph in (1, 2)
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
    if CONV_MATH in _X3_PRODUCTS and W in (8, 16, 32, 64) and Cin % 64 == 0 and Cout % 64 == 0:
        ws = _ws(query("tag_conv3x3_wgrad_x3_ws_bytes", B, H, W, Cin, Cout), x)
        with _timed(("conv3x3_wgrad_x3_kernel", B, H, W, Cin, Cout), 2.0 * B * H * W * 9 * Cin * Cout):
            call("tag_conv3x3_wgrad_x3", ptr(x), prologue, ptr(scale), ptr(shift), ptr(dy), ptr(dw), B, H, W, Cin, Cout,
                 _X3_PRODUCTS[CONV_MATH], ptr(ws))
        return dw
    if (x.dtype == F32 and dy.dtype == F32 and _wino_shape(W, Cin, Cout)
            and B * ((H + 1) // 2) * ((W + 1) // 2) * max(Cin, Cout) >= WINO_MIN_WORK and query("tag_conv3x3_wino_ok", B, H, W, Cin, Cout)):
        global WINO_LAUNCHES
        WINO_LAUNCHES += 1
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
    P = query("tag_conv3x3_c1_stats_rows", B, H, W, Cout) if (want_stats and FUSE_BN_STATS) else 0
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
# Cnn8Rnn: the whole audio encoder as one autograd node (rows F1-F3, A1-A4 forward + backward)
# ------------------------------------------------------------------------------------------------

CNN8_POOLS = [(2, 2), (2, 2), (1, 2), (1, 2)]

#: weight-gradient convolutions are off the critical path of backward (only the optimiser needs them): on a second HIP stream
#: their workgroups fill the tails / small-grid gaps of the dgrad + BatchNorm chain.  TAG_WGRAD_STREAM = 1 / 0 forces it on / off;
#: the default ("auto", None here) is ON for the arithmetics on the bf16 MFMA (bf16 mode, x3, x9) and OFF for exact fp32: with
#: round 4's halo / all-taps kernels the fp32 step IS the sum of its kernels' isolated times and co-running two of them only
#: stretches both (same box, alternating processes: 54.82 / 54.92 ms with the side stream, 54.18 / 54.15 ms without), while the
#: bf16-MFMA modes' shorter kernels still gain (bf16 10.52-10.53 against 10.70-10.78 ms, x3 34.6 against 35.3, x9 44.8 against 45.1).
import os as _os
_side_env = _os.environ.get("TAG_WGRAD_STREAM", "auto")
WGRAD_SIDE_STREAM = None if _side_env == "auto" else (_side_env != "0")
_side_streams = {}


def side_stream_enabled():
    return WGRAD_SIDE_STREAM if WGRAD_SIDE_STREAM is not None else (CONV_MATH != "fp32")


#: TAG_WGRAD_CU_SKIP=k (k >= 2): the side stream may not use every k-th compute unit (hipExtStreamCreateWithCUMask), so that the
#: short kernels of the main stream (BatchNorm finalizes, reductions) never queue behind a full residency round of
#: weight-gradient workgroups.  0 = an ordinary stream.
def _env_int(name, default=0):
    """An integer environment switch parsed defensively: a malformed value is reported with its name, not as a bare ValueError at
    import time."""
    raw = _os.environ.get(name, "")
    if raw.strip() == "":
        return default
    try:
        return int(raw)
    except ValueError:
        raise RuntimeError(f"environment variable {name}={raw!r} must be an integer") from None


WGRAD_CU_SKIP = _env_int("TAG_WGRAD_CU_SKIP", 0)


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _side_streams:
        st = None
        if WGRAD_CU_SKIP >= 2:
            import ctypes
            ncu = query("tag_device_cu_count")
            words = (ncu + 31) // 32
            mask = (ctypes.c_uint32 * words)()
            for i in range(ncu):
                if i % WGRAD_CU_SKIP != WGRAD_CU_SKIP - 1:
                    mask[i // 32] |= 1 << (i % 32)
            out = ctypes.c_void_p()
            with torch.cuda.device(device):
                rc = lib.load().tag_stream_create_cu_mask(mask, words, ctypes.byref(out))
            if rc != 0:
                raise RuntimeError(f"tag_stream_create_cu_mask failed: {lib.load().tag_last_error().decode()}")
            st = torch.cuda.ExternalStream(out.value, device=device)
        _side_streams[key] = st if st is not None else torch.cuda.Stream(device=device)
    return _side_streams[key]


def side_streams(device):
    """Side streams this process has used on ``device`` (the gradient all-reduce must wait for their wgrad kernels)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    return [_side_streams[key]] if key in _side_streams else []


#: the side stream's wgrad of a layer is RELEASED one kernel late -- when the dgrad conv that consumes the same dy has been
#: enqueued -- so that it starts together with the HBM-bound BatchNorm / pool backward passes that follow that dgrad
#: instead of beside the dgrad itself (two MFMA-bound kernels of equal length co-running finish together and leave the
#: bandwidth-bound passes alone on the chip; lagged, every such pass has MFMA work to hide under).
WGRAD_LAG = _os.environ.get("TAG_WGRAD_LAG", "0") != "0"     # measured: 57.7 ms lagged vs 57.2 ms not -- off by default


#: parameter-gradient work of the GRU (1) -- and of fc1 (2) -- on the wgrad side stream; 0 = on the main stream
#: (TAG_SIDE_PARAM_GRADS).  Measured on one box, B = 64 (fp32 / bf16 mode ms per step): 0: 54.64-54.75 / 11.23-11.30,
#: 1: 54.57-54.69 / 11.02-11.11, 2: 54.91-55.20 / 11.08-11.13 -- fc1's GEMM on the side stream delays the block-4 wgrads more
#: than it overlaps, so the default stops at the GRU.
SIDE_PARAM_GRADS = _env_int("TAG_SIDE_PARAM_GRADS", 1)


#: TAG_SIDE_RECORD_STREAM=1 restores round 4's lifetime rule for the side stream's operands (Tensor.record_stream instead of keeping
#: them alive until join()) -- kept for the A/B that shows the allocator growth it causes under an unsynchronised host.
SIDE_RECORD_STREAM = _env_int("TAG_SIDE_RECORD_STREAM", 0) != 0


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
        if not WGRAD_LAG:
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
                if SIDE_RECORD_STREAM:
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
                             seeds=seeds, sinks=_sinks(params), params=params if DIRECT_GRADS else None)
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
        dfc, ggru = gru_bidir_backward(dy, fc, sv["gsave"], outs=sk[28:36], side=sw if SIDE_PARAM_GRADS >= 1 else None)
        for k in range(8):
            _deliver(grads, sk, 28 + k, ggru[k])
        M = fc.shape[0]
        dfc = relu_backward(fc, dfc)
        xm = sv["xm"]
        fc_w = p[26]
        if SIDE_PARAM_GRADS >= 2 and sk[26] is not None and sk[27] is not None:
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
            defer = i == 0 and FUSE_C1_BN_BWD and (y1.shape[2], y1.shape[3]) == C1_BWD_FUSED_SHAPE
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
# One conv3x3 -> BatchNorm -> ReLU (-> pool) stage as a standalone operator: what SURVEY.md section 8(b) lists as
# ``conv3x3_bn_relu[_pool]`` and what ConvBlock.forward (models/panns.py:46-62) is made of.  The fused Cnn8Rnn engine above
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
# CrnnEncoder (row A1'): cdur_block = BN -> conv3x3 -> LeakyReLU(0.1), LPPool2d(4), Dropout(0.3), BiGRU(128)
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
                             params=params if DIRECT_GRADS else None)
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
        ctx.params = [w, b] if DIRECT_GRADS else None
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
        ctx.params = [w_h, b_h, v] if DIRECT_GRADS else None
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
        ctx.params = [w_u, b_u, w_s, b_s] if DIRECT_GRADS else None
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
        ctx.params = list(params) if DIRECT_GRADS else None
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
        ctx.params = [w, b] if DIRECT_GRADS else None
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
