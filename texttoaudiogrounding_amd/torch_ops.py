"""``torch.library`` registration of the hot-path kernels (namespace ``tag``) -- the op list SURVEY.md section 8(b) asks a
native replacement to export, with schemas, fake (meta) kernels for shape inference / tracing and autograd formulas, on top
of the SAME C ABI (libtag_hip.so through ctypes, texttoaudiogrounding_amd.lib): PyTorch sees these as first-class operators
(``torch.ops.tag.logmel`` ...).  They are the FRONT door: the reference-shaped modules call them (models/match.py ->
``frame_match``, models/align.py -> ``align_dot``, losses.py -> ``frame_bce``, models/text_encoder.py -> ``embed_mean``,
models/panns.py ConvBlock.forward -> ``conv3x3_bn_relu_pool``, utils/eval_util.py -> ``segments``).  Every formula exists
once: an operator's forward / backward body is a plain function of ``ops.py`` (``ops.match_forward`` / ``ops.match_backward``
...).  The fused encoders are operators too (``tag::cnn8rnn_encoder`` / ``tag::crnn_encoder``, round 4): they wrap the engine
of ``ops.py`` (``ops.Cnn8RnnFunction`` / ``ops.CrnnFunction`` never materialise the intermediate activations and write
parameter gradients straight into the flat buffer -- per-step state that travels beside the operator, see below);
``ops.EmbedMeanFunction`` (the direct-gradient scatter) is the one autograd node the modules still apply directly.

    import texttoaudiogrounding_amd.torch_ops          # registers torch.ops.tag.*

Every op takes contiguous fp32 device tensors on the current stream and raises on CPU tensors (no fallback).  Backward passes
are ops themselves (``tag::*_backward``), so a graph captured through these operators contains only ``tag::`` nodes for the
path.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import Tensor
from torch.library import custom_op, register_autograd

from . import engine, ops

__all__ = ["OP_NAMES"]


# ------------------------------------------------------------------------------------------------ F1/F2 log-mel
@custom_op("tag::logmel", mutates_args=())
def logmel(waveform: Tensor, n_fft: int, win_length: int, hop: int, window: Tensor, fb: Tensor) -> Tensor:
    """(B,S) -> (B, S//hop + 1, n_mels) dB log-mel, time-major (rows F1/F2)."""
    return ops.logmel(waveform, n_fft, win_length, hop, window, fb)


@logmel.register_fake
def _(waveform, n_fft, win_length, hop, window, fb):
    return waveform.new_empty(waveform.shape[0], waveform.shape[1] // hop + 1, fb.shape[1])


# ------------------------------------------------------------------------------------------------ A1 conv 3x3
@custom_op("tag::conv3x3", mutates_args=())
def conv3x3(x: Tensor, weight: Tensor, prologue: int, scale: Optional[Tensor], shift: Optional[Tensor]) -> Tensor:
    """y = conv3x3(prologue(x)), channels-last (B,H,W,Cin) x (Cout,Cin,3,3) -> (B,H,W,Cout); prologue 1 folds the producer's
    BatchNorm + ReLU (relu(x * scale + shift)) into the operand load."""
    wf, _ = ops.pack_conv_weight(weight, want_dgrad=False, W=x.shape[2])
    return ops.conv3x3(x, wf, weight.shape[0], prologue, scale, shift)


@conv3x3.register_fake
def _(x, weight, prologue, scale, shift):
    return x.new_empty(x.shape[0], x.shape[1], x.shape[2], weight.shape[0])


@custom_op("tag::conv3x3_dgrad", mutates_args=())
def conv3x3_dgrad(dy: Tensor, weight: Tensor) -> Tensor:
    """Gradient of conv3x3 (prologue 0) with respect to its input: (B,H,W,Cout) -> (B,H,W,Cin)."""
    _, wd = ops.pack_conv_weight(weight, want_dgrad=True, W=dy.shape[2])
    return ops.conv3x3(dy, wd, weight.shape[1])


@conv3x3_dgrad.register_fake
def _(dy, weight):
    return dy.new_empty(dy.shape[0], dy.shape[1], dy.shape[2], weight.shape[1])


@custom_op("tag::conv3x3_wgrad", mutates_args=())
def conv3x3_wgrad(x: Tensor, dy: Tensor, prologue: int, scale: Optional[Tensor], shift: Optional[Tensor]) -> Tensor:
    """Weight gradient (Cout,Cin,3,3) of y = conv3x3(prologue(x))."""
    return ops.conv3x3_wgrad(x, dy, prologue, scale, shift)


@conv3x3_wgrad.register_fake
def _(x, dy, prologue, scale, shift):
    return x.new_empty(dy.shape[3], x.shape[3], 3, 3)


def _conv_setup(ctx, inputs, output):
    x, weight, prologue, scale, shift = inputs
    if prologue != 0 and (x.requires_grad or (scale is not None and scale.requires_grad)):
        raise RuntimeError("tag::conv3x3: the autograd formula covers prologue = 0 (the fused BatchNorm prologue is "
                           "differentiated by ops.Cnn8RnnFunction, which owns the batch statistics)")
    ctx.save_for_backward(x, weight)
    ctx.pro = (prologue, scale, shift)


def _conv_backward(ctx, dy):
    x, weight = ctx.saved_tensors
    prologue, scale, shift = ctx.pro
    dy = dy.contiguous()
    dx = torch.ops.tag.conv3x3_dgrad(dy, weight) if ctx.needs_input_grad[0] else None
    dw = torch.ops.tag.conv3x3_wgrad(x, dy, prologue, scale, shift) if ctx.needs_input_grad[1] else None
    return dx, dw, None, None, None


register_autograd("tag::conv3x3", _conv_backward, setup_context=_conv_setup)


# ------------------------------------------------------------------------------------------------ A1 conv3x3 -> BN -> ReLU (-> pool)
@custom_op("tag::conv3x3_bn_relu_pool", mutates_args=())
def conv3x3_bn_relu_pool(x: Tensor, weight: Tensor, gamma: Tensor, beta: Tensor, running_mean: Tensor, running_var: Tensor,
                         training: bool, momentum: float, eps: float, ph: int, pw: int,
                         pool: int) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """One stage of ConvBlock.forward (models/panns.py:46-62), channels-last: relu(bn(conv3x3(x))) then pooling with kernel =
    stride = (ph, pw) (1,1 = none); pool 0 'avg+max' | 2 'avg' | 3 'max'.  x (B,H,W,Cin) with Cin = 1 or a multiple of 32,
    weight (Cout,Cin,3,3).  Functional (an operator with an autograd formula may not mutate its inputs): train mode uses the
    batch statistics and RETURNS the updated running statistics, which the caller copies into the BatchNorm buffers.
    -> (out (B,H/ph,W/pw,Cout), y = raw conv output, mean, invstd, scale, shift, new_running_mean, new_running_var)."""
    rm, rv = running_mean.clone(), running_var.clone()
    out, y, st = ops.conv_bn_relu_pool_forward(x, weight, gamma, beta, rm, rv, training, momentum, eps, ph, pw, pool)
    mean = st.mean.clone() if not training else st.mean           # eval: st.mean IS rm (no aliased outputs)
    return out, y, mean, st.invstd, st.scale, st.shift, rm, rv


@conv3x3_bn_relu_pool.register_fake
def _(x, weight, gamma, beta, running_mean, running_var, training, momentum, eps, ph, pw, pool):
    B, H, W, _ = x.shape
    C = weight.shape[0]
    v = lambda: x.new_empty(C)
    return x.new_empty(B, H // ph, W // pw, C), x.new_empty(B, H, W, C), v(), v(), v(), v(), v(), v()


@custom_op("tag::conv3x3_bn_relu_pool_backward", mutates_args=())
def conv3x3_bn_relu_pool_backward(dout: Tensor, x: Tensor, weight: Tensor, y: Tensor, mean: Tensor, invstd: Tensor, scale: Tensor,
                                  shift: Tensor, gamma: Tensor, training: bool, ph: int, pw: int, pool: int,
                                  need_dx: bool) -> List[Tensor]:
    """-> [dx (empty (0,) tensor when not needed), dweight, dgamma, dbeta]."""
    st = ops.BNStat()
    st.mean, st.invstd, st.scale, st.shift, st.train = mean, invstd, scale, shift, bool(training)
    dx, dw, dg, db = ops.conv_bn_relu_pool_backward(dout, x, weight, y, st, gamma, ph, pw, pool, need_dx)
    return [dx if dx is not None else x.new_empty(0), dw, dg, db]


@conv3x3_bn_relu_pool_backward.register_fake
def _(dout, x, weight, y, mean, invstd, scale, shift, gamma, training, ph, pw, pool, need_dx):
    return [torch.empty_like(x) if need_dx else x.new_empty(0), torch.empty_like(weight), torch.empty_like(gamma),
            torch.empty_like(gamma)]


def _cbrp_setup(ctx, inputs, output):
    x, weight, gamma, _beta, _rm, _rv, training, _mom, _eps, ph, pw, pool = inputs
    _out, y, mean, invstd, scale, shift, _rm, _rv = output
    ctx.save_for_backward(x, weight, y, mean, invstd, scale, shift, gamma)
    ctx.cfg = (training, ph, pw, pool)


def _cbrp_backward(ctx, dout, *_unused):
    x, weight, y, mean, invstd, scale, shift, gamma = ctx.saved_tensors
    training, ph, pw, pool = ctx.cfg
    dx, dw, dg, db = torch.ops.tag.conv3x3_bn_relu_pool_backward(dout.contiguous(), x, weight, y, mean, invstd, scale, shift, gamma,
                                                                 training, ph, pw, pool, ctx.needs_input_grad[0])
    return (dx if ctx.needs_input_grad[0] else None, dw, dg, db, None, None, None, None, None, None, None, None)


register_autograd("tag::conv3x3_bn_relu_pool", _cbrp_backward, setup_context=_cbrp_setup)


# ------------------------------------------------------------------------------------------------ A4 BiGRU
@custom_op("tag::gru_bidir", mutates_args=())
def gru_bidir(x: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor, w_ih_r: Tensor, w_hh_r: Tensor,
              b_ih_r: Tensor, b_hh_r: Tensor) -> Tuple[Tensor, Tensor]:
    """nn.GRU(I, H, bidirectional=True, batch_first=True), h0 = 0: x (B,T,I) -> y (B,T,2H) and the saved gates (B,T,2,4H)."""
    B, T, I = x.shape
    y, sv = ops.gru_bidir_forward(x.reshape(B * T, I), [w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r], B, T, True)
    return y, sv["gates"]


@gru_bidir.register_fake
def _(x, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
    B, T, _ = x.shape
    H = w_hh.shape[1]
    return x.new_empty(B, T, 2 * H), x.new_empty(B, T, 2, 4 * H)


@custom_op("tag::gru_bidir_backward", mutates_args=())
def gru_bidir_backward(dy: Tensor, x: Tensor, y: Tensor, gates: Tensor, w_ih: Tensor, w_hh: Tensor, w_ih_r: Tensor,
                       w_hh_r: Tensor) -> List[Tensor]:
    """-> [dx, dw_ih, dw_hh, db_ih, db_hh, dw_ih_r, dw_hh_r, db_ih_r, db_hh_r]."""
    B, T, I = x.shape
    H = w_hh.shape[1]
    sv = dict(Hh=H, y=y, gates=gates, w_ih=torch.cat([w_ih, w_ih_r], 0), w_hh=torch.stack([w_hh, w_hh_r], 0).contiguous())
    dx, g = ops.gru_bidir_backward(dy.contiguous(), x.reshape(B * T, I), sv)
    return [dx.view(B, T, I)] + [t.clone() for t in g]          # bias slices share storage: operators may not return aliases


@gru_bidir_backward.register_fake
def _(dy, x, y, gates, w_ih, w_hh, w_ih_r, w_hh_r):
    H = w_hh.shape[1]
    b = x.new_empty(3 * H)
    return [torch.empty_like(x), torch.empty_like(w_ih), torch.empty_like(w_hh), b, torch.empty_like(b),
            torch.empty_like(w_ih_r), torch.empty_like(w_hh_r), torch.empty_like(b), torch.empty_like(b)]


def _gru_setup(ctx, inputs, output):
    x, w_ih, w_hh, _, _, w_ih_r, w_hh_r, _, _ = inputs
    ctx.save_for_backward(x, output[0], output[1], w_ih, w_hh, w_ih_r, w_hh_r)


def _gru_backward(ctx, dy, _dgates):
    x, y, gates, w_ih, w_hh, w_ih_r, w_hh_r = ctx.saved_tensors
    g = torch.ops.tag.gru_bidir_backward(dy, x, y, gates, w_ih, w_hh, w_ih_r, w_hh_r)
    return tuple(g)


register_autograd("tag::gru_bidir", _gru_backward, setup_context=_gru_setup)


# ------------------------------------------------------------------------------------------------ T1/T2 embedding + mean
@custom_op("tag::embed_mean", mutates_args=())
def embed_mean(table: Tensor, text: Tensor, text_len: Tensor) -> Tuple[Tensor, Tensor]:
    """nn.Embedding gather + mean over the valid tokens: -> seq_emb (B,D), token_emb (B,L,D)."""
    return ops.embed_mean_forward(table, text, text_len, True)


@embed_mean.register_fake
def _(table, text, text_len):
    B, L = text.shape
    return table.new_empty(B, table.shape[1]), table.new_empty(B, L, table.shape[1])


@custom_op("tag::embed_mean_backward", mutates_args=())
def embed_mean_backward(dseq: Optional[Tensor], dtok: Optional[Tensor], text: Tensor, text_len: Tensor, V: int, D: int) -> Tensor:
    """Deterministic scatter of the seq_emb / token_emb gradients into a zeroed (V,D) table gradient."""
    return ops.embed_mean_backward_into(torch.zeros(V, D, device=text.device, dtype=torch.float32), dseq, dtok, text, text_len)


@embed_mean_backward.register_fake
def _(dseq, dtok, text, text_len, V, D):
    return torch.empty(V, D, device=text.device, dtype=torch.float32)


def _embed_setup(ctx, inputs, output):
    table, text, text_len = inputs
    ctx.save_for_backward(text, text_len)
    ctx.vd = tuple(table.shape)


def _embed_backward(ctx, dseq, dtok):
    text, text_len = ctx.saved_tensors
    return torch.ops.tag.embed_mean_backward(dseq, dtok, text, text_len, ctx.vd[0], ctx.vd[1]), None, None


register_autograd("tag::embed_mean", _embed_backward, setup_context=_embed_setup)


# ------------------------------------------------------------------------------------------------ M1/M2 frame x phrase heads
@custom_op("tag::frame_match", mutates_args=())
def frame_match(audio: Tensor, text: Tensor, kind: int, l2norm: bool, scale: bool) -> Tensor:
    """kind 0 = match.DotProduct (sigmoid(a.t [/sqrt D]).clamp(1e-7,1)), 1 = match.ExpNegL2 (exp(-||a - t||)): (B,T,D),(B,D) -> (B,T)."""
    return ops.match_forward(audio, text, kind, l2norm, scale)


@frame_match.register_fake
def _(audio, text, kind, l2norm, scale):
    return audio.new_empty(audio.shape[0], audio.shape[1])


@custom_op("tag::frame_match_backward", mutates_args=())
def frame_match_backward(audio: Tensor, text: Tensor, sim: Tensor, dsim: Tensor, kind: int, l2norm: bool,
                         scale: bool) -> Tuple[Tensor, Tensor]:
    return ops.match_backward(audio, text, sim, dsim, kind, l2norm, scale)


@frame_match_backward.register_fake
def _(audio, text, sim, dsim, kind, l2norm, scale):
    return torch.empty_like(audio), torch.empty_like(text)


def _match_setup(ctx, inputs, output):
    audio, text, kind, l2norm, scale = inputs
    ctx.save_for_backward(audio, text, output)
    ctx.cfg = (kind, l2norm, scale)


def _match_backward(ctx, dsim):
    audio, text, sim = ctx.saved_tensors
    da, dt = torch.ops.tag.frame_match_backward(audio, text, sim, dsim, *ctx.cfg)
    return da, dt, None, None, None


register_autograd("tag::frame_match", _match_backward, setup_context=_match_setup)


# ------------------------------------------------------------------------------------------------ M3 align.DotProduct
@custom_op("tag::align_dot", mutates_args=())
def align_dot(audio: Tensor, text: Tensor, l2norm: bool, scaled: bool) -> Tensor:
    """align.DotProduct: (B,T,D),(B,N,D) -> (B,B,T,N) on the MFMA GEMM with the sigmoid/clamp/scatter epilogue."""
    return ops.align_dot(audio, text, l2norm, scaled)


@align_dot.register_fake
def _(audio, text, l2norm, scaled):
    B, T, _ = audio.shape
    return audio.new_empty(B, B, T, text.shape[1])


@custom_op("tag::align_dot_backward", mutates_args=())
def align_dot_backward(audio: Tensor, text: Tensor, out: Tensor, dout: Tensor, l2norm: bool, scaled: bool) -> Tuple[Tensor, Tensor]:
    return ops.align_dot_backward(audio, text, out, dout, l2norm, scaled)


@align_dot_backward.register_fake
def _(audio, text, out, dout, l2norm, scaled):
    return torch.empty_like(audio), torch.empty_like(text)


def _align_setup(ctx, inputs, output):
    audio, text, l2norm, scaled = inputs
    ctx.save_for_backward(audio, text, output)
    ctx.cfg = (l2norm, scaled)


def _align_backward(ctx, dout):
    audio, text, out = ctx.saved_tensors
    da, dt = torch.ops.tag.align_dot_backward(audio, text, out, dout, *ctx.cfg)
    return da, dt, None, None


register_autograd("tag::align_dot", _align_backward, setup_context=_align_setup)


# ------------------------------------------------------------------------------------------------ L1 frame BCE
@custom_op("tag::frame_bce", mutates_args=())
def frame_bce(frame_sim: Tensor, label: Tensor, length: Tensor, Tt: int) -> Tensor:
    """FrameBceLoss over the first Tt frames, masked by clamp(length, 1, Tt): -> 0-dim loss."""
    return ops.frame_bce_forward(frame_sim, label, length, Tt)


@frame_bce.register_fake
def _(frame_sim, label, length, Tt):
    return frame_sim.new_empty(())


@custom_op("tag::frame_bce_backward", mutates_args=())
def frame_bce_backward(frame_sim: Tensor, label: Tensor, length: Tensor, Tt: int, dloss: Tensor) -> Tensor:
    return ops.frame_bce_backward(frame_sim, label, length, Tt, dloss)


@frame_bce_backward.register_fake
def _(frame_sim, label, length, Tt, dloss):
    return torch.empty_like(frame_sim)


def _bce_setup(ctx, inputs, output):
    frame_sim, label, length, Tt = inputs
    ctx.save_for_backward(frame_sim, label, length)
    ctx.Tt = Tt


def _bce_backward(ctx, dloss):
    frame_sim, label, length = ctx.saved_tensors
    return torch.ops.tag.frame_bce_backward(frame_sim, label, length, ctx.Tt, dloss), None, None, None


register_autograd("tag::frame_bce", _bce_backward, setup_context=_bce_setup)


# ------------------------------------------------------------------------------------------------ the fused audio encoders
# ``tag::cnn8rnn_encoder`` / ``tag::crnn_encoder``: log-mel -> conv stack -> (fc1) -> BiGRU as ONE operator each (rows F1-F3,
# A1-A4), what models/audio_encoder.py Cnn8Rnn.forward / CrnnEncoder.forward call -- the hot 97 % of the step goes through the
# operator registry like the heads do.  The operator wraps the engine of ops.py (ops.Cnn8RnnFunction / ops.CrnnFunction: the
# same forward / backward bodies, called with a plain context object): that engine keeps per-step state no functional
# operator could return cheaply (GBs of raw conv outputs saved for backward, dropout seeds, the direct-gradient sinks of the
# flat buffer, the weight-gradient side stream), so the saved state travels from the forward kernel to ``setup_context`` through
# a one-slot hand-over and the module (eps / momentum / dropout probabilities / frozen flags) is named by a token.
# EAGER-ONLY: the fake kernels serve shape inference (opcheck, meta tensors); a backward traced by torch.compile / AOTAutograd has
# no saved state and raises a clear error (_enc_backward).  A second backward through one forward (retain_graph) raises its own.
import weakref

_ENC_MODULES = weakref.WeakValueDictionary()          # token -> nn.Module
_ENC_HANDOVER = [None]                                # saved state of the forward that just ran -> its setup_context


def encoder_token(mod) -> int:
    """Registers ``mod`` (Cnn8Rnn / CrnnEncoder) for the encoder operators and returns its token."""
    tok = id(mod)
    _ENC_MODULES[tok] = mod
    return tok


class _EncCtx:
    """What ops.Cnn8RnnFunction.forward / .backward expect of their ctx."""

    def __init__(self, needs):
        self.needs_input_grad = needs
        self.saved = None


def _enc_forward(engine, waveform, params, module_token, need_grad):
    mod = _ENC_MODULES[module_token]
    ctx = _EncCtx((False, False) + tuple(bool(need_grad and p.requires_grad) for p in params))
    y = engine.forward(ctx, waveform, mod, *params)
    _ENC_HANDOVER[0] = (engine, ctx) if need_grad else None
    return y


def _enc_setup(ctx, inputs, output):
    ctx.enc, _ENC_HANDOVER[0] = _ENC_HANDOVER[0], None
    ctx.enc_consumed = False


def _enc_backward(ctx, dy):
    if ctx.enc is None:
        if getattr(ctx, "enc_consumed", False):
            raise RuntimeError("tag encoder operator: second backward through the same forward (retain_graph=True): the saved "
                               "activations (GBs of raw conv outputs) are released by the first backward -- run the forward again")
        raise RuntimeError("tag encoder operator: backward without saved state.  The operator is EAGER-ONLY: its saved state is "
                           "handed from the forward kernel to setup_context out of band, which torch.compile / AOTAutograd "
                           "tracing (fake tensors) cannot reproduce; call it in eager mode with need_grad=True under autograd")
    engine, ectx = ctx.enc
    ctx.enc = None
    ctx.enc_consumed = True
    grads = engine.backward(ectx, dy.contiguous())            # (None, None, *parameter gradients)
    return None, list(grads[2:]), None, None


def _enc_fake_frames(mod, waveform):
    return (waveform.shape[1] // mod.hop_length + 1) // mod.downsample_ratio


@custom_op("tag::cnn8rnn_encoder", mutates_args=())
def cnn8rnn_encoder(waveform: Tensor, params: List[Tensor], module_token: int, need_grad: bool) -> Tensor:
    """Cnn8Rnn.forward of models/audio_encoder.py:88-227: waveform (B,S) -> embedding (B, T', 512).  params in
    Cnn8Rnn._flat_params() order.  In train mode the BatchNorm running statistics of the MODULE named by the token are updated
    as nn.BatchNorm2d updates its buffers: module state, not operator arguments (torch.library refuses an autograd formula on an
    operator that declares mutated arguments)."""
    return _enc_forward(ops.Cnn8RnnFunction, waveform, params, module_token, need_grad)


@cnn8rnn_encoder.register_fake
def _(waveform, params, module_token, need_grad):
    mod = _ENC_MODULES[module_token]
    return waveform.new_empty(waveform.shape[0], _enc_fake_frames(mod, waveform), mod.embed_dim)


register_autograd("tag::cnn8rnn_encoder", _enc_backward, setup_context=_enc_setup)


@custom_op("tag::crnn_encoder", mutates_args=())
def crnn_encoder(waveform: Tensor, params: List[Tensor], module_token: int, need_grad: bool) -> Tensor:
    """CrnnEncoder.forward of models/audio_encoder.py:16-86: waveform (B,S) -> embedding (B, T', embed_dim)."""
    return _enc_forward(ops.CrnnFunction, waveform, params, module_token, need_grad)


@crnn_encoder.register_fake
def _(waveform, params, module_token, need_grad):
    mod = _ENC_MODULES[module_token]
    return waveform.new_empty(waveform.shape[0], _enc_fake_frames(mod, waveform), 2 * mod.gru.hidden_size)


register_autograd("tag::crnn_encoder", _enc_backward, setup_context=_enc_setup)


def run_encoder(op, mod, waveform, params):
    """Front door of the two encoder modules: calls ``op`` with the caller's grad mode made visible to the engine (inside an
    operator grad mode is always off) and the module's token."""
    need = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    prev, engine._RECORDING = engine._RECORDING, torch.is_grad_enabled()
    try:
        return op(waveform, list(params), encoder_token(mod), need)
    finally:
        engine._RECORDING = prev
        # setup_context has taken the saved state by now; if it was skipped (the dispatcher decided that no input needs a
        # gradient) the slot must not keep GBs of activations alive until the next forward
        _ENC_HANDOVER[0] = None


# ------------------------------------------------------------------------------------------------ P1 segments
@custom_op("tag::segments", mutates_args=())
def segments(frame_sim: Tensor, thresholds: Tensor, window_size: int, n_connect: int) -> Tuple[Tensor, Tensor]:
    """binarize -> median filter -> connect clusters -> contiguous regions for every (clip, threshold):
    regions (B,NT,ceil(T/2),2) int64 rows [onset, offset), counts (B,NT) int32."""
    return ops.segments(frame_sim, thresholds, window_size, n_connect)


@segments.register_fake
def _(frame_sim, thresholds, window_size, n_connect):
    B, T = frame_sim.shape
    NT = thresholds.numel()
    return (torch.empty(B, NT, (T + 1) // 2, 2, device=frame_sim.device, dtype=torch.int64),
            torch.empty(B, NT, device=frame_sim.device, dtype=torch.int32))


OP_NAMES = ["logmel", "conv3x3", "conv3x3_dgrad", "conv3x3_wgrad", "conv3x3_bn_relu_pool", "conv3x3_bn_relu_pool_backward",
            "gru_bidir", "gru_bidir_backward", "embed_mean", "embed_mean_backward", "frame_match", "frame_match_backward",
            "align_dot", "align_dot_backward", "frame_bce", "frame_bce_backward", "segments", "cnn8rnn_encoder", "crnn_encoder"]
