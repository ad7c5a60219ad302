"""Engine of the torch bindings over libtag_hip.so: tensor checks and scratch allocation, launch timing for bench.py, dropout
seeds, the direct-gradient bookkeeping of the autograd nodes (flat-gradient sinks, claims, the second-writer guard), the base
class of those nodes and the weight-gradient side stream.  No kernel is chosen here (dispatch.py) and no autograd node is
defined here (functions.py); ops.py is the namespace callers import.
"""
from __future__ import annotations

import torch

from . import lib
from . import settings as cfg
from .lib import query

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


class _timed:
    def __init__(self, key, flops):
        self.key, self.flops = key, flops

    def __enter__(self):
        if cfg.PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream())

    def __exit__(self, *a):
        if cfg.PROFILE is not None:
            self.e1.record(torch.cuda.current_stream())
            cfg.PROFILE.setdefault(self.key, []).append((self.e0, self.e1, self.flops))


def new_seed() -> int:
    """Dropout seed drawn from torch's global CPU generator (so torch.manual_seed controls it), decorrelated per rank."""
    s = int(torch.randint(0, 2 ** 62, (1,)).item())
    return (s + cfg.SEED_RANK * 0x9E3779B97F4A7C15) % (2 ** 62)


# ------------------------------------------------------------------------------------------------
# Direct gradients.  runner.FlatParams gives every trainable parameter a view of ONE flat gradient buffer
# (``p._tag_grad_sink``).  While settings.DIRECT_GRADS is on (StrongRunner.forward_backward, after its zero_grad) the autograd
# nodes (functions.py) write parameter gradients straight into those views and return None for them: no AccumulateGrad
# ``grad += new`` kernels, and a node can announce "these gradients are final" (GRAD_READY) so that the data-parallel
# all-reduce of a bucket starts while the rest of backward is still running (runner.GradBuckets).
# ------------------------------------------------------------------------------------------------

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
    return _LazySinks(params, cfg.DIRECT_GRADS and _RECORDING)


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
        if g is None or not cfg.DIRECT_GRADS or _CLAIMS.get(id(p), 0) != 1:
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
    if cfg.GRAD_READY is not None and params:
        cfg.GRAD_READY([t for t in params if isinstance(t, torch.Tensor) and _CLAIMS.get(id(t), 0) == 1])


def _flush():
    if cfg.GRAD_FLUSH is not None:
        cfg.GRAD_FLUSH()


# ------------------------------------------------------------------------------------------------
# The weight-gradient side stream (settings.WGRAD_SIDE_STREAM; scheduled by functions._SideWgrad)
# ------------------------------------------------------------------------------------------------
_side_streams = {}


def side_stream_enabled():
    return cfg.WGRAD_SIDE_STREAM if cfg.WGRAD_SIDE_STREAM is not None else (cfg.CONV_MATH != "fp32")


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _side_streams:
        st = None
        if cfg.WGRAD_CU_SKIP >= 2:
            import ctypes
            ncu = query("tag_device_cu_count")
            words = (ncu + 31) // 32
            mask = (ctypes.c_uint32 * words)()
            for i in range(ncu):
                if i % cfg.WGRAD_CU_SKIP != cfg.WGRAD_CU_SKIP - 1:
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
