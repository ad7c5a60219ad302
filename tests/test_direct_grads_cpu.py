"""CPU: the host logic of direct gradient delivery (ops._sinks / _deliver / second_writer_guard) on a stand-in autograd node
-- the claim counting and the guard are pure Python and do not need the HIP library."""
import pytest
import torch

from texttoaudiogrounding_amd import ops
from texttoaudiogrounding_amd.runner import FlatParams


class _Scale(ops.TagFunction):
    """y = x * w with the parameter gradient delivered the way the HIP nodes deliver theirs."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        ctx.sinks = ops._sinks([x, w])
        return x * w

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        grads = [dy * w, None]
        ops._deliver(grads, ctx.sinks, 1, dy * x)
        return tuple(grads)


@pytest.fixture
def flat_w():
    m = torch.nn.Linear(4, 1, bias=False)
    fp = FlatParams(m)
    old = ops.DIRECT_GRADS
    ops.DIRECT_GRADS = True
    ops.begin_direct_step()
    yield m.weight, fp
    ops.DIRECT_GRADS = old


def test_single_claim_is_delivered_in_place(flat_w):
    w, fp = flat_w
    x = torch.arange(4.0).view(1, 4).requires_grad_()
    _Scale.apply(x, w).sum().backward()
    assert torch.equal(fp.grad, torch.arange(4.0))
    assert w.grad.data_ptr() == fp.grad.data_ptr()


def test_plain_torch_second_consumer_raises(flat_w):
    """a parameter claimed by one HIP node AND used by a plain torch op must not be summed into the view the
    node overwrites (order-dependent result): the parameter's tensor hook raises."""
    w, fp = flat_w
    x = torch.arange(4.0).view(1, 4)
    loss = _Scale.apply(x, w).sum() + 0.5 * (w * w).sum()
    with pytest.raises(RuntimeError, match="also received a gradient through plain autograd"):
        loss.backward()


def test_no_grad_forward_claims_nothing(flat_w):
    """A forward under no_grad (evaluation between steps, the first pass of a checkpointed segment) must not turn the real
    node's parameter into a 'shared' one."""
    w, fp = flat_w
    x = torch.arange(4.0).view(1, 4)
    with torch.no_grad():
        _Scale.apply(x, w)
    assert ops._CLAIMS.get(id(w), 0) == 0
    _Scale.apply(x, w).sum().backward()
    assert ops._CLAIMS[id(w)] == 1 and torch.equal(fp.grad, torch.arange(4.0))


def test_two_claims_fall_back_to_accumulate(flat_w):
    w, fp = flat_w
    x = torch.arange(4.0).view(1, 4)
    (_Scale.apply(x, w).sum() + _Scale.apply(2 * x, w).sum()).backward()      # hook fires legitimately: claims == 2
    assert torch.equal(fp.grad, 3 * torch.arange(4.0))


def test_claimed_but_unused_node_plus_plain_consumer_has_one_writer(flat_w):
    """the guard decides on what HAPPENED in the step.  A parameter claimed by a HIP node whose output never takes part
    in this backward (a metric-only forward under grad mode) and also used by a plain torch op has ONE writer -- autograd --
    and must train normally instead of raising."""
    w, fp = flat_w
    x = torch.arange(4.0).view(1, 4)
    _ = _Scale.apply(x, w)                                   # claimed (claims == 1), never backpropagated
    (0.5 * (w * w).sum()).backward()
    assert ops._CLAIMS[id(w)] == 1
    assert torch.equal(fp.grad, w.detach().reshape(-1))      # d/dw 0.5 w^2 = w, accumulated by AccumulateGrad into the flat view


def test_second_writer_is_caught_in_either_order(flat_w):
    """Autograd may run the plain consumer's gradient before OR after the node's in-place delivery: both orders raise."""
    w, fp = flat_w
    x = torch.arange(4.0).view(1, 4)
    for build in (lambda: _Scale.apply(x, w).sum() + 0.5 * (w * w).sum(), lambda: 0.5 * (w * w).sum() + _Scale.apply(x, w).sum()):
        ops.begin_direct_step()
        fp.zero_grad()
        with pytest.raises(RuntimeError, match="also received a gradient through plain autograd"):
            build().backward()
