"""The reference's OWN training loop over the HIP modules (python_scripts/training/run_strong.py:92-152 restated as a test):
``Runner.forward`` + ``train_epoch`` with ``torch.optim.Adam``, ``optimizer.zero_grad()`` (set_to_none), ``loss.backward()``,
``clip_grad_norm_``, ``optimizer.step()``, ``ReduceLROnPlateau`` -- no StrongRunner, no flat buffers, no direct gradients: the
plain-autograd path through ``ops.Cnn8RnnFunction`` and the ``torch.ops.tag.*`` heads.  Then the checkpoint round trip of
run_strong.py:679-709 (``torch.save({"model": state_dict, ...})`` -> fresh model / oracle state).  Plus the latent hazards the
direct-gradient mode must survive: a parameter shared by two autograd nodes, and ``optimizer.zero_grad()`` mixed with
``StrongRunner``."""
import numpy as np
import pytest
import torch

from oracle import tag_oracle as O

pytestmark = pytest.mark.gpu


def build_model(st, dev):
    from texttoaudiogrounding_amd.models import audio_encoder, audio_text_model, match, text_encoder
    model = audio_text_model.BiEncoder(audio_encoder.Cnn8Rnn(32000), text_encoder.EmbeddingAgg(5221, 512), match.DotProduct(),
                                       512)
    missing = model.load_state_dict(st, strict=False)
    assert not missing.unexpected_keys and all("melspec_extractor" in k for k in missing.missing_keys)
    return model.to(dev)


class RefRunner:
    """run_strong.py:92-152 with the data loader replaced by a fixed batch (the loop body is the reference's, line for line
    in meaning: forward -> label alignment -> loss_fn -> backward -> clip -> step)."""

    def __init__(self, model, loss_fn, optimizer, lr_scheduler, device, max_grad_norm=1.0):
        self.model, self.loss_fn, self.optimizer, self.lr_scheduler = model, loss_fn, optimizer, lr_scheduler
        self.device, self.max_grad_norm = device, max_grad_norm

    def forward(self, batch, training=True):
        for k, v in batch.items():
            if isinstance(v, torch.Tensor):
                batch[k] = v.long().to(self.device) if k == "text" else v.float().to(self.device)
        input_dict = {"specaug": False}
        input_dict.update(batch)
        output = self.model(input_dict)
        if training:
            label, frame_sim = batch["label"], output["frame_sim"]
            tt = min(frame_sim.size(1), label.size(1))
            output.update({"frame_sim": frame_sim[..., :tt], "label": label[..., :tt],
                           "length": torch.clamp(output["length"], 1, tt)})
        return output

    def train_epoch(self, batches):
        history = []
        self.model.train()
        for batch in batches:
            self.optimizer.zero_grad()
            output = self.forward(batch, training=True)
            loss = self.loss_fn(output)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.max_grad_norm)
            self.optimizer.step()
            history.append(loss.item())
        return history


def test_reference_training_loop_runs_unchanged(dev, tmp_path):
    """10 steps of the reference loop on the HIP BiEncoder follow the fp64 oracle driven by the same loop; the checkpoint
    written the reference's way loads into a fresh HIP model and into the oracle."""
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd.losses import FrameBceLoss
    st = O.init_state(seed=23, logit_gain=20.0)
    batch = O.synthetic_batch(4, 32000, seed=9, ragged=True)
    model = build_model(st, dev)
    model.audio_encoder.dropout_p = (0.0, 0.0)
    optimizer = torch.optim.Adam(model.parameters(), lr=2e-4)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=0.5, patience=0)
    runner = RefRunner(model, FrameBceLoss(), optimizer, sched, dev)
    assert not ops.DIRECT_GRADS
    fresh = lambda: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    hip = runner.train_epoch([fresh() for _ in range(5)])
    assert all(p.grad is not None and not hasattr(p, "_tag_grad_sink") for p in model.parameters())
    sched.step(1e9)                                 # a validation loss that did not improve -> lr halves (run_strong.py:773-776)
    sched.step(1e9)
    assert abs(optimizer.param_groups[0]["lr"] - 1e-4) < 1e-12
    hip += runner.train_epoch([fresh() for _ in range(5)])
    ops.check_async_errors()
    # ---- the same loop on the fp64 oracle ----
    s64 = O.state_to(st, torch.float64, requires_grad=True)
    b64 = dict(batch)
    b64["waveform"], b64["label"] = batch["waveform"].double(), batch["label"].double()
    names = [n for n, _ in model.named_parameters()]
    params = [s64[n] for n in names]
    opt = torch.optim.Adam(params, lr=2e-4)
    ref = []
    for i in range(10):
        if i == 5:
            opt.param_groups[0]["lr"] = 1e-4
        opt.zero_grad()
        loss, _ = O.train_step_loss(s64, b64, "dot", "cnn8rnn", True, (0.0, 0.0))
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        ref.append(loss.item())
    print(f"reference loop: hip {['%.5f' % v for v in hip]}")
    print(f"             oracle {['%.5f' % v for v in ref]}")
    assert abs(hip[0] - ref[0]) < 2e-5
    assert max(abs(a - b) for a, b in zip(hip, ref)) < 3e-3            # same budget as test_training_trajectory_matches_oracle
    assert ref[-1] < ref[0] - 0.003
    # ---- checkpoint round trip (run_strong.py:679-709 / utils/train_util.py:287-297) ----
    path = tmp_path / "ckpt.pth"
    torch.save({"model": model.state_dict(), "optimizer": optimizer.state_dict(), "lr_scheduler": sched.state_dict()}, path)
    ck = torch.load(path, map_location="cpu")
    assert len(ck["model"]) == 66 and "audio_encoder.melspec_extractor.mel_scale.fb" in ck["model"]
    model2 = build_model(O.init_state(seed=1), dev)
    model2.load_state_dict(ck["model"])                                # strict: every key and shape matches
    model.eval(), model2.eval()
    with torch.no_grad():
        inp = lambda: {**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, "specaug": False}
        o1 = model({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp().items()})
        o2 = model2({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp().items()})
    assert torch.equal(o1["frame_sim"], o2["frame_sim"])
    st_o = {k: v for k, v in ck["model"].items() if "melspec" not in k}
    oo = O.biencoder_forward(st_o, batch, "dot", "cnn8rnn", training=False)
    err = (o1["frame_sim"].cpu() - oo["frame_sim"]).abs().max().item()
    print(f"checkpoint -> oracle: eval frame_sim err {err:.2e}")
    assert err < 1e-4 and torch.equal(o1["length"].cpu(), oo["length"])


def test_direct_grads_shared_parameter_matches_autograd(dev):
    """a parameter seen by TWO autograd nodes in one forward (one Linear applied to two tensors) under
    ops.DIRECT_GRADS -- neither node may overwrite the other's contribution in the flat-gradient sink.  The result must
    equal plain autograd's sum."""
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd.runner import FlatParams
    torch.manual_seed(0)
    lin = torch.nn.Linear(64, 32).to(dev)
    other = torch.nn.Linear(32, 16).to(dev)                               # claimed once: stays on the direct path
    x1, x2 = torch.randn(40, 64, device=dev), torch.randn(24, 64, device=dev)

    def loss_of():
        a = ops.LinearFunction.apply(x1, lin.weight, lin.bias)
        b = ops.LinearFunction.apply(x2, lin.weight, lin.bias)            # same parameters, second node
        c = ops.LinearFunction.apply(a, other.weight, other.bias)
        return (a * a).sum() + 3.0 * b.sum() + c.sum()

    loss_of().backward()
    want = [p.grad.clone() for p in (*lin.parameters(), *other.parameters())]
    mod = torch.nn.ModuleList([lin, other])
    flat = FlatParams(mod)
    flat.zero_grad()
    ready = []
    prev = (ops.DIRECT_GRADS, ops.GRAD_READY)
    ops.DIRECT_GRADS, ops.GRAD_READY = True, lambda ps: ready.extend(id(p) for p in ps)
    ops.begin_direct_step()
    try:
        loss_of().backward()
    finally:
        ops.DIRECT_GRADS, ops.GRAD_READY = prev
    for p, w in zip((*lin.parameters(), *other.parameters()), want):
        assert torch.allclose(p.grad, w, rtol=1e-5, atol=1e-5), (p.shape, (p.grad - w).abs().max())
        assert p.grad.data_ptr() == p._tag_grad_sink.data_ptr()           # still the flat view
    # the shared Linear was never announced as final to the gradient buckets; the single-use one was
    assert id(lin.weight) not in ready and id(lin.bias) not in ready
    assert id(other.weight) in ready and id(other.bias) in ready


def test_strong_runner_survives_optimizer_zero_grad(dev):
    """model.zero_grad() / optimizer.zero_grad(set_to_none=True) between StrongRunner steps: the flat-gradient views are
    re-attached at the next forward_backward (gradients are being zeroed there anyway) instead of the kernels writing into
    sinks p.grad no longer aliases; a re-homed parameter (.to() / .data = ...) fails fast at forward_backward entry."""
    from texttoaudiogrounding_amd.runner import StrongRunner
    st = O.init_state(seed=23, logit_gain=20.0)
    batch = O.synthetic_batch(2, 32000, seed=9)
    fresh = lambda: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    model = build_model(st, dev).train()
    model.audio_encoder.dropout_p = (0.0, 0.0)
    runner = StrongRunner(model, device=str(dev))
    runner.forward_backward(fresh())
    g0 = runner.flat.grad.clone()
    model.zero_grad()                                                     # set_to_none: every p.grad is now None
    assert all(p.grad is None for p in model.parameters())
    runner.forward_backward(fresh())
    runner.flat.check()
    assert torch.equal(runner.flat.grad, g0)                              # same step, same gradients, in the flat buffer
    assert all(p.grad is not None and p.grad.data_ptr() == p._tag_grad_sink.data_ptr() for p in model.parameters())
    p0 = next(model.parameters())
    p0.data = p0.data.clone()                                             # re-homed outside the flat buffer
    with pytest.raises(RuntimeError, match="flat"):
        runner.forward_backward(fresh())
