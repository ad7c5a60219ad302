"""The reference's sub-modules run ON THEIR OWN on the HIP path: ``ConvBlock.forward(input, pool_size, pool_type)``
(models/panns.py:46-62, three pool types), ``Seq2SeqAttention.forward`` with d_q != d_kv (models/cross_encoder.py:11-42) and
``CrossGating.forward`` (:45-57), against tests/golden/submodules.npz -- outputs, gradients and updated BatchNorm buffers of
the imported reference classes (fp64 twin; the reference's own fp32 run is the round-off floor).  Modules are constructed
under the same seeds as in tests/golden/make_golden_submodules.py; a checksum proves the parameters are the fixture's."""
import numpy as np
import pytest
import torch

from tests.golden import make_golden_submodules as M

pytestmark = pytest.mark.gpu


def close(got, want64, floor32=None, tol=2e-5):
    """max-normalised distance from the fp64 twin (sampled layout of make_golden_submodules.sample for big tensors)."""
    want64 = np.asarray(want64, dtype=np.float64).ravel()
    got = got.detach().cpu()
    got = got.double().flatten().numpy() if got.numel() == want64.size else M.sample(got)
    assert got.shape == want64.shape, (got.shape, want64.shape)
    err = np.abs(got - want64).max() / (np.abs(want64).max() + 1e-30)
    return err


@pytest.mark.parametrize("case", M.CONV_CASES, ids=[c[0] for c in M.CONV_CASES])
def test_convblock_forward_standalone(dev, golden_dir, case):
    from texttoaudiogrounding_amd.models.panns import ConvBlock
    name, cin, cout, shape, psz, ptype = case
    gold = np.load(f"{golden_dir}/submodules.npz")
    blk = M.make_convblock(ConvBlock, name, cin, cout)
    x, g = M.convblock_inputs(name, cin, shape)
    assert np.allclose(np.concatenate([M.checksum(x), M.checksum(blk.conv2.weight.detach())]), gold[f"{name}/checksum"],
                       rtol=1e-12), "seeded inputs drifted from the fixture"
    blk = blk.to(dev)
    blk.eval()
    with torch.no_grad():
        ev = blk(x.to(dev), pool_size=psz, pool_type=ptype)
    want = gold[f"{name}/eval_f64"]
    assert ev.shape == want.shape                                   # NCHW out, like the reference
    e_eval = close(ev, want)
    blk.train()
    xi = x.to(dev).requires_grad_(True)
    y = blk(xi, pool_size=psz, pool_type=ptype)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.to(dev))
    errs = {"eval": e_eval, "train": close(y, gold[f"{name}/train_f64"]), "dx": close(xi.grad, gold[f"{name}/dx_f64"])}
    for n, p in blk.named_parameters():
        errs["grad " + n] = close(p.grad, gold[f"{name}/grad_f64/{n}"])
    for n, b in blk.named_buffers():
        want_b = gold[f"{name}/buf_f64/{n}"]
        if n.endswith("num_batches_tracked"):
            assert int(b.item()) == int(want_b) == 1
        else:
            errs["buf " + n] = close(b, want_b)
    floor = np.abs(gold[f"{name}/train_f32"].astype(np.float64) - gold[f"{name}/train_f64"]).max() / np.abs(gold[f"{name}/train_f64"]).max()
    print(f"ConvBlock {name}:", {k: f"{v:.1e}" for k, v in errs.items()}, f"(reference f32 vs f64 on train out: {floor:.1e})")
    # max-pool / ReLU decisions: none sits within fp32 round-off of a tie at these seeds -> everything is round-off
    assert all(v < 3e-5 for v in errs.values()), errs
    with pytest.raises(Exception, match="Incorrect argument"):
        blk(x.to(dev), pool_size=psz, pool_type="lp")


def test_seq2seq_attention_forward_standalone(dev, golden_dir):
    from texttoaudiogrounding_amd.models.cross_encoder import Seq2SeqAttention
    gold = np.load(f"{golden_dir}/submodules.npz")
    m, q, kv, ql, kl, dout = M.make_attn(Seq2SeqAttention)
    assert np.allclose(np.concatenate([M.checksum(q), M.checksum(m.h2attn.weight.detach())]), gold["attn/checksum"], rtol=1e-12)
    m = m.to(dev)
    qi, ki = q.to(dev).requires_grad_(True), kv.to(dev).requires_grad_(True)
    o = m(qi, ki, ql, kl)
    assert o.shape == (M.ATTN["B"], M.ATTN["Lq"], M.ATTN["d_kv"])
    o.backward(dout.to(dev))
    errs = {"out": close(o, gold["attn/out_f64"]), "dq": close(qi.grad, gold["attn/dq_f64"]),
            "dkv": close(ki.grad, gold["attn/dkv_f64"])}
    for n, p in m.named_parameters():
        errs["grad " + n] = close(p.grad, gold[f"attn/grad_f64/{n}"])
    print("Seq2SeqAttention (d_q 128, d_kv 64, d_attn 64):", {k: f"{v:.1e}" for k, v in errs.items()})
    assert all(v < 2e-5 for v in errs.values()), errs


def test_cross_gating_forward_standalone(dev, golden_dir):
    from texttoaudiogrounding_amd.models.cross_encoder import CrossGating
    gold = np.load(f"{golden_dir}/submodules.npz")
    m, u, s, du, ds = M.make_gate(CrossGating)
    assert np.allclose(np.concatenate([M.checksum(u), M.checksum(m.fc_u.weight.detach())]), gold["gate/checksum"], rtol=1e-12)
    m = m.to(dev)
    ui, si = u.to(dev).requires_grad_(True), s.to(dev).requires_grad_(True)
    uo, so = m(ui, si)
    torch.autograd.backward([uo, so], [du.to(dev), ds.to(dev)])
    errs = {"u_out": close(uo, gold["gate/u_out_f64"]), "s_out": close(so, gold["gate/s_out_f64"]),
            "du": close(ui.grad, gold["gate/du_f64"]), "ds": close(si.grad, gold["gate/ds_f64"])}
    for n, p in m.named_parameters():
        errs["grad " + n] = close(p.grad, gold[f"gate/grad_f64/{n}"])
    print("CrossGating:", {k: f"{v:.1e}" for k, v in errs.items()})
    assert all(v < 2e-5 for v in errs.values()), errs


def test_conv3x3_bn_relu_pool_operator_opcheck(dev):
    """The operator SURVEY.md section 8(b) lists, as PyTorch sees it: schema / fake kernel / autograd registration."""
    import texttoaudiogrounding_amd.torch_ops  # noqa: F401
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 6, 8, 32, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(64, 32, 3, 3, generator=g) / 17).to(dev).requires_grad_(True)
    gamma, beta = torch.rand(64, generator=g).add(0.5).to(dev).requires_grad_(True), torch.randn(64, generator=g).to(dev).requires_grad_(True)
    rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
    for pool in (0, 2, 3):
        args = (x, w, gamma, beta, rm, rv, True, 0.1, 1e-5, 2, 2, pool)
        torch.library.opcheck(torch.ops.tag.conv3x3_bn_relu_pool, args,
                              test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    assert torch.equal(rm, torch.zeros_like(rm))                 # functional: the inputs are not mutated
    out = torch.ops.tag.conv3x3_bn_relu_pool(*args)
    assert out[0].shape == (2, 3, 4, 64) and not torch.equal(out[6], rm)


@pytest.mark.parametrize("case", ["rank1", "rank2", "rank3"])
def test_embedding_layer_forward_standalone(dev, golden_dir, case):
    """``EmbeddingLayer.forward`` on its own (models/text_encoder.py:39-43; round 3 raised here): tokens of any rank, padding
    and repeated ids, output bit-equal to the imported reference's lookup and the table gradient (a deterministic gather) equal
    to its fp64 twin -- tests/golden/embedding_layer.npz (make_golden_embedding.py)."""
    from tests.golden import make_golden_embedding as E
    from texttoaudiogrounding_amd.models.text_encoder import EmbeddingLayer
    gold = np.load(f"{golden_dir}/embedding_layer.npz")
    m = E.make_layer(EmbeddingLayer)
    assert np.allclose(E.checksum(m.core.weight.detach()), gold["table_checksum"], rtol=1e-12), "seeded table drifted"
    m = m.to(dev)
    t, dout = E.tokens(case)
    out = m({"text": t.int()})                                   # host ids, int32: cast like the reference's .long()
    assert tuple(out.shape) == E.CASES[case] + (E.D,)
    assert np.array_equal(out.detach().cpu().numpy(), gold[f"{case}/out"])         # a gather: bit-exact
    out.backward(dout.to(dev))
    err = close(m.core.weight.grad, gold[f"{case}/dtable_f64"])
    assert err < 1e-6, err
    with pytest.raises(IndexError):                              # nn.Embedding raises on an id outside the table
        m({"text": torch.tensor([0, E.V])})
    out2 = m({"text": t.to(dev)})                                # device-resident ids take the same path
    assert torch.equal(out2, out)


def test_encoders_run_through_the_operator_registry(dev, monkeypatch):
    """north_star: "exposed to Python through PyTorch-ROCm custom ops" -- round 4: Cnn8Rnn.forward / CrnnEncoder.forward call
    ``torch.ops.tag.cnn8rnn_encoder`` / ``tag.crnn_encoder`` (torch.library operators with fake kernels and a registered
    autograd formula around the fused engine).  The operator's result and gradients are bit-equal to the engine applied as a
    plain autograd.Function (the same kernels in the same order), and opcheck accepts schema / fake kernel / autograd registration."""
    import texttoaudiogrounding_amd.torch_ops as T
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd.models import audio_encoder
    calls = []
    monkeypatch.setattr(T, "run_encoder", lambda op, *a, _r=T.run_encoder: (calls.append(op), _r(op, *a))[1])
    for cls, engine, opname in ((audio_encoder.Cnn8Rnn, ops.Cnn8RnnFunction, "cnn8rnn_encoder"),
                                (audio_encoder.CrnnEncoder, ops.CrnnFunction, "crnn_encoder")):
        calls.clear()
        torch.manual_seed(3)
        m = (cls(32000) if cls is audio_encoder.Cnn8Rnn else cls(32000, 256)).to(dev).train()
        if cls is audio_encoder.Cnn8Rnn:
            m.dropout_p = (0.0, 0.0)
        else:
            m.dropout_p = 0.0
        wave = (0.1 * torch.randn(3, 16000, generator=torch.Generator().manual_seed(5))).to(dev)
        real = getattr(torch.ops.tag, opname)
        out = m({"waveform": wave, "waveform_len": [16000] * 3, "specaug": False})["embedding"]
        assert calls == [real], "the module's forward must go through torch.ops.tag." + opname
        dy = torch.randn(out.shape, generator=torch.Generator().manual_seed(6)).to(dev)
        out.backward(dy)
        g_op = [p.grad.clone() for p in m.parameters()]
        buf_op = [b.clone() for b in m.buffers()]
        # the same step through the engine as a plain autograd node, from the same starting state
        torch.manual_seed(3)
        m2 = (cls(32000) if cls is audio_encoder.Cnn8Rnn else cls(32000, 256)).to(dev).train()
        m2.dropout_p = m.dropout_p
        out2 = engine.apply(wave, m2, *m2._flat_params())
        out2.backward(dy)
        assert torch.equal(out, out2)
        assert all(torch.equal(a, p.grad) for a, p in zip(g_op, m2.parameters()))
        # running statistics were updated by the operator exactly as by the node (num_batches_tracked is bumped by the module)
        for (n, b), b2 in zip(m.named_buffers(), m2.buffers()):
            if not n.endswith("num_batches_tracked"):
                assert torch.equal(b, b2), n
        # eval / no_grad: no saved state is kept
        with torch.no_grad():
            m.eval()
            m({"waveform": wave, "waveform_len": [16000] * 3, "specaug": False})
        assert T._ENC_HANDOVER[0] is None
        torch.library.opcheck(real, (wave, list(m._flat_params()), T.encoder_token(m), False),
                              test_utils=("test_schema", "test_faketensor"))
        torch.library.opcheck(real, (wave, list(m.train()._flat_params()), T.encoder_token(m), True),
                              test_utils=("test_autograd_registration",))
