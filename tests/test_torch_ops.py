"""torch.library registration (namespace ``tag``): schemas + fake kernels on CPU (no compute), numerics and
torch.library.opcheck on the GPU."""
import pytest
import torch


def test_ops_registered_with_schemas_and_fake_kernels():
    import texttoaudiogrounding_amd.torch_ops as T
    from torch._subclasses.fake_tensor import FakeTensorMode
    for name in T.OP_NAMES:
        assert hasattr(torch.ops.tag, name), name
        schema = str(getattr(torch.ops.tag, name).default._schema)
        assert schema.startswith(f"tag::{name}("), schema
    with FakeTensorMode():
        wave = torch.empty(4, 32000)
        lm = torch.ops.tag.logmel(wave, 1024, 1024, 320, torch.empty(1024), torch.empty(513, 64))
        assert lm.shape == (4, 101, 64)
        y = torch.ops.tag.conv3x3(torch.empty(2, 25, 16, 128), torch.empty(256, 128, 3, 3), 0, None, None)
        assert y.shape == (2, 25, 16, 256)
        assert torch.ops.tag.conv3x3_wgrad(torch.empty(2, 25, 16, 128), y, 0, None, None).shape == (256, 128, 3, 3)
        seq, tok = torch.ops.tag.embed_mean(torch.empty(5221, 512), torch.empty(4, 6, dtype=torch.long), torch.empty(4, dtype=torch.long))
        assert seq.shape == (4, 512) and tok.shape == (4, 6, 512)
        sim = torch.ops.tag.frame_match(torch.empty(4, 250, 512), seq, 0, False, True)
        assert sim.shape == (4, 250)
        assert torch.ops.tag.align_dot(torch.empty(4, 250, 512), tok, False, False).shape == (4, 4, 250, 6)
        assert torch.ops.tag.frame_bce(sim, torch.empty(4, 250), torch.empty(4, dtype=torch.long), 250).shape == ()
        regions, counts = torch.ops.tag.segments(sim, torch.empty(50, dtype=torch.float64), 1, 13)
        assert regions.shape == (4, 50, 125, 2) and counts.dtype == torch.int32
        rnn = [torch.empty(768, 512), torch.empty(768, 256), torch.empty(768), torch.empty(768)] * 2
        yy, gates = torch.ops.tag.gru_bidir(torch.empty(4, 250, 512), *rnn)
        assert yy.shape == (4, 250, 512) and gates.shape == (4, 250, 2, 1024)
    # the fused encoders (round 4): shape inference needs the module (hop, downsample ratio, width) named by the token
    from texttoaudiogrounding_amd.models import audio_encoder
    cnn8, crnn = audio_encoder.Cnn8Rnn(32000), audio_encoder.CrnnEncoder(32000, 256)
    t8, tc = T.encoder_token(cnn8), T.encoder_token(crnn)
    with FakeTensorMode():
        emb = torch.ops.tag.cnn8rnn_encoder(torch.empty(4, 320000), [], t8, False)
        assert emb.shape == (4, 250, 512)
        assert torch.ops.tag.crnn_encoder(torch.empty(3, 64000), [], tc, False).shape == (3, 25, 256)      # hop 640, downsample 4


def test_ops_raise_on_cpu_tensors():
    import texttoaudiogrounding_amd.torch_ops  # noqa: F401
    with pytest.raises(RuntimeError):
        torch.ops.tag.logmel(torch.zeros(1, 3200), 1024, 1024, 320, torch.hann_window(1024), torch.zeros(513, 64))


@pytest.mark.gpu
def test_ops_match_the_autograd_nodes_and_pass_opcheck(dev):
    import texttoaudiogrounding_amd.torch_ops  # noqa: F401
    from oracle import tag_oracle as O
    from texttoaudiogrounding_amd import ops
    g = torch.Generator().manual_seed(0)
    B, T, D, L, V = 3, 11, 64, 4, 50
    table = torch.randn(V, D, generator=g).to(dev).requires_grad_(True)
    text = torch.randint(1, V, (B, L), generator=g).to(dev)
    lens = torch.tensor([4, 1, 3]).to(dev)
    audio = torch.randn(B, T, D, generator=g).to(dev).requires_grad_(True)
    label = (torch.rand(B, T, generator=g) < 0.5).float().to(dev)
    length = torch.tensor([11, 7, 9]).to(dev)
    # composed through the registered operators ...
    seq, tok = torch.ops.tag.embed_mean(table, text, lens)
    sim = torch.ops.tag.frame_match(audio, seq, 0, False, True)
    loss = torch.ops.tag.frame_bce(sim, label, length, T)
    loss.backward()
    g1 = (audio.grad.clone(), table.grad.clone())
    audio.grad = table.grad = None
    # ... equals the same formulas driven by hand (ops.py's plain forward / backward functions, which ARE the operators'
    # bodies) and the direct-gradient EmbedMeanFunction node kept for StrongRunner
    seq2, _ = ops.EmbedMeanFunction.apply(table, text, lens, True)
    sim2 = ops.match_forward(audio.detach(), seq2.detach(), 0, False, True)
    loss2 = ops.frame_bce_forward(sim2, label, length, T)
    dsim = ops.frame_bce_backward(sim2, label, length, T, torch.ones((), device=dev))
    da, dseq = ops.match_backward(audio.detach(), seq2.detach(), sim2, dsim, 0, False, True)
    seq2.backward(dseq)
    assert torch.equal(loss, loss2) and torch.equal(g1[0], da) and torch.equal(g1[1], table.grad)
    # conv + GRU operators against the oracle
    x = torch.randn(2, 9, 8, 64, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(128, 64, 3, 3, generator=g) / 24).to(dev).requires_grad_(True)
    y = torch.ops.tag.conv3x3(x, w, 0, None, None)
    y.square().sum().backward()
    xd, wd = x.detach().cpu().double().requires_grad_(True), w.detach().cpu().double().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xd.permute(0, 3, 1, 2), wd, padding=1).permute(0, 2, 3, 1)
    yr.square().sum().backward()
    rel = lambda a, b: (a.detach().cpu().double() - b).abs().max().item() / b.abs().max().item()
    assert rel(y, yr.detach()) < 5e-6 and rel(x.grad, xd.grad) < 1e-5 and rel(w.grad, wd.grad) < 1e-5
    # opcheck: schema / fake-kernel / autograd registration consistency
    torch.library.opcheck(torch.ops.tag.frame_match, (audio.detach().requires_grad_(True), seq.detach().requires_grad_(True), 1, True, False),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    torch.library.opcheck(torch.ops.tag.logmel, (0.1 * torch.randn(2, 8000, device=dev), 1024, 1024, 320,
                                                 torch.hann_window(1024, device=dev), O.frontend_tables("cnn8rnn")[1].to(dev)),
                          test_utils=("test_schema", "test_faketensor"))
