"""BASELINE configs[3], the head its text names: match.CrossAttention (models/match.py:63-88 in the reference) on the HIP
path -- against tests/golden/cross_attention.npz (outputs and fp64 gradients of the IMPORTED reference module, both
parameter layouts, head dims 16 / 32 / 64), against the fp64 oracle with the HIP path's own dropout masks replayed, at
the benched shape (B = 64, T' = 250, E = 512, 8 heads), and inside a whole BiEncoder training step."""
import numpy as np
import pytest
import torch

from oracle import tag_oracle as O

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


def build(E, H, Dk, p, st, dev):
    from texttoaudiogrounding_amd.models import match
    m = match.CrossAttention(E, H, p, kvdim=None if Dk == E else Dk)
    m.load_state_dict({k[len("match_fn."):]: torch.as_tensor(v) for k, v in st.items()})
    return m.to(dev)


@pytest.mark.parametrize("case", ["packed", "kv", "wide"])
def test_cross_attention_golden(dev, golden_dir, case):
    gold = np.load(f"{golden_dir}/cross_attention.npz")
    E, H, Dk, B, T, L = (int(v) for v in gold[f"{case}/cfg"])
    st = {k[len(case) + 3:]: gold[k] for k in gold.files if k.startswith(f"{case}/w/")}
    m = build(E, H, Dk, 0.0, st, dev).train()
    a = torch.from_numpy(gold[f"{case}/audio"]).to(dev).requires_grad_(True)
    t = torch.from_numpy(gold[f"{case}/token"]).to(dev).requires_grad_(True)
    sim = m({"audio_emb": a, "text_emb": {"token_emb": t}, "text_len": gold[f"{case}/text_len"]})
    assert sim.shape == (B, T)
    e_sim = (sim.detach().cpu().double() - torch.from_numpy(gold[f"{case}/sim_f64"])).abs().max().item()
    sim.backward(torch.from_numpy(gold[f"{case}/dsim"]).to(dev))
    floor = float(gold[f"{case}/grad_floor"])
    bound = 4.0 * max(floor, 2e-6)
    worst = max(relerr(a.grad, gold[f"{case}/daudio"]), relerr(t.grad, gold[f"{case}/dtoken"]))
    for n, p in m.named_parameters():
        e = relerr(p.grad, gold[f"{case}/grad/match_fn.{n}"])
        worst = max(worst, e)
        assert e <= bound, (n, e, floor)
    print(f"CrossAttention[{case}] E={E} H={H} kv={Dk}: sim err {e_sim:.2e}, worst gradient err {worst:.2e} "
          f"(reference fp32 floor {floor:.2e})")
    assert e_sim < 1e-5 and worst <= bound                 # north_star tolerance for scores: 1e-4


@pytest.mark.parametrize("B,T,L,E,H,p", [(3, 11, 4, 128, 4, 0.3), (64, 250, 6, 512, 8, 0.1), (2, 5, 32, 64, 4, 0.0)])
def test_cross_attention_dropout_replay_vs_oracle(dev, B, T, L, E, H, p):
    """Train mode with dropout: the two keep masks the kernels drew (attention weights, residual branch) are exported with
    tag_dropout_mask and replayed in the fp64 oracle.  (64, 250, 6, 512, 8) = the benched shape of configs[3]."""
    from texttoaudiogrounding_amd import ops
    g = torch.Generator().manual_seed(B + T + E)
    torch.manual_seed(3)
    from texttoaudiogrounding_amd.models import match
    m = match.CrossAttention(E, H, p).to(dev).train()
    st = {"match_fn." + k: v.detach().cpu() for k, v in m.state_dict().items()}
    audio, token = torch.randn(B, T, E, generator=g), torch.randn(B, L, E, generator=g)
    text_len = 1 + torch.arange(B) % L
    dsim = torch.randn(B, T, generator=g)
    seeds = [11, 22]
    it = iter(seeds)
    from texttoaudiogrounding_amd import functions
    old = functions.new_seed
    functions.new_seed = lambda: next(it)        # (patched where the nodes look it up)
    try:
        a = audio.to(dev).requires_grad_(True)
        t = token.to(dev).requires_grad_(True)
        sim = m({"audio_emb": a, "text_emb": {"token_emb": t}, "text_len": text_len})
        sim.backward(dsim.to(dev))
    finally:
        functions.new_seed = old
    ak = rk = None
    if p > 0:
        ak = ops.dropout_mask(seeds[0], (B, T, H, L), p, dev).cpu().double()
        rk = ops.dropout_mask(seeds[1], (B, T, E), p, dev).cpu().double()
    st64 = {k: v.double().requires_grad_(True) for k, v in st.items()}
    ad, td = audio.double().requires_grad_(True), token.double().requires_grad_(True)
    ref = O.match_cross_attention(st64, ad, td, text_len, H, attn_keep=ak, res_keep=rk, p_drop=p)
    ref.backward(dsim.double())
    e_sim = (sim.detach().cpu().double() - ref.detach()).abs().max().item()
    worst = max(relerr(a.grad, ad.grad), relerr(t.grad, td.grad))
    for n, q in m.named_parameters():
        worst = max(worst, relerr(q.grad, st64["match_fn." + n].grad))
    print(f"CrossAttention B={B} T={T} L={L} E={E} H={H} p={p}: sim err {e_sim:.2e}, worst gradient err {worst:.2e}")
    assert e_sim < 5e-6 and worst < 5e-5


def test_biencoder_with_cross_attention_train_step(dev):
    """The head inside the whole model: BiEncoder(Cnn8Rnn, EmbeddingAgg(512), match.CrossAttention(512, 8, 0.0)) -- one
    training step (dropout off) against the fp64 oracle of the same composition."""
    from texttoaudiogrounding_amd.models import audio_encoder, audio_text_model, match, text_encoder
    from texttoaudiogrounding_amd.runner import StrongRunner
    st = O.init_state(seed=13, logit_gain=1.0)
    batch = O.synthetic_batch(3, 64000, seed=4, ragged=True)
    torch.manual_seed(5)
    head = match.CrossAttention(512, 8, 0.0)
    model = audio_text_model.BiEncoder(audio_encoder.Cnn8Rnn(32000), text_encoder.EmbeddingAgg(5221, 512), head, 512)
    model.load_state_dict(st, strict=False)
    model.audio_encoder.dropout_p = (0.0, 0.0)
    runner = StrongRunner(model.train(), device=str(dev))
    loss = runner.forward_backward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    lv = runner.loss_value(loss)
    st64 = O.state_to(st, torch.float64, requires_grad=True)
    for k, v in head.state_dict().items():
        st64["match_fn." + k] = v.detach().cpu().double().requires_grad_(True)
    b64 = dict(batch)
    b64["waveform"], b64["label"] = batch["waveform"].double(), batch["label"].double()
    enc = O.cnn8rnn_forward(st64, b64["waveform"], b64["waveform_len"], training=True, p_drop=(0.0, 0.0))
    txt = O.embedding_agg_mean(st64, batch["text"], batch["text_len"])
    sim = O.match_cross_attention(st64, enc["embedding"], txt["token_emb"], batch["text_len"], 8)
    out = O.runner_truncate({"frame_sim": sim, "length": enc["length"]}, b64["label"])
    oloss = O.frame_bce_loss(out["frame_sim"], out["label"], out["length"])
    oloss.backward()
    print(f"BiEncoder + CrossAttention: loss {lv:.7f} vs oracle {oloss.item():.7f}")
    assert abs(lv - oloss.item()) < 2e-5
    for name in ("match_fn.attn.in_proj_weight", "match_fn.linear.weight", "match_fn.norm.bias",
                 "text_encoder.embedding.core.weight", "audio_encoder.rnn.weight_hh_l0", "audio_encoder.fc1.weight"):
        p = dict(model.named_parameters())[name]
        e = relerr(p.grad, st64[name].grad)
        print(f"  {name:45s} {e:.2e}")
        assert e < 1e-4, (name, e)


def test_cross_attention_empty_phrase_is_nan_and_long_phrase_is_rejected(dev):
    """a phrase with no valid token is a softmax over a fully masked row -- nn.MultiheadAttention (the reference,
    models/match.py:75-79) returns NaN for that clip, and so must the HIP head (loud, not a silent 0); the other clips are
    untouched.  A phrase longer than the kernel's compile-time token limit (32) raises instead of truncating."""
    from texttoaudiogrounding_amd.models import match
    torch.manual_seed(5)
    E, H, B, T, L = 128, 4, 3, 7, 4
    m = match.CrossAttention(E, H, 0.0).to(dev).eval()
    audio, token = torch.randn(B, T, E, device=dev), torch.randn(B, L, E, device=dev)
    sim = m({"audio_emb": audio, "text_emb": {"token_emb": token}, "text_len": torch.tensor([3, 0, 4])})
    ref = m({"audio_emb": audio, "text_emb": {"token_emb": token}, "text_len": torch.tensor([3, 2, 4])})
    assert torch.isnan(sim[1]).all() and torch.equal(sim[[0, 2]], ref[[0, 2]]) and torch.isfinite(ref).all()
    # the same through torch's own module on the CPU: fully masked row -> NaN
    cpu = torch.nn.MultiheadAttention(E, H, 0.0, batch_first=True).eval()
    mask = torch.arange(L)[None, :] >= torch.tensor([3, 0, 4])[:, None]
    with torch.no_grad():
        out, _ = cpu(audio.cpu(), token.cpu(), token.cpu(), key_padding_mask=mask)
    assert torch.isnan(out[1]).all() and torch.isfinite(out[0]).all()
    with pytest.raises(RuntimeError):
        m({"audio_emb": audio, "text_emb": {"token_emb": torch.randn(B, 33, E, device=dev)}, "text_len": torch.tensor([3, 2, 33])})
