"""BASELINE configs[3]: cross-encoder (models/cross_encoder.py in the reference) + token-level DotProduct on the HIP path,
against the golden vectors the imported reference produced (tests/golden/cross_encoder.npz, fp64 twin) and, at the
BASELINE shape (B=64, T'=250, D=512), against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import tag_oracle as O

pytestmark = pytest.mark.gpu


def build(st, D, dev):
    from texttoaudiogrounding_amd.models.cross_encoder import CrossAttentionGating
    ce = CrossAttentionGating(D)
    ce.load_state_dict({k[len("cross_encoder."):]: v for k, v in st.items()})
    return ce.to(dev)


def run(ce, audio, token, audio_len, text_len, dsim, dev):
    from texttoaudiogrounding_amd.models.match import DotProduct
    a = audio.to(dev).requires_grad_(True)
    t = token.to(dev).requires_grad_(True)
    enc = ce({"audio_emb": a, "text_emb": {"token_emb": t}, "audio_len": audio_len, "text_len": text_len})
    sim = DotProduct(text_level="token")({"audio_emb": enc["audio_emb"], "text_emb": enc["text_emb"]})
    sim.backward(dsim.to(dev))
    return enc, sim, a.grad, t.grad


def rel(got, want):
    want = torch.as_tensor(want).double()
    return (got.detach().cpu().double() - want).abs().max().item() / (want.abs().max().item() + 1e-30)


def test_cross_encoder_golden(dev, golden_dir):
    g = np.load(f"{golden_dir}/cross_encoder.npz")
    st = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w/")}
    D = g["audio"].shape[-1]
    ce = build(st, D, dev)
    enc, sim, da, dt = run(ce, torch.from_numpy(g["audio"]), torch.from_numpy(g["token"]), torch.from_numpy(g["audio_len"]),
                           torch.from_numpy(g["text_len"]), torch.from_numpy(g["dsim"]), dev)
    errs = {"audio_out": rel(enc["audio_emb"], g["audio_out_f64"]), "text_out": rel(enc["text_emb"]["token_emb"], g["text_out_f64"]),
            "sim": rel(sim, g["sim_f64"]), "daudio": rel(da, g["daudio_f64"]), "dtoken": rel(dt, g["dtoken_f64"])}
    for n, p in ce.named_parameters():
        errs["grad " + n] = rel(p.grad, g[f"grad_f64/cross_encoder.{n}"])
    ref32 = max(np.abs(g["daudio_f32"].astype(np.float64) - g["daudio_f64"]).max() / np.abs(g["daudio_f64"]).max(),
                np.abs(g["sim_f32"].astype(np.float64) - g["sim_f64"]).max())
    print("cross-encoder golden:", {k: f"{v:.1e}" for k, v in errs.items()}, f"(reference's own f32 vs f64: {ref32:.1e})")
    assert all(v < 2e-5 for v in errs.values()), errs


def test_cross_encoder_baseline_shape_vs_oracle(dev):
    """B=64, T'=250, D=512, phrases of 1..6 tokens, ragged audio lengths; forward and every gradient vs the fp64 oracle."""
    B, T, L, D = 64, 250, 6, 512
    st = O.init_cross_state(seed=3, dim=D, scale=2.0)
    g = torch.Generator().manual_seed(8)
    audio = torch.randn(B, T, D, generator=g) * 0.7
    token = torch.randn(B, L, D, generator=g) * 0.7
    audio_len = torch.randint(120, T + 1, (B,), generator=g)
    text_len = torch.randint(1, L + 1, (B,), generator=g)
    dsim = torch.randn(B, T, generator=g) * (torch.arange(T)[None, :] < audio_len[:, None])      # loss mask
    ce = build(st, D, dev)
    enc, sim, da, dt = run(ce, audio, token, audio_len, text_len, dsim, dev)
    st64 = {k: v.double().requires_grad_(True) for k, v in st.items()}
    a64, t64 = audio.double().requires_grad_(True), token.double().requires_grad_(True)
    ao, to = O.cross_attention_gating(st64, a64, t64, audio_len, text_len)
    so = O.match_dot_product_token(ao, to)
    so.backward(dsim.double())
    errs = {"sim": (sim.cpu().double() - so.detach()).abs().max().item(), "audio_out": rel(enc["audio_emb"], ao.detach()),
            "text_out": rel(enc["text_emb"]["token_emb"], to.detach()), "daudio": rel(da, a64.grad), "dtoken": rel(dt, t64.grad)}
    for n, p in ce.named_parameters():
        errs["grad " + n] = rel(p.grad, st64["cross_encoder." + n].grad)
    print("cross-encoder B=64:", {k: f"{v:.1e}" for k, v in errs.items()}, f"sim range [{so.min().item():.3f}, {so.max().item():.3f}]")
    assert errs["sim"] < 1e-5 and all(v < 5e-5 for v in errs.values()), errs


def test_cross_encoder_whole_train_step(dev):
    """BiEncoder(Cnn8Rnn, EmbeddingAgg(512), DotProduct(text_level="token"), cross_encoder=CrossAttentionGating(512)):
    one training step (train-mode BN, dropout off) through StrongRunner vs the fp64 oracle: loss, frame_sim, gradients of
    the cross-encoder, the embedding table (token path) and the audio encoder's last layers."""
    from texttoaudiogrounding_amd.models import audio_encoder, audio_text_model, match, text_encoder
    from texttoaudiogrounding_amd.models.cross_encoder import CrossAttentionGating
    from texttoaudiogrounding_amd.runner import StrongRunner
    st = O.init_state(seed=13, logit_gain=1.0)
    st.update(O.init_cross_state(seed=4, dim=512, scale=2.0))
    st["text_encoder.embedding.core.weight"] = st["text_encoder.embedding.core.weight"] * 8.0
    batch = O.synthetic_batch(2, 48000, seed=77, ragged=True)
    model = audio_text_model.BiEncoder(audio_encoder.Cnn8Rnn(32000), text_encoder.EmbeddingAgg(5221, 512),
                                       match.DotProduct(text_level="token"), 512, cross_encoder=CrossAttentionGating(512))
    missing = model.load_state_dict(st, strict=False)
    assert not missing.unexpected_keys and all("melspec" in k for k in missing.missing_keys), missing
    model = model.to(dev).train()
    model.audio_encoder.dropout_p = (0.0, 0.0)
    runner = StrongRunner(model, device=str(dev))
    loss = runner.forward_backward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    # ---- oracle (fp64) ----
    s64 = O.state_to(st, torch.float64, requires_grad=True)
    wav = batch["waveform"].double()
    ao = O.cnn8rnn_forward(s64, wav, batch["waveform_len"], training=True, p_drop=(0.0, 0.0))
    tok = s64["text_encoder.embedding.core.weight"][batch["text"]]
    u, s_ = O.cross_attention_gating(s64, ao["embedding"], tok, ao["length"], batch["text_len"])
    out = O.runner_truncate({"frame_sim": O.match_dot_product_token(u, s_), "length": ao["length"]}, batch["label"].double())
    oloss = O.frame_bce_loss(out["frame_sim"], out["label"], out["length"])
    oloss.backward()
    print(f"cross-encoder train step: loss {loss.item():.7f} vs {oloss.item():.7f}")
    assert abs(loss.item() - oloss.item()) < 2e-5
    worst = 0.0
    for name, p in model.named_parameters():
        if not (name.startswith("cross_encoder") or name.startswith("text_encoder") or "rnn" in name or "fc1" in name):
            continue                                    # conv/BN gradients at B=2 carry decision-flip noise
        e = rel(p.grad, s64[name].grad)
        worst = max(worst, e)
        assert e < 1e-4, (name, e)
    print(f"  worst gradient error (cross-encoder, embedding table, fc1, GRU) {worst:.1e}")
