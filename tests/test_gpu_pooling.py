"""Every similarity reducer of the reference (models/sim_pooling.py:6-204; models/utils.py *_with_lens as selected by
MultiTextBiEncoder.pooling) on the HIP path, against tests/golden/sim_pooling.npz -- outputs and input-gradients of the
IMPORTED reference classes in fp64 -- and at the sentence-level runner's shape (B = 64, T' = 250, 6 phrases)."""
import numpy as np
import pytest
import torch

from oracle import tag_oracle as O

pytestmark = pytest.mark.gpu

PAIR = ["AudioMeanTextMean", "AudioMeanTextSum", "AudioMaxTextMean", "AudioMaxTextMax", "AudioMaxTextSum",
        "AudioMaxTextMeanSum", "AudioLinearSoftTextMean", "AudioLinearSoftTextSum", "AudioExpSoftTextMean",
        "AudioExpSoftTextSum"]


def relerr(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


@pytest.mark.parametrize("name", PAIR)
def test_pair_reducers_golden(dev, golden_dir, name):
    from texttoaudiogrounding_amd.models import sim_pooling
    gold = np.load(f"{golden_dir}/sim_pooling.npz")
    sim = torch.from_numpy(gold["sim"]).float().to(dev).requires_grad_(True)
    out = getattr(sim_pooling, name)()({"sim": sim, "audio_len": gold["audio_len"], "text_len": gold["text_len"]})
    out.backward(torch.from_numpy(gold["dout"]).float().to(dev))
    e_o, e_g = relerr(out, gold[f"{name}/out"]), relerr(sim.grad, gold[f"{name}/dsim"])
    print(f"{name}: out {e_o:.2e} dsim {e_g:.2e}")
    assert e_o < 2e-6 and e_g < 5e-6


@pytest.mark.parametrize("name", ["MultiTextLinearSoft", "MultiTextMax"])
def test_multitext_reducers_golden(dev, golden_dir, name):
    from texttoaudiogrounding_amd.models import sim_pooling
    gold = np.load(f"{golden_dir}/sim_pooling.npz")
    sim = torch.from_numpy(gold["frame_sim"]).float().transpose(1, 2).contiguous().to(dev).requires_grad_(True)   # (B,n,T)
    out = getattr(sim_pooling, name)()({"sim": sim, "audio_len": gold["audio_len"]})
    out.backward(torch.from_numpy(gold["dclip"]).float().to(dev))
    assert relerr(out, gold[f"{name}/out"]) < 2e-6 and relerr(sim.grad, gold[f"{name}/dsim"]) < 5e-6


@pytest.mark.parametrize("mode", ["linear_softmax", "max", "mean", "exp_softmax"])
def test_multitext_biencoder_pooling_modes_golden(dev, golden_dir, mode):
    """The four pooling modes of MultiTextBiEncoder (models/audio_text_model.py:205-215) through ops.SimPoolFunction in the
    layout the model uses ((B*N, T') rows of frame scores)."""
    from texttoaudiogrounding_amd import ops
    gold = np.load(f"{golden_dir}/sim_pooling.npz")
    fs = torch.from_numpy(gold["frame_sim"]).float()                      # (B, T, N) as the reference pools it
    B, T, N = fs.shape
    rows = fs.transpose(1, 2).reshape(B * N, T, 1).contiguous().to(dev).requires_grad_(True)
    al = torch.from_numpy(gold["audio_len"]).long().to(dev)
    clip = ops.SimPoolFunction.apply(rows, al, None, N, 1, ops.POOL_MODES[mode], -1).view(B, N)
    clip.backward(torch.from_numpy(gold["dclip"]).float().to(dev))
    dfs = rows.grad.view(B, N, T).transpose(1, 2)
    assert relerr(clip, gold[f"pool_{mode}/out"]) < 2e-6 and relerr(dfs, gold[f"pool_{mode}/dsim"]) < 5e-6


@pytest.mark.parametrize("am,tm", [("exp_softmax", "mean"), ("max", "mean_sum"), ("linear_softmax", "sum")])
def test_pair_reducers_runner_shape_vs_oracle(dev, am, tm):
    """(64, 64, 250, 6): the matrix align.DotProduct hands to the reducers in the sentence-level runner."""
    from texttoaudiogrounding_amd import ops
    B, T, N = 64, 250, 6
    g = torch.Generator().manual_seed(3)
    sim = torch.rand(B, B, T, N, generator=g) * 0.98 + 0.01
    al = torch.randint(1, T + 1, (B,), generator=g); al[0] = T
    tl = torch.randint(1, N + 1, (B,), generator=g); tl[1] = N
    dout = torch.randn(B, B, generator=g)
    s = sim.to(dev).requires_grad_(True)
    out = ops.SimPoolFunction.apply(s.view(B * B, T, N), al.to(dev), tl.to(dev), B, B, ops.POOL_MODES[am], ops.TEXT_MODES[tm])
    out.view(B, B).backward(dout.to(dev))
    sd = sim.double().requires_grad_(True)
    ref = O.sim_pooling(sd, al, tl, am, tm)
    ref.backward(dout.double())
    assert relerr(out.view(B, B), ref) < 5e-6 and relerr(s.grad, sd.grad) < 2e-5


def test_attention_pooling_golden(dev, golden_dir):
    """EmbeddingAgg(aggregation="attention"): AttentionPooling forward/backward vs the imported reference (fp64)."""
    from texttoaudiogrounding_amd import ops
    gold = np.load(f"{golden_dir}/sim_pooling.npz")
    x = torch.from_numpy(gold["attnpool/x"]).float().to(dev).requires_grad_(True)
    w = torch.from_numpy(gold["attnpool/w"]).float().to(dev).requires_grad_(True)
    b = torch.from_numpy(gold["attnpool/b"]).float().to(dev).requires_grad_(True)
    out = ops.AttnPoolFunction.apply(x, torch.from_numpy(gold["attnpool/lens"]).long().to(dev), w, b)
    out.backward(torch.from_numpy(gold["attnpool/dout"]).float().to(dev))
    e = (relerr(out, gold["attnpool/out"]), relerr(x.grad, gold["attnpool/dx"]), relerr(w.grad, gold["attnpool/dw"]))
    db, db_ref = b.grad.reshape(-1)[0].item(), np.asarray(gold["attnpool/db"]).reshape(-1)[0].item()
    print(f"AttentionPooling: out {e[0]:.2e} dx {e[1]:.2e} dw {e[2]:.2e}; db {db:.2e} (reference {db_ref:.2e})")
    assert max(e) < 5e-6 and abs(db - db_ref) < 1e-5


def test_embedding_agg_attention_module(dev):
    from texttoaudiogrounding_amd.models import text_encoder
    torch.manual_seed(1)
    m = text_encoder.EmbeddingAgg(300, 128, aggregation="attention").to(dev)
    assert [n for n, _ in m.named_parameters()] == ["embedding.core.weight", "attn.fc.weight", "attn.fc.bias"]
    text = torch.randint(2, 300, (6, 4))
    lens = torch.tensor([4, 1, 2, 3, 4, 2])
    out = m({"text": text, "text_len": lens})
    (out["seq_emb"].sum() + 0.5 * out["token_emb"].sum()).backward()
    st = {k: v.detach().cpu().double().requires_grad_(True) for k, v in m.named_parameters()}
    tok = st["embedding.core.weight"][text]
    ref = O.attention_pooling(tok, lens, st["attn.fc.weight"], st["attn.fc.bias"])
    (ref.sum() + 0.5 * tok.sum()).backward()
    assert relerr(out["seq_emb"], ref) < 2e-6
    for k, p in m.named_parameters():
        if k != "attn.fc.bias":
            assert relerr(p.grad, st[k].grad) < 1e-5, k


@pytest.mark.parametrize("R,T,ratio", [(3, 7, 4), (64, 250, 4), (2, 1, 4), (5, 33, 2)])
def test_upsample_linear_vs_torch(dev, R, T, ratio):
    """BiEncoder(upsample=True): F.interpolate(mode="linear", align_corners=False) -- torch itself is the oracle."""
    from texttoaudiogrounding_amd import ops
    g = torch.Generator().manual_seed(R + T)
    x = torch.rand(R, T, generator=g)
    dout = torch.randn(R, T * ratio, generator=g)
    xs = x.to(dev).requires_grad_(True)
    out = ops.UpsampleLinearFunction.apply(xs, ratio)
    out.backward(dout.to(dev))
    xd = x.double().requires_grad_(True)
    ref = O.upsample_linear(xd, ratio)
    ref.backward(dout.double())
    assert out.shape == (R, T * ratio) and relerr(out, ref) < 1e-6 and relerr(xs.grad, xd.grad) < 1e-6


def test_audio_text_align_by_word(dev):
    """AudioTextAlignByWord (models/audio_text_model.py:843-904): Cnn8Rnn frames x word embeddings (256 -> projections to
    128) -> align.DotProduct (B,B,T',n_word) -> AudioExpSoftTextMean -> (B,B) -> MaxMarginRankingLoss, vs the fp64 oracle."""
    from texttoaudiogrounding_amd.models import align, audio_encoder, audio_text_model, sim_pooling, text_encoder
    from texttoaudiogrounding_amd.losses import MaxMarginRankingLoss
    torch.manual_seed(9)
    st = O.init_state(seed=2, text_dim=256, shared_dim=128, logit_gain=1.0)
    model = audio_text_model.AudioTextAlignByWord(audio_encoder.Cnn8Rnn(32000), text_encoder.EmbeddingAgg(5221, 256),
                                                  align.DotProduct(l2norm=True, scaled=False), sim_pooling.AudioExpSoftTextMean(), 128)
    assert hasattr(model, "audio_proj")
    sd = {k: v for k, v in st.items() if not k.startswith(("audio_proj", "text_proj"))}
    model.load_state_dict(sd, strict=False)
    model.audio_encoder.dropout_p = (0.0, 0.0)
    model = model.to(dev).train()
    batch = O.synthetic_batch(4, 48000, seed=6, ragged=True)
    out = model({"waveform": batch["waveform"].to(dev), "waveform_len": batch["waveform_len"], "text": batch["text"].to(dev),
                 "text_len": batch["text_len"], "specaug": False})
    loss = MaxMarginRankingLoss(margin=0.2)(out)
    loss.backward()
    st64 = O.state_to({k: v.detach().cpu() for k, v in model.state_dict().items() if "melspec" not in k}, torch.float64,
                      requires_grad=True)
    enc = O.cnn8rnn_forward(st64, batch["waveform"].double(), batch["waveform_len"], training=True, p_drop=(0.0, 0.0))
    a = torch.nn.functional.linear(enc["embedding"], st64["audio_proj.weight"], st64["audio_proj.bias"])
    tok = st64["text_encoder.embedding.core.weight"][batch["text"]]
    w = torch.nn.functional.linear(tok, st64["text_proj.weight"], st64["text_proj.bias"])
    sim_m = O.align_dot_product(a, w, l2norm=True, scaled=False)
    ref = O.sim_pooling(sim_m, enc["length"], batch["text_len"], "exp_softmax", "mean")
    rloss = O.max_margin_ranking_loss(ref, margin=0.2)
    rloss.backward()
    print(f"AlignByWord: sim err {relerr(out['sim'], ref):.2e}; loss {loss.item():.6f} vs {rloss.item():.6f}")
    assert relerr(out["sim"], ref) < 1e-5 and abs(loss.item() - rloss.item()) < 1e-5
    for name in ("text_proj.weight", "audio_proj.bias", "text_encoder.embedding.core.weight", "audio_encoder.fc1.weight"):
        assert relerr(dict(model.named_parameters())[name].grad, st64[name].grad) < 2e-4, name
