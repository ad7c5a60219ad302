"""Inference path (BASELINE config 5: models/hf_modeling_grounding.py in the reference):
LAION-CLAP text tower kernels against the Hugging Face golden vectors and the CPU restatement, and the whole
Cnn8Rnn + CLAP + projections + DotProduct model on 30 s clips against the CPU oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import clap_text_oracle as C
from oracle import tag_oracle as O

pytestmark = pytest.mark.gpu


def build_text_encoder(st, cfg, dev):
    from texttoaudiogrounding_amd.models.hf_modeling_grounding import LaionClapEncoder
    enc = LaionClapEncoder(config=cfg)
    missing = enc.load_state_dict(st, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return enc.to(dev).eval()


def test_clap_text_tower_golden(dev, golden_dir):
    """HIP text tower == real transformers ClapTextModel + ClapProjectionLayer outputs (tiny config fixture)."""
    g = np.load(f"{golden_dir}/clap_text_tiny.npz")
    st = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w/")}
    D = st["model.embeddings.word_embeddings.weight"].shape[1]
    cfg = dict(vocab_size=st["model.embeddings.word_embeddings.weight"].shape[0], hidden_size=D,
               num_hidden_layers=2, num_attention_heads=int(g["n_heads"]),
               intermediate_size=st["model.encoder.layer.0.intermediate.dense.weight"].shape[0],
               max_position_embeddings=st["model.embeddings.position_embeddings.weight"].shape[0],
               projection_dim=st["projection.linear2.weight"].shape[0], layer_norm_eps=float(g["eps"]))
    enc = build_text_encoder(st, cfg, dev)
    out = enc({"input_ids": torch.from_numpy(g["input_ids"]), "attention_mask": torch.from_numpy(g["attention_mask"])})
    for key in ("last_hidden_state", "pooler_output", "token_emb", "seq_emb"):
        err = np.abs(out[key].cpu().numpy() - g[key]).max()
        print(f"clap tiny {key}: max abs err {err:.2e} (|ref| max {np.abs(g[key]).max():.2f})")
        assert err < 2e-5 * max(1.0, np.abs(g[key]).max()), key


def test_clap_text_tower_roberta_base_shape(dev):
    """RoBERTa-base shape (12 x 768, FFN 3072, vocab 50265, projection 512), seeded random weights, ragged phrases."""
    st = C.init_text_state(seed=5)
    ids, mask = C.synthetic_tokens(6, 12, seed=9)
    enc = build_text_encoder(st, None, dev)
    out = enc({"input_ids": ids, "attention_mask": mask})
    st64 = {k: v.double() for k, v in st.items()}
    ref = C.laion_clap_encoder_forward(st64, ids, mask, 12, 1e-12)
    e_seq = (out["seq_emb"].cpu().double() - ref["seq_emb"]).abs().max().item()
    e_tok = (out["token_emb"].cpu().double() - ref["token_emb"]).abs().max().item() / ref["token_emb"].abs().max().item()
    print(f"clap base-shape: seq_emb err {e_seq:.2e}, token_emb rel err {e_tok:.2e}")
    assert e_seq < 1e-5 and e_tok < 1e-5          # unit-norm 512-vector: 1e-5 per component


def test_grounding_model_30s_clips_vs_oracle(dev):
    """Cnn8RnnLaionClapGroundingModel.forward on 30 s clips (T' = 750), B = 3 processed in passes of 2: frame_sim within
    1e-4 of the CPU oracle (log-mel -> Cnn8Rnn eval -> audio_proj; CLAP tower -> text_proj; DotProduct) and
    bit-exact integer segments."""
    from texttoaudiogrounding_amd.models.hf_modeling_grounding import Cnn8RnnLaionClapGroundingModel
    from texttoaudiogrounding_amd.utils import eval_util
    S = 960000
    st_a = O.init_state(seed=31, add_proj=True, shared_dim=512, logit_gain=40.0)
    st_t = C.init_text_state(seed=6)
    st_a["audio_proj.weight"] = st_a["audio_proj.weight"] * 8.0        # spread the logits over several units
    st_a["text_proj.weight"] = st_a["text_proj.weight"] * 30.0
    batch = O.synthetic_batch(3, S, seed=41, ragged=True)
    ids, mask = C.synthetic_tokens(3, 10, seed=2)
    model = Cnn8RnnLaionClapGroundingModel(max_clips_per_pass=2)
    sd = {}
    for k, v in st_a.items():
        if k.startswith("text_encoder."):
            continue
        sd["model." + k] = v
    sd.update({"model.text_encoder." + k: v for k, v in st_t.items()})
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing
    assert all("melspec" in k or k.endswith(("position_ids", "token_type_ids")) for k in missing.missing_keys), missing
    model = model.to(dev)
    # calibrate the running statistics (momentum 1 on one training-mode pass of the audio encoder), then eval
    enc = model.model.audio_encoder
    enc.train()
    enc.dropout_p = (0.0, 0.0)
    for m in enc.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 1.0
    with torch.no_grad():
        enc({"waveform": batch["waveform"][:2].to(dev), "waveform_len": batch["waveform_len"][:2], "specaug": False})
    model.eval()
    fs = model(batch["waveform"], batch["waveform_len"], {"input_ids": ids, "attention_mask": mask}).cpu()
    assert fs.shape == (3, 750)
    # ---- CPU oracle on the same (calibrated) weights ----
    st2 = {k[len("model."):]: v.detach().cpu() for k, v in model.state_dict().items() if "melspec" not in k}
    audio = O.cnn8rnn_forward({k: v for k, v in st2.items() if k.startswith("audio_encoder.")}, batch["waveform"],
                              batch["waveform_len"], training=False)["embedding"]
    a = F.linear(audio, st2["audio_proj.weight"], st2["audio_proj.bias"])
    tx = C.laion_clap_encoder_forward({k[len("text_encoder."):]: v for k, v in st2.items() if k.startswith("text_encoder.")},
                                      ids, mask, 12, 1e-12)
    t = F.linear(tx["seq_emb"], st2["text_proj.weight"], st2["text_proj.bias"])
    ref = O.match_dot_product(a, t)
    err = (fs - ref).abs().max().item()
    print(f"30 s clips: frame_sim err {err:.2e}; range [{ref.min():.3f}, {ref.max():.3f}]")
    assert err < 1e-4
    th = eval_util.eval_thresholds(50)
    got = eval_util.segments_for_thresholds(fs.to(dev), th, 1, eval_util.n_connect_for(0.04))
    for b in range(3):
        for ti, tt in enumerate(th):
            assert np.array_equal(got[b][ti], O.segments(fs[b].numpy(), tt, 1, 13))


def test_grounding_model_30s_full_pass_b67(dev):
    """BASELINE configs[4] at its real pass size (models/hf_modeling_grounding.py:319-352 in the reference): 30 s clips,
    B = 67 = ONE FULL 64-clip pass (3.15 GB first-conv output, the 32-bit activation-offset guard's regime) + a ragged
    3-clip remainder.  Asserted: shape / finiteness, `length` semantics (frames beyond a clip's length are still scored,
    like the reference, and the valid count is floor((len // hop + 1) / 4)), the pass split is invisible (the same clips run
    alone give bit-identical rows), and 3 sampled clips (first of the full pass, last of it, last of the remainder) agree
    with the CPU oracle to 1e-4 with bit-exact integer segments at the 50 thresholds."""
    from texttoaudiogrounding_amd.models.hf_modeling_grounding import Cnn8RnnLaionClapGroundingModel
    from texttoaudiogrounding_amd.utils import eval_util
    S, B = 960000, 67
    st_a = O.init_state(seed=31, add_proj=True, shared_dim=512, logit_gain=40.0)
    st_t = C.init_text_state(seed=6)
    st_a["audio_proj.weight"] = st_a["audio_proj.weight"] * 8.0
    st_a["text_proj.weight"] = st_a["text_proj.weight"] * 30.0
    g = torch.Generator().manual_seed(4101)
    wave = 0.1 * torch.randn(B, S, generator=g)
    lens = torch.randint(S // 2, S + 1, (B,), generator=g)
    lens[0] = S
    for i in range(B):
        wave[i, lens[i]:] = 0.0
    ids, mask = C.synthetic_tokens(B, 10, seed=2)
    model = Cnn8RnnLaionClapGroundingModel()          # default max_clips_per_pass = 64
    sd = {"model." + k: v for k, v in st_a.items() if not k.startswith("text_encoder.")}
    sd.update({"model.text_encoder." + k: v for k, v in st_t.items()})
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing
    model = model.to(dev)
    enc = model.model.audio_encoder
    enc.train()
    enc.dropout_p = (0.0, 0.0)
    for m in enc.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 1.0
    with torch.no_grad():          # calibrate the running statistics on two clips, then eval
        enc({"waveform": wave[:2].to(dev), "waveform_len": lens[:2], "specaug": False})
    model.eval()
    text = {"input_ids": ids, "attention_mask": mask}
    torch.cuda.reset_peak_memory_stats()
    fs_dev = model(wave, lens, text)
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    fs = fs_dev.cpu()
    assert fs.shape == (B, 750) and torch.isfinite(fs).all()
    assert fs.min() >= 1e-7 and fs.max() <= 1.0                      # sigmoid().clamp(1e-7, 1.0)
    want_len = (lens // 320 + 1) // 4
    out_len = model.model.audio_encoder({"waveform": wave[64:].to(dev), "waveform_len": lens[64:], "specaug": False})["length"]
    assert torch.equal(out_len.cpu(), want_len[64:])
    # the split into passes is invisible: clips from both passes, run alone in a different pass composition
    pick = [0, 63, 64, 66]
    alone = model(wave[pick], lens[pick], {"input_ids": ids[pick], "attention_mask": mask[pick]}).cpu()
    assert torch.equal(alone, fs[pick]), (alone - fs[pick]).abs().max()
    # ---- CPU oracle on three sampled clips (eval-mode BatchNorm: clips are independent) ----
    st2 = {k[len("model."):]: v.detach().cpu() for k, v in model.state_dict().items() if "melspec" not in k}
    samp = [0, 63, 66]
    audio = O.cnn8rnn_forward({k: v for k, v in st2.items() if k.startswith("audio_encoder.")}, wave[samp],
                              lens[samp].numpy(), training=False)["embedding"]
    a = F.linear(audio, st2["audio_proj.weight"], st2["audio_proj.bias"])
    tx = C.laion_clap_encoder_forward({k[len("text_encoder."):]: v for k, v in st2.items() if k.startswith("text_encoder.")},
                                      ids[samp], mask[samp], 12, 1e-12)
    t = F.linear(tx["seq_emb"], st2["text_proj.weight"], st2["text_proj.bias"])
    ref = O.match_dot_product(a, t)
    err = (fs[samp] - ref).abs().max().item()
    print(f"30 s clips, B = 67 (64 + 3): peak memory {peak:.1f} GiB; frame_sim err on 3 sampled clips {err:.2e}; "
          f"range [{ref.min():.3f}, {ref.max():.3f}]")
    assert err < 1e-4
    th = eval_util.eval_thresholds(50)
    got = eval_util.segments_for_thresholds(fs_dev[samp].contiguous(), th, 1, eval_util.n_connect_for(0.04))
    mism = 0
    for j in range(3):
        for ti, tt in enumerate(th):
            assert np.array_equal(got[j][ti], O.segments(fs[samp[j]].numpy(), tt, 1, 13))     # kernel == oracle, same scores
            mism += int(not np.array_equal(got[j][ti], O.segments(ref[j].numpy(), tt, 1, 13)))
    print(f"segments vs the oracle's own scores: {mism} of {3 * len(th)} (clip, threshold) pairs differ")
    assert mism == 0


def test_hf_surface_round_trip_and_string_prompts(dev, tmp_path):
    """The reference's entry point as the README uses it (models/hf_modeling_grounding.py:305-352; README.md:7-39): the model is
    saved with save_pretrained, loaded back with from_pretrained (and through AutoModel), and called with STRINGS -- tokenised by
    a tokenizer that resolves locally (a tiny word-level one built in the test: there is no network for the CLAP tokenizer).
    frame_sim is bit-identical across the round trip, and the string call equals the call with the tokenizer's output."""
    transformers = pytest.importorskip("transformers")
    from tests.test_hf_surface import TINY, tiny_tokenizer_dir
    from texttoaudiogrounding_amd.models.hf_modeling_grounding import (Cnn8RnnLaionClapGroundingConfig,
                                                                      Cnn8RnnLaionClapGroundingModel)
    tok_dir = tiny_tokenizer_dir(str(tmp_path / "tok"))
    torch.manual_seed(12)
    cfg = Cnn8RnnLaionClapGroundingConfig(text_encoder_name=tok_dir, text_config=TINY)
    model = Cnn8RnnLaionClapGroundingModel(cfg)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_var.uniform_(0.5, 1.5)
                m.running_mean.uniform_(-0.2, 0.2)
    model = model.to(dev).eval()
    g = torch.Generator().manual_seed(5)
    wave = 0.1 * torch.randn(3, 64000, generator=g)
    lens = [64000, 50000, 33333]
    text = ["a man speaks", "the dog is barking loudly", "rain falls on the roof"]
    fs = model(audio=wave, audio_len=lens, text=text)                      # the README's call
    assert fs.shape == (3, 50) and torch.isfinite(fs).all()
    tokens = model.text_tokenizer(text, padding=True, return_tensors="pt", truncation=True)
    assert torch.equal(fs, model(wave, lens, tokens))
    model.save_pretrained(tmp_path / "ckpt")
    for loader in (Cnn8RnnLaionClapGroundingModel, transformers.AutoModel):
        back = loader.from_pretrained(tmp_path / "ckpt").to(dev).eval()
        assert back.text_tokenizer is not None
        assert torch.equal(back(audio=wave, audio_len=lens, text=text), fs)
    model.text_tokenizer = None
    with pytest.raises(RuntimeError, match="no tokenizer"):
        model(wave, lens, text)
