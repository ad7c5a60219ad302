#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the IMPORTED REFERENCE
(/root/reference, stubbed per SURVEY.md section 8c) in the build container.

    python tests/golden/make_golden.py

Fixtures are data only: inputs are regenerated from seeds by oracle.tag_oracle
(``synthetic_batch`` / ``init_state``; an input checksum is stored to catch RNG drift),
expected outputs are stored by value.  While generating, the oracle is checked against
the reference and the script aborts if they disagree.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_import  # noqa: E402
from oracle import tag_oracle as O  # noqa: E402

torch.set_num_threads(8)
mods = ref_import.install()
AE, TE, MA, AL = (mods["models.audio_encoder"], mods["models.text_encoder"], mods["models.match"],
                  mods["models.align"])
ATM, LOSS, EU = mods["models.audio_text_model"], mods["losses"], mods["utils.eval_util"]

S = 48000          # F = 151 -> 75 -> 37 (odd at two pooling stages)
B = 2
SAMPLE_IDX_SEED = 99


def checksum(t):
    t = t.detach().double().flatten()
    return [float(t.sum()), float(t.abs().max()), float(t[:: max(1, t.numel() // 7)][:7].sum())]


def grad_summary(named_grads):
    """per tensor: l2, maxabs, 16 sampled entries (fixed pseudo-random indices)."""
    out = {}
    for k, g in named_grads.items():
        g = g.detach().double().flatten()
        gi = torch.Generator().manual_seed(SAMPLE_IDX_SEED)
        idx = torch.randint(0, g.numel(), (16,), generator=gi)
        out[k] = np.concatenate([[g.norm().item(), g.abs().max().item()], g[idx].numpy()])
    return out


def make_batch(hop, ragged=True):
    b = O.synthetic_batch(B, S, seed=1234, ragged=False, hop=hop)
    if ragged:
        lens = np.array([S, S - 5 * hop * 4 - 123])
        b["waveform"][1, lens[1]:] = 0.0
        b["waveform_len"] = lens
    return b


def calibrate_running_stats(st, batch, audio):
    """Set every BN's running stats to the batch statistics so eval mode is well-conditioned."""
    st2 = O.state_to(st)
    bn_prefixes = sorted({k[: -len("running_mean")] for k in st2 if k.endswith("running_mean")})
    # momentum=1 pass: running <- batch stats (unbiased var), done via F.batch_norm momentum
    import torch.nn.functional as F
    orig = O._bn

    def bn_m1(x, s, prefix, training, momentum=0.1, eps=1e-5):
        return orig(x, s, prefix, True, 1.0, eps)

    O._bn = bn_m1
    try:
        with torch.no_grad():
            O.biencoder_forward(st2, batch, "dot" if "audio_proj.weight" not in st2 else "expnegl2",
                                audio, training=True, p_drop=(0.0, 0.0) if audio == "cnn8rnn" else 0.0)
    finally:
        O._bn = orig
    for p in bn_prefixes:
        st[p + "running_mean"] = st2[p + "running_mean"].clone()
        st[p + "running_var"] = st2[p + "running_var"].clone()
    return st


def build_reference(st, audio, match):
    text_dim = st["text_encoder.embedding.core.weight"].shape[1]
    if audio == "cnn8rnn":
        ae = AE.Cnn8Rnn(sample_rate=32000)
    else:
        ae = AE.CrnnEncoder(sample_rate=32000, embed_dim=256)
    te = TE.EmbeddingAgg(vocab_size=5221, embed_dim=text_dim, aggregation="mean")
    mf = MA.DotProduct() if match == "dot" else MA.ExpNegL2()
    model = ATM.BiEncoder(ae, te, mf, shared_dim=256 if "audio_proj.weight" in st else text_dim)
    missing = model.load_state_dict(st, strict=True)
    return model


class DropoutReplay:
    """Replace torch.nn.functional.dropout with a recorded-mask version while the reference runs."""

    def __init__(self, seed=None, off=False):
        self.seed, self.off, self.masks = seed, off, []

    def __enter__(self):
        import torch.nn.functional as F
        self.F = F
        self.orig = F.dropout
        g = torch.Generator().manual_seed(self.seed or 0)

        def drop(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0 or self.off:
                return x
            m = (torch.rand(x.shape, generator=g) >= p)
            self.masks.append(m)
            return x * m.to(x.dtype) / (1.0 - p)

        F.dropout = drop
        return self

    def __exit__(self, *a):
        self.F.dropout = self.orig


def run_reference_step(model, batch, training, dtype=torch.float32, dropout=None):
    model = model.to(dtype)
    model.train(training)
    model.zero_grad()
    bd = {"waveform": batch["waveform"].to(dtype), "waveform_len": batch["waveform_len"],
          "text": batch["text"], "text_len": batch["text_len"], "specaug": False}
    hooks, taps = [], {}
    hooks.append(model.audio_encoder.register_forward_hook(lambda m, i, o: taps.__setitem__("embedding", o["embedding"])))
    with (dropout or DropoutReplay(off=True)) as dr:
        out = model(bd)
        # Runner.forward label alignment (run_strong.py:107-118), restated inline for the reference side
        label = batch["label"].to(dtype)
        fs = out["frame_sim"]
        tt = min(fs.size(1), label.size(1))
        o2 = {"frame_sim": fs[..., :tt], "label": label[..., :tt],
              "length": torch.clamp(out["length"], 1, tt)}
        loss = LOSS.FrameBceLoss()(dict(o2))
        grads = {}
        if training:
            loss.backward()
            grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    for h in hooks:
        h.remove()
    return out, loss, grads, taps, dr.masks


def case_whole_path(name, audio, match, text_dim, training, dropout_seed=None, save=None):
    hop = 320 if audio == "cnn8rnn" else 640
    batch = make_batch(hop)
    if audio == "cnn8rnn":
        st = O.init_state(seed=7, text_dim=text_dim, shared_dim=256 if text_dim != 512 else 512,
                          logit_gain=120.0)
    else:
        st = O.init_crnn_state(seed=7)
        g = torch.Generator().manual_seed(8)
        st["text_encoder.embedding.core.weight"] = (torch.rand(5221, 256, generator=g) * 2 - 1) * 0.9
        for k in list(st):
            if k.endswith(".0.weight"):
                st[k] = 0.5 + torch.rand(st[k].shape, generator=g)
            if k.endswith(".0.bias"):
                st[k] = 0.2 * torch.randn(st[k].shape, generator=g)
    st = calibrate_running_stats(st, batch, audio)

    res = {}
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        st_d = O.state_to(st, dtype)
        model = build_reference(st_d, audio, match)
        dr = DropoutReplay(seed=dropout_seed) if dropout_seed is not None else None
        out, loss, grads, taps, masks = run_reference_step(model, batch, training, dtype, dr)
        # ---- oracle vs reference (abort on disagreement) ----
        st_o = O.state_to(st, dtype, requires_grad=training)
        bo = dict(batch)
        bo["waveform"] = batch["waveform"].to(dtype)
        bo["label"] = batch["label"].to(dtype)
        mk = None
        if masks:
            names = [f"drop{i}" for i in range(1, 6)] if audio == "cnn8rnn" else ["drop"]
            mk = dict(zip(names, masks))
        otaps = {}
        p_drop = None if masks else ((0.0, 0.0) if audio == "cnn8rnn" else 0.0)
        oloss, oout = O.train_step_loss(st_o, bo, match, audio, training, p_drop, mk, otaps)
        tol = 2e-5 if dtype == torch.float32 else 1e-10
        d_fs = (oout["frame_sim"] - out["frame_sim"][..., : oout["frame_sim"].shape[1]]).abs().max().item()
        d_emb = (otaps["audio_emb"] if "audio_proj.weight" not in st else None)
        print(f"[{name}/{tag}] frame_sim diff {d_fs:.3e}  loss ref {loss.item():.8f} oracle {oloss.item():.8f}")
        assert d_fs < tol, "oracle != reference"
        assert abs(loss.item() - oloss.item()) < tol * 10
        assert torch.equal(oout["length"], torch.clamp(out["length"], 1, oout["frame_sim"].shape[1]))
        if "audio_proj.weight" not in st:
            de = (otaps["audio_emb"] - taps["embedding"]).abs().max().item()
            print(f"    embedding diff {de:.3e} (max {taps['embedding'].abs().max().item():.3f})")
            assert de < tol * 10
        if training:
            oloss.backward()
            worst = 0.0
            for k, gref in grads.items():
                go = st_o[k].grad
                rel = (go - gref).abs().max().item() / (gref.abs().max().item() + 1e-30)
                worst = max(worst, rel)
            print(f"    worst per-tensor rel grad diff oracle-vs-reference: {worst:.3e}")
            assert worst < (5e-3 if dtype == torch.float32 else 1e-8)
        # running stats after a training step
        res[tag] = dict(out=out, loss=loss, grads=grads, taps=taps, masks=masks,
                        state_after=model.state_dict())
    # ---- store ----
    f32, f64 = res["f32"], res["f64"]
    store = {
        "input_checksum": np.array(checksum(batch["waveform"]) + checksum(batch["text"].float())
                                   + checksum(st["audio_encoder.fc1.weight"] if audio == "cnn8rnn"
                                              else st["audio_encoder.gru.weight_hh_l0"])),
        "waveform_len": np.asarray(batch["waveform_len"]),
        "frame_sim_f32": f32["out"]["frame_sim"].detach().numpy(),
        "frame_sim_f64": f64["out"]["frame_sim"].detach().numpy(),
        "length": f32["out"]["length"].numpy(),
        "loss_f32": np.array(f32["loss"].item()), "loss_f64": np.array(f64["loss"].item()),
        "embedding_f64_as_f32": f64["taps"]["embedding"].detach().float().numpy(),
    }
    if match == "dot":
        emb = f64["taps"]["embedding"].detach()
        # pre-activation logits recomputed from the reference's own outputs
        fs = f64["out"]["frame_sim"].detach()
        store["logit_f64"] = torch.log(fs / (1 - fs)).numpy()
    if training:
        for tag in ("f32", "f64"):
            for k, v in grad_summary(res[tag]["grads"]).items():
                store[f"grad_{tag}/{k}"] = v
        for k, v in f32["state_after"].items():
            if "running_" in k:
                store[f"after/{k}"] = v.numpy()
    for k, v in st.items():
        if "running_" in k:
            store[f"before/{k}"] = v.numpy()
    if f32["masks"]:
        # masks are regenerated from dropout_seed by shape in call order; store only the seed
        store["dropout_seed"] = np.array(dropout_seed)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **store)
    print(f"  wrote {name}.npz")


def case_postprocessing():
    rng = np.random.RandomState(2024)
    thresholds = np.arange(1 / 100, 1, 1 / 50)
    rows = []
    for T in (250, 125, 37):
        for kind in range(6):
            if kind == 0:
                x = rng.rand(T)
            elif kind == 1:   # smooth blobs
                t = np.arange(T)
                x = 0.5 + 0.5 * np.sin(t / (3.0 + kind) + rng.rand() * 6) * np.cos(t / 17.0)
            elif kind == 2:
                x = np.zeros(T)
            elif kind == 3:
                x = np.ones(T)
            elif kind == 4:   # values exactly on float32(threshold) and its neighbours
                x = rng.rand(T)
                th32 = thresholds.astype(np.float32)
                for j in range(0, T - 2, 3):
                    v = th32[(j // 3) % 50]
                    x[j] = v
                    x[j + 1] = np.nextafter(v, np.float32(1), dtype=np.float32)
                    x[j + 2] = np.nextafter(v, np.float32(0), dtype=np.float32)
            else:             # sparse spikes: exercises gap merging around n_connect
                x = np.zeros(T)
                pos = rng.choice(T, size=max(2, T // 9), replace=False)
                x[pos] = 0.2 + 0.8 * rng.rand(len(pos))
            rows.append(x.astype(np.float32))
    flat_in = []
    seg_rows = []   # (row, th_idx, window, n_connect, onset, offset)
    for ri, x in enumerate(rows):
        xt = torch.from_numpy(x)
        for window in (1, 3, 4):
            for n_connect in (7, 13):
                for ti, th in enumerate(thresholds):
                    filt = EU.median_filter(xt.unsqueeze(0), window_size=window, threshold=th)[0]
                    reg = EU.find_contiguous_regions(EU.connect_clusters(filt, n_connect))
                    mine = O.segments(x, th, window, n_connect)
                    assert np.array_equal(np.asarray(reg, dtype=np.int64).reshape(-1, 2), mine), \
                        (ri, window, n_connect, ti, reg, mine)
                    for on, off in reg:
                        seg_rows.append((ri, ti, window, n_connect, int(on), int(off)))
    lens = np.array([len(r) for r in rows])
    np.savez_compressed(os.path.join(HERE, "postproc.npz"), rows=np.concatenate(rows), row_len=lens,
                        thresholds=thresholds, segments=np.asarray(seg_rows, dtype=np.int64))
    print(f"  wrote postproc.npz: {len(rows)} rows, {len(seg_rows)} segments; oracle == reference")


def case_align():
    g = torch.Generator().manual_seed(5)
    audio = torch.randn(2, 50, 64, generator=g)
    text = torch.randn(2, 3, 64, generator=g)
    out = {}
    for l2norm in (False, True):
        for scaled in (False, True):
            ref = AL.DotProduct(l2norm=l2norm, scaled=scaled)(audio, text)
            mine = O.align_dot_product(audio, text, l2norm, scaled)
            assert (ref - mine).abs().max().item() < 1e-6
            out[f"l2{int(l2norm)}_sc{int(scaled)}"] = ref.numpy()
    np.savez_compressed(os.path.join(HERE, "align.npz"), audio=audio.numpy(), text=text.numpy(), **out)
    print("  wrote align.npz")


def case_heads():
    """match.ExpNegL2 / match.DotProduct / FrameBceLoss on small explicit inputs."""
    g = torch.Generator().manual_seed(6)
    audio = torch.randn(3, 11, 32, generator=g) * 2
    text = torch.randn(3, 32, generator=g) * 2
    label = (torch.rand(3, 11, generator=g) < 0.5).float()
    length = torch.tensor([11, 7, 1])
    out = {"audio": audio.numpy(), "text": text.numpy(), "label": label.numpy(), "length": length.numpy()}
    fd = {"audio_emb": audio, "text_emb": {"seq_emb": text}, "audio_len": length}
    for nm, mod, mine in (
        ("expnegl2", MA.ExpNegL2(), O.match_exp_neg_l2(audio, text)),
        ("expnegl2_raw", MA.ExpNegL2(l2norm=False), O.match_exp_neg_l2(audio, text, l2norm=False)),
        ("dot", MA.DotProduct(), O.match_dot_product(audio, text)),
        ("dot_l2", MA.DotProduct(l2norm=True, scale=False), O.match_dot_product(audio, text, True, False)),
    ):
        ref = mod(fd)
        assert (ref - mine).abs().max().item() < 1e-6, nm
        out["sim_" + nm] = ref.numpy()
        l_ref = LOSS.FrameBceLoss()({"frame_sim": ref.clone(), "label": label, "length": length})
        l_mine = O.frame_bce_loss(mine, label, length)
        assert abs(l_ref.item() - l_mine.item()) < 1e-6
        out["loss_" + nm] = np.array(l_ref.item())
    np.savez_compressed(os.path.join(HERE, "heads.npz"), **out)
    print("  wrote heads.npz")


def case_frontend():
    """Known answers of the restated frontend (PARITY UNPINNED vs torchaudio; regression vectors)."""
    n = torch.arange(32000, dtype=torch.float32)
    x = (0.5 * torch.sin(2 * math.pi * 1000.0 * n / 32000.0)).unsqueeze(0)
    g = torch.Generator().manual_seed(11)
    noise = 0.1 * torch.randn(2, 32000, generator=g)
    out = {}
    for kind in ("cnn8rnn", "crnn"):
        out[f"sine_{kind}"] = O.logmel(x, kind).numpy()
        out[f"noise_{kind}"] = O.logmel(noise, kind).numpy()
        out[f"noise_power_{kind}"] = O.mel_spectrogram(noise, kind).numpy()
        w, fb = O.frontend_tables(kind)
        out[f"fb_sum_{kind}"] = np.array([fb.sum().item(), fb.max().item()])
        # independent cross-check of the filterbank (third-party, build container only)
        from transformers.audio_utils import mel_filter_bank
        p = O.FRONTEND[kind]
        alt = mel_filter_bank(p["n_fft"] // 2 + 1, p["n_mels"], p["f_min"], p["f_max"], p["sample_rate"],
                              p["norm"], p["mel_scale"])
        d = np.abs(alt - fb.numpy()).max()
        print(f"  fb[{kind}] vs transformers.audio_utils.mel_filter_bank: max abs diff {d:.2e}")
        assert d < 1e-5
    np.savez_compressed(os.path.join(HERE, "frontend.npz"), **out)
    print("  wrote frontend.npz")
    case_frontend_witness(noise)


def case_frontend_witness(noise):
    """INDEPENDENT witness for rows F1/F2: the same two seeded noise clips through
    transformers.audio_utils (window_function -> spectrogram(center, reflect, power 2, onesided) -> mel_filter_bank), in
    fp64, with ITS OWN window and filterbank.  Nothing of the oracle is used to produce these arrays; the oracle (and,
    on the GPU, logmel.hip) must agree with them in the power domain.  torchaudio itself stays absent: this pins the
    oracle's reading of torch.stft / MelScale / AmplitudeToDB against a second implementation of the published algorithm
    for both parameter sets (Cnn8Rnn n_fft = win = 1024; CrnnEncoder win 1280 zero-padded centred to n_fft 2048)."""
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function
    out = {}
    for kind in ("cnn8rnn", "crnn"):
        p = O.FRONTEND[kind]
        fb = mel_filter_bank(p["n_fft"] // 2 + 1, p["n_mels"], p["f_min"], p["f_max"], p["sample_rate"], p["norm"],
                             p["mel_scale"])
        win = window_function(p["win_length"], "hann", periodic=True, frame_length=p["n_fft"], center=True)
        P = np.stack([spectrogram(noise[b].double().numpy(), win, frame_length=p["n_fft"], hop_length=p["hop_length"],
                                  fft_length=p["n_fft"], power=2.0, center=True, pad_mode="reflect", onesided=True,
                                  mel_filters=fb, mel_floor=0.0, dtype=np.float64) for b in range(noise.shape[0])])
        out[f"power_{kind}"] = P
        out[f"db_{kind}"] = 10.0 * np.log10(np.maximum(P, 1e-10))
        mine = O.mel_spectrogram(noise.double(), kind).numpy()
        d = np.abs(P - mine).max() / np.abs(P).max()
        ddb = np.abs(out[f"db_{kind}"] - O.logmel(noise.double(), kind).numpy()).max()
        print(f"  witness[{kind}] vs oracle(fp64): power rel {d:.2e}, dB {ddb:.2e}")
        assert d < 1e-5 and ddb < 2e-4
    np.savez_compressed(os.path.join(HERE, "frontend_witness.npz"), **out)
    print("  wrote frontend_witness.npz")


if __name__ == "__main__":
    import math
    case_frontend()
    case_postprocessing()
    case_align()
    case_heads()
    case_whole_path("cnn8rnn_dot_eval", "cnn8rnn", "dot", 512, training=False)
    case_whole_path("cnn8rnn_dot_train", "cnn8rnn", "dot", 512, training=True)
    case_whole_path("cnn8rnn_proj_expnegl2_train_dropout", "cnn8rnn", "expnegl2", 256, training=True,
                    dropout_seed=4321)
    case_whole_path("crnn_expnegl2_train", "crnn", "expnegl2", 256, training=True)
    case_whole_path("crnn_expnegl2_eval", "crnn", "expnegl2", 256, training=False)
    sizes = {f: os.path.getsize(os.path.join(HERE, f)) for f in sorted(os.listdir(HERE)) if f.endswith(".npz")}
    print(json.dumps(sizes, indent=1), "total", sum(sizes.values()))
