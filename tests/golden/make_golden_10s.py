#!/usr/bin/env python3
"""Fixture of the IMPORTED REFERENCE at the benched clip length: B = 2 clips of 10 s @ 32 kHz (S = 320 000, second clip ragged),
eval mode -- the frame chain F = 1001 -> 500 -> 250 of models/audio_encoder.py:202-227 pinned to the reference itself, not to
the oracle (every other whole-path fixture generated from the reference is 1.5 s long: make_golden.py, S = 48 000).

    python tests/golden/make_golden_10s.py        (build container only: needs /root/reference)

Stored: frame_sim of the reference in fp32 and of its fp64 twin (model.double()), `length`, the audio embedding, and the
segments utils/eval_util.py's own functions (median_filter -> connect_clusters -> find_contiguous_regions, run_strong.py:203-252)
produce from the reference's fp32 frame_sim at the 50 evaluation thresholds with window 1 and n_connect = ceil(0.5 / 0.04) = 13,
plus `margin`: the distance of the nearest fp64 score to each threshold (a threshold whose margin is below the 1e-4 parity
tolerance is undecidable for ANY fp32 implementation and is skipped by the test; at these seeds none is).
The oracle is checked against the reference on the way (abort on disagreement).
"""
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
from make_golden import build_reference, calibrate_running_stats, checksum, mods  # noqa: E402  (installs the reference import)

EU = mods["utils.eval_util"]
S, B, HOP = 320000, 2, 320


def make_batch():
    b = O.synthetic_batch(B, S, seed=4321, ragged=False, hop=HOP)
    lens = np.array([S, S - 37 * HOP * 4 - 77])            # 250 and 213 valid output frames
    b["waveform"][1, lens[1]:] = 0.0
    b["waveform_len"] = lens
    return b


def main():
    torch.set_num_threads(8)
    batch = make_batch()
    st = O.init_state(seed=11, text_dim=512, shared_dim=512, logit_gain=120.0)
    st = calibrate_running_stats(st, batch, "cnn8rnn")
    res = {}
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        model = build_reference(O.state_to(st, dtype), "cnn8rnn", "dot").to(dtype).eval()
        taps = {}
        h = model.audio_encoder.register_forward_hook(lambda m, i, o: taps.__setitem__("embedding", o["embedding"]))
        with torch.no_grad():
            out = model({"waveform": batch["waveform"].to(dtype), "waveform_len": batch["waveform_len"],
                         "text": batch["text"], "text_len": batch["text_len"], "specaug": False})
        h.remove()
        bo = dict(batch)
        bo["waveform"] = batch["waveform"].to(dtype)
        oout = O.biencoder_forward(O.state_to(st, dtype), bo, "dot", "cnn8rnn", training=False)
        d = (oout["frame_sim"] - out["frame_sim"]).abs().max().item()
        print(f"[10s/{tag}] frame_sim {tuple(out['frame_sim'].shape)} oracle-vs-reference {d:.3e}; length {out['length'].tolist()}")
        assert d < (2e-5 if dtype == torch.float32 else 1e-10), "oracle != reference"
        assert torch.equal(oout["length"], out["length"])
        res[tag] = (out, taps["embedding"])
    fs32, fs64 = res["f32"][0]["frame_sim"], res["f64"][0]["frame_sim"]
    assert fs32.shape == (B, 250) and res["f32"][0]["length"].tolist() == [250, 213]
    thresholds = np.arange(1 / 100, 1, 1 / 50)
    n_connect = int(np.ceil(0.5 / 0.04))
    seg_rows, margin = [], np.zeros((B, len(thresholds)))
    for b in range(B):
        for ti, th in enumerate(thresholds):
            filt = EU.median_filter(fs32[b].unsqueeze(0), window_size=1, threshold=th)[0]
            reg = EU.find_contiguous_regions(EU.connect_clusters(filt, n_connect))
            mine = O.segments(fs32[b].numpy(), th, 1, n_connect)
            assert np.array_equal(np.asarray(reg, dtype=np.int64).reshape(-1, 2), mine)
            margin[b, ti] = np.abs(fs64[b].numpy() - th).min()
            for on, off in reg:
                seg_rows.append((b, ti, int(on), int(off)))
    print(f"  segments: {len(seg_rows)} rows; smallest threshold margin {margin.min():.3e}")
    np.savez_compressed(
        os.path.join(HERE, "cnn8rnn_dot_eval_10s.npz"),
        input_checksum=np.array(checksum(batch["waveform"]) + checksum(batch["text"].float())
                                + checksum(st["audio_encoder.fc1.weight"])),
        waveform_len=np.asarray(batch["waveform_len"]),
        frame_sim_f32=fs32.numpy(), frame_sim_f64=fs64.numpy(), length=res["f32"][0]["length"].numpy(),
        embedding_f64_as_f32_every5=res["f64"][1][:, ::5].float().numpy(),     # frames 0, 5, ...: 200 KB instead of 1 MB
        thresholds=thresholds, n_connect=np.array(n_connect), segments=np.asarray(seg_rows, dtype=np.int64),
        margin=margin,
        **{f"before/{k}": v.numpy() for k, v in st.items() if "running_" in k})
    print("  wrote cnn8rnn_dot_eval_10s.npz", os.path.getsize(os.path.join(HERE, "cnn8rnn_dot_eval_10s.npz")), "bytes")


def main_crnn():
    """The same at 10 s for the variant the strong eg_config literally instantiates (row A1': CrnnEncoder(256) + EmbeddingAgg(256) +
    ExpNegL2, hop 640: F = 501 -> 250 -> 125; models/audio_encoder.py:16-86), from the imported reference."""
    hop = 640
    b = O.synthetic_batch(B, S, seed=4322, ragged=False, hop=hop)
    lens = np.array([S, S - 19 * hop * 4 - 55])
    b["waveform"][1, lens[1]:] = 0.0
    b["waveform_len"] = lens
    st = O.init_crnn_state(seed=13)
    g = torch.Generator().manual_seed(14)
    st["text_encoder.embedding.core.weight"] = (torch.rand(5221, 256, generator=g) * 2 - 1) * 0.9
    for k in list(st):
        if k.endswith(".0.weight"):
            st[k] = 0.5 + torch.rand(st[k].shape, generator=g)
        if k.endswith(".0.bias"):
            st[k] = 0.2 * torch.randn(st[k].shape, generator=g)
    st = calibrate_running_stats(st, b, "crnn")
    res = {}
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        model = build_reference(O.state_to(st, dtype), "crnn", "expnegl2").to(dtype).eval()
        with torch.no_grad():
            out = model({"waveform": b["waveform"].to(dtype), "waveform_len": b["waveform_len"], "text": b["text"],
                         "text_len": b["text_len"], "specaug": False})
        bo = dict(b)
        bo["waveform"] = b["waveform"].to(dtype)
        oout = O.biencoder_forward(O.state_to(st, dtype), bo, "expnegl2", "crnn", training=False)
        d = (oout["frame_sim"] - out["frame_sim"]).abs().max().item()
        print(f"[crnn 10s/{tag}] frame_sim {tuple(out['frame_sim'].shape)} oracle-vs-reference {d:.3e}; length {out['length'].tolist()}")
        assert d < (2e-5 if dtype == torch.float32 else 1e-10) and torch.equal(oout["length"], out["length"])
        res[tag] = out
    assert res["f32"]["frame_sim"].shape == (B, 125)
    np.savez_compressed(
        os.path.join(HERE, "crnn_expnegl2_eval_10s.npz"),
        input_checksum=np.array(checksum(b["waveform"]) + checksum(b["text"].float())
                                + checksum(st["audio_encoder.gru.weight_hh_l0"])),
        waveform_len=np.asarray(b["waveform_len"]), frame_sim_f32=res["f32"]["frame_sim"].numpy(),
        frame_sim_f64=res["f64"]["frame_sim"].numpy(), length=res["f32"]["length"].numpy(),
        **{f"before/{k}": v.numpy() for k, v in st.items() if "running_" in k})
    print("  wrote crnn_expnegl2_eval_10s.npz", os.path.getsize(os.path.join(HERE, "crnn_expnegl2_eval_10s.npz")), "bytes")


if __name__ == "__main__":
    main()
    main_crnn()
