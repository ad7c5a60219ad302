#!/usr/bin/env python3
"""Fixture for CrnnEncoder parity AT THE SIZE `bench.py --crnn` times (the strong eg_config as written, cdur_w2vmean.yaml:
CrnnEncoder(256) + EmbeddingAgg(256, mean) + ExpNegL2; batch 64, 10 s @ 32 kHz clips, ragged lengths, train-mode BatchNorm,
dropout 0.3 ON): one training step of the CPU oracle in fp64 (the truth) and in fp32 (the noise floor of ANY fp32
implementation), with the HIP path's counter-based dropout mask replayed from a fixed seed.

The oracle's crnn_forward is pinned against the imported reference by make_golden.py (crnn_expnegl2_{train,eval}.npz,
models/audio_encoder.py:25-86); this script only scales it to the benched size.

Writes tests/golden/crnn_b64_train_step.npz (< 1 MB): loss, frame_sim (64,125) fp64, per-tensor gradient norm / max / 1024
sampled entries (fp64) and per tensor the fp32 oracle's own distance from fp64.

    python tests/golden/make_golden_crnn_b64.py          (build container only)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import tag_oracle as O  # noqa: E402

B, S, HOP = 64, 320000, 640
STATE_SEED, TABLE_SEED, BATCH_SEED = 7, 8, 199
DROPOUT_SEED = 6000011                      # what the test feeds ops.new_seed()
P_DROP = 0.3
N_SAMPLE = 1024


def sample_index(numel, name):
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    return torch.randint(0, numel, (min(N_SAMPLE, numel),), generator=g)


def checksum(t):
    t = t.detach().double().flatten()
    return [float(t.sum()), float(t.abs().max()), float(t[:: max(1, t.numel() // 7)][:7].sum())]


def crnn_b64_state():
    """Seeded CrnnEncoder + embedding table with non-trivial BatchNorm affines (shared with the GPU test)."""
    st = O.init_crnn_state(seed=STATE_SEED)
    g = torch.Generator().manual_seed(TABLE_SEED)
    st["text_encoder.embedding.core.weight"] = (torch.rand(5221, 256, generator=g) * 2 - 1) * 0.9
    for k in list(st):
        if k.endswith(".0.weight"):
            st[k] = 0.5 + torch.rand(st[k].shape, generator=g)
        if k.endswith(".0.bias"):
            st[k] = 0.2 * torch.randn(st[k].shape, generator=g)
    return st


def crnn_dropout_mask(seed, batch, n_frames, dtype=torch.float32):
    """Keep mask of CrnnEncoder's Dropout(0.3) for the HIP path's seed: the kernel indexes the channels-last pooled
    output (B, T', 1, 128) flat; the oracle applies it to (B, 128, T', 1)."""
    tp = (n_frames // 2) // 2
    m = O.dropout_keep_mask4(seed, batch * tp * 128, P_DROP).reshape(batch, tp, 1, 128)
    return {"drop": torch.from_numpy(m).permute(0, 3, 1, 2).to(dtype)}


def run(dtype, st0, batch, masks):
    st = O.state_to(st0, dtype, requires_grad=True)
    b = dict(batch)
    b["waveform"], b["label"] = batch["waveform"].to(dtype), batch["label"].to(dtype)
    t0 = time.time()
    loss, out = O.train_step_loss(st, b, "expnegl2", "crnn", True, None, {k: v.to(dtype) for k, v in masks.items()})
    loss.backward()
    print(f"  {dtype}: loss {loss.item():.9f}  ({time.time() - t0:.0f} s)", flush=True)
    grads = {k: v.grad.detach().double() for k, v in st.items() if v.is_floating_point() and v.grad is not None}
    return float(loss.item()), out["frame_sim"].detach().double().numpy(), grads


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    st0 = crnn_b64_state()
    batch = O.synthetic_batch(B, S, seed=BATCH_SEED, ragged=True, hop=HOP)
    masks = crnn_dropout_mask(DROPOUT_SEED, B, S // HOP + 1)
    out = {"input_checksum": np.array(checksum(batch["waveform"]) + checksum(batch["text"].float())
                                      + checksum(st0["audio_encoder.gru.weight_ih_l0"])),
           "dropout_seed": np.array(DROPOUT_SEED, dtype=np.int64),
           "mask_keep_count": np.array(int(torch.count_nonzero(masks["drop"]).item()), dtype=np.int64)}
    l64, fs64, g64 = run(torch.float64, st0, batch, masks)
    l32, fs32, g32 = run(torch.float32, st0, batch, masks)
    out["loss_f64"], out["loss_f32"] = np.array(l64), np.array(l32)
    out["frame_sim_f64"] = fs64
    out["frame_sim_floor"] = np.array(np.abs(fs32 - fs64).max())
    for name, g in g64.items():
        flat, f32 = g.flatten(), g32[name].flatten()
        idx = sample_index(flat.numel(), name)
        scale = flat.abs().max().item() + 1e-300
        out[f"grad/{name}"] = np.concatenate([[flat.norm().item(), flat.abs().max().item()], flat[idx].numpy()])
        out[f"floor/{name}"] = np.array([(f32 - flat).abs().max().item() / scale,
                                         (f32[idx] - flat[idx]).abs().max().item() / scale,
                                         abs(f32.norm().item() - flat.norm().item()) / (flat.norm().item() + 1e-300)])
        print(f"  {name:45s} |g| {flat.norm().item():.3e}  fp32 floor {out[f'floor/{name}'][0]:.2e} "
              f"(sample {out[f'floor/{name}'][1]:.2e})")
    path = os.path.join(HERE, "crnn_b64_train_step.npz")
    np.savez_compressed(path, **out)
    print("wrote crnn_b64_train_step.npz", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
