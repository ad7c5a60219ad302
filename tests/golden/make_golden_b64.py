#!/usr/bin/env python3
"""Fixture for parity AT THE BENCHED SIZE (BASELINE configs[1]: batch 64, 10 s @ 32 kHz clips, train-mode BatchNorm,
dropout ON): one training step of the CPU oracle in fp64 (the truth) and in fp32 (the noise floor of ANY fp32
implementation: ReLU / max-pool decisions that flip under rounding), with the HIP path's own counter-based dropout masks
replayed from fixed seeds (oracle.dropout_keep_mask restates the generator bit for bit).

The oracle itself is pinned against the imported reference by make_golden.py (bit-equal in train mode); this script only
scales it to the size the CPU of the GPU box cannot afford inside a test (fp64 twin: ~6 min, ~27 GB here).

Writes tests/golden/b64_train_step.npz (< 1 MB): loss, frame_sim (64,250) fp64, per-tensor gradient norm / max /
1024 sampled entries (fp64), and per tensor the fp32 oracle's own distance from fp64 (max-normalised), over the whole
tensor and over the sample.  Inputs are regenerated from seeds on the GPU box; their checksums are stored.

    python tests/golden/make_golden_b64.py          (build container only; needs ~30 GB of RAM)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import tag_oracle as O  # noqa: E402

B, S = 64, 320000
STATE_SEED, BATCH_SEED, LOGIT_GAIN = 5, 99, 120.0
DROPOUT_SEEDS = [1000003, 2000003, 3000017, 4000037, 5000011]          # what the test feeds ops.new_seed()
N_SAMPLE = 1024


def sample_index(numel, name):
    g = torch.Generator().manual_seed(sum(map(ord, name)))          # deterministic per tensor name (no hash())
    return torch.randint(0, numel, (min(N_SAMPLE, numel),), generator=g)


def checksum(t):
    t = t.detach().double().flatten()
    return [float(t.sum()), float(t.abs().max()), float(t[:: max(1, t.numel() // 7)][:7].sum())]


def run(dtype, st0, batch, masks):
    st = O.state_to(st0, dtype, requires_grad=True)
    b = dict(batch)
    b["waveform"], b["label"] = batch["waveform"].to(dtype), batch["label"].to(dtype)
    t0 = time.time()
    loss, out = O.train_step_loss(st, b, "dot", "cnn8rnn", True, None, {k: v.to(dtype) for k, v in masks.items()})
    loss.backward()
    print(f"  {dtype}: loss {loss.item():.9f}  ({time.time() - t0:.0f} s)", flush=True)
    grads = {k: v.grad.detach().double() for k, v in st.items() if v.is_floating_point() and v.grad is not None}
    return float(loss.item()), out["frame_sim"].detach().double().numpy(), grads


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    st0 = O.init_state(seed=STATE_SEED, logit_gain=LOGIT_GAIN)
    batch = O.synthetic_batch(B, S, seed=BATCH_SEED, ragged=True)
    masks = O.cnn8rnn_dropout_masks(DROPOUT_SEEDS, B, S // 320 + 1)
    out = {"input_checksum": np.array(checksum(batch["waveform"]) + checksum(batch["text"].float())
                                      + checksum(st0["audio_encoder.fc1.weight"])),
           "dropout_seeds": np.array(DROPOUT_SEEDS, dtype=np.int64),
           # exact counts (a float32 sum is not)
           "mask_keep_counts": np.array([int(torch.count_nonzero(m).item()) for m in masks.values()], dtype=np.int64)}
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
        print(f"  {name:55s} |g| {flat.norm().item():.3e}  fp32 floor {out[f'floor/{name}'][0]:.2e} (sample {out[f'floor/{name}'][1]:.2e})")
    np.savez_compressed(os.path.join(HERE, "b64_train_step.npz"), **out)
    print("wrote b64_train_step.npz", os.path.getsize(os.path.join(HERE, "b64_train_step.npz")), "bytes")


if __name__ == "__main__":
    main()
