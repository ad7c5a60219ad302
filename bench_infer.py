#!/usr/bin/env python3
"""Secondary benchmark (NOT the driver's contract -- that is bench.py): BASELINE.json configs[4], inference-only throughput of
cnn8rnn + LAION-CLAP text tower (models/hf_modeling_grounding.py in the reference) on 30 s @ 32 kHz clips, batch 256, 1 GPU.
Random-init weights of the published architecture (RoBERTa-base text tower, 512-d projections), synthetic clips and tokens.

    python bench_infer.py [--batch 256] [--steps 3] [--warmup 1] [--tokens 8] [--conv-math fp32|x3]
prints one JSON line: clips/s for the whole forward (log-mel -> Cnn8Rnn -> audio_proj; tokens -> CLAP -> text_proj; DotProduct).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_CLIP_30S = 33.90e9 * 3.0          # forward of a 10 s clip x 3 (T' = 750)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tokens", type=int, default=8)
    ap.add_argument("--conv-math", default="fp32", choices=["fp32", "x3", "x9", "bf16"])
    args = ap.parse_args()
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd.models.hf_modeling_grounding import Cnn8RnnLaionClapGroundingModel
    ops.CONV_MATH = args.conv_math
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = Cnn8RnnLaionClapGroundingModel().to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(1234)
    audio = 0.1 * torch.randn(args.batch, 960000, device=dev, generator=g)
    audio_len = torch.full((args.batch,), 960000)
    ids = torch.randint(3, 50265, (args.batch, args.tokens), device=dev, generator=g)
    ids[:, 0], ids[:, -1] = 0, 2
    text = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
    for _ in range(args.warmup):
        model(audio, audio_len, text)
    torch.cuda.synchronize()
    ops.PROFILE = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fs = model(audio, audio_len, text)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    fam = {}
    for key, evs in prof.items():
        d = fam.setdefault(key[0], [0.0, 0.0])
        d[0] += sum(e0.elapsed_time(e1) for e0, e1, _ in evs)
        d[1] += sum(f for _, _, f in evs)
    # text tower alone
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        model.model.text_encoder(text)
    torch.cuda.synchronize()
    dtt = time.perf_counter() - t1
    value = args.batch * args.steps / dt
    out = {"metric": "clips/sec (30 s@32 kHz, 1 phrase) inference", "value": round(value, 2), "unit": "clips/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
           "higher_is_better": True, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "configs[4]: Cnn8Rnn + LAION-CLAP text tower (RoBERTa-base shape, random init) + "
                                  "audio/text proj + DotProduct, forward only", "batch": args.batch, "clip": "30 s @ 32 kHz",
                      "tokens_per_phrase": args.tokens, "conv_math": args.conv_math},
           "frame_sim_shape": list(fs.shape),
           "whole_forward_mfma_frac": round(value * FLOP_PER_CLIP_30S / 1e12 / 157.3, 4),
           "text_tower_ms_per_batch": round(dtt / args.steps * 1e3, 2),
           "conv_families": {k: {"TFLOP/s": round(v[1] / (v[0] * 1e-3) / 1e12, 1), "ms_per_step": round(v[0] / args.steps, 2)}
                             for k, v in fam.items()}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
