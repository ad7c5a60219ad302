#!/usr/bin/env python3
"""Host-side cost of one training step: time to ENQUEUE a step (Python + ctypes launches, no sync) vs its GPU time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_batch
from texttoaudiogrounding_amd import ops
from texttoaudiogrounding_amd.models import audio_encoder, audio_text_model, match, text_encoder
from texttoaudiogrounding_amd.runner import StrongRunner
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = audio_text_model.BiEncoder(audio_encoder.Cnn8Rnn(32000), text_encoder.EmbeddingAgg(5221, 512), match.DotProduct(), 512)
runner = StrongRunner(model, device=str(dev))
batch = synthetic_batch(int(os.environ.get("B", "64")), 320000, 1234, dev)
for _ in range(2):
    runner.train_step(dict(batch))
torch.cuda.synchronize()
K = 5
t0 = time.perf_counter()
for _ in range(K):
    runner.train_step(dict(batch))
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / K:.2f} ms/step, total {1e3 * (t2 - t0) / K:.2f} ms/step, cpu count {os.cpu_count()}, "
      f"affinity {len(os.sched_getaffinity(0))}")
