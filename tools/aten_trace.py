#!/usr/bin/env python3
"""Which ATen / runtime kernels does one training step still launch, and from where?  (torch.profiler, with stacks)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from texttoaudiogrounding_amd import ops
from texttoaudiogrounding_amd.models import audio_encoder, audio_text_model, match, text_encoder
from texttoaudiogrounding_amd.runner import StrongRunner
dev = torch.device("cuda:0")
if len(sys.argv) > 1 and sys.argv[1] == "bf16":
    ops.CONV_MATH = "bf16"; ops.ACT_DTYPE = "bf16"
torch.manual_seed(0)
model = audio_text_model.BiEncoder(audio_encoder.Cnn8Rnn(32000), text_encoder.EmbeddingAgg(5221, 512), match.DotProduct(), 512)
runner = StrongRunner(model, lr=1e-3, max_grad_norm=1.0, device=str(dev))
batch = bench.synthetic_batch(64, 320000, 1234, dev)
for _ in range(3):
    runner.train_step(dict(batch))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    runner.train_step(dict(batch))
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::")]
from collections import Counter
cnt = Counter()
for e in ev:
    if e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue                                  # top-level ATen calls only
    st = [s for s in (e.stack or []) if "texttoaudiogrounding_amd" in s or "bench.py" in s]
    cnt[(e.name, st[0].split("/")[-1] if st else "?")] += 1
for (name, where), n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d}  {name:28s} {where}")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
