#!/usr/bin/env python3
"""The PCIe-inclusive rate of the contract's step (never `value`): the same 20 steps as `python bench.py`, but every step's batch
starts in pinned HOST memory (the padded fp32 waveforms the reference's collate builds: 64 x 320000 x 4 B = 82 MB, plus lengths,
token ids, labels) and is copied to the device inside the timed region -- (a) on the compute stream in front of the step, (b) on a
copy stream one step ahead (double-buffered), and (c) as float16 waveform packs (datasets/: what the reference's HDF5 files hold)
widened on the device by tag_waveform_f16_to_f32_padded.

    python tools/pcie_inclusive.py [steps=20]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from texttoaudiogrounding_amd.runner import StrongRunner  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench.build_workload("biencoder", dev)
runner = StrongRunner(model, lr=1e-3, max_grad_norm=1.0, device=str(dev))
dbatch = bench.synthetic_batch(64, 320000, 1234, dev)
hbatch = {k: (v.cpu().pin_memory() if torch.is_tensor(v) else v) for k, v in dbatch.items()}
nbytes = sum(v.numel() * v.element_size() for v in hbatch.values() if torch.is_tensor(v))


def to_dev(hb, stream=None):
    with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
        return {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in hb.items()}


def timed(fn, k=K):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


res = {}
res["resident (bench.py's value)"] = timed(lambda: runner.train_step(dict(dbatch)))
res["H2D on the compute stream, then the step"] = timed(lambda: runner.train_step(to_dev(hbatch)))
copy_stream = torch.cuda.Stream(dev)
state = {"next": None}


def prefetched():
    main = torch.cuda.current_stream()
    if state["next"] is None:
        state["next"] = to_dev(hbatch, copy_stream)
    cur = state["next"]
    main.wait_stream(copy_stream)
    state["next"] = to_dev(hbatch, copy_stream)          # the NEXT step's batch crosses PCIe while this step computes
    runner.train_step(cur)
    copy_stream.wait_stream(main)                        # (the buffers of `cur` are released on the main stream after their use)


res["H2D on a copy stream one step ahead"] = timed(prefetched)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    to_dev(hbatch)
e1.record()
torch.cuda.synchronize()
h2d_ms = e0.elapsed_time(e1) / 10
print(f"batch in host memory: {nbytes / 1e6:.1f} MB; H2D alone {h2d_ms:.3f} ms = {nbytes / h2d_ms / 1e6:.1f} GB/s")
for k_, v in res.items():
    print(f"{k_:45s} {v:8.3f} ms/step  {64 / v * 1e3:8.1f} clips/s")
