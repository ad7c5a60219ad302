#!/usr/bin/env python3
"""Phase clocks of the 4-row persistent GRU forward (build gru.o with -DTAG_GRU_PROF first: the kernel then accumulates
s_memtime deltas of its phases over all steps in one workgroup and leaves them behind the scratch's error word)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texttoaudiogrounding_amd import ops
dev = torch.device("cuda:0")
B, T, I, H = 64, 250, 512, 256
g = torch.Generator().manual_seed(0)
k = 1 / math.sqrt(H)
rnn = []
for _ in range(2):
    rnn += [((torch.rand(3 * H, I, generator=g) * 2 - 1) * k).to(dev), ((torch.rand(3 * H, H, generator=g) * 2 - 1) * k).to(dev),
            ((torch.rand(3 * H, generator=g) * 2 - 1) * k).to(dev), ((torch.rand(3 * H, generator=g) * 2 - 1) * k).to(dev)]
x = torch.randn(B * T, I, device=dev)
for _ in range(3):
    y, sv = ops.gru_bidir_forward(x, rnn, B, T, True)
torch.cuda.synchronize()
for key, ws in ops._gru_scratch.items():
    if key[-1] != "fwd":
        continue
    words = ws.view(torch.int64)[(ws._tag_err_index * 4 + 64) // 8: (ws._tag_err_index * 4 + 64) // 8 + 10].cpu().tolist()
    names = ["sweep", "barrier", "deal+MFMA", "lane reduction", "math+stores"]
    for w, off in (("wave 0", 0), ("wave 3", 5)):
        tot = sum(words[off:off + 5])
        print(w, {n: f"{v / T:.0f} clk/step" for n, v in zip(names, words[off:off + 5])}, f"total {tot / T:.0f} clk/step (s_memtime = 100 MHz constant clock: x10 ns)")
