#!/usr/bin/env python3
"""Where does the rows kernel move the bf16-mode gradients of the B = 64 fixture step?  (GPU box)  Runs the fixture step with the
row-streaming conv kernel on and off and prints, per conv-block tensor, the norms against the fp64 fixture and the relative
distance between the two runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import tag_oracle as O
from tests.test_gpu_path import build_hip_model
from texttoaudiogrounding_amd import functions, ops
from texttoaudiogrounding_amd.lib import query
from texttoaudiogrounding_amd.runner import StrongRunner
dev = torch.device("cuda:0")
gold = np.load("tests/golden/b64_train_step.npz")
ops.CONV_MATH, ops.ACT_DTYPE = "bf16", "bf16"
res = {}
for on in (0, 1):
    query("tag_conv_rows_enable", on)
    st = O.init_state(seed=5, logit_gain=120.0)
    batch = O.synthetic_batch(64, 320000, seed=99, ragged=True)
    seeds = iter(int(v) for v in gold["dropout_seeds"])
    functions.new_seed = lambda: next(seeds)
    model = build_hip_model(st, "dot", dev).train()
    runner = StrongRunner(model, device=str(dev))
    loss = runner.forward_backward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    print("rows", on, "loss", runner.loss_value(loss))
    res[on] = {n: p.grad.detach().double().cpu().clone() for n, p in model.named_parameters()}
for n in res[0]:
    if "conv_block1" in n or "conv_block2.conv" in n or "bn0" in n:
        a, b = res[0][n], res[1][n]
        w = gold[f"grad/{n}"]
        print(f"{n:45s} |tile| {a.norm():.5e} |rows| {b.norm():.5e} |fp64| {w[0]:.5e}  |rows-tile|/|tile| {(a - b).norm() / a.norm():.3e}  "
              f"cos(rows,tile) {torch.dot(a.flatten(), b.flatten()) / (a.norm() * b.norm()):.6f}")
