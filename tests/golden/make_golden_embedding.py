#!/usr/bin/env python3
"""tests/golden/embedding_layer.npz: the REFERENCE's ``models.text_encoder.EmbeddingLayer`` run on its own (imported from
/root/reference; models/text_encoder.py:14-43): ``core(tokens.long())`` for token tensors of rank 1, 2 and 3 (padding ids and
repeated ids included), outputs and the table gradient for seeded upstream gradients (fp64 twin for the gradient, whose
summation order is the implementation's own).  The module is constructed under a seed, so the fixture stores only outputs /
gradients plus a checksum of the table.  Build container only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

V, D = 53, 96
CASES = {"rank1": (7,), "rank2": (4, 6), "rank3": (2, 3, 5)}


def make_layer(cls):
    torch.manual_seed(4242)
    return cls(V, D)


def tokens(name):
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    t = torch.randint(0, V, CASES[name], generator=g)
    flat = t.view(-1)
    flat[-2:] = 0                                # padding ids
    flat[1] = flat[0]                            # a repeated id: its table row receives two gradient rows
    dout = torch.randn(*CASES[name], D, generator=g)
    return t, dout


def checksum(t):
    t = t.double().flatten()
    return np.array([t.sum().item(), (t * torch.arange(1, t.numel() + 1, dtype=torch.float64)).sum().item() / t.numel()])


if __name__ == "__main__":
    ref_import.install()
    from models.text_encoder import EmbeddingLayer  # noqa: E402  (the reference)

    out = {}
    for name in CASES:
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            m = make_layer(EmbeddingLayer).to(dt)
            out["table_checksum"] = checksum(m.core.weight.detach())
            t, dout = tokens(name)
            e = m({"text": t.int()})             # the reference casts with .long()
            e.backward(dout.to(dt))
            assert tuple(e.shape) == CASES[name] + (D,)
            if tag == "f32":
                out[f"{name}/out"] = e.detach().numpy()
            else:
                out[f"{name}/dtable_f64"] = m.core.weight.grad.numpy()
    path = os.path.join(HERE, "embedding_layer.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))
