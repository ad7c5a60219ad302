#!/usr/bin/env python3
"""tests/golden/submodules.npz: the REFERENCE's sub-modules run on their own (imported from /root/reference) --
``models.panns.ConvBlock.forward(input, pool_size, pool_type)`` for the three pool types (models/panns.py:46-62), in train
and eval mode, ``models.cross_encoder.Seq2SeqAttention.forward`` with d_q != d_kv (:11-42) and ``CrossGating.forward``
(:45-57) -- outputs and gradients in fp32 with an fp64 twin.  Parameters and inputs are seeded (module construction under
``torch.manual_seed``), so the fixture stores only outputs / gradients / updated BatchNorm buffers plus an input checksum.
Build container only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

CONV_CASES = [  # name, in_ch, out_ch, (B, T, F), pool_size, pool_type
    ("c64_avg", 64, 128, (2, 12, 8), (2, 2), "avg"),
    ("c64_max", 64, 128, (2, 12, 8), (2, 2), "max"),
    ("c64_avgmax", 64, 128, (2, 12, 8), (2, 2), "avg+max"),
    ("c1_avg_t", 1, 64, (3, 10, 16), (2, 1), "avg"),           # pool over time only, Cin = 1
    ("c64_max_11", 64, 64, (2, 6, 8), (1, 1), "max"),          # (1,1): no pooling
]
ATTN = dict(B=3, Lq=11, Lk=5, d_q=128, d_kv=64, d_attn=64)
GATE = dict(B=3, T=11, D=64)


def convblock_inputs(name, cin, shape, seed_mod=None):
    g = torch.Generator().manual_seed(1000 + sum(map(ord, name)))
    B, T, Fq = shape
    x = torch.randn(B, cin, T, Fq, generator=g)
    return x, g


def make_convblock(cls, name, cin, cout):
    torch.manual_seed(77 + sum(map(ord, name)))
    blk = cls(cin, cout)
    g = torch.Generator().manual_seed(5 + sum(map(ord, name)))
    with torch.no_grad():                      # non-trivial BatchNorm affine + running statistics
        for bn in (blk.bn1, blk.bn2):
            bn.weight.copy_(0.5 + torch.rand(cout, generator=g))
            bn.bias.copy_(0.2 * torch.randn(cout, generator=g))
            bn.running_mean.copy_(0.1 * torch.randn(cout, generator=g))
            bn.running_var.copy_(0.5 + torch.rand(cout, generator=g))
    return blk


def make_attn(cls):
    torch.manual_seed(901)
    m = cls(ATTN["d_q"], ATTN["d_kv"], ATTN["d_attn"])
    with torch.no_grad():
        m.h2attn.weight.mul_(3.0)
    g = torch.Generator().manual_seed(902)
    q = torch.randn(ATTN["B"], ATTN["Lq"], ATTN["d_q"], generator=g)
    kv = torch.randn(ATTN["B"], ATTN["Lk"], ATTN["d_kv"], generator=g)
    dout = torch.randn(ATTN["B"], ATTN["Lq"], ATTN["d_kv"], generator=g)
    return m, q, kv, torch.tensor([11, 7, 9]), torch.tensor([5, 1, 3]), dout


def make_gate(cls):
    torch.manual_seed(911)
    m = cls(GATE["D"])
    g = torch.Generator().manual_seed(912)
    u = torch.randn(GATE["B"], GATE["T"], GATE["D"], generator=g)
    s = torch.randn(GATE["B"], GATE["T"], GATE["D"], generator=g)
    du, ds = torch.randn(u.shape, generator=g), torch.randn(s.shape, generator=g)
    return m, u, s, du, ds


def sample(t, n=512):
    """[norm, absmax, n entries at seeded positions] of a tensor / array (big tensors are stored sampled)."""
    t = torch.as_tensor(t).detach().double().flatten()
    if t.numel() <= 4096:
        return t.numpy()
    idx = torch.randint(0, t.numel(), (n,), generator=torch.Generator().manual_seed(99))
    return np.concatenate([[t.norm().item(), t.abs().max().item()], t[idx].numpy()])


def checksum(t):
    t = t.double().flatten()
    return np.array([t.sum().item(), (t * torch.arange(1, t.numel() + 1, dtype=torch.float64)).sum().item() / t.numel()])


if __name__ == "__main__":
    ref_import.install()
    from models.cross_encoder import CrossGating, Seq2SeqAttention  # noqa: E402  (the reference)
    from models.panns import ConvBlock  # noqa: E402

    out = {}
    for name, cin, cout, shape, psz, ptype in CONV_CASES:
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            blk = make_convblock(ConvBlock, name, cin, cout).to(dt)
            x, g = convblock_inputs(name, cin, shape)
            out[f"{name}/checksum"] = np.concatenate([checksum(x), checksum(blk.conv2.weight.detach())])
            blk.eval()
            with torch.no_grad():
                out[f"{name}/eval_{tag}"] = blk(x.to(dt), pool_size=psz, pool_type=ptype).numpy()
            blk.train()
            xi = x.to(dt).requires_grad_(True)
            y = blk(xi, pool_size=psz, pool_type=ptype)
            dy = torch.randn(y.shape, generator=g)
            y.backward(dy.to(dt))
            out[f"{name}/train_{tag}"] = y.detach().numpy() if tag == "f64" else y.detach().numpy().astype(np.float32)
            out[f"{name}/dx_{tag}"] = sample(xi.grad)
            for n, p in blk.named_parameters():
                out[f"{name}/grad_{tag}/{n}"] = sample(p.grad)
            for n, b in blk.named_buffers():
                out[f"{name}/buf_{tag}/{n}"] = b.numpy()
        print(name, "train out", out[f"{name}/train_f32"].shape,
              "f32 vs f64", np.abs(out[f"{name}/train_f32"] - out[f"{name}/train_f64"]).max())
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        m, q, kv, ql, kl, dout = make_attn(Seq2SeqAttention)
        m = m.to(dt)
        qi, ki = q.to(dt).requires_grad_(True), kv.to(dt).requires_grad_(True)
        o = m(qi, ki, ql, kl)
        o.backward(dout.to(dt))
        out[f"attn/out_{tag}"], out[f"attn/dq_{tag}"], out[f"attn/dkv_{tag}"] = o.detach().numpy(), qi.grad.numpy(), ki.grad.numpy()
        for n, p in m.named_parameters():
            out[f"attn/grad_{tag}/{n}"] = p.grad.numpy()
        out["attn/checksum"] = np.concatenate([checksum(q), checksum(m.h2attn.weight.detach())])
        m, u, s, du, ds = make_gate(CrossGating)
        m = m.to(dt)
        ui, si = u.to(dt).requires_grad_(True), s.to(dt).requires_grad_(True)
        uo, so = m(ui, si)
        torch.autograd.backward([uo, so], [du.to(dt), ds.to(dt)])
        out[f"gate/u_out_{tag}"], out[f"gate/s_out_{tag}"] = uo.detach().numpy(), so.detach().numpy()
        out[f"gate/du_{tag}"], out[f"gate/ds_{tag}"] = ui.grad.numpy(), si.grad.numpy()
        for n, p in m.named_parameters():
            out[f"gate/grad_{tag}/{n}"] = p.grad.numpy()
        out["gate/checksum"] = np.concatenate([checksum(u), checksum(m.fc_u.weight.detach())])
    # keep the fixture small: fp32 copies only where they serve as the round-off floor (outputs), fp64 everywhere
    keep = {k: v for k, v in out.items() if not (k.split("/")[1].startswith(("grad_f32", "buf_f32")))}
    path = os.path.join(HERE, "submodules.npz")
    np.savez_compressed(path, **keep)
    print("wrote", path, os.path.getsize(path))
