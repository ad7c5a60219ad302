"""Every similarity reducer of the reference (models/sim_pooling.py:6-204; models/utils.py *_with_lens as selected by
MultiTextBiEncoder.pooling) on the HIP path, against tests/golden/sim_pooling.npz -- outputs and input-gradients of the
IMPORTED reference classes in fp64 -- and at the sentence-level runner's shape (B = 64, T' = 250, 6 phrases)."""
import numpy as np
import pytest
import torch

from oracle import tag_oracle as O

pytestmark = pytest.mark.gpu

PAIR = ["AudioMeanTextMean", "AudioMeanTextSum", "AudioMaxTextMean", "AudioMaxTextMax", "AudioMaxTextSum",
        "AudioMaxTextMeanSum", "AudioLinearSoftTextMean", "AudioLinearSoftTextSum", "AudioExpSoftTextMean",
        "AudioExpSoftTextSum"]


def relerr(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


@pytest.mark.parametrize("name", PAIR)
def test_pair_reducers_golden(dev, golden_dir, name):
    from texttoaudiogrounding_amd.models import sim_pooling
    gold = np.load(f"{golden_dir}/sim_pooling.npz")
    sim = torch.from_numpy(gold["sim"]).float().to(dev).requires_grad_(True)
    out = getattr(sim_pooling, name)()({"sim": sim, "audio_len": gold["audio_len"], "text_len": gold["text_len"]})
    out.backward(torch.from_numpy(gold["dout"]).float().to(dev))
    e_o, e_g = relerr(out, gold[f"{name}/out"]), relerr(sim.grad, gold[f"{name}/dsim"])
    print(f"{name}: out {e_o:.2e} dsim {e_g:.2e}")
    assert e_o < 2e-6 and e_g < 5e-6


@pytest.mark.parametrize("name", ["MultiTextLinearSoft", "MultiTextMax"])
def test_multitext_reducers_golden(dev, golden_dir, name):
    from texttoaudiogrounding_amd.models import sim_pooling
    gold = np.load(f"{golden_dir}/sim_pooling.npz")
    sim = torch.from_numpy(gold["frame_sim"]).float().transpose(1, 2).contiguous().to(dev).requires_grad_(True)   # (B,n,T)
    out = getattr(sim_pooling, name)()({"sim": sim, "audio_len": gold["audio_len"]})
    out.backward(torch.from_numpy(gold["dclip"]).float().to(dev))
    assert relerr(out, gold[f"{name}/out"]) < 2e-6 and relerr(sim.grad, gold[f"{name}/dsim"]) < 5e-6


@pytest.mark.parametrize("mode", ["linear_softmax", "max", "mean", "exp_softmax"])
def test_multitext_biencoder_pooling_modes_golden(dev, golden_dir, mode):
    """The four pooling modes of MultiTextBiEncoder (models/audio_text_model.py:205-215) through ops.SimPoolFunction in the
    layout the model uses ((B*N, T') rows of frame scores)."""
    from texttoaudiogrounding_amd import ops
    gold = np.load(f"{golden_dir}/sim_pooling.npz")
    fs = torch.from_numpy(gold["frame_sim"]).float()                      # (B, T, N) as the reference pools it
    B, T, N = fs.shape
    rows = fs.transpose(1, 2).reshape(B * N, T, 1).contiguous().to(dev).requires_grad_(True)
    al = torch.from_numpy(gold["audio_len"]).long().to(dev)
    clip = ops.SimPoolFunction.apply(rows, al, None, N, 1, ops.POOL_MODES[mode], -1).view(B, N)
    clip.backward(torch.from_numpy(gold["dclip"]).float().to(dev))
    dfs = rows.grad.view(B, N, T).transpose(1, 2)
    assert relerr(clip, gold[f"pool_{mode}/out"]) < 2e-6 and relerr(dfs, gold[f"pool_{mode}/dsim"]) < 5e-6


@pytest.mark.parametrize("am,tm", [("exp_softmax", "mean"), ("max", "mean_sum"), ("linear_softmax", "sum")])
def test_pair_reducers_runner_shape_vs_oracle(dev, am, tm):
    """(64, 64, 250, 6): the matrix align.DotProduct hands to the reducers in the sentence-level runner."""
    from texttoaudiogrounding_amd import ops
    B, T, N = 64, 250, 6
    g = torch.Generator().manual_seed(3)
    sim = torch.rand(B, B, T, N, generator=g) * 0.98 + 0.01
    al = torch.randint(1, T + 1, (B,), generator=g); al[0] = T
    tl = torch.randint(1, N + 1, (B,), generator=g); tl[1] = N
    dout = torch.randn(B, B, generator=g)
    s = sim.to(dev).requires_grad_(True)
    out = ops.SimPoolFunction.apply(s.view(B * B, T, N), al.to(dev), tl.to(dev), B, B, ops.POOL_MODES[am], ops.TEXT_MODES[tm])
    out.view(B, B).backward(dout.to(dev))
    sd = sim.double().requires_grad_(True)
    ref = O.sim_pooling(sd, al, tl, am, tm)
    ref.backward(dout.double())
    assert relerr(out.view(B, B), ref) < 5e-6 and relerr(s.grad, sd.grad) < 2e-5
