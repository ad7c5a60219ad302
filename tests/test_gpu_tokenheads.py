"""Heads with text_level="token" in general (match.ExpNegL2 / match.DotProduct with F.normalize, models/match.py:10-60) and
MaxMarginRankingLoss(fix_norm=False) (losses.py:226-264) against tests/golden/token_heads.npz -- outputs and gradients of the
IMPORTED reference's fp64 twin (tests/golden/make_golden_tokenheads.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(got, want):
    want = torch.as_tensor(want).double()
    return (got.detach().cpu().double() - want).abs().max().item() / (want.abs().max().item() + 1e-30)


@pytest.mark.parametrize("case", ["expnegl2_norm", "expnegl2_raw", "dot_norm", "dot_norm_noscale"])
def test_token_level_heads_vs_reference(dev, golden_dir, case):
    from texttoaudiogrounding_amd.models import match
    g = np.load(f"{golden_dir}/token_heads.npz")
    head = {"expnegl2_norm": match.ExpNegL2(l2norm=True, text_level="token"),
            "expnegl2_raw": match.ExpNegL2(l2norm=False, text_level="token"),
            "dot_norm": match.DotProduct(l2norm=True, scale=True, text_level="token"),
            "dot_norm_noscale": match.DotProduct(l2norm=True, scale=False, text_level="token")}[case]
    a = torch.from_numpy(g["audio"]).to(dev).requires_grad_(True)
    t = torch.from_numpy(g["text"]).to(dev).requires_grad_(True)
    sim = head({"audio_emb": a, "text_emb": {"token_emb": t}})
    sim.backward(torch.from_numpy(g["dsim"]).to(dev))
    floor = np.abs(g[f"{case}/sim_f32"].astype(np.float64) - g[f"{case}/sim_f64"]).max()
    errs = {"sim": (sim.detach().cpu().double() - torch.from_numpy(g[f"{case}/sim_f64"])).abs().max().item(),
            "daudio": rel(a.grad, g[f"{case}/daudio"]), "dtext": rel(t.grad, g[f"{case}/dtext"])}
    print(f"{case}: {({k: f'{v:.1e}' for k, v in errs.items()})} (reference fp32-vs-fp64 sim {floor:.1e})")
    assert errs["sim"] < max(4 * floor, 2e-7) and errs["daudio"] < 5e-6 and errs["dtext"] < 5e-6, errs
    assert torch.isfinite(a.grad).all() and torch.isfinite(t.grad).all()          # the zero-distance row has gradient 0


def test_token_level_head_needs_one_vector_per_frame(dev):
    from texttoaudiogrounding_amd.models import match
    with pytest.raises(RuntimeError):
        match.ExpNegL2(text_level="token")({"audio_emb": torch.zeros(2, 5, 8, device=dev),
                                            "text_emb": {"token_emb": torch.zeros(2, 3, 8, device=dev)}})


@pytest.mark.parametrize("key", ["mm_n6_fix1", "mm_n6_fix0", "mm_n9_fix1", "mm_n9_fix0"])
def test_maxmargin_fix_norm_vs_reference(dev, golden_dir, key):
    from texttoaudiogrounding_amd.losses import MaxMarginRankingLoss
    g = np.load(f"{golden_dir}/token_heads.npz")
    margin, lam = (float(v) for v in g[f"{key}/cfg"])
    x = torch.from_numpy(g[f"{key}/x"]).float().to(dev).requires_grad_(True)
    loss = MaxMarginRankingLoss(margin=margin, fix_norm=key.endswith("fix1"), lamda1=lam)({"sim": x})
    loss.backward()
    assert abs(loss.item() - float(g[f"{key}/loss"])) < 2e-6
    assert rel(x.grad, g[f"{key}/dx"]) < 1e-6
