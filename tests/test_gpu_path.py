"""Whole-path parity of the HIP hot path (through the reference-shaped modules and the C ABI):
against the golden vectors the imported reference produced (tests/golden/*.npz), against the live
oracle with the HIP path's own dropout masks replayed, and -- at BASELINE.json's clip length --
through frame_sim <= 1e-4 and bit-exact integer segments."""
import numpy as np
import pytest
import torch

from oracle import tag_oracle as O
from texttoaudiogrounding_amd import engine as _engine, functions as _functions   # where tests patch what the nodes look up

pytestmark = pytest.mark.gpu

S_GOLD = 48000


def make_batch(hop):
    b = O.synthetic_batch(2, S_GOLD, seed=1234, ragged=False, hop=hop)
    lens = np.array([S_GOLD, S_GOLD - 5 * hop * 4 - 123])
    b["waveform"][1, lens[1]:] = 0.0
    b["waveform_len"] = lens
    return b


def checksum(t):
    t = t.detach().double().flatten()
    return [float(t.sum()), float(t.abs().max()), float(t[:: max(1, t.numel() // 7)][:7].sum())]


def build_hip_model(st, match, dev):
    from texttoaudiogrounding_amd.models import audio_encoder, audio_text_model, text_encoder
    from texttoaudiogrounding_amd.models import match as match_mod
    text_dim = st["text_encoder.embedding.core.weight"].shape[1]
    mf = match_mod.DotProduct() if match == "dot" else match_mod.ExpNegL2()
    model = audio_text_model.BiEncoder(audio_encoder.Cnn8Rnn(32000), text_encoder.EmbeddingAgg(5221, text_dim), mf,
                                       256 if "audio_proj.weight" in st else text_dim)
    missing = model.load_state_dict(st, strict=False)
    assert not missing.unexpected_keys
    assert all("melspec_extractor" in k for k in missing.missing_keys)
    return model.to(dev)


def gold_state(gold, text_dim=512):
    st = O.init_state(seed=7, text_dim=text_dim, shared_dim=256 if text_dim != 512 else 512, logit_gain=120.0)
    for k in gold.files:
        if k.startswith("before/"):
            st[k[len("before/"):]] = torch.from_numpy(gold[k])
    return st


def sample_grad(g):
    g = g.detach().double().flatten().cpu()
    gi = torch.Generator().manual_seed(99)
    idx = torch.randint(0, g.numel(), (16,), generator=gi)
    return np.concatenate([[g.norm().item(), g.abs().max().item()], g[idx].numpy()])


def assert_grad_close(name, err, floor):
    """Per-tensor gradient rule (errors are max-normalised distances from the fp64 twin; ``floor`` = the fp32 CPU
    reference's / oracle's own distance on the same tensor):

    * tensors ABOVE the last ReLU / max-pool of the encoder (fc1, GRU, embedding, projections): pure round-off ->
      err <= 4 x max(floor, 1e-6);
    * conv-block / bn0 tensors: a ReLU or max-pool decision whose operands differ by less than fp32 rounding can flip in
      ANY fp32 implementation, and one flip moves O(1e-3) of such a tensor's gradient at these 2-clip sizes.  With 0 or 1
      flips per tensor the floor of ONE other fp32 run is not a statistic (it is 1e-6 when that run happened not to
      flip), so here the rule is err <= max(4 x floor, 1e-2); the strict 4 x floor rule for EVERY tensor is asserted
      where flips are counted in thousands: test_full_length_train_step_vs_oracle (B = 6, 10 s) and
      test_benched_size_train_step_vs_fixture (B = 64, 10 s)."""
    if "conv_block" in name or "bn0" in name or ".cnn." in name:
        assert err <= max(4.0 * floor, 1e-2), (name, err, floor)
    else:
        assert err <= 4.0 * max(floor, 1e-6), (name, err, floor)


@pytest.fixture(params=["fp32"])
def conv_math(request):
    """The arithmetic of the 3x3 convolutions the whole-path tests run under: exact fp32 MFMA.  (Rounds 2-5 also ran every test
    here under the opt-in 3 x bf16 splits "x3" / "x9" of conv_x3.hip; they earn no roofline credit and since round 6 the exact fp32
    path is as fast as x3, so they left the default matrix -- their kernels keep their own tests in tests/test_gpu_kernels.py and one
    whole-path smoke test each below.)"""
    from texttoaudiogrounding_amd import ops
    old = ops.CONV_MATH
    ops.CONV_MATH = request.param
    yield request.param
    ops.CONV_MATH = old


@pytest.mark.parametrize("split", ["x3", "x9"])
def test_split_arithmetic_whole_path_smoke(dev, golden_dir, split, monkeypatch):
    """One whole-path check per opt-in split arithmetic: the reference's eval fixture under CONV_MATH = x3 / x9."""
    from texttoaudiogrounding_amd import ops
    monkeypatch.setattr(ops, "CONV_MATH", split)
    test_golden_eval(dev, golden_dir, split)


def test_golden_eval(dev, golden_dir, conv_math):
    gold = np.load(f"{golden_dir}/cnn8rnn_dot_eval.npz")
    st = gold_state(gold)
    batch = make_batch(320)
    chk = checksum(batch["waveform"]) + checksum(batch["text"].float()) + checksum(st["audio_encoder.fc1.weight"])
    assert np.allclose(chk, gold["input_checksum"], rtol=1e-9), "seeded inputs drifted from the fixture"
    model = build_hip_model(st, "dot", dev).eval()
    with torch.no_grad():
        emb = model.audio_encoder({"waveform": batch["waveform"].to(dev), "waveform_len": batch["waveform_len"],
                                   "specaug": False})
        out = model({"waveform": batch["waveform"].to(dev), "waveform_len": batch["waveform_len"],
                     "text": batch["text"], "text_len": batch["text_len"], "specaug": False})
    assert np.array_equal(out["length"].numpy(), gold["length"])                   # A5: integer, exact
    e_err = (emb["embedding"].cpu() - torch.from_numpy(gold["embedding_f64_as_f32"])).abs().max().item()
    fs_err = np.abs(out["frame_sim"].cpu().numpy().astype(np.float64) - gold["frame_sim_f64"]).max()
    fs = out["frame_sim"].cpu().double()
    logit_err = np.abs(torch.log(fs / (1 - fs)).numpy() - gold["logit_f64"]).max()
    print(f"eval: embedding err {e_err:.2e}, frame_sim err {fs_err:.2e}, logit err {logit_err:.2e}")
    assert e_err < 1e-4            # GRU outputs are in [-1,1]
    assert fs_err < 1e-4           # north_star tolerance (fp32, 1e-4)
    assert logit_err < 2e-3        # logits span +-3.3


def test_golden_eval_at_the_benched_clip_length(dev, golden_dir, conv_math):
    """The HIP path against the IMPORTED REFERENCE at 10 s clips (tests/golden/make_golden_10s.py: B = 2, second clip ragged, eval
    mode): the frame chain F = 1001 -> 500 -> 250 (models/audio_encoder.py:202-227) pinned to the reference itself.  frame_sim
    within the north_star tolerance of the reference's fp64 twin, `length` exact, and the reference's OWN segment lists
    (utils/eval_util.py functions on its fp32 scores, 50 thresholds, n_connect 13) reproduced bit-exactly by tag_segments at
    every threshold that no score sits on (margin > 2 x (this path's error + the reference's own fp32 error); >= 98 of 100)."""
    from tests.test_oracle_golden import gold_10s_inputs, gold_10s_segments
    from texttoaudiogrounding_amd.utils import eval_util
    gold = np.load(f"{golden_dir}/cnn8rnn_dot_eval_10s.npz")
    st, batch = gold_10s_inputs(gold)
    model = build_hip_model(st, "dot", dev).eval()
    with torch.no_grad():
        emb = model.audio_encoder({"waveform": batch["waveform"].to(dev), "waveform_len": batch["waveform_len"],
                                   "specaug": False})
        out = model({"waveform": batch["waveform"].to(dev), "waveform_len": batch["waveform_len"],
                     "text": batch["text"], "text_len": batch["text_len"], "specaug": False})
    assert np.array_equal(out["length"].numpy(), gold["length"]) and out["frame_sim"].shape == (2, 250)
    e_err = (emb["embedding"].cpu()[:, ::5] - torch.from_numpy(gold["embedding_f64_as_f32_every5"])).abs().max().item()
    fs_err = np.abs(out["frame_sim"].cpu().numpy().astype(np.float64) - gold["frame_sim_f64"]).max()
    ref32_err = np.abs(gold["frame_sim_f32"].astype(np.float64) - gold["frame_sim_f64"]).max()
    assert e_err < 1e-4 and fs_err < 1e-4
    th = eval_util.eval_thresholds(50)
    assert np.allclose(th, gold["thresholds"])
    got = eval_util.segments_for_thresholds(out["frame_sim"], th, 1, eval_util.n_connect_for(0.04))
    checked = 0
    for b in range(2):
        for ti in range(len(th)):
            if gold["margin"][b, ti] <= 2 * (fs_err + ref32_err):
                continue
            assert np.array_equal(got[b][ti], gold_10s_segments(gold, b, ti)), (b, ti)
            checked += 1
    print(f"10 s reference fixture: embedding err {e_err:.2e}, frame_sim err {fs_err:.2e} (reference fp32: {ref32_err:.2e}); "
          f"{checked} of 100 (clip, threshold) segment lists checked, all identical to the reference's")
    assert checked >= 98


def test_crnn_golden_eval_at_the_benched_clip_length(dev, golden_dir):
    """CrnnEncoder (the strong eg_config's encoder) against the IMPORTED REFERENCE at 10 s clips (F = 501 -> 250 -> 125): frame_sim
    within the north_star tolerance of the reference's fp64 twin, `length` exact."""
    from tests.test_oracle_golden import gold_10s_crnn_inputs
    gold = np.load(f"{golden_dir}/crnn_expnegl2_eval_10s.npz")
    st, batch = gold_10s_crnn_inputs(gold)
    model = build_crnn_model(st, dev).eval()
    with torch.no_grad():
        out = model({"waveform": batch["waveform"].to(dev), "waveform_len": batch["waveform_len"], "text": batch["text"],
                     "text_len": batch["text_len"], "specaug": False})
    assert np.array_equal(out["length"].numpy(), gold["length"]) and out["frame_sim"].shape == (2, 125)
    fs_err = np.abs(out["frame_sim"].cpu().numpy().astype(np.float64) - gold["frame_sim_f64"]).max()
    print(f"crnn 10 s reference fixture: frame_sim err {fs_err:.2e}")
    assert fs_err < 1e-4


def test_golden_eval_at_the_benched_clip_length_bf16_conv_math(dev, golden_dir, monkeypatch):
    """The same 10 s reference fixture under the bf16 conv arithmetic (configs[2]'s): the frame probabilities stay within bf16
    tolerance of the reference's fp64 twin and are visibly NOT the fp32 path's."""
    from tests.test_oracle_golden import gold_10s_inputs
    from texttoaudiogrounding_amd import ops
    gold = np.load(f"{golden_dir}/cnn8rnn_dot_eval_10s.npz")
    st, batch = gold_10s_inputs(gold)
    model = build_hip_model(st, "dot", dev).eval()
    monkeypatch.setattr(ops, "CONV_MATH", "bf16")
    with torch.no_grad():
        out = model({"waveform": batch["waveform"].to(dev), "waveform_len": batch["waveform_len"], "text": batch["text"],
                     "text_len": batch["text_len"], "specaug": False})
    assert np.array_equal(out["length"].numpy(), gold["length"])
    fs_err = np.abs(out["frame_sim"].cpu().numpy().astype(np.float64) - gold["frame_sim_f64"]).max()
    print(f"10 s reference fixture, bf16 conv math: frame_sim err {fs_err:.2e}")
    assert 1e-6 < fs_err < 3e-2


def test_golden_eval_bf16_conv_math(dev, golden_dir):
    """BASELINE configs[2] arithmetic for the convolutions (operands rounded to bf16, fp32 accumulate, everything else
    fp32): the frame probabilities stay within bf16 tolerance of the fp64 golden values."""
    from texttoaudiogrounding_amd import ops
    gold = np.load(f"{golden_dir}/cnn8rnn_dot_eval.npz")
    model = build_hip_model(gold_state(gold), "dot", dev).eval()
    batch = make_batch(320)
    old = ops.CONV_MATH
    ops.CONV_MATH = "bf16"
    try:
        with torch.no_grad():
            out = model({"waveform": batch["waveform"].to(dev), "waveform_len": batch["waveform_len"],
                         "text": batch["text"], "text_len": batch["text_len"], "specaug": False})
    finally:
        ops.CONV_MATH = old
    fs_err = np.abs(out["frame_sim"].cpu().numpy().astype(np.float64) - gold["frame_sim_f64"]).max()
    print(f"bf16 conv math: frame_sim err {fs_err:.2e}")
    assert 1e-6 < fs_err < 3e-2         # visibly bf16 (not silently the fp32 path), yet within bf16 tolerance


def test_golden_train_step_grads(dev, golden_dir, conv_math):
    """Train-mode BN, dropout off: loss, frame_sim, running stats and every parameter gradient
    against the reference's fp64 twin.  Gradient tolerance is per tensor, normalised by max|g|:
    the reference's own fp32-vs-fp64 error with train-mode BN at B=2 is up to 4e-3."""
    gold = np.load(f"{golden_dir}/cnn8rnn_dot_train.npz")
    st = gold_state(gold)
    batch = make_batch(320)
    model = build_hip_model(st, "dot", dev).train()
    model.audio_encoder.dropout_p = (0.0, 0.0)
    from texttoaudiogrounding_amd.runner import StrongRunner
    runner = StrongRunner(model, device=str(dev))
    loss = runner.forward_backward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    assert abs(loss.item() - float(gold["loss_f64"])) < 2e-5
    worst, worst32 = 0.0, 0.0
    for name, p in model.named_parameters():
        want = gold[f"grad_f64/{name}"]
        ref32 = gold[f"grad_f32/{name}"]
        got = sample_grad(p.grad)
        scale = want[1] + 1e-30
        err = np.abs(got[2:] - want[2:]).max() / scale
        nerr = abs(got[0] - want[0]) / (want[0] + 1e-30)
        worst = max(worst, err, nerr)
        e32 = np.abs(ref32[2:] - want[2:]).max() / scale
        worst32 = max(worst32, e32)
        print(f"  {name:55s} hip {err:.2e} (norm {nerr:.2e})  reference-f32 {e32:.2e}")
        assert_grad_close(name, err, e32)
        assert nerr < 1e-2, (name, nerr)
    print(f"train: loss {loss.item():.7f} (ref f64 {float(gold['loss_f64']):.7f}); worst grad err vs f64 "
          f"{worst:.2e} (reference's own f32 {worst32:.2e})")
    sd = model.state_dict()
    for k in gold.files:
        if k.startswith("after/"):
            name = k[len("after/"):]
            assert np.allclose(sd[name].cpu().numpy(), gold[k], rtol=2e-4, atol=1e-5), name


def test_golden_train_step_grads_winograd_forced(dev, golden_dir, monkeypatch):
    """The reference-imported 2-clip training fixture (tests/golden/cnn8rnn_dot_train.npz: /root/reference's own loss and fp64
    gradients) on the DEFAULT training arithmetic of the benched size: ops.WINO_MIN_WORK = 1 puts the fused Winograd kernels
    (csrc/conv_wino_fused.hip) into every layer with >= 64 channels -- 21 launches -- where the dispatch rule would keep this small
    step on the direct kernels.  Loss and running statistics against the reference's values; every gradient tensor (a) against the
    reference's fp64 gradient at the test's usual per-tensor bound where no decision flipped at or above its layer, and (b) with the
    step's OWN ReLU / arg-max decisions imposed on the fp64 oracle (oracle.conv_block(decisions=...)) within round-off: 4 x
    max(floor, 1e-6) above the conv stack, max(4 x floor, 1e-5) for conv blocks / bn0 -- a flipped near-tie is explained, not hidden."""
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd.runner import StrongRunner
    gold = np.load(f"{golden_dir}/cnn8rnn_dot_train.npz")
    st = gold_state(gold)
    batch = make_batch(320)
    monkeypatch.setattr(ops, "WINO_MIN_WORK", 1)
    captured = []
    orig_fwd = ops.Cnn8RnnFunction.forward

    def fwd(ctx, *a):
        y = orig_fwd(ctx, *a)
        captured.append(hip_decisions(ctx.saved))
        return y

    monkeypatch.setattr(ops.Cnn8RnnFunction, "forward", staticmethod(fwd))
    wino0 = ops.WINO_LAUNCHES
    model = build_hip_model(st, "dot", dev).train()
    model.audio_encoder.dropout_p = (0.0, 0.0)
    runner = StrongRunner(model, device=str(dev))
    loss = runner.forward_backward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    assert ops.WINO_LAUNCHES - wino0 == 21, ops.WINO_LAUNCHES - wino0
    assert abs(loss.item() - float(gold["loss_f64"])) < 2e-5
    dec = captured[0]

    def oracle_grads(dt, decisions, taps=None):
        st_o = O.state_to(st, dt, requires_grad=True)
        bo = dict(batch)
        bo["waveform"], bo["label"] = batch["waveform"].to(dt), batch["label"].to(dt)
        oloss, _ = O.train_step_loss(st_o, bo, "dot", "cnn8rnn", True, (0.0, 0.0), dict(decisions or {}), taps)
        oloss.backward()
        return oloss.item(), {k: v.grad.double() for k, v in st_o.items() if v.is_floating_point() and v.grad is not None}, st_o

    taps = {}
    l64, g64, st64 = oracle_grads(torch.float64, None, taps)
    assert abs(l64 - float(gold["loss_f64"])) < 1e-6          # the oracle IS the reference on this fixture (frontend round-off apart)
    own = oracle_decisions(st64, taps)
    flips = {k: int((own[k] != dec[k]).sum()) for k in dec}
    _, g64i, _ = oracle_grads(torch.float64, dec)
    _, g32i, _ = oracle_grads(torch.float32, dec)
    order = [f"conv_block{i}" for i in range(1, 5)]

    def flips_at_or_above(name):
        lo = 0 if "bn0" in name else next((i for i, b in enumerate(order) if b in name), 4)
        return sum(v for k, v in flips.items() if any(b in k for b in order[lo:]) or "fc1" in k)

    worst_i = 0.0
    for name, p in model.named_parameters():
        g = p.grad.cpu().double()
        scale_i = g64i[name].abs().max().item() + 1e-30
        err_i, e32_i = (g - g64i[name]).abs().max().item() / scale_i, (g32i[name] - g64i[name]).abs().max().item() / scale_i
        conv = "conv_block" in name or "bn0" in name
        assert err_i <= (max(4.0 * e32_i, 1e-5) if conv else 4.0 * max(e32_i, 1e-6)), (name, err_i, e32_i)
        worst_i = max(worst_i, err_i)
        if flips_at_or_above(name) == 0:                        # untouched by a flip: the reference's own fp64 gradient, sampled
            want, ref32 = gold[f"grad_f64/{name}"], gold[f"grad_f32/{name}"]
            got = sample_grad(p.grad)
            sc = want[1] + 1e-30
            assert_grad_close(name, np.abs(got[2:] - want[2:]).max() / sc, np.abs(ref32[2:] - want[2:]).max() / sc)
    print(f"reference train fixture on the fused Winograd kernels: loss {loss.item():.7f} (reference {float(gold['loss_f64']):.7f}); "
          f"flipped decisions { {k: v for k, v in flips.items() if v} }; worst gradient error with the step's decisions imposed {worst_i:.2e}")
    sd = model.state_dict()
    for k in gold.files:
        if k.startswith("after/"):
            name = k[len("after/"):]
            assert np.allclose(sd[name].cpu().numpy(), gold[k], rtol=2e-4, atol=1e-5), name


def _window_first_argmax(r, ph, pw):
    """r (B,C,H,W) -> (B,C,Ho,Wo) position dh * pw + dw of the FIRST maximum of each pooling window (scan order h then w: ATen's
    max_pool2d and bn_pool.hip pick that one; torch.argmax documents first-index tie breaking)."""
    B, C, H, W = r.shape
    Ho, Wo = H // ph, W // pw
    win = r[:, :, :Ho * ph, :Wo * pw].reshape(B, C, Ho, ph, Wo, pw).permute(0, 1, 2, 4, 3, 5).reshape(B, C, Ho, Wo, ph * pw)
    return win.argmax(-1)


def hip_decisions(saved, pools=((2, 2), (2, 2), (1, 2), (1, 2))):
    """The hard decisions the HIP step took, recomputed from ITS OWN saved tensors (raw conv outputs + the BatchNorm scale / shift
    its kernels apply as fmaf(y, scale, shift)): sign of the exact y * scale + shift (fp64 product of fp32 factors is exact) = sign
    of the kernels' fmaf; arg-max over the fp32-rounded values.  Keys as oracle.conv_block(decisions=...) reads them."""
    dec = {}
    for i, (x, y1, s1, y2, s2, _, _) in enumerate(saved["acts"], start=1):
        pre = f"audio_encoder.conv_block{i}."
        for j, (y, stt) in enumerate(((y1, s1), (y2, s2)), start=1):
            a = (y.double() * stt.scale.double() + stt.shift.double()).permute(0, 3, 1, 2)          # NCHW
            dec[f"relu/{pre}bn{j}"] = (a > 0).cpu()
        dec[f"argmax/{pre}"] = _window_first_argmax(torch.relu(a.float()), *pools[i - 1]).cpu()
    fc = saved["fc"]                                                                            # (B * T', 512) after the ReLU
    B = saved["acts"][0][1].shape[0]
    dec["relu/audio_encoder.fc1"] = (fc > 0).view(B, -1, fc.shape[1]).cpu()
    return dec


def oracle_decisions(st, taps, pools=((2, 2), (2, 2), (1, 2), (1, 2))):
    """The same decisions as the oracle run that filled `taps` took them (train-mode BatchNorm of its own raw conv outputs)."""
    import torch.nn.functional as F
    dec = {}
    for i in range(1, 5):
        pre = f"audio_encoder.conv_block{i}."
        for j in (1, 2):
            y = taps[pre + f"conv{j}"].detach()
            a = F.batch_norm(y, None, None, st[pre + f"bn{j}.weight"].detach().to(y.dtype), st[pre + f"bn{j}.bias"].detach().to(y.dtype),
                             True, 0.0, 1e-5)
            dec[f"relu/{pre}bn{j}"] = a > 0
        dec[f"argmax/{pre}"] = _window_first_argmax(torch.relu(a), *pools[i - 1])
    dec["relu/audio_encoder.fc1"] = taps["fc1"].detach() > 0
    return dec


@pytest.mark.parametrize("algo", ["default", "winograd-forced"])
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 1234])
def test_proj_expnegl2_dropout_replay_vs_oracle(dev, seed, algo, monkeypatch):
    """Cnn8Rnn(512)+EmbeddingAgg(256)+audio/text proj+ExpNegL2, train mode WITH dropout, over seven generator seeds (round 4 kept
    ONE hand-picked seed: "a realisation without a flipped decision").  The HIP path's keep-masks are exported (tag_dropout_mask)
    and replayed in the oracle.  At this 2-clip size a conv-block gradient hangs on single ReLU / arg-max decisions whose operands
    two fp32 implementations can round to different sides; what is asserted for EVERY seed:

    1. loss within 2e-5 of the fp64 oracle; every tensor ABOVE the last ReLU of the conv stack (fc1, GRU, embedding, projections)
       within 4 x max(floor, 1e-6) of the plain fp64 oracle -- no decision of the conv stack reaches them in backward;
    2. with the HIP step's OWN decisions imposed on the oracle (oracle.conv_block(decisions=...): every BatchNorm+ReLU mask, every
       max-pool arg-max, recomputed from the step's saved raw conv outputs), EVERY tensor is within round-off of the fp64 oracle:
       4 x max(floor, 1e-6) above the conv stack, max(4 x floor, 1e-5) for conv blocks / bn0 (floor = the fp32 CPU oracle under
       the same decisions; measured 3.5e-6 ... 4.7e-6 over the seven seeds where the plain comparison shows up to 4.6e-2) -- a
       bound a thousand times below round 4's 1e-2: all that separates the HIP path from the reference arithmetic is round-off
       plus the counted decisions;
    3. a conv-block tensor that misses the strict bound against the PLAIN oracle must have at least one flipped decision at or
       above its layer (the error is explained, not tolerated)."""
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd.runner import StrongRunner
    st = O.init_state(seed=11, text_dim=256, shared_dim=256, logit_gain=120.0)
    batch = make_batch(320)
    # "winograd-forced" (round 6): the work threshold of ops.WINO_MIN_WORK set to 1, so that this 2-clip step runs the arithmetic
    # the benched size runs by default -- the fused Winograd kernels in every layer with >= 64 channels (21 launches) -- against the
    # same rules: a near-tied decision that this arithmetic rounds to the other side is counted and explained, not shielded by the
    # dispatch rule
    wino0 = ops.WINO_LAUNCHES
    if algo == "winograd-forced":
        monkeypatch.setattr(ops, "WINO_MIN_WORK", 1)
    captured = []
    orig_fwd = ops.Cnn8RnnFunction.forward

    def fwd(ctx, *a):
        y = orig_fwd(ctx, *a)
        captured.append(hip_decisions(ctx.saved))            # now: backward frees the saved activations
        return y

    monkeypatch.setattr(ops.Cnn8RnnFunction, "forward", staticmethod(fwd))
    torch.manual_seed(seed)                       # dropout seeds are drawn from torch's global generator
    model = build_hip_model(st, "expnegl2", dev).train()
    runner = StrongRunner(model, device=str(dev))
    loss = runner.forward_backward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    dec = captured[0]
    assert ops.WINO_LAUNCHES - wino0 == (21 if algo == "winograd-forced" else 0)
    info = model.audio_encoder._last_dropout
    assert info["p"] == (0.2, 0.5)
    shapes = [(2, 75, 32, 64), (2, 37, 16, 128), (2, 37, 8, 256), (2, 37, 4, 512)]
    masks = {}
    for i, shp in enumerate(shapes):
        m = ops.dropout_mask(info["seeds"][i], shp, 0.2, dev, pooled=True).cpu()
        masks[f"drop{i + 1}"] = m.permute(0, 3, 1, 2).double()
        assert 0.7 < m.float().mean().item() < 0.9
    masks["drop5"] = ops.dropout_mask(info["seeds"][4], (2, 37, 512), 0.5, dev).cpu().double()

    def oracle_grads(dt, decisions, taps=None):
        st_o = O.state_to(st, dt, requires_grad=True)
        bo = dict(batch)
        bo["waveform"], bo["label"] = batch["waveform"].to(dt), batch["label"].to(dt)
        mk = {k: v.to(dt) for k, v in masks.items()}
        mk.update(decisions or {})
        oloss, _ = O.train_step_loss(st_o, bo, "expnegl2", "cnn8rnn", True, None, mk, taps)
        oloss.backward()
        return oloss.item(), {k: v.grad.double() for k, v in st_o.items() if v.is_floating_point() and v.grad is not None}, st_o

    taps = {}
    l64, g64, st64 = oracle_grads(torch.float64, None, taps)
    _, g32, _ = oracle_grads(torch.float32, None)
    assert abs(loss.item() - l64) < 2e-5
    own = oracle_decisions(st64, taps)
    flips = {k: int((own[k] != dec[k]).sum()) for k in dec}
    _, g64i, _ = oracle_grads(torch.float64, dec)
    _, g32i, _ = oracle_grads(torch.float32, dec)
    # layers whose decisions can reach a tensor's gradient: its own block's and everything above it (backward flows downwards)
    order = [f"conv_block{i}" for i in range(1, 5)]

    def flips_at_or_above(name):
        if "bn0" in name:
            lo = 0
        else:
            lo = next((i for i, b in enumerate(order) if b in name), 4)
        return sum(v for k, v in flips.items() if any(b in k for b in order[lo:]) or "fc1" in k)

    worst_plain, worst_imposed = 0.0, 0.0
    for name, p in model.named_parameters():
        g = p.grad.cpu().double()
        scale = g64[name].abs().max().item() + 1e-30
        err, e32 = (g - g64[name]).abs().max().item() / scale, (g32[name] - g64[name]).abs().max().item() / scale
        scale_i = g64i[name].abs().max().item() + 1e-30
        err_i, e32_i = (g - g64i[name]).abs().max().item() / scale_i, (g32i[name] - g64i[name]).abs().max().item() / scale_i
        conv = "conv_block" in name or "bn0" in name
        print(f"  {name:55s} plain {err:.2e} (cpu-f32 {e32:.2e})   decisions imposed {err_i:.2e} (cpu-f32 {e32_i:.2e})")
        assert err_i <= (max(4.0 * e32_i, 1e-5) if conv else 4.0 * max(e32_i, 1e-6)), (name, err_i, e32_i)   # rule 2: every tensor
        if not conv:
            assert err <= 4.0 * max(e32, 1e-6), (name, err, e32)                          # rule 1
        elif err > 4.0 * max(e32, 1e-6):
            assert flips_at_or_above(name) > 0, (name, err, e32, flips)                   # rule 3
        worst_plain, worst_imposed = max(worst_plain, err), max(worst_imposed, err_i)
    print(f"seed {seed} [{algo}]: flipped decisions vs the fp64 oracle {sum(flips.values())} ({ {k: v for k, v in flips.items() if v} }); "
          f"worst error plain {worst_plain:.2e}, with the HIP decisions imposed {worst_imposed:.2e}")


@pytest.mark.parametrize("freeze_cnn", [False, True])
def test_frozen_batchnorm_and_frozen_cnn_train_step(dev, golden_dir, freeze_cnn):
    """The reference's fine-tuning switches (models/audio_encoder.py:90-97,159-169): Cnn8Rnn(freeze_bn=True).train() keeps every
    BatchNorm in eval mode (running statistics normalise, nothing is updated) while dropout stays on; freeze_cnn=True additionally
    freezes everything below the GRU.  One training step (dropout off for a mask-free comparison) against the fp64 oracle with
    bn_training=False: loss, every trainable gradient, running statistics and counters untouched, frozen parameters without a
    gradient -- and with the CNN frozen the engine skips the conv stack's backward altogether."""
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd.models import audio_encoder, audio_text_model, match as match_mod, text_encoder
    from texttoaudiogrounding_amd.runner import StrongRunner
    gold = np.load(f"{golden_dir}/cnn8rnn_dot_eval.npz")
    st = gold_state(gold)                              # calibrated running statistics: eval-mode BatchNorm is well conditioned
    batch = make_batch(320)
    model = audio_text_model.BiEncoder(audio_encoder.Cnn8Rnn(32000, freeze_cnn=freeze_cnn, freeze_bn=True),
                                       text_encoder.EmbeddingAgg(5221, 512), match_mod.DotProduct(), 512)
    model.load_state_dict(st, strict=False)
    model = model.to(dev).train()
    ae = model.audio_encoder
    assert ae.training and not ae.bn0.training and not ae.conv_block3.bn2.training
    ae.dropout_p = (0.0, 0.0)
    before = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}
    launched = []
    orig = _functions.conv3x3_wgrad                    # (patched where the Cnn8Rnn node looks it up)
    _functions.conv3x3_wgrad = lambda *a, **k: (launched.append(1), orig(*a, **k))[1]
    try:
        runner = StrongRunner(model, device=str(dev))
        loss = runner.forward_backward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    finally:
        _functions.conv3x3_wgrad = orig
    assert (len(launched) == 0) == freeze_cnn          # frozen CNN: not one weight-gradient conv was launched
    grads = {}
    for dt in (torch.float64, torch.float32):
        st_o = O.state_to(st, dt, requires_grad=True)
        bo = dict(batch)
        bo["waveform"], bo["label"] = batch["waveform"].to(dt), batch["label"].to(dt)
        oloss, _ = O.train_step_loss(st_o, bo, "dot", "cnn8rnn", True, (0.0, 0.0), bn_training=False)
        oloss.backward()
        grads[dt] = {k: v.grad.double() for k, v in st_o.items() if v.is_floating_point() and v.grad is not None}
        if dt == torch.float64:
            assert abs(loss.item() - oloss.item()) < 2e-5
    for name, p in model.named_parameters():
        frozen = freeze_cnn and name.startswith("audio_encoder.") and ".rnn." not in name
        assert p.requires_grad != frozen
        if frozen:
            assert p.grad is None
            continue
        g64, g32 = grads[torch.float64][name], grads[torch.float32][name]
        scale = g64.abs().max().item() + 1e-30
        err = (p.grad.cpu().double() - g64).abs().max().item() / scale
        assert_grad_close(name, err, (g32 - g64).abs().max().item() / scale)
    after = model.state_dict()
    assert all(torch.equal(after[k], v) for k, v in before.items())          # eval-mode BatchNorm updates nothing


def test_full_length_frame_sim_and_segments(dev, conv_math):
    """BASELINE clip length (10 s @ 32 kHz -> T'=250), eval mode, B=3 ragged: frame_sim within 1e-4 of the
    CPU oracle and bit-exact integer segments at all 50 thresholds (n_connect 13 @ 0.04 s)."""
    from texttoaudiogrounding_amd.utils import eval_util
    st = O.init_state(seed=21, logit_gain=120.0)
    batch = O.synthetic_batch(3, 320000, seed=77, ragged=True)
    # calibrated running statistics so eval-mode BN is well conditioned
    model = build_hip_model(st, "dot", dev)
    model.train()
    model.audio_encoder.dropout_p = (0.0, 0.0)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 1.0
    inp = {"waveform": batch["waveform"].to(dev), "waveform_len": batch["waveform_len"], "text": batch["text"],
           "text_len": batch["text_len"], "specaug": False}
    with torch.no_grad():
        model(inp)          # momentum 1.0: running stats <- batch stats
    model.eval()
    with torch.no_grad():
        out = model(inp)
    st2 = {k: v.detach().cpu() for k, v in model.state_dict().items() if "melspec" not in k}
    oout = O.biencoder_forward(st2, batch, "dot", "cnn8rnn", training=False)
    assert torch.equal(out["length"], oout["length"]) and out["frame_sim"].shape == (3, 250)
    fs, ofs = out["frame_sim"].cpu(), oout["frame_sim"]
    err = (fs - ofs).abs().max().item()
    print(f"full-length frame_sim err {err:.2e}; range [{ofs.min():.3f}, {ofs.max():.3f}]")
    assert err < 1e-4
    th = eval_util.eval_thresholds(50)
    got = eval_util.segments_for_thresholds(out["frame_sim"], th, 1, eval_util.n_connect_for(0.04))
    mismatched = 0
    for b in range(3):
        for ti, t in enumerate(th):
            want = O.segments(ofs[b].numpy(), t, 1, 13)
            mine_on_ref = O.segments(fs[b].numpy(), t, 1, 13)
            assert np.array_equal(got[b][ti], mine_on_ref)          # kernel == oracle on identical scores
            mismatched += int(not np.array_equal(got[b][ti], want))
    # asserted budget: at these seeds no oracle score lies within the HIP path's frame_sim error of a threshold, so every
    # one of the 150 (clip, threshold) segment lists must be identical to the oracle's (north_star: bit-exact indices)
    print(f"segments: {mismatched} of {3 * len(th)} (clip,threshold) pairs differ from the oracle's")
    assert mismatched == 0


# ------------------------------------------------------------------------------------------- CrnnEncoder (row A1')
def crnn_state(gold):
    st = O.init_crnn_state(seed=7)
    g = torch.Generator().manual_seed(8)
    st["text_encoder.embedding.core.weight"] = (torch.rand(5221, 256, generator=g) * 2 - 1) * 0.9
    for k in list(st):
        if k.endswith(".0.weight"):
            st[k] = 0.5 + torch.rand(st[k].shape, generator=g)
        if k.endswith(".0.bias"):
            st[k] = 0.2 * torch.randn(st[k].shape, generator=g)
    for k in gold.files:
        if k.startswith("before/"):
            st[k[len("before/"):]] = torch.from_numpy(gold[k])
    return st


def build_crnn_model(st, dev):
    """The strong eg_config verbatim: CrnnEncoder(256) + EmbeddingAgg(256, mean) + ExpNegL2 (cdur_w2vmean.yaml)."""
    from texttoaudiogrounding_amd.models import audio_encoder, audio_text_model, match, text_encoder
    model = audio_text_model.BiEncoder(audio_encoder.CrnnEncoder(sample_rate=32000, embed_dim=256),
                                       text_encoder.EmbeddingAgg(5221, 256), match.ExpNegL2(), 256)
    missing = model.load_state_dict(st, strict=False)
    assert not missing.unexpected_keys and all("melspec_extractor" in k for k in missing.missing_keys)
    assert not hasattr(model, "audio_proj")
    return model.to(dev)


def assert_crnn_grad_close(name, err, nerr, floor):
    """CrnnEncoder has no hard decision (LeakyReLU is continuous, LPPool2d smooth), so every gradient tensor is pure
    round-off: max-normalised distance from the fp64 twin and relative norm error <= 4 x max(floor, 1e-6), where ``floor``
    is the imported fp32 reference's (or the fp32 oracle's) own distance from fp64 on the same tensor."""
    bound = 4.0 * max(float(floor), 1e-6)
    assert err <= bound and nerr <= bound, (name, err, nerr, floor)


def test_crnn_golden_eval(dev, golden_dir):
    gold = np.load(f"{golden_dir}/crnn_expnegl2_eval.npz")
    model = build_crnn_model(crnn_state(gold), dev).eval()
    batch = make_batch(640)
    with torch.no_grad():
        emb = model.audio_encoder({"waveform": batch["waveform"].to(dev), "waveform_len": batch["waveform_len"]})
        out = model({"waveform": batch["waveform"].to(dev), "waveform_len": batch["waveform_len"],
                     "text": batch["text"], "text_len": batch["text_len"], "specaug": False})
    assert np.array_equal(out["length"].numpy(), gold["length"])
    e_err = (emb["embedding"].cpu() - torch.from_numpy(gold["embedding_f64_as_f32"])).abs().max().item()
    fs_err = np.abs(out["frame_sim"].cpu().numpy().astype(np.float64) - gold["frame_sim_f64"]).max()
    print(f"crnn eval: embedding err {e_err:.2e}, frame_sim err {fs_err:.2e}")
    assert e_err < 1e-4 and fs_err < 1e-4


def test_crnn_golden_train_step_grads(dev, golden_dir):
    gold = np.load(f"{golden_dir}/crnn_expnegl2_train.npz")
    model = build_crnn_model(crnn_state(gold), dev).train()
    model.audio_encoder.dropout_p = 0.0
    from texttoaudiogrounding_amd.runner import StrongRunner
    runner = StrongRunner(model, device=str(dev))
    batch = make_batch(640)
    loss = runner.forward_backward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    assert abs(loss.item() - float(gold["loss_f64"])) < 2e-5
    worst = 0.0
    for name, p in model.named_parameters():
        want, ref32 = gold[f"grad_f64/{name}"], gold[f"grad_f32/{name}"]
        got = sample_grad(p.grad)
        scale = want[1] + 1e-30
        err = np.abs(got[2:] - want[2:]).max() / scale
        nerr = abs(got[0] - want[0]) / (want[0] + 1e-30)
        e32 = max(np.abs(ref32[2:] - want[2:]).max() / scale, abs(ref32[0] - want[0]) / (want[0] + 1e-30))
        print(f"  {name:45s} hip {err:.2e} (norm {nerr:.2e})  reference-f32 {e32:.2e}")
        worst = max(worst, err, nerr)
        assert_crnn_grad_close(name, err, nerr, e32)
    sd = model.state_dict()
    for k in gold.files:
        if k.startswith("after/"):
            assert np.allclose(sd[k[len("after/"):]].cpu().numpy(), gold[k], rtol=2e-4, atol=1e-5), k
    print(f"crnn train: loss {loss.item():.7f} (ref f64 {float(gold['loss_f64']):.7f}); worst grad err {worst:.2e}")


def test_crnn_benched_size_train_step_vs_fixture(dev, golden_dir, monkeypatch):
    """CrnnEncoder parity AT THE SIZE `bench.py --crnn` times (the strong eg_config as written: B = 64, 10 s clips, ragged
    lengths, train-mode BatchNorm, Dropout(0.3) ON; models/audio_encoder.py:25-86 in the reference).
    tests/golden/crnn_b64_train_step.npz holds the CPU oracle's fp64 step and its own fp32 step with THIS dropout mask
    (make_golden_crnn_b64.py replays the HIP generator from the same seed): loss <= 2e-5, frame_sim <= 1e-4, every
    gradient tensor within 4 x the fp32 oracle's own distance from fp64 (round-off only: no hard decisions in this encoder)."""
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd.runner import StrongRunner
    from tests.golden.make_golden_crnn_b64 import BATCH_SEED, HOP, P_DROP, crnn_b64_state
    gold = np.load(f"{golden_dir}/crnn_b64_train_step.npz")
    st = crnn_b64_state()
    batch = O.synthetic_batch(64, 320000, seed=BATCH_SEED, ragged=True, hop=HOP)
    chk = checksum(batch["waveform"]) + checksum(batch["text"].float()) + checksum(st["audio_encoder.gru.weight_ih_l0"])
    assert np.allclose(chk, gold["input_checksum"], rtol=1e-9), "seeded inputs drifted from the fixture"
    monkeypatch.setattr(_functions, "new_seed", lambda: int(gold["dropout_seed"]))
    model = build_crnn_model(st, dev).train()
    assert model.audio_encoder.dropout_p == P_DROP
    runner = StrongRunner(model, device=str(dev))
    loss = runner.forward_backward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    lv = runner.loss_value(loss)                       # also raises if the GRU(128) exchange timed out at this grid
    info = model.audio_encoder._last_dropout
    assert info["seeds"] == [int(gold["dropout_seed"])] and info["p"] == P_DROP
    kept = int(ops.dropout_mask(info["seeds"][0], (64, 125, 1, 128), P_DROP, dev, pooled=True).sum().item())
    assert kept == int(gold["mask_keep_count"]), kept
    assert abs(lv - float(gold["loss_f64"])) < 2e-5, (lv, float(gold["loss_f64"]))
    with torch.no_grad():
        out = runner.forward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, training=True)
    assert out["frame_sim"].shape == (64, 125)
    fs_err = np.abs(out["frame_sim"].cpu().numpy().astype(np.float64) - gold["frame_sim_f64"]).max()
    print(f"crnn B=64: loss {lv:.7f} vs {float(gold['loss_f64']):.7f}; frame_sim err {fs_err:.2e} "
          f"(fp32 oracle itself {float(gold['frame_sim_floor']):.2e})")
    assert fs_err < 1e-4
    worst = 0.0
    for name, p in model.named_parameters():
        want, floor = gold[f"grad/{name}"], gold[f"floor/{name}"]
        g = p.grad.detach().double().flatten().cpu()
        gi = torch.Generator().manual_seed(sum(map(ord, name)))
        idx = torch.randint(0, g.numel(), (min(1024, g.numel()),), generator=gi)
        err = np.abs(g[idx].numpy() - want[2:]).max() / (want[1] + 1e-300)
        nerr = abs(g.norm().item() - want[0]) / (want[0] + 1e-300)
        fl = max(float(floor[0]), float(floor[2]))
        worst = max(worst, max(err, nerr) / (4.0 * max(fl, 1e-6)))
        print(f"  {name:45s} hip {err:.2e} (norm {nerr:.2e})  fp32-oracle floor {fl:.2e}")
        assert_crnn_grad_close(name, err, nerr, fl)
    print(f"crnn B=64 gradients: worst tensor at {worst:.2f} of its 4 x floor bound")


def test_full_length_train_step_vs_oracle(dev):
    """BASELINE clip length, B=6, one full training step (train-mode BN, dropout replayed): loss, frame_sim and the
    gradient of every parameter against the fp64 CPU oracle; then clip_grad_norm_ + Adam against torch.optim.Adam."""
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd.runner import StrongRunner
    st = O.init_state(seed=5, logit_gain=120.0)
    batch = O.synthetic_batch(6, 320000, seed=99, ragged=True)
    torch.manual_seed(7)
    model = build_hip_model(st, "dot", dev).train()
    runner = StrongRunner(model, lr=1e-3, max_grad_norm=1.0, device=str(dev))
    before = {k: v.detach().cpu().clone() for k, v in model.named_parameters()}
    loss = runner.forward_backward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    info = model.audio_encoder._last_dropout
    shapes = [(6, 500, 32, 64), (6, 250, 16, 128), (6, 250, 8, 256), (6, 250, 4, 512)]
    masks = {f"drop{i + 1}": ops.dropout_mask(info["seeds"][i], shp, 0.2, dev, pooled=True).cpu().permute(0, 3, 1, 2).double()
             for i, shp in enumerate(shapes)}
    masks["drop5"] = ops.dropout_mask(info["seeds"][4], (6, 250, 512), 0.5, dev).cpu().double()
    st_o = O.state_to(st, torch.float64, requires_grad=True)
    bo = dict(batch)
    bo["waveform"], bo["label"] = batch["waveform"].double(), batch["label"].double()
    oloss, oout = O.train_step_loss(st_o, bo, "dot", "cnn8rnn", True, None, masks)
    oloss.backward()
    assert abs(loss.item() - oloss.item()) < 2e-5
    # the same step by the CPU oracle in fp32: its distance from fp64 is the noise floor of ANY fp32 implementation
    # (ReLU / max-pool decisions that flip under rounding; 10 M+ decisions per layer at this size)
    st_f = O.state_to(st, torch.float32, requires_grad=True)
    m32 = {k: v.float() for k, v in masks.items()}
    floss, _ = O.train_step_loss(st_f, batch, "dot", "cnn8rnn", True, None, m32)
    floss.backward()
    names = [n for n, _ in model.named_parameters()]
    scale = {n: st_o[n].grad.abs().max().item() + 1e-30 for n in names}
    floor = [(st_f[n].grad.double() - st_o[n].grad).abs().max().item() / scale[n] for n in names]
    errs = [(p.grad.cpu().double() - st_o[n].grad).abs().max().item() / scale[n] for n, p in model.named_parameters()]
    # Above the last conv block a gradient tensor sees only a handful of near-tied ReLU / max-pool decisions, so whether ONE
    # of them flips under fp32 rounding is a coin toss (tools/diag_flip.py: waveforms perturbed by 6e-8 relative move the fp64
    # oracle's gradients by 1e-8, but toggle this path's fc1.weight error between 5e-7 and 6.9e-4 -- always the same element).
    # A per-tensor bound against ONE other fp32 run is therefore asserted on the best of four such perturbed steps (an
    # implementation error shows in all of them), and every single step must stay inside the flip budget of the conv blocks.
    best = list(errs)
    for trial in (1, 2, 3):
        g = torch.Generator().manual_seed(trial)
        bt = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        bt["waveform"] = (batch["waveform"].double() * (1 + 6e-8 * torch.randn(batch["waveform"].shape, generator=g,
                                                                              dtype=torch.float64))).float()
        torch.manual_seed(7)
        m2 = build_hip_model(st, "dot", dev).train()
        StrongRunner(m2, lr=1e-3, max_grad_norm=1.0, device=str(dev)).forward_backward(bt)
        assert m2.audio_encoder._last_dropout["seeds"] == info["seeds"]
        for i, (n, p) in enumerate(m2.named_parameters()):
            e = (p.grad.cpu().double() - st_o[n].grad).abs().max().item() / scale[n]
            assert e <= max(4.0 * max(floor), 2e-2), (n, e)
            best[i] = min(best[i], e)
    for n, e, bst, f in zip(names, errs, best, floor):
        print(f"  {n:55s} hip {e:.2e} (best of 4 perturbed steps {bst:.2e})  cpu-fp32 {f:.2e}")
    print(f"full-length train step: loss {loss.item():.7f} vs {oloss.item():.7f}; grad err median {np.median(errs):.2e} "
          f"max {max(errs):.2e}; cpu-fp32-oracle floor median {np.median(floor):.2e} max {max(floor):.2e}")
    assert float(np.median(errs)) < max(2e-5, 4 * float(np.median(floor)))
    # conv-block tensors flip in EVERY fp32 run (6 M+ decisions per layer): one CPU run that happened not to flip inside
    # block 4 (its floor there is 1e-5 .. 4e-7) is not a statistic, the median floor over the conv-block tensors is
    conv_floor = float(np.median([f for n, f in zip(names, floor) if "conv_block" in n or "bn0" in n]))
    for n, e, bst, f in zip(names, errs, best, floor):
        assert e <= max(4.0 * max(floor), 2e-2), (n, e)
        fl = max(f, conv_floor) if ("conv_block" in n or "bn0" in n) else max(f, 2e-6)
        assert bst <= 4.0 * fl, (n, bst, f)
    # optimiser at full parameter size: clip_grad_norm_(1.0) + Adam in fp64 on the SAME (HIP) gradients
    # (the first Adam step is lr * g/|g|, so it must be fed identical gradients to be comparable)
    params = [before[k].double().requires_grad_(True) for k, _ in model.named_parameters()]
    for q, (_, p) in zip(params, model.named_parameters()):
        q.grad = p.grad.detach().cpu().double()
    opt = torch.optim.Adam(params, lr=1e-3)
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    opt.step()
    runner.optimizer_step()
    for (name, p), q in zip(model.named_parameters(), params):
        assert (p.detach().cpu().double() - q.detach()).abs().max().item() < 1e-6, name


@pytest.mark.parametrize("math_", ["fp32", "fp32-direct"])
def test_benched_size_train_step_vs_fixture(dev, golden_dir, monkeypatch, math_):
    """Parity AT THE BENCHED SIZE (BASELINE configs[1]: B = 64, 10 s clips, train-mode BatchNorm, dropout ON) -- the grids
    where the XCD remap, 3-workgroup/CU residency, split-K counts and the 128-workgroup persistent GRU actually live.
    tests/golden/b64_train_step.npz holds the CPU oracle's fp64 step (truth) and its own fp32 step (the noise floor of
    any fp32 implementation) with THESE dropout masks (make_golden_b64.py; the oracle replays the HIP generator from the
    same seeds).  loss <= 2e-5, frame_sim <= 1e-4, every gradient tensor within 4 x the fp32 oracle's own distance from
    fp64 (floor clamped below by 2e-5 of the tensor's max: above the last ReLU the floor is pure round-off)."""
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd.runner import StrongRunner
    gold = np.load(f"{golden_dir}/b64_train_step.npz")
    st = O.init_state(seed=5, logit_gain=120.0)
    batch = O.synthetic_batch(64, 320000, seed=99, ragged=True)
    chk = checksum(batch["waveform"]) + checksum(batch["text"].float()) + checksum(st["audio_encoder.fc1.weight"])
    assert np.allclose(chk, gold["input_checksum"], rtol=1e-9), "seeded inputs drifted from the fixture"
    seeds = iter(int(v) for v in gold["dropout_seeds"])
    monkeypatch.setattr(_functions, "new_seed", lambda: next(seeds))
    # "x9" (round 4): the all-nine-products split arithmetic is held to the SAME bounds as the exact-fp32 MFMA kernels
    # "fp32" (the default path, round 5): blocks 3 and 4 run their forward and conv2-dgrad launches as Winograd F(2x2,3x3)
    # (csrc/conv_wino.hip) at this size; "fp32-direct" = the direct halo-tile kernels everywhere (TAG_CONV_WINOGRAD=0) -- same bounds
    monkeypatch.setattr(ops, "CONV_MATH", "fp32" if math_ == "fp32-direct" else math_)
    monkeypatch.setattr(ops, "CONV_WINOGRAD", math_ == "fp32")
    wino0 = ops.WINO_LAUNCHES
    model = build_hip_model(st, "dot", dev).train()
    runner = StrongRunner(model, device=str(dev))
    loss = runner.forward_backward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    lv = runner.loss_value(loss)                       # also raises if the GRU exchange timed out at this grid
    # the four convs of blocks 3 and 4 (128->256, 256->256, 256->512, 512->512): 4 forward launches, 2 dgrad launches carrying
    # BatchNorm-backward sums (their conv2), 2 dgrads carrying the pool-backward sums of the block below (their conv1), 4 weight gradients
    # (round 6: the fused Winograd kernels take every layer with >= 64 channels: 7 forward launches, 4 dgrads carrying BatchNorm-backward
    # sums, 3 dgrads carrying the pool-backward sums of the block below, 7 weight gradients)
    assert ops.WINO_LAUNCHES - wino0 == (21 if math_ == "fp32" else 0), ops.WINO_LAUNCHES - wino0
    info = model.audio_encoder._last_dropout
    assert info["seeds"] == [int(v) for v in gold["dropout_seeds"]]
    # the masks the kernels drew are the masks the oracle replayed (CPU restatement of the generator, checked by count)
    shapes = [(64, 500, 32, 64), (64, 250, 16, 128), (64, 250, 8, 256), (64, 250, 4, 512), (64, 250, 512)]
    for i, shp in enumerate(shapes):
        kept = int(ops.dropout_mask(info["seeds"][i], shp, 0.2 if i < 4 else 0.5, dev, pooled=i < 4).sum().item())
        assert kept == int(gold["mask_keep_counts"][i]), (i, kept)
    assert abs(lv - float(gold["loss_f64"])) < 2e-5, (lv, float(gold["loss_f64"]))
    # frame_sim of the training forward: re-run the forward with the same seeds (BatchNorm batch statistics, same masks)
    seeds2 = iter(int(v) for v in gold["dropout_seeds"])
    monkeypatch.setattr(_functions, "new_seed", lambda: next(seeds2))
    with torch.no_grad():
        out = runner.forward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, training=True)
    fs_err = np.abs(out["frame_sim"].cpu().numpy().astype(np.float64) - gold["frame_sim_f64"]).max()
    print(f"B=64: loss {lv:.7f} vs {float(gold['loss_f64']):.7f}; frame_sim err {fs_err:.2e} "
          f"(fp32 oracle itself {float(gold['frame_sim_floor']):.2e})")
    assert fs_err < 1e-4
    worst = 0.0
    for name, p in model.named_parameters():
        want, floor = gold[f"grad/{name}"], gold[f"floor/{name}"]
        g = p.grad.detach().double().flatten().cpu()
        gi = torch.Generator().manual_seed(sum(map(ord, name)))
        idx = torch.randint(0, g.numel(), (min(1024, g.numel()),), generator=gi)
        scale = want[1] + 1e-300
        err = np.abs(g[idx].numpy() - want[2:]).max() / scale
        nerr = abs(g.norm().item() - want[0]) / (want[0] + 1e-300)
        bound = 4.0 * max(float(floor[0]), 2e-5)
        worst = max(worst, err / bound)
        print(f"  {name:55s} hip {err:.2e} (norm {nerr:.2e})  fp32-oracle floor {floor[0]:.2e}  -> {err / bound:.2f} of the bound")
        assert err <= bound, (name, err, float(floor[0]))
        assert nerr <= bound, (name, nerr, float(floor[0]))       # |‖a‖ - ‖b‖| <= ‖a - b‖: the norm obeys the same budget
    print(f"B=64 gradients: worst tensor at {worst:.2f} of its 4 x floor bound")


BF16_BUDGET_REALISATIONS = 5


def test_bf16_mode_train_step_budget(dev, golden_dir, monkeypatch):
    """BASELINE configs[2] as a real mode (TAG_CONV_MATH=bf16 + TAG_ACT_DTYPE=bf16): bf16 conv arithmetic AND bf16 storage
    of the conv stack's activations / gradients, everything else fp32 -- the B = 64, 10 s training step (T' = 250, dropout
    on) against the fp64 fixture of the benched step, as a STATISTIC over rounding realisations instead of one sample.

    bf16 has 8 significand bits: per-element agreement is 1e-2-ish by construction; what training needs is an unbiased gradient
    direction, which the (norm, cosine) pair measures -- but ONE step is one realisation of the rounding noise: rounds 2-4 moved
    the conv-block norm bound 5 % -> 8 % when a new kernel changed which ties round which way (round-4 review).  Here the step is
    run BF16_BUDGET_REALISATIONS times, realisation r > 0 on the waveform perturbed by one part in 1.7e7 (6e-8 relative, a coin
    flip on the last bit of each fp32 sample: the fp64 fixture's gradients move by 1e-8 under it, tools/diag_flip.py) so that
    the rounding ties of all eight bf16 layers fall differently, and per gradient tensor the SIGNED relative norm error
    e_r = (|g_r| - |g_fp64|) / |g_fp64| is summarised by its mean (the mode's bias) and standard deviation (its noise):

      * loss within 2e-3 and frame_sim within 6e-2 of fp64 for every realisation (logit gain 120 spreads the logits over +-3.3);
      * bias:  |mean e| <= 6 % (conv blocks / bn0: eight bf16 layers and their BatchNorm cancellations deep), 3 % (the rest);
      * noise: std e <= 2.5 % (conv blocks / bn0), 1 % (the rest) -- and therefore no single realisation is asserted against a
        constant that a kernel change could move: a kernel that only re-orders ties changes e_r, not (mean, std);
      * direction, every realisation: cosine on the sampled entries >= 0.97 (conv blocks / bn0), 0.999 (fc1), 0.9999 (GRU,
        embedding: above the encoder's last ReLU); the MEAN gradient over the realisations must be at least as close in cosine as
        the worst single one (noise averages out, a bias would not).
    What pins the arithmetic itself are the per-stage rounding-point emulation tests below; this is the end-to-end check."""
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd.runner import StrongRunner
    gold = np.load(f"{golden_dir}/b64_train_step.npz")
    st = O.init_state(seed=5, logit_gain=120.0)
    batch = O.synthetic_batch(64, 320000, seed=99, ragged=True)
    monkeypatch.setattr(ops, "CONV_MATH", "bf16")
    monkeypatch.setattr(ops, "ACT_DTYPE", "bf16")
    names, sampled, norms, stats = None, {}, {}, []
    for r in range(BF16_BUDGET_REALISATIONS):
        b = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        if r:
            gen = torch.Generator().manual_seed(1000 + r)
            b["waveform"] = (b["waveform"].double() * (1 + 6e-8 * torch.randn(b["waveform"].shape, generator=gen,
                                                                               dtype=torch.float64))).float()
        seeds = iter(int(v) for v in gold["dropout_seeds"])
        monkeypatch.setattr(_functions, "new_seed", lambda: next(seeds))
        model = build_hip_model(st, "dot", dev).train()
        runner = StrongRunner(model, device=str(dev))
        torch.cuda.reset_peak_memory_stats()
        loss = runner.forward_backward(dict(b))
        lv = runner.loss_value(loss)
        peak = torch.cuda.max_memory_allocated() / 2 ** 30
        seeds2 = iter(int(v) for v in gold["dropout_seeds"])
        monkeypatch.setattr(_functions, "new_seed", lambda: next(seeds2))
        with torch.no_grad():
            out = runner.forward(dict(b), training=True)
        fs_err = np.abs(out["frame_sim"].cpu().numpy().astype(np.float64) - gold["frame_sim_f64"]).max()
        stats.append((lv, fs_err))
        assert abs(lv - float(gold["loss_f64"])) < 2e-3 and fs_err < 6e-2, (r, lv, fs_err)
        names = [n for n, _ in model.named_parameters()]
        for name, p in model.named_parameters():
            g = p.grad.detach().double().flatten().cpu()
            gi = torch.Generator().manual_seed(sum(map(ord, name)))
            idx = torch.randint(0, g.numel(), (min(1024, g.numel()),), generator=gi)
            sampled.setdefault(name, []).append(g[idx].numpy())
            norms.setdefault(name, []).append(g.norm().item())
        del runner, model
    print(f"bf16 mode B=64, {BF16_BUDGET_REALISATIONS} rounding realisations: loss {[round(s[0], 6) for s in stats]} vs "
          f"{float(gold['loss_f64']):.6f}; frame_sim err {[f'{s[1]:.1e}' for s in stats]}; peak memory {peak:.1f} GiB")
    bad = []
    for name in names:
        want = gold[f"grad/{name}"]
        e = np.array([(n - want[0]) / (want[0] + 1e-300) for n in norms[name]])
        cos = [float(np.dot(a, want[2:]) / (np.linalg.norm(a) * np.linalg.norm(want[2:]) + 1e-300)) for a in sampled[name]]
        mean_g = np.mean(sampled[name], axis=0)
        cos_mean = float(np.dot(mean_g, want[2:]) / (np.linalg.norm(mean_g) * np.linalg.norm(want[2:]) + 1e-300))
        deep = "conv_block" in name or "bn0" in name
        floor = 0.97 if deep else (0.999 if "fc1" in name else 0.9999)
        print(f"  {name:55s} norm err mean {e.mean():+.2e} std {e.std(ddof=1):.2e} [{e.min():+.2e}, {e.max():+.2e}]  "
              f"cosine min {min(cos):.6f} of-the-mean {cos_mean:.6f}")
        ok = (abs(e.mean()) <= (6e-2 if deep else 3e-2) and e.std(ddof=1) <= (2.5e-2 if deep else 1e-2)
              and min(cos) >= floor and cos_mean >= min(cos) - 1e-6)
        bad += [] if ok else [(name, float(e.mean()), float(e.std(ddof=1)), min(cos), cos_mean)]
    assert not bad, bad


def test_bf16_mode_stages_match_rounding_point_emulation(dev, monkeypatch):
    """The bf16 mode held to MORE than a budget against fp64 (round-2 review: "no check that the bf16 path equals the fp32 path
    run on bf16-rounded tensors layer by layer").  The oracle restates the mode's rounding points (O.cnn8rnn_forward_bf16_mode:
    bf16 storage of raw conv outputs and pooled activations, bf16 conv / GEMM operands, fp32 accumulation and everything else);
    here EVERY stage of the eval-mode encoder is run on the device and compared with that restatement evaluated in float64 ON
    THE DEVICE'S OWN INPUT of the stage, so that what remains is fp32 accumulation order and rounding ties: mean error <= 1e-6
    of the stage's range (measured <= 1.3e-7) and <= 5e-4 of the elements off by more than one bf16 ulp (measured <= 5e-5).  (End to end the same ties amplify
    chaotically through eight layers -- 15 % of block 4's elements differ by a bf16 ulp -- which is why the whole-path bf16
    tests are budgets; a dropped or extra rounding point shows up HERE as a whole-tensor ulp-level error.)"""
    import torch.nn.functional as F
    from texttoaudiogrounding_amd import ops
    monkeypatch.setattr(ops, "CONV_MATH", "bf16")
    monkeypatch.setattr(ops, "ACT_DTYPE", "bf16")
    st = O.init_state(seed=13, logit_gain=30.0)
    batch = O.synthetic_batch(4, 64000, seed=21, ragged=True)
    model = build_hip_model(st, "dot", dev).eval()
    mod = model.audio_encoder
    s64 = O.state_to(st, torch.float64)
    q, P = O._q_bf16, "audio_encoder."
    worst = {"mean": 0.0, "frac": 0.0}

    def check(name, dev_t, ref, nchw=True):
        d = dev_t.double().cpu()
        r = ref.permute(0, 2, 3, 1) if (nchw and ref.dim() == 4) else ref
        e = (d - r).abs()
        rng = r.abs().max().item()
        mean, frac = e.mean().item() / rng, (e > r.abs() * 2.0 ** -7 + 1e-6 * rng).double().mean().item()
        print(f"  {name:26s} mean err / range {mean:.1e}   elements off by > 1 bf16 ulp {frac:.1e}")
        worst["mean"], worst["frac"] = max(worst["mean"], mean), max(worst["frac"], frac)
        assert mean <= 1e-6 and frac <= 5e-4, (name, mean, frac)

    def nchw(t):
        return t.double().cpu().permute(0, 3, 1, 2)

    bn = lambda y, m: ops.bn_stats(y.view(-1, y.shape[-1]), m.weight.detach(), m.bias.detach(), m.running_mean, m.running_var, False,
                                   m.eps, m.momentum)
    with torch.no_grad():
        lm = ops.logmel(batch["waveform"].to(dev), mod.n_fft, mod.win_length, mod.hop_length, mod.window, mod.mel_fb)
        st0 = bn(lm, mod.bn0)
        x_dev, x_in = None, lm.double().cpu().unsqueeze(1)                      # (B,1,F,64) = the device's own log-mel
        s0, t0 = O._bn_affine(x_in.transpose(1, 3), s64, P + "bn0.", False)
        x_in = (x_in.transpose(1, 3) * s0 + t0).transpose(1, 3)
        for i, ps in enumerate([(2, 2), (2, 2), (1, 2), (1, 2)], start=1):
            blk, bp = getattr(mod, f"conv_block{i}"), f"{P}conv_block{i}."
            w1, w2 = s64[bp + "conv1.weight"], s64[bp + "conv2.weight"]
            if i == 1:
                y1, _ = ops.conv3x3_c1_stats(lm, blk.conv1.weight.detach(), st0.scale, st0.shift, want_stats=False,
                                             out_dtype=torch.bfloat16)
            else:
                wf, _ = ops.pack_conv_weight(blk.conv1.weight.detach(), want_dgrad=False, W=x_dev.shape[2])
                y1, _ = ops.conv3x3_stats(x_dev, wf, blk.conv1.weight.shape[0], want_stats=False)
            assert y1.dtype == torch.bfloat16
            check(f"block{i}.conv1", y1, q(F.conv2d(x_in, w1 if i == 1 else q(w1), None, 1, 1)))
            s1 = bn(y1, blk.bn1)
            r1s, r1t = O._bn_affine(None, s64, bp + "bn1.", False)
            wf2, _ = ops.pack_conv_weight(blk.conv2.weight.detach(), want_dgrad=False, W=y1.shape[2])
            y2, _ = ops.conv3x3_stats(y1, wf2, y1.shape[3], prologue=1, scale=s1.scale, shift=s1.shift, want_stats=False)
            check(f"block{i}.conv2 (BN+ReLU fed)", y2, q(F.conv2d(q(F.relu(nchw(y1) * r1s + r1t)), q(w2), None, 1, 1)))
            s2 = bn(y2, blk.bn2)
            r2s, r2t = O._bn_affine(None, s64, bp + "bn2.", False)
            x_dev = ops.bnact_pool(y2, s2, ps[0], ps[1], act=1, pool=0)
            a2 = F.relu(nchw(y2) * r2s + r2t)
            check(f"block{i}.pool", x_dev, q(F.avg_pool2d(a2, kernel_size=ps) + F.max_pool2d(a2, kernel_size=ps)))
            x_in = nchw(x_dev)
        B_, Tp, Wp, C = x_dev.shape
        xm = torch.empty(B_ * Tp, C, device=dev)
        ops.call("tag_mean_w_forward_bf16", ops.ptr(x_dev), B_ * Tp, Wp, C, 0.0, 0, ops.ptr(xm))
        check("mean over mel", xm.view(B_, Tp, C), x_in.mean(dim=3).transpose(1, 2), nchw=False)
        fw, fb = mod.fc1.weight.detach(), mod.fc1.bias.detach()
        fc = ops.gemm(xm, fw, B_ * Tp, fw.shape[0], C, transB=True, bias=fb, act=1)
        fc_ref = F.relu(F.linear(q(xm.double().cpu()), q(s64[P + "fc1.weight"]), s64[P + "fc1.bias"]))
        check("fc1 (bf16 operands)", fc, fc_ref, nchw=False)
        w_ih = torch.cat([mod.rnn.weight_ih_l0.detach(), mod.rnn.weight_ih_l0_reverse.detach()], 0)
        b_ih = torch.cat([mod.rnn.bias_ih_l0.detach(), mod.rnn.bias_ih_l0_reverse.detach()], 0)
        gi = ops.gemm(fc, w_ih, B_ * Tp, w_ih.shape[0], fc.shape[1], transB=True, bias=b_ih)
        gi_ref = F.linear(q(fc.double().cpu()), q(w_ih.double().cpu()), b_ih.double().cpu())
        check("GRU input projection", gi, gi_ref, nchw=False)
        # whole encoder, for the record: the ties above amplify through the layers
        inp = {k: (v.clone().to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
        inp["specaug"] = False
        emb = mod(inp)["embedding"].double().cpu()
    emu = O.cnn8rnn_forward_bf16_mode(dict(s64), batch["waveform"].double(), batch["waveform_len"], training=False)["embedding"]
    ref = O.cnn8rnn_forward(dict(O.state_to(st, torch.float64)), batch["waveform"].double(), batch["waveform_len"], training=False)["embedding"]
    m_emu, m_ref = (emb - emu).abs().mean().item(), (emb - ref).abs().mean().item()
    print(f"bf16 mode, per stage: worst mean err / range {worst['mean']:.1e}, worst > 1 ulp fraction {worst['frac']:.1e}; whole encoder: "
          f"embedding mean |err| {m_emu:.1e} vs the emulation, {m_ref:.1e} vs the plain fp64 oracle")
    assert m_emu < m_ref and m_emu < 2e-3


def test_bf16_mode_backward_stages_match_rounding_point_emulation(dev, monkeypatch):
    """The backward twin of the test above (round-3 review: "the bf16 BACKWARD -- dgrad / wgrad / bf16 gradient storage -- is
    only budget-checked").  One training-mode forward of the encoder on the device (bf16 mode, dropout off) leaves the saved
    activations of every block; then the backward of the conv stack is driven STAGE BY STAGE with the engine's own calls, from a
    seeded gradient at the top, and every stage is compared with the oracle's restatement of its rounding points
    (O.bf16_bnrelu_pool_backward / bf16_conv_dgrad / bf16_dgrad_bnrelu_backward / bf16_conv_wgrad) evaluated in float64 ON THE
    DEVICE'S OWN INPUTS of that stage.  bf16-stored results: mean error <= 1e-6 of the range and <= 5e-4 of the elements off
    by more than one bf16 ulp; fp32 results (weight gradients, dgamma / dbeta): <= 2e-5 of their maximum.  Blocks 4..2 in full,
    block 1 down to dL/dy1 (its Cin = 1 backward has its own kernel tests)."""
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd import torch_ops as T
    monkeypatch.setattr(ops, "CONV_MATH", "bf16")
    monkeypatch.setattr(ops, "ACT_DTYPE", "bf16")
    st = O.init_state(seed=17, logit_gain=30.0)
    batch = O.synthetic_batch(4, 64000, seed=23, ragged=False)
    model = build_hip_model(st, "dot", dev).train()
    mod = model.audio_encoder
    mod.dropout_p = (0.0, 0.0)
    ctx = T._EncCtx((False, False) + (True,) * len(mod._flat_params()))
    ops.Cnn8RnnFunction.forward(ctx, batch["waveform"].to(dev), mod, *mod._flat_params())
    sv = ctx.saved
    p = sv["p"]
    worst = {"mean": 0.0, "frac": 0.0, "f32": 0.0}
    nchw = lambda t: t.double().cpu().permute(0, 3, 1, 2).contiguous()
    vec = lambda t: t.double().cpu()

    def check16(name, dev_t, ref):
        d, r = nchw(dev_t), ref
        e = (d - r).abs()
        rng = r.abs().max().item()
        mean, frac = e.mean().item() / rng, (e > r.abs() * 2.0 ** -7 + 1e-6 * rng).double().mean().item()
        print(f"  {name:34s} mean err / range {mean:.1e}   elements off by > 1 bf16 ulp {frac:.1e}")
        worst["mean"], worst["frac"] = max(worst["mean"], mean), max(worst["frac"], frac)
        assert mean <= 1e-6 and frac <= 5e-4, (name, mean, frac)

    def check32(name, dev_t, ref):
        e = (dev_t.double().cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-300)
        print(f"  {name:34s} max err / max {e:.1e}")
        worst["f32"] = max(worst["f32"], e)
        assert e <= 2e-5, (name, e)

    g = torch.Generator().manual_seed(5)
    dx = (0.05 * torch.randn(sv["x_last"].shape, generator=g)).to(dev).bfloat16()
    for i in range(3, -1, -1):
        x_in, y1, s1, y2, s2, wd1, wd2 = sv["acts"][i]
        c1w, g1, b1, c2w, g2, b2 = p[2 + 6 * i: 8 + 6 * i]
        ph, pw = ops.CNN8_POOLS[i]
        # (1) pool + ReLU + BatchNorm backward of the block's second half
        dy2, dg2, db2 = ops.bnrelu_pool_backward(y2, s2, g2, dx, ph, pw, 0.0, 0)
        r_dy2, r_dg2, r_db2 = O.bf16_bnrelu_pool_backward(nchw(y2), vec(s2.scale), vec(s2.shift), vec(s2.mean), vec(s2.invstd),
                                                          vec(g2), nchw(dx), (ph, pw))
        check16(f"block{i + 1}: pool/ReLU/bn2 backward", dy2, r_dy2)
        check32(f"block{i + 1}: dgamma2", dg2, r_dg2)
        check32(f"block{i + 1}: dbeta2", db2, r_db2)
        # (2) weight gradient of conv2 (operand = relu(bn1(y1)) rounded to bf16, as in the forward)
        dw2 = ops.conv3x3_wgrad(y1, dy2, prologue=1, scale=s1.scale, shift=s1.shift)
        check32(f"block{i + 1}: conv2 weight gradient", dw2,
                O.bf16_conv_wgrad(nchw(y1), nchw(dy2), tuple(c2w.shape), vec(s1.scale), vec(s1.shift)))
        # (3) dgrad of conv2 + BatchNorm-backward sums in its epilogue + the apply pass
        dy1, dg1, db1 = ops.conv3x3_dgrad_bnrelu_backward(dy2, wd2, y1, s1, g1)
        da_f = O.bf16_conv_dgrad(nchw(dy2), vec(c2w), tuple(nchw(y1).shape))
        r_dy1, r_dg1, r_db1, _ = O.bf16_dgrad_bnrelu_backward(da_f, nchw(y1), vec(s1.scale), vec(s1.shift), vec(s1.mean),
                                                              vec(s1.invstd), vec(g1))
        check16(f"block{i + 1}: conv2 dgrad + bn1 backward", dy1, r_dy1)
        check32(f"block{i + 1}: dgamma1", dg1, r_dg1)
        check32(f"block{i + 1}: dbeta1", db1, r_db1)
        if i == 0:
            break
        # (4) conv1: weight gradient (operand = the stored pooled activation) and input gradient (stored bf16)
        dw1 = ops.conv3x3_wgrad(x_in, dy1)
        check32(f"block{i + 1}: conv1 weight gradient", dw1, O.bf16_conv_wgrad(nchw(x_in), nchw(dy1), tuple(c1w.shape)))
        dx = ops.conv3x3(dy1, wd1, x_in.shape[3])
        check16(f"block{i + 1}: conv1 dgrad", dx, O._q_bf16(O.bf16_conv_dgrad(nchw(dy1), vec(c1w), tuple(nchw(x_in).shape))))
    print(f"bf16 mode backward, per stage: worst mean err / range {worst['mean']:.1e}, worst > 1 ulp fraction {worst['frac']:.1e}, "
          f"worst fp32 result {worst['f32']:.1e}")


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_batch_of_300_clips_in_one_pass(dev, monkeypatch, mode):
    """Round-3 review, missing #4: `ops.py` raised at B >= 262 x 10 s because the conv kernels indexed activations with 32-bit
    byte offsets over the whole batch; the reference takes any batch that fits memory (run_strong.py:123-152).  The kernels now
    add a 64-bit per-image base to 32-bit in-image offsets.  B = 300 x 10 s (first conv output 4.9 GB in fp32: beyond 2^32 bytes) in
    ONE eval pass: rows 0..63 / 236..299 equal the same clips run as passes of 64 to 2e-6 (every stage up to the GRU is
    batch-invariant bit for bit; 300 sequences take the per-step GRU kernels instead of the persistent 4-row ones, a different
    summation order), and one TRAINING step at B = 300 runs (loss finite, every gradient finite and non-zero)."""
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd.runner import StrongRunner
    if mode == "bf16":
        monkeypatch.setattr(ops, "CONV_MATH", "bf16")
        monkeypatch.setattr(ops, "ACT_DTYPE", "bf16")
    B, S = 300, 320000
    st = O.init_state(seed=9, logit_gain=40.0)
    g = torch.Generator().manual_seed(77)
    wave = (0.1 * torch.randn(B, S, generator=g)).to(dev)
    text = torch.randint(2, 5221, (B, 3), generator=g)
    batch = {"waveform": wave, "waveform_len": np.full((B,), S), "text": text.to(dev), "text_len": torch.full((B,), 3).to(dev),
             "specaug": False}
    model = build_hip_model(st, "dot", dev).eval()
    with torch.no_grad():
        full = model(dict(batch))["frame_sim"]
        for lo in (0, 236):
            part = model({k: (v[lo:lo + 64] if hasattr(v, "__len__") and not isinstance(v, bool) else v) for k, v in batch.items()})
            d = (full[lo:lo + 64] - part["frame_sim"]).abs().max().item()
            print(f"B = 300 in one pass vs a pass of 64 (clips {lo}..{lo + 63}): max |d frame_sim| = {d:.1e}")
            assert d <= 2e-6, (lo, d)
    assert torch.isfinite(full).all()
    if mode == "fp32":
        return                                                  # the fp32 training step at B = 300 needs ~25 GB more; bf16 covers it
    model.train()
    runner = StrongRunner(model, device=str(dev))
    tb = dict(batch)
    tb["label"] = (torch.rand(B, 250, generator=g) < 0.5).float().to(dev)
    loss = runner.forward_backward(tb)
    assert np.isfinite(runner.loss_value(loss))
    for n, p in model.named_parameters():
        assert torch.isfinite(p.grad).all() and p.grad.abs().max() > 0, n


@pytest.mark.parametrize("B,S", [(1, 4000), (3, 9999), (2, 32000)])
def test_edge_shapes_train_step(dev, B, S):
    """Edge cases of the path: a single clip, clips of a few frames (T' = 3), odd sample counts, one-token phrases:
    one training step (dropout off) against the fp64 oracle."""
    from texttoaudiogrounding_amd.runner import StrongRunner
    st = O.init_state(seed=B + S, logit_gain=40.0)
    batch = O.synthetic_batch(B, S, seed=S, ragged=(B > 1))
    model = build_hip_model(st, "dot", dev).train()
    model.audio_encoder.dropout_p = (0.0, 0.0)
    runner = StrongRunner(model, device=str(dev))
    loss = runner.forward_backward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    s64 = O.state_to(st, torch.float64, requires_grad=True)
    b64 = dict(batch)
    b64["waveform"], b64["label"] = batch["waveform"].double(), batch["label"].double()
    oloss, oout = O.train_step_loss(s64, b64, "dot", "cnn8rnn", True, (0.0, 0.0))
    oloss.backward()
    print(f"edge B={B} S={S}: T'={oout['frame_sim'].shape[1]}, loss {loss.item():.6f} vs {oloss.item():.6f}")
    assert abs(loss.item() - oloss.item()) < 5e-5 * max(1.0, abs(oloss.item()))
    for name in ("text_encoder.embedding.core.weight", "audio_encoder.rnn.weight_hh_l0", "audio_encoder.fc1.bias"):
        p = dict(model.named_parameters())[name]
        g = s64[name].grad
        assert (p.grad.cpu().double() - g).abs().max().item() / (g.abs().max().item() + 1e-30) < 1e-3, name


_TRAJECTORY_REF = {}


def _oracle_trajectory(st, batch, names):
    """The fp64 oracle's ten optimiser steps on the trajectory test's fixed batch: the same for every arithmetic mode of the HIP
    path, so it is computed once per session (20 s of CPU work on a shared host) and shared by the four parametrisations."""
    key = tuple(names)
    if key not in _TRAJECTORY_REF:
        s64 = O.state_to(st, torch.float64, requires_grad=True)
        b64 = dict(batch)
        b64["waveform"], b64["label"] = batch["waveform"].double(), batch["label"].double()
        params = [s64[n] for n in names]
        opt = torch.optim.Adam(params, lr=2e-4)
        ref = []
        for _ in range(10):
            opt.zero_grad()
            loss, _ = O.train_step_loss(s64, b64, "dot", "cnn8rnn", True, (0.0, 0.0))
            loss.backward()
            torch.nn.utils.clip_grad_norm_(params, 1.0)
            opt.step()
            ref.append(loss.item())
        _TRAJECTORY_REF[key] = ref
    return list(_TRAJECTORY_REF[key])


@pytest.mark.parametrize("math_", ["fp32", "bf16mode"])
def test_training_trajectory_matches_oracle(dev, math_):
    """Ten optimiser steps (forward, backward, clip_grad_norm_(1.0), Adam) on a fixed batch, dropout off: the loss
    trajectory of the HIP path follows the fp64 oracle's (torch.optim.Adam on the oracle's parameters).  "bf16mode" =
    BASELINE configs[2] as a mode (bf16 conv arithmetic + bf16 activation storage + bf16 GEMM operands): the same descent
    within a bf16-sized budget."""
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd.runner import StrongRunner
    st = O.init_state(seed=23, logit_gain=20.0)
    batch = O.synthetic_batch(4, 32000, seed=9, ragged=True)
    old = (ops.CONV_MATH, ops.ACT_DTYPE)
    ops.CONV_MATH, ops.ACT_DTYPE = ("bf16", "bf16") if math_ == "bf16mode" else (math_, "fp32")
    try:
        model = build_hip_model(st, "dot", dev).train()
        model.audio_encoder.dropout_p = (0.0, 0.0)
        runner = StrongRunner(model, lr=2e-4, max_grad_norm=1.0, device=str(dev))
        hip = []
        for _ in range(10):
            hip.append(runner.train_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}).item())
    finally:
        ops.CONV_MATH, ops.ACT_DTYPE = old
    ref = _oracle_trajectory(st, batch, [n for n, _ in model.named_parameters()])
    print(f"trajectory [{math_}]: hip {['%.5f' % v for v in hip]}")
    print(f"                 oracle {['%.5f' % v for v in ref]}")
    assert ref[-1] < ref[0] - 0.005                                  # the oracle actually learns on this batch
    if math_ == "bf16mode":                                          # 8 significand bits through 8 conv layers + the GEMMs
        assert max(abs(a - b) for a, b in zip(hip, ref)) < 2e-2 and abs(hip[0] - ref[0]) < 5e-3
        assert hip[-1] < hip[0] - 0.005                              # and it learns
    else:
        assert max(abs(a - b) for a, b in zip(hip, ref)) < 3e-3       # Adam's sign-like first steps amplify fp32 noise
        assert abs(hip[0] - ref[0]) < 2e-5


@pytest.mark.parametrize("math_", ["fp32"])
def test_training_step_is_bitwise_deterministic(dev, math_):
    """Every reduction of the path folds its partials in a fixed order and nothing uses atomics (the embedding-table
    gradient is summed per row by its first occurrence, in pair order; the batch repeats token ids on purpose): two runs
    of the same step from the same state give bit-identical losses and gradients -- with the weight-gradient convs on
    the side stream."""
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd.runner import StrongRunner
    st = O.init_state(seed=3, logit_gain=40.0)
    batch = O.synthetic_batch(6, 64000, seed=8, ragged=True)
    batch["text"][3:] = batch["text"][:3]                # repeated ids: several (clip, position) pairs share a table row
    batch["text_len"][3:] = batch["text_len"][:3]
    old = ops.CONV_MATH
    ops.CONV_MATH = math_
    try:
        res = []
        for _ in range(2):
            torch.manual_seed(123)                       # same dropout seeds
            model = build_hip_model(st, "dot", dev).train()
            runner = StrongRunner(model, device=str(dev))
            loss = runner.forward_backward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
            res.append((loss.item(), runner.flat.grad.clone()))
    finally:
        ops.CONV_MATH = old
    assert res[0][0] == res[1][0]
    same = res[0][1] == res[1][1]
    assert bool(same.all()), f"{int((~same).sum())} gradient elements differ between two identical runs"


@pytest.mark.parametrize("cu_skip", [0, 2])
def test_fp32_step_with_the_wgrad_side_stream_switched_on(dev, monkeypatch, cu_skip):
    """The switches that are off by default in fp32 (round-4 advice: untested): TAG_WGRAD_STREAM=1 runs the weight-gradient convs
    on the side stream, TAG_WGRAD_CU_SKIP=k makes that stream a CU-masked hipExtStreamCreateWithCUMask stream.  Stream placement
    changes WHEN a kernel runs, never what it computes: loss and every gradient are bit-identical to the single-stream step."""
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd.runner import StrongRunner
    st = O.init_state(seed=3, logit_gain=40.0)
    batch = O.synthetic_batch(4, 64000, seed=8, ragged=True)
    res = []
    for side in (False, True):
        monkeypatch.setattr(ops, "WGRAD_SIDE_STREAM", side)
        monkeypatch.setattr(ops, "WGRAD_CU_SKIP", cu_skip if side else 0)
        monkeypatch.setattr(_engine, "_side_streams", {})             # the masked stream is created on first use
        torch.manual_seed(123)
        model = build_hip_model(st, "dot", dev).train()
        runner = StrongRunner(model, device=str(dev))
        assert ops.side_stream_enabled() == side
        loss = runner.forward_backward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
        torch.cuda.synchronize()
        res.append((loss.item(), runner.flat.grad.clone()))
        if side:
            assert len(ops.side_streams(dev)) == 1
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])


def test_side_stream_steps_without_a_host_sync_do_not_grow_the_allocator(dev, monkeypatch):
    """K training steps enqueued WITHOUT a host synchronisation (what bench.py times, what a loop that reads the loss every N steps
    does) with the weight-gradient convs on the side stream: the caching allocator's reserve must stay at the level of one step.
    Round 4 marked the side stream's operands with Tensor.record_stream, which defers a block's reuse until the HOST sees the side
    stream's event complete -- a host running ahead of the GPU saw none complete and took new memory every step (53-75 GB after
    30-60 steps of the benched size, with single hipMalloc calls blocking for 0.7-2.6 s in every second process of a row).  The operands are now kept alive until
    join(); the old rule (TAG_SIDE_RECORD_STREAM=1) stays switchable and must give bit-identical gradients."""
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd.runner import StrongRunner
    st = O.init_state(seed=3, logit_gain=40.0)
    batch = O.synthetic_batch(16, 320000, seed=8, ragged=True)    # large enough that the GPU step outlasts the host's enqueue
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}     # resident, as in bench.py: no blocking copies
    monkeypatch.setattr(ops, "WGRAD_SIDE_STREAM", True)
    grads = {}
    for record_stream in (False, True):
        monkeypatch.setattr(ops, "SIDE_RECORD_STREAM", record_stream)
        torch.manual_seed(123)
        model = build_hip_model(st, "dot", dev).train()
        runner = StrongRunner(model, device=str(dev))
        fresh = lambda: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        runner.forward_backward(fresh())
        torch.cuda.synchronize()
        grads[record_stream] = runner.flat.grad.clone()
        if record_stream:
            continue
        for _ in range(2):                                    # the allocator's steady state of a synchronised loop
            runner.train_step(fresh())
            torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(dev)
        base = torch.cuda.memory_reserved(dev)
        for _ in range(16):
            runner.train_step(fresh())                        # no synchronisation: the host runs ahead
        torch.cuda.synchronize()
        grown = torch.cuda.max_memory_reserved(dev) - base
        one_step = torch.cuda.max_memory_allocated(dev)
        assert grown <= 0.5 * one_step, (f"allocator reserve grew by {grown / 2 ** 20:.0f} MiB over 16 unsynchronised steps "
                                         f"(one step's peak: {one_step / 2 ** 20:.0f} MiB)")
        del runner, model
    assert torch.equal(grads[False], grads[True])


def test_bf16_mode_pool_sum_fusion_switch(dev, monkeypatch):
    """TAG_FUSE_POOL_BWD_BF16=1 (off by default: measured level in step time, docs/experiments_r05.md) through a whole bf16-mode
    training step: the forward is untouched (loss identical) and every gradient tensor agrees with the two-pass pool backward up to
    the bf16 rounding ties the 1e-7 differences of the folded sums can flip (cosine >= 0.99999, norms within 1e-3)."""
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd.runner import StrongRunner
    st = O.init_state(seed=3, logit_gain=40.0)
    batch = O.synthetic_batch(4, 160000, seed=8, ragged=True)
    monkeypatch.setattr(ops, "CONV_MATH", "bf16")
    monkeypatch.setattr(ops, "ACT_DTYPE", "bf16")
    res, fused_calls = [], []
    orig = ops.conv3x3_dgrad_poolsums
    monkeypatch.setattr(_functions, "conv3x3_dgrad_poolsums", lambda *a, **k: (fused_calls.append(a[0].dtype), orig(*a, **k))[1])
    for on in (False, True):
        monkeypatch.setattr(ops, "FUSE_POOL_BWD_SUMS_BF16", on)
        torch.manual_seed(123)
        model = build_hip_model(st, "dot", dev).train()
        runner = StrongRunner(model, device=str(dev))
        loss = runner.forward_backward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
        res.append((runner.loss_value(loss), {n: p.grad.detach().double().cpu().flatten().clone() for n, p in model.named_parameters()}))
        assert (len(fused_calls) > 0) == on
    assert fused_calls and all(d == torch.bfloat16 for d in fused_calls)
    assert res[0][0] == res[1][0]
    for n in res[0][1]:
        a, b = res[0][1][n], res[1][1][n]
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300))
        assert cos >= 0.99999 and abs(a.norm() - b.norm()) <= 1e-3 * a.norm(), (n, cos, a.norm().item(), b.norm().item())


def test_env_switches_are_parsed_defensively(monkeypatch):
    from texttoaudiogrounding_amd import ops
    monkeypatch.setenv("TAG_WGRAD_CU_SKIP", "two")
    with pytest.raises(RuntimeError, match="TAG_WGRAD_CU_SKIP"):
        ops._env_int("TAG_WGRAD_CU_SKIP", 0)
    monkeypatch.setenv("TAG_WGRAD_CU_SKIP", " ")
    assert ops._env_int("TAG_WGRAD_CU_SKIP", 0) == 0


def test_device_segments_to_th_auc_and_psds(dev):
    """Scores -> device segments (tag_segments, 50 thresholds) -> operating-point tables -> threshold-AUC / PSDS: identical
    to the same evaluation over the CPU oracle's segments (the segments are bit-exact, so the metrics are equal)."""
    from texttoaudiogrounding_amd.utils import eval_util, grounding_eval as GE
    g = torch.Generator().manual_seed(17)
    B, T, res = 12, 250, 0.04
    base = torch.rand(B, 1, generator=g) * 0.5 + 0.2
    fs = (base + 0.35 * torch.sin(torch.arange(T)[None, :] / (3.0 + 5.0 * torch.rand(B, 1, generator=g))) +
          0.05 * torch.randn(B, T, generator=g)).clamp(1e-7, 1.0)
    th = eval_util.eval_thresholds(50)
    names = [f"clip{i}_0" for i in range(B)]
    gt = {n: np.array([[1.0 + 0.3 * i, 3.5 + 0.3 * i], [6.0, 7.0 + 0.1 * i]]) for i, n in enumerate(names)}
    dur = {n: 10.0 for n in names}
    seg_dev = eval_util.segments_for_thresholds(fs.to(dev), th, 1, eval_util.n_connect_for(res))
    seg_cpu = [[O.segments(fs[b].numpy(), t, 1, 13) for t in th] for b in range(B)]
    t_dev = GE.tables_from_segments(seg_dev, names, th, res)
    t_cpu = GE.tables_from_segments(seg_cpu, names, th, res)
    a_dev, a_cpu = GE.compute_th_auc(t_dev, gt), GE.compute_th_auc(t_cpu, gt)
    p_dev, p_cpu = GE.psds_intersection(t_dev, gt, dur, max_efpr=800.0), GE.psds_intersection(t_cpu, gt, dur, max_efpr=800.0)
    print(f"th_auc {a_dev:.6f} (cpu {a_cpu:.6f}); psds {p_dev:.6f} (cpu {p_cpu:.6f})")
    assert a_dev == a_cpu and p_dev == p_cpu and 0.0 < a_dev < 1.0 and 0.0 < p_dev <= 1.0
