"""Per-kernel parity: every C-ABI entry point of libtag_hip.so against the oracle / a plain torch
fp32-or-fp64 CPU restatement of the same op, on seeded inputs small enough for the CPU to finish in
seconds.  Tolerances are stated per test; integer outputs must be bit-exact."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tag_oracle as O

pytestmark = pytest.mark.gpu


def nhwc(x):   # (B,C,H,W) -> (B,H,W,C) contiguous
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


@pytest.fixture(scope="module")
def ops(dev):
    from texttoaudiogrounding_amd import ops as _ops
    from texttoaudiogrounding_amd import torch_ops  # noqa: F401  (registers torch.ops.tag.*, which the head tests call)
    return _ops


@pytest.fixture(autouse=True)
def _direct_kernels_only(ops, monkeypatch):
    """This file tests the DIRECT convolution kernels (csrc/conv.hip & co.) through the ops-level entry points; the Winograd
    kernels that take the same launches above ops.WINO_MIN_WORK have their own file (tests/test_gpu_wino.py)."""
    monkeypatch.setattr(ops, "CONV_WINOGRAD", False)


# ------------------------------------------------------------------------------------------- frontend
@pytest.mark.parametrize("kind", ["cnn8rnn", "crnn"])
def test_logmel_vs_oracle(ops, dev, kind):
    g = torch.Generator().manual_seed(1)
    wave = 0.1 * torch.randn(3, 20000, generator=g)
    wave[1, 15000:] = 0.0                                   # zero-padded tail -> -100 dB floor
    wave[2] = 0.5 * torch.sin(2 * math.pi * 1000.0 * torch.arange(20000) / 32000.0)
    p = O.FRONTEND[kind]
    window, fb = O.frontend_tables(kind)
    db, power = ops.logmel(wave.to(dev), p["n_fft"], p["win_length"], p["hop_length"], window.to(dev), fb.to(dev),
                           want_power=True)
    ref_power = O.mel_spectrogram(wave.double(), kind).transpose(1, 2)        # fp64 oracle, (B,F,mel)
    ref32 = O.mel_spectrogram(wave, kind).transpose(1, 2)
    scale = ref_power.abs().amax(dim=(1, 2), keepdim=True)
    err = ((power.cpu().double() - ref_power).abs() / scale).max().item()
    err32 = ((ref32.double() - ref_power).abs() / scale).max().item()
    print(f"logmel[{kind}] power err vs fp64 oracle {err:.2e} (fp32 oracle itself {err32:.2e})")
    assert err < 5e-6                                       # power-domain tolerance, relative to clip max
    ref_db = O.amplitude_to_db(ref_power)
    big = ref_power > 1e-4 * scale                          # away from the rounding-sensitive floor
    assert (db.cpu().double() - ref_db)[big].abs().max().item() < 1e-3       # dB
    assert torch.all(db[1, -5:].cpu() == -100.0)            # silent frames hit the clamp exactly


@pytest.mark.parametrize("kind", ["cnn8rnn", "crnn"])
def test_logmel_vs_independent_witness(ops, dev, golden_dir, kind):
    """logmel.hip against the fixture a SECOND implementation produced (transformers.audio_utils, fp64, its own window
    and filterbank; tests/golden/make_golden.py:case_frontend_witness) -- not against the oracle."""
    gold = np.load(f"{golden_dir}/frontend_witness.npz")
    g = torch.Generator().manual_seed(11)
    noise = 0.1 * torch.randn(2, 32000, generator=g)
    p = O.FRONTEND[kind]
    window, fb = O.frontend_tables(kind)          # the buffers the mirror modules hold (torchaudio's fp32 tables)
    db, power = ops.logmel(noise.to(dev), p["n_fft"], p["win_length"], p["hop_length"], window.to(dev), fb.to(dev),
                           want_power=True)
    P = torch.from_numpy(gold[f"power_{kind}"]).transpose(1, 2)               # (B,F,mel) fp64
    scale = P.abs().amax(dim=(1, 2), keepdim=True)
    err = ((power.cpu().double() - P).abs() / scale).max().item()
    ddb = (db.cpu().double() - torch.from_numpy(gold[f"db_{kind}"]).transpose(1, 2)).abs().max().item()
    print(f"logmel[{kind}] vs independent witness: power {err:.2e} of clip max, dB {ddb:.2e}")
    assert err < 1e-5 and ddb < 1e-3


def test_logmel_known_answers(ops, dev, golden_dir):
    gold = np.load(f"{golden_dir}/frontend.npz")
    n = torch.arange(32000, dtype=torch.float32)
    x = (0.5 * torch.sin(2 * math.pi * 1000.0 * n / 32000.0)).unsqueeze(0)
    for kind in ("cnn8rnn", "crnn"):
        p = O.FRONTEND[kind]
        window, fb = O.frontend_tables(kind)
        db = ops.logmel(x.to(dev), p["n_fft"], p["win_length"], p["hop_length"], window.to(dev), fb.to(dev))
        ref = torch.from_numpy(gold[f"sine_{kind}"]).transpose(1, 2)          # (1,F,64)
        # bins within 60 dB of the clip's peak: further down a single-precision FFT of any factorisation sits on its own
        # rounding floor (the fp32 CPU oracle itself is 1e-3 dB off its fp64 twin at 80 dB below the peak)
        strong = ref > ref.max() - 60.0
        assert (db.cpu() - ref)[strong].abs().max().item() < 5e-4
    # SURVEY appendix A known answers (Cnn8Rnn set): frame 50 peak bin 17 = 23.6652 dB
    p = O.FRONTEND["cnn8rnn"]
    window, fb = O.frontend_tables("cnn8rnn")
    db = ops.logmel(x.to(dev), p["n_fft"], p["win_length"], p["hop_length"], window.to(dev), fb.to(dev)).cpu()
    assert int(db[0, 50].argmax()) == 17 and abs(db[0, 50, 17].item() - 23.6652) < 2e-3


# ------------------------------------------------------------------------------------------- batch norm
@pytest.mark.parametrize("C,rows", [(64, 3003), (128, 1000), (512, 777)])
def test_bn_stats(ops, dev, C, rows):
    g = torch.Generator().manual_seed(C)
    x = torch.randn(rows, C, generator=g) * 3 - 20.0 * (C == 64)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    rm_d, rv_d = rm.clone().to(dev), rv.clone().to(dev)
    st = ops.bn_stats(x.to(dev), gamma.to(dev), beta.to(dev), rm_d, rv_d, True)
    xd = x.double()
    mean, var = xd.mean(0), xd.var(0, unbiased=False)
    assert relerr(st.mean, mean) < 1e-6
    assert relerr(st.invstd, 1 / torch.sqrt(var + 1e-5)) < 1e-6
    assert relerr(st.scale, gamma.double() / torch.sqrt(var + 1e-5)) < 1e-6
    assert relerr(rm_d, 0.9 * rm.double() + 0.1 * mean) < 1e-6
    assert relerr(rv_d, 0.9 * rv.double() + 0.1 * xd.var(0, unbiased=True)) < 1e-6
    st_e = ops.bn_stats(x.to(dev), gamma.to(dev), beta.to(dev), rm.to(dev), rv.to(dev), False)
    assert relerr(st_e.scale, gamma / torch.sqrt(rv + 1e-5)) < 1e-6
    assert relerr(st_e.shift, beta - rm * gamma / torch.sqrt(rv + 1e-5)) < 1e-5


# ------------------------------------------------------------------------------------------- convolutions
@pytest.mark.parametrize("B,H,W,Cin,Cout,pro", [(2, 9, 8, 32, 64, 0), (1, 17, 16, 64, 128, 1), (2, 5, 4, 128, 256, 1),
                                                (1, 7, 8, 64, 64, 3), (1, 6, 8, 32, 32, 2), (3, 11, 8, 256, 512, 0),
                                                # the BASELINE (10 s clip) layer shapes of Cnn8Rnn, B=2
                                                (2, 1001, 64, 64, 64, 1), (2, 500, 32, 64, 128, 0),
                                                (2, 500, 32, 128, 128, 1), (2, 250, 16, 128, 256, 0),
                                                (2, 250, 16, 256, 256, 1), (2, 250, 8, 256, 512, 0),
                                                (3, 250, 8, 512, 512, 1),
                                                # round 4: 64-wide images run as two 4 x 32 tile columns, their weight gradient (and
                                                # the 32-wide one) walks down a strip with the input rows in a ring of four slots:
                                                # images shorter than the ring / the tile, a split that crosses strips and images
                                                (1, 1, 64, 64, 64, 0), (3, 2, 64, 64, 128, 1), (2, 7, 64, 128, 64, 1),
                                                (1, 1, 32, 64, 64, 1), (5, 3, 32, 64, 128, 0), (2, 5, 32, 128, 64, 3),
                                                # round 5: 4-wide images (the last two CrnnEncoder layers, 125 x 4) on the halo-tile
                                                # kernel as 32 x 4 tiles: a tile taller than the image, a partial last tile, prologues 2 / 3
                                                (2, 125, 4, 128, 128, 2), (3, 125, 4, 128, 128, 3), (1, 33, 4, 64, 128, 0), (2, 31, 4, 128, 64, 1)])
def test_conv3x3_forward_dgrad_wgrad(ops, dev, B, H, W, Cin, Cout, pro):
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    s, t = torch.rand(Cin, generator=g) + 0.5, 0.3 * torch.randn(Cin, generator=g)

    def pro_fn(v):
        sc, sh = s.view(1, -1, 1, 1).double(), t.view(1, -1, 1, 1).double()
        if pro == 1:
            return F.relu(v * sc + sh)
        if pro == 2:
            return F.leaky_relu(v, 0.1) * sc + sh
        if pro == 3:
            return v * sc + sh
        return v

    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    a = pro_fn(xd)
    a.retain_grad()
    y_ref = F.conv2d(a, wd, None, 1, 1)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy.double())

    wf, wdg = ops.pack_conv_weight(w.to(dev))
    sd, td = (s.to(dev), t.to(dev)) if pro else (None, None)
    y = ops.conv3x3(nhwc(x).to(dev), wf, Cout, pro, sd, td)
    tol = 5e-6     # fp32 MFMA = sequential fmaf chain over K = 9*Cin <= 4608 terms (vs an fp64 reference)
    assert relerr(nchw(y), y_ref) < tol
    da = ops.conv3x3(nhwc(dy).to(dev), wdg, Cin)                                 # dgrad wrt prologue output
    assert relerr(nchw(da), a.grad) < tol
    dw = ops.conv3x3_wgrad(nhwc(x).to(dev), nhwc(dy).to(dev), pro, sd, td)
    assert relerr(dw, wd.grad) < tol


@pytest.mark.parametrize("B,H,W,Cin,Cout,pro", [(2, 9, 8, 32, 64, 0), (1, 17, 16, 64, 128, 1), (2, 21, 32, 64, 64, 1),
                                                (1, 7, 64, 64, 64, 3), (1, 6, 8, 32, 128, 2), (3, 11, 8, 256, 512, 0),
                                                (2, 1001, 64, 64, 64, 1), (2, 500, 32, 64, 128, 0),
                                                (2, 500, 32, 128, 128, 1), (2, 250, 16, 128, 256, 0),
                                                (2, 250, 16, 256, 256, 1), (2, 250, 8, 256, 512, 0),
                                                (3, 250, 8, 512, 512, 1)])
@pytest.mark.parametrize("math_", ["x3", "x9"])
def test_conv3x3_x3_forward_dgrad(ops, dev, B, H, W, Cin, Cout, pro, math_):
    """Opt-in arithmetic (conv_x3.hip): fp32 operands split exactly into 3 bf16 terms, 6 ("x3") or all 9 ("x9") partial
    products on the bf16 MFMA, fp32 accumulate -- held to the SAME tolerance against fp64 as the exact-fp32 kernels."""
    g = torch.Generator().manual_seed(B * 1000 + H + W)
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(B, Cin, 1, 1, generator=g))   # wide dynamic range
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    s, t = torch.rand(Cin, generator=g) + 0.5, 0.3 * torch.randn(Cin, generator=g)
    sc, sh = s.view(1, -1, 1, 1).double(), t.view(1, -1, 1, 1).double()
    xd = x.double()
    a = {0: xd, 1: F.relu(xd * sc + sh), 2: F.leaky_relu(xd, 0.1) * sc + sh, 3: xd * sc + sh}[pro]
    y_ref = F.conv2d(a, w.double(), None, 1, 1)
    dy = torch.randn(y_ref.shape, generator=g)
    da_ref = torch.nn.grad.conv2d_input(a.shape, w.double(), dy.double(), 1, 1)
    old = ops.CONV_MATH
    ops.CONV_MATH = math_
    try:
        wf, wdg = ops.pack_conv_weight(w.to(dev), W=W)
        assert wf.dtype == torch.uint8
        sd, td = (s.to(dev), t.to(dev)) if pro else (None, None)
        y = ops.conv3x3(nhwc(x).to(dev), wf, Cout, pro, sd, td)
        e_f = relerr(nchw(y), y_ref)
        e_d = None
        if wdg.dtype == torch.uint8:                       # dgrad needs Cin % 64 == 0
            da = ops.conv3x3(nhwc(dy).to(dev), wdg, Cin)
            e_d = relerr(nchw(da), da_ref)
        e_w = None
        if Cin % 64 == 0:
            dw_ref = torch.nn.grad.conv2d_weight(a, w.shape, dy.double(), 1, 1)
            dw = ops.conv3x3_wgrad(nhwc(x).to(dev), nhwc(dy).to(dev), pro, sd, td)
            e_w = relerr(dw, dw_ref)
    finally:
        ops.CONV_MATH = old
    # what the exact-fp32 kernel gives on the same data
    pf, _ = ops.pack_conv_weight(w.to(dev))
    e_32 = relerr(nchw(ops.conv3x3(nhwc(x).to(dev), pf, Cout, pro, sd, td)), y_ref)
    print(f"{math_} conv {B}x{H}x{W} {Cin}->{Cout} pro {pro}: fwd err {e_f:.2e} (exact-fp32 kernel {e_32:.2e})"
          + (f", dgrad err {e_d:.2e}" if e_d is not None else "") + (f", wgrad err {e_w:.2e}" if e_w is not None else ""))
    assert e_f < 5e-6 and (e_d is None or e_d < 5e-6) and (e_w is None or e_w < 5e-6)


@pytest.mark.parametrize("math_", ["x3", "x9"])
def test_conv3x3_x3_extreme_dynamic_range(ops, dev, math_):
    """x3 / x9 arithmetic with operands spanning 16 orders of magnitude across channels (1e-8 ... 1e+8): the exact 3-way bf16
    split keeps fp32 accuracy wherever fp32 itself does (bf16 has fp32's exponent range)."""
    B, H, W, Cin, Cout = 2, 33, 16, 128, 128
    g = torch.Generator().manual_seed(99)
    sc = torch.exp(6.0 * torch.randn(1, Cin, 1, 1, generator=g)).clamp(1e-8, 1e8)
    x = torch.randn(B, Cin, H, W, generator=g) * sc
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin) / sc.view(1, Cin, 1, 1)     # products are O(1)
    dy = torch.randn(B, Cout, H, W, generator=g)
    y_ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    dw_ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), 1, 1)
    old = ops.CONV_MATH
    ops.CONV_MATH = math_
    try:
        wf, _ = ops.pack_conv_weight(w.to(dev), W=W)
        assert wf.products == (6 if math_ == "x3" else 9)
        e_f = relerr(nchw(ops.conv3x3(nhwc(x).to(dev), wf, Cout)), y_ref)
        dw = ops.conv3x3_wgrad(nhwc(x).to(dev), nhwc(dy).to(dev))
    finally:
        ops.CONV_MATH = old
    pf, _ = ops.pack_conv_weight(w.to(dev))
    e_32 = relerr(nchw(ops.conv3x3(nhwc(x).to(dev), pf, Cout)), y_ref)
    # weight gradients span the same 16 decades: compare per input channel
    e_w = ((dw.cpu().double() - dw_ref).abs().amax(dim=(0, 2, 3)) / dw_ref.abs().amax(dim=(0, 2, 3))).max().item()
    print(f"{math_} extreme range: fwd err {e_f:.2e} (exact-fp32 kernel {e_32:.2e}), wgrad per-channel err {e_w:.2e}; "
          f"operand scales {sc.min().item():.1e} .. {sc.max().item():.1e}")
    assert e_f < 5e-6 and e_w < 5e-6


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 21, 32, 64, 64), (1, 17, 16, 64, 128), (2, 250, 8, 256, 512), (2, 500, 32, 128, 128)])
def test_conv3x3_bf16_mode(ops, dev, B, H, W, Cin, Cout):
    """CONV_MATH = "bf16" (BASELINE configs[2] arithmetic): operands rounded to nearest bf16, ONE product per multiply on the
    bf16 MFMA, fp32 accumulation -> must equal the convolution of the bf16-rounded operands to fp32 round-off."""
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    dy = torch.randn(B, Cout, H, W, generator=g)
    xb, wb, dyb = x.bfloat16().double(), w.bfloat16().double(), dy.bfloat16().double()
    y_ref = F.conv2d(xb, wb, None, 1, 1)
    da_ref = torch.nn.grad.conv2d_input(x.shape, wb, dyb, 1, 1)
    dw_ref = torch.nn.grad.conv2d_weight(xb, w.shape, dyb, 1, 1)
    old = ops.CONV_MATH
    ops.CONV_MATH = "bf16"
    try:
        wf, wdg = ops.pack_conv_weight(w.to(dev), W=W)
        assert wf.products == 1
        e_f = relerr(nchw(ops.conv3x3(nhwc(x).to(dev), wf, Cout)), y_ref)
        e_d = relerr(nchw(ops.conv3x3(nhwc(dy).to(dev), wdg, Cin)), da_ref)
        e_w = relerr(ops.conv3x3_wgrad(nhwc(x).to(dev), nhwc(dy).to(dev)), dw_ref)
    finally:
        ops.CONV_MATH = old
    e_vs_fp32 = relerr(y_ref, F.conv2d(x.double(), w.double(), None, 1, 1))
    print(f"bf16 conv {B}x{H}x{W} {Cin}->{Cout}: vs bf16-rounded operands fwd {e_f:.2e} dgrad {e_d:.2e} wgrad {e_w:.2e}; "
          f"(bf16 rounding itself moves the result by {e_vs_fp32:.2e})")
    assert e_f < 5e-6 and e_d < 5e-6 and e_w < 5e-6


# ------------------------------------------------------------------------------------------- bf16 activation storage
def bf(t):
    return t.bfloat16()


def ulp_bf16(ref):
    """Half a bf16 ulp of |ref| (8 significand bits): the rounding budget of a value stored as bf16."""
    return ref.abs().double() * 2.0 ** -8 + 1e-30


@pytest.mark.parametrize("pro", [0, 1])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 21, 32, 64, 64), (1, 17, 16, 64, 128), (2, 250, 8, 256, 512), (2, 37, 64, 64, 64),
                                            (2, 500, 32, 128, 128), (2, 250, 16, 128, 256), (1, 19, 16, 256, 256),
                                            (1, 33, 32, 64, 256),
                                            # shapes of the row-streaming kernel (conv_rows.hip): strips of 8+ steps, a strip of
                                            # 2 steps, two rows per step (dgrad 128 -> 64 at W = 32; W = 16), two n-tiles
                                            (3, 1001, 64, 64, 64), (1, 2, 64, 64, 64), (2, 64, 32, 64, 128), (2, 40, 32, 64, 128),
                                            (1, 4, 32, 128, 128), (3, 6, 16, 128, 256), (2, 90, 16, 128, 128)])
def test_conv3x3_bf16_storage(ops, dev, B, H, W, Cin, Cout, pro):
    """configs[2] storage: bf16 tensors in, bf16 tensors out, fp32 accumulate.  Forward (with / without the producer's
    BN+ReLU folded into the operand load), dgrad and wgrad equal the fp64 convolution of the SAME bf16 operands up to one
    bf16 rounding of the result (forward / dgrad outputs are bf16) or fp32 round-off (wgrad output is fp32)."""
    _bf16_storage_case(ops, dev, B, H, W, Cin, Cout, pro)


@pytest.mark.parametrize("rows,dma", [(0, 1), (1, 0), (1, 2), (0, 0)])
@pytest.mark.parametrize("pro", [0, 1])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 1001, 64, 64, 64), (2, 64, 32, 64, 128), (2, 90, 16, 128, 128), (1, 19, 8, 256, 256)])
def test_conv3x3_bf16_storage_switchable_paths(ops, dev, B, H, W, Cin, Cout, pro, rows, dma):
    """The switchable code paths behind the bf16 entry points hold the SAME bounds as the defaults (round-4 advice: they had no
    test): tag_conv_rows_enable(0) = the tile-kernel fallback for the 64-channel shapes (unreached by default since conv_rows.hip),
    tag_wgrad_dma_enable(0) = the register-staged weight gradient everywhere, (2) = the DMA weight gradient with the producer's
    BatchNorm+ReLU applied in place in LDS (prologue-1 layers; off by default because it measures 3-5 % slower)."""
    from texttoaudiogrounding_amd.lib import query
    was_rows, was_dma = query("tag_conv_rows_enable", rows), query("tag_wgrad_dma_enable", dma)
    try:
        _bf16_storage_case(ops, dev, B, H, W, Cin, Cout, pro)
    finally:
        query("tag_conv_rows_enable", was_rows)
        query("tag_wgrad_dma_enable", was_dma)


def _bf16_storage_case(ops, dev, B, H, W, Cin, Cout, pro):
    g = torch.Generator().manual_seed(H + W + pro)
    x = bf(torch.randn(B, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    dy = bf(torch.randn(B, Cout, H, W, generator=g))
    sc, sh = torch.rand(Cin, generator=g) + 0.5, 0.3 * torch.randn(Cin, generator=g)
    wb = w.bfloat16().double()
    a = x.double()
    if pro:
        a = F.relu(a * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
        a = (a.float()).bfloat16().double()     # the kernel rounds the prologue's fp32 result to bf16 for the MFMA
    y_ref = F.conv2d(a, wb, None, 1, 1)
    da_ref = torch.nn.grad.conv2d_input(x.shape, wb, dy.double(), 1, 1)
    dw_ref = torch.nn.grad.conv2d_weight(a, w.shape, dy.double(), 1, 1)
    old = ops.CONV_MATH
    ops.CONV_MATH = "bf16"
    try:
        wf, wdg = ops.pack_conv_weight(w.to(dev), W=W)
        xs, dys = nhwc(x).to(dev), nhwc(dy).to(dev)
        assert xs.dtype == torch.bfloat16
        kw = dict(prologue=1, scale=sc.to(dev), shift=sh.to(dev)) if pro else {}
        y, part = ops.conv3x3_stats(xs, wf, Cout, want_stats=True, **kw)
        da = ops.conv3x3(dys, wdg, Cin)
        dw = ops.conv3x3_wgrad(xs, dys, **kw)
    finally:
        ops.CONV_MATH = old
    assert y.dtype == torch.bfloat16 and da.dtype == torch.bfloat16 and dw.dtype == torch.float32
    ey = ((nchw(y).cpu().double() - y_ref).abs() / (ulp_bf16(y_ref) + 1e-3 * 2.0 ** -8)).max().item()
    ed = ((nchw(da).cpu().double() - da_ref).abs() / (ulp_bf16(da_ref) + 1e-3 * 2.0 ** -8)).max().item()
    ew = relerr(dw, dw_ref)
    print(f"bf16 storage conv {B}x{H}x{W} {Cin}->{Cout} pro={pro}: fwd {ey:.2f} dgrad {ed:.2f} (in half-ulps of bf16), wgrad {ew:.2e}")
    # a prologue value that sits on a bf16 rounding boundary may round the other way in fp32 vs fp64: allow 2 half-ulps
    assert ey <= (2.0 if pro else 1.01) and ed <= 1.01 and ew < (2e-3 if pro else 1e-5)
    # BatchNorm statistics come from the fp32 accumulators (before the bf16 rounding of y)
    st = ops.bn_stats(y.view(-1, Cout), torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev), None, None, True,
                      partials=part)
    yd = y_ref.permute(0, 2, 3, 1).reshape(-1, Cout)
    assert relerr(st.mean, yd.mean(0)) < (5e-3 if pro else 2e-5)
    assert relerr(1.0 / st.invstd.cpu().double() ** 2, yd.var(0, unbiased=False) + 1e-5) < (5e-3 if pro else 2e-5)


@pytest.mark.parametrize("H,W,C,ph,pw,p", [(9, 8, 64, 2, 2, 0.0), (7, 8, 128, 1, 2, 0.2), (1001, 64, 64, 2, 2, 0.2),
                                           (250, 8, 512, 1, 2, 0.2)])
def test_bn_pool_kernels_bf16_storage(ops, dev, H, W, C, ph, pw, p):
    """The BatchNorm / ReLU / pool / dropout kernels on bf16 tensors = the fp32 kernels on the same (bf16-representable)
    values, up to the bf16 rounding of their outputs; per-channel sums (fp64) to fp32 round-off."""
    B = 2
    g = torch.Generator().manual_seed(H + C)
    y = bf(torch.randn(B, H, W, C, generator=g))
    gamma, beta = torch.rand(C, generator=g) + 0.5, 0.2 * torch.randn(C, generator=g)
    y32 = y.float().to(dev)
    st = ops.bn_stats(y32.view(-1, C), gamma.to(dev), beta.to(dev), None, None, True)
    seed = 777
    out32 = ops.bnact_pool(y32, st, ph, pw, 1, 0, p, seed)
    out16 = ops.bnact_pool(y.to(dev), st, ph, pw, 1, 0, p, seed)
    assert out16.dtype == torch.bfloat16
    assert ((out16.float() - out32).abs().cpu().double() / (ulp_bf16(out32.cpu()) + 1e-9)).max().item() <= 1.01
    dout = bf(torch.randn(out32.shape, generator=g))
    dy32, dg32, db32 = ops.bnrelu_pool_backward(y32, st, gamma.to(dev), dout.float().to(dev), ph, pw, p, seed)
    dy16, dg16, db16 = ops.bnrelu_pool_backward(y.to(dev), st, gamma.to(dev), dout.to(dev), ph, pw, p, seed)
    assert dy16.dtype == torch.bfloat16
    assert ((dy16.float() - dy32).abs().cpu().double() / (ulp_bf16(dy32.cpu()) + 1e-9)).max().item() <= 1.01
    assert relerr(dg16, dg32) < 1e-6 and relerr(db16, db32) < 1e-6
    da = bf(torch.randn(B, H, W, C, generator=g))
    e32, eg32, eb32 = ops.bnrelu_backward(y32, st, gamma.to(dev), da.float().to(dev), inplace=False)
    e16, eg16, eb16 = ops.bnrelu_backward(y.to(dev), st, gamma.to(dev), da.to(dev), inplace=False)
    assert ((e16.float() - e32).abs().cpu().double() / (ulp_bf16(e32.cpu()) + 1e-9)).max().item() <= 1.01
    assert relerr(eg16, eg32) < 1e-6 and relerr(eb16, eb32) < 1e-6


def test_conv_c1_bf16_storage(ops, dev):
    g = torch.Generator().manual_seed(5)
    B, H, W, Cout = 3, 101, 64, 64
    x = torch.randn(B, H, W, generator=g) * 10 - 30
    cs, ct = torch.rand(W, generator=g) * 0.1 + 0.05, torch.randn(W, generator=g)
    w = torch.randn(Cout, 1, 3, 3, generator=g) / 3
    y32, p32 = ops.conv3x3_c1_stats(x.to(dev), w.to(dev), cs.to(dev), ct.to(dev), want_stats=True)
    y16, p16 = ops.conv3x3_c1_stats(x.to(dev), w.to(dev), cs.to(dev), ct.to(dev), want_stats=True, out_dtype=torch.bfloat16)
    assert y16.dtype == torch.bfloat16 and torch.equal(y16, y32.bfloat16())          # same fp32 values, rounded once
    assert torch.equal(p16[1], p32[1])                                                # statistics from the fp32 values
    dy = bf(torch.randn(B, H, W, Cout, generator=g))
    dw32, dx32 = ops.conv3x3_c1_backward(x.to(dev), dy.float().to(dev), w.to(dev), cs.to(dev), ct.to(dev))
    dw16, dx16 = ops.conv3x3_c1_backward(x.to(dev), dy.to(dev), w.to(dev), cs.to(dev), ct.to(dev))
    assert torch.equal(dw16, dw32) and torch.equal(dx16, dx32)                        # fp32 outputs of identical inputs


@pytest.mark.parametrize("math_", ["fp32", "x3"])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 9, 8, 64, 128), (1, 17, 16, 64, 64), (2, 1001, 64, 64, 64), (3, 250, 8, 256, 512),
                                            (2, 501, 32, 64, 128)])
def test_conv3x3_fused_bn_stats(ops, dev, math_, B, H, W, Cin, Cout):
    """Batch statistics from the conv kernel's epilogue (per-tile sum / squared deviations merged in fp64) == the
    statistics of the conv output (fp64), incl. ragged last tiles (H not a multiple of the tile height) and a large mean."""
    g = torch.Generator().manual_seed(H + Cout)
    x = torch.randn(B, Cin, H, W, generator=g) + 2.0                   # non-zero mean -> cancellation matters
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin) + 0.02
    w[:3] = w[:3] * 1e-4 + 0.05                                        # channels whose |mean| is ~1e3 x their std
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    rm, rv = torch.randn(Cout, generator=g), torch.rand(Cout, generator=g) + 0.5
    old = ops.CONV_MATH
    ops.CONV_MATH = math_
    try:
        wf, _ = ops.pack_conv_weight(w.to(dev), W=W)
        y, part = ops.conv3x3_stats(nhwc(x).to(dev), wf, Cout)
    finally:
        ops.CONV_MATH = old
    assert part is not None
    rmd, rvd = rm.clone().to(dev), rv.clone().to(dev)
    st = ops.bn_stats(y.view(-1, Cout), gamma.to(dev), beta.to(dev), rmd, rvd, True, partials=part)
    yd = y.cpu().double().view(-1, Cout)                               # statistics OF THE KERNEL'S OUTPUT
    mean, var = yd.mean(0), yd.var(0, unbiased=False)
    n = yd.shape[0]
    assert relerr(st.mean, mean) < 1e-6
    assert ((st.invstd.cpu().double() - 1.0 / torch.sqrt(var + 1e-5)).abs() * torch.sqrt(var + 1e-5)).max() < 2e-6   # per channel
    assert relerr(st.scale, gamma.double() / torch.sqrt(var + 1e-5)) < 2e-6
    assert relerr(rmd, 0.9 * rm.double() + 0.1 * mean) < 1e-6
    assert relerr(rvd, 0.9 * rv.double() + 0.1 * var * n / (n - 1)) < 2e-6


@pytest.mark.parametrize("B,H", [(2, 21), (3, 1001)])
def test_conv3x3_c1(ops, dev, B, H):
    g = torch.Generator().manual_seed(7)
    W, Cout = 64, 64
    x = torch.randn(B, H, W, generator=g) * 10 - 30
    cs, ct = torch.rand(W, generator=g) * 0.1 + 0.05, torch.randn(W, generator=g)
    w = torch.randn(Cout, 1, 3, 3, generator=g) / 3
    xin = (x.double() * cs.double() + ct.double()).unsqueeze(1).requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y_ref = F.conv2d(xin, wd, None, 1, 1)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy.double())
    y = ops.conv3x3_c1(x.to(dev), w.to(dev), cs.to(dev), ct.to(dev))
    assert relerr(nchw(y), y_ref) < 2e-6
    dw = ops.conv3x3_c1_wgrad(x.to(dev), nhwc(dy).to(dev), cs.to(dev), ct.to(dev))
    assert relerr(dw, wd.grad) < 2e-6
    dx = ops.conv3x3_c1_dgrad(nhwc(dy).to(dev), w.to(dev))
    assert relerr(dx, xin.grad[:, 0]) < 2e-6
    dw2, dx2 = ops.conv3x3_c1_backward(x.to(dev), nhwc(dy).to(dev), w.to(dev), cs.to(dev), ct.to(dev))   # fused pass
    assert relerr(dw2, wd.grad) < 2e-6 and relerr(dx2, xin.grad[:, 0]) < 2e-6


@pytest.mark.parametrize("train", [True, False], ids=["train_bn", "eval_bn"])
@pytest.mark.parametrize("B,H", [(2, 21), (3, 1001), (5, 8)])
def test_conv3x3_c1_backward_with_fused_bnrelu_backward(ops, dev, B, H, train):
    """tag_conv3x3_c1_backward_bnrelu (bn1's backward applied while da is loaded) == tag_bnrelu_backward followed by
    tag_conv3x3_c1_backward: bit for bit in fp32 (same arithmetic, same order); with bf16 storage the fused pass skips the
    bf16 rounding of dy, so it is compared with the fp32 result of the same bf16-valued inputs (also bit for bit) and with
    the two-pass bf16 result within that one rounding.  Rows at the strip and image borders included (H = 8: one strip)."""
    g = torch.Generator().manual_seed(100 * B + H)
    W, C = 64, 64
    x = (torch.randn(B, H, W, generator=g) * 10 - 30).to(dev)
    cs, ct = (torch.rand(W, generator=g) * 0.1 + 0.05).to(dev), torch.randn(W, generator=g).to(dev)
    w = (torch.randn(C, 1, 3, 3, generator=g) / 3).to(dev)
    y = bf(torch.randn(B, H, W, C, generator=g) * 2 + 0.5).to(dev)
    da = bf(torch.randn(B, H, W, C, generator=g)).to(dev)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.3).to(dev)
    rm, rv = torch.zeros(C, device=dev) + 0.4, torch.ones(C, device=dev) * 3.0
    st = ops.bn_stats(y.float().view(-1, C), gamma, beta, rm, rv, train, 1e-5, 0.1)
    dy, dg, db = ops.bnrelu_backward(y.float(), st, gamma, da.float(), inplace=False)
    dw_ref, dx_ref = ops.conv3x3_c1_backward(x, dy, w, cs, ct)
    dw, dx = ops.conv3x3_c1_backward(x, da.float(), w, cs, ct, bn_bwd=(y.float(), st, gamma, dg, db))
    assert torch.equal(dw, dw_ref) and torch.equal(dx, dx_ref)
    dw16, dx16 = ops.conv3x3_c1_backward(x, da, w, cs, ct, bn_bwd=(y, st, gamma, dg, db))
    assert torch.equal(dw16, dw_ref) and torch.equal(dx16, dx_ref)
    dy16, dg16, db16 = ops.bnrelu_backward(y, st, gamma, da, inplace=False)
    dw2, dx2 = ops.conv3x3_c1_backward(x, dy16, w, cs, ct)
    # dy rounded to bf16 (relative 2^-9 per value) against an input of mean / std = 3: a few 1e-2 of max|dw| at most
    assert relerr(dw2, dw_ref) < 5e-2 and relerr(dx2, dx_ref) < 1e-2
    with pytest.raises(RuntimeError):
        ops.conv3x3_c1_backward(x, da, w, cs, ct, bn_bwd=(y.float(), st, gamma, dg, db))
    ops.check_async_errors()


@pytest.mark.parametrize("B,H", [(2, 21), (3, 1001), (64, 37)])
def test_conv3x3_c1_fused_bn_stats(ops, dev, B, H):
    """The Cin = 1 forward kernel writes the BatchNorm partial statistics of its own output (no second pass over the
    1 GB tensor); channels whose |mean| >> std keep their variance (pivoted sums)."""
    g = torch.Generator().manual_seed(B * H)
    W, Cout = 64, 64
    x = torch.randn(B, H, W, generator=g) * 10 - 30
    cs, ct = torch.rand(W, generator=g) * 0.1 + 0.05, torch.randn(W, generator=g) + 40.0     # strongly offset input
    w = torch.randn(Cout, 1, 3, 3, generator=g) / 3
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, 0.2 * torch.randn(Cout, generator=g)
    y, part = ops.conv3x3_c1_stats(x.to(dev), w.to(dev), cs.to(dev), ct.to(dev), want_stats=True)
    assert part is not None and part[0] == ops.query("tag_conv3x3_c1_stats_rows", B, H, W, Cout)
    rm, rv = torch.zeros(Cout), torch.ones(Cout)
    st = ops.bn_stats(y.view(-1, Cout), gamma.to(dev), beta.to(dev), rm.clone().to(dev), rv.clone().to(dev), True,
                      partials=part)
    st2 = ops.bn_stats(y.view(-1, Cout), gamma.to(dev), beta.to(dev), rm.clone().to(dev), rv.clone().to(dev), True)
    yd = y.cpu().double().view(-1, Cout)
    mean, var = yd.mean(0), yd.var(0, unbiased=False)
    assert relerr(st.mean, mean) < 1e-6 and relerr(st2.mean, mean) < 1e-6
    e = ((st.invstd.cpu().double() - 1.0 / torch.sqrt(var + 1e-5)).abs() * torch.sqrt(var + 1e-5)).max().item()
    print(f"c1 fused stats: invstd rel err {e:.2e}; |mean|/std up to {(mean.abs() / var.sqrt()).max().item():.1f}")
    assert e < 5e-6


# ------------------------------------------------------------------------------------------- bn+relu+pool
@pytest.mark.parametrize("H,W,C,ph,pw,train,p", [(9, 8, 64, 2, 2, True, 0.0), (7, 8, 128, 1, 2, True, 0.2),
                                                 (5, 6, 256, 2, 2, False, 0.0), (4, 4, 512, 1, 2, True, 0.0),
                                                 # BASELINE (10 s clip) shapes of the four Cnn8Rnn blocks
                                                 (1001, 64, 64, 2, 2, True, 0.2), (500, 32, 128, 2, 2, True, 0.2),
                                                 (250, 16, 256, 1, 2, True, 0.2), (250, 8, 512, 1, 2, True, 0.2)])
def test_bnrelu_pool_fwd_bwd(ops, dev, H, W, C, ph, pw, train, p):
    B = 2
    g = torch.Generator().manual_seed(H * 100 + C)
    y = torch.randn(B, C, H, W, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, 0.2 * torch.randn(C, generator=g)
    rm, rv = 0.1 * torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    yh = nhwc(y).to(dev)
    st = ops.bn_stats(yh.view(-1, C), gamma.to(dev), beta.to(dev), rm.clone().to(dev), rv.clone().to(dev), train)
    seed = 12345
    out = ops.bnact_pool(yh, st, ph, pw, 1, 0, p, seed)
    Ho, Wo = H // ph, W // pw
    mask = ops.dropout_mask(seed, (B, Ho, Wo, C), p, dev, pooled=True).cpu().permute(0, 3, 1, 2) if p > 0 else None
    yd = y.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    a = F.relu(F.batch_norm(yd, rm.double().clone(), rv.double().clone(), gd, bd, train, 0.1, 1e-5))
    ref = F.avg_pool2d(a, (ph, pw)) + F.max_pool2d(a, (ph, pw))
    if mask is not None:
        ref = ref * mask.double() / (1 - p)
    assert relerr(nchw(out), ref) < 2e-6
    dout = torch.randn(ref.shape, generator=g)
    ref.backward(dout.double())
    dy, dg, db = ops.bnrelu_pool_backward(yh, st, gamma.to(dev), nhwc(dout).to(dev), ph, pw, p, seed)
    if H < 100:
        assert relerr(nchw(dy), yd.grad) < 5e-6
        assert relerr(dg, gd.grad) < 5e-6 and relerr(db, bd.grad) < 5e-6
    else:
        # millions of ReLU / arg-max decisions: the handful whose operands sit within fp32 rounding of a tie may
        # legitimately fall the other way than in the fp64 reference; everything else must agree to round-off
        d = (nchw(dy).cpu().double() - yd.grad).abs() / yd.grad.abs().max()
        nbad = int((d > 5e-6).sum())
        print(f"bnrelu_pool_backward {B}x{H}x{W}x{C}: {nbad} of {d.numel()} elements beyond 5e-6 (decision ties)")
        assert nbad <= max(8, d.numel() // 500000)
        assert relerr(dg, gd.grad) < 1e-4 and relerr(db, bd.grad) < 1e-4


@pytest.mark.parametrize("B,Hf,Wf,Cin,C,ph,train,p", [
    (2, 18, 16, 64, 128, 2, True, 0.2), (1, 35, 32, 128, 64, 2, True, 0.0), (2, 21, 16, 256, 128, 1, True, 0.2),
    (2, 43, 64, 128, 64, 2, True, 0.2), (2, 13, 128, 64, 64, 2, False, 0.2), (3, 250, 16, 512, 256, 1, True, 0.2),
    (2, 1001, 64, 128, 64, 2, True, 0.2),
    # a 32-channel tensor (half of a 64-cout tile is empty: the BatchNorm / pool passes take C | 256 or multiples of 256), one clip,
    # a single pooled row, fewer rows than a tile
    (1, 5, 16, 32, 32, 2, False, 0.2), (3, 2, 32, 64, 32, 1, True, 0.0), (1, 9, 16, 64, 256, 2, True, 0.2)])
def test_conv_dgrad_fused_pool_backward_sums(ops, dev, B, Hf, Wf, Cin, C, ph, train, p):
    """One-read pool backward: tag_conv3x3_dgrad_poolsums (the dgrad conv of the NEXT block's first conv, with the sums of the
    BatchNorm+ReLU+pool backward below it in the epilogue) + tag_bn_grad_from_partials + tag_bnrelu_pool_backward_apply, against
    (a) the fp64 chain  a = relu(bn(y)); o = dropout(avg_pool(a) + max_pool(a)); u = conv(o, w); backward(du)  and (b) the
    two-pass kernels it replaces on the same dx (dy bit-identical: same apply kernel; dgamma / dbeta to summation round-off).
    Odd Hf (floor-dropped last row), 64-wide pooled images (two tile columns), 1x2 and 2x2 windows, eval-mode statistics."""
    pw = 2
    H, W = Hf // ph, Wf // pw
    g = torch.Generator().manual_seed(Hf * Wf + C)
    y = torch.randn(B, C, Hf, Wf, generator=g) * (1.0 + torch.arange(C).view(1, C, 1, 1) % 5) + 0.3
    w = torch.randn(Cin, C, 3, 3, generator=g) / math.sqrt(9 * C)       # the conv that CONSUMES the pooled output: C -> Cin
    gamma, beta = torch.rand(C, generator=g) + 0.5, 0.2 * torch.randn(C, generator=g)
    rm, rv = 0.1 * torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    du = torch.randn(B, Cin, H, W, generator=g)
    yh = nhwc(y).to(dev)
    st = ops.bn_stats(yh.view(-1, C), gamma.to(dev), beta.to(dev), rm.clone().to(dev), rv.clone().to(dev), train)
    seed = 4242
    mask = ops.dropout_mask(seed, (B, H, W, C), p, dev, pooled=True).cpu().permute(0, 3, 1, 2) if p > 0 else None
    _, wdg = ops.pack_conv_weight(w.to(dev), W=W)
    duh = nhwc(du).to(dev)
    assert ops.pool_sums_fusable(duh, wdg, yh, ph, pw)
    dx, part = ops.conv3x3_dgrad_poolsums(duh, wdg, yh, st, ph, pw, p, seed)
    dy, dg, db = ops.bnrelu_pool_backward(yh, st, gamma.to(dev), dx, ph, pw, p, seed, partials=part)
    # (b) the two-pass kernels on the same dx
    dx_plain = ops.conv3x3(duh, wdg, C)
    assert torch.equal(dx, dx_plain)
    dy2, dg2, db2 = ops.bnrelu_pool_backward(yh, st, gamma.to(dev), dx, ph, pw, p, seed)
    assert relerr(dg, dg2.cpu().double()) < 2e-6 and relerr(db, db2.cpu().double()) < 2e-6
    if not train:
        assert torch.equal(dy, dy2)          # eval statistics: dy does not depend on the sums
    else:
        assert relerr(dy, dy2.cpu().double()) < 2e-6
    if Hf > 300:
        return                                # the big shape is a kernel-vs-kernel check (fp64 reference of 1001 x 64 x 64: slow)
    # (a) fp64 autograd
    yd = y.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    a = F.relu(F.batch_norm(yd, rm.double().clone(), rv.double().clone(), gd, bd, train, 0.1, 1e-5))
    o = F.avg_pool2d(a, (ph, pw)) + F.max_pool2d(a, (ph, pw))
    if mask is not None:
        o = o * mask.double() / (1 - p)
    F.conv2d(o, w.double(), None, 1, 1).backward(du.double())
    e_dy, e_dg, e_db = relerr(nchw(dy), yd.grad), relerr(dg, gd.grad), relerr(db, bd.grad)
    print(f"fused pool sums {B}x{Hf}x{Wf} {C}<-{Cin} window {ph}x2 train={train} p={p}: dy {e_dy:.2e} dgamma {e_dg:.2e} dbeta {e_db:.2e}")
    assert e_dy < 1e-5 and e_dg < 1e-5 and e_db < 1e-5


@pytest.mark.parametrize("B,Hf,Wf,Cin,C,ph,p", [(2, 36, 16, 256, 128, 2, 0.2), (2, 21, 16, 512, 256, 1, 0.2), (3, 250, 16, 512, 256, 1, 0.2),
                                               (2, 70, 64, 128, 64, 2, 0.2), (1, 35, 32, 256, 128, 2, 0.0), (2, 500, 32, 256, 128, 2, 0.2)])
def test_conv_dgrad_fused_pool_backward_sums_bf16(ops, dev, B, Hf, Wf, Cin, C, ph, p):
    """The bf16-storage twin (BASELINE configs[2] mode): tag_conv3x3_dgrad_poolsums_bf16 takes the sums from the bf16 tile it stages
    for its own output -- the very values the two-pass kernels read back from HBM -- so dgamma / dbeta agree with the two-pass
    kernels to fp32 summation round-off and dy (bf16) is identical up to the rare element whose fp32 value sits on a bf16 rounding
    boundary.  The conv output itself is bit-identical to the plain dgrad launch."""
    pw = 2
    H, W = Hf // ph, Wf // pw
    g = torch.Generator().manual_seed(Hf * Wf + C)
    y = bf(torch.randn(B, Hf, Wf, C, generator=g) * (1.0 + torch.arange(C).view(1, 1, 1, C) % 5) + 0.3).to(dev)
    w = torch.randn(Cin, C, 3, 3, generator=g) / math.sqrt(9 * C)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (0.2 * torch.randn(C, generator=g)).to(dev)
    du = bf(torch.randn(B, H, W, Cin, generator=g)).to(dev)
    st = ops.bn_stats(y.float().view(-1, C), gamma, beta, None, None, True)
    seed = 777
    old, old_sw = ops.CONV_MATH, ops.FUSE_POOL_BWD_SUMS_BF16
    ops.CONV_MATH, ops.FUSE_POOL_BWD_SUMS_BF16 = "bf16", True        # off by default (measured neutral): exercised here
    try:
        _, wdg = ops.pack_conv_weight(w.to(dev), W=W)
        if not ops.pool_sums_fusable(du, wdg, y, ph, pw):
            pytest.skip("this shape goes to the row-streaming kernel: two-pass pool backward")
        dx, part = ops.conv3x3_dgrad_poolsums(du, wdg, y, st, ph, pw, p, seed)
        dx_plain = ops.conv3x3(du, wdg, C)
    finally:
        ops.CONV_MATH, ops.FUSE_POOL_BWD_SUMS_BF16 = old, old_sw
    assert dx.dtype == torch.bfloat16 and torch.equal(dx, dx_plain)
    dy, dg, db = ops.bnrelu_pool_backward(y, st, gamma, dx, ph, pw, p, seed, partials=part)
    dy2, dg2, db2 = ops.bnrelu_pool_backward(y, st, gamma, dx, ph, pw, p, seed)
    e_g, e_b = relerr(dg, dg2.cpu().double()), relerr(db, db2.cpu().double())
    diff = (dy.float() != dy2.float()).float().mean().item()
    print(f"bf16 fused pool sums {B}x{Hf}x{Wf} {C}<-{Cin} window {ph}x2 p={p}: dgamma {e_g:.2e} dbeta {e_b:.2e}, dy elements differing {diff:.2e}")
    assert e_g < 5e-6 and e_b < 5e-6 and diff < 5e-4
    # an element differs only where the 1e-7 relative difference of the folded sums moves its fp32 value across a bf16 rounding
    # boundary: by one bf16 ulp of the value, or -- where dy is a near-cancellation of its three terms -- by 1e-5 of the tensor's range
    d, r = (dy.float() - dy2.float()).abs().cpu().double(), dy2.float().abs().cpu().double()
    assert bool((d <= r * 2.0 ** -7 + 1e-5 * r.max()).all())


@pytest.mark.parametrize("B,H,W,Cin,C,ph,pro,pool", [
    (2, 19, 16, 64, 128, 2, 1, 0), (1, 35, 32, 128, 64, 2, 0, 0), (2, 21, 16, 256, 256, 1, 1, 0), (2, 43, 64, 64, 64, 2, 1, 0),
    (2, 250, 8, 512, 512, 1, 1, 0), (1, 3001, 64, 64, 64, 2, 1, 0), (2, 18, 8, 64, 128, 2, 0, 2), (2, 17, 32, 64, 64, 1, 1, 3),
    # half-empty n-tile (32 channels), fewer rows than a tile, one row
    (1, 9, 16, 64, 32, 2, 1, 0), (3, 2, 32, 64, 256, 1, 0, 0), (2, 1, 8, 32, 32, 1, 1, 0)])
def test_conv_fused_bnrelu_pool_eval(ops, dev, monkeypatch, B, H, W, Cin, C, ph, pro, pool):
    """Inference forward of a ConvBlock stage in ONE kernel (tag_conv3x3_forward_bnrelu_pool_eval, EPI == 3: the conv pools its own
    output tile, the raw conv output never touches HBM) against the two kernels it replaces: BIT-identical, and against the fp64
    chain conv -> BatchNorm(eval) -> ReLU -> avg / max pool (models/panns.py:49-60).  Odd heights (floor-dropped last row), 64-wide
    images (two tile columns), 1x2 and 2x2 windows, with and without the producer's BatchNorm+ReLU prologue, the three pool types,
    and the 30 s clip length of BASELINE configs[4].  (The DIRECT kernel: the Winograd twin the inference forward uses for the
    deep layers has its own test, tests/test_gpu_wino.py::test_wino_forward_bnrelu_pool_eval.)"""
    monkeypatch.setattr(ops, "CONV_WINOGRAD_EVAL", False)
    pw = 2
    g = torch.Generator().manual_seed(H * W + C + pool)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(C, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    sc_in, sh_in = torch.rand(Cin, generator=g) + 0.5, 0.3 * torch.randn(Cin, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, 0.2 * torch.randn(C, generator=g)
    rm, rv = 0.1 * torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    xh = nhwc(x).to(dev)
    wf, _ = ops.pack_conv_weight(w.to(dev), W=W)
    st = ops.bn_stats(gamma.view(1, C).to(dev), gamma.to(dev), beta.to(dev), rm.to(dev), rv.to(dev), False)
    kw = dict(prologue=1, scale=sc_in.to(dev), shift=sh_in.to(dev)) if pro else {}
    assert ops.eval_pool_fusable(xh, wf, ph, pw, pool)
    out = ops.conv3x3_bnrelu_pool_eval(xh, wf, C, st, ph, pw, pool=pool, **kw)
    y = ops.conv3x3(xh, wf, C, **kw)
    two = ops.bnact_pool(y, st, ph, pw, 1, pool, 0.0, 0)
    assert out.shape == (B, H // ph, W // pw, C) and torch.equal(out, two)
    if H > 1000:
        return
    a = x.double()
    if pro:
        a = F.relu(a * sc_in.double().view(1, -1, 1, 1) + sh_in.double().view(1, -1, 1, 1))
    z = F.relu(F.batch_norm(F.conv2d(a, w.double(), None, 1, 1), rm.double(), rv.double(), gamma.double(), beta.double(), False, 0.1, 1e-5))
    ref = {0: F.avg_pool2d(z, (ph, pw)) + F.max_pool2d(z, (ph, pw)), 2: F.avg_pool2d(z, (ph, pw)), 3: F.max_pool2d(z, (ph, pw))}[pool]
    assert relerr(nchw(out), ref) < 5e-6


def test_bnrelu_backward(ops, dev):
    B, C, H, W = 2, 128, 5, 6
    g = torch.Generator().manual_seed(3)
    y = torch.randn(B, C, H, W, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, 0.2 * torch.randn(C, generator=g)
    yh = nhwc(y).to(dev)
    st = ops.bn_stats(yh.view(-1, C), gamma.to(dev), beta.to(dev), None, None, True)
    yd = y.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    a = F.relu(F.batch_norm(yd, None, None, gd, bd, True, 0.1, 1e-5))
    da = torch.randn(a.shape, generator=g)
    a.backward(da.double())
    dy, dg, db = ops.bnrelu_backward(yh, st, gamma.to(dev), nhwc(da).to(dev))
    assert relerr(nchw(dy), yd.grad) < 5e-6
    assert relerr(dg, gd.grad) < 5e-6 and relerr(db, bd.grad) < 5e-6


@pytest.mark.parametrize("B,H,W,Cin,C", [(2, 9, 8, 64, 128), (1, 17, 16, 128, 64), (2, 21, 32, 64, 64), (2, 33, 64, 64, 64),
                                         (3, 250, 8, 512, 512)])
def test_conv_dgrad_fused_bnrelu_backward(ops, dev, B, H, W, Cin, C):
    """tag_conv3x3_dgrad_bnsums + tag_bn_grad_from_partials + tag_bnrelu_backward_apply (BatchNorm-backward sums in the
    dgrad conv epilogue) against the fp64 chain  a = relu(bn(y)); u = conv(a, w); backward(du)."""
    g = torch.Generator().manual_seed(H * W + C)
    y = torch.randn(B, C, H, W, generator=g) * (1.0 + torch.arange(C).view(1, C, 1, 1) % 5) + 0.3
    w = torch.randn(Cin, C, 3, 3, generator=g) / math.sqrt(9 * C)       # the conv that CONSUMES relu(bn(y)): C -> Cin
    gamma, beta = torch.rand(C, generator=g) + 0.5, 0.2 * torch.randn(C, generator=g)
    du = torch.randn(B, Cin, H, W, generator=g)
    yd = y.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    a = F.relu(F.batch_norm(yd, None, None, gd, bd, True, 0.1, 1e-5))
    F.conv2d(a, w.double(), padding=1).backward(du.double())
    yh = nhwc(y).to(dev)
    st = ops.bn_stats(yh.view(-1, C), gamma.to(dev), beta.to(dev), None, None, True)
    _, wd = ops.pack_conv_weight(w.to(dev), W=W)                          # dgrad pack: Cin -> C
    assert ops.FUSE_BN_BWD_SUMS and ops.query("tag_conv3x3_stats_rows", B, H, W, C) > 0
    dy, dg, db = ops.conv3x3_dgrad_bnrelu_backward(nhwc(du).to(dev), wd, yh, st, gamma.to(dev))
    e = (relerr(nchw(dy), yd.grad), relerr(dg, gd.grad), relerr(db, bd.grad))
    print(f"fused dgrad+BN backward: dy {e[0]:.2e} dgamma {e[1]:.2e} dbeta {e[2]:.2e}")
    assert max(e) < 1e-5
    # and equal (to rounding) to the unfused kernels: like with like -- the direct conv both times (a launch large enough for the
    # Winograd form, tests/test_gpu_wino.py, would differ from the direct conv by both kernels' rounding)
    old, oldw = ops.FUSE_BN_BWD_SUMS, ops.CONV_WINOGRAD
    ops.CONV_WINOGRAD = False
    try:
        dy, dg, db = ops.conv3x3_dgrad_bnrelu_backward(nhwc(du).to(dev), wd, yh, st, gamma.to(dev))
        ops.FUSE_BN_BWD_SUMS = False
        dy2, dg2, db2 = ops.conv3x3_dgrad_bnrelu_backward(nhwc(du).to(dev), wd, yh, st, gamma.to(dev))
    finally:
        ops.FUSE_BN_BWD_SUMS, ops.CONV_WINOGRAD = old, oldw
    assert relerr(dy, dy2) < 2e-6 and relerr(dg, dg2) < 2e-6 and relerr(db, db2) < 2e-6


@pytest.mark.parametrize("B,H,W,Cin,C", [(2, 9, 8, 64, 128), (1, 17, 16, 128, 64), (2, 21, 32, 64, 64), (2, 33, 64, 64, 64),
                                         (3, 250, 8, 512, 512), (2, 45, 16, 256, 256),
                                         # row-streaming kernel (conv_rows.hip), epilogue 2: sums over whole strips
                                         (2, 1001, 64, 64, 64), (2, 500, 32, 128, 128), (1, 3, 32, 128, 128)])
def test_conv_dgrad_fused_bnrelu_backward_bf16(ops, dev, monkeypatch, B, H, W, Cin, C):
    """BASELINE configs[2] mode: tag_conv3x3_dgrad_bnsums_bf16 (sums from the fp32 accumulators of the bf16 dgrad conv) +
    tag_bnrelu_backward_apply_bf16 against (a) the unfused bf16 kernels and (b) the fp64 chain on the same bf16 tensors."""
    monkeypatch.setattr(ops, "CONV_MATH", "bf16")
    g = torch.Generator().manual_seed(H * W + C + 1)
    bf = lambda t: t.bfloat16().float()
    y = bf(torch.randn(B, C, H, W, generator=g) * (1.0 + torch.arange(C).view(1, C, 1, 1) % 5) + 0.3)
    w = bf(torch.randn(Cin, C, 3, 3, generator=g) / math.sqrt(9 * C))
    gamma, beta = torch.rand(C, generator=g) + 0.5, 0.2 * torch.randn(C, generator=g)
    du = bf(torch.randn(B, Cin, H, W, generator=g))
    yd = y.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    a = F.relu(F.batch_norm(yd, None, None, gd, bd, True, 0.1, 1e-5))
    F.conv2d(a, w.double(), padding=1).backward(du.double())
    yh = nhwc(y).to(dev)
    st = ops.bn_stats(yh.view(-1, C), gamma.to(dev), beta.to(dev), None, None, True)
    _, wd = ops.pack_conv_weight(w.to(dev), W=W)
    assert wd.dtype == torch.uint8 and wd.products == 1
    yb, dub = yh.bfloat16(), nhwc(du).to(dev).bfloat16()
    dy, dg, db = ops.conv3x3_dgrad_bnrelu_backward(dub, wd, yb, st, gamma.to(dev))
    assert dy.dtype == torch.bfloat16
    monkeypatch.setattr(ops, "FUSE_BN_BWD_SUMS", False)
    dy2, dg2, db2 = ops.conv3x3_dgrad_bnrelu_backward(dub, wd, yb, st, gamma.to(dev))
    e_un = (relerr(dy.float(), dy2.float()), relerr(dg, dg2), relerr(db, db2))
    e_64 = (relerr(nchw(dy.float()), yd.grad), relerr(dg, gd.grad), relerr(db, bd.grad))
    e_64u = (relerr(nchw(dy2.float()), yd.grad), relerr(dg2, gd.grad), relerr(db2, bd.grad))
    print(f"bf16 fused dgrad+BN backward vs unfused {e_un[0]:.1e} {e_un[1]:.1e} {e_un[2]:.1e}; vs fp64: fused "
          f"{e_64[0]:.1e} {e_64[1]:.1e} {e_64[2]:.1e}, unfused {e_64u[0]:.1e} {e_64u[1]:.1e} {e_64u[2]:.1e}")
    # dgamma / dbeta: sums over >= 1000 pixels of values whose bf16 rounding (2^-9) the fused path skips -> it must be at least
    # as close to fp64 as the unfused path (up to noise), and both agree to a few 1e-3; dy carries one bf16 rounding
    assert e_64[1] < max(2 * e_64u[1], 2e-3) and e_64[2] < max(2 * e_64u[2], 2e-3)
    assert e_un[1] < 1e-2 and e_un[2] < 1e-2 and e_64[0] < 2e-2 and e_un[0] < 2e-2


@pytest.mark.parametrize("pre", [0, 1])
def test_bn_act_backward(ops, dev, pre):
    """BatchNorm in front of a conv (CrnnEncoder cdur_block): u = bn(pre(x)), pre = identity | leaky_relu(0.1)."""
    B, C, H, W = 2, 32, 7, 6
    g = torch.Generator().manual_seed(11 + pre)
    x = torch.randn(B, C, H, W, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, 0.2 * torch.randn(C, generator=g)
    xh = nhwc(x).to(dev)
    st = ops.bn_stats(xh.view(-1, C), gamma.to(dev), beta.to(dev), None, None, True, pre_op=pre)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    v = F.leaky_relu(xd, 0.1) if pre else xd
    u = F.batch_norm(v, None, None, gd, bd, True, 0.1, 1e-5)
    du = torch.randn(u.shape, generator=g)
    u.backward(du.double())
    dx, dg, db = ops.bn_act_backward(xh, pre, st, gamma.to(dev), nhwc(du).to(dev))
    assert relerr(nchw(dx), xd.grad) < 5e-6
    assert relerr(dg, gd.grad) < 5e-6 and relerr(db, bd.grad) < 5e-6


@pytest.mark.parametrize("H,W,C,ph,pw,p", [(9, 16, 32, 2, 4, 0.0), (6, 4, 128, 1, 4, 0.3), (5, 8, 128, 2, 4, 0.0)])
def test_lppool_leaky_fwd_bwd(ops, dev, H, W, C, ph, pw, p):
    B = 2
    g = torch.Generator().manual_seed(H + C)
    y = torch.randn(B, C, H, W, generator=g)
    yh = nhwc(y).to(dev)
    seed = 777
    out = ops.bnact_pool(yh, None, ph, pw, act=2, pool=1, drop_p=p, seed=seed)
    yd = y.double().requires_grad_(True)
    ref = F.lp_pool2d(F.leaky_relu(yd, 0.1), 4.0, (ph, pw))
    if p > 0:
        mask = ops.dropout_mask(seed, (B, H // ph, W // pw, C), p, dev, pooled=True).cpu().permute(0, 3, 1, 2)
        ref = ref * mask.double() / (1 - p)
    assert relerr(nchw(out), ref) < 2e-6
    dout = torch.randn(ref.shape, generator=g)
    ref.backward(dout.double())
    dy = ops.lppool_leaky_backward(yh, nhwc(dout).to(dev), ph, pw, p, seed)
    assert relerr(nchw(dy), yd.grad) < 5e-6


def test_bn_param_grad_and_mean_w(ops, dev):
    g = torch.Generator().manual_seed(4)
    rows, C = 1001, 64
    x = torch.randn(rows, C, generator=g) * 5 - 30
    dy = torch.randn(rows, C, generator=g)
    st = ops.bn_stats(x.to(dev), None, None, None, None, True)
    dg, db = ops.bn_param_grad(x.to(dev), dy.to(dev), st)
    xd = x.double()
    xhat = (xd - xd.mean(0)) / torch.sqrt(xd.var(0, unbiased=False) + 1e-5)
    assert relerr(dg, (dy.double() * xhat).sum(0)) < 1e-5 and relerr(db, dy.double().sum(0)) < 1e-5
    from texttoaudiogrounding_amd.lib import call, ptr
    xm = torch.randn(37, 4, 512, generator=g)
    out = torch.empty(37, 512, device=dev)
    call("tag_mean_w_forward", ptr(xm.to(dev)), 37, 4, 512, 0.0, 0, ptr(out))
    assert relerr(out, xm.double().mean(1)) < 1e-6
    dx = torch.empty(37, 4, 512, device=dev)
    call("tag_mean_w_backward", ptr(out), 37, 4, 512, 0.0, 0, ptr(dx))
    assert relerr(dx, (out.cpu().double() / 4).unsqueeze(1).expand(37, 4, 512)) < 1e-6


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K,ta,tb", [(100, 512, 512, False, True), (70, 33, 129, False, False),
                                         (65, 64, 96, True, False), (130, 7, 260, True, True),
                                         (1536, 512, 300, True, False)])
def test_gemm(ops, dev, M, N, K, ta, tb):
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    Bm = torch.randn((N, K) if tb else (K, N), generator=g)
    bias = torch.randn(N, generator=g)
    ref = (A.double().t() if ta else A.double()) @ (Bm.double().t() if tb else Bm.double())
    out = ops.gemm(A.to(dev), Bm.to(dev), M, N, K, ta, tb)
    assert relerr(out, ref) < 2e-6
    out2 = ops.gemm(A.to(dev), Bm.to(dev), M, N, K, ta, tb, bias=bias.to(dev), act=1)
    assert relerr(out2, F.relu(ref + bias.double())) < 2e-6
    acc = out.clone()
    ops.gemm(A.to(dev), Bm.to(dev), M, N, K, ta, tb, out=acc, accumulate=True)
    assert relerr(acc, 2 * ref) < 2e-6
    cs = ops.colsum(out2, M, N)
    assert relerr(cs, out2.cpu().double().sum(0)) < 1e-6


@pytest.mark.parametrize("M,N,K,ta,tb", [(750, 512, 512, False, True), (300, 1536, 512, False, True), (768, 512, 4000, True, False),
                                         (100, 96, 40, False, False), (129, 65, 33, True, True)])
def test_gemm_bf16_mode(ops, dev, monkeypatch, M, N, K, ta, tb):
    """tag_gemm_bf16 (BASELINE configs[2] mode): equal to the fp64 product of the bf16-ROUNDED operands up to fp32 accumulation
    (products of bf16 values are exact in fp32), and within bf16 rounding of the fp32 product."""
    monkeypatch.setattr(ops, "GEMM_MATH", "bf16")
    g = torch.Generator().manual_seed(M + N + 1)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    Bm = torch.randn((N, K) if tb else (K, N), generator=g)
    bias = torch.randn(N, generator=g)
    r = lambda t: t.bfloat16().double()
    ref = (r(A).t() if ta else r(A)) @ (r(Bm).t() if tb else r(Bm))
    out = ops.gemm(A.to(dev), Bm.to(dev), M, N, K, ta, tb)
    assert relerr(out, ref) < 2e-6
    out2 = ops.gemm(A.to(dev), Bm.to(dev), M, N, K, ta, tb, bias=bias.to(dev), act=1)
    assert relerr(out2, F.relu(ref + bias.double())) < 2e-6
    exact = (A.double().t() if ta else A.double()) @ (Bm.double().t() if tb else Bm.double())
    assert relerr(out, exact) < 2e-2 * math.sqrt(K) / math.sqrt(K)          # one bf16 rounding per operand
    monkeypatch.setattr(ops, "GEMM_MATH", "fp32")
    assert relerr(ops.gemm(A.to(dev), Bm.to(dev), M, N, K, ta, tb), exact) < 2e-6


@pytest.mark.parametrize("math_", ["fp32", "bf16"])
@pytest.mark.parametrize("M,N,K,ta,tb", [(256, 128, 512, False, True), (128, 192, 320, False, False), (192, 64, 2048, True, False),
                                         (64, 64, 64, True, True)])
def test_gemm_whole_tile_loader_is_the_same_arithmetic(ops, dev, monkeypatch, math_, M, N, K, ta, tb):
    """Whole, 16-byte-aligned problems take the kernel variant whose loader has no tail handling (gemm.hip FAST).  The same
    operands placed 4 bytes off a 16-byte boundary force the tail-aware loader: identical K order, identical products ->
    the two results are bit-identical (fp32 MFMA and bf16-operand MFMA, every transposition, split-K included)."""
    monkeypatch.setattr(ops, "GEMM_MATH", math_)
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    Bm = torch.randn((N, K) if tb else (K, N), generator=g)
    bias = torch.randn(N, generator=g).to(dev)

    def off4(t):                                       # same values, storage starting 4 bytes after a 16-byte boundary
        buf = torch.empty(t.numel() + 8, device=dev)
        start = 1 + ((-(buf.data_ptr() // 4)) % 4)
        v = buf[start:start + t.numel()].view(t.shape)
        v.copy_(t)
        assert v.data_ptr() % 16 == 4
        return v
    fast = ops.gemm(A.to(dev), Bm.to(dev), M, N, K, ta, tb, bias=bias, act=1)
    slow = ops.gemm(off4(A), off4(Bm), M, N, K, ta, tb, bias=bias, act=1)
    assert torch.equal(fast, slow)
    exact = F.relu((A.double().t() if ta else A.double()) @ (Bm.double().t() if tb else Bm.double()) + bias.cpu().double())
    assert relerr(fast, exact) < (2e-6 if math_ == "fp32" else 2e-2)
    ops.check_async_errors()


# ------------------------------------------------------------------------------------------- GRU
@pytest.mark.parametrize("B,T,I,H", [(3, 9, 64, 32), (18, 6, 512, 256), (64, 40, 512, 256), (5, 33, 128, 128),
                                     (130, 5, 64, 256)])
def test_gru_forward_backward(ops, dev, B, T, I, H):
    from texttoaudiogrounding_amd.lib import call, ptr
    g = torch.Generator().manual_seed(B)
    k = 1 / math.sqrt(H)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    shapes = [(3 * H, I), (3 * H, H), (3 * H,), (3 * H,)]
    st = {}
    for sfx in ("", "_reverse"):
        for n, s in zip(names, shapes):
            st["rnn." + n + sfx] = ((torch.rand(s, generator=g) * 2 - 1) * k * 2).double().requires_grad_(True)
    x = torch.randn(B, T, I, generator=g)
    xd = x.double().requires_grad_(True)
    y_ref = O.gru_bidir(xd, st, "rnn.")
    y_man = O.gru_bidir_manual(xd, st, "rnn.")
    assert (y_ref - y_man).abs().max().item() < 1e-12        # the oracle's two GRU statements agree
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy.double())

    f = lambda k_: st[k_].detach().float().to(dev)
    w_ih = torch.cat([f("rnn.weight_ih_l0"), f("rnn.weight_ih_l0_reverse")], 0)
    b_ih = torch.cat([f("rnn.bias_ih_l0"), f("rnn.bias_ih_l0_reverse")], 0)
    w_hh = torch.stack([f("rnn.weight_hh_l0"), f("rnn.weight_hh_l0_reverse")], 0).contiguous()
    b_hh = torch.stack([f("rnn.bias_hh_l0"), f("rnn.bias_hh_l0_reverse")], 0).contiguous()
    M = B * T
    xdv = x.to(dev).view(M, I)
    gi = ops.gemm(xdv, w_ih, M, 6 * H, I, transB=True, bias=b_ih)
    y = torch.empty(B, T, 2 * H, device=dev)
    gates = torch.empty(B, T, 2, 4 * H, device=dev)
    from texttoaudiogrounding_amd.lib import query
    nws = query("tag_gru_ws_bytes", B, T, H)
    ws = torch.zeros(nws // 4 + 1, device=dev)
    call("tag_gru_forward", ptr(gi), ptr(w_hh), ptr(b_hh), ptr(y), ptr(gates), ptr(ws), B, T, H)
    assert relerr(y, y_ref) < 5e-6
    assert int(ws.view(torch.int32)[(nws - 256) // 4].item()) == 0        # no bounded spin timed out
    dgi = torch.empty(B, T, 2, 3 * H, device=dev)
    dgh = torch.empty(B, T, 2, 3 * H, device=dev)
    hprev = torch.empty(B, T, 2, H, device=dev)
    scratch = torch.zeros(nws // 4 + 1, device=dev)
    call("tag_gru_backward", ptr(dy.to(dev)), ptr(y), ptr(gates), ptr(w_hh), ptr(dgi), ptr(dgh), ptr(hprev),
         ptr(scratch), B, T, H)
    dx = ops.gemm(dgi, w_ih, M, I, 6 * H)
    assert relerr(dx.view(B, T, I), xd.grad) < 2e-5
    dw_ih = ops.gemm(dgi, xdv, 6 * H, I, M, transA=True, lda=6 * H)
    assert relerr(dw_ih[:3 * H], st["rnn.weight_ih_l0"].grad) < 2e-5
    assert relerr(dw_ih[3 * H:], st["rnn.weight_ih_l0_reverse"].grad) < 2e-5
    for d, sfx in enumerate(("", "_reverse")):
        a = dgh.view(M, 6 * H)[:, d * 3 * H:]
        hb = hprev.view(M, 2 * H)[:, d * H:]
        dw_hh = ops.gemm(a, hb, 3 * H, H, M, transA=True, lda=6 * H, ldb=2 * H)
        assert relerr(dw_hh, st["rnn.weight_hh_l0" + sfx].grad) < 2e-5
    assert relerr(ops.colsum(dgh, M, 6 * H)[:3 * H], st["rnn.bias_hh_l0"].grad) < 2e-5
    assert relerr(ops.colsum(dgi, M, 6 * H)[3 * H:], st["rnn.bias_ih_l0_reverse"].grad) < 2e-5


# ------------------------------------------------------------------------------------------- heads
def test_embed_mean(ops, dev):
    g = torch.Generator().manual_seed(9)
    V, D, B, L = 50, 256, 5, 4
    table = torch.randn(V, D, generator=g)
    text = torch.randint(2, V, (B, L), generator=g)
    lens = torch.tensor([4, 1, 2, 3, 4])
    text[1, 1:] = 0
    td = table.double().requires_grad_(True)
    ref = O.embedding_agg_mean({"text_encoder.embedding.core.weight": td}, text, lens)
    tab = table.to(dev).requires_grad_(True)
    seq, tok = ops.EmbedMeanFunction.apply(tab, text.to(dev), lens.to(dev), True)
    assert relerr(seq, ref["seq_emb"]) < 1e-6 and relerr(tok, ref["token_emb"]) == 0
    dseq = torch.randn(B, D, generator=g)
    ref["seq_emb"].backward(dseq.double())
    seq.backward(dseq.to(dev))
    assert relerr(tab.grad, td.grad) < 1e-6


def test_embed_backward_repeated_ids_is_deterministic_and_flags_bad_ids(ops, dev):
    """Many (clip, position) pairs share table rows; seq_emb AND token_emb gradients flow; the table gradient equals the
    fp64 oracle and is bit-identical run to run (no atomics).  An id outside [0,V) raises at the next host check, as
    nn.Embedding would (models/text_encoder.py:39)."""
    g = torch.Generator().manual_seed(11)
    V, D, B, L = 7, 512, 64, 4                                   # 256 pairs over 7 rows
    table = torch.randn(V, D, generator=g)
    text = torch.randint(0, V, (B, L), generator=g)
    lens = 1 + torch.arange(B) % 4
    td = table.double().requires_grad_(True)
    ref = O.embedding_agg_mean({"text_encoder.embedding.core.weight": td}, text, lens)
    dseq, dtok = torch.randn(B, D, generator=g), torch.randn(B, L, D, generator=g)
    (ref["seq_emb"] * dseq.double()).sum().add((ref["token_emb"] * dtok.double()).sum()).backward()
    outs = []
    for _ in range(2):
        tab = table.to(dev).requires_grad_(True)
        seq, tok = ops.EmbedMeanFunction.apply(tab, text.to(dev), lens.to(dev), True)
        torch.autograd.backward([seq, tok], [dseq.to(dev), dtok.to(dev)])
        outs.append(tab.grad.clone())
    assert relerr(outs[0], td.grad) < 2e-6
    assert torch.equal(outs[0], outs[1])
    ops.check_async_errors()                                     # all ids valid: nothing raised
    bad = text.clone()
    bad[3, 0] = V + 5
    ops.EmbedMeanFunction.apply(table.to(dev), bad.to(dev), lens.to(dev), False)
    with pytest.raises(IndexError):
        ops.check_async_errors()
    ops.check_async_errors()                                     # the flag was cleared by the raise


def test_adam_skips_non_finite_gradient_norm(ops, dev):
    """One poisoned step (NaN from a timed-out GRU exchange) must not destroy parameters or moments."""
    n = 4099
    p = torch.randn(n, device=dev)
    m, v = torch.rand(n, device=dev), torch.rand(n, device=dev)
    p0, m0, v0 = p.clone(), m.clone(), v.clone()
    gr = torch.randn(n, device=dev)
    gr[17] = float("nan")
    gsq = ops.grad_sumsq(gr)
    assert not torch.isfinite(gsq).item()
    ops.adam_step(p, gr, m, v, 1e-3, 0.9, 0.999, 1e-8, 1, gsq, 1.0, 1.0)
    torch.cuda.synchronize()
    assert torch.equal(p, p0) and torch.equal(m, m0) and torch.equal(v, v0)


@pytest.mark.parametrize("kind,l2norm,scale", [(0, False, True), (0, True, False), (1, True, False), (1, False, False)])
def test_match_heads(ops, dev, golden_dir, kind, l2norm, scale):
    gold = np.load(f"{golden_dir}/heads.npz")
    audio, text = torch.from_numpy(gold["audio"]), torch.from_numpy(gold["text"])
    key = {(0, False, True): "dot", (0, True, False): "dot_l2", (1, True, False): "expnegl2",
           (1, False, False): "expnegl2_raw"}[(kind, l2norm, scale)]
    a = audio.to(dev).requires_grad_(True)
    t = text.to(dev).requires_grad_(True)
    sim = torch.ops.tag.frame_match(a, t, kind, l2norm, scale)
    assert (sim.cpu() - torch.from_numpy(gold["sim_" + key])).abs().max().item() < 1e-6    # golden (reference)
    ad, td = audio.double().requires_grad_(True), text.double().requires_grad_(True)
    ref = O.match_dot_product(ad, td, l2norm, scale) if kind == 0 else O.match_exp_neg_l2(ad, td, l2norm)
    g = torch.Generator().manual_seed(2)
    ds = torch.randn(ref.shape, generator=g)
    ref.backward(ds.double())
    sim.backward(ds.to(dev))
    assert relerr(a.grad, ad.grad) < 2e-5 and relerr(t.grad, td.grad) < 2e-5


def test_frame_bce(ops, dev, golden_dir):
    gold = np.load(f"{golden_dir}/heads.npz")
    sim = torch.from_numpy(gold["sim_dot"])
    label, length = torch.from_numpy(gold["label"]), torch.from_numpy(gold["length"])
    s = sim.to(dev).requires_grad_(True)
    loss = torch.ops.tag.frame_bce(s, label.to(dev), length.to(dev), sim.shape[1])
    assert abs(loss.item() - float(gold["loss_dot"])) < 1e-6                               # golden (reference)
    sd = sim.double().requires_grad_(True)
    ref = O.frame_bce_loss(sd, label.double(), length)
    ref.backward()
    loss.backward()
    assert relerr(s.grad, sd.grad) < 1e-5
    # saturated probabilities: log clamp at -100 and the 1e-12 guard of the backward
    edge = torch.tensor([[1.0, 1e-7, 0.5, 1.0]]), torch.tensor([[0.0, 1.0, 1.0, 1.0]])
    e = edge[0].to(dev).requires_grad_(True)
    le = torch.ops.tag.frame_bce(e, edge[1].to(dev), torch.tensor([4]).to(dev), 4)
    ed = edge[0].clone().requires_grad_(True)
    lr = O.frame_bce_loss(ed, edge[1], torch.tensor([4]))
    lr.backward()
    le.backward()
    assert abs(le.item() - lr.item()) < 1e-5 and relerr(e.grad, ed.grad) < 1e-5


def test_align_dot(ops, dev, golden_dir):
    gold = np.load(f"{golden_dir}/align.npz")
    audio, text = torch.from_numpy(gold["audio"]).to(dev), torch.from_numpy(gold["text"]).to(dev)
    for l2 in (0, 1):
        for sc in (0, 1):
            out = ops.align_dot(audio, text, bool(l2), bool(sc))
            ref = torch.from_numpy(gold[f"l2{l2}_sc{sc}"])
            assert out.shape == ref.shape and (out.cpu() - ref).abs().max().item() < 1e-6


@pytest.mark.parametrize("l2norm,scaled", [(False, False), (True, True), (False, True)])
def test_align_dot_backward(dev, l2norm, scaled):
    from texttoaudiogrounding_amd.models import align
    g = torch.Generator().manual_seed(31)
    audio, text = torch.randn(3, 21, 64, generator=g), torch.randn(3, 5, 64, generator=g)
    a, t = audio.to(dev).requires_grad_(True), text.to(dev).requires_grad_(True)
    out = align.DotProduct(l2norm=l2norm, scaled=scaled)(a, t)
    ad, td = audio.double().requires_grad_(True), text.double().requires_grad_(True)
    ref = O.align_dot_product(ad, td, l2norm, scaled)
    assert relerr(out, ref) < 2e-6
    dout = torch.randn(ref.shape, generator=g)
    ref.backward(dout.double())
    out.backward(dout.to(dev))
    assert relerr(a.grad, ad.grad) < 2e-5 and relerr(t.grad, td.grad) < 2e-5


def test_segments_bit_exact_vs_reference_golden(ops, dev, golden_dir):
    """P1: integer segment indices, bit-exact against what the reference's own eval_util produced."""
    gold = np.load(f"{golden_dir}/postproc.npz")
    rows, lens, th, segs = gold["rows"], gold["row_len"], gold["thresholds"], gold["segments"]
    off = np.concatenate([[0], np.cumsum(lens)])
    for T in sorted(set(lens.tolist())):
        idx = [i for i, l in enumerate(lens) if l == T]
        x = torch.from_numpy(np.stack([rows[off[i]:off[i + 1]] for i in idx])).to(dev)
        for window in (1, 3, 4):
            for n_connect in (7, 13):
                regions, counts = ops.segments(x, th, window, n_connect)
                regions, counts = regions.cpu().numpy(), counts.cpu().numpy()
                for bi, ri in enumerate(idx):
                    for ti in range(len(th)):
                        sel = (segs[:, 0] == ri) & (segs[:, 1] == ti) & (segs[:, 2] == window) & (segs[:, 3] == n_connect)
                        want = segs[sel][:, 4:6]
                        got = regions[bi, ti, :counts[bi, ti]]
                        assert np.array_equal(got, want), (ri, ti, window, n_connect)


def test_adam_clip_vs_torch(ops, dev):
    g = torch.Generator().manual_seed(5)
    n = 100003
    p0, grads = torch.randn(n, generator=g), [torch.randn(n, generator=g) * 0.01 * (i + 1) for i in range(3)]
    pr = p0.clone().double().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3)
    p = p0.clone().to(dev)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for i, gr in enumerate(grads):
        pr.grad = gr.double().clone()
        torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        gd = gr.to(dev)
        gsq = ops.grad_sumsq(gd)
        assert abs(gsq.item() - float((gr.double() ** 2).sum())) / float((gr.double() ** 2).sum()) < 1e-10
        ops.adam_step(p, gd, m, v, 1e-3, 0.9, 0.999, 1e-8, i + 1, gsq, 1.0, 1.0)
    assert (p.cpu().double() - pr.detach()).abs().max().item() < 1e-6


def test_halo_256_cout_tiles_forced_everywhere(dev):
    """TAG_HALO_BN256=2 puts EVERY halo launch with W <= 16 and Cout % 256 == 0 on the 256-cout workgroups -- also the launches the
    default keeps on 128-cout tiles (plain forward / dgrad without statistics, the inference epilogue EPI == 3).  The switch is read once
    per process, so the affected kernel tests run in a child process with it set."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TAG_HALO_BN256="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_kernels.py"), "-x", "-q", "-m", "gpu", "-k",
                        "fused_bnrelu_pool_eval or forward_dgrad_wgrad or fused_bn_stats or fused_pool_backward_sums and not bf16 "
                        "and not forced_everywhere"], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
