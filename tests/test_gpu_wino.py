"""Winograd F(2x2,3x3) form of the deep-layer convolutions (csrc/conv_wino.hip; models/panns.py:29-38,49-50) against an fp64
convolution: forward with every producer prologue and the fused BatchNorm batch statistics, dgrad with the fused
BatchNorm+ReLU-backward sums, odd heights / widths (tiles that hang over the image), tile counts that do and do not fill the
GEMM tiles, and the dispatch rule of dispatch.py (which launches take this path).  Tolerances: 5e-6 of the output range -- the
direct fp32 kernel's own bound in tests/test_gpu_kernels.py (measured: 1e-6, against 2e-6 ... 4e-6 of the direct kernel at 512
input channels: 16 chains of Cin products round less than one chain of 9 Cin)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


@pytest.fixture(scope="module")
def ops(dev):
    from texttoaudiogrounding_amd import ops as _ops
    return _ops


def wino_pack(ops, w):
    Cout, Cin = w.shape[:2]
    uf = torch.empty(16, Cin, Cout, device=w.device)
    ud = torch.empty(16, Cout, Cin, device=w.device)
    ops.call("tag_pack_conv_weight_wino", ops.ptr(w), ops.ptr(uf), ops.ptr(ud), Cin, Cout)
    return uf, ud


def prologue64(x, mode, s, t):
    if mode == 1:
        return torch.relu(x * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1))
    if mode == 2:
        return F.leaky_relu(x, 0.1) * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)
    if mode == 3:
        return x * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)
    return x


SHAPES = [(2, 10, 8, 64, 128, 1),      # whole tiles, 40-tile launch: ragged GEMM tiles
          (3, 9, 7, 128, 64, 0),       # odd height AND width: the last tile row / column hangs over the image
          (1, 1, 2, 32, 32, 3),        # a single tile row of one-pixel height
          (2, 37, 16, 256, 256, 2),    # the 1.5 s fixture's block-3 geometry
          (8, 64, 8, 512, 256, 1),     # 1024 tiles = whole 128-row GEMM tiles (the loader without tail handling)
          (2, 33, 32, 64, 128, 1),     # block 2's first conv (32-wide image, odd height), fused kernels (round 6)
          (1, 21, 64, 64, 64, 0),      # block 1's second conv (64-wide image, one 64-cout block)
          (2, 1, 8, 64, 64, 1),        # one-pixel-high images: every tile hangs over the bottom edge (th = 1)
          (1, 2, 16, 64, 1024, 0),     # the widest cout count the kernels take (16 cout blocks)
          (1, 6, 8, 1024, 64, 1),      # the deepest K loop (128 chunks) and the full scale / shift staging area
          (3, 10, 16, 128, 192, 1)]    # 192 couts = three 64-cout blocks; 120 tiles: a ragged last tile block


@pytest.mark.parametrize("B,H,W,Cin,Cout,pro", SHAPES)
def test_wino_forward_and_statistics(ops, dev, B, H, W, Cin, Cout, pro):
    """tag_conv3x3_wino_forward: y = conv(prologue(x)) and the partial statistics rows folded by tag_bn_stats_from_partials."""
    g = torch.Generator().manual_seed(B * H + W + Cin)
    x = torch.randn(B, Cin, H, W, generator=g) * (1.0 + torch.arange(Cin).view(1, Cin, 1, 1) % 3)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    s, t = torch.rand(Cin, generator=g) + 0.5, 0.3 * torch.randn(Cin, generator=g)
    ref = F.conv2d(prologue64(x.double(), pro, s.double(), t.double()), w.double(), padding=1)
    assert ops.query("tag_conv3x3_wino_ok", B, H, W, Cin, Cout) == 1
    xh, wh, sd, td = nhwc(x).to(dev), w.to(dev), s.to(dev), t.to(dev)     # (kept alive: the calls below take raw pointers)
    uf, _ = wino_pack(ops, wh)
    P = ops.query("tag_conv3x3_wino_stats_rows", B, H, W, Cout)
    y = torch.full((B, H, W, Cout), float("nan"), device=dev)
    part = torch.full((P * (3 * Cout + 1),), float("nan"), device=dev)
    ws = torch.empty(ops.query("tag_conv3x3_wino_ws_bytes", B, H, W, Cin, Cout) // 4, device=dev)
    ops.call("tag_conv3x3_wino_forward", ops.ptr(xh), ops.ptr(uf), pro, ops.ptr(sd), ops.ptr(td), ops.ptr(y),
             ops.ptr(part), B, H, W, Cin, Cout, ops.ptr(ws), None)
    e = relerr(nchw(y), ref)
    assert e < 5e-6, e
    # statistics: every pixel counted exactly once, mean / invstd of the kernel's OWN output
    assert int(part[P * 3 * Cout:].sum().item()) == B * H * W
    gam, bet = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    st = ops.bn_stats(y.view(-1, Cout), gam, bet, None, None, True, partials=(P, part))
    y64 = y.double().view(-1, Cout)
    m64, v64 = y64.mean(0), y64.var(0, unbiased=False)
    assert (st.mean.double() - m64).abs().max().item() < 2e-6 * (m64.abs().max().item() + v64.sqrt().max().item())
    assert ((st.invstd.double() * torch.sqrt(v64 + 1e-5)) - 1.0).abs().max().item() < 5e-6
    # without a statistics buffer the same output, bit for bit
    y2 = torch.empty_like(y)
    ops.call("tag_conv3x3_wino_forward", ops.ptr(xh), ops.ptr(uf), pro, ops.ptr(sd), ops.ptr(td), ops.ptr(y2), None,
             B, H, W, Cin, Cout, ops.ptr(ws), None)
    assert torch.equal(y, y2)
    print(f"wino forward {B}x{H}x{W} {Cin}->{Cout} prologue {pro}: err {e:.2e}, P {P}")


@pytest.mark.parametrize("B,H,W,Cin,Cout,pro", SHAPES)
def test_wino_dgrad_and_bn_backward_sums(ops, dev, B, H, W, Cin, Cout, pro):
    """tag_conv3x3_wino_dgrad_bnsums: da = conv_transpose(dy, w) and sum g / sum g xhat of the BatchNorm+ReLU backward it flows into."""
    g = torch.Generator().manual_seed(B * H + W + Cout + 1)
    dy = torch.randn(B, Cout, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cout)
    yref = torch.randn(B, Cin, H, W, generator=g) * 2.0 + 0.2
    sc, sh = torch.rand(Cin, generator=g) + 0.5, 0.3 * torch.randn(Cin, generator=g)
    mean, invstd = 0.1 * torch.randn(Cin, generator=g), torch.rand(Cin, generator=g) + 0.5
    ref = F.conv_transpose2d(dy.double(), w.double(), padding=1)
    wh = w.to(dev)
    _, ud = wino_pack(ops, wh)
    P = ops.query("tag_conv3x3_wino_stats_rows", B, H, W, Cin)
    da = torch.full((B, H, W, Cin), float("nan"), device=dev)
    part = torch.full((P * 2 * Cin,), float("nan"), device=dev)
    ws = torch.empty(ops.query("tag_conv3x3_wino_ws_bytes", B, H, W, Cout, Cin) // 4, device=dev)
    yh, dyh = nhwc(yref).to(dev), nhwc(dy).to(dev)
    scd, shd, md, isd = sc.to(dev), sh.to(dev), mean.to(dev), invstd.to(dev)      # (kept alive: raw pointers below)
    ops.call("tag_conv3x3_wino_dgrad_bnsums", ops.ptr(dyh), ops.ptr(ud), ops.ptr(da), ops.ptr(yh), ops.ptr(scd),
             ops.ptr(shd), ops.ptr(md), ops.ptr(isd), ops.ptr(part), B, H, W, Cout, Cin, ops.ptr(ws))
    e = relerr(nchw(da), ref)
    assert e < 5e-6, e
    dg, db = torch.empty(Cin, device=dev), torch.empty(Cin, device=dev)
    wsb = ops._ws(ops.query("tag_bn_grad_from_partials_ws_bytes", P, Cin), da)
    ops.call("tag_bn_grad_from_partials", ops.ptr(part), P, Cin, ops.ptr(dg), ops.ptr(db), ops.ptr(wsb))
    # against fp64 sums over the kernel's own da with the mask the kernel's arithmetic gives (fmaf(y, scale, shift) > 0)
    # (a single-rounding fmaf keeps the sign of the exact y * scale + shift, which fp64 holds exactly)
    mask = (nhwc(yref).double() * sc.double() + sh.double()) > 0
    gg = da.double().cpu() * mask
    db64 = gg.sum(dim=(0, 1, 2))
    dg64 = (gg * (nhwc(yref).double() - mean.double()) * invstd.double()).sum(dim=(0, 1, 2))
    scale = max(db64.abs().max().item(), dg64.abs().max().item())
    assert (db.cpu().double() - db64).abs().max().item() < 5e-6 * scale
    assert (dg.cpu().double() - dg64).abs().max().item() < 5e-6 * scale
    print(f"wino dgrad {B}x{H}x{W} {Cout}->{Cin}: err {e:.2e}")


@pytest.mark.parametrize("B,H,W,Cin,Cout,pro", SHAPES + [(16, 64, 8, 256, 256, 1)])      # the last: 2048 tiles, 2 K slices
def test_wino_wgrad(ops, dev, B, H, W, Cin, Cout, pro):
    """tag_conv3x3_wino_wgrad: dw = d/dw sum(conv(prologue(x), w) * dy) against fp64 autograd -- the bound of the direct
    weight-gradient kernel's own test (1e-5 of the largest entry)."""
    g = torch.Generator().manual_seed(B * H + W + Cin + 2)
    x = torch.randn(B, Cin, H, W, generator=g)
    dy = torch.randn(B, Cout, H, W, generator=g)
    s, t = torch.rand(Cin, generator=g) + 0.5, 0.3 * torch.randn(Cin, generator=g)
    w64 = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(prologue64(x.double(), pro, s.double(), t.double()), w64, padding=1).backward(dy.double())
    xh, dyh, sd, td = nhwc(x).to(dev), nhwc(dy).to(dev), s.to(dev), t.to(dev)
    dw = torch.full((Cout, Cin, 3, 3), float("nan"), device=dev)
    ws = torch.empty(ops.query("tag_conv3x3_wino_wgrad_ws_bytes", B, H, W, Cin, Cout) // 4, device=dev)
    ops.call("tag_conv3x3_wino_wgrad", ops.ptr(xh), pro, ops.ptr(sd), ops.ptr(td), ops.ptr(dyh), ops.ptr(dw), B, H, W, Cin, Cout,
             ops.ptr(ws), None)
    e = relerr(dw, w64.grad)
    dw2 = torch.empty_like(dw)
    ops.call("tag_conv3x3_wino_wgrad", ops.ptr(xh), pro, ops.ptr(sd), ops.ptr(td), ops.ptr(dyh), ops.ptr(dw2), B, H, W, Cin, Cout,
             ops.ptr(ws), None)
    print(f"wino wgrad {B}x{H}x{W} {Cin}->{Cout} prologue {pro}: err {e:.2e}")
    assert e < 1e-5, e
    assert torch.equal(dw, dw2)                   # fixed summation order: bit-reproducible


@pytest.mark.parametrize("B,H,W,Cin,Cout,pro", SHAPES[1:-1] + [(2, 11, 8, 64, 64, 1)])     # (the two-pass reference takes C | 256)
@pytest.mark.parametrize("ph,pool", [(2, 0), (1, 0), (2, 2), (1, 3)])
def test_wino_forward_bnrelu_pool_eval(ops, dev, B, H, W, Cin, Cout, pro, ph, pool):
    """tag_conv3x3_wino_forward_bnrelu_pool_eval (inference: the output transform pools its own tile) against the fp64 chain
    pool(relu(conv(prologue(x)) * scale + shift)), windows 2x2 / 1x2 (floor), pool types avg+max / avg / max."""
    if H // ph == 0 or W < 2:
        pytest.skip("no whole pool window")
    g = torch.Generator().manual_seed(B * H + W + Cin + 7 * ph + pool)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    s, t = torch.rand(Cin, generator=g) + 0.5, 0.3 * torch.randn(Cin, generator=g)
    bs, bt = torch.rand(Cout, generator=g) + 0.5, 0.3 * torch.randn(Cout, generator=g)
    yy = F.conv2d(prologue64(x.double(), pro, s.double(), t.double()), w.double(), padding=1)
    a = torch.relu(yy * bs.double().view(1, -1, 1, 1) + bt.double().view(1, -1, 1, 1))
    avg, mx = F.avg_pool2d(a, (ph, 2)), F.max_pool2d(a, (ph, 2))
    ref = avg + mx if pool == 0 else (avg if pool == 2 else mx)
    xh, wh, sd, td, bsd, btd = nhwc(x).to(dev), w.to(dev), s.to(dev), t.to(dev), bs.to(dev), bt.to(dev)
    uf, _ = wino_pack(ops, wh)
    out = torch.full((B, H // ph, W // 2, Cout), float("nan"), device=dev)
    ws = torch.empty(ops.query("tag_conv3x3_wino_ws_bytes", B, H, W, Cin, Cout) // 4, device=dev)
    ops.call("tag_conv3x3_wino_forward_bnrelu_pool_eval", ops.ptr(xh), ops.ptr(uf), pro, ops.ptr(sd), ops.ptr(td), ops.ptr(out),
             ops.ptr(bsd), ops.ptr(btd), B, H, W, Cin, Cout, ph, 2, pool, ops.ptr(ws))
    assert torch.isfinite(out).all()
    e = relerr(nchw(out), ref)
    assert e < 5e-6, e
    # = the unfused Winograd forward followed by the pool pass of bn_pool.hip, bit for bit (same expression, same order)
    y = torch.empty(B, H, W, Cout, device=dev)
    ops.call("tag_conv3x3_wino_forward", ops.ptr(xh), ops.ptr(uf), pro, ops.ptr(sd), ops.ptr(td), ops.ptr(y), None, B, H, W, Cin,
             Cout, ops.ptr(ws), None)
    st = ops.BNStat()
    st.scale, st.shift, st.train = bsd, btd, False
    two = ops.bnact_pool(y, st, ph, 2, act=1, pool=pool)
    assert torch.equal(out, two)


@pytest.mark.parametrize("B,Hf,Wf,Cin,C,ph,train,p", [(2, 21, 16, 256, 128, 2, True, 0.2), (2, 9, 16, 128, 256, 1, True, 0.0),
                                                      (3, 11, 14, 64, 64, 2, False, 0.2), (2, 16, 32, 512, 256, 1, True, 0.2)])
def test_wino_dgrad_pool_backward_sums(ops, dev, B, Hf, Wf, Cin, C, ph, train, p):
    """tag_conv3x3_wino_dgrad_poolsums (one-read pool backward on the Winograd path: the output transform of the NEXT block's first
    conv dgrad carries the sums of the BatchNorm+ReLU+pool+dropout backward below it) against the two-pass kernels on the same dx
    (tag_bnrelu_pool_backward: pool_bwd_reduce + apply) -- dx bit-identical to the plain Winograd dgrad, dgamma / dbeta to summation
    round-off, dy accordingly.  2x2 and 1x2 windows, odd Hf / Wf-derived tile overhang, dropout on and off, eval statistics."""
    pw = 2
    H, W = Hf // ph, Wf // pw
    g = torch.Generator().manual_seed(Hf * Wf + C)
    y = torch.randn(B, C, Hf, Wf, generator=g) * (1.0 + torch.arange(C).view(1, C, 1, 1) % 5) + 0.3
    w = torch.randn(Cin, C, 3, 3, generator=g) / math.sqrt(9 * C)       # the conv that CONSUMES the pooled output: C -> Cin
    gamma, beta = torch.rand(C, generator=g) + 0.5, 0.2 * torch.randn(C, generator=g)
    rm, rv = 0.1 * torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    du = torch.randn(B, Cin, H, W, generator=g)
    yh, duh, gd = nhwc(y).to(dev), nhwc(du).to(dev), gamma.to(dev)
    st = ops.bn_stats(yh.view(-1, C), gd, beta.to(dev), rm.clone().to(dev), rv.clone().to(dev), train)
    seed = 4242
    _, ud = wino_pack(ops, w.to(dev))                                   # dgrad planes: Cin -> C
    P = ops.query("tag_conv3x3_wino_stats_rows", B, H, W, C)
    dx = torch.full((B, H, W, C), float("nan"), device=dev)
    part = torch.full((P * 2 * C,), float("nan"), device=dev)
    ws = torch.empty(ops.query("tag_conv3x3_wino_ws_bytes", B, H, W, Cin, C) // 4, device=dev)
    ops.call("tag_conv3x3_wino_dgrad_poolsums", ops.ptr(duh), ops.ptr(ud), ops.ptr(dx), ops.ptr(yh), ops.ptr(st.scale), ops.ptr(st.shift),
             ops.ptr(st.mean), ops.ptr(st.invstd), ops.ptr(part), B, H, W, Cin, C, Hf, Wf, ph, pw, 0, float(p), seed, ops.ptr(ws))
    dx_plain = torch.empty_like(dx)
    ops.call("tag_conv3x3_wino_forward", ops.ptr(duh), ops.ptr(ud), 0, None, None, ops.ptr(dx_plain), None, B, H, W, Cin, C, ops.ptr(ws),
             None)
    assert torch.equal(dx, dx_plain)
    dy, dg, db = ops.bnrelu_pool_backward(yh, st, gd, dx, ph, pw, p, seed, partials=(P, part))
    dy2, dg2, db2 = ops.bnrelu_pool_backward(yh, st, gd, dx, ph, pw, p, seed)
    assert relerr(dg, dg2) < 2e-6 and relerr(db, db2) < 2e-6
    assert torch.equal(dy, dy2) if not train else relerr(dy, dy2) < 2e-6
    # ... and against an fp64 reduction over the kernel's OWN dx (round 6; not another HIP kernel): dz = the gradient reaching
    # a = relu(bn(y)) through dropout and avg + max pool, with the mask from the exact y * scale + shift (an fp64 product of fp32
    # factors is exact, so its sign is the kernels' fmaf's) and the FIRST maximum of each window; dbeta = sum dz, dgamma = sum dz xhat
    sc, sh = st.scale.double().cpu().view(1, C, 1, 1), st.shift.double().cpu().view(1, C, 1, 1)
    mu, isd = st.mean.double().cpu().view(1, C, 1, 1), st.invstd.double().cpu().view(1, C, 1, 1)
    y64 = y.double()
    a = (y64.float().double() * sc + sh)[:, :, :H * ph, :W * pw]
    gp = nchw(dx).double().cpu()
    if p > 0:
        keep = ops.dropout_mask(seed, (B, H, W, C), p, dev, pooled=True).cpu().permute(0, 3, 1, 2).double()
        gp = gp * keep * float(torch.tensor(1.0 / (1.0 - p), dtype=torch.float32))
    win = a.reshape(B, C, H, ph, W, pw).permute(0, 1, 2, 4, 3, 5).reshape(B, C, H, W, ph * pw)
    first = torch.zeros_like(win)
    first.scatter_(-1, torch.relu(win.float()).argmax(-1, keepdim=True), 1.0)
    dz = (win > 0) * (gp.unsqueeze(-1) * (1.0 / (ph * pw)) + gp.unsqueeze(-1) * first)
    xh = ((y64.float().double() - mu) * isd)[:, :, :H * ph, :W * pw].reshape(B, C, H, ph, W, pw).permute(0, 1, 2, 4, 3, 5).reshape(B, C, H, W, ph * pw)
    db64, dg64 = dz.sum(dim=(0, 2, 3, 4)), (dz * xh).sum(dim=(0, 2, 3, 4))
    norm = max(db64.abs().max().item(), dg64.abs().max().item())
    e_b, e_g = (db.double().cpu() - db64).abs().max().item() / norm, (dg.double().cpu() - dg64).abs().max().item() / norm
    print(f"wino pool sums {B}x{Hf}x{Wf} {C}<-{Cin} window {ph}x2 train={train} p={p}: vs fp64 dbeta {e_b:.1e} dgamma {e_g:.1e}")
    assert e_b < 5e-6 and e_g < 5e-6


def test_plane_form_wgrad_reuses_the_forward_planes(ops, dev):
    """Channel counts the fused kernels do not take (Cout % 64 != 0) keep round 5's plane form: its forward launch can leave the
    transformed input in a caller's buffer (v_keep) and the weight gradient multiplies those planes (v_saved) instead of
    transforming x again -- bit-identical dw.  The fused shapes report that they have no planes to keep."""
    B, H, W, C = 8, 64, 8, 32                     # 1024 tiles
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, H, W, C, generator=g).to(dev)
    dy = torch.randn(B, H, W, C, generator=g).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C)).to(dev)
    s, t = (torch.rand(C, generator=g) + 0.5).to(dev), (0.3 * torch.randn(C, generator=g)).to(dev)
    assert ops.query("tag_conv3x3_wino_wgrad_can_reuse_v", B, H, W, C, C) == 1
    assert ops.query("tag_conv3x3_wino_wgrad_can_reuse_v", B, H, W, 256, 256) == 0
    uf, _ = wino_pack(ops, w)
    T = B * (H // 2) * (W // 2)
    vkeep = torch.empty(16 * T * C, device=dev)
    y = torch.empty(B, H, W, C, device=dev)
    ws = torch.empty(ops.query("tag_conv3x3_wino_ws_bytes", B, H, W, C, C) // 4, device=dev)
    ops.call("tag_conv3x3_wino_forward", ops.ptr(x), ops.ptr(uf), 1, ops.ptr(s), ops.ptr(t), ops.ptr(y), None, B, H, W, C, C,
             ops.ptr(ws), ops.ptr(vkeep))
    wsw = torch.empty(ops.query("tag_conv3x3_wino_wgrad_ws_bytes", B, H, W, C, C) // 4, device=dev)
    dw_plain, dw_kept = torch.empty(C, C, 3, 3, device=dev), torch.empty(C, C, 3, 3, device=dev)
    ops.call("tag_conv3x3_wino_wgrad", ops.ptr(x), 1, ops.ptr(s), ops.ptr(t), ops.ptr(dy), ops.ptr(dw_plain), B, H, W, C, C,
             ops.ptr(wsw), None)
    ops.call("tag_conv3x3_wino_wgrad", None, 1, ops.ptr(s), ops.ptr(t), ops.ptr(dy), ops.ptr(dw_kept), B, H, W, C, C,
             ops.ptr(wsw), ops.ptr(vkeep))
    assert torch.equal(dw_kept, dw_plain)
    ref = F.conv2d(torch.relu(nchw(x).double() * s.double().view(1, -1, 1, 1) + t.double().view(1, -1, 1, 1)).transpose(0, 1),
                   nchw(dy).double().transpose(0, 1), padding=1).transpose(0, 1)
    assert relerr(dw_plain, ref) < 1e-5


def test_wino_dispatch_rule(ops, dev, monkeypatch):
    """ops.conv3x3_stats / conv3x3_dgrad_bnrelu_backward take the Winograd form exactly for fp32 training launches on 8- ... 64-wide
    images with both channel counts >= WINO_MIN_C and at least WINO_MIN_WORK tiles x output channels; everything else keeps the direct kernel; the
    two forms agree to both kernels' rounding, and a run is bit-reproducible."""
    B, H, W, C = 4, 64, 16, 256                   # 1024 tiles
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, H, W, C, generator=g).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C)).to(dev)
    monkeypatch.setattr(ops, "WINO_MIN_WORK", 1024 * 256)
    wf, wd = ops.pack_conv_weight(w, W=W)
    assert hasattr(wf, "wino_u") and hasattr(wd, "wino_u")
    n0 = ops.WINO_LAUNCHES
    y_w, part_w = ops.conv3x3_stats(x, wf, C)
    y_w2, _ = ops.conv3x3_stats(x, wf, C)
    assert ops.WINO_LAUNCHES == n0 + 2 and torch.equal(y_w, y_w2)
    y_e = ops.conv3x3(x, wf, C)                   # no statistics wanted (inference): direct
    assert ops.WINO_LAUNCHES == n0 + 2
    monkeypatch.setattr(ops, "WINO_MIN_WORK", 1024 * 256 + 1)
    y_d, part_d = ops.conv3x3_stats(x, wf, C)
    assert ops.WINO_LAUNCHES == n0 + 2 and torch.equal(y_d, y_e)
    assert relerr(y_w, y_d) < 5e-6
    assert part_w[0] == ops.query("tag_conv3x3_wino_stats_rows", B, H, W, C) and part_d[0] == ops.query("tag_conv3x3_stats_rows", B, H, W, C)
    dyv = torch.randn(B, H, W, C, generator=g).to(dev)
    dw_d = ops.conv3x3_wgrad(x, dyv)
    assert ops.WINO_LAUNCHES == n0 + 2
    monkeypatch.setattr(ops, "WINO_MIN_WORK", 1)
    dw_w = ops.conv3x3_wgrad(x, dyv)
    assert ops.WINO_LAUNCHES == n0 + 3 and relerr(dw_w, dw_d) < 1e-5
    monkeypatch.setattr(ops, "CONV_WINOGRAD", False)
    ops.conv3x3_stats(x, wf, C)
    ops.conv3x3_wgrad(x, dyv)
    assert ops.WINO_LAUNCHES == n0 + 3
    monkeypatch.setattr(ops, "CONV_WINOGRAD", True)
    wf64, _ = ops.pack_conv_weight(w[:128, :128].contiguous(), W=W)     # 128 -> 128: taken since the fused kernels (round 6)
    assert hasattr(wf64, "wino_u")
    wf32, _ = ops.pack_conv_weight(w, W=32)                              # 32-wide image: taken
    assert hasattr(wf32, "wino_u")
    wf4, _ = ops.pack_conv_weight(w, W=4)                                # the 4-wide images of the CrnnEncoder: direct
    assert not hasattr(wf4, "wino_u")
    wfs, _ = ops.pack_conv_weight(w[:32, :32].contiguous(), W=W)         # 32 channels: direct
    assert not hasattr(wfs, "wino_u")


def test_inference_batch_cuts_give_the_same_rows(ops, dev, monkeypatch):
    """The inference launches are cut into batch slices that keep every tensor inside the fused kernels' 32-bit descriptor range
    (ops.WINO_MAX_BYTES; at 30 s x 256 clips the 64-channel tensors are 4 x that range).  Any cut gives the same rows bit for bit
    -- every tile is computed independently of the others -- for the plain forward and for the forward with the BatchNorm / ReLU /
    pool epilogue; a limit that forces ragged slices (5 clips as 2 + 2 + 1) is compared with the uncut launch."""
    B, H, W, Cin, Cout = 5, 21, 16, 64, 128
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dev)
    s, t = (torch.rand(Cin, generator=g) + 0.5).to(dev), (0.3 * torch.randn(Cin, generator=g)).to(dev)
    st = ops.BNStat()
    st.scale, st.shift, st.train = (torch.rand(Cout, generator=g) + 0.5).to(dev), (0.3 * torch.randn(Cout, generator=g)).to(dev), False
    wf, _ = ops.pack_conv_weight(w, want_dgrad=False, W=W)
    assert hasattr(wf, "wino_u")
    n0 = ops.WINO_LAUNCHES
    y1, _ = ops.conv3x3_stats(x, wf, Cout, 1, s, t, want_stats=False, inference=True)
    p1 = ops.conv3x3_bnrelu_pool_eval(x, wf, Cout, st, 2, 2, 1, s, t)
    assert ops.WINO_LAUNCHES == n0 + 2
    per_clip = H * W * max(Cin, Cout) * 4
    monkeypatch.setattr(ops, "WINO_MAX_BYTES", 2 * per_clip + 17)
    assert ops._batch_chunks(B, per_clip) == [(0, 2), (2, 4), (4, 5)]
    y2, _ = ops.conv3x3_stats(x, wf, Cout, 1, s, t, want_stats=False, inference=True)
    p2 = ops.conv3x3_bnrelu_pool_eval(x, wf, Cout, st, 2, 2, 1, s, t)
    assert torch.equal(y1, y2) and torch.equal(p1, p2)
    monkeypatch.setattr(ops, "WINO_MAX_BYTES", 1)            # below one clip: one clip per launch, never zero
    y3, _ = ops.conv3x3_stats(x, wf, Cout, 1, s, t, want_stats=False, inference=True)
    assert torch.equal(y1, y3)
    ref = F.conv2d(prologue64(nchw(x.cpu()).double(), 1, s.cpu().double(), t.cpu().double()), w.cpu().double(), padding=1)
    assert relerr(nchw(y1), ref) < 5e-6
