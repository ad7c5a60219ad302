#!/usr/bin/env python3
"""Winograd F(2x2,3x3) conv (csrc/conv_wino.hip) against the direct halo-tile kernel on the deep Cnn8Rnn layer shapes at B = 64:
time per launch (HIP events, 10 repetitions), error of both against an fp64 convolution of a sub-batch, and the fused
statistics / BatchNorm-backward sums against the direct kernel's.

    python tools/wino_bench.py [B]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from texttoaudiogrounding_amd import ops  # noqa: E402
from texttoaudiogrounding_amd.ops import call, ptr, query  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def ref64(x, w, scale, shift, nb=2):
    xx = x[:nb].double()
    if scale is not None:
        xx = torch.relu(xx * scale.double() + shift.double())
    return torch.nn.functional.conv2d(xx.permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)


shapes = [(250, 8, 512, 512), (250, 8, 256, 512), (250, 16, 256, 256), (250, 16, 128, 256), (500, 32, 128, 128), (500, 32, 64, 128),
          (1001, 64, 64, 64)]
if len(sys.argv) > 2:                      # one shape only (under rocprofv3)
    shapes = [shapes[int(sys.argv[2])]]
for (H, W, Cin, Cout) in shapes:
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * (2.0 / (9 * Cin)) ** 0.5
    scale = torch.rand(Cin, device=dev) + 0.5
    shift = torch.randn(Cin, device=dev) * 0.3
    pf, pd = ops.pack_conv_weight(w, True, W)
    uf = torch.empty(16, Cin, Cout, device=dev)
    ud = torch.empty(16, Cout, Cin, device=dev)
    call("tag_pack_conv_weight_wino", ptr(w), ptr(uf), ptr(ud), Cin, Cout)
    ws = torch.empty(query("tag_conv3x3_wino_ws_bytes", B, H, W, max(Cin, Cout), max(Cin, Cout)) // 4 + 16, device=dev)
    flops = 2.0 * B * H * W * 9 * Cin * Cout
    # ---- forward with prologue 1 and statistics
    Pd = query("tag_conv3x3_stats_rows", B, H, W, Cout)
    Pw = query("tag_conv3x3_wino_stats_rows", B, H, W, Cout)
    yd = torch.empty(B, H, W, Cout, device=dev)
    yw = torch.empty(B, H, W, Cout, device=dev)
    sd = torch.zeros(Pd * (3 * Cout + 1), device=dev)
    sw = torch.zeros(Pw * (3 * Cout + 1), device=dev)
    f_d = lambda: call("tag_conv3x3_forward", ptr(x), ptr(pf), 1, ptr(scale), ptr(shift), ptr(yd), ptr(sd), B, H, W, Cin, Cout)
    f_w = lambda: call("tag_conv3x3_wino_forward", ptr(x), ptr(uf), 1, ptr(scale), ptr(shift), ptr(yw), ptr(sw), B, H, W, Cin, Cout,
                       ptr(ws), None)
    td, tw = timeit(f_d), timeit(f_w)
    r = ref64(x, w, scale, shift)
    rng = r.abs().max().item()
    ed = ((yd[:2].double() - r).abs().max().item() / rng, ((yd[:2].double() - r).pow(2).mean().sqrt().item()) / rng)
    ew = ((yw[:2].double() - r).abs().max().item() / rng, ((yw[:2].double() - r).pow(2).mean().sqrt().item()) / rng)
    g = torch.ones(Cout, device=dev)
    bt = torch.zeros(Cout, device=dev)
    std = ops.bn_stats(yd.view(-1, Cout), g, bt, None, None, True, partials=(Pd, sd))
    stw = ops.bn_stats(yw.view(-1, Cout), g, bt, None, None, True, partials=(Pw, sw))
    m64 = yw.double().mean(dim=(0, 1, 2))
    v64 = yw.double().var(dim=(0, 1, 2), unbiased=False)
    em = (stw.mean.double() - m64).abs().max().item()
    ei = ((stw.invstd.double() - 1.0 / torch.sqrt(v64 + 1e-5)).abs() * torch.sqrt(v64 + 1e-5)).max().item()
    print(f"fwd  {H}x{W} {Cin}->{Cout}: direct {td:.3f} ms ({flops / td * 1e-9:.1f} TF/s)  wino {tw:.3f} ms ({flops / tw * 1e-9:.1f} eff)  "
          f"x{td / tw:.2f} | err/range max,rms direct {ed[0]:.2e},{ed[1]:.2e} wino {ew[0]:.2e},{ew[1]:.2e} | stats mean {em:.1e} invstd rel {ei:.1e}")
    # ---- dgrad (Cout -> Cin) with BatchNorm-backward sums against yref
    dy = torch.randn(B, H, W, Cout, device=dev)
    yref = torch.randn(B, H, W, Cin, device=dev)
    mean = torch.randn(Cin, device=dev) * 0.1
    invstd = torch.rand(Cin, device=dev) + 0.5
    Pd2 = query("tag_conv3x3_stats_rows", B, H, W, Cin)
    Pw2 = query("tag_conv3x3_wino_stats_rows", B, H, W, Cin)
    dad = torch.empty(B, H, W, Cin, device=dev)
    daw = torch.empty(B, H, W, Cin, device=dev)
    bd = torch.zeros(Pd2 * 2 * Cin, device=dev)
    bw = torch.zeros(Pw2 * 2 * Cin, device=dev)
    g_d = lambda: call("tag_conv3x3_dgrad_bnsums", ptr(dy), ptr(pd), ptr(dad), ptr(yref), ptr(scale), ptr(shift), ptr(mean),
                       ptr(invstd), ptr(bd), B, H, W, Cout, Cin)
    g_w = lambda: call("tag_conv3x3_wino_dgrad_bnsums", ptr(dy), ptr(ud), ptr(daw), ptr(yref), ptr(scale), ptr(shift), ptr(mean),
                       ptr(invstd), ptr(bw), B, H, W, Cout, Cin, ptr(ws))
    td, tw = timeit(g_d), timeit(g_w)
    r = torch.nn.functional.conv_transpose2d(dy[:2].double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)
    rng = r.abs().max().item()
    ed = (dad[:2].double() - r).abs().max().item() / rng
    ew = (daw[:2].double() - r).abs().max().item() / rng
    outs = []
    for (Pn, buf) in ((Pd2, bd), (Pw2, bw)):
        dg, db = torch.empty(Cin, device=dev), torch.empty(Cin, device=dev)
        wsb = ops._ws(query("tag_bn_grad_from_partials_ws_bytes", Pn, Cin), dy)
        call("tag_bn_grad_from_partials", ptr(buf), Pn, Cin, ptr(dg), ptr(db), ptr(wsb))
        outs.append((dg, db))
    mask = (yref.double() * scale.double() + shift.double()) > 0
    gg = daw.double() * mask
    db64 = gg.sum(dim=(0, 1, 2))
    dg64 = (gg * (yref.double() - mean.double()) * invstd.double()).sum(dim=(0, 1, 2))
    sc = max(db64.abs().max().item(), dg64.abs().max().item())
    print(f"dgrad {H}x{W} {Cout}->{Cin}: direct {td:.3f} ms  wino {tw:.3f} ms  x{td / tw:.2f} | err/range max direct {ed:.2e} wino {ew:.2e} | "
          f"sums vs fp64 of wino's own da: dbeta {(outs[1][1].double() - db64).abs().max().item() / sc:.1e} dgamma "
          f"{(outs[1][0].double() - dg64).abs().max().item() / sc:.1e}; direct-vs-wino dbeta {(outs[0][1] - outs[1][1]).abs().max().item() / sc:.1e}")
    # ---- weight gradient
    dwd = torch.empty(Cout, Cin, 3, 3, device=dev)
    dww = torch.empty(Cout, Cin, 3, 3, device=dev)
    wsd = ops._ws(query("tag_conv3x3_wgrad_ws_bytes", B, H, W, Cin, Cout), x)
    wsw = ops._ws(query("tag_conv3x3_wino_wgrad_ws_bytes", B, H, W, Cin, Cout), x)
    w_d = lambda: call("tag_conv3x3_wgrad", ptr(x), 1, ptr(scale), ptr(shift), ptr(dy), ptr(dwd), B, H, W, Cin, Cout, ptr(wsd))
    w_w = lambda: call("tag_conv3x3_wino_wgrad", ptr(x), 1, ptr(scale), ptr(shift), ptr(dy), ptr(dww), B, H, W, Cin, Cout, ptr(wsw), None)
    td, tw = timeit(w_d), timeit(w_w)
    print(f"wgrad {H}x{W} {Cin}->{Cout}: direct {td:.3f} ms  wino {tw:.3f} ms  x{td / tw:.2f} | direct-vs-wino "
          f"{(dwd - dww).abs().max().item() / dwd.abs().max().item():.2e}")
ops.check_async_errors()
