#!/usr/bin/env python3
"""Is co-running an HBM-bound pass with an MFMA-bound kernel positive-sum on this part?  (bf16 mode: the weight-gradient side
stream.)  Two HIP streams: A = N launches of one bf16 weight-gradient conv, B = M launches of one BatchNorm+ReLU backward apply
pass (3 tensors of the block-1 size streamed through HBM).  Reports the wall time of A alone, B alone, and A beside B --
max(A, B) would be perfect overlap, A + B none.  TAG_WGRAD_WGS sets how many workgroups the weight gradient launches (512 = one
full residency round of 2 per CU: its workgroups live as long as the kernel and hold 2 x 200-240 of a SIMD's 512 VGPRs).

    python tools/overlap_probe.py [layer: 16|32|8] [prologue 0|1]
"""
import sys
import time

import torch

sys.path.insert(0, ".")
from texttoaudiogrounding_amd import ops  # noqa: E402
from texttoaudiogrounding_amd.ops import BNStat  # noqa: E402


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    pro = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = torch.device("cuda:0")
    B = 64
    H, Cin, Cout = {8: (250, 256, 512), 16: (250, 128, 256), 32: (500, 64, 128), 64: (1001, 64, 64)}[W]
    if pro:
        Cin = Cout
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g).to(torch.bfloat16)
    dy = torch.randn(B, H, W, Cout, device=dev, generator=g).to(torch.bfloat16)
    sc = torch.rand(Cin, device=dev, generator=g) + 0.5
    sh = torch.randn(Cin, device=dev, generator=g)
    dw = torch.empty(Cout, Cin, 3, 3, device=dev)
    # the HBM-bound pass: block-1 size
    C = 64
    rows = B * 1001 * 64
    y = torch.randn(rows, C, device=dev, generator=g).to(torch.bfloat16)
    da = torch.randn(rows, C, device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty_like(da)
    st = BNStat()
    st.mean, st.invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    st.scale, st.shift, st.train = torch.ones(C, device=dev), torch.zeros(C, device=dev), True
    gamma, dg, db = torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)

    def wgrad():
        ops.conv3x3_wgrad(x, dy, prologue=pro, scale=sc if pro else None, shift=sh if pro else None, out=dw)

    def apply():
        ops.call("tag_bnrelu_backward_apply_bf16", ops.ptr(y), ops.ptr(st.scale), ops.ptr(st.shift), ops.ptr(st.mean),
                 ops.ptr(st.invstd), ops.ptr(gamma), ops.ptr(da), ops.ptr(out), ops.ptr(dg), ops.ptr(db), rows, C, 1)

    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def run(na, nb, reps=5):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(s1):
                for _ in range(na):
                    wgrad()
            with torch.cuda.stream(s2):
                for _ in range(nb):
                    apply()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3

    for _ in range(3):
        wgrad(); apply()
    torch.cuda.synchronize()
    ta1 = run(20, 0) / 20
    tb1 = run(0, 20) / 20
    nb = max(1, round(20 * ta1 / tb1))
    ta, tb, tab = run(20, 0), run(0, nb), run(20, nb)
    print(f"W={W} {Cin}->{Cout} prologue {pro}: wgrad {ta1 * 1e3:.1f} us/launch, apply {tb1 * 1e3:.1f} us/launch")
    print(f"  20 wgrads alone {ta:.3f} ms | {nb} applies alone {tb:.3f} ms | together {tab:.3f} ms "
          f"(max {max(ta, tb):.3f}, sum {ta + tb:.3f}): overlap efficiency {(ta + tb - tab) / min(ta, tb):.2f}")


if __name__ == "__main__":
    main()
