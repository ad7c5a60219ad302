#!/usr/bin/env python3
"""Does a real MFMA convolution keep its rate while packed-fp32 VALU waves run on the same CUs?  (GPU box)
    python tools/hybrid_probe.py              (3 workgroups of the conv per CU: no registers left for anything else)
    TAG_HALO_LDS_PAD=24000 python tools/hybrid_probe.py   (2 per CU: 176 VGPRs per lane and SIMD are free)
Stream A: the 512->512 conv at 250 x 8 (B = 64, 604 GFLOP), REPS launches; stream B: csrc/probe.hip's v_pk_fma_f32 loop, one
4-wave workgroup per CU, sized to outlast them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texttoaudiogrounding_amd import ops
from texttoaudiogrounding_amd.lib import call, query

dev = torch.device("cuda:0")
B, H, W, Cin, Cout, REPS = 64, 250, 8, 512, 512, 6
x = torch.randn(B, H, W, Cin, device=dev)
w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
wf, _ = ops.pack_conv_weight(w, W=W)
flop = 2.0 * B * H * W * 9 * Cin * Cout
clocks = torch.zeros(3, dtype=torch.int64, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ncu = query("tag_device_cu_count")


def conv_ms(with_valu, valu_iters=0, wgs_per_cu=1):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    v0, v1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if with_valu:
        with torch.cuda.stream(sb):
            v0.record()
            call("tag_valu_probe", valu_iters, ncu * wgs_per_cu, 7, clocks.data_ptr())
            v1.record()
    with torch.cuda.stream(sa):
        e0.record()
        for _ in range(REPS):
            ops.conv3x3(x, wf, Cout)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS, (v0.elapsed_time(v1) if with_valu else None)


for _ in range(2):
    conv_ms(False)
alone, _ = conv_ms(False)
print(f"TAG_HALO_LDS_PAD={os.environ.get('TAG_HALO_LDS_PAD', '0')}: conv alone {alone:.3f} ms = {flop / alone / 1e9:.1f} TFLOP/s")
# VALU alone: calibrate iterations to ~ REPS * alone
it = 20000
call("tag_valu_probe", it, ncu, 7, clocks.data_ptr()); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); call("tag_valu_probe", it, ncu, 7, clocks.data_ptr()); e1.record(); torch.cuda.synchronize()
vms = e0.elapsed_time(e1)
vflop = lambda iters, wgs: wgs * 4.0 * iters * 32 * 256
print(f"VALU loop alone (1 workgroup/CU): {vflop(it, ncu) / vms / 1e9:.1f} TFLOP/s")
for wpc in (1, 2):
    iters = int(it * (REPS * alone * 1.3) / vms / wpc)
    ms, vms2 = conv_ms(True, iters, wpc)
    c = clocks.cpu().tolist()
    print(f"  beside {wpc} VALU workgroup(s)/CU: conv {ms:.3f} ms = {flop / ms / 1e9:.1f} TFLOP/s ({ms / alone:.3f} x alone); "
          f"VALU {vflop(iters, ncu * wpc) / vms2 / 1e9:.1f} TFLOP/s over its {vms2:.1f} ms (conv ran {ms * REPS:.1f} ms of them), "
          f"sclk {c[0] / max(c[1], 1) * 100:.0f} MHz")
