import os, sys
sys.path.insert(0, "/root/repo")
import torch
from texttoaudiogrounding_amd import ops
from texttoaudiogrounding_amd.lib import query
ops.CONV_MATH, ops.ACT_DTYPE = "bf16", "bf16"
dev = torch.device("cuda:0")
B, H, W, Cin, Cout = 64, 1001, 64, 64, 64
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, xs, wsc in (("random", 1.0, 0.05), ("zero x", 0.0, 0.05), ("zero x, zero w", 0.0, 0.0), ("const", None, None)):
    if xs is None:
        x = torch.full((B, H, W, Cin), 0.5, device=dev).bfloat16(); w = torch.full((Cout, Cin, 3, 3), 0.01, device=dev)
    else:
        x = (torch.randn(B, H, W, Cin, device=dev) * xs).bfloat16(); w = torch.randn(Cout, Cin, 3, 3, device=dev) * wsc
    wf, _ = ops.pack_conv_weight(w, W=W)
    s, t = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
    for on in (1, 0):
        query("tag_conv_rows_enable", on)
        us0 = timeit(lambda: ops.conv3x3_stats(x, wf, Cout, 0, None, None, want_stats=True))
        us1 = timeit(lambda: ops.conv3x3_stats(x, wf, Cout, 1, s, t, want_stats=True))
        print(f"{name:16s} {'rows' if on else 'tile'}: pro0 {us0:7.1f} us  pro1 {us1:7.1f} us", flush=True)
# pure copy of the same bytes for reference
x = torch.randn(B, H, W, Cin, device=dev).bfloat16(); y = torch.empty_like(x)
print("copy_ of the same tensor:", timeit(lambda: y.copy_(x)), "us")
