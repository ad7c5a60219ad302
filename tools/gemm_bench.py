#!/usr/bin/env python3
"""Every dense GEMM of one training step of the bench workload (B = 64, 10 s clips): the calls are recorded from a live
step, then each distinct shape is timed alone (20 repetitions, HIP events) on tag_gemm and tag_gemm_bf16.

    python tools/gemm_bench.py [B]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from texttoaudiogrounding_amd import ops  # noqa: E402
from texttoaudiogrounding_amd.models import audio_encoder, audio_text_model, match, text_encoder  # noqa: E402
from texttoaudiogrounding_amd.runner import StrongRunner  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = audio_text_model.BiEncoder(audio_encoder.Cnn8Rnn(32000), text_encoder.EmbeddingAgg(5221, 512), match.DotProduct(), 512)
runner = StrongRunner(model, lr=1e-3, max_grad_norm=1.0, device=str(dev))
batch = bench.synthetic_batch(B, 320000, 1234, dev)
runner.train_step(batch)
calls = []
orig = ops.gemm


def rec(A, Bm, M, N, K, transA=False, transB=False, lda=None, ldb=None, out=None, ldc=None, bias=None, act=0, accumulate=False):
    calls.append((M, N, K, bool(transA), bool(transB), bias is not None, act, bool(accumulate)))
    return orig(A, Bm, M, N, K, transA, transB, lda, ldb, out, ldc, bias, act, accumulate)


from texttoaudiogrounding_amd import dispatch, functions  # noqa: E402
dispatch.gemm = functions.gemm = rec           # (patched where the launches look it up)
runner.train_step(batch)
dispatch.gemm = functions.gemm = orig
torch.cuda.synchronize()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {"tag_gemm": 0.0, "tag_gemm_bf16": 0.0}
print(f"{len(calls)} GEMM calls per step")
for (M, N, K, tA, tB, hb, act, acc) in calls:
    A = torch.randn((K, M) if tA else (M, K), device=dev)
    Bm = torch.randn((N, K) if tB else (K, N), device=dev)
    C = torch.zeros(M, N, device=dev)
    bias = torch.randn(N, device=dev) if hb else None
    nws = ops.query("tag_gemm_ws_bytes", M, N, K) if tA else 0
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
    row = []
    for fn in ("tag_gemm", "tag_gemm_bf16"):
        us = timeit(lambda: ops.call(fn, ops.ptr(A), M if tA else K, int(tA), ops.ptr(Bm), K if tB else N, int(tB), ops.ptr(C), N, M, N,
                                     K, ops.ptr(bias), act, int(acc), ops.ptr(ws) if nws else None))
        tot[fn] += us
        row.append(f"{fn[4:]:9s} {us:7.1f} us {2.0 * M * N * K / us * 1e-6:6.1f} TFLOP/s")
    print(f"M={M:5d} N={N:5d} K={K:5d} transA={int(tA)} transB={int(tB)} bias={int(hb)} act={act} acc={int(acc)} splits_ws={nws >> 20:4d} MiB | " + " | ".join(row))
print("sum per step: " + ", ".join(f"{k} {v / 1e3:.3f} ms" for k, v in tot.items()))
ops.check_async_errors()
