"""Evidence for the power ceiling of the bf16 kernels, written to stdout for profiles/rNN_mfma_peak.txt
and profiles/rNN_conv_rows_power.txt:

  python tools/power_evidence.py mfma   -> the register-resident MFMA loop of csrc/probe.hip (no LDS, no memory) for ~70 ms per
                                           case: bf16 32x32x16 and fp32 32x32x2, constant vs random operands; TFLOP/s, the shader
                                           clock the kernel itself measured (s_memtime / s_memrealtime), hwmon power and clock
  python tools/power_evidence.py conv   -> the bf16 64->64 layer of block 1 (B = 64, 1001 x 64) on zeros / constants / random
                                           data, each for >= 0.3 s under the hwmon sampler: us per launch, TFLOP/s, W, MHz
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from texttoaudiogrounding_amd import lib, ops
from texttoaudiogrounding_amd.utils.telemetry import BoardSampler, mfma_probe

dev = torch.device("cuda:0")


def header():
    pr = torch.cuda.get_device_properties(0)
    print(f"# {time.strftime('%Y-%m-%dT%H:%MZ', time.gmtime())}  {pr.name}, {pr.multi_processor_count} CUs; "
          f"libtag_hip build id {lib.build_id()[:12]}")
    s = BoardSampler(0)
    print(f"# hwmon sources: power={s.power_file} sclk={s.freq_file or s.dpm_file}")
    return s


def mfma():
    s = header()
    print(f"{'case':<16} {'ms':>8} {'TFLOP/s':>9} {'sclk MHz (in-kernel)':>22} {'avg W':>8} {'max W':>8} {'hwmon MHz':>10}")
    for rep in range(2):                                    # the second round starts from a warm part
        for kind in ("bf16_constant", "bf16_random", "f32_constant", "f32_random"):
            s.start()
            r = mfma_probe(kind, 70.0, 0)
            b = s.stop()
            print(f"{kind:<16} {r['ms']:8.2f} {r['TFLOP/s']:9.1f} {str(r['sclk_MHz']):>22} {str(b['avg_W']):>8} {str(b['max_W']):>8} "
                  f"{str(b['avg_sclk_MHz']):>10}", flush=True)
    # a long random-operand run: the clock after the power controller has settled
    s.start()
    r = mfma_probe("bf16_random", 400.0, 0)
    b = s.stop()
    print(f"{'bf16_random 0.4s':<16} {r['ms']:8.2f} {r['TFLOP/s']:9.1f} {str(r['sclk_MHz']):>22} {str(b['avg_W']):>8} {str(b['max_W']):>8} "
          f"{str(b['avg_sclk_MHz']):>10}")


def conv():
    s = header()
    ops.CONV_MATH, ops.ACT_DTYPE = "bf16", "bf16"
    B, H, W, Cin, Cout = 64, 1001, 64, 64, 64
    flop = 2.0 * B * H * W * 9 * Cin * Cout
    print(f"# bf16 conv 64->64 at {B} x {H} x {W} (block 1, conv_rows.hip), forward with BatchNorm statistics; "
          f"{flop / 1e9:.1f} GFLOP per launch")
    print(f"{'operands':<22} {'us/launch':>10} {'TFLOP/s':>9} {'avg W':>8} {'max W':>8} {'hwmon MHz':>10}")

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.start()
        n = 0
        e0.record()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3 or n < 20:
            for _ in range(20):
                fn()
            n += 20
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3, s.stop()

    cases = (("zero x, zero w", 0.0, 0.0), ("zero x, random w", 0.0, 0.05), ("constant x, constant w", None, None),
             ("random x, random w", 1.0, 0.05))
    for name, xs, wsc in cases:
        if xs is None:
            x = torch.full((B, H, W, Cin), 0.5, device=dev).bfloat16()
            w = torch.full((Cout, Cin, 3, 3), 0.01, device=dev)
        else:
            x = (torch.randn(B, H, W, Cin, device=dev) * xs).bfloat16()
            w = torch.randn(Cout, Cin, 3, 3, device=dev) * wsc
        wf, _ = ops.pack_conv_weight(w, W=W)
        us, b = timed(lambda: ops.conv3x3_stats(x, wf, Cout, 0, None, None, want_stats=True))
        print(f"{name:<22} {us:10.1f} {flop / us / 1e6:9.1f} {str(b['avg_W']):>8} {str(b['max_W']):>8} {str(b['avg_sclk_MHz']):>10}",
              flush=True)
    x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
    y = torch.empty_like(x)
    us, b = timed(lambda: y.copy_(x))
    print(f"{'copy_ of the same bytes':<22} {us:10.1f} {'-':>9} {str(b['avg_W']):>8} {str(b['max_W']):>8} {str(b['avg_sclk_MHz']):>10}")


if __name__ == "__main__":
    {"mfma": mfma, "conv": conv}[sys.argv[1] if len(sys.argv) > 1 else "mfma"]()
