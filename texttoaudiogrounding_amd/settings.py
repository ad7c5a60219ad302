"""Switches of the HIP path, in ONE module so that every consumer reads the same object: `settings.NAME`, or -- the spelling the
tests, bench.py and the tools use -- `ops.NAME`, which ops.py forwards here for reading AND assignment.  Defaults come from the
TAG_* environment variables at import time; everything may be re-assigned at run time (the parity tests do).
"""
import os


def _env_int(name, default=0):
    """An integer environment switch parsed defensively: a malformed value is reported with its name, not as a bare ValueError at
    import time."""
    raw = os.environ.get(name, "")
    if raw.strip() == "":
        return default
    try:
        return int(raw)
    except ValueError:
        raise RuntimeError(f"environment variable {name}={raw!r} must be an integer") from None


#: when set to a dict by bench.py, MFMA kernel launches are bracketed by HIP events recorded on the
#: launch stream: key -> list of (start_event, end_event, algorithmic_flops)
PROFILE = None

#: data-parallel rank folded into every dropout seed (set by runner.StrongRunner): ranks seeded alike by
#: torch.manual_seed still draw different masks for their different clips
SEED_RANK = 0

#: direct gradients (engine.py): autograd nodes write parameter gradients straight into the flat-gradient views
#: (runner.FlatParams) while this is on (StrongRunner.forward_backward, after its zero_grad)
DIRECT_GRADS = False

GRAD_READY = None        # callable(list of parameters) -> None: their gradient kernels are enqueued
GRAD_FLUSH = None        # callable() -> None: a safe point to launch the all-reduce of every complete bucket

# Arithmetic of the 3x3 convolutions (forward, dgrad, wgrad): "fp32" = exact fp32 MFMA (default); opt-in, on the bf16
# MFMA with fp32 accumulation (conv_x3.hip): "x3" = fp32 operands split exactly into 3 bf16 terms, 6 partial products;
# "x9" = all 9 partial products; "bf16" = operands rounded to bf16, one product (BASELINE configs[2] arithmetic).
CONV_MATH = os.environ.get("TAG_CONV_MATH", "fp32")

# Storage of the big activations of the conv stack (raw conv outputs, pooled block outputs and their gradients):
# "fp32" (default) or "bf16" = BASELINE configs[2] proper -- bf16 tensors in HBM, fp32 accumulation / BatchNorm statistics /
# GRU / heads / loss / master weights.  Only meaningful with CONV_MATH == "bf16" (the one-product bf16 MFMA kernels);
# with any other conv arithmetic the setting is ignored.
ACT_DTYPE = os.environ.get("TAG_ACT_DTYPE", "fp32")

#: BASELINE configs[2] mode only: GEMM operands (nn.Linear fc1 / projections, GRU input projections and their backward GEMMs)
#: rounded to bf16 on the bf16 MFMA with fp32 accumulation, as autocast would; "auto" = on exactly when ACT_DTYPE is bf16
GEMM_MATH = os.environ.get("TAG_GEMM_MATH", "auto")

#: BatchNorm batch statistics in the epilogue of the conv that produces the tensor instead of a pass of their own
FUSE_BN_STATS = os.environ.get("TAG_FUSE_BN_STATS", "1") != "0"

#: Winograd F(2x2,3x3) form of the 3x3 convolutions, all fp32 (csrc/conv_wino_fused.hip: ONE kernel per launch, the transforms inside
#: the product kernel; round 5's plane form, csrc/conv_wino.hip, remains for channel counts the fused kernels do not take): forward
#: (training: + BatchNorm statistics; inference: + BatchNorm / ReLU / pool), dgrad (+ BatchNorm-backward or pool-backward sums) and
#: weight gradient.  2.25 x fewer MFMA FLOP than the direct halo-tile kernels; since round 6 faster on EVERY layer with >= 64
#: channels on both sides (tools/wino_bench.py, B = 64: x1.5 ... x1.9 per launch).  "0" = direct kernels only.
CONV_WINOGRAD = os.environ.get("TAG_CONV_WINOGRAD", "1") != "0"

#: channel rule: the smaller count >= WINO_MIN_C and the larger >= WINO_MIN_CMAX
WINO_MIN_C = int(os.environ.get("TAG_WINO_MIN_C", "64"))
WINO_MIN_CMAX = int(os.environ.get("TAG_WINO_MIN_CMAX", "64"))

#: ... and only training launches of at least this much work, tiles x output channels (tiles = B * ceil(H/2) * ceil(W/2); 2^20 = one
#: 64-tile x 64-cout workgroup of the fused kernel per CU): smaller launches cannot fill the chip with those blocks and keep the
#: direct kernel.  (At B = 64 every layer is 30 ... 120 times above it; the 2-clip fixtures of the parity tests are below it and
#: are ALSO run with the rule forced to 1 and the step's decisions imposed on the oracle: tests/test_gpu_path.py.)
WINO_MIN_WORK = int(os.environ.get("TAG_WINO_MIN_WORK", str(1 << 20)))

#: the inference forward (BatchNorm in eval mode, nothing saved) of the same layers as Winograd too, at EVERY launch size: the choice
#: must not depend on the batch, or the same clip would score differently in a 4-clip and in a 64-clip pass (the forward is
#: batch-invariant, tests/test_gpu_infer.py)
CONV_WINOGRAD_EVAL = os.environ.get("TAG_CONV_WINOGRAD_EVAL", "1") != "0"

#: the fused kernels address a tensor through a buffer descriptor (32-bit byte offsets): launches without per-batch sums are cut into
#: batch slices below this many bytes per tensor (any cut gives the same rows: every tile is computed independently of the others)
WINO_MAX_BYTES = int(os.environ.get("TAG_WINO_MAX_BYTES", str((1 << 31) - (1 << 20))))

#: launches that took the Winograd path since import (tests assert that the benched-size step really runs through it)
WINO_LAUNCHES = 0

#: BatchNorm-backward sums in the dgrad conv epilogue (tag_conv3x3_dgrad_bnsums) instead of a separate two-tensor pass
FUSE_BN_BWD_SUMS = os.environ.get("TAG_FUSE_BN_BWD", "1") != "0"

#: block 1: bn1's backward applied inside the Cin = 1 conv backward (tag_conv3x3_c1_backward_bnrelu) instead of a separate pass
FUSE_C1_BN_BWD = os.environ.get("TAG_FUSE_C1_BN_BWD", "1") != "0"

#: the reduction half of the pool backward in the epilogue of the dgrad conv that PRODUCES the pooled gradient
#: (tag_conv3x3_dgrad_poolsums) instead of a pass of its own over the largest tensors (pool_bwd_reduce_kernel)
FUSE_POOL_BWD_SUMS = os.environ.get("TAG_FUSE_POOL_BWD", "1") != "0"

#: the same for the bf16-storage kernels (tag_conv3x3_dgrad_poolsums_bf16): built and tested, OFF by default -- the bf16 convs are
#: HBM / power-bound, the epilogue's window reads are not hidden there and the step time is level (10.62 vs 10.62-10.70 ms) while the
#: conv family's own time grows by what the removed pass cost (docs/experiments_r05.md)
FUSE_POOL_BWD_SUMS_BF16 = os.environ.get("TAG_FUSE_POOL_BWD_BF16", "0") != "0"

#: inference: conv2 of a ConvBlock writes the POOLED relu(bn(.)) straight from its output tile (tag_conv3x3_forward_bnrelu_pool_eval)
FUSE_EVAL_POOL = os.environ.get("TAG_FUSE_EVAL_POOL", "1") != "0"

#: weight-gradient convolutions are off the critical path of backward (only the optimiser needs them): on a second HIP stream
#: their workgroups fill the tails / small-grid gaps of the dgrad + BatchNorm chain.  TAG_WGRAD_STREAM = 1 / 0 forces it on / off;
#: the default ("auto", None here) is ON for the arithmetics on the bf16 MFMA (bf16 mode, x3, x9) and OFF for exact fp32: with
#: round 4's halo / all-taps kernels the fp32 step IS the sum of its kernels' isolated times and co-running two of them only
#: stretches both (same box, alternating processes: 54.82 / 54.92 ms with the side stream, 54.18 / 54.15 ms without), while the
#: bf16-MFMA modes' shorter kernels still gain (bf16 10.52-10.53 against 10.70-10.78 ms, x3 34.6 against 35.3, x9 44.8 against 45.1).
_side_env = os.environ.get("TAG_WGRAD_STREAM", "auto")
WGRAD_SIDE_STREAM = None if _side_env == "auto" else (_side_env != "0")

#: TAG_WGRAD_CU_SKIP=k (k >= 2): the side stream may not use every k-th compute unit (hipExtStreamCreateWithCUMask), so that the
#: short kernels of the main stream (BatchNorm finalizes, reductions) never queue behind a full residency round of
#: weight-gradient workgroups.  0 = an ordinary stream.
WGRAD_CU_SKIP = _env_int("TAG_WGRAD_CU_SKIP", 0)

#: the side stream's wgrad of a layer is RELEASED one kernel late -- when the dgrad conv that consumes the same dy has been
#: enqueued -- so that it starts together with the HBM-bound BatchNorm / pool backward passes that follow that dgrad
#: instead of beside the dgrad itself (two MFMA-bound kernels of equal length co-running finish together and leave the
#: bandwidth-bound passes alone on the chip; lagged, every such pass has MFMA work to hide under).
WGRAD_LAG = os.environ.get("TAG_WGRAD_LAG", "0") != "0"     # measured: 57.7 ms lagged vs 57.2 ms not -- off by default

#: parameter-gradient work of the GRU (1) -- and of fc1 (2) -- on the wgrad side stream; 0 = on the main stream
#: (TAG_SIDE_PARAM_GRADS).  Measured on one box, B = 64 (fp32 / bf16 mode ms per step): 0: 54.64-54.75 / 11.23-11.30,
#: 1: 54.57-54.69 / 11.02-11.11, 2: 54.91-55.20 / 11.08-11.13 -- fc1's GEMM on the side stream delays the block-4 wgrads more
#: than it overlaps, so the default stops at the GRU.
SIDE_PARAM_GRADS = _env_int("TAG_SIDE_PARAM_GRADS", 1)

#: TAG_SIDE_RECORD_STREAM=1 restores round 4's lifetime rule for the side stream's operands (Tensor.record_stream instead of keeping
#: them alive until join()) -- kept for the A/B that shows the allocator growth it causes under an unsynchronised host.
SIDE_RECORD_STREAM = _env_int("TAG_SIDE_RECORD_STREAM", 0) != 0


#: the names ops.py forwards
NAMES = frozenset(k for k in list(globals()) if k.isupper())
