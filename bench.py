#!/usr/bin/env python3
"""bench.py -- clips/s of the hot path (log-mel -> Cnn8Rnn -> w2v-mean text encoder -> frame
similarity -> FrameBceLoss), forward + backward + gradient all-reduce + clip + Adam, on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: biencoder Cnn8Rnn + EmbeddingAgg(512, mean) + match.DotProduct, fp32,
batch 64 per GPU, 10 s @ 32 kHz synthetic clips, dropout on, train-mode BatchNorm (weak scaling: per-GPU batch
fixed).  Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_CLIP = 101.7e9          # 3 x 33.90 GFLOP forward
PEAK_FP32_MFMA = 157.3           # TFLOP/s, MI355X_MICROARCH.md (v_mfma_f32_32x32x2_f32)


def synthetic_batch(B, n_samples, seed, device):
    g = torch.Generator().manual_seed(seed)
    wave = 0.1 * torch.randn(B, n_samples, generator=g)
    text = torch.randint(2, 5221, (B, 4), generator=g)
    text_len = 1 + torch.arange(B) % 4
    for i in range(B):
        text[i, text_len[i]:] = 0
    label = (torch.rand(B, (n_samples // 320 + 1) // 4, generator=g) < 0.5).float()
    return {"waveform": wave.to(device), "waveform_len": torch.full((B,), n_samples).numpy(), "text": text.to(device),
            "text_len": text_len.to(device), "label": label.to(device)}


def usable_cores():
    """Host cores this process may actually use: min(affinity, cgroup v2 cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _time_oracle(batch, iters, threads):
    from oracle import tag_oracle as O
    torch.set_num_threads(threads)
    st = O.state_to(O.init_state(seed=0), torch.float32, requires_grad=True)
    b = O.synthetic_batch(batch, 320000, seed=1234)
    times = []
    for i in range(iters + 1):                 # 1 warm-up + iters timed
        for v in st.values():
            if v.is_floating_point() and v.grad is not None:
                v.grad = None
        t0 = time.perf_counter()
        loss, _ = O.train_step_loss(st, b, "dot", "cnn8rnn", True)
        loss.backward()
        times.append(time.perf_counter() - t0)
    return sorted(times[1:])[len(times[1:]) // 2]


def cpu_baseline(batch=16, iters=5):
    """the oracle (plain PyTorch eager CPU restatement of the reference) timed on the host cores of
    this box: one training step's fwd+bwd, dropout on, B = 16 (configs[0]), 1 warm-up + 5 timed, median; all usable cores
    and -- on a smaller sample of the same workload, to bound the run -- one thread."""
    cores = usable_cores()
    t_all = _time_oracle(batch, iters, cores)
    t_one = _time_oracle(2, iters, 1)
    return {"value": round(batch / t_all, 3), "unit": "clips/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": f"oracle fwd+bwd (Cnn8Rnn + EmbeddingAgg(512) + DotProduct, dropout on), B={batch} x 10 s clips, "
                      f"median of {iters} after 1 warm-up",
            "single_thread": {"value": round(2 / t_one, 3), "unit": "clips/s", "cores": 1,
                              "sample": f"same step, B=2, median of {iters} after 1 warm-up"}}


#: profile family (ops._timed key) -> the kernels behind it, as a trace names them
KERNEL_TEXT = {
    "conv3x3_wino": "wino_fused_kernel<PRO,EPI> (Winograd F(2x2,3x3) forward and dgrad launches: transforms, 16 products and "
                    "epilogue in one kernel; csrc/conv_wino_fused.hip)",
    "conv3x3_wino_wgrad": "wino_fused_wgrad_kernel<PRO> + wino_fused_wgrad_finish_kernel (Winograd weight gradient)",
    "conv3x3_halo_kernel": "conv3x3_halo_kernel<BN,PRO,TW,EPI> (direct MFMA convolution, forward and dgrad)",
    "conv3x3_wgrad_alltaps_kernel": "conv3x3_wgrad_alltaps_kernel / conv3x3_wgrad_rowring_kernel (direct weight gradient)",
}


def families(prof):
    """ops.PROFILE (key -> [(start event, end event, algorithmic FLOP)]) folded per kernel family."""
    fam = {}
    for key, evs in prof.items():
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in evs)
        fl = sum(f for _, _, f in evs)
        d = fam.setdefault(key[0], {"ms": 0.0, "flop": 0.0, "launches": 0})
        d["ms"] += ms; d["flop"] += fl; d["launches"] += len(evs)
    return fam


def roofline_of(fam, fam_iso, steps, mode, conv_math, batch, overlap):
    """`roofline` object of the dominant MFMA kernel family of a timed leg: achieved = sum of the algorithmic FLOP of its
    launches / sum of their HIP-event durations (events on the launch stream).  mode = "fp32" | "bf16" selects the committed PMC
    traffic profile (profiles/r*_pmc_hbm_traffic_<mode>.json); the profile is only quoted when it was collected on the kernel
    sources the loaded libtag_hip.so attests it was built from (tag_build_id = sha256 of csrc/ at compile time)."""
    if not fam:
        return None
    # the dominant family is the one that owns the most TIME of the timed region (round 5's record picked by FLOP and named a family
    # that no longer described the step)
    dom = max(fam, key=lambda k: fam[k]["ms"])
    ach = fam[dom]["flop"] / (fam[dom]["ms"] * 1e-3) / 1e12
    iso = fam_iso[dom]["flop"] / (fam_iso[dom]["ms"] * 1e-3) / 1e12
    # HBM bytes per launch of the dominant family: PMC counters cannot be collected from inside this process, so the
    # figure is read from the committed profile of the SAME command (tools/collect_profiles.sh: two separate --pmc passes,
    # FETCH_SIZE x2 + WRITE_SIZE) and labelled with where and when it was collected -- it is offline data.
    traffic, traffic_src = None, None
    from texttoaudiogrounding_amd.lib import build_id as csrc_sha256   # the id the LOADED BINARY attests (tag_build_id), not a hash of the checkout
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_hbm_traffic_{mode}.json")), reverse=True)
    if not cands or conv_math not in ("fp32", "bf16"):
        traffic_src = {"reason": "no PMC traffic profile for this arithmetic under profiles/"}
    else:
        tj = json.load(open(cands[0]))
        meta = tj.pop("_meta", {})
        tname = os.path.basename(cands[0])
        if meta.get("csrc_sha256") != csrc_sha256():
            # a profile of OTHER kernel sources says nothing about this build: no number rather than a stale one
            traffic_src = {"file": f"profiles/{tname}", "collected_utc": meta.get("collected_utc"),
                           "reason": "profile was collected on different kernel sources (csrc sha256 "
                                     f"{str(meta.get('csrc_sha256'))[:12]} != built {csrc_sha256()[:12]}); re-run "
                                     "tools/collect_profiles.sh"}
        else:
            # the bf16 conv family is two kernels since round 4: the tile kernel and the row-streaming kernel of conv_rows.hip
            names = {"conv3x3_x3_kernel": ("conv3x3_x3_kernel", "conv3x3_rows_kernel"),
                     "conv3x3_wgrad_x3_kernel": ("conv3x3_wgrad_x3_kernel", "conv3x3_wgrad_dma_kernel"),
                     "conv3x3_wino": ("wino_fused_kernel",), "conv3x3_wino_wgrad": ("wino_fused_wgrad_kernel",)}.get(dom, (dom,))
            rows = [v for k, v in tj.items() if k.startswith(names)]
            n = sum(v["launches_in_run"] for v in rows)
            if n:
                traffic = round(sum((v["fetch_GB_per_launch"] + v["write_GB_per_launch"]) * v["launches_in_run"]
                                    for v in rows) / n, 3)
                traffic_src = {"file": f"profiles/{tname}", "collected_utc": meta.get("collected_utc"),
                               "csrc_sha256": meta.get("csrc_sha256"), "measured_in_this_run": False,
                               "whole_step_GB": round(meta.get("fetch_GB_per_step", 0) + meta.get("write_GB_per_step", 0), 2),
                               "algorithmic_lower_bound_GB_per_step": round((0.082 if mode == "bf16" else 0.164) * batch, 2)}
    # x3 kernels: 6 bf16 MFMA products per fp32 multiply -> peak = dense bf16 MFMA peak (2500 TFLOP/s) / 6
    nprod = {"x3": 6.0, "x9": 9.0, "bf16": 1.0}.get(conv_math, 6.0)
    peak = PEAK_FP32_MFMA if not dom.startswith("conv3x3_x3") and not dom.startswith("conv3x3_pc") else round(2500.0 / nprod, 1)
    def fam_peak(k):
        return PEAK_FP32_MFMA if not k.startswith("conv3x3_x3") and not k.startswith("conv3x3_wgrad_x3") and not k.startswith("conv3x3_pc") \
            else round(2500.0 / nprod, 1)

    wino = dom.startswith("conv3x3_wino")
    secondary = {KERNEL_TEXT.get(k, k): {"TFLOP/s": round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 2),
                                         "frac": round(v["flop"] / (v["ms"] * 1e-3) / 1e12 / fam_peak(k), 4),
                                         "ms_per_step": round(v["ms"] / steps, 3), "launches_per_step": v["launches"] // steps}
                 for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]) if k != dom}
    return {"bound": "mfma", "kernel": KERNEL_TEXT.get(dom, dom), "family": dom, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(ach / peak, 4),
            # what `achieved` counts: the FLOP the launches EXECUTE on the matrix pipe.  A Winograd launch executes 16 products of
            # (tiles x Cin x Cout) per 2 x 2 output tile = the direct-convolution FLOP / 2.25
            "flop_counted": ("executed: 2 * 16 * tiles * Cin * Cout per launch (direct-convolution FLOP / 2.25)" if wino
                             else "2 * B * H * W * 9 * Cin * Cout per launch"),
            "ms_per_step": round(fam[dom]["ms"] / steps, 3),
            "secondary": secondary,
            "traffic": traffic, "traffic_unit": "GB per launch (HBM, PMC, offline profile)",
            "traffic_source": traffic_src,
            "avg_launch_ms": round(fam[dom]["ms"] / fam[dom]["launches"], 4),
            "launches_per_step": fam[dom]["launches"] // steps,
            "streams": "overlapped (wgrad on a side stream)" if overlap else "single",
            "isolated_achieved": round(iso, 2), "isolated_frac": round(iso / peak, 4),
            "isolated_avg_launch_ms": round(fam_iso[dom]["ms"] / fam_iso[dom]["launches"], 4),
            "families_isolated": {k: {"TFLOP/s": round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 2),
                                      "ms_per_step": round(v["ms"] / steps, 3)} for k, v in fam_iso.items()}}


def mem_record(device):
    """Peak device memory since the last reset: what the caching allocator handed out (allocated) and what it holds (reserved).
    Reserved far above allocated in a leg = the allocator could not reuse freed blocks (a host that ran ahead of events recorded on
    a second stream) -- the symptom of the round-4 side-stream growth (ops._SideWgrad)."""
    st = torch.cuda.memory_stats(device)
    return {"max_allocated_GB": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2),
            "max_reserved_GB": round(torch.cuda.max_memory_reserved(device) / 2 ** 30, 2),
            # since process start: hipMalloc calls the allocator had to make, and allocations that failed first (the allocator then
            # synchronises the device and frees its cache before retrying: a stall)
            "device_mallocs": int(st.get("num_device_alloc", 0)), "alloc_retries": int(st.get("num_alloc_retries", 0))}


def run_steps(runner, batch, k, sync, spread=None):
    """k training steps; returns (wall seconds incl. the closing sync, host seconds spent INSIDE train_step = the time one core
    needs to enqueue a step: ctypes launches + autograd bookkeeping, no synchronisation), and the last loss.
    spread: a dict that receives {"min", "median", "max"} of the k steps' own GPU durations (one HIP event per step boundary on the
    compute stream -- a marker packet, no synchronisation): a stall inside the timed region (round 5: one hipMalloc blocking for
    seconds in the middle of an unsynchronised run) shows as max >> median instead of hiding in the mean."""
    host = 0.0
    loss = None
    marks = None
    if spread is not None and torch.cuda.is_available():
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
        marks[0].record()
    t0 = time.perf_counter()
    for i in range(k):
        th = time.perf_counter()
        loss = runner.train_step(dict(batch))
        host += time.perf_counter() - th
        if marks is not None:
            marks[i + 1].record()
    sync()
    dt = time.perf_counter() - t0
    if marks is not None:
        d = [marks[i].elapsed_time(marks[i + 1]) for i in range(k)]
        first, d = d[0], sorted(d)
        spread.update({"min": round(d[0], 3), "median": round(d[len(d) // 2], 3), "max": round(d[-1], 3), "first": round(first, 3)})
    return dt, host, loss


def build_workload(name, device):
    """The models of the secondary workloads, on the same kernels as the contract's (DESIGN.md sections 9-10)."""
    from texttoaudiogrounding_amd.models import audio_encoder, audio_text_model, match, text_encoder
    if name == "crnn":
        return audio_text_model.BiEncoder(audio_encoder.CrnnEncoder(32000, 256), text_encoder.EmbeddingAgg(5221, 256),
                                          match.ExpNegL2(), 256)
    if name == "cross_attention":
        return audio_text_model.BiEncoder(audio_encoder.Cnn8Rnn(32000), text_encoder.EmbeddingAgg(5221, 512),
                                          match.CrossAttention(512, 8, 0.1), 512)
    if name == "cross_encoder":
        from texttoaudiogrounding_amd.models.cross_encoder import CrossAttentionGating
        return audio_text_model.BiEncoder(audio_encoder.Cnn8Rnn(32000), text_encoder.EmbeddingAgg(5221, 512),
                                          match.DotProduct(text_level="token"), 512, cross_encoder=CrossAttentionGating(512))
    return audio_text_model.BiEncoder(audio_encoder.Cnn8Rnn(32000), text_encoder.EmbeddingAgg(5221, 512),
                                      match.DotProduct(), 512)


WORKLOAD_TEXT = {
    "cross_encoder": "configs[3]: biencoder Cnn8Rnn + EmbeddingAgg(512) + CrossAttentionGating(512) + match.DotProduct(token), "
                     "fwd+bwd+clip+Adam, B=64 x 10 s",
    "cross_attention": "configs[3]: biencoder Cnn8Rnn + EmbeddingAgg(512) + match.CrossAttention(512, 8 heads, p 0.1), "
                       "fwd+bwd+clip+Adam, B=64 x 10 s",
    "crnn": "strong eg_config as written (cdur_w2vmean.yaml): CrnnEncoder(256) + EmbeddingAgg(256) + match.ExpNegL2, "
            "fwd+bwd+clip+Adam, B=64 x 10 s",
    "infer_30s_b256": "configs[4]: Cnn8Rnn + LAION-CLAP text tower (RoBERTa-base shape, random init) + audio/text proj + "
                      "DotProduct, forward only, B=256 x 30 s @ 32 kHz, 8-token phrases",
}


def other_workloads(args, device, log, budget_s=5.0):
    """BASELINE configs[3] (both heads), the strong eg_config's CrnnEncoder and configs[4]'s 30 s inference, each timed for a
    bounded number of steps (<= ~5 s each) on the same device after the contract's legs; reported BESIDE `value`, never as it.
    fp32 (exact MFMA) like `value`."""
    from texttoaudiogrounding_amd.runner import StrongRunner
    res = {}
    for name in ("cross_encoder", "cross_attention", "crnn"):
        torch.manual_seed(0)
        runner = StrongRunner(build_workload(name, device), lr=1e-3, max_grad_norm=1.0, device=str(device))
        batch = synthetic_batch(args.batch, 320000, 1234, device)
        for _ in range(3):
            runner.train_step(dict(batch))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.train_step(dict(batch))
        torch.cuda.synchronize()
        one = time.perf_counter() - t0
        k = max(3, min(args.steps, int(budget_s / 2 / max(one, 1e-4))))
        blocks = []
        for _ in range(2):                    # two timed blocks of k steps; BOTH are reported, `value` is the faster one (the first
            d_, h_, loss = run_steps(runner, batch, k, torch.cuda.synchronize)   # block after a model switch has been seen 3x slow once:
            blocks.append((d_, h_))           # allocator / clock state, not the workload) and the estimator is named in the record
        dt, host = min(blocks)
        res[name] = {"workload": WORKLOAD_TEXT[name], "value": round(args.batch * k / dt, 2), "unit": "clips/s",
                     "ms_per_step": round(dt / k * 1e3, 3), "steps": k, "dtype": "f32", "loss": round(runner.loss_value(loss), 6),
                     "estimator": "min_of_2_blocks", "blocks_ms_per_step": [round(b[0] / k * 1e3, 3) for b in blocks],
                     "host_enqueue_ms_per_step": round(host / k * 1e3, 3)}
        log(f"other workload {name}: {res[name]['value']} clips/s")
        del runner
        torch.cuda.empty_cache()
    # configs[4]: inference, 30 s clips, batch 256 (models/hf_modeling_grounding.py:319-352 of the reference)
    from texttoaudiogrounding_amd.models.hf_modeling_grounding import Cnn8RnnLaionClapGroundingModel
    torch.manual_seed(0)
    B, tokens = 256, 8
    model = Cnn8RnnLaionClapGroundingModel().to(device).eval()
    g = torch.Generator(device=device).manual_seed(1234)
    audio = 0.1 * torch.randn(B, 960000, device=device, generator=g)
    audio_len = torch.full((B,), 960000)
    ids = torch.randint(3, 50265, (B, tokens), device=device, generator=g)
    ids[:, 0], ids[:, -1] = 0, 2
    text = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
    model(audio, audio_len, text)
    torch.cuda.synchronize()
    k = 3
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(k):
        th = time.perf_counter()
        fs = model(audio, audio_len, text)
        host += time.perf_counter() - th
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for _ in range(k):
        model.model.text_encoder(text)
    torch.cuda.synchronize()
    dtt = time.perf_counter() - t1
    res["infer_30s_b256"] = {"workload": WORKLOAD_TEXT["infer_30s_b256"], "value": round(B * k / dt, 2), "unit": "clips/s",
                             "ms_per_step": round(dt / k * 1e3, 2), "steps": k, "dtype": "f32", "frame_sim_shape": list(fs.shape),
                             "estimator": "one_block_after_one_warm_up_pass", "host_enqueue_ms_per_step": round(host / k * 1e3, 3),
                             # forward of a 30 s clip = 3 x 33.90 GFLOP (T' = 750) -- numerically FLOP_PER_CLIP
                             "whole_forward_mfma_frac": round(B * k / dt * 3 * 33.90e9 / 1e12 / PEAK_FP32_MFMA, 4),
                             "whole_forward_mfma_frac_counts": "algorithmic direct-convolution FLOP (effective rate: blocks 3-4 run as "
                                                               "Winograd F(2x2,3x3) and execute 2.25 x fewer)",
                             "text_tower_ms_per_batch": round(dtt / k * 1e3, 2)}
    log(f"other workload infer_30s_b256: {res['infer_30s_b256']['value']} clips/s")
    return res


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run with N ranks on this node
    (one process per GPU, rendezvous on 127.0.0.1); the child rank 0 prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def workload_name(args):
    tail = ", fwd+bwd+clip+Adam, dropout on, train-mode BN"
    if args.crnn:
        return ("strong eg_config as written (cdur_w2vmean.yaml): CrnnEncoder(256) + EmbeddingAgg(256) + match.ExpNegL2, "
                "fwd+bwd+clip+Adam")
    if args.cross_attention:
        return "configs[3]: biencoder Cnn8Rnn + EmbeddingAgg(512) + match.CrossAttention(512, 8 heads, p 0.1)" + tail
    if args.cross_encoder:
        return "configs[3]: biencoder Cnn8Rnn + EmbeddingAgg(512) + CrossAttentionGating(512) + match.DotProduct(token)" + tail
    return "configs[1]: biencoder Cnn8Rnn + EmbeddingAgg(512,mean) + match.DotProduct" + tail


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 20 timed steps (0.7 s of the contract workload).  The first timed step starts on an idle queue -- the host has
    # nothing enqueued ahead, so the GPU waits for launches (+1.5 ... 2.5 ms on that step, `step_ms.first`); a training job runs in
    # the steady state, where the host is a step ahead.  With 5 steps that one step was 1 % of the mean.
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cross-encoder", action="store_true",
                    help="time BASELINE configs[3] instead: the same biencoder with CrossAttentionGating(512) and the "
                         "token-level DotProduct (NOT the contract's workload; for DESIGN.md section 10)")
    ap.add_argument("--cross-attention", action="store_true",
                    help="time BASELINE configs[3] with the head its text names: the same biencoder scored by "
                         "match.CrossAttention(512, 8 heads, dropout 0.1) (nn.MultiheadAttention of every frame over the "
                         "phrase tokens + residual + LayerNorm + Linear(512,1)); NOT the contract's workload")
    ap.add_argument("--crnn", action="store_true",
                    help="time the variant the strong eg_config literally instantiates instead (cdur_w2vmean.yaml: CrnnEncoder "
                         "(256) + EmbeddingAgg(256) + match.ExpNegL2); NOT the contract's workload")
    ap.add_argument("--comm-only", action="store_true",
                    help="N > 1: time ONLY the gradient buckets' all-reduces (fp32 and bf16 payload), nothing else, and print "
                         "that as the JSON line (diagnosis of a scaling run; not the contract's metric)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra legs (direct kernels; the bf16 mode)")
    ap.add_argument("--alt-math", default="bf16",
                    help="comma list of the other conv arithmetics timed after the contract's leg: bf16 (configs[2]'s per-rank mode; "
                         "default), x3, x9 (opt-in splits: no roofline credit, retired from the default record in round 6)")
    ap.add_argument("--no-probe", action="store_true",
                    help="skip the ~50 ms register-resident MFMA probes behind `measured_ceiling` (profile collection: they are "
                         "not kernels of the step)")
    ap.add_argument("--no-others", action="store_true",
                    help="skip the `other_workloads` block (configs[3] cross-encoder / cross-attention, the strong eg_config's "
                         "CrnnEncoder, configs[4] 30 s inference at batch 256: a few seconds each, never `value`)")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"],
                    help="bf16 = BASELINE configs[2] as a mode: bf16 conv arithmetic (fp32 accumulate) + bf16 storage of the conv "
                         "stack's activations and gradients (the all-reduce payload stays fp32 unless --grad-wire bf16); fp32 BatchNorm statistics / GRU / heads "
                         "/ loss / master weights.  The default (and the contract's `value`) is fp32")
    ap.add_argument("--grad-wire", default="fp32", choices=["fp32", "bf16"],
                    help="payload of the gradient all-reduce.  fp32 (default, also with --dtype bf16): the 35 MB of a step hide "
                         "under the remaining backward on xGMI, and the collective then sums in fp32.  bf16 halves the bytes but "
                         "the collective sums in bf16 (rms error ~ sqrt(N) * 2^-9): switch only when `comm.exposed_ms_per_step` "
                         "of a scaling run shows exposed communication")
    ap.add_argument("--conv-math", default="fp32", choices=["fp32", "x3", "x9", "bf16"],
                    help="arithmetic of the 3x3 conv forward/dgrad kernels of the TIMED run: fp32 = exact fp32 MFMA "
                         "(the contract's number); x3 = opt-in 3 x bf16 split on the bf16 MFMA (conv_x3.hip)")
    args = ap.parse_args()
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        sys.exit(self_launch(args))

    import torch.distributed as dist
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd.models import audio_encoder, audio_text_model, match, text_encoder
    from texttoaudiogrounding_amd.runner import StrongRunner, barrier, comm_environment, init_distributed

    if args.dtype == "bf16":
        args.conv_math = "bf16"
        ops.ACT_DTYPE = "bf16"
    ops.CONV_MATH = args.conv_math
    rank, world, local = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    device = torch.device(f"cuda:{local}")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) are visible to this process")
    torch.cuda.set_device(device)
    if world > 1 and os.environ.get("TAG_SHARE_GPU", "0") != "1":      # (the test hook puts every rank on cuda:0 on purpose)
        # one line of diagnosis instead of a run that silently times one GPU: every rank on its own device, that device = LOCAL_RANK
        me = (rank, local, torch.cuda.current_device(), str(getattr(torch.cuda.get_device_properties(local), "uuid", local)))
        seen = [None] * world
        dist.all_gather_object(seen, me)
        bad = [m for m in seen if m[1] != m[2]]
        if bad or len({m[3] for m in seen}) != world or len(seen) != args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: ranks do not map one-to-one onto GPUs (rank, LOCAL_RANK, current device, "
                             f"uuid) = {seen}")
    torch.manual_seed(0)
    model = build_workload("crnn" if args.crnn else "cross_attention" if args.cross_attention else
                           "cross_encoder" if args.cross_encoder else "biencoder", device)
    runner = StrongRunner(model, lr=1e-3, max_grad_norm=1.0, device=str(device),
                          grad_comm_dtype=torch.bfloat16 if args.grad_wire == "bf16" else None)
    batch = synthetic_batch(args.batch, 320000, 1234 + rank, device)

    def sync():
        if world > 1:
            barrier(local)                     # names this rank's device to RCCL (no "device under current context" guess)
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    def comm_only(iters=10):
        """The buckets' collectives alone, back to back on the communication stream: per payload type the time of one
        step's worth of all-reduces, each bucket's own duration and the bus bandwidth 2(n-1)/n x bytes / time."""
        bk = runner.buckets
        res = {}
        for name, dt_ in (("fp32", None), ("bf16", torch.bfloat16)):
            bufs = [runner.flat.grad[s0:e].clone() if dt_ is None else runner.flat.grad[s0:e].to(dt_) for (s0, e, _, _) in bk.bounds]
            evs = []
            for it in range(iters + 2):
                sync()
                with torch.cuda.stream(bk.comm_stream):
                    row = [torch.cuda.Event(enable_timing=True) for _ in range(len(bufs) + 1)]
                    row[0].record(bk.comm_stream)
                    for i, b_ in enumerate(bufs):
                        b_.zero_()                                   # keep the payload finite over the repetitions
                        w = dist.all_reduce(b_, async_op=True)
                        w.wait()
                        row[i + 1].record(bk.comm_stream)
                if it >= 2:
                    evs.append(row)
            sync()
            per = [sum(r[i].elapsed_time(r[i + 1]) for r in evs) / len(evs) for i in range(len(bufs))]
            nbytes = sum(b_.numel() * b_.element_size() for b_ in bufs)
            tot = sum(per)
            t = torch.tensor([tot], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tot = t.item()
            res[name] = {"ms_per_step": round(tot, 4), "bucket_ms": [round(v, 4) for v in per], "payload_MB": round(nbytes / 2 ** 20, 2),
                         "bus_GB_per_s": round(2 * (world - 1) / world * nbytes / (tot * 1e-3) / 1e9, 2) if tot > 0 else None}
        return res

    if args.comm_only:
        if runner.buckets is None:
            raise SystemExit("--comm-only needs --gpus N > 1 (one rank has no gradient exchange)")
        res = comm_only()
        if rank == 0:
            print(json.dumps({"metric": "gradient all-reduce alone (diagnosis, not the contract's metric)", "n_gpus": world,
                              "backend": dist.get_backend(),
                              "buckets_MB": [round((e - s0) * 4 / 2 ** 20, 2) for (s0, e, _, _) in runner.buckets.bounds],
                              "comm_only": res}))
        dist.destroy_process_group()
        return
    from texttoaudiogrounding_amd.utils.telemetry import BoardSampler, mfma_probe
    sampler = BoardSampler(local)              # hwmon power + shader clock every 25 ms while a leg runs (a sysfs read: no GPU work)
    log(f"model on {device}, batch {args.batch}/GPU; warm-up {args.warmup} step(s)")
    for i in range(args.warmup):
        tw = time.perf_counter()
        runner.train_step(dict(batch))
        torch.cuda.synchronize()
        log(f"warm-up step {i}: {time.perf_counter() - tw:.3f} s")
    ops.PROFILE = {}
    if runner.buckets is not None:
        runner.buckets.record = True          # per-bucket events on the communication stream + the exposed wait
    sampler.start()
    torch.cuda.reset_peak_memory_stats(device)
    step_spread = {}
    sync()                                     # barrier + synchronize, then NOTHING between it and the first timed enqueue (an idle
    dt, host_s, loss = run_steps(runner, batch, args.steps, sync, step_spread)     # gap of some ms lets the shader clock fall)
    board = sampler.stop()
    mem_main = mem_record(device)
    log(f"timed {args.steps} steps in {dt:.3f} s (host enqueue {host_s / args.steps * 1e3:.2f} ms/step)")
    prof, ops.PROFILE = ops.PROFILE, None
    comm_timing = None
    if runner.buckets is not None:
        runner.buckets.record = False
        comm_timing = runner.buckets.timing_summary()
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    clips = world * args.batch * args.steps
    value = clips / dt

    # ---- roofline of the dominant kernel family (halo-tile conv: forward+dgrad launches), HIP events recorded on
    # the launch stream over the timed region.  The timed region runs the weight-gradient convs on a second stream
    # (they overlap the dgrad/BatchNorm chain), which stretches every overlapped kernel's own start-to-end time;
    # a second, untimed pass of the same K steps with that overlap switched off gives the isolated kernel rate.
    fam = families(prof)
    overlap = ops.side_stream_enabled()         # auto: off for the fp32-accurate arithmetics, on in the bf16 mode (settings.py)
    side_setting = ops.WGRAD_SIDE_STREAM
    fam_iso = fam
    if overlap:
        ops.WGRAD_SIDE_STREAM = False
        ops.PROFILE = {}
        for _ in range(args.steps):
            runner.train_step(dict(batch))
        sync()
        fam_iso, ops.PROFILE = families(ops.PROFILE), None
        ops.WGRAD_SIDE_STREAM = side_setting
    # ---- the same K steps with the direct halo-tile kernels everywhere (TAG_CONV_WINOGRAD=0): what the Winograd form of the deep
    # layers is worth on this box, in the driver-timed record
    alt_algo = None
    if args.conv_math == "fp32" and ops.CONV_WINOGRAD and not args.no_alt:
        ops.CONV_WINOGRAD = False
        runner.train_step(dict(batch))
        sync()
        dtd, host_d, _ = run_steps(runner, batch, args.steps, sync, None)
        ops.CONV_WINOGRAD = True
        if world > 1:
            t = torch.tensor([dtd], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtd = t.item()
        alt_algo = {"direct": {"conv_algo": "direct MFMA convolution in every layer (TAG_CONV_WINOGRAD=0)",
                               "value": round(clips / dtd, 2), "unit": "clips/s", "ms_per_step": round(dtd / args.steps * 1e3, 3)}}
    # ---- the same K steps with the opt-in conv arithmetic (reported beside the contract's number, never as `value`)
    alt = None
    if args.conv_math == "fp32" and not args.no_alt:
        alt = {}
        desc = {"x3": "fp32 operands split exactly into 3 bf16 terms, 6 partial products on v_mfma_f32_32x32x16_bf16, f32 "
                      "accumulate (forward, dgrad and wgrad convs; same parity tolerances as fp32; opt-in TAG_CONV_MATH=x3)",
                "bf16": "BASELINE configs[2] as a mode (python bench.py --dtype bf16): conv operands bf16, one product, f32 "
                        "accumulate; raw conv outputs, pooled activations and their gradients STORED as bf16; BatchNorm "
                        "statistics, GRU, heads, loss, Adam and master weights f32"}
        desc["x9"] = ("fp32 operands split exactly into 3 bf16 terms, ALL 9 partial products on v_mfma_f32_32x32x16_bf16 (every "
                      "partial product exact, f32 accumulate; forward, dgrad and wgrad convs; opt-in TAG_CONV_MATH=x9)")
        for mode in [m for m in args.alt_math.split(",") if m in ("x3", "x9", "bf16")]:
            ops.CONV_MATH = mode
            ops.ACT_DTYPE = "bf16" if mode == "bf16" else "fp32"
            runner.train_step(dict(batch))
            ops.PROFILE = {} if mode == "bf16" else None          # the bf16 mode gets its own roofline (events as in the main leg)
            sampler.start()
            torch.cuda.reset_peak_memory_stats(device)
            spread_a = {}
            sync()
            dta, host_a, _ = run_steps(runner, batch, args.steps, sync, spread_a)
            board_a = sampler.stop()
            mem_a = mem_record(device)
            prof_a, ops.PROFILE = ops.PROFILE, None
            if world > 1:
                t = torch.tensor([dta], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dta = t.item()
            alt[mode] = {"conv_math": desc[mode], "value": round(clips / dta, 2), "unit": "clips/s",
                         "ms_per_step": round(dta / args.steps * 1e3, 3),
                         "host_enqueue_ms_per_step": round(host_a / args.steps * 1e3, 3), "board": board_a,
                         "device_memory": mem_a, "step_ms": spread_a}
            alt[mode]["wgrad_side_stream"] = bool(ops.side_stream_enabled())
            if mode == "bf16":
                fam_a = fam_a_iso = families(prof_a)
                overlap_a = ops.side_stream_enabled()
                if overlap_a:                                     # isolated kernel rate: the same steps, wgrad convs in line
                    ops.WGRAD_SIDE_STREAM = False
                    ops.PROFILE = {}
                    for _ in range(args.steps):
                        runner.train_step(dict(batch))
                    sync()
                    fam_a_iso, ops.PROFILE = families(ops.PROFILE), None
                    ops.WGRAD_SIDE_STREAM = side_setting
                alt[mode]["whole_step_mfma_frac"] = round(clips / dta / world * FLOP_PER_CLIP / 1e12 / 2500.0, 4)
                alt[mode]["roofline"] = roofline_of(fam_a, fam_a_iso, args.steps, "bf16", "bf16", args.batch, overlap_a)
                if alt[mode]["roofline"] is not None and not args.no_probe:
                    # what the bf16 matrix pipe sustains on THIS box right now with operands that toggle like a kernel's (the
                    # part's power limit) and with constant operands (datasheet conditions), ~50 ms each
                    alt[mode]["roofline"]["measured_ceiling"] = {"random_operands": mfma_probe("bf16_random", 50.0, local),
                                                                 "constant_operands": mfma_probe("bf16_constant", 50.0, local)}
                    mc = alt[mode]["roofline"]["measured_ceiling"]["random_operands"]["TFLOP/s"]
                    alt[mode]["roofline"]["isolated_frac_of_measured_ceiling"] = round(
                        alt[mode]["roofline"]["isolated_achieved"] / mc, 4) if mc else None
        ops.CONV_MATH = "fp32"
        ops.ACT_DTYPE = "fp32"
    loss_value = round(runner.loss_value(loss), 6)
    roof = roofline_of(fam, fam_iso, args.steps, "bf16" if args.dtype == "bf16" else "fp32", args.conv_math, args.batch, overlap)
    if roof is not None and rank == 0 and not args.no_probe:
        kind = "bf16" if args.conv_math in ("bf16", "x3", "x9") else "f32"
        roof["measured_ceiling"] = {"random_operands": mfma_probe(f"{kind}_random", 50.0, local),
                                    "constant_operands": mfma_probe(f"{kind}_constant", 50.0, local)}
    others = None
    main_wl = not (args.crnn or args.cross_attention or args.cross_encoder)
    if world == 1 and main_wl and args.conv_math == "fp32" and not args.no_others:
        del runner, model
        torch.cuda.empty_cache()
        others = other_workloads(args, device, log)
    comm_only_res = comm_only() if world > 1 else None        # every rank takes part; a few ms
    if rank == 0:
        out = {"metric": "clips/sec (10 s@32 kHz, 1-phrase) fwd+bwd", "value": round(value, 2), "unit": "clips/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3),
               # wall time one host core spends inside train_step per step (launch enqueue through ctypes + autograd bookkeeping,
               # no synchronisation): must stay below ms_per_step or the GPU starves
               "host_enqueue_ms_per_step": round(host_s / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None,
               "dtype": ("bf16 (conv arithmetic + activation storage; f32 accumulate, statistics, GRU, heads, master weights, "
                         f"{args.grad_wire} all-reduce payload)" if args.dtype == "bf16" else
                         {"fp32": "f32", "bf16": "bf16 conv operands, f32 accumulate, f32 elsewhere"}.get(
                             args.conv_math, "f32 (conv products as 3 x bf16 split, f32 accumulate)")),
               "data": "synthetic",
               "config": {"workload": workload_name(args), "batch_per_gpu": args.batch,
                          "global_batch": world * args.batch, "clip": "10 s @ 32 kHz", "parallelism": f"dp{world}",
                          "conv_math": args.conv_math,
                          "conv_algo": ("fp32 Winograd F(2x2,3x3), transforms fused into the product kernel, for the forward, dgrad and "
                                        "weight-gradient convs of every layer with >= 64 channels (21 launches per step; "
                                        "csrc/conv_wino_fused.hip); the Cin = 1 conv stays direct"
                                        if (args.conv_math == "fp32" and ops.CONV_WINOGRAD) else "direct MFMA convolution")},
               "loss": loss_value,
               "whole_step_mfma_frac": round(value / world * FLOP_PER_CLIP / 1e12 /
                                             (2500.0 if args.conv_math == "bf16" else PEAK_FP32_MFMA), 4),
               # counts the ALGORITHMIC (direct-convolution) FLOP of the step, 101.7 GFLOP per clip; with the Winograd form of the deep
               # layers the matrix pipe executes fewer (2.25 x fewer in those launches), so this is an effective rate, not pipe utilisation
               "whole_step_mfma_frac_counts": "algorithmic direct-convolution FLOP (effective rate)",
               # the FLOP the step EXECUTES on the matrix pipe (the timed conv families as launched + 2.36 GFLOP per clip of GEMM / GRU /
               # mel work) over the whole step time: pipe utilisation, comparable with roofline.frac
               "whole_step_mfma_frac_executed": round((sum(v["flop"] for v in fam.values()) / args.steps + 2.36e9 * args.batch)
                                                      / (dt / args.steps) / 1e12 /
                                                      (2500.0 if args.conv_math == "bf16" else PEAK_FP32_MFMA), 4) if fam else None,
               "roofline": roof,
               # socket power and shader clock sampled (hwmon, every 25 ms) while the timed region ran
               "board": board,
               # peak device memory over the timed region (allocated by tensors / held by the caching allocator)
               "device_memory": mem_main,
               # the timed steps' own GPU durations (events at the step boundaries): a stall would show as max >> median
               "step_ms": step_spread}
        if alt_algo:
            out["alt_conv_algo"] = alt_algo
        if alt:
            out["alt_conv_math"] = alt
        if others:
            out["other_workloads"] = others
        out["ranks_observed"] = dist.get_world_size() if world > 1 else 1
        out["backend"] = dist.get_backend() if world > 1 else None
        if world > 1:
            out["comm"] = {"collective": "all-reduce(sum) of the flat fp32 gradient in buckets, launched from inside backward",
                           "buckets_MB": [round((e - s0) * 4 / 2 ** 20, 2) for (s0, e, _, _) in runner.buckets.bounds],
                           "overlap": runner.overlap_comm,
                           "environment": comm_environment(),
                           "payload": args.grad_wire,
                           # rank 0's events over the timed region: each bucket's all-reduce start->end on the communication
                           # stream (includes waiting for slower ranks to arrive) and the time the compute stream spent blocked
                           # on the communication stream at the end of backward (= communication NOT hidden under compute)
                           "bucket_ms_in_step": comm_timing["bucket_ms"] if comm_timing else None,
                           "exposed_ms_per_step": comm_timing["exposed_ms_per_step"] if comm_timing else None,
                           "comm_only": comm_only_res}
        if world == 1 and not args.no_cpu_baseline:
            log("timing the CPU oracle (bounded sample)")
            out["cpu_baseline"] = cpu_baseline()
        # compact digest of every timed leg as the LAST key: a reader who only keeps the tail of this line still sees them
        legs = {"value": [out["value"], out["ms_per_step"]]}
        for k_, v_ in (alt_algo or {}).items():
            legs[k_] = [v_["value"], v_["ms_per_step"]]
        for k_, v_ in (alt or {}).items():
            legs[k_] = [v_["value"], v_["ms_per_step"]]
        for k_, v_ in (others or {}).items():
            legs[k_] = [v_["value"], v_["ms_per_step"]]
        out["legs_clips_per_s_and_ms"] = legs
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line must be the LAST line on stdout: RCCL / HIP print banners through C stdio, whose buffer would
        # otherwise be flushed at exit, after Python's own output
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
