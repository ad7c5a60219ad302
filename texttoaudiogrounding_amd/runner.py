"""Training-step driver for the strongly-supervised path -- the counterpart of
Runner.forward / Runner.train_epoch in python_scripts/training/run_strong.py:92-152.

One process per GPU.  Parameters live in ONE flat fp32 buffer (and gradients in another) so that clip_grad_norm_ +
Adam are two kernels over the flat buffers (tag_sumsq, tag_adam_step) instead of ~40 small launches each, and the
data-parallel exchange is a handful of large all-reduces over contiguous slices of the flat gradient (RCCL over xGMI),
launched from inside backward as soon as a slice is final so that they overlap the remaining dgrad / wgrad kernels.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import ops
from .losses import FrameBceLoss
from .utils.train_util import init_obj_from_str


def build_model(model_cfg: Dict):
    """get_model of run_strong.py:71-89: instantiate sub-modules first, then the composition."""
    kwargs = {}
    for k in ("audio_encoder", "text_encoder", "match_fn"):
        if k in model_cfg:
            kwargs[k] = init_obj_from_str(model_cfg[k])
    return init_obj_from_str({"type": model_cfg["type"], "args": model_cfg.get("args", {})}, **kwargs)


class FlatParams:
    """Re-homes every trainable parameter (and its .grad) into one contiguous buffer each.

    ``p._tag_grad_sink`` is the parameter's view of the flat gradient: inside ``direct_grads()`` the HIP autograd nodes
    write gradients there themselves (no ``grad += new`` kernels).  ``model.zero_grad()`` / ``optimizer.zero_grad()``
    (set_to_none) detach ``p.grad`` from the flat buffer: ``zero_grad()`` re-attaches such parameters, and ``check()``
    (called at the entry of StrongRunner.forward_backward and by optimizer_step) raises when a parameter or a gradient was
    re-homed elsewhere instead of silently writing into / stepping on memory nobody reads."""

    def __init__(self, model: torch.nn.Module):
        self.params = self._ordered([p for p in model.parameters() if p.requires_grad], model)
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.empty(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.offsets: List[int] = []
        off = 0
        for p in self.params:
            k = p.numel()
            self.offsets.append(off)
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.grad[off:off + k].view_as(p)
            p._tag_grad_sink = p.grad
            if getattr(p, "_tag_guard_hook", None) is None:       # one hook per parameter however often a runner is rebuilt
                p._tag_guard_hook = p.register_hook(ops.second_writer_guard(p))
            off += k
        self.numel = n

    @staticmethod
    def _ordered(params, model):
        """model.parameters() order, except that the two directions of every one-layer bidirectional nn.GRU are interleaved
        (w_ih, w_ih_reverse, w_hh, w_hh_reverse, b_ih, b_ih_reverse, b_hh, b_hh_reverse): the HIP GRU takes each pair as one
        tensor, and adjacent storage makes that a view instead of a torch.cat per step (ops._joined)."""
        pos = {id(p): i for i, p in enumerate(params)}
        for m in model.modules():
            if isinstance(m, torch.nn.GRU) and m.bidirectional and m.num_layers == 1:
                names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
                group = [getattr(m, n + sfx) for n in names for sfx in ("", "_reverse")]
                if all(id(t) in pos for t in group):
                    first = min(pos[id(t)] for t in group)
                    rest = [p for p in params if all(p is not t for t in group)]
                    params = rest[:first] + group + rest[first:]
                    pos = {id(p): i for i, p in enumerate(params)}
        return params

    def zero_grad(self):
        """Zero the flat gradient; parameters whose ``.grad`` was dropped (``model.zero_grad()`` /
        ``optimizer.zero_grad(set_to_none=True)`` between steps) get their flat view back -- nothing is lost, the
        gradients are being zeroed anyway."""
        self.grad.zero_()
        for p, off in zip(self.params, self.offsets):
            if p.grad is None:
                p.grad = self.grad[off:off + p.numel()].view_as(p)
                p._tag_grad_sink = p.grad

    def check(self):
        base, esz = self.grad.data_ptr(), self.grad.element_size()
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != base + off * esz or p.data.data_ptr() != self.flat.data_ptr() + off * esz:
                raise RuntimeError("a parameter or its .grad no longer aliases the flat buffers (p.grad = <other tensor> / "
                                   "p.data = ... / .to() after FlatParams was built); rebuild the runner, or clear gradients "
                                   "with FlatParams.zero_grad() / set_to_none only")


class GradBuckets:
    """Bucketed data-parallel gradient exchange over the flat gradient buffer.

    Buckets are contiguous slices of ``flat.grad`` holding whole parameters, in parameter order, cut (walking from the
    last parameter to the first) whenever a slice reaches ``bucket_bytes``.  The autograd nodes call ``ready(params)`` when the kernels producing those gradients are
    enqueued and ``flush()`` at points where a collective may start; ``flush`` launches ONE asynchronous all-reduce(sum)
    per complete bucket on a communication stream that waits for the compute stream (and the wgrad side stream) as of
    that moment, so the exchange of conv_block4 runs beside the backward of conv_block3...1.  ``finish()`` (after
    loss.backward()) launches whatever is left and makes the compute stream wait for all of them.  The mean (1/world) is
    folded into the Adam kernel; with equal per-rank batch sizes the result is the gradient of the mean over ranks of
    the per-replica mean losses (= the reference's single-device loss on the concatenated batch when all clips have the
    same number of valid frames; otherwise each replica's frames are weighted by its own mask count -- DESIGN.md section 5).
    Device-agnostic: on CPU tensors (gloo, tests/test_dp_gloo.py) the same code runs without streams."""

    def __init__(self, flat: FlatParams, bucket_bytes: int = 8 << 20, group=None, comm_dtype=None):
        self.flat, self.group = flat, group
        # comm_dtype=torch.bfloat16 (BASELINE configs[2]): the payload on the wire is a bf16 copy of the bucket (17.6 MB
        # instead of 35.2 MB per step), summed in bf16 by the collective and widened back into the fp32 flat gradient
        self.comm_dtype = comm_dtype
        self.cuda = flat.grad.is_cuda
        self.bounds: List[tuple] = []          # (start element, end element, first param index, last param index + 1)
        # cut walking the parameters from the LAST to the first: backward produces gradients roughly in reverse parameter
        # order, so the bucket that completes last (the first conv blocks: tiny tensors) is the small remainder and the
        # exposed tail of the exchange is ~1 MB instead of a full bucket
        n = len(flat.params)
        end_i, acc = n, 0
        for i in range(n - 1, -1, -1):
            acc += flat.params[i].numel() * 4
            if acc >= bucket_bytes or i == 0:
                self.bounds.append((flat.offsets[i], flat.offsets[end_i - 1] + flat.params[end_i - 1].numel(), i, end_i))
                end_i, acc = i, 0
        self.bounds.reverse()
        self.bucket_of = {}
        for b, (_, _, i0, i1) in enumerate(self.bounds):
            for i in range(i0, i1):
                self.bucket_of[id(flat.params[i])] = b
        self.comm_stream = torch.cuda.Stream(device=flat.grad.device) if self.cuda else None
        self.stream_wait = self.cuda and dist.get_backend(group) == "nccl"
        self.record = False                    # bench.py: per-bucket events on the communication stream
        self.events, self.exposed = [], []
        self.reset()

    def reset(self):
        self.missing = [i1 - i0 for (_, _, i0, i1) in self.bounds]
        self.seen = set()
        self.launched = [False] * len(self.bounds)
        self.works = []

    def ready(self, params):
        for p in params:
            b = self.bucket_of.get(id(p))
            if b is None or id(p) in self.seen:
                continue
            self.seen.add(id(p))
            self.missing[b] -= 1

    def _launch(self, b):
        s, e, _, _ = self.bounds[b]
        buf = self.flat.grad[s:e]
        wire = buf
        if self.cuda:
            main = torch.cuda.current_stream(buf.device)
            self.comm_stream.wait_stream(main)
            for side in ops.side_streams(buf.device):
                self.comm_stream.wait_stream(side)
            ev = None
            with torch.cuda.stream(self.comm_stream):
                if self.record:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record(self.comm_stream)
                if self.comm_dtype is not None:
                    wire = buf.to(self.comm_dtype)
                w = dist.all_reduce(wire, group=self.group, async_op=True)
                if self.stream_wait:
                    # RCCL: Work.wait() orders the CURRENT stream (= the communication stream) after the collective without
                    # blocking the host, so the widening copy of a bf16 payload and the end-of-bucket event also live on
                    # the communication stream, off the compute stream's critical path
                    w.wait()
                    if wire is not buf:
                        buf.copy_(wire)
                        wire = buf
                    w = None
                    if ev is not None:
                        ev[1].record(self.comm_stream)
                        self.events.append((b, ev[0], ev[1]))
                elif wire is not buf:
                    wire.record_stream(main)             # widened back on the compute stream in finish()
        else:
            if self.comm_dtype is not None:
                wire = buf.to(self.comm_dtype)
            w = dist.all_reduce(wire, group=self.group, async_op=True)
        self.works.append((w, buf, wire))
        self.launched[b] = True

    def flush(self):
        for b in range(len(self.bounds)):
            if not self.launched[b] and self.missing[b] <= 0:
                self._launch(b)

    def finish(self):
        """Everything not yet exchanged goes now (parameters no node announced, e.g. unused ones whose gradient is the
        zero fill); then the compute stream waits for every collective.  With ``record`` on, the wait is bracketed by two
        events on the compute stream: their distance is the EXPOSED communication time of the step."""
        for b in range(len(self.bounds)):
            if not self.launched[b]:
                self._launch(b)
        for w, _, _ in self.works:
            if w is not None:
                w.wait()
        if self.cuda:
            main = torch.cuda.current_stream(self.flat.grad.device)
            if self.record:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main)
            main.wait_stream(self.comm_stream)
            if self.record:
                e1.record(main)
                self.exposed.append((e0, e1))
        for _, buf, wire in self.works:
            if wire is not buf:
                buf.copy_(wire)
        self.works = []

    def timing_summary(self):
        """After a synchronize: {"bucket_ms": [mean duration of each bucket's all-reduce on the communication stream],
        "exposed_ms_per_step": mean time the compute stream spent blocked on the communication stream}; clears the log."""
        per = [[] for _ in self.bounds]
        for b, e0, e1 in self.events:
            per[b].append(e0.elapsed_time(e1))
        exp = [e0.elapsed_time(e1) for e0, e1 in self.exposed]
        self.events, self.exposed = [], []
        return {"bucket_ms": [round(sum(v) / len(v), 4) if v else None for v in per],
                "exposed_ms_per_step": round(sum(exp) / len(exp), 4) if exp else None}


class StrongRunner:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0, device="cuda",
                 bucket_bytes: int = 8 << 20, overlap_comm: bool = True, grad_comm_dtype=None, force_comm: bool = False):
        self.device = torch.device(device)
        self.model = model.to(self.device)
        self.loss_fn = FrameBceLoss()
        self.flat = FlatParams(self.model)
        self.m = torch.zeros_like(self.flat.flat)
        self.v = torch.zeros_like(self.flat.flat)
        self.lr, self.betas, self.eps, self.max_grad_norm = lr, betas, eps, max_grad_norm
        self.step_count = 0
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if on else 1
        self.rank = dist.get_rank() if on else 0
        self.overlap_comm = overlap_comm
        self.buckets: Optional[GradBuckets] = None
        if self.world > 1 or (force_comm and on):
            # (force_comm: a 1-rank group still runs the whole bucket / stream machinery -- tests, bench --comm-only)
            # replicas must START identical whatever each rank's seed / checkpoint was: parameters and every buffer
            # (BatchNorm running statistics, num_batches_tracked) come from rank 0; Adam moments start at zero everywhere
            dist.broadcast(self.flat.flat, 0)
            for b in self.model.buffers():
                dist.broadcast(b, 0)
            self.buckets = GradBuckets(self.flat, bucket_bytes, comm_dtype=grad_comm_dtype)
        # ranks seeded alike still need different dropout masks for their different clips
        ops.SEED_RANK = self.rank

    # Runner.forward (run_strong.py:92-120)
    def forward(self, batch: Dict, training: bool = True):
        for k, v in batch.items():
            if k in ("waveform_len", "text_len"):
                # small integer arrays: staged through pinned memory so that no copy blocks the host mid-step (a
                # pageable host->device copy waits for the stream to drain); everything downstream (frame lengths,
                # masks) then stays on the device
                if not (isinstance(v, torch.Tensor) and v.is_cuda):
                    v = torch.as_tensor(v).long()
                    v = (v.pin_memory() if self.device.type == "cuda" else v).to(self.device, non_blocking=True)
                batch[k] = v.long()
            elif isinstance(v, torch.Tensor):
                batch[k] = v.long().to(self.device) if k == "text" else v.float().to(self.device)
        input_dict = {"specaug": False}
        input_dict.update(batch)
        output = self.model(input_dict)
        if training:
            label = batch["label"]
            frame_sim = output["frame_sim"]
            tt = min(frame_sim.size(1), label.size(1))
            output.update({"frame_sim": frame_sim[..., :tt], "label": label[..., :tt],
                           "length": torch.clamp(output["length"], 1, tt)})
        return output

    def forward_backward(self, batch: Dict):
        """zero_grad -> forward -> FrameBceLoss -> backward [with the bucketed gradient all-reduce inside it]."""
        self.flat.zero_grad()                    # also re-attaches gradients dropped by model.zero_grad()
        self.flat.check()                        # fail fast: the kernels are about to write into the flat views
        prev = (ops.DIRECT_GRADS, ops.GRAD_READY, ops.GRAD_FLUSH)
        ops.DIRECT_GRADS = True
        ops.begin_direct_step()
        if self.buckets is not None:
            self.buckets.reset()
            ops.GRAD_READY = self.buckets.ready
            ops.GRAD_FLUSH = self.buckets.flush if self.overlap_comm else None
        try:
            output = self.forward(batch, training=True)
            loss = self.loss_fn(output)
            loss.backward()
        finally:
            ops.DIRECT_GRADS, ops.GRAD_READY, ops.GRAD_FLUSH = prev
        if self.buckets is not None:
            self.buckets.finish()                # RCCL sums; the mean is folded into the Adam kernel
        return loss

    def optimizer_step(self):
        """clip_grad_norm_(max_grad_norm) + Adam (run_strong.py:143-145) on the flat buffers."""
        self.flat.check()
        self.step_count += 1
        gsq = ops.grad_sumsq(self.flat.grad)
        ops.adam_step(self.flat.flat, self.flat.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps,
                      self.step_count, gsq, self.max_grad_norm or 0.0, 1.0 / self.world)

    def train_step(self, batch: Dict):
        self.model.train()
        loss = self.forward_backward(batch)
        self.optimizer_step()
        return loss

    def loss_value(self, loss) -> float:
        """loss.item() + the deferred device-side error checks (GRU exchange timeout, embedding ids out of range)."""
        v = float(loss.item())
        ops.check_async_errors()
        return v


def init_distributed(backend: Optional[str] = None):
    """One process per GPU, torch.distributed over RCCL (backend 'nccl' on ROCm).  TAG_DIST_BACKEND=gloo and
    TAG_SHARE_GPU=1 are test hooks: N ranks on ONE GPU exchanging through gloo (how the N > 1 path is exercised on a
    1-GPU box; RCCL refuses two ranks on one device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = backend or os.environ.get("TAG_DIST_BACKEND", "nccl")
    if os.environ.get("TAG_SHARE_GPU", "0") == "1":
        local = 0
    torch.cuda.set_device(local)
    if not dist.is_initialized():
        # device_id binds the communicator to THIS rank's GPU at init time (eager RCCL initialisation): without it the first
        # collective / barrier picks "the device under current context" lazily -- the classic wrong-device hang of a first
        # N > 1 run -- and torch warns about it on every barrier
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def barrier(local: Optional[int] = None):
    """dist.barrier() naming the rank's device to RCCL (no lazy device guess); plain barrier on gloo / without a group."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    if dist.get_backend() == "nccl":
        dist.barrier(device_ids=[torch.cuda.current_device() if local is None else local])
    else:
        dist.barrier()


def comm_environment() -> Dict:
    """What a scaling record needs to be read later (rank 0 calls it): RCCL version, the NCCL_* / RCCL_* / HSA_* settings in
    force, and the GPU-to-GPU link types `rocm-smi --showtopo` reports (XGMI vs PCIE), when the tool is there."""
    import subprocess
    info: Dict = {"env": {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_", "HSA_ENABLE_IPC"))}}
    try:
        info["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception as e:                                    # noqa: BLE001 - diagnostic only
        info["rccl_version"] = f"unavailable ({type(e).__name__})"
    try:
        out = subprocess.run(["rocm-smi", "--showtopotype"], capture_output=True, text=True, timeout=20).stdout
        rows = [ln.split() for ln in out.splitlines() if ln.startswith("GPU") and len(ln.split()) > 1]
        kinds = sorted({c for r in rows for c in r[1:] if c.isalpha()})
        info["link_types"] = kinds or None
        info["gpus_in_topology"] = len(rows) or None
    except Exception as e:                                    # noqa: BLE001
        info["link_types"] = f"rocm-smi unavailable ({type(e).__name__})"
    return info
