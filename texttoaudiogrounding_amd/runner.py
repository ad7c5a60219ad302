"""Training-step driver for the strongly-supervised path -- the counterpart of
Runner.forward / Runner.train_epoch in python_scripts/training/run_strong.py:92-152.

One process per GPU.  Parameters live in ONE flat fp32 buffer (and gradients in another) so that
the data-parallel exchange is a single RCCL all-reduce over xGMI and clip_grad_norm_ + Adam are two
kernels over the flat buffers (tag_sumsq, tag_adam_step) instead of ~40 small launches each.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

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
    """Re-homes every trainable parameter (and its .grad) into one contiguous buffer each."""

    def __init__(self, model: torch.nn.Module):
        self.params = [p for p in model.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.empty(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.grad[off:off + k].view_as(p)
            off += k
        self.numel = n

    def zero_grad(self):
        self.grad.zero_()


class StrongRunner:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0, device="cuda"):
        self.device = torch.device(device)
        self.model = model.to(self.device)
        self.loss_fn = FrameBceLoss()
        self.flat = FlatParams(self.model)
        self.m = torch.zeros_like(self.flat.flat)
        self.v = torch.zeros_like(self.flat.flat)
        self.lr, self.betas, self.eps, self.max_grad_norm = lr, betas, eps, max_grad_norm
        self.step_count = 0
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

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
        """zero_grad -> forward -> FrameBceLoss -> backward [-> gradient all-reduce]."""
        self.flat.zero_grad()
        output = self.forward(batch, training=True)
        loss = self.loss_fn(output)
        loss.backward()
        if self.world > 1:
            dist.all_reduce(self.flat.grad)          # RCCL sum; the mean is folded into the Adam kernel
        return loss

    def optimizer_step(self):
        """clip_grad_norm_(max_grad_norm) + Adam (run_strong.py:143-145) on the flat buffers."""
        self.step_count += 1
        gsq = ops.grad_sumsq(self.flat.grad)
        ops.adam_step(self.flat.flat, self.flat.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps,
                      self.step_count, gsq, self.max_grad_norm or 0.0, 1.0 / self.world)

    def train_step(self, batch: Dict):
        self.model.train()
        loss = self.forward_backward(batch)
        self.optimizer_step()
        return loss


def init_distributed():
    """One process per GPU, torch.distributed over RCCL (backend 'nccl' on ROCm)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=rank, world_size=world)
    return rank, world, local
