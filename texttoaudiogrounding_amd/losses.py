"""FrameBceLoss behind the reference interface (losses.py:11-35 in the reference)."""
from typing import Dict

import torch
import torch.nn as nn

from . import ops
from . import torch_ops  # noqa: F401  (registers torch.ops.tag.*)


class FrameBceLoss(nn.Module):
    def forward(self, output: Dict):
        frame_sim = output["frame_sim"]
        if frame_sim.ndim == 3 and frame_sim.size(2) == 1:
            frame_sim = frame_sim.squeeze(2)
        return self.forward_tensor(frame_sim, output["label"], output["length"])

    def forward_tensor(self, frame_sim, label, length):
        Tt = min(frame_sim.size(1), label.size(1))
        if frame_sim.size(1) != label.size(1):
            raise RuntimeError("frame_sim and label must be aligned first (Runner.forward does it)")
        length = torch.as_tensor(length).long().to(frame_sim.device).contiguous()
        return torch.ops.tag.frame_bce(frame_sim, label.to(frame_sim.device).float(), length, Tt)


class ClipBceLoss(nn.Module):
    """losses.py:38-43 in the reference: F.binary_cross_entropy(clip_sim, label) -- the mean over the (B,N) clip matrix,
    evaluated by the frame-BCE kernels with every row of full length."""

    def forward(self, output: Dict):
        return self.forward_tensor(output["clip_sim"], output["label"])

    def forward_tensor(self, prob, label):
        B, N = prob.shape
        length = torch.full((B,), N, dtype=torch.long, device=prob.device)
        return torch.ops.tag.frame_bce(prob.contiguous(), label.to(prob.device).float().contiguous(), length, N)


class MaxMarginRankingLoss(nn.Module):
    """losses.py:226-264 in the reference (triplet ranking over the (B,B) clip-vs-caption matrix)."""

    def __init__(self, margin=1, fix_norm=True, lamda1=1, sim_key="sim"):
        super().__init__()
        self.fix_norm, self.margin, self.lamda1, self.sim_key = fix_norm, margin, lamda1, sim_key

    def forward(self, x):
        return ops.MaxMarginFunction.apply(x[self.sim_key], self.margin, self.lamda1, self.fix_norm)


class ClipFrameBceLoss(nn.Module):
    """losses.py:186-210 in the reference: (1 - w) * ClipBceLoss(clip_sim, weak_label) + w * FrameBceLoss(frame_sim (B,T,N),
    strong_label (B,T,N), length).  The 3-D frame term equals the 2-D kernel over rows (b, n) of length[b] (the mask count
    is sum_b len_b * N either way), so the (B,T,N) view of MultiTextBiEncoder's (B*N,T) scores is used as it lies."""

    def __init__(self, frame_weight, clip_label_key="weak_label", clip_prob_key="clip_sim", frame_label_key="strong_label",
                 frame_prob_key="frame_sim"):
        super().__init__()
        self.clip_loss_fn, self.frame_loss_fn = ClipBceLoss(), FrameBceLoss()
        self.frame_weight = frame_weight
        self.clip_label_key, self.clip_prob_key = clip_label_key, clip_prob_key
        self.frame_label_key, self.frame_prob_key = frame_label_key, frame_prob_key

    def forward(self, output: Dict):
        fs, lab = output[self.frame_prob_key], output[self.frame_label_key]
        B, T, N = fs.shape
        fs2 = fs.transpose(1, 2).reshape(B * N, T)                      # a view for MultiTextBiEncoder's output
        lab2 = lab.to(fs.device).float().transpose(1, 2).reshape(B * N, T).contiguous()
        length = torch.as_tensor(output["length"]).long().to(fs.device).repeat_interleave(N).contiguous()
        frame = torch.ops.tag.frame_bce(fs2.contiguous(), lab2, length, T)
        clip = self.clip_loss_fn.forward_tensor(output[self.clip_prob_key], output[self.clip_label_key])
        return (1 - self.frame_weight) * clip + self.frame_weight * frame
