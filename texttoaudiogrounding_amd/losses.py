"""FrameBceLoss behind the reference interface (losses.py:11-35 in the reference)."""
from typing import Dict

import torch
import torch.nn as nn

from . import ops


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
        return ops.FrameBceFunction.apply(frame_sim, label.to(frame_sim.device).float(), length, Tt)
