"""Masks and initialisers used on the hot path (mirror of the reference's models/utils.py:5-58)."""
import torch
import torch.nn as nn


def init_weights(m):
    """Same per-layer-type initialisation policy as the reference (models/utils.py:5-20)."""
    if isinstance(m, (nn.Conv2d, nn.Conv1d)):
        nn.init.kaiming_normal_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.BatchNorm2d):
        nn.init.constant_(m.weight, 1)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.Linear):
        nn.init.kaiming_uniform_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.Embedding):
        nn.init.kaiming_uniform_(m.weight)


def generate_length_mask(lens, max_length=None):
    """(N, max_length) bool, True where position < len (models/utils.py:22-30)."""
    lens = torch.as_tensor(lens)
    if max_length is None:
        max_length = int(lens.max().item())
    pos = torch.arange(max_length, device=lens.device)
    return pos.unsqueeze(0) < lens.view(-1, 1)
