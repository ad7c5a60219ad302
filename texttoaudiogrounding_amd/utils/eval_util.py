"""Post-processing of frame probabilities into integer segments (row P1; mirror of
utils/eval_util.py:18-116 in the reference as it is driven by run_strong.py:203-252).

``segments_for_thresholds`` runs binarize (strict >, float64 compare) -> median filter ->
connect_clusters -> find_contiguous_regions for every (clip, threshold) pair in one HIP launch
and returns the reference's rows ``[onset_idx, offset_idx)``.
"""
import math

import numpy as np
import torch

from .. import ops


def eval_thresholds(n_thresholds: int = 50) -> np.ndarray:
    return np.arange(1 / (n_thresholds * 2), 1, 1 / n_thresholds)


def n_connect_for(time_resolution: float) -> int:
    return math.ceil(0.5 / time_resolution)


def segments_for_thresholds(frame_sim: torch.Tensor, thresholds, window_size: int, n_connect: int):
    """frame_sim (B,T) on the device -> list over clips of list over thresholds of (K,2) int64 arrays."""
    regions, counts = ops.segments(frame_sim, thresholds, window_size, n_connect)
    regions = regions.cpu().numpy()
    counts = counts.cpu().numpy()
    B, NT = counts.shape
    return [[regions[b, t, :counts[b, t]].copy() for t in range(NT)] for b in range(B)]


def predictions_to_time(regions: np.ndarray, ratio: float) -> np.ndarray:
    return regions.astype(np.float64) * ratio
