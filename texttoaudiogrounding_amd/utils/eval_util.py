"""Post-processing of frame probabilities into integer segments (row P1; mirror of
utils/eval_util.py:18-116 in the reference as it is driven by run_strong.py:203-252).

Two layers:

* ``segments_for_thresholds`` runs binarize (strict >, float64 compare) -> median filter ->
  connect_clusters -> find_contiguous_regions for every (clip, threshold) pair in ONE HIP launch
  (``tag_segments``) and returns the reference's rows ``[onset_idx, offset_idx)``.
* the reference's own function names -- ``find_contiguous_regions``, ``binarize``,
  ``median_filter``, ``connect_clusters`` / ``connect_clusters_`` / ``connect_``,
  ``predictions_to_time`` -- with the reference's signatures, so that its evaluate loop
  (run_strong.py:234-252) runs unchanged after ``install_aliases()``.  They are host-side numpy
  bookkeeping over a few hundred frames per call, like the reference's; both layers are pinned by
  the same fixture (tests/golden/postproc.npz, generated from the imported reference).
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
    ops.check_async_errors()          # the copies above synchronised anyway: surface a GRU exchange timeout / a bad token id here
    B, NT = counts.shape
    return [[regions[b, t, :counts[b, t]].copy() for t in range(NT)] for b in range(B)]


# ------------------------------------------------------------------------------------------------
# the reference's per-stage functions (utils/eval_util.py:18-116)
# ------------------------------------------------------------------------------------------------

def _as_numpy(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def find_contiguous_regions(activity_array) -> np.ndarray:
    """utils/eval_util.py:18-44: 0/1 (or bool) vector -> (K, 2) rows ``[onset, offset)``."""
    a = _as_numpy(activity_array).astype(bool).ravel()
    edges = np.diff(np.concatenate(([0], a.astype(np.int8), [0])))
    return np.stack([np.flatnonzero(edges == 1), np.flatnonzero(edges == -1)], axis=1)


def binarize(x, threshold=0.5):
    """utils/eval_util.py:47-52 (sklearn.preprocessing.binarize): strict ``x > threshold`` -- compared in float64, as numpy
    promotes the float32 scores against the np.float64 thresholds of run_strong.py:203-205 -- returned as 0/1 in x's dtype."""
    x = _as_numpy(x)
    if x.ndim not in (2, 3):
        raise ValueError(f"binarize expects a 2-D or 3-D array, got {x.ndim}-D")          # sklearn's check_array does too
    return (x.astype(np.float64) > np.float64(threshold)).astype(x.dtype)


def _median_along(b: np.ndarray, window_size: int, axis: int) -> np.ndarray:
    """scipy.ndimage.median_filter(size = window_size along ``axis``, 1 elsewhere, mode='reflect', origin 0): window
    [i - w//2, i - w//2 + w - 1], boundary (d c b a | a b c d | d c b a), element of rank w//2."""
    if window_size <= 1:
        return b.copy()
    b = np.moveaxis(b, axis, -1)
    T = b.shape[-1]
    left = window_size // 2
    idx = np.mod(np.arange(-left, T + window_size - left - 1), 2 * T)
    idx = np.where(idx >= T, 2 * T - 1 - idx, idx)
    win = np.lib.stride_tricks.sliding_window_view(b[..., idx], window_size, axis=-1)
    out = np.partition(win, window_size // 2, axis=-1)[..., window_size // 2]
    return np.moveaxis(out, -1, axis).astype(b.dtype)


def median_filter(x, window_size, threshold=0.5):
    """utils/eval_util.py:55-63: binarize, then a median filter along the TIME axis, which the reference picks by shape:
    (batch, time, classes) -> axis 1; (1, time) -> axis 1; (time, classes) with more than one row -> axis 0."""
    x = binarize(x, threshold=threshold)
    if x.ndim == 3 or (x.ndim == 2 and x.shape[0] == 1):
        axis = 1
    else:
        axis = 0
    return _median_along(x, int(window_size), axis)


def predictions_to_time(df, ratio):
    """utils/eval_util.py:66-71 for a DataFrame with onset / offset columns (scaled in place, returned); an integer region
    array (K,2) is scaled to float64 seconds."""
    if hasattr(df, "onset") and hasattr(df, "offset"):
        if len(df) == 0:
            return df
        df.onset = df.onset * ratio
        df.offset = df.offset * ratio
        return df
    return np.asarray(df).astype(np.float64) * ratio


def connect_(pairs, n=1):
    """utils/eval_util.py:97-116: merge neighbouring ``(onset, offset)`` clusters whose gap ``next.onset - cur.offset`` is
    <= n; returns a list of (onset, offset) tuples ([] for no clusters)."""
    pairs = [(int(p[0]), int(p[1])) for p in pairs]
    if not pairs:
        return []
    merged = [list(pairs[0])]
    for on, off in pairs[1:]:
        if on - merged[-1][1] <= n:
            merged[-1][1] = off
        else:
            merged.append([on, off])
    return [(a, b) for a, b in merged]


def connect_clusters_(x, n=1):
    """utils/eval_util.py:81-94: 1-D 0/1 vector -> 0/1 int vector with gaps <= n frames between clusters filled."""
    x = _as_numpy(x)
    assert x.ndim == 1, "input needs to be 1d"
    out = np.zeros_like(x, dtype=int)
    for on, off in connect_(find_contiguous_regions(x), n=n):
        out[on:off] = 1
    return out


def connect_clusters(x, n=1):
    """utils/eval_util.py:74-78: 1-D input directly; N-D input along axis -2 (the reference's np.apply_along_axis(..., -2, x))."""
    x = _as_numpy(x)
    if x.ndim == 1:
        return connect_clusters_(x, n)
    return np.apply_along_axis(lambda a: connect_clusters_(a, n=n), -2, x)
