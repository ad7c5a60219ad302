"""Evaluation of grounding segments: threshold-AUC (``Grounding_PrecisionRecall``) and intersection-based PSDS.

Host code (numpy) over the integer segments the device produces (``eval_util.segments_for_thresholds``); SURVEY.md section
8(f) rank 4.  Reference: utils/eval_util.py:431-663 (Grounding_PrecisionRecall, its own pandas code) and :136-225
(compute_psds -> psds_eval.PSDSEval with dtc = gtc = 0.5, cttc = 0, alpha_ct = alpha_st = 0), driven by
python_scripts/training/run_strong.py:170-275.

Pinning:
* ``GroundingPrecisionRecall`` restates the reference's OWN arithmetic (the pandas merges / group-bys of
  ``_ground_truth_intersections``, ``_recall_criteria``, ``_precision_criteria``, ``th_auc``) on arrays; it is pinned by
  tests/golden/grounding_eval.npz, produced by running the imported reference class on seeded tables
  (tests/golden/make_golden_eval.py).
* ``psds_intersection`` restates the published single-class intersection-based PSDS (Bilen et al., ICASSP 2020) as
  ``psds_eval`` applies it to operating-point tables.  psds_eval / sed_scores_eval are third-party packages that are neither
  vendored in the reference nor installed here: **parity with them is unpinned**; the function is checked on
  hand-derivable known answers only (tests/test_grounding_eval.py).

Tables are ``{filename: (K, 2) float array of [onset, offset] seconds}``; files without rows may be absent or empty.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

Table = Dict[str, np.ndarray]
EPS = 1e-15


def _rows(t: Table, f: str) -> np.ndarray:
    a = t.get(f)
    if a is None:
        return np.zeros((0, 2))
    return np.asarray(a, dtype=np.float64).reshape(-1, 2)


def _n_rows(t: Table) -> int:
    return int(sum(np.asarray(v).reshape(-1, 2).shape[0] for v in t.values()))


def _pair_ratios(det: np.ndarray, gt: np.ndarray):
    """For one file: crossing mask (D,G) and the two ratios of every (detection, ground truth) pair
    (utils/eval_util.py:543-560): a pair crosses when onset_det <= offset_gt and onset_gt <= offset_det;
    det_precision = intersection / detection duration, gt_coverage = intersection / ground-truth duration."""
    on_d, off_d = det[:, None, 0], det[:, None, 1]
    on_g, off_g = gt[None, :, 0], gt[None, :, 1]
    cross = (on_d <= off_g) & (on_g <= off_d)
    inter = np.minimum(off_d, off_g) - np.maximum(on_d, on_g)
    with np.errstate(divide="ignore", invalid="ignore"):
        det_prec = np.where(cross, inter / (off_d - on_d), 0.0)
        gt_cov = np.where(cross, inter / (off_g - on_g), 0.0)
    return cross, det_prec, gt_cov


def evaluate_detections(detections: Table, ground_truth: Table, dtc: float, gtc: float) -> Tuple[float, float]:
    """(precision, recall) of one operating point -- Grounding_PrecisionRecall._evaluate_detections (:631-641).

    recall: a ground truth counts when the detections that pass the DTC (sum of det_precision over the ground truths they
    cross >= dtc) cover it by >= gtc in total (:562-595).  precision: a detection counts when the ground truths that pass
    the GTC (total coverage by ALL detections >= gtc) account for >= dtc of it (:598-629)."""
    tp_refs = tp_preds = 0
    for f in set(detections) | set(ground_truth):
        det, gt = _rows(detections, f), _rows(ground_truth, f)
        if det.shape[0] == 0 or gt.shape[0] == 0:
            continue
        cross, det_prec, gt_cov = _pair_ratios(det, gt)
        # recall criterion
        dtc_pass = (det_prec.sum(1) >= dtc) & cross.any(1)
        cov_by_passing = (gt_cov * dtc_pass[:, None]).sum(0)
        tp_refs += int(((cov_by_passing >= gtc) & (cross & dtc_pass[:, None]).any(0)).sum())
        # precision criterion
        gtc_pass = (gt_cov.sum(0) >= gtc) & cross.any(0)
        prec_by_passing = (det_prec * gtc_pass[None, :]).sum(1)
        tp_preds += int(((prec_by_passing >= dtc) & (cross & gtc_pass[None, :]).any(1)).sum())
    n_refs, n_preds = _n_rows(ground_truth), _n_rows(detections)
    return tp_preds / max(n_preds, EPS), tp_refs / max(n_refs, EPS)


def _table_key(t: Table):
    rows = []
    for f in sorted(t):
        for on, off in _rows(t, f):
            rows.append((f, float(on), float(off)))
    return tuple(sorted(rows))


class GroundingPrecisionRecall:
    """Restatement of utils/eval_util.py:431-663 (Grounding_PrecisionRecall)."""

    def __init__(self, dtc_threshold: float, gtc_threshold: float, ground_truth: Table):
        if not 0.0 <= dtc_threshold <= 1.0:
            raise ValueError("dtc_threshold must be between 0 and 1")
        if not 0.0 <= gtc_threshold <= 1.0:
            raise ValueError("gtc_threshold must be between 0 and 1")
        if ground_truth is None:
            raise ValueError("The ground truth cannot be set without data")
        self.dtc, self.gtc = dtc_threshold, gtc_threshold
        self.ground_truth = {f: _rows(ground_truth, f) for f in ground_truth}
        self.operating_points: List[dict] = []
        self._seen = set()

    def add_operating_point(self, detections: Table, threshold: float):
        """:527-540.  A detection table identical to ANY earlier one is not re-evaluated: the reference appends a copy of the
        LAST row of its table with the new threshold (:530-537) -- restated as it is."""
        key = _table_key(detections)
        if key in self._seen and self.operating_points:
            row = dict(self.operating_points[-1])
            row["threshold"] = threshold
            self.operating_points.append(row)
            return
        self._seen.add(key)
        precision, recall = evaluate_detections(detections, self.ground_truth, self.dtc, self.gtc)
        self.operating_points.append({"precision": precision, "recall": recall, "threshold": threshold})

    def f_scores(self, beta: float = 1.0) -> np.ndarray:
        p = np.array([r["precision"] for r in self.operating_points])
        r = np.array([r["recall"] for r in self.operating_points])
        return (1 + beta ** 2) * p * r / np.maximum(beta ** 2 * p + r, EPS)

    def th_auc(self, beta: float = 1.0, low_th: float = 0.0, high_th: float = 1.0) -> float:
        """:643-655: trapezoidal area under F(threshold) over [low_th, high_th], divided by the interval length."""
        f = self.f_scores(beta)
        th = np.array([r["threshold"] for r in self.operating_points])
        keep = (th >= low_th) & (th <= high_th)
        th, f = th[keep], f[keep]
        order = np.argsort(th)
        th, f = th[order], f[order]
        area = float(np.sum((th[1:] - th[:-1]) * (f[1:] + f[:-1]) / 2.0))
        return area / (high_th - low_th)


def compute_th_auc(prediction_tables: Dict[float, Table], ground_truth: Table, dtc_threshold=0.5, gtc_threshold=0.5,
                   min_threshold=0.0, max_threshold=1.0, beta=1.0) -> float:
    """utils/eval_util.py:296-330 without the file output."""
    ev = GroundingPrecisionRecall(dtc_threshold, gtc_threshold, ground_truth)
    for th, det in prediction_tables.items():
        ev.add_operating_point(det, th)
    return ev.th_auc(beta=beta, low_th=min_threshold, high_th=max_threshold)


def psds_operating_point(detections: Table, ground_truth: Table, total_duration_s: float, dtc: float, gtc: float):
    """(TP ratio, false positives per hour) of one operating point, single class, no cross-triggers:
    a detection is RELEVANT when the ground truths it intersects account for >= dtc of it, otherwise it is a false
    positive; a ground truth is DETECTED when the relevant detections cover >= gtc of it."""
    tp = fp = 0
    for f in set(detections) | set(ground_truth):
        det, gt = _rows(detections, f), _rows(ground_truth, f)
        if det.shape[0] == 0:
            continue
        if gt.shape[0] == 0:
            fp += det.shape[0]
            continue
        _, det_prec, gt_cov = _pair_ratios(det, gt)
        relevant = det_prec.sum(1) >= dtc
        fp += int((~relevant).sum())
        tp += int(((gt_cov * relevant[:, None]).sum(0) >= gtc).sum())
    n_gt = _n_rows(ground_truth)
    return tp / max(n_gt, EPS), fp / (total_duration_s / 3600.0)


def psd_roc(points: Iterable[Tuple[float, float]]) -> Tuple[np.ndarray, np.ndarray]:
    """(eFPR, eTPR) staircase from (tpr, fpr) operating points: origin first, points sorted by eFPR, the best TPR at equal
    eFPR, made non-decreasing."""
    pts = sorted((round(fpr, 6), tpr) for tpr, fpr in points)
    xs, ys = [0.0], [0.0]
    for x, y in pts:
        if x == xs[-1]:
            ys[-1] = max(ys[-1], y)
        else:
            xs.append(x)
            ys.append(y)
    return np.array(xs), np.maximum.accumulate(np.array(ys))


def staircase_auc(x: np.ndarray, y: np.ndarray, max_x: Optional[float] = None) -> float:
    """Area under the right-continuous step function through (x, y) on [0, max_x] (the last level extends to max_x)."""
    if max_x is None:
        max_x = float(x.max())
    keep = x <= max_x
    xs = np.concatenate([x[keep], [max_x]])
    return float(np.sum(np.diff(xs) * y[keep]))


def psds_intersection(prediction_tables: Dict[float, Table], ground_truth: Table, durations: Dict[str, float],
                      dtc_threshold=0.5, gtc_threshold=0.5, max_efpr: Optional[float] = None) -> float:
    """Single-class intersection-based PSDS with alpha_ct = alpha_st = 0 over operating-point tables -- the computation
    utils/eval_util.py:136-225 delegates to psds_eval (parity with that package unpinned, see the module docstring).
    durations: seconds of audio per filename (the PSDSEval metadata table); max_efpr: per hour, None = the largest eFPR."""
    total = float(sum(durations[f] for f in ground_truth))
    pts = [psds_operating_point(det, ground_truth, total, dtc_threshold, gtc_threshold) for det in prediction_tables.values()]
    x, y = psd_roc(pts)
    if max_efpr is None:
        max_efpr = float(x.max())
    if max_efpr <= 0:
        return float(y.max())
    return staircase_auc(x, y, max_efpr) / max_efpr


def tables_from_segments(segments, filenames, thresholds, time_resolution: float) -> Dict[float, Table]:
    """eval_util.segments_for_thresholds output (list over clips of list over thresholds of (K,2) int64 rows) ->
    {threshold: {filename: (K,2) seconds}} -- what run_strong.py:226-265 accumulates in pred_buffer."""
    out: Dict[float, Table] = {float(th): {} for th in thresholds}
    for b, f in enumerate(filenames):
        for ti, th in enumerate(thresholds):
            out[float(th)][f] = np.asarray(segments[b][ti], dtype=np.float64) * time_resolution
    return out
