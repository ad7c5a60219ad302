#!/usr/bin/env python3
"""tests/golden/grounding_eval.npz: precision / recall of every operating point and the threshold-AUC computed by the
REFERENCE's Grounding_PrecisionRecall (/root/reference/utils/eval_util.py:431-663, imported with the psds_eval base class
stubbed -- the class only inherits input validation from it) on seeded ground-truth / detection tables.

pandas >= 2 removed DataFrame.append, which the reference's bookkeeping (_add_op) calls; the arithmetic under test
(_ground_truth_intersections, _recall_criteria, _precision_criteria, th_auc) is called directly and the operating-point
table is assembled with pd.concat -- no line of the reference's arithmetic is replaced.  Build container only."""
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

mods = ref_import.install()
EU = mods["utils.eval_util"]
EU.Grounding_PrecisionRecall._validate_simple_dataframe = lambda self, *a, **k: None      # inherited validation only

rng = np.random.default_rng(12)
files = [f"clip{i}_{j}" for i in range(5) for j in range(2)]


def random_table(n_per_file, jitter):
    rows = []
    for f in files:
        k = rng.integers(0, n_per_file + 1)
        on = np.sort(rng.uniform(0, 9, k))
        for o in on:
            rows.append({"filename": f, "onset": round(float(o), 2), "offset": round(float(o + rng.uniform(0.2, jitter)), 2)})
    return pd.DataFrame(rows, columns=["filename", "onset", "offset"])


gt = random_table(3, 3.0)
out = {"gt_file": np.array(gt.filename.values, dtype="U16"), "gt": gt[["onset", "offset"]].values}
cases = {}
for dtc, gtc in ((0.5, 0.5), (0.3, 0.7), (0.0, 1.0)):
    ev = EU.Grounding_PrecisionRecall(dtc, gtc, gt.copy())
    ths = np.round(np.arange(0.05, 1.0, 0.1), 2)
    ops = []
    for i, th in enumerate(ths):
        # detections: the ground truth perturbed more and more, plus spurious ones; operating point 4 repeats point 3
        if i == 4:
            det = dets_prev.copy()
        else:
            det = gt.copy()
            det["onset"] = np.round(det.onset + rng.normal(0, 0.1 + 0.15 * i, len(det)), 2)
            det["offset"] = np.round(np.maximum(det.onset + 0.1, det.offset + rng.normal(0, 0.1 + 0.15 * i, len(det))), 2)
            det = det[rng.uniform(size=len(det)) > 0.08 * i]
            det = pd.concat([det, random_table(1, 1.5)], ignore_index=True)
        dets_prev = det
        det_t = ev._init_det_table(det.reset_index(drop=True))
        precision, recall = ev._evaluate_detections(det_t)
        ops.append({"id": str(i), "precision": precision, "recall": recall, "threshold": float(th)})
        out[f"det{dtc}_{gtc}/{i}/file"] = np.array(det.filename.values, dtype="U16")
        out[f"det{dtc}_{gtc}/{i}/rows"] = det[["onset", "offset"]].values
    ev.operating_points = pd.DataFrame(ops)
    auc_full = ev.th_auc(beta=1.0, low_th=0.0, high_th=1.0)
    auc_sub = ev.th_auc(beta=2.0, low_th=0.2, high_th=0.8)
    out[f"pr{dtc}_{gtc}"] = np.array([[o["precision"], o["recall"]] for o in ops])
    out[f"thauc{dtc}_{gtc}"] = np.array([auc_full, auc_sub])
    out[f"ths{dtc}_{gtc}"] = ths
    print(f"dtc {dtc} gtc {gtc}: th_auc {auc_full:.6f} / {auc_sub:.6f}; P/R at op 0 {ops[0]['precision']:.3f}/{ops[0]['recall']:.3f}, "
          f"op 9 {ops[-1]['precision']:.3f}/{ops[-1]['recall']:.3f}")
np.savez_compressed(os.path.join(HERE, "grounding_eval.npz"), **out)
print("wrote grounding_eval.npz", os.path.getsize(os.path.join(HERE, "grounding_eval.npz")))
