"""Host-side evaluators over the device segments, CPU tests.

* GroundingPrecisionRecall / th_auc: against tests/golden/grounding_eval.npz -- per-operating-point precision / recall and
  the threshold-AUC computed by the IMPORTED reference class (utils/eval_util.py:431-663) on seeded tables.
* psds_intersection: hand-derivable known answers (psds_eval / sed_scores_eval are absent third-party packages: parity with
  them is unpinned, stated in texttoaudiogrounding_amd/utils/grounding_eval.py)."""
import numpy as np
import pytest

from texttoaudiogrounding_amd.utils import grounding_eval as GE


def table(files, rows):
    t = {}
    for f, r in zip(files, rows):
        t.setdefault(str(f), []).append(r)
    return {f: np.array(v, dtype=np.float64) for f, v in t.items()}


@pytest.mark.parametrize("dtc,gtc", [(0.5, 0.5), (0.3, 0.7), (0.0, 1.0)])
def test_precision_recall_and_th_auc_vs_reference(golden_dir, dtc, gtc):
    gold = np.load(f"{golden_dir}/grounding_eval.npz")
    gt = table(gold["gt_file"], gold["gt"])
    ev = GE.GroundingPrecisionRecall(dtc, gtc, gt)
    ths = gold[f"ths{dtc}_{gtc}"]
    for i, th in enumerate(ths):
        det = table(gold[f"det{dtc}_{gtc}/{i}/file"], gold[f"det{dtc}_{gtc}/{i}/rows"])
        p, r = GE.evaluate_detections(det, gt, dtc, gtc)
        assert abs(p - gold[f"pr{dtc}_{gtc}"][i, 0]) < 1e-12 and abs(r - gold[f"pr{dtc}_{gtc}"][i, 1]) < 1e-12, (i, p, r)
        ev.add_operating_point(det, float(th))
    # operating point 4 repeats the detection table of point 3: the class copies its last row (reference quirk, :530-537)
    assert ev.operating_points[4]["precision"] == ev.operating_points[3]["precision"]
    assert abs(ev.th_auc(1.0, 0.0, 1.0) - gold[f"thauc{dtc}_{gtc}"][0]) < 1e-12
    assert abs(ev.th_auc(2.0, 0.2, 0.8) - gold[f"thauc{dtc}_{gtc}"][1]) < 1e-12


def test_precision_recall_known_answers():
    gt = {"a": np.array([[0.0, 4.0], [6.0, 8.0]]), "b": np.array([[1.0, 2.0]])}
    # a: one detection covering half of the first event exactly, one spurious; b: nothing detected
    det = {"a": np.array([[0.0, 2.0], [9.0, 9.5]])}
    p, r = GE.evaluate_detections(det, gt, 0.5, 0.5)
    assert p == 0.5 and r == pytest.approx(1 / 3)            # 1 of 2 detections, 1 of 3 events
    p, r = GE.evaluate_detections(det, gt, 0.5, 0.51)
    assert p == 0.0 and r == 0.0                             # coverage 0.5 < 0.51: neither criterion holds
    # two detections that only TOGETHER cover the event; each lies fully inside it
    det = {"a": np.array([[0.0, 1.5], [2.0, 3.0]])}
    p, r = GE.evaluate_detections(det, gt, 0.5, 0.5)
    assert p == 1.0 and r == pytest.approx(1 / 3)
    assert GE.evaluate_detections({}, gt, 0.5, 0.5) == (0.0, 0.0)


def test_psds_known_answers():
    """Three operating points over one hour of audio, 4 events.  By hand:
       strict : 1 event found, 0 false positives      -> (eFPR 0,  TPR 0.25)
       medium : 2 events found, 3 false positives     -> (3 /h,    0.50)
       loose  : 4 events found, 12 false positives    -> (12 /h,   1.00)
    staircase from the origin: max_efpr 12 -> (3*0.25 + 9*0.5) / 12 = 0.4375; max_efpr 6 -> (3*0.25 + 3*0.5) / 6 = 0.375;
    max_efpr 100 -> (0.75 + 4.5 + 88) / 100 = 0.9325."""
    gt = {f"f{i}": np.array([[10.0 * i, 10.0 * i + 4.0]]) for i in range(4)}
    dur = {f"f{i}": 900.0 for i in range(4)}

    def op(n_found, n_fp):
        t = {}
        for i in range(n_found):
            t[f"f{i}"] = [[10.0 * i + 0.5, 10.0 * i + 3.5]]                       # 75 % coverage, fully inside
        for k in range(n_fp):
            t.setdefault(f"f{k % 4}", []).append([100.0 + k, 100.5 + k])           # touches no event
        return {f: np.array(v) for f, v in t.items()}

    ops = {0.9: op(1, 0), 0.5: op(2, 3), 0.1: op(4, 12)}
    assert GE.psds_operating_point(ops[0.5], gt, 3600.0, 0.5, 0.5) == (0.5, 3.0)
    assert GE.psds_intersection(ops, gt, dur, max_efpr=12.0) == pytest.approx(0.4375)
    assert GE.psds_intersection(ops, gt, dur, max_efpr=None) == pytest.approx(0.4375)
    assert GE.psds_intersection(ops, gt, dur, max_efpr=6.0) == pytest.approx(0.375)
    assert GE.psds_intersection(ops, gt, dur, max_efpr=100.0) == pytest.approx(0.9325)
    # a detection that is mostly outside its event fails the DTC: it is a false positive AND does not count for the event
    bad = {"f0": np.array([[3.0, 9.0]])}                                          # 1 s of 6 s inside the event
    assert GE.psds_operating_point(bad, gt, 3600.0, 0.5, 0.5) == (0.0, 1.0)
    # a perfect system: PSDS 1
    perfect = {0.5: {f: v.copy() for f, v in gt.items()}}
    assert GE.psds_intersection(perfect, gt, dur, max_efpr=100.0) == pytest.approx(1.0)


def test_tables_from_segments():
    segs = [[np.array([[0, 5], [10, 12]]), np.zeros((0, 2), dtype=np.int64)], [np.array([[3, 4]]), np.array([[3, 4]])]]
    t = GE.tables_from_segments(segs, ["x_0", "y_3"], [0.25, 0.75], 0.04)
    assert np.allclose(t[0.25]["x_0"], [[0.0, 0.2], [0.4, 0.48]]) and t[0.75]["x_0"].shape == (0, 2)
    assert np.allclose(t[0.75]["y_3"], [[0.12, 0.16]])
